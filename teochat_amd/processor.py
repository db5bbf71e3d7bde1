"""Image preprocessing for the tower: host restatement (CPU, used by the parity tests and when no engine is bound)
and the on-device kernel (teo_preprocess_frames) used when an engine is bound.

Mirrors LanguageBindImageProcessor / get_image_transform of the reference
(languagebind/image/processing_image.py:7-8,15-25,33-68): ToTensor -> Resize(224, bicubic) ->
CenterCrop(224) -> Normalize(OPENAI mean/std).  torchvision is not available offline; the same arithmetic is
written with torch (tensor bicubic resize with antialias, as torchvision does for tensors).  For inputs that are
already 224x224 the resize and crop are identities.
"""
import numpy as np
import torch
import torch.nn.functional as F

OPENAI_DATASET_MEAN = (0.48145466, 0.4578275, 0.40821073)
OPENAI_DATASET_STD = (0.26862954, 0.26130258, 0.27577711)


def _to_chw_float(img):
    if isinstance(img, str):
        from PIL import Image
        img = Image.open(img).convert("RGB")
    if isinstance(img, torch.Tensor):
        if img.dtype == torch.uint8:                      # HWC uint8
            return img.permute(2, 0, 1).to(torch.float32) / 255.0
        return img.to(torch.float32)                      # already CHW float in [0,1]
    arr = np.asarray(img)
    if arr.ndim == 2:
        arr = arr[:, :, None]
    t = torch.from_numpy(np.array(arr)).permute(2, 0, 1)
    return t.to(torch.float32) / 255.0 if t.dtype == torch.uint8 else t.to(torch.float32)


def _to_hwc_uint8(img):
    """PIL image / path / uint8 array or tensor -> contiguous uint8 HWC torch tensor (RGB); None if it is float data."""
    if isinstance(img, str):
        from PIL import Image
        img = Image.open(img).convert("RGB")
    if isinstance(img, torch.Tensor):
        return img.contiguous() if (img.dtype == torch.uint8 and img.dim() == 3 and img.shape[-1] == 3) else None
    if hasattr(img, "convert") and getattr(img, "mode", "RGB") != "RGB":
        img = img.convert("RGB")
    arr = np.asarray(img)
    if arr.dtype != np.uint8 or arr.ndim != 3 or arr.shape[-1] != 3:
        return None
    return torch.from_numpy(np.array(arr))


class TeoImageProcessor:
    def __init__(self, size=224, image_mean=OPENAI_DATASET_MEAN, image_std=OPENAI_DATASET_STD, engine=None):
        self.size = size
        self.image_mean = tuple(image_mean)
        self.image_std = tuple(image_std)
        self.crop_size = {"height": size, "width": size}
        self.engine = engine              # bound engine: uint8 frames are preprocessed on the device

    def transform_device(self, frames_u8):
        """uint8 [T, H, W, 3] (host or device) -> [T, 3, S, S] on the engine's device in the engine's dtype."""
        import ctypes as C
        from . import _lib as L
        eng = self.engine
        T, H, W, _ = frames_u8.shape
        mean = (C.c_float * 3)(*self.image_mean)
        std = (C.c_float * 3)(*self.image_std)
        with eng.phase() as st:
            src = frames_u8.to(eng.device, non_blocking=True).contiguous()
            out = torch.empty(T, 3, self.size, self.size, dtype=eng.dtype, device=eng.device)
            L.check(eng.lib.teo_preprocess_frames(src.data_ptr(), out.data_ptr(), T, H, W, self.size, mean, std, eng.dt, st),
                    "teo_preprocess_frames")
        return out

    def transform(self, img):
        x = _to_chw_float(img)
        _, h, w = x.shape
        s = self.size
        if (h, w) != (s, s):
            if min(h, w) != s:                            # Resize(s): shorter edge -> s, aspect kept
                nh, nw = (s, int(s * w / h)) if h <= w else (int(s * h / w), s)
                x = F.interpolate(x[None], size=(nh, nw), mode="bicubic", align_corners=False, antialias=True)[0]
                h, w = nh, nw
            top, left = int(round((h - s) / 2.0)), int(round((w - s) / 2.0))
            x = x[:, top:top + s, left:left + s]
        mean = torch.tensor(self.image_mean, dtype=torch.float32).view(-1, 1, 1)
        std = torch.tensor(self.image_std, dtype=torch.float32).view(-1, 1, 1)
        return (x - mean) / std

    def __call__(self, images=None, text=None, return_tensors=None, **kwargs):
        if images is None:
            raise ValueError("You have to specify either text or images. Both cannot be none.")
        if not isinstance(images, (list, tuple)):
            images = [images]
        if self.engine is not None:
            u8 = [_to_hwc_uint8(im) for im in images]
            if all(u is not None for u in u8):
                outs = [None] * len(u8)
                groups = {}
                for i, u in enumerate(u8):                      # one launch per distinct frame size
                    groups.setdefault(tuple(u.shape), []).append(i)
                for idx in groups.values():
                    res = self.transform_device(torch.stack([u8[i] for i in idx]))
                    for j, i in enumerate(idx):
                        outs[i] = res[j]
                return {"pixel_values": torch.stack(outs)}
        return {"pixel_values": torch.stack([self.transform(im) for im in images])}

    def preprocess(self, images, return_tensors=None):
        return self.__call__(images=images, return_tensors=return_tensors)
