"""Image preprocessing for the tower, on the device (teo_preprocess_frames).

Mirrors LanguageBindImageProcessor / get_image_transform of the reference
(languagebind/image/processing_image.py:7-8,15-25,33-68): ToTensor -> Resize(224, bicubic, antialias) -> CenterCrop(224) ->
Normalize(OPENAI mean/std), and process_images' `image_aspect_ratio == 'pad'` mode (mm_utils.py:14-36: expand2square first).
The host only decodes the file and uploads the raw uint8 pixels (4x fewer PCIe bytes than a float tensor); every arithmetic
step is one HIP kernel writing the layout and dtype the tower consumes.  There is no host implementation of the transform in
this package (the CPU restatement the kernel is checked against is test infrastructure, outside the package): a processor
that is not bound to an engine raises when asked to preprocess.
"""
import ctypes as C

import numpy as np
import torch

OPENAI_DATASET_MEAN = (0.48145466, 0.4578275, 0.40821073)
OPENAI_DATASET_STD = (0.26862954, 0.26130258, 0.27577711)


def _to_hwc_uint8(img):
    """path / PIL image / uint8 HWC array or tensor -> contiguous uint8 HWC RGB torch tensor (what ToTensor divides by 255)."""
    if isinstance(img, str):
        from PIL import Image
        img = Image.open(img).convert("RGB")                # load_and_transform_image, processing_image.py:28-31
    if isinstance(img, torch.Tensor):
        if img.dtype == torch.uint8 and img.dim() == 3 and img.shape[-1] == 3:
            return img.contiguous()
        raise TypeError(f"image tensor must be uint8 [H, W, 3] (got {img.dtype} {tuple(img.shape)}); already-preprocessed float "
                        "frames go straight to the model as `images=`")
    if hasattr(img, "convert"):
        if getattr(img, "mode", "RGB") != "RGB":
            img = img.convert("RGB")
        return torch.from_numpy(np.array(img))
    arr = np.asarray(img)
    if arr.dtype != np.uint8 or arr.ndim != 3 or arr.shape[-1] != 3:
        raise TypeError(f"image array must be uint8 [H, W, 3] (got {arr.dtype} {arr.shape})")
    return torch.from_numpy(np.array(arr))


class TeoImageProcessor:
    def __init__(self, size=224, image_mean=OPENAI_DATASET_MEAN, image_std=OPENAI_DATASET_STD, engine=None):
        self.size = size
        self.image_mean = tuple(image_mean)
        self.image_std = tuple(image_std)
        self.crop_size = {"height": size, "width": size}
        self.engine = engine              # the frames are preprocessed on this engine's device

    def transform_device(self, frames_u8, pad_rgb=None):
        """uint8 [T, H, W, 3] (host or device) -> [T, 3, S, S] on the engine's device in the engine's dtype.
        pad_rgb: (r, g, b) bytes -> expand2square with that background first (image_aspect_ratio == 'pad')."""
        from . import _lib as L
        eng = self.engine
        if eng is None:
            raise RuntimeError("TeoImageProcessor is not bound to an engine: preprocessing runs on the MI355X only "
                               "(no CPU fallback); load the model with load_pretrained_model()")
        T, H, W, _ = frames_u8.shape
        mean = (C.c_float * 3)(*self.image_mean)
        std = (C.c_float * 3)(*self.image_std)
        with eng.phase() as st:
            src = frames_u8.to(eng.device, non_blocking=True).contiguous()
            out = torch.empty(T, 3, self.size, self.size, dtype=eng.dtype, device=eng.device)
            if pad_rgb is None:
                L.check(eng.lib.teo_preprocess_frames(src.data_ptr(), out.data_ptr(), T, H, W, self.size, mean, std, eng.dt, st),
                        "teo_preprocess_frames")
            else:
                fill = (C.c_ubyte * 3)(*[int(v) & 255 for v in pad_rgb])
                L.check(eng.lib.teo_preprocess_frames_pad(src.data_ptr(), out.data_ptr(), T, H, W, self.size, mean, std, fill,
                                                          eng.dt, st), "teo_preprocess_frames_pad")
        return out

    def __call__(self, images=None, text=None, return_tensors=None, pad_rgb=None, **kwargs):
        if images is None:
            raise ValueError("You have to specify either text or images. Both cannot be none.")
        if not isinstance(images, (list, tuple)):
            images = [images]
        u8 = [_to_hwc_uint8(im) for im in images]
        outs = [None] * len(u8)
        groups = {}
        for i, u in enumerate(u8):                              # one launch per distinct frame size
            groups.setdefault(tuple(u.shape), []).append(i)
        for idx in groups.values():
            res = self.transform_device(torch.stack([u8[i] for i in idx]), pad_rgb=pad_rgb)
            for j, i in enumerate(idx):
                outs[i] = res[j]
        return {"pixel_values": torch.stack(outs)}

    def preprocess(self, images, return_tensors=None, pad_rgb=None):
        return self.__call__(images=images, return_tensors=return_tensors, pad_rgb=pad_rgb)
