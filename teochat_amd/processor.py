"""Image preprocessing for the tower (host side).

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
    t = torch.from_numpy(np.ascontiguousarray(arr)).permute(2, 0, 1)
    return t.to(torch.float32) / 255.0 if t.dtype == torch.uint8 else t.to(torch.float32)


class TeoImageProcessor:
    def __init__(self, size=224, image_mean=OPENAI_DATASET_MEAN, image_std=OPENAI_DATASET_STD):
        self.size = size
        self.image_mean = tuple(image_mean)
        self.image_std = tuple(image_std)
        self.crop_size = {"height": size, "width": size}

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
        return {"pixel_values": torch.stack([self.transform(im) for im in images])}

    def preprocess(self, images, return_tensors=None):
        return self.__call__(images=images, return_tensors=return_tensors)
