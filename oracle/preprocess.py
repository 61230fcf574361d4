"""ORACLE (test infrastructure) -- the CLIP image transform the reference applies on the host
(`clip.load(...)[1]`, used as `data.transform` in e.g. methods/clip_baseline.py:53 and data/dataset.py:64-79):
openai-CLIP's `_transform(n_px)` = Resize(n_px, interpolation=BICUBIC) -> CenterCrop(n_px) -> convert("RGB") -> ToTensor ->
Normalize(mean, std).  torchvision is not installed; for PIL inputs its Resize / CenterCrop call exactly the PIL
operations below, so PIL + numpy IS the reference here (Pillow's own resampler is the arithmetic being matched)."""
import numpy as np
from PIL import Image

MEAN = np.array((0.48145466, 0.4578275, 0.40821073), dtype=np.float32)
STD = np.array((0.26862954, 0.26130258, 0.27577711), dtype=np.float32)


def resized_size(h, w, n_px):
    """torchvision.transforms.Resize(int): shorter side -> n_px, longer side -> int(n_px * long / short)."""
    if w <= h:
        return int(n_px * h / w), n_px
    return n_px, int(n_px * w / h)


def clip_transform(img, n_px):
    """PIL image or uint8 [H,W,3] array -> float32 [3, n_px, n_px]."""
    im = img if isinstance(img, Image.Image) else Image.fromarray(np.asarray(img, dtype=np.uint8))
    w, h = im.size
    oh, ow = resized_size(h, w, n_px)
    im = im.resize((ow, oh), Image.BICUBIC)
    top, left = int(round((oh - n_px) / 2.0)), int(round((ow - n_px) / 2.0))      # torchvision CenterCrop
    im = im.crop((left, top, left + n_px, top + n_px)).convert("RGB")
    x = np.asarray(im, dtype=np.float32).transpose(2, 0, 1) / np.float32(255.0)    # ToTensor
    return (x - MEAN[:, None, None]) / STD[:, None, None]                          # Normalize
