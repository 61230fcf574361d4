"""CLIP preprocessing (`_transform(n_px)` of openai-CLIP: Resize(n_px, BICUBIC) -> CenterCrop(n_px) -> RGB -> ToTensor ->
Normalize) with the resize / crop / normalise on the GPU (csrc/preprocess.hip).  The reference runs this on the host with
PIL, three times per item (data/dataset.py:64-79).  JPEG decoding stays with PIL; the decoded uint8 HWC image is
uploaded once and everything after that is native.  The bicubic resize is bit-exact with Pillow's (tests compare against
PIL.Image.resize): the coefficient windows are computed here exactly as Pillow's `precompute_coeffs` /
`normalize_coeffs_8bpc` do (float64, a = -0.5, 22-bit fixed point)."""
import functools
import math
from ctypes import c_void_p

import numpy as np
import torch

from . import native

MEAN = (0.48145466, 0.4578275, 0.40821073)
STD = (0.26862954, 0.26130258, 0.27577711)
_PRECISION_BITS = 32 - 8 - 2


def _bicubic(x, a=-0.5):
    x = np.abs(x)
    return np.where(x < 1.0, ((a + 2.0) * x - (a + 3.0)) * x * x + 1, np.where(x < 2.0, (((x - 5) * x + 8) * x - 4) * a, 0.0))


@functools.lru_cache(maxsize=256)
def resample_coeffs(in_size, out_size):
    """(coef int32 [out, ksize], bounds int32 [out, 2] = (first input index, count), ksize) as Pillow computes them."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    coef = np.zeros((out_size, ksize), dtype=np.int32)
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = _bicubic((np.arange(xmax, dtype=np.float64) + xmin - center + 0.5) * ss)
        ww = w.sum()
        if ww != 0.0:
            w = w / ww
        fixed = np.where(w < 0, -0.5 + w * (1 << _PRECISION_BITS), 0.5 + w * (1 << _PRECISION_BITS))
        coef[xx, :xmax] = np.trunc(fixed).astype(np.int32)
        bounds[xx] = (xmin, xmax)
    return coef, bounds, ksize


def resized_size(h, w, n_px):
    """torchvision Resize(int): shorter side -> n_px, the other int(n_px * long / short)."""
    if w <= h:
        return int(n_px * h / w), n_px
    return n_px, int(n_px * w / h)


class ClipPreprocess:
    """Callable with the role of the `preprocess` that clip.load returns: PIL image (any mode) or uint8 [H,W,3] array ->
    float32 [3, n_px, n_px] tensor on the device.  Tensors that are already float [3,R,R] pass through."""

    def __init__(self, n_px, device="cuda"):
        self.n_px = n_px
        self.device = torch.device(device)
        self._tables = {}
        self._mean = torch.tensor(MEAN, dtype=torch.float32)
        self._std = torch.tensor(STD, dtype=torch.float32)

    def _table(self, in_size, out_size):
        key = (in_size, out_size)
        if key not in self._tables:
            coef, bounds, ksize = resample_coeffs(in_size, out_size)
            self._tables[key] = (torch.from_numpy(coef).to(self.device), torch.from_numpy(bounds).to(self.device), ksize)
            if len(self._tables) > 512:
                self._tables.pop(next(iter(self._tables)))
        return self._tables[key]

    def __call__(self, img):
        if torch.is_tensor(img) and img.is_floating_point():
            return img
        if not torch.is_tensor(img):
            if hasattr(img, "convert"):
                img = img.convert("RGB")
            img = torch.from_numpy(np.array(img, dtype=np.uint8))
        if img.dim() != 3 or img.shape[2] != 3 or img.dtype != torch.uint8:
            raise ValueError(f"expected a uint8 [H, W, 3] image, got {tuple(img.shape)} {img.dtype}")
        lib = native.lib()
        h, w = int(img.shape[0]), int(img.shape[1])
        n = self.n_px
        oh, ow = resized_size(h, w, n)
        top, left = int(round((oh - n) / 2.0)), int(round((ow - n) / 2.0))
        x = img.to(self.device, non_blocking=True).contiguous()
        out = torch.empty(3, n, n, dtype=torch.float32, device=self.device)
        s = c_void_p(torch.cuda.current_stream().cuda_stream)
        p = lambda t: c_void_p(t.data_ptr()) if t is not None else c_void_p(0)
        if (oh, ow) == (h, w):
            native.check(lib.grip_preprocess_image(p(x), h, w, None, None, 0, ow, None, None, 0, oh, left, top, n, p(self._mean), p(self._std), None, p(out), s))
            return out
        hc, hb, hk = self._table(w, ow)
        vc, vb, vk = self._table(h, oh)
        tmp = torch.empty(h * n * 3, dtype=torch.uint8, device=self.device)
        native.check(lib.grip_preprocess_image(p(x), h, w, p(hc), p(hb), hk, ow, p(vc), p(vb), vk, oh, left, top, n, p(self._mean), p(self._std),
                                               p(tmp), p(out), s))
        return out

    # ------------------------------------------------------------------ batched path
    @staticmethod
    def _as_u8(img):
        if torch.is_tensor(img):
            img = img.cpu().numpy()
        if hasattr(img, "convert"):
            img = np.asarray(img.convert("RGB"), dtype=np.uint8)
        img = np.ascontiguousarray(img, dtype=np.uint8)
        if img.ndim != 3 or img.shape[2] != 3:
            raise ValueError(f"expected a uint8 [H, W, 3] image, got {img.shape}")
        return img

    def batch(self, images, out=None):
        """A list of decoded images (PIL, or uint8 [H, W, 3] arrays, any sizes) -> float32 [B, 3, n_px, n_px] on the device with
        ONE host-to-device copy and ONE launch pair (grip_preprocess_batch): the images are packed into a single pinned buffer,
        each gets a descriptor with its own Pillow coefficient tables.  Bit-identical to calling the transform per image."""
        import ctypes
        imgs = [self._as_u8(im) for im in images]
        B, n = len(imgs), self.n_px
        if out is None:
            out = torch.empty(B, 3, n, n, dtype=torch.float32, device=self.device)
        if B == 0:
            return out
        lib = native.lib()
        sizes = [im.shape[0] * im.shape[1] * 3 for im in imgs]
        offs = np.concatenate([[0], np.cumsum([(s + 255) // 256 * 256 for s in sizes])]).astype(np.int64)
        host = torch.empty(int(offs[-1]), dtype=torch.uint8).pin_memory()
        hv = host.numpy()
        for im, o, s in zip(imgs, offs, sizes):
            hv[o:o + s] = im.reshape(-1)
        dev = host.to(self.device, non_blocking=True)
        tmp_offs = np.concatenate([[0], np.cumsum([(im.shape[0] * n * 3 + 255) // 256 * 256 for im in imgs])]).astype(np.int64)
        tmp = torch.empty(int(tmp_offs[-1]), dtype=torch.uint8, device=self.device)
        items = (native.PreprocessItem * B)()
        for i, im in enumerate(imgs):
            h, w = im.shape[0], im.shape[1]
            oh, ow = resized_size(h, w, n)
            it = items[i]
            it.img, it.H, it.W = dev.data_ptr() + int(offs[i]), h, w
            it.W_out, it.H_out = ow, oh
            it.crop_top, it.crop_left = int(round((oh - n) / 2.0)), int(round((ow - n) / 2.0))
            if (oh, ow) != (h, w):
                hc, hb, hk = self._table(w, ow)
                vc, vb, vk = self._table(h, oh)
                it.hcoef, it.hbounds, it.hksize = hc.data_ptr(), hb.data_ptr(), hk
                it.vcoef, it.vbounds, it.vksize = vc.data_ptr(), vb.data_ptr(), vk
            it.tmp = tmp.data_ptr() + int(tmp_offs[i])
            it.out = out.data_ptr() + i * 3 * n * n * 4
        desc = torch.frombuffer(bytearray(bytes(items)), dtype=torch.uint8).to(self.device)
        s = c_void_p(torch.cuda.current_stream().cuda_stream)
        native.check(lib.grip_preprocess_batch(c_void_p(desc.data_ptr()), B, max(im.shape[0] for im in imgs), n,
                                               c_void_p(self._mean.data_ptr()), c_void_p(self._std.data_ptr()), s))
        for t in (dev, tmp, desc):
            t.record_stream(torch.cuda.current_stream())
        return out

    def load_batch(self, paths, workers=8, out=None):
        """Image files -> preprocessed batch: JPEG / PNG decoding on a thread pool (Pillow's decoders release the GIL), then
        `batch`.  This replaces the per-item host transform of the reference's datasets (data/dataset.py:56-89)."""
        from concurrent.futures import ThreadPoolExecutor

        from PIL import Image

        def decode(p):
            with Image.open(p) as im:
                return np.asarray(im.convert("RGB"), dtype=np.uint8)
        if workers > 1 and len(paths) > 1:
            pool = self.__dict__.get("_pool")
            if pool is None or pool._max_workers != workers:
                pool = self.__dict__["_pool"] = ThreadPoolExecutor(max_workers=workers)
            imgs = list(pool.map(decode, paths))
        else:
            imgs = [decode(p) for p in paths]
        return self.batch(imgs, out=out)
