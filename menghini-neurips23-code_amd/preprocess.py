"""CLIP preprocessing (`_transform(n_px)` of openai-CLIP: Resize(n_px, BICUBIC) -> CenterCrop(n_px) -> RGB -> ToTensor ->
Normalize) with the resize / crop / normalise on the GPU (csrc/preprocess.hip).  The reference runs this on the host with
PIL, three times per item (data/dataset.py:64-79).  JPEG decoding stays with PIL; the decoded uint8 HWC image is
uploaded once and everything after that is native.  The bicubic resize is bit-exact with Pillow's (tests compare against
PIL.Image.resize): the coefficient windows are computed here exactly as Pillow's `precompute_coeffs` /
`normalize_coeffs_8bpc` do (float64, a = -0.5, 22-bit fixed point)."""
import functools
import math
import os
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
    # grip_preprocess_item (include/grip_amd.h), as a numpy record so a chunk's descriptors are filled column-wise
    _ITEM = np.dtype([("img", "<u8"), ("H", "<i4"), ("W", "<i4"), ("hcoef", "<u8"), ("hbounds", "<u8"), ("hksize", "<i4"), ("W_out", "<i4"),
                      ("vcoef", "<u8"), ("vbounds", "<u8"), ("vksize", "<i4"), ("H_out", "<i4"), ("crop_left", "<i4"), ("crop_top", "<i4"),
                      ("tmp", "<u8"), ("out", "<u8")])
    N_STAGING = 3          # staging buffers in rotation: one being decoded into, one in flight to the device, one spare

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

    def _staging(self, min_bytes):
        """The next page-locked staging buffer of the rotation (grow-only), free to overwrite: the upload that last read it has
        finished.  Returns (slot index, uint8 numpy view)."""
        st = self.__dict__.setdefault("_stage", {"k": 0, "bufs": [None] * self.N_STAGING, "events": [None] * self.N_STAGING})
        k = st["k"] = (st["k"] + 1) % self.N_STAGING
        if st["events"][k] is not None:
            st["events"][k].synchronize()
        if st["bufs"][k] is None or st["bufs"][k].numel() < min_bytes:
            st["bufs"][k] = torch.empty(max(int(min_bytes), 1 << 20), dtype=torch.uint8).pin_memory()
        return k, st["bufs"][k].numpy()

    def _upload_and_launch(self, host, packed, extra, out, done_event_slot=None):
        """`host`: page-locked uint8 tensor holding the chunk's pixels as `packed` (data.decode.Packed) describes; `extra`:
        {index: uint8 [H,W,3] array} for images outside the buffer.  One host-to-device copy, one descriptor upload, one launch pair."""
        B, n = len(packed.shapes), self.n_px
        if out is None:
            out = torch.empty(B, 3, n, n, dtype=torch.float32, device=self.device)
        if B == 0:
            return out
        lib = native.lib()
        stream = torch.cuda.current_stream()
        dev = host[:max(packed.used, 1)].to(self.device, non_blocking=True)
        if done_event_slot is not None:
            ev = torch.cuda.Event()
            ev.record(stream)
            self._stage["events"][done_event_slot] = ev
        keep = [dev]
        H, W = packed.shapes[:, 0].astype(np.int64), packed.shapes[:, 1].astype(np.int64)
        items = np.zeros(B, dtype=self._ITEM)
        items["img"] = dev.data_ptr() + packed.offsets
        for i, arr in extra.items():
            t = torch.from_numpy(np.ascontiguousarray(arr)).to(self.device)
            keep.append(t)
            items["img"][i] = t.data_ptr()
        items["H"], items["W"] = H, W
        # torchvision Resize(int): shorter side -> n, the other int(n * long / short); crop offsets as CenterCrop rounds them
        tall = W <= H
        oh = np.where(tall, (n * H / W).astype(np.int64), n)
        ow = np.where(tall, n, (n * W / H).astype(np.int64))
        items["H_out"], items["W_out"] = oh, ow
        items["crop_top"] = [int(round((int(v) - n) / 2.0)) for v in oh]
        items["crop_left"] = [int(round((int(v) - n) / 2.0)) for v in ow]
        for (h, w, o_h, o_w) in {(int(a), int(b), int(c), int(d)) for a, b, c, d in zip(H, W, oh, ow)}:
            if (o_h, o_w) == (h, w):
                continue
            hc, hb, hk = self._table(w, o_w)
            vc, vb, vk = self._table(h, o_h)
            sel = (H == h) & (W == w)
            items["hcoef"][sel], items["hbounds"][sel], items["hksize"][sel] = hc.data_ptr(), hb.data_ptr(), hk
            items["vcoef"][sel], items["vbounds"][sel], items["vksize"][sel] = vc.data_ptr(), vb.data_ptr(), vk
        tmp_sizes = (H * n * 3 + 255) // 256 * 256
        tmp_offs = np.concatenate([[0], np.cumsum(tmp_sizes)])
        tmp = torch.empty(int(tmp_offs[-1]), dtype=torch.uint8, device=self.device)
        items["tmp"] = tmp.data_ptr() + tmp_offs[:-1]
        items["out"] = out.data_ptr() + np.arange(B, dtype=np.int64) * (3 * n * n * 4)
        desc = torch.from_numpy(items.view(np.uint8)).to(self.device)
        keep += [tmp, desc]
        native.check(lib.grip_preprocess_batch(c_void_p(desc.data_ptr()), B, int(H.max()), n,
                                               c_void_p(self._mean.data_ptr()), c_void_p(self._std.data_ptr()), c_void_p(stream.cuda_stream)))
        for t in keep:
            t.record_stream(stream)
        return out

    def batch(self, images, out=None):
        """A list of decoded images (PIL, or uint8 [H, W, 3] arrays, any sizes) -> float32 [B, 3, n_px, n_px] on the device with
        ONE host-to-device copy and ONE launch pair (grip_preprocess_batch): the images are packed into a page-locked staging
        buffer, each gets a descriptor with its own Pillow coefficient tables.  Bit-identical to calling the transform per image."""
        from .data.decode import Packed, _round
        imgs = [self._as_u8(im) for im in images]
        sizes = [im.shape[0] * im.shape[1] * 3 for im in imgs]
        offs = np.concatenate([[0], np.cumsum([_round(s) for s in sizes])]).astype(np.int64)
        k, hv = self._staging(int(offs[-1]))
        for im, o, s in zip(imgs, offs, sizes):
            hv[o:o + s] = im.reshape(-1)
        packed = Packed(offs[:-1], np.array([im.shape[:2] for im in imgs], dtype=np.int32).reshape(len(imgs), 2), int(offs[-1]))
        return self._upload_and_launch(self._stage["bufs"][k], packed, {}, out, done_event_slot=k)

    def decode_chunk(self, paths, workers=8, processes=0):
        """Host half of `load_batch`: the files decoded (in parallel) into a staging buffer.  Returns an opaque handle for
        `finish_chunk`.  Thread-safe against one concurrent `finish_chunk` (the staging buffers rotate), so a caller can decode
        chunk i+1 on a background thread while chunk i is uploaded and encoded."""
        from .data import decode as D
        n = len(paths)
        hint = self.__dict__.get("_bytes_per_image", 600 * 1024)
        if processes > 0:
            dec = self.__dict__.get("_procs")
            if dec is None or dec.n_proc != processes:
                self.close()
                dec = self.__dict__["_procs"] = D.ProcessDecoder(processes, max(32 << 20, int(n * hint * 1.5)), slots=1)
            if dec.segs[0].size < n * hint * 1.1:
                dec.ensure(0, int(n * hint * 1.5))
            packed, overflow = dec.decode(paths, 0)
            # shared segment -> page-locked staging buffer: one plain copy per worker region, gaps between the regions left behind
            # (Packed.compact_into), split over a few threads; the segment is free again at once.  The segment itself is NOT
            # registered with the HIP runtime for DMA.
            k, hv = self._staging(sum(r[1] for r in packed.regions) + 1 if packed.regions else packed.used + 1)
            cp = self.__dict__.get("_copy_pool") or self.__dict__.setdefault("_copy_pool", D.make_thread_pool(4))
            packed = packed.compact_into(dec.slot_view(0), hv, cp)
            if n:
                total = packed.used + sum(a.nbytes for a in overflow.values())
                self.__dict__["_bytes_per_image"] = max(hint if not overflow else 0, total // n + 1)
            return ("thr", k, packed, overflow)
        pool = self.__dict__.get("_pool")
        if workers > 1 and (pool is None or pool._max_workers != workers):
            pool = self.__dict__["_pool"] = D.make_thread_pool(workers)
        k, hv = self._staging(int(n * hint * 1.25) + (1 << 20))
        packed, overflow = D.decode_threads(paths, hv, pool if workers > 1 else None)
        if n:
            total = packed.used + sum(a.nbytes for a in overflow.values())
            self.__dict__["_bytes_per_image"] = max(hint if not overflow else 0, total // n + 1)
        return ("thr", k, packed, overflow)

    def finish_chunk(self, handle, out=None):
        """Device half of `load_batch`: upload + descriptors + the batched launch pair for a `decode_chunk` handle."""
        kind, k, packed, overflow = handle
        return self._upload_and_launch(self._stage["bufs"][k], packed, overflow, out, done_event_slot=k)

    def load_batch(self, paths, workers=8, out=None, processes=0):
        """Image files -> preprocessed batch [B, 3, n_px, n_px] on the device: JPEG / PNG decoding in parallel (`workers` threads --
        Pillow's decoders release the GIL -- or `processes` decode processes around a shared-memory segment, see
        data/decode.py), then one upload from a page-locked staging buffer and one batched launch pair.  This replaces the per-item host transform of the
        reference's datasets (data/dataset.py:56-89); results are bit-identical to the per-image transform."""
        return self.finish_chunk(self.decode_chunk(list(paths), workers=workers, processes=processes), out=out)

    def close(self):
        """Stop the decode processes and release their shared segment (idempotent; also runs at interpreter exit)."""
        dec = self.__dict__.pop("_procs", None)
        if dec is not None:
            dec.close()
