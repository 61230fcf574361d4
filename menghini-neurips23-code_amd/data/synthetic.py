"""Seeded synthetic image pools shared by the parity tests, the golden-vector generator and bench.py.

`structured_images` gives every image a colour cast and a low-frequency ramp on top of noise: random-init towers
separate such images far better than i.i.d. noise, so class scores have real margins.  Images are generated in
independent 64-image blocks (one RNG stream per block), so any slice [lo, hi) of a pool is reproducible on both
boxes without materialising the whole pool."""
import numpy as np
import torch

from .. import rng

_BLOCK = 64


def _block(seed, blk, res):
    x = rng.normal(seed, rng.stream_id(f"pool.x.{res}.{blk}"), (_BLOCK, 3, res, res))
    mu = rng.normal(seed, rng.stream_id(f"pool.mu.{res}.{blk}"), (_BLOCK, 3, 1, 1)) * 2.0
    r = rng.normal(seed, rng.stream_id(f"pool.r.{res}.{blk}"), (_BLOCK, 3, 1, 1))
    ramp = np.linspace(-1.0, 1.0, res, dtype=np.float32).reshape(1, 1, 1, res) * r
    return (x * np.float32(0.5) + mu + ramp).astype(np.float32)


def structured_images(seed, lo, hi, res):
    """float32 tensor [hi - lo, 3, res, res]: images lo .. hi-1 of the pool `seed`."""
    out = np.empty((hi - lo, 3, res, res), dtype=np.float32)
    for blk in range(lo // _BLOCK, (hi + _BLOCK - 1) // _BLOCK):
        b = _block(seed, blk, res)
        a, e = max(lo, blk * _BLOCK), min(hi, (blk + 1) * _BLOCK)
        out[a - lo: e - lo] = b[a - blk * _BLOCK: e - blk * _BLOCK]
    return torch.from_numpy(out)


def pool_paths(n, stem="/data/pool/train"):
    """Path strings whose lexicographic order differs from the dataset order (score ties break on the path)."""
    return [f"{stem}/{(i * 7919) % 100000:05d}_{i}.jpg" for i in range(n)]
