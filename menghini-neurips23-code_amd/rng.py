"""Build-owned counter-based RNG (Philox-4x64 raw stream + Box-Muller).

Both boxes (this container and the GPU box) regenerate identical synthetic
weights / inputs from (seed, stream) without shipping blobs.  Only numpy's
BitGenerator raw stream is used (numpy guarantees its stability); the
uniform->normal transform is spelled out here so it cannot drift.
"""
import hashlib

import numpy as np

_TWO_PI = 6.283185307179586


def stream_id(name: str) -> int:
    """Stable 63-bit id for a named tensor (e.g. 'visual.transformer.resblocks.3.mlp.c_fc.weight')."""
    return int.from_bytes(hashlib.sha256(name.encode()).digest()[:8], "little") >> 1


def uniform(seed: int, stream: int, n: int) -> np.ndarray:
    """n float64 uniforms in (0, 1]."""
    bg = np.random.Philox(key=np.array([seed, stream], dtype=np.uint64))
    raw = bg.random_raw(n)
    return ((raw >> np.uint64(11)).astype(np.float64) + 1.0) * (1.0 / 9007199254740992.0)


def normal(seed: int, stream: int, shape, mean=0.0, std=1.0) -> np.ndarray:
    """float32 N(mean, std) of `shape`, Box-Muller on the Philox stream."""
    n = int(np.prod(shape)) if len(shape) else 1
    m = (n + 1) // 2
    u = uniform(seed, stream, 2 * m)
    r = np.sqrt(-2.0 * np.log(u[:m]))
    t = _TWO_PI * u[m:]
    z = np.concatenate([r * np.cos(t), r * np.sin(t)])[:n]
    return (z * std + mean).astype(np.float32).reshape(shape)


def uniform_range(seed: int, stream: int, shape, lo, hi) -> np.ndarray:
    n = int(np.prod(shape)) if len(shape) else 1
    u = uniform(seed, stream, n)
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


def integers(seed: int, stream: int, shape, lo, hi) -> np.ndarray:
    """int64 in [lo, hi)."""
    n = int(np.prod(shape)) if len(shape) else 1
    u = uniform(seed, stream, n)
    return np.minimum((lo + np.floor((hi - lo) * (1.0 - u))).astype(np.int64), hi - 1).reshape(shape)
