"""ORACLE (test infrastructure) -- `clip.load` / `clip.tokenize` stand-ins.

`load` builds the CPU fp32 oracle CLIP with the build's seeded synthetic weights
(no checkpoints exist offline).  `tokenize` is a STAND-IN, not BPE: the real
vocabulary file is absent.  It keeps the structure the reference
relies on -- [SOT, word ids..., EOT, 0...] of width 77 with EOT the largest id so
`argmax` finds it (models/clip_encoders.py:86-89) and "X" -> 343 for the CoOp
placeholder (models/clip_encoders.py:54-57) -- and maps every other word to a
stable hash id.  The same function body lives in grip_amd.clip (product side);
tests assert the two agree.
"""
import hashlib
import os
import re
import sys

import torch

_repo = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _repo not in sys.path:
    sys.path.insert(0, _repo)
import grip_amd  # noqa: E402  (shared dims / seeded weights only -- no engine code)
from grip_amd import config as _cfg, weights as _weights  # noqa: E402

from . import model  # noqa: E402,F401
from .model import CLIP  # noqa: E402

_WORD = re.compile(r"[a-z]+|[0-9]|[^\sa-z0-9]+")


def _word_id(w: str) -> int:
    if w == "x":
        return _cfg.X_TOKEN
    h = int.from_bytes(hashlib.sha256(w.encode()).digest()[:4], "little")
    return 1000 + h % 39000


def tokenize(texts, context_length: int = 77, truncate: bool = False):
    if isinstance(texts, str):
        texts = [texts]
    out = torch.zeros(len(texts), context_length, dtype=torch.int)
    for i, t in enumerate(texts):
        ids = [_cfg.SOT_TOKEN] + [_word_id(w) for w in _WORD.findall(t.lower())] + [_cfg.EOT_TOKEN]
        if len(ids) > context_length:
            if not truncate:
                raise RuntimeError(f"Input {t} is too long for context length {context_length}")
            ids = ids[:context_length]
            ids[-1] = _cfg.EOT_TOKEN
        out[i, : len(ids)] = torch.tensor(ids, dtype=torch.int)
    return out


def build_model(name: str, seed: int = 0) -> CLIP:
    d = _cfg.get_dims(name)
    m = CLIP(d.embed_dim, d.image_resolution, d.vision_layers, d.vision_width, d.vision_patch_size,
             d.context_length, d.vocab_size, d.transformer_width, d.transformer_heads, d.transformer_layers)
    sd = {k: torch.from_numpy(v) for k, v in _weights.init_state_dict(d, seed).items()}
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert not [k for k in missing if "attn_mask" not in k], missing
    return m.float().eval()


def _transform(n_px):
    def t(img):
        raise RuntimeError("oracle transform: torchvision is not installed; feed tensors")
    t.n_px = n_px
    return t


def load(name: str, device="cpu", jit: bool = False, download_root=None, seed: int = 0):
    m = build_model(name, seed)
    for p in m.parameters():
        p.requires_grad_(False)
    return m.to(device), _transform(m.visual.input_resolution)


def available_models():
    return list(_cfg.CLIP_CONFIGS)
