"""`clip.load` / `clip.tokenize` of the native engine (same call signatures as openai-CLIP's)."""
import hashlib
import os
import re

import torch

from .. import config as _cfg, weights as _weights
from . import model  # noqa: F401
from .model import CLIP

_WORD = re.compile(r"[a-z]+|[0-9]|[^\sa-z0-9]+")


def _word_id(w: str) -> int:
    if w == "x":
        return _cfg.X_TOKEN
    h = int.from_bytes(hashlib.sha256(w.encode()).digest()[:4], "little")
    return 1000 + h % 39000


_TOKENIZER = None


def _bpe():
    """The real BPE tokenizer when its merges file is available ($CLIP_BPE_VOCAB), else None."""
    global _TOKENIZER
    if _TOKENIZER is None:
        from . import simple_tokenizer as st
        _TOKENIZER = st.SimpleTokenizer() if os.path.exists(st.default_bpe_path()) else False
    return _TOKENIZER or None


def tokenize(texts, context_length: int = 77, truncate: bool = False):
    """[n, 77] int tensor [SOT, ids..., EOT, 0...] (same contract as openai-CLIP's clip.tokenize).
    With a merges file ($CLIP_BPE_VOCAB) this is byte-level BPE (clip/simple_tokenizer.py); without one --
    the vocabulary does not exist offline -- it is a STAND-IN word-hash tokenizer that keeps
    the structure CustomTextEncoder relies on (SOT/EOT, "X" -> 343, EOT = largest id;
    models/clip_encoders.py:54-60,86-89)."""
    if isinstance(texts, str):
        texts = [texts]
    tk = _bpe()
    out = torch.zeros(len(texts), context_length, dtype=torch.int)
    for i, t in enumerate(texts):
        if tk is not None:
            ids = [tk.sot] + tk.encode(t) + [tk.eot]
            eot = tk.eot
        else:
            ids = [_cfg.SOT_TOKEN] + [_word_id(w) for w in _WORD.findall(t.lower())] + [_cfg.EOT_TOKEN]
            eot = _cfg.EOT_TOKEN
        if len(ids) > context_length:
            if not truncate:
                raise RuntimeError(f"Input {t} is too long for context length {context_length}")
            ids = ids[:context_length]
            ids[-1] = eot
        out[i, : len(ids)] = torch.tensor(ids, dtype=torch.int)
    return out


def available_models():
    return [k for k in _cfg.CLIP_CONFIGS]


def _preprocess(n_px, device):
    from ..preprocess import ClipPreprocess
    return ClipPreprocess(n_px, device)


def load(name: str, device="cuda", jit: bool = False, download_root=None, seed: int = 0):
    """(model, preprocess).  Weights: $CLIP_WEIGHTS (a torch-saved OpenAI state_dict) when set,
    else the seeded synthetic init of grip_amd.weights (no checkpoints exist offline)."""
    d = _cfg.get_dims(name)
    m = CLIP(d, device)
    path = os.environ.get("CLIP_WEIGHTS")
    if path:
        sd = torch.load(path, map_location="cpu")
        sd = sd.get("state_dict", sd)
    else:
        sd = {k: torch.from_numpy(v) for k, v in _weights.init_state_dict(d, seed).items()}
    load_openai_state_dict(m, sd)
    return m, _preprocess(d.image_resolution, device)


def load_openai_state_dict(m: CLIP, sd):
    own = dict(m.named_parameters())
    missing = [k for k in own if k not in sd]
    if missing:
        raise RuntimeError(f"state_dict is missing keys: {missing[:5]}...")
    with torch.no_grad():
        for k, p in own.items():
            p.copy_(sd[k].reshape(p.shape).to(p.dtype))
    m.visual.tower.finalize()
    m.tower.finalize()
