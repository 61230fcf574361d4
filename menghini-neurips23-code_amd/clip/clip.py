"""`clip.load` / `clip.tokenize` of the native engine (same call signatures as openai-CLIP's)."""
import hashlib
import logging
import os
import re

import torch

from .. import config as _cfg, weights as _weights
from . import model  # noqa: F401
from .model import CLIP

log = logging.getLogger(__name__)
_WARNED = set()
PROVENANCE = {"weights": None, "tokenizer": None}   # what the last clip.load / clip.tokenize actually used (stored with results)


def _warn_once(key, msg):
    if key not in _WARNED:
        _WARNED.add(key)
        log.warning(msg)


_WORD = re.compile(r"[a-z]+|[0-9]|[^\sa-z0-9]+")


def _word_id(w: str) -> int:
    if w == "x":
        return _cfg.X_TOKEN
    h = int.from_bytes(hashlib.sha256(w.encode()).digest()[:4], "little")
    return 1000 + h % 39000


_TOKENIZER = None
_SD_CACHE = {}


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
    if tk is None:
        if os.environ.get("CLIP_WEIGHTS"):
            raise RuntimeError("CLIP_WEIGHTS is set but no BPE vocabulary is available ($CLIP_BPE_VOCAB): the stand-in word-hash "
                               "tokenizer would index a real token_embedding with meaningless ids")
        _warn_once("tok", "clip.tokenize: no BPE vocabulary ($CLIP_BPE_VOCAB) -- using the STAND-IN word-hash tokenizer; "
                          "text features are only meaningful together with the synthetic weights")
    PROVENANCE["tokenizer"] = "bpe" if tk is not None else "stand-in"
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


def load(name: str, device="cuda", jit: bool = False, download_root=None, seed: int = 0, exact=None, synthetic="standard", fp16_checkpoint=None):
    """(model, preprocess).  Weights: $CLIP_WEIGHTS (a torch-saved OpenAI state_dict) when set,
    else the seeded synthetic init of grip_amd.weights (no checkpoints exist offline).
    exact=True (or GRIP_EXACT=1): f32 towers -- weights, activations, attention and residual stream in fp32, the
    arithmetic the reference's CPU path uses (clip.load(..., device="cpu") keeps fp32) -- for index-exact comparison of
    the pseudolabel lists; inference only, ~1/10 of the f16 throughput.
    synthetic="stress": the synthetic init with outlier channels and an image-dependent f16 overflow (weights.stress_state_dict: tests / bench only).
    fp16_checkpoint=True (or GRIP_SYNTHETIC_FP16=1; synthetic weights only): the matrix weights rounded to f16 numbers, as every published CLIP
    checkpoint holds them (weights.on_f16_grid) -- the f32 twin then computes on exactly the values the f16 towers hold, as the reference's CPU
    path does on a real checkpoint."""
    d = _cfg.get_dims(name)
    if fp16_checkpoint is None:
        fp16_checkpoint = os.environ.get("GRIP_SYNTHETIC_FP16", "0") == "1"
    if fp16_checkpoint:
        synthetic = synthetic + "+fp16"
    if exact is None:
        exact = os.environ.get("GRIP_EXACT", "0") == "1"
    exact = int(exact)          # 2 = split-f16 towers (developer / tests: the middle tier on its own)
    m = CLIP(d, device, exact=exact)
    path = os.environ.get("CLIP_WEIGHTS")
    if path:
        # an OpenAI checkpoint: TorchScript archive (what OpenAI publishes) or a pickled state_dict; every dimension is
        # re-derived from the tensor shapes and must agree with the encoder that was asked for
        sd = _weights.read_checkpoint(path)
        got = _weights.dims_from_state_dict(sd, name)
        if got != d:
            raise RuntimeError(f"$CLIP_WEIGHTS={path} holds {got}, not the requested {d}")
        _weights.check_state_dict(sd, d)
        PROVENANCE["weights"] = path
    else:
        _warn_once("weights", "clip.load: no $CLIP_WEIGHTS -- using SYNTHETIC seeded random-init weights (accuracies are meaningless)")
        PROVENANCE["weights"] = "synthetic"
        if synthetic.split("+")[0] not in ("standard", "stress"):
            raise ValueError(f"synthetic={synthetic!r}: expected 'standard' or 'stress'")
        if synthetic != "standard":
            PROVENANCE["weights"] = "synthetic-" + synthetic
        sd = {k: torch.from_numpy(v) for k, v in _synthetic_sd(name, d, seed, synthetic).items()}    # a second load of the same synthetic model reuses the arrays
    load_openai_state_dict(m, sd)
    if not exact:
        src = None if path is None else sd       # a checkpoint's tensors are kept for the twin; the synthetic init is regenerated

        def build_twin(precision=1):
            t = CLIP(d, device, exact=precision, vision_only=precision == 2)      # the middle tier only ever encodes images
            load_openai_state_dict(t, src if src is not None else {k: torch.from_numpy(v) for k, v in _synthetic_sd(name, d, seed, synthetic).items()})
            return t
        m._twin[1] = build_twin
        m._split[1] = lambda: build_twin(2)
    return m, _preprocess(d.image_resolution, device)


def _synthetic_sd(name, d, seed, variant="standard"):
    if _SD_CACHE.get("key") != (name, seed, variant):
        base, _, grid = variant.partition("+")
        sd = _weights.stress_state_dict(d, seed) if base == "stress" else _weights.init_state_dict(d, seed)
        _SD_CACHE.update(key=(name, seed, variant), sd=_weights.on_f16_grid(sd) if grid == "fp16" else sd)
    return _SD_CACHE["sd"]


def load_openai_state_dict(m: CLIP, sd):
    own = dict(m.named_parameters())
    missing = [k for k in own if k not in sd]
    if missing:
        raise RuntimeError(f"state_dict is missing keys: {missing[:5]}...")
    with torch.no_grad():
        for k, p in own.items():
            p.copy_(sd[k].reshape(p.shape).to(p.dtype))
    m.visual.tower.finalize()
    if m.tower is not None:
        m.tower.finalize()
