"""Seeded synthetic CLIP weights under the OpenAI `state_dict` key names.

There are no CLIP checkpoints offline, so parity and throughput
are measured on random-init towers of the true dimensions.  Standard deviations
follow the published openai/CLIP `initialize_parameters`; LayerNorm affine
parameters and biases are perturbed away from (1, 0) so that a kernel which
drops one of them cannot pass a parity test.  A user who has real weights loads
them through `load_state_dict` with the same keys.
"""
import math

import numpy as np

from . import rng
from .config import ClipDims


def _block_keys(prefix, width):
    w = width
    return [
        (f"{prefix}.ln_1.weight", (w,), "ln_w"),
        (f"{prefix}.ln_1.bias", (w,), "ln_b"),
        (f"{prefix}.attn.in_proj_weight", (3 * w, w), "attn"),
        (f"{prefix}.attn.in_proj_bias", (3 * w,), "bias"),
        (f"{prefix}.attn.out_proj.weight", (w, w), "proj"),
        (f"{prefix}.attn.out_proj.bias", (w,), "bias"),
        (f"{prefix}.ln_2.weight", (w,), "ln_w"),
        (f"{prefix}.ln_2.bias", (w,), "ln_b"),
        (f"{prefix}.mlp.c_fc.weight", (4 * w, w), "fc"),
        (f"{prefix}.mlp.c_fc.bias", (4 * w,), "bias"),
        (f"{prefix}.mlp.c_proj.weight", (w, 4 * w), "proj"),
        (f"{prefix}.mlp.c_proj.bias", (w,), "bias"),
    ]


def weight_spec(d: ClipDims):
    """Ordered [(key, shape, kind)] for the ViT + text CLIP the reference uses."""
    vw, tw, p = d.vision_width, d.transformer_width, d.vision_patch_size
    spec = [
        ("visual.conv1.weight", (vw, 3, p, p), "conv"),
        ("visual.class_embedding", (vw,), "vscale"),
        ("visual.positional_embedding", (d.vision_seq, vw), "vscale"),
        ("visual.ln_pre.weight", (vw,), "ln_w"),
        ("visual.ln_pre.bias", (vw,), "ln_b"),
    ]
    for i in range(d.vision_layers):
        spec += [(k, s, "v" + kind) for k, s, kind in _block_keys(f"visual.transformer.resblocks.{i}", vw)]
    spec += [
        ("visual.ln_post.weight", (vw,), "ln_w"),
        ("visual.ln_post.bias", (vw,), "ln_b"),
        ("visual.proj", (vw, d.embed_dim), "vscale"),
        ("token_embedding.weight", (d.vocab_size, tw), "tok"),
        ("positional_embedding", (d.context_length, tw), "pos"),
    ]
    for i in range(d.transformer_layers):
        spec += [(k, s, "t" + kind) for k, s, kind in _block_keys(f"transformer.resblocks.{i}", tw)]
    spec += [
        ("ln_final.weight", (tw,), "ln_w"),
        ("ln_final.bias", (tw,), "ln_b"),
        ("text_projection", (tw, d.embed_dim), "tproj"),
        ("logit_scale", (), "logit_scale"),
    ]
    return spec


def _std(kind, d: ClipDims):
    vw, tw = d.vision_width, d.transformer_width
    if kind in ("vscale",):
        return vw ** -0.5
    if kind == "conv":
        return (3 * d.vision_patch_size ** 2) ** -0.5
    if kind == "tok":
        return 0.02
    if kind == "pos":
        return 0.01
    if kind == "tproj":
        return tw ** -0.5
    tower, k = kind[0], kind[1:]
    w = vw if tower == "v" else tw
    layers = d.vision_layers if tower == "v" else d.transformer_layers
    if k == "attn":
        return w ** -0.5
    if k == "proj":
        return (w ** -0.5) * ((2 * layers) ** -0.5)
    if k == "fc":
        return (2 * w) ** -0.5
    raise KeyError(kind)


def init_state_dict(d: ClipDims, seed: int = 0, logit_scale: float = math.log(100.0)):
    """dict[key] -> float32 numpy array, deterministic in (dims, seed)."""
    out = {}
    for key, shape, kind in weight_spec(d):
        sid = rng.stream_id(key)
        base = kind[1:] if kind[0] in "vt" and kind[1:] in ("ln_w", "ln_b", "bias", "attn", "proj", "fc") else kind
        if base == "ln_w":
            a = rng.normal(seed, sid, shape, 1.0, 0.1)
        elif base in ("ln_b", "bias"):
            a = rng.normal(seed, sid, shape, 0.0, 0.05)
        elif base == "logit_scale":
            a = np.array(logit_scale, dtype=np.float32)
        else:
            a = rng.normal(seed, sid, shape, 0.0, _std(kind, d))
        out[key] = a
    return out


def on_f16_grid(sd):
    """The state dict with every matrix operand rounded to the nearest f16 number (kept as float32): what a PUBLISHED CLIP checkpoint holds.  OpenAI's
    archives store the convolution / Linear / attention / projection weights in fp16 (clip.model.convert_weights), and the reference's CPU path --
    clip.load(name, "cpu"), methods/clip_baseline.py:39-41 -- computes in fp32 on those values cast up: its weights ARE f16 numbers.  The seeded
    synthetic init is not; with this applied the f16 towers round no weight (as with a real checkpoint) and the split-f16 tier drops its a_hi w_lo
    product (csrc/gemm_split.hip, GemmArgs::w_exact).  Embeddings tables and vectors (biases, LayerNorm affine) stay as they are: the towers keep
    them in f32 either way."""
    out = {}
    for k, v in sd.items():
        matrix = v.ndim >= 2 and not k.endswith("positional_embedding") and "token_embedding" not in k
        out[k] = v.astype(np.float16).astype(np.float32) if matrix else v
    return out


STRESS_OUTLIER_CHANNELS = (5, 77, 300, 511)     # (taken modulo the vision width)
STRESS_OVERFLOW_GAIN = (100.0, 800.0)           # (last block ln_2 gain, one c_proj output row): that CLS stream channel is ~ N(0, (5e4)^2) at ViT-B/16, a fifth of the images beyond 65 504


def stress_state_dict(d: ClipDims, seed: int = 0, outlier: float = 200.0, overflow_gain=None, channels=None):
    """The synthetic weights of init_state_dict bent towards what trained CLIP checkpoints do to low-precision arithmetic and the seeded random init does
    not: (a) MASSIVE ACTIVATIONS -- four channels of the vision residual stream sit at x ~ +200 on every token (an ln_pre bias, as the
    near-constant outlier channels of trained ViTs are), so the row mean is ~ 1 and the row variance ~ 200 in every later LayerNorm, which stresses the
    statistics, the LayerNorm-folded GEMMs' `acc - mean * colsum` and the f16 stream's 0.125 ulp at 200; as trained models do, the LayerNorms that READ
    the stream carry small gains on those channels and correspondingly larger ones elsewhere, so the blocks compute on inputs of the usual scale.
    (Outliers whose magnitude varies from token to token -- a GAIN on ln_pre instead of a bias -- make the network ill-conditioned in fp32 itself:
    oracle/gen_golden_stress.py measures fp32 against fp64, 3e-2 in cosine for that variant, 2e-7 for this one.)  (b) an F16 OVERFLOW that depends on the
    image -- the last block's ln_2 gain and one output row of its c_proj are scaled until that stream channel of the CLS row leaves the f16 range
    (> 65 504) for roughly a fifth of the images: their f16 embeddings are non-finite (the `nonfinite_screen_rows` path of pseudolabels.refine_scan
    through the REAL towers) while the f32 / split-f16 towers, whose stream is f32, stay finite; the channel does not reach the embedding (ln_post gain
    0).  Every GEMM operand stays far inside the f16 range.  The text tower is untouched.  Test / bench model (`clip.load(..., synthetic="stress")`)."""
    sd = init_state_dict(d, seed)
    vw = d.vision_width
    overflow_gain = STRESS_OVERFLOW_GAIN if overflow_gain is None else overflow_gain
    ch = sorted({c % vw for c in (STRESS_OUTLIER_CHANNELS if channels is None else channels)})
    rest = np.ones(vw, dtype=bool)
    rest[ch] = False
    if ch and outlier:
        b = sd["visual.ln_pre.bias"].copy()
        b[ch] += np.float32(outlier)
        sd["visual.ln_pre.bias"] = b
        std = math.sqrt(1.0 + len(ch) * outlier * outlier / vw)        # what a row's standard deviation becomes
        for key in [k for k in sd if k.startswith("visual.") and (k.endswith("ln_1.weight") or k.endswith("ln_2.weight") or k == "visual.ln_post.weight")]:
            g = sd[key].copy()
            g[ch] *= np.float32(1.0 / outlier)
            g[rest] *= np.float32(std)
            sd[key] = g
    blk = f"visual.transformer.resblocks.{d.vision_layers - 1}"
    sd[blk + ".ln_2.weight"] = sd[blk + ".ln_2.weight"] * np.float32(overflow_gain[0])
    w = sd[blk + ".mlp.c_proj.weight"].copy()
    row = (ch[0] + 1) % vw if ch else 6
    w[row] *= np.float32(overflow_gain[1])
    sd[blk + ".mlp.c_proj.weight"] = w
    g = sd["visual.ln_post.weight"].copy()
    g[row] = 0.0
    sd["visual.ln_post.weight"] = g
    return sd


def init_upt_mixer(coop_dim: int, vpt_dim: int, tdim: int = 128, seed: int = 0):
    """Seeded parameters of UPTModel's prompt mixer (models/prompts_models.py:99-119):
    four nn.Linear and a 1-layer, 1-head CLIP Transformer of width `tdim`."""
    out = {}

    def lin(name, o, i):
        out[f"{name}.weight"] = rng.normal(seed, rng.stream_id("upt." + name + ".weight"), (o, i), 0.0, i ** -0.5)
        out[f"{name}.bias"] = rng.normal(seed, rng.stream_id("upt." + name + ".bias"), (o,), 0.0, 0.05)

    lin("proj_coop_pre", tdim, coop_dim)
    lin("proj_coop_post", coop_dim, tdim)
    lin("proj_vpt_pre", tdim, vpt_dim)
    lin("proj_vpt_post", vpt_dim, tdim)
    for key, shape, kind in _block_keys("transformer.resblocks.0", tdim):
        sid = rng.stream_id("upt." + key)
        if kind == "ln_w":
            a = rng.normal(seed, sid, shape, 1.0, 0.1)
        elif kind in ("ln_b", "bias"):
            a = rng.normal(seed, sid, shape, 0.0, 0.05)
        else:
            a = rng.normal(seed, sid, shape, 0.0, {"attn": tdim ** -0.5, "proj": tdim ** -0.5 * 2 ** -0.5, "fc": (2 * tdim) ** -0.5}[kind])
        out[key] = a
    return out


# ---------------------------------------------------------------------------------------------- OpenAI checkpoints
# clip.load of the published openai/CLIP reads `ViT-B-16.pt` etc. as TorchScript archives (torch.jit.load) and falls back to a
# plain torch.load state_dict; build_model then infers every dimension from the tensor shapes and drops three scalar
# book-keeping entries.  The reference reaches all of that through clip.load(VIS_ENCODER, device) (methods/clip_baseline.py:39-41).
NON_WEIGHT_KEYS = ("input_resolution", "context_length", "vocab_size")


def read_checkpoint(path):
    """state_dict (key -> tensor, on the CPU) of an OpenAI CLIP checkpoint: a TorchScript archive (the files OpenAI publishes),
    a pickled state_dict, or a {"state_dict": ...} wrapper.  f16 tensors stay f16 here; the towers convert on load."""
    import torch
    try:
        sd = torch.jit.load(path, map_location="cpu").state_dict()
    except Exception:
        sd = torch.load(path, map_location="cpu")
        if hasattr(sd, "state_dict") and not isinstance(sd, dict):
            sd = sd.state_dict()
        sd = sd.get("state_dict", sd)
    return {k: v for k, v in sd.items() if k not in NON_WEIGHT_KEYS and not k.endswith("attn_mask")}


def dims_from_state_dict(sd, name="checkpoint") -> ClipDims:
    """Every dimension of a ViT CLIP from its tensor shapes (the inference of the published build_model): width and patch from
    conv1, layers from the resblock keys, resolution from the positional embedding, heads = width / 64."""
    if "visual.proj" not in sd:
        raise RuntimeError("not a ViT CLIP state_dict (ModifiedResNet towers are not supported: the reference only names ViT encoders)")
    vw, _, p, _ = sd["visual.conv1.weight"].shape
    vl = len({k.split(".")[3] for k in sd if k.startswith("visual.transformer.resblocks.") and k.endswith(".attn.in_proj_weight")})
    grid = round((sd["visual.positional_embedding"].shape[0] - 1) ** 0.5)
    tw = sd["ln_final.weight"].shape[0]
    tl = len({k.split(".")[2] for k in sd if k.startswith("transformer.resblocks.") and k.endswith(".attn.in_proj_weight")})
    return ClipDims(name, sd["text_projection"].shape[1], grid * p, vl, vw, p, sd["positional_embedding"].shape[0], sd["token_embedding.weight"].shape[0],
                    tw, tw // 64, tl)


def check_state_dict(sd, d: ClipDims):
    """Raises with a precise message when `sd` is not a complete set of weights for dims `d` (missing / unexpected keys, shapes)."""
    spec = {k: tuple(s) for k, s, _ in weight_spec(d)}
    missing = [k for k in spec if k not in sd]
    extra = [k for k in sd if k not in spec]
    if missing or extra:
        raise RuntimeError(f"state_dict does not match {d.name}: missing {missing[:4]}{'...' if len(missing) > 4 else ''}, "
                           f"unexpected {extra[:4]}{'...' if len(extra) > 4 else ''}")
    for k, shape in spec.items():
        if tuple(sd[k].shape) != shape:
            raise RuntimeError(f"state_dict does not match {d.name}: {k} has shape {tuple(sd[k].shape)}, expected {shape}")
