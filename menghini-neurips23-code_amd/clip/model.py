"""Native-engine CLIP with the attribute surface of `clip.model.CLIP` that the reference touches
(SURVEY.md 8b): visual.{input_resolution, output_dim, conv1, class_embedding, positional_embedding,
ln_pre, transformer, ln_post, proj}, transformer, token_embedding, positional_embedding, ln_final,
text_projection, logit_scale, encode_image, encode_text, __call__.

Every parameter is a frozen view into the tower's HBM weight blob (OpenAI state_dict key names, so
real checkpoints load with load_state_dict); the forward passes run in libgrip_amd.so.
"""
import math

import torch
from torch import nn

from .. import engine
from ..config import ClipDims


class _Holder(nn.Module):
    """Name-space node so that named_parameters() yields the OpenAI key names."""


def _attach(root, dotted, param):
    node = root
    parts = dotted.split(".")
    for p in parts[:-1]:
        if not hasattr(node, p):
            node.add_module(p, _Holder())
        node = getattr(node, p)
    node.register_parameter(parts[-1], param)


def _frozen(t):
    return nn.Parameter(t, requires_grad=False)


class _TowerModule(nn.Module):
    def _bind(self, tower):
        self._tower = [tower]   # list: keep the Tower out of nn.Module's registry
        for name in tower.primary_names():
            v = tower.view(name)
            if v.shape[0] == 1:
                v = v[0]
            if name == "conv1.weight":
                d, p = tower.width, tower.dims.patch
                _, _, off, rows, cols, ld = tower.slots[name]
                v = tower.blob16.as_strided((d, 3, p, p), (ld, p * p, p, 1), off)
            _attach(self, name, _frozen(v))

    @property
    def tower(self):
        return self._tower[0]

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)
        if self.tower is not None:
            self.tower._finalized = False


class VisionTransformer(_TowerModule):
    """clip_model.visual: frozen ViT on the native vision tower."""

    def __init__(self, d: ClipDims, device, exact=False):
        super().__init__()
        self.input_resolution = d.image_resolution
        self.output_dim = d.embed_dim
        self._bind(engine.vision_tower(d, device, exact=exact))

    def forward(self, x: torch.Tensor, prefix=None):
        return self.tower.vit_forward(x, prefix)[0]


class _Embedding(_Holder):
    def forward(self, ids):
        return torch.nn.functional.embedding(ids.long(), self.weight)


class Transformer(nn.Module):
    """clip.model.Transformer(width, layers, heads): the small TRAINABLE prompt mixer of
    UPTModel (models/prompts_models.py:116-119; width 128, 1 layer, 1 head, input [2, P, 128]).
    < 1 MFLOP per step and it needs weight gradients, so it stays a torch module (plumbing scale);
    the frozen CLIP towers never go through this class."""

    class _Block(nn.Module):
        def __init__(self, d, h):
            super().__init__()
            self.attn = nn.MultiheadAttention(d, h)
            self.ln_1 = nn.LayerNorm(d)
            self.mlp = nn.Sequential()
            self.mlp.add_module("c_fc", nn.Linear(d, 4 * d))
            self.mlp.add_module("c_proj", nn.Linear(4 * d, d))
            self.ln_2 = nn.LayerNorm(d)

        def forward(self, x):
            y = self.ln_1(x.float()).to(x.dtype)
            x = x + self.attn(y, y, y, need_weights=False)[0]
            y = self.mlp.c_fc(self.ln_2(x.float()).to(x.dtype))
            return x + self.mlp.c_proj(y * torch.sigmoid(1.702 * y))

    def __init__(self, width: int, layers: int, heads: int, attn_mask=None):
        super().__init__()
        if attn_mask is not None:
            raise NotImplementedError("masked Transformer is only used inside the native text tower")
        self.width, self.layers = width, layers
        self.resblocks = nn.Sequential(*[Transformer._Block(width, heads) for _ in range(layers)])

    def forward(self, x):
        return self.resblocks(x)


class CLIP(_TowerModule):
    def __init__(self, d: ClipDims, device="cuda", exact=False, vision_only=False):
        super().__init__()
        self.dims = d
        self.precision = int(exact)        # 0 f16 towers, 1 f32 towers, 2 split-f16 towers (include/grip_amd.h: grip_dims.precision)
        self.exact = self.precision == 1   # f32 towers (comparison mode): `dtype` is float32, like the reference's clip.load on a CPU
        self.context_length = d.context_length
        self.vocab_size = d.vocab_size
        self.visual = VisionTransformer(d, device, exact=exact)
        self.vision_only = bool(vision_only)     # the split-f16 twin: the middle tier re-encodes IMAGES only (no text tower: its blob + #S copies would sit unused)
        if vision_only:
            self._tower = [None]
        else:
            self.add_module("token_embedding", _Embedding())
            self._bind(engine.text_tower(d, device, exact=exact))
        self.logit_scale = _frozen(torch.ones([], device=device) * math.log(1 / 0.07))
        self._twin = [None, None]       # [the exact (f32) twin, a callable that builds it]: set by clip.load, out of nn.Module's registry
        self._split = [None, None]      # the same for the split-f16 twin

    def exact_twin(self):
        """The same model with f32 towers (weights from the same checkpoint, NOT re-derived from this model's f16 blobs): what
        the screen-and-refine pseudolabel pass re-encodes its undecidable rows with.  Built on first use (ViT-B/16: 0.7 GB)."""
        if self.exact:
            return self
        if self._twin[0] is None:
            if self._twin[1] is None:
                raise engine.native.GripError("this CLIP was not created by clip.load: no source for its exact twin")
            self._twin[0] = self._twin[1]()
        return self._twin[0]

    def split_twin(self):
        """The same model with split-f16 towers (precision 2: every GEMM operand carried as an f16 hi / lo pair, three f16 MFMA products,
        f32 everywhere else): the MIDDLE tier of the screen-and-refine pseudolabel pass -- probabilities within ~1e-5 of the f32 twin's at
        several times its throughput.  None when the model has no source for it (pseudolabels.mid_tower decides whether a pool uses it)."""
        if self.precision != 0 or self._split[1] is None:
            return None
        if self._split[0] is None:
            self._split[0] = self._split[1]()
        return self._split[0]

    @property
    def text_tower(self):
        return self.tower

    @property
    def dtype(self):
        return self.visual.conv1.weight.dtype

    def encode_image(self, image):
        return self.visual(image)

    def encode_text(self, text):
        if self.tower is None:
            raise engine.native.GripError("this CLIP holds a vision tower only (the split-f16 twin of the refinement's middle tier)")
        return self.tower.text_forward(text)[0]

    def forward(self, image, text):
        """(logits_per_image, logits_per_text), as utils/clip_pseudolabels.py:35 consumes them."""
        img = self.encode_image(image)
        txt = self.encode_text(text)
        logits, _, _, _ = engine.cosine_head(img, txt, self.logit_scale.exp().item(), want_probs=False)
        return logits, logits.t()
