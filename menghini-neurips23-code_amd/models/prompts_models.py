"""Prompt models with the reference's names, constructor/forward signatures and parameter names
(models/prompts_models.py): they own the only trainable tensors."""
import logging

import torch
from torch import nn

from .. import clip

log = logging.getLogger(__name__)


_SIDE = {}


def _side_stream(device):
    key = torch.device(device).index or 0
    if key not in _SIDE:
        _SIDE[key] = torch.cuda.Stream(device=device)
    return _SIDE[key]


class _ModuleShim:
    """Reference callers reach `.module.classes` whenever torch.cuda.is_available() (DDP-wrapped
    model, e.g. methods/semi_supervised_learning/textual_prompt.py:94-97); PyTorch-ROCm reports
    cuda available, so an un-wrapped model must answer to `.module` too."""

    @property
    def module(self):
        return self


class TextPrefixModel(_ModuleShim, nn.Module):
    def __init__(self, initial_prefix, text_encoder, classes, temperature=0.07, device="cpu"):
        super().__init__()
        self.device = device
        self.initialized_prefix = initial_prefix
        self.classes = classes
        self.prefix = nn.Parameter(initial_prefix)
        self.text_encoder = text_encoder

    def forward(self, classes):
        return self.text_encoder(self.prefix, classes)     # un-normalised, as reference :31-36


class ImagePrefixModel(_ModuleShim, nn.Module):
    def __init__(self, initial_prefix, image_encoder, temperature=0.07, device="cpu"):
        super().__init__()
        self.device = device
        self.initialized_prefix = initial_prefix
        self.prefix = nn.Parameter(initial_prefix)
        self.image_encoder = image_encoder

    def forward(self, x):
        return self.image_encoder(x, self.prefix)          # un-normalised, as reference :55-61


class UPTModel(_ModuleShim, nn.Module):
    def __init__(self, coop_embeddings, vpt_embeddings, vpt_embeddings_deep, image_encoder, text_encoder, classes,
                 dim_transformer, temperature=0.07, device="cpu", dtype=torch.float32):
        super().__init__()
        self.device = device
        self.classes = classes
        self.temperature = temperature
        self.dtype = dtype
        self.coop_embeddings = nn.Parameter(coop_embeddings)
        self.vpt_embeddings = nn.Parameter(vpt_embeddings)
        self.coop_length, self.coop_dim = self.coop_embeddings.size()[1], self.coop_embeddings.size()[2]
        self.vpt_length, self.vpt_dim = self.vpt_embeddings.size()[1], self.vpt_embeddings.size()[2]
        self.vpt_embeddings_deep = nn.Parameter(vpt_embeddings_deep) if vpt_embeddings_deep is not None else None
        self.proj_coop_pre = nn.Linear(self.coop_dim, dim_transformer, dtype=self.dtype).to(self.device)
        self.proj_coop_post = nn.Linear(dim_transformer, self.coop_dim, dtype=self.dtype).to(self.device)
        self.proj_vpt_pre = nn.Linear(self.vpt_dim, dim_transformer, dtype=self.dtype).to(self.device)
        self.proj_vpt_post = nn.Linear(dim_transformer, self.vpt_dim, dtype=self.dtype).to(self.device)
        self.transformer = clip.model.Transformer(width=dim_transformer, layers=1, heads=1).to(self.device)
        self.image_encoder = image_encoder
        self.text_encoder = text_encoder

    def _native_mixer_ok(self):
        """The native mixer covers what the reference builds (:99-119): one block, one head, on the GPU, float32 -- or the float16 branch
        of multimodal_prompt.py:46 (fp16 prompt embeddings and projection Linears around the fp32 block: the same kernels with fp16 rounding
        points, grip_upt_mixer.half_linears).  Any other shape runs the same arithmetic through the framework's kernels."""
        import os
        t = self.transformer
        dtypes_ok = all(p.dtype == self.dtype for p in (self.coop_embeddings, self.vpt_embeddings, self.proj_coop_pre.weight, self.proj_vpt_post.weight)) \
            and t.resblocks[0].ln_1.weight.dtype == torch.float32
        return (self.dtype in (torch.float32, torch.float16) and dtypes_ok and self.coop_embeddings.is_cuda and os.environ.get("GRIP_NATIVE_MIXER", "1") != "0"
                and getattr(t, "layers", 0) == 1 and t.resblocks[0].attn.num_heads == 1 and len(self.coop_embeddings) == 1
                and self.coop_length == self.vpt_length and self.coop_length <= 16 and t.width % 64 == 0 and t.width <= 256)

    def mix(self):
        """Reference :129-146 (incl. the fp32 -> fp16 -> dtype round trip of :138-145)."""
        if self._native_mixer_ok():
            from ..engine import UptMixerFn
            b = self.transformer.resblocks[0]
            coop_embs, vpt_embs = UptMixerFn.apply(
                self.coop_embeddings, self.vpt_embeddings, self.proj_coop_pre.weight, self.proj_coop_pre.bias, self.proj_vpt_pre.weight,
                self.proj_vpt_pre.bias, b.ln_1.weight, b.ln_1.bias, b.attn.in_proj_weight, b.attn.in_proj_bias, b.attn.out_proj.weight,
                b.attn.out_proj.bias, b.ln_2.weight, b.ln_2.bias, b.mlp.c_fc.weight, b.mlp.c_fc.bias, b.mlp.c_proj.weight, b.mlp.c_proj.bias,
                self.proj_coop_post.weight, self.proj_coop_post.bias, self.proj_vpt_post.weight, self.proj_vpt_post.bias)
            return (coop_embs.reshape(-1, self.coop_length, self.coop_dim).to(self.dtype), vpt_embs.reshape(-1, self.vpt_length, self.vpt_dim).to(self.dtype))
        coop = self.proj_coop_pre(self.coop_embeddings)
        vpt = self.proj_vpt_pre(self.vpt_embeddings)
        seq = torch.cat((coop, vpt), dim=0).to(torch.float32)
        out = self.transformer(seq).to(torch.float16)
        n = len(self.coop_embeddings)
        coop_embs = self.proj_coop_post(out[:n].to(self.dtype)).reshape(-1, self.coop_length, self.coop_dim)
        vpt_embs = self.proj_vpt_post(out[n:].to(self.dtype)).reshape(-1, self.vpt_length, self.vpt_dim)
        return coop_embs, vpt_embs

    def forward(self, x, classes):
        coop_embs, vpt_embs = self.mix()
        if x.is_cuda:
            # the two towers are independent until the head: the image tower (forward and, through autograd's
            # stream tracking, its backward) runs on a side stream next to the text tower
            main = torch.cuda.current_stream()
            side = _side_stream(x.device)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                visual_out = self.image_encoder(x, vpt_embs)
            vpt_embs.record_stream(side)
            text_out = self.text_encoder(coop_embs, classes)
            main.wait_stream(side)
            visual_out.record_stream(main)
            return text_out, visual_out
        text_out = self.text_encoder(coop_embs, classes)
        visual_out = self.image_encoder(x, vpt_embs)
        return text_out, visual_out
