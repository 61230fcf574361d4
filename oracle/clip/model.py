"""ORACLE (test infrastructure) -- CPU fp32 restatement of the openai/CLIP
ViT + text-transformer arithmetic the reference drives.

The arithmetic is NOT in /root/reference: it lives in the third-party package
`clip` (requirements.txt:2, git+https://github.com/openai/CLIP.git, un-pinned),
which is absent from this image.  This file restates its published
architecture (clip/model.py upstream) with the attribute surface the
reference touches:
  models/clip_encoders.py:33-37   transformer, positional_embedding, ln_final,
                                  text_projection, token_embedding
  models/clip_encoders.py:108-119 visual.{input_resolution, output_dim, conv1,
                                  class_embedding, positional_embedding, ln_pre,
                                  transformer, ln_post, proj}
  models/prompts_models.py:116    clip.model.Transformer(width, layers, heads)
  utils/clip_pseudolabels.py:35   clip_model(image, text) -> (logits_per_image, logits_per_text)
  methods/*/textual_prompt.py:100 encode_image / encode_text / logit_scale
Parameter names equal the OpenAI state_dict keys so real weights load unchanged.
Attention is written out explicitly (packed in_proj, per-head softmax(QK^T/sqrt(dh))V,
out_proj) instead of calling nn.MultiheadAttention, so it is an independent
statement of the math; tests/test_oracle_vs_hf.py cross-checks it against
transformers.CLIPModel.
"""
import math
from collections import OrderedDict

import torch
import torch.nn.functional as F
from torch import nn


class LayerNorm(nn.LayerNorm):
    """fp32 LayerNorm, cast back to the input dtype (eps 1e-5)."""

    def forward(self, x: torch.Tensor):
        orig_type = x.dtype
        ret = super().forward(x.type(torch.float32))
        return ret.type(orig_type)


class QuickGELU(nn.Module):
    def forward(self, x: torch.Tensor):
        return x * torch.sigmoid(1.702 * x)


class _OutProj(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(d, d))
        self.bias = nn.Parameter(torch.zeros(d))


class PackedSelfAttention(nn.Module):
    """Self-attention on LND input with nn.MultiheadAttention's parameter names."""

    def __init__(self, d_model: int, n_head: int):
        super().__init__()
        self.d_model, self.n_head = d_model, n_head
        self.in_proj_weight = nn.Parameter(torch.empty(3 * d_model, d_model))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * d_model))
        self.out_proj = _OutProj(d_model)
        nn.init.normal_(self.in_proj_weight, std=d_model ** -0.5)
        nn.init.normal_(self.out_proj.weight, std=d_model ** -0.5)

    def forward(self, x, attn_mask=None):
        L, N, D = x.shape
        H, dh = self.n_head, D // self.n_head
        qkv = F.linear(x, self.in_proj_weight, self.in_proj_bias)  # [L,N,3D]
        q, k, v = qkv.split(D, dim=-1)
        # [N*H, L, dh]
        q = q.reshape(L, N * H, dh).transpose(0, 1) * (dh ** -0.5)
        k = k.reshape(L, N * H, dh).transpose(0, 1)
        v = v.reshape(L, N * H, dh).transpose(0, 1)
        s = torch.bmm(q, k.transpose(1, 2))
        if attn_mask is not None:
            s = s + attn_mask.to(dtype=s.dtype, device=s.device)
        p = torch.softmax(s, dim=-1)
        o = torch.bmm(p, v).transpose(0, 1).reshape(L, N, D)
        return F.linear(o, self.out_proj.weight, self.out_proj.bias)


class ResidualAttentionBlock(nn.Module):
    def __init__(self, d_model: int, n_head: int, attn_mask: torch.Tensor = None):
        super().__init__()
        self.attn = PackedSelfAttention(d_model, n_head)
        self.ln_1 = LayerNorm(d_model)
        self.mlp = nn.Sequential(OrderedDict([
            ("c_fc", nn.Linear(d_model, d_model * 4)),
            ("gelu", QuickGELU()),
            ("c_proj", nn.Linear(d_model * 4, d_model)),
        ]))
        self.ln_2 = LayerNorm(d_model)
        self.attn_mask = attn_mask

    def forward(self, x: torch.Tensor):
        x = x + self.attn(self.ln_1(x), self.attn_mask)
        x = x + self.mlp(self.ln_2(x))
        return x


class Transformer(nn.Module):
    def __init__(self, width: int, layers: int, heads: int, attn_mask: torch.Tensor = None):
        super().__init__()
        self.width = width
        self.layers = layers
        self.resblocks = nn.Sequential(*[ResidualAttentionBlock(width, heads, attn_mask) for _ in range(layers)])

    def forward(self, x: torch.Tensor):
        return self.resblocks(x)


class VisionTransformer(nn.Module):
    def __init__(self, input_resolution, patch_size, width, layers, heads, output_dim):
        super().__init__()
        self.input_resolution = input_resolution
        self.output_dim = output_dim
        self.conv1 = nn.Conv2d(3, width, kernel_size=patch_size, stride=patch_size, bias=False)
        scale = width ** -0.5
        self.class_embedding = nn.Parameter(scale * torch.randn(width))
        self.positional_embedding = nn.Parameter(scale * torch.randn((input_resolution // patch_size) ** 2 + 1, width))
        self.ln_pre = LayerNorm(width)
        self.transformer = Transformer(width, layers, heads)
        self.ln_post = LayerNorm(width)
        self.proj = nn.Parameter(scale * torch.randn(width, output_dim))

    def forward(self, x: torch.Tensor):
        x = self.conv1(x)
        x = x.reshape(x.shape[0], x.shape[1], -1).permute(0, 2, 1)
        cls = self.class_embedding.to(x.dtype) + torch.zeros(x.shape[0], 1, x.shape[-1], dtype=x.dtype, device=x.device)
        x = torch.cat([cls, x], dim=1)
        x = x + self.positional_embedding.to(x.dtype)
        x = self.ln_pre(x)
        x = x.permute(1, 0, 2)
        x = self.transformer(x)
        x = x.permute(1, 0, 2)
        x = self.ln_post(x[:, 0, :])
        if self.proj is not None:
            x = x @ self.proj
        return x


class CLIP(nn.Module):
    def __init__(self, embed_dim, image_resolution, vision_layers, vision_width, vision_patch_size,
                 context_length, vocab_size, transformer_width, transformer_heads, transformer_layers):
        super().__init__()
        self.context_length = context_length
        self.visual = VisionTransformer(image_resolution, vision_patch_size, vision_width, vision_layers,
                                        vision_width // 64, embed_dim)
        self.transformer = Transformer(transformer_width, transformer_layers, transformer_heads,
                                       attn_mask=self.build_attention_mask())
        self.vocab_size = vocab_size
        self.token_embedding = nn.Embedding(vocab_size, transformer_width)
        self.positional_embedding = nn.Parameter(torch.empty(context_length, transformer_width))
        self.ln_final = LayerNorm(transformer_width)
        self.text_projection = nn.Parameter(torch.empty(transformer_width, embed_dim))
        self.logit_scale = nn.Parameter(torch.ones([]) * math.log(1 / 0.07))

    def build_attention_mask(self):
        mask = torch.empty(self.context_length, self.context_length)
        mask.fill_(float("-inf"))
        mask.triu_(1)
        return mask

    @property
    def dtype(self):
        return self.visual.conv1.weight.dtype

    def encode_image(self, image):
        return self.visual(image.type(self.dtype))

    def encode_text(self, text):
        x = self.token_embedding(text).type(self.dtype)
        x = x + self.positional_embedding.type(self.dtype)
        x = x.permute(1, 0, 2)
        x = self.transformer(x)
        x = x.permute(1, 0, 2)
        x = self.ln_final(x).type(self.dtype)
        x = x[torch.arange(x.shape[0]), text.argmax(dim=-1)] @ self.text_projection
        return x

    def forward(self, image, text):
        image_features = self.encode_image(image)
        text_features = self.encode_text(text)
        image_features = image_features / image_features.norm(dim=1, keepdim=True)
        text_features = text_features / text_features.norm(dim=1, keepdim=True)
        logit_scale = self.logit_scale.exp()
        logits_per_image = logit_scale * image_features @ text_features.t()
        logits_per_text = logits_per_image.t()
        return logits_per_image, logits_per_text
