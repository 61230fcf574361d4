"""Encoder wrappers with the reference's names and signatures (models/clip_encoders.py), running
on the native towers.  Backbone parameters stay frozen; autograd delivers a gradient only to the
prompt tensors."""
import logging

import torch
import torch.nn as nn

from .. import clip
from ..engine import text_prefix_forward, vit_prefix_forward

log = logging.getLogger(__name__)


class TextEncoder(nn.Module):
    """CLIP text encoder (reference :13-22)."""

    def __init__(self, clip_model):
        super().__init__()
        self.clip_model = clip_model

    def forward(self, text):
        return self.clip_model.encode_text(text)


class CustomTextEncoder(nn.Module):
    """Reference :25-90: splice a learnable prefix over token positions 1..P of 'X .. X <class>'."""

    def __init__(self, clip_model, device, dtype):
        super().__init__()
        self.dtype = dtype
        self.clip_model = clip_model
        self.transformer = clip_model.transformer
        self.positional_embedding = clip_model.positional_embedding
        self.ln_final = clip_model.ln_final
        self.text_projection = clip_model.text_projection
        self.token_embedding = clip_model.token_embedding
        self.device = device
        self._tok_cache = {}

    def tokenize(self, text):
        return torch.cat([clip.tokenize(tok) for tok in text])

    def _token_ids(self, n_prefix, classes):
        key = (n_prefix, tuple(classes))
        ids = self._tok_cache.get(key)
        if ids is None:
            prompts = [" ".join([" ".join(["X"] * n_prefix).strip(), c]) for c in classes]   # reference :54-57
            ids = clip.tokenize(prompts).to(self.device)
            if len(self._tok_cache) > 64:
                self._tok_cache.clear()
            self._tok_cache[key] = ids
        return ids

    def forward(self, class_embeddings, classes, enable_pos_emb=True):
        token_ids = self._token_ids(class_embeddings.size()[1], classes)
        return text_prefix_forward(self.clip_model.text_tower, token_ids, class_embeddings, pos_emb=bool(enable_pos_emb))   # :70-74


class ImageEncoder(nn.Module):
    """CLIP image encoder (reference :93-102)."""

    def __init__(self, clip_model):
        super().__init__()
        self.clip_model = clip_model

    def forward(self, text):
        return self.clip_model.encode_image(text)


class CustomVisionTransformer(nn.Module):
    """Reference :105-194: ViT with a visual prompt inserted between CLS and the patches."""

    def __init__(self, vision_transformer):
        super().__init__()
        self.input_resolution = vision_transformer.input_resolution
        self.output_dim = vision_transformer.output_dim
        self.conv1 = vision_transformer.conv1
        self.class_embedding = vision_transformer.class_embedding
        self.positional_embedding = vision_transformer.positional_embedding
        self.ln_pre = vision_transformer.ln_pre
        self.transformer = vision_transformer.transformer
        self.ln_post = vision_transformer.ln_post
        self.proj = vision_transformer.proj
        self._vt = [vision_transformer]

    def forward(self, x, image_prefix, pos_emb=True, deep_embs=None):
        if deep_embs is not None:
            # reference :158-174 reads self.visual / self.mvlpt_model, attributes this class never has: the branch raises
            # AttributeError upstream as well (VPT_DEEP is False in every config)
            raise NotImplementedError("deep prompts are unreachable in the reference (VPT_DEEP: False; :158-174 reads attributes that do not exist)")
        return vit_prefix_forward(self._vt[0].tower, x, image_prefix, pos_emb=bool(pos_emb))   # :141


class CustomImageEncoder(nn.Module):
    """Reference :198-208."""

    def __init__(self, visual):
        super().__init__()
        self.visual = CustomVisionTransformer(visual)
        self.dtype = self.visual.conv1.weight.dtype

    def forward(self, image, prefix, deep_embds=None):
        return self.visual(image, prefix, deep_embs=deep_embds)
