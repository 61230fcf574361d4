"""Prompt-tuning steps on the native engine (forward + input-gradient backward + SGD on the prompt
tensors only), shaped after the reference's `_train_epoch` bodies:
  CoOp  methods/semi_supervised_learning/textual_prompt.py:92-135
  VPT   methods/unsupervised_learning/visual_prompt.py:113-140
  UPT   methods/transductive_zsl/multimodal_prompt.py:98-133
The per-sample `.item()` host syncs of the reference loops are gone: labels stay on the device.
"""
import torch

from . import dist as gdist
from .engine import CosineHeadFn, WeightedCEFn


def fpl_row_weights(is_pseudo, gamma_seen=1.0, gamma_pseudo=1.0):
    """Per-row weights that turn `gamma_a * CE(rows_a) + gamma_b * CE(rows_b)` (each CE a mean over
    its own rows) into one weighted sum: the FPL losses of
    methods/semi_supervised_learning/textual_fpl.py:117-165 (gamma_seen = |unseen|/|seen|, gamma_pseudo = 1),
    methods/transductive_zsl/textual_fpl.py:117-147 (gamma_seen = 1, gamma_pseudo = |seen|/|unseen|) and
    methods/unsupervised_learning/visual_fpl.py:107-122 (all rows one group)."""
    is_pseudo = torch.as_tensor(is_pseudo, dtype=torch.bool)
    n_p = int(is_pseudo.sum())
    n_s = is_pseudo.numel() - n_p
    w = torch.zeros(is_pseudo.numel(), dtype=torch.float32)
    if n_s:
        w[~is_pseudo] = gamma_seen / n_s
    if n_p:
        w[is_pseudo] = gamma_pseudo / n_p
    return w


_SIDE = {}


def _side_stream(device):
    key = torch.device(device).index or 0
    if key not in _SIDE:
        _SIDE[key] = torch.cuda.Stream(device=device)
    return _SIDE[key]


def _finish(loss, params, optimizer):
    loss.backward()
    gdist.allreduce_mean_([p.grad for p in params if p.grad is not None])
    optimizer.step()
    optimizer.zero_grad(set_to_none=False)
    return loss.detach()


def coop_step(model, clip_model, images, labels, row_weight, optimizer, image_features=None):
    """Textual prompt step: text tower forward+backward over all class prompts, frozen image tower
    forward only (or cached features)."""
    side = None
    if image_features is None:
        # the frozen image tower does not depend on the prompt: run it on a second stream next to the text
        # tower's forward (both are small-batch launches that leave CUs idle on their own)
        side = _side_stream(images.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            image_features = clip_model.encode_image(images)
    text_features = model(model.classes)
    if side is not None:
        torch.cuda.current_stream().wait_stream(side)
        image_features.record_stream(torch.cuda.current_stream())
    logits = CosineHeadFn.apply(image_features, text_features, clip_model.logit_scale.exp().item())
    loss = WeightedCEFn.apply(logits, labels, row_weight)
    return _finish(loss, [model.prefix], optimizer)


def vpt_step(model, text_features, logit_scale, images, labels, row_weight, optimizer):
    """Visual prompt step: image tower forward+backward; text features fixed for the epoch."""
    image_features = model(images)
    logits = CosineHeadFn.apply(image_features, text_features, logit_scale)
    loss = WeightedCEFn.apply(logits, labels, row_weight)
    return _finish(loss, [model.prefix], optimizer)


def upt_step(model, logit_scale, images, labels, row_weight, optimizer):
    """Multimodal prompt step: mixer + both towers forward and backward."""
    text_features, image_features = model(images, model.classes)
    logits = CosineHeadFn.apply(image_features, text_features, logit_scale)
    loss = WeightedCEFn.apply(logits, labels, row_weight)
    return _finish(loss, [p for p in model.parameters() if p.requires_grad], optimizer)
