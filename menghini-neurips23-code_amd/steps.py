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


class GraphedCoopStep:
    """coop_step with its forward + backward captured ONCE in a HIP graph and replayed per step (shapes are static within an
    epoch: same batch size, same class list).  The CoOp step is ~330 launches of 10-25 us on two streams; replaying them
    as one graph removes the per-launch host cost and the launch gaps, nothing else changes: the captured kernels are the
    ones coop_step launches, the prompt-gradient all-reduce and the optimizer step stay outside the graph (eager), so the
    same object serves one GPU and N.  A batch of another size falls back to the eager step.

    Static inputs: images [B,3,R,R] / labels [B] / row weights [B] are copied into fixed buffers before each replay; the
    prompt parameter is read in place, its .grad is written in place."""

    def __init__(self, model, clip_model, optimizer):
        self.model, self.clip_model, self.optimizer = model, clip_model, optimizer
        self.scale = clip_model.logit_scale.exp().item()
        self.graph = None
        self.key = None

    def _body(self):
        side = _side_stream(self.x.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            image_features = self.clip_model.encode_image(self.x)
        text_features = self.model(self.model.classes)
        torch.cuda.current_stream().wait_stream(side)
        logits = CosineHeadFn.apply(image_features, text_features, self.scale)
        loss = WeightedCEFn.apply(logits, self.y, self.w)
        loss.backward()
        return loss.detach()

    def _capture(self, images, labels, row_weight):
        dev = images.device
        self.x, self.y, self.w = images.clone(), labels.to(torch.int32).clone(), row_weight.to(torch.float32).clone()
        self.key = (tuple(images.shape), images.dtype, tuple(self.model.classes))
        prefix = self.model.prefix
        warm = torch.cuda.Stream(device=dev)
        warm.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(warm):              # warm-up off the default stream: workspaces, lazy kernel attributes, cached token ids
            for _ in range(2):
                prefix.grad = None
                self._body()
        torch.cuda.current_stream().wait_stream(warm)
        torch.cuda.synchronize()
        prefix.grad = torch.zeros_like(prefix)     # the captured backward accumulates into this buffer
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            prefix.grad.zero_()
            self.loss = self._body()
        self.grad = prefix.grad

    def __call__(self, images, labels, row_weight):
        key = (tuple(images.shape), images.dtype, tuple(self.model.classes))
        if self.graph is None:
            self._capture(images, labels, row_weight)
        if key != self.key:
            self.model.prefix.grad = None          # (the graph's gradient buffer holds the previous replay's values)
            return coop_step(self.model, self.clip_model, images, labels, row_weight, self.optimizer)
        self.x.copy_(images)
        self.y.copy_(labels)
        self.w.copy_(row_weight)
        self.model.prefix.grad = self.grad
        self.graph.replay()
        gdist.allreduce_mean_([self.grad])
        self.optimizer.step()
        return self.loss
