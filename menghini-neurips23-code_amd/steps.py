"""Prompt-tuning steps on the native engine (forward + input-gradient backward + SGD on the prompt
tensors only), shaped after the reference's `_train_epoch` bodies:
  CoOp  methods/semi_supervised_learning/textual_prompt.py:92-135
  VPT   methods/unsupervised_learning/visual_prompt.py:113-140
  UPT   methods/transductive_zsl/multimodal_prompt.py:98-133
The per-sample `.item()` host syncs of the reference loops are gone: labels stay on the device.
"""
import torch

from . import dist as gdist
from . import engine
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


def _finish(loss, params, optimizer, logits=None):
    loss.backward()
    gdist.allreduce_mean_([p.grad for p in params if p.grad is not None])
    optimizer.step()
    optimizer.zero_grad(set_to_none=True)       # the next backward adopts its gradient tensors (no fill, no add kernels)
    return loss.detach() if logits is None else (loss.detach(), logits.detach())


def coop_step(model, clip_model, images, labels, row_weight, optimizer, image_features=None, return_logits=False):
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
    return _finish(loss, [model.prefix], optimizer, logits if return_logits else None)


def vpt_step(model, text_features, logit_scale, images, labels, row_weight, optimizer, return_logits=False):
    """Visual prompt step: image tower forward+backward; text features fixed for the epoch."""
    image_features = model(images)
    logits = CosineHeadFn.apply(image_features, text_features, logit_scale)
    loss = WeightedCEFn.apply(logits, labels, row_weight)
    return _finish(loss, [model.prefix], optimizer, logits if return_logits else None)


def upt_step(model, logit_scale, images, labels, row_weight, optimizer, return_logits=False):
    """Multimodal prompt step: mixer + both towers forward and backward."""
    text_features, image_features = model(images, model.classes)
    logits = CosineHeadFn.apply(image_features, text_features, logit_scale)
    loss = WeightedCEFn.apply(logits, labels, row_weight)
    return _finish(loss, [p for p in model.parameters() if p.requires_grad], optimizer, logits if return_logits else None)


class GraphedStep:
    """A prompt step whose forward + backward is captured ONCE in a HIP graph and replayed per step (shapes are static within an
    epoch: same batch size, same class list).  A CoOp step is ~330 launches of 10-25 us on two streams, a UPT step adds the
    mixer's dozens of tiny torch kernels; replaying them as one graph removes the per-launch host cost, nothing else changes: the
    captured kernels are the ones the eager step launches, the prompt-gradient all-reduce and the optimizer step stay outside
    the graph (eager), so the same object serves one GPU and N.  A batch of another shape falls back to the eager step.

    Static inputs: images [B,3,R,R] / labels [B] / row weights [B] are copied into fixed buffers before each replay; the
    trainable parameters are read in place, their .grad buffers are written in place.  Subclasses give `forward_logits`
    (static inputs -> logits), `params` and the eager fallback."""

    def __init__(self, optimizer):
        self.optimizer = optimizer
        self.graph = None
        self.key = None
        self._pins = []

    def __del__(self):
        try:
            engine.release_pins(self._pins)
        except Exception:
            pass

    # -- subclass interface
    def params(self):
        raise NotImplementedError

    def forward_logits(self):
        raise NotImplementedError

    def eager(self, images, labels, row_weight):
        raise NotImplementedError

    def shape_key(self, images):
        return (tuple(images.shape), images.dtype)

    # -- machinery
    def _body(self):
        logits = self.forward_logits()
        self.logits = logits.detach()      # [B, C] of the step just run (a static buffer of the graph; the eager fallback's own tensor)
        loss = WeightedCEFn.apply(logits, self.y, self.w)
        loss.backward()
        return loss.detach()

    def _capture(self, images, labels, row_weight):
        dev = images.device
        self.x, self.y, self.w = images.clone(), labels.to(torch.int32).clone(), row_weight.to(torch.float32).clone()
        self.key = self.shape_key(images)
        ps = self.params()
        warm = torch.cuda.Stream(device=dev)
        warm.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(warm):              # warm-up off the default stream: workspaces, lazy kernel attributes, cached token ids
            for _ in range(2):
                for p in ps:
                    p.grad = None
                self._body()
        torch.cuda.current_stream().wait_stream(warm)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        # the workspaces the captured forwards take are dedicated to this graph (never handed to an eager forward while it lives)
        with engine.pin_workspaces() as self._pins, torch.cuda.graph(self.graph):
            # .grad = None inside the capture: autograd then ADOPTS each gradient tensor the backward produces (memory of the graph's
            # private pool, rewritten in place by every replay) instead of zero-filling a buffer and adding into it -- for the UPT
            # step that was one fill and one add kernel per mixer parameter, 44 launches of ~2 us in a 3.4-ms step (r03)
            for p in ps:
                p.grad = None
            self.loss = self._body()
        self._graph_logits = self.logits           # the graph's static output buffer
        for p in ps:
            if p.grad is None:                     # (a parameter the loss does not reach)
                p.grad = torch.zeros_like(p)
        self.grads = [p.grad for p in ps]

    def __call__(self, images, labels, row_weight):
        if self.graph is None:
            self._capture(images, labels, row_weight)
        if self.shape_key(images) != self.key:
            for p in self.params():
                p.grad = None                      # (the graph's gradient buffers hold the previous replay's values)
            loss, self.logits = self.eager(images, labels, row_weight)
            return loss
        self.x.copy_(images)
        self.y.copy_(labels)
        self.w.copy_(row_weight)
        for p, g in zip(self.params(), self.grads):
            p.grad = g
        self.graph.replay()
        self.logits = self._graph_logits
        gdist.allreduce_mean_(self.grads)
        self.optimizer.step()
        return self.loss


class GraphedCoopStep(GraphedStep):
    """coop_step (textual prompt: text tower forward + backward, frozen image tower on a side stream) replayed from a HIP graph."""

    def __init__(self, model, clip_model, optimizer):
        super().__init__(optimizer)
        self.model, self.clip_model = model, clip_model
        self.scale = clip_model.logit_scale.exp().item()

    def params(self):
        return [self.model.prefix]

    def shape_key(self, images):
        return (tuple(images.shape), images.dtype, tuple(self.model.classes))

    def forward_logits(self):
        side = _side_stream(self.x.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            image_features = self.clip_model.encode_image(self.x)
        text_features = self.model(self.model.classes)
        torch.cuda.current_stream().wait_stream(side)
        return CosineHeadFn.apply(image_features, text_features, self.scale)

    def eager(self, images, labels, row_weight):
        return coop_step(self.model, self.clip_model, images, labels, row_weight, self.optimizer, return_logits=True)


class GraphedCoopFeatureStep(GraphedStep):
    """coop_step on image features that were encoded ahead (lookahead_image_features): the text tower's forward + backward, the
    cosine head and the loss, replayed from a HIP graph; the static input is the feature block [B, embed_dim]."""

    def __init__(self, model, clip_model, optimizer):
        super().__init__(optimizer)
        self.model, self.clip_model = model, clip_model
        self.scale = clip_model.logit_scale.exp().item()

    def params(self):
        return [self.model.prefix]

    def shape_key(self, feats):
        return (tuple(feats.shape), feats.dtype, tuple(self.model.classes))

    def forward_logits(self):
        return CosineHeadFn.apply(self.x, self.model(self.model.classes), self.scale)

    def eager(self, feats, labels, row_weight):
        return coop_step(self.model, self.clip_model, None, labels, row_weight, self.optimizer, image_features=feats, return_logits=True)


def lookahead_overlap():
    """Quarters of the chip an overlapped look-ahead encode may use (1-3), 0 = no overlap (the default).  $GRIP_LOOKAHEAD_OVERLAP (developer A/B: measured
    and rejected in r05 -- on this runtime the masked stream's kernels and the graph replays of the steps do not run side by side)."""
    import os
    v = os.environ.get("GRIP_LOOKAHEAD_OVERLAP", "0")
    return int(v) if v in ("0", "1", "2", "3") else 0


def lookahead_image_features(clip_model, batches, group=8, overlap=None):
    """The frozen image tower of a textual-prompt epoch, run `group` batches ahead: the image features of a step do not depend
    on the prompt being trained, so the batches of `group` consecutive steps are encoded in ONE inference forward (group x B x S
    rows feed the persistent 256x256 GEMMs at the pool-encode rate; a 16-image forward leaves them a quarter full) and each
    step receives its slice.  Every image is still encoded every time a step uses it (nothing is cached across steps or epochs),
    and the engine's forward is chunk-independent, so the features are bit-identical to a per-step encode.

    r05 experiment, OFF by default (`overlap` / GRIP_LOOKAHEAD_OVERLAP = 1..3 quarters of the chip): the encode of group g + 1 enqueued on a CU-masked
    side stream (engine.masked_stream; the persistent kernels size their grids to it, grip_set_cu_budget) BEFORE group g's features are yielded, so that
    it could run beside the prompt steps -- a latency-bound chain of ~176 launches of 28 - 60 workgroups each -- on the CUs the mask leaves free.  Same
    features bit for bit (tests/test_gpu_determinism.py), but no overlap happens on this runtime: encode 28.5 ms + 51 steps 56.3 ms take 102 ms
    "together" (88 ms with an ordinary side stream), tools/overlap_probe.py; profiles/HISTORY.md 11.6.

    `batches` yields tuples whose first element is the image batch [B, 3, R, R] (any further elements are passed through);
    yields (features [B, embed_dim], *rest) in the same order."""
    quarters = lookahead_overlap() if overlap is None else int(overlap)

    def groups():
        pending = []
        for b in batches:
            pending.append(b)
            if len(pending) == group:
                yield pending
                pending = []
        if pending:
            yield pending

    def encode(pending, side=None, budget=0):
        x = torch.cat([b[0] for b in pending])
        with torch.no_grad():
            if side is None:
                return clip_model.encode_image(x)
            main = torch.cuda.current_stream()
            side.wait_stream(main)
            with torch.cuda.stream(side):
                engine.set_cu_budget(budget)
                try:
                    feats = clip_model.encode_image(x)
                finally:
                    engine.set_cu_budget(0)
            x.record_stream(side)
            return feats

    def slices(pending, feats):
        at = 0
        for b in pending:
            n = b[0].shape[0]
            yield (feats[at:at + n],) + tuple(b[1:])
            at += n

    it = groups()
    cur = next(it, None)
    if cur is None:
        return
    ms = engine.masked_stream(cur[0][0].device, quarters) if quarters and cur[0][0].is_cuda else None
    if ms is None:
        while cur is not None:
            yield from slices(cur, encode(cur))
            cur = next(it, None)
        return
    side, n_cus = ms
    feats = encode(cur)
    while cur is not None:
        nxt = next(it, None)
        nxt_feats = encode(nxt, side, n_cus) if nxt is not None else None      # enqueued BEFORE this group's steps: runs beside them
        yield from slices(cur, feats)
        if nxt is not None:
            torch.cuda.current_stream().wait_stream(side)
            nxt_feats.record_stream(torch.cuda.current_stream())
        cur, feats = nxt, nxt_feats


class GraphedVptStep(GraphedStep):
    """vpt_step (visual prompt: image tower forward + backward; text features fixed for the epoch) replayed from a HIP graph."""

    def __init__(self, model, text_features, logit_scale, optimizer):
        super().__init__(optimizer)
        self.model, self.text_features, self.scale = model, text_features, float(logit_scale)

    def params(self):
        return [self.model.prefix]

    def forward_logits(self):
        return CosineHeadFn.apply(self.model(self.x), self.text_features, self.scale)

    def eager(self, images, labels, row_weight):
        return vpt_step(self.model, self.text_features, self.scale, images, labels, row_weight, self.optimizer, return_logits=True)


class GraphedUptStep(GraphedStep):
    """upt_step (multimodal prompt: mixer + both towers forward and backward, towers on two streams) replayed from a HIP graph."""

    def __init__(self, model, logit_scale, optimizer):
        super().__init__(optimizer)
        self.model, self.scale = model, float(logit_scale)

    def params(self):
        return [p for p in self.model.parameters() if p.requires_grad]

    def shape_key(self, images):
        return (tuple(images.shape), images.dtype, tuple(self.model.classes))

    def forward_logits(self):
        text_features, image_features = self.model(self.x, self.model.classes)
        return CosineHeadFn.apply(image_features, text_features, self.scale)

    def eager(self, images, labels, row_weight):
        return upt_step(self.model, self.scale, images, labels, row_weight, self.optimizer, return_logits=True)
