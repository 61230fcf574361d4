"""Host side of the native towers: weight blobs, handles, workspaces, autograd glue.

torch is used for device memory, streams and autograd bookkeeping only; every FLOP of the
encoders, the head and the losses runs in libgrip_amd.so (csrc/*.hip).
"""
import ctypes
import os
import weakref
from ctypes import byref, c_int64, c_size_t, c_uint64, c_void_p

import torch

from . import native
from .config import ClipDims


def _ptr(t):
    return c_void_p(t.data_ptr()) if t is not None else c_void_p(0)


def _stream():
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def require_gpu(device):
    device = torch.device(device)
    if device.type != "cuda" or not torch.cuda.is_available():
        raise native.GripError(
            "grip_amd runs only on an AMD GPU (device 'cuda' under PyTorch-ROCm); there is no CPU path. "
            f"Requested device: {device}")
    return device


_PINNING = []      # stack of lists: workspaces handed out while a HIP graph is being captured (see pin_workspaces)


class pin_workspaces:
    """Context manager around a HIP-graph capture (steps.GraphedStep): every workspace a tower hands out inside it is DEDICATED to
    that graph -- a train-mode one is never handed to another forward again (a replay would overwrite the activations an eager
    forward on the same workspace is still holding for its backward), an inference one is a private allocation (the shared
    inference workspace may be replaced by a larger one and freed under the graph).  The list it yields keeps them alive; the
    owner un-pins with release_pins when the graph dies."""

    def __enter__(self):
        self.pins = []
        _PINNING.append(self.pins)
        return self.pins

    def __exit__(self, *exc):
        _PINNING.pop()
        return False


def release_pins(pins):
    for tower, ws in pins:
        tower._pinned.pop(ws.data_ptr(), None)
    pins.clear()


class Tower:
    """One frozen CLIP tower (vision or text) living in two HBM blobs behind a native handle."""

    def __init__(self, kind, width, layers, heads, embed_dim, seq0, patch=0, resolution=0, vocab=0,
                 max_prefix=64, device="cuda", exact=False):
        self.device = require_gpu(device)
        self.lib = native.lib()
        self.precision = int(exact)  # grip_dims.precision: 0 f16 towers, 1 f32 (exact comparison mode), 2 split-f16 (the middle tier of screen-and-refine)
        self.exact = self.precision != 0     # f32 weights / activations / attention: inference only
        self.dims = native.Dims(kind, width, layers, heads, embed_dim, seq0, patch, resolution, vocab, max_prefix, self.precision)
        self.kind, self.width, self.embed_dim, self.seq0 = kind, width, embed_dim, seq0
        n16, n32 = c_int64(), c_int64()
        native.check(self.lib.grip_layout_size(byref(self.dims), byref(n16), byref(n32)))
        # the GEMM-operand blob: f16, or f32 elements in exact mode (same element offsets)
        self.blob16 = torch.zeros(n16.value, dtype=torch.float32 if self.exact else torch.float16, device=self.device)
        self.blob32 = torch.zeros(n32.value, dtype=torch.float32, device=self.device)
        self.slots = {}
        s, i = native.Slot(), 0
        while self.lib.grip_layout_slot(byref(self.dims), i, byref(s)) == 0:
            self.slots[s.name.decode()] = (s.dtype, s.derived, s.offset, s.rows, s.cols, s.ld)
            i += 1
        h = c_void_p()
        native.check(self.lib.grip_tower_create(byref(self.dims), _ptr(self.blob16), _ptr(self.blob32), byref(h)))
        self.handle = h
        self._ws = {}
        self._owners = {}
        self._pinned = {}        # data_ptr -> workspace dedicated to a captured HIP graph
        self._finalized = False

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.lib.grip_tower_destroy(self.handle)
        except Exception:
            pass

    # ---- weights
    def primary_names(self):
        return [n for n, v in self.slots.items() if not v[1]]

    def view(self, name):
        """Tensor view (rows x cols, row stride ld) of a slot inside its blob."""
        dtype, _, off, rows, cols, ld = self.slots[name]
        blob = self.blob16 if dtype == 0 else self.blob32
        return blob.as_strided((rows, cols), (ld, 1), off)

    def load(self, name, tensor):
        v = self.view(name)
        v.copy_(tensor.reshape(v.shape).to(device=self.device, dtype=v.dtype))
        self._finalized = False

    def finalize(self):
        native.check(self.lib.grip_tower_finalize(self.handle, _stream()))
        self._finalized = True

    # ---- workspaces
    def workspace(self, batch, n_prefix, train, seq_len=0):
        """Inference workspaces are interchangeable (the largest is kept).  Train-mode workspaces hold the activations a
        backward still needs: each shape keeps a small pool, and a workspace whose forward is still waiting for its
        backward (its autograd ctx is alive and has not run) is never handed out again -- model(aug_1) and model(aug_2)
        before one loss.backward() get two workspaces instead of overwriting each other."""
        key = (batch, n_prefix, bool(train), seq_len)
        if train:
            pool = self._ws.setdefault(key, [])
            ws = next((w for w in pool if not self._busy(w)), None)
            if ws is None:
                nbytes = c_size_t()
                native.check(self.lib.grip_workspace_bytes(self.handle, batch, n_prefix, seq_len, 1, byref(nbytes)))
                ws = torch.empty(nbytes.value + 256, dtype=torch.uint8, device=self.device)
                pool.append(ws)
            if _PINNING:
                self._pinned[ws.data_ptr()] = ws
                _PINNING[-1].append((self, ws))
            return ws
        if _PINNING:      # inside a graph capture: a private inference workspace (kept alive by the graph's owner)
            nbytes = c_size_t()
            native.check(self.lib.grip_workspace_bytes(self.handle, batch, n_prefix, seq_len, 0, byref(nbytes)))
            ws = torch.empty(nbytes.value + 256, dtype=torch.uint8, device=self.device)
            _PINNING[-1].append((self, ws))
            return ws
        ws = self._ws.get(key)
        if ws is None:
            nbytes = c_size_t()
            native.check(self.lib.grip_workspace_bytes(self.handle, batch, n_prefix, seq_len, 0, byref(nbytes)))
            for k in [k for k in self._ws if not k[2]]:
                if self._ws[k].numel() >= nbytes.value:
                    return self._ws[k]
                del self._ws[k]
            ws = torch.empty(nbytes.value + 256, dtype=torch.uint8, device=self.device)
            self._ws[key] = ws
        return ws

    def _busy(self, ws):
        if ws.data_ptr() in self._pinned:
            return True
        owner = self._owners.get(ws.data_ptr())
        return owner is not None and owner() is not None

    def hold(self, ws, ctx):
        """Mark a train-mode workspace as owned by the autograd ctx of its forward until that ctx's backward ran
        (release) or the graph was dropped (the weak reference dies)."""
        self._owners[ws.data_ptr()] = weakref.ref(ctx)

    def release(self, ws):
        self._owners.pop(ws.data_ptr(), None)

    @staticmethod
    def _aligned(ws):
        off = (-ws.data_ptr()) % 256
        return c_void_p(ws.data_ptr() + off), ws.numel() - off

    # ---- forward / backward
    def vit_forward(self, images, prefix=None, train=False, pos_emb=True):
        if not self._finalized:
            self.finalize()
        assert self.kind == 0
        images = images.contiguous()
        if images.dtype not in (torch.float32, torch.float16):
            images = images.float()
        B = images.shape[0]
        P = 0 if prefix is None else prefix.shape[-2]
        if prefix is not None:
            prefix = prefix.reshape(P, self.width).contiguous().float()
        out = torch.empty(B, self.embed_dim, dtype=torch.float32, device=self.device)
        ws = self.workspace(B, P, train)
        p, n = self._aligned(ws)
        gen = c_uint64(0)
        native.check(self.lib.grip_vit_forward(self.handle, _ptr(images), int(images.dtype == torch.float16), _ptr(prefix), P, B,
                                               _ptr(out), p, n, (native.FWD_TRAIN if train else 0) | (0 if pos_emb else native.FWD_NO_POS_EMB),
                                               byref(gen), _stream()))
        ws.generation = gen.value
        return out, ws

    @torch.no_grad()
    def encode_chunks(self, images, out, lo, hi, chunk, prefix=None, streams=2, hilo=False):
        """Inference encode of images[lo:hi] into out[0:hi-lo] in chunks, alternating between two HIP streams (each
        with its own workspace): the HBM-bound kernels of one chunk (LayerNorm, attention, patch gather) run next to
        the power-bound GEMMs of the other.  Rows are independent of the chunking, so the result is bit-identical to
        a single-stream pass (+6 % on the 50k-image pass).  streams=1 keeps everything on the current stream (per-kernel
        timings are only meaningful that way).  `images` is a tensor or a callable (a, b) -> tensor.
        hilo=True (f16 towers): the residual stream as a compensated f16 pair (GRIP_FWD_STREAM_HILO, include/grip_amd.h): the screen
        of the pseudolabel pass -- a different (more accurate) function of the image than the plain f16 stream's, equally chunk-independent."""
        if not self._finalized:
            self.finalize()
        P = 0 if prefix is None else prefix.shape[-2]
        if prefix is not None:
            prefix = prefix.reshape(P, self.width).contiguous().float()
        if not hasattr(self, "_enc_streams"):
            self._enc_streams = [torch.cuda.Stream(device=self.device), torch.cuda.Stream(device=self.device)]
            self._enc_ws = {}
        nbytes = c_size_t()
        native.check(self.lib.grip_workspace_bytes(self.handle, min(chunk, max(hi - lo, 1)), P, 0, 0, byref(nbytes)))
        for k in (0, 1):
            if k not in self._enc_ws or self._enc_ws[k].numel() < nbytes.value + 256:
                self._enc_ws[k] = torch.empty(nbytes.value + 256, dtype=torch.uint8, device=self.device)
        main = torch.cuda.current_stream()
        for st in self._enc_streams:
            st.wait_stream(main)
        if hasattr(images, "plan"):
            images.plan(lo, hi, chunk)      # lazy file pools decode one chunk ahead, never past the shard
        for i, s in enumerate(range(lo, hi, chunk)):
            e = min(s + chunk, hi)
            k = (i & 1) if streams == 2 else 0
            with torch.cuda.stream(self._enc_streams[k] if streams == 2 else main):
                x = images[s:e] if torch.is_tensor(images) else images(s, e)
                x = x.to(self.device, non_blocking=True).contiguous()
                if x.dtype not in (torch.float32, torch.float16):
                    x = x.float()
                if streams == 2:
                    x.record_stream(self._enc_streams[k])
                o = out[s - lo: e - lo]
                p, n = self._aligned(self._enc_ws[k])
                native.check(self.lib.grip_vit_forward(self.handle, _ptr(x), int(x.dtype == torch.float16), _ptr(prefix), P, e - s, _ptr(o), p, n,
                                                       native.FWD_STREAM_HILO if (hilo and self.precision == 0) else 0, None,
                                                       c_void_p((self._enc_streams[k] if streams == 2 else main).cuda_stream)))
        for st in self._enc_streams:
            main.wait_stream(st)
        return out

    def vit_backward(self, grad_emb, prefix, ws, generation=0):
        P = prefix.shape[-2]
        prefix = prefix.reshape(P, self.width).contiguous().float()
        grad_emb = grad_emb.contiguous().float()
        g = torch.empty(P, self.width, dtype=torch.float32, device=self.device)
        p, n = self._aligned(ws)
        native.check(self.lib.grip_vit_backward_prefix(self.handle, _ptr(grad_emb), _ptr(prefix), _ptr(g), p, n, generation, _stream()))
        return g

    # Encode only the positions up to the longest prompt's EOT: the text transformer is causal and only the EOT row is
    # read, so later positions cannot influence any output (exact; the reference encodes all 77).
    truncate_text_at_eot = True

    def text_forward(self, token_ids, prefix=None, train=False, seq_len=None, pos_emb=True):
        if not self._finalized:
            self.finalize()
        assert self.kind == 1
        # int32 ids and EOT row indices are kept on the token tensor (keyed by its version counter: an in-place edit recomputes
        # them) -- a prompt step would otherwise spend three tiny launches per forward re-deriving the same indices
        memo = getattr(token_ids, "_grip_ids", None)
        if memo is not None and memo[0] == token_ids._version and memo[1].device == self.device:
            ids, eot = memo[1], memo[2]
        else:
            ids = token_ids.to(device=self.device, dtype=torch.int32).contiguous()
            eot = ids.argmax(dim=-1).to(torch.int32).contiguous()
            token_ids._grip_ids = (token_ids._version, ids, eot)
        C = ids.shape[0]
        if seq_len is None:
            seq_len = min(int(eot.max().item()) + 1, self.seq0) if self.truncate_text_at_eot else 0
        P, pc = 0, 1
        if prefix is not None:
            pc, P = prefix.shape[0], prefix.shape[1]
            prefix = prefix.contiguous().float()
        flags = (native.FWD_TRAIN if train else 0) | (0 if pos_emb else native.FWD_NO_POS_EMB)
        if P and pc == 1 and not self.exact and self.share_text_prefix and self._shares_prefix(token_ids, ids, eot, P):
            flags |= native.FWD_SHARED_PREFIX
        self.last_text_flags = flags
        out = torch.empty(C, self.embed_dim, dtype=torch.float32, device=self.device)
        ws = self.workspace(C, P, train, seq_len)
        p, n = self._aligned(ws)
        gen = c_uint64(0)
        native.check(self.lib.grip_text_forward(self.handle, _ptr(ids), _ptr(eot), _ptr(prefix), P, pc, C, seq_len, _ptr(out), p, n,
                                                flags, byref(gen), _stream()))
        ws.generation = gen.value
        return out, ws, (ids, eot, seq_len)

    # One shared context (CoOp / UPT text side): positions 0 .. P hold the same tokens for every class and the mask is causal, so
    # the engine encodes them once (include/grip_amd.h, GRIP_FWD_SHARED_PREFIX).  GRIP_TEXT_SHARED_PREFIX=0 keeps the plain
    # n_class x seq_len layout (developer A/B; the tests hold the two equal).
    share_text_prefix = os.environ.get("GRIP_TEXT_SHARED_PREFIX", "1") != "0"

    @staticmethod
    def _shares_prefix(token_ids, ids, eot, P):
        """True when every class has the same tokens at positions 0 .. P and its EOT after them (checked once per token tensor
        and context length, remembered with the tensor's version counter)."""
        memo = getattr(token_ids, "_grip_shared", None)
        if memo is not None and memo[0] == token_ids._version and memo[1] == P:
            return memo[2]
        ok = bool(ids.shape[0] >= 2 and ids.shape[1] > P + 1 and (ids[:, :P + 1] == ids[:1, :P + 1]).all().item() and int(eot.min().item()) > P)
        token_ids._grip_shared = (token_ids._version, P, ok)
        return ok

    def text_backward(self, grad_emb, prefix_shape, ws, generation=0):
        grad_emb = grad_emb.contiguous().float()
        g = torch.empty(prefix_shape, dtype=torch.float32, device=self.device)
        p, n = self._aligned(ws)
        native.check(self.lib.grip_text_backward_prefix(self.handle, _ptr(grad_emb), _ptr(g), p, n, generation, _stream()))
        return g


def vision_tower(d: ClipDims, device="cuda", max_prefix=64, exact=False):
    return Tower(0, d.vision_width, d.vision_layers, d.vision_heads, d.embed_dim, d.vision_seq, d.vision_patch_size,
                 d.image_resolution, 0, max_prefix, device, exact)


def text_tower(d: ClipDims, device="cuda", max_prefix=64, exact=False):
    return Tower(1, d.transformer_width, d.transformer_layers, d.transformer_heads, d.embed_dim, d.context_length, 0, 0,
                 d.vocab_size, max_prefix, device, exact)


# ------------------------------------------------------------------------------------------ autograd
class VitPrefixFn(torch.autograd.Function):
    """CustomVisionTransformer.forward with autograd to the visual prompt only (frozen backbone)."""

    @staticmethod
    def forward(ctx, tower, images, prefix, pos_emb=True):
        need = ctx.needs_input_grad[2]   # grad mode is off inside Function.forward
        if need and tower.exact:
            raise native.GripError("exact (f32) towers are inference-only: prompt gradients need a default-precision tower")
        out, ws = tower.vit_forward(images, prefix.detach(), train=need, pos_emb=pos_emb)
        ctx.tower, ctx.ws, ctx.generation = tower, ws, ws.generation
        if need:
            tower.hold(ws, ctx)
        ctx.save_for_backward(prefix.detach())
        ctx.pshape, ctx.pdtype = prefix.shape, prefix.dtype
        return out

    @staticmethod
    def backward(ctx, grad_out):
        (prefix,) = ctx.saved_tensors
        g = ctx.tower.vit_backward(grad_out, prefix, ctx.ws, ctx.generation)
        ctx.tower.release(ctx.ws)
        return None, None, g.reshape(ctx.pshape).to(ctx.pdtype), None


class TextPrefixFn(torch.autograd.Function):
    """CustomTextEncoder.forward with autograd to the textual prompt only."""

    @staticmethod
    def forward(ctx, tower, token_ids, prefix, pos_emb=True):
        need = ctx.needs_input_grad[2]
        if need and tower.exact:
            raise native.GripError("exact (f32) towers are inference-only: prompt gradients need a default-precision tower")
        cached = getattr(token_ids, "_grip_seq_len", None)
        out, ws, keep = tower.text_forward(token_ids, prefix.detach(), train=need, seq_len=cached, pos_emb=pos_emb)
        token_ids._grip_seq_len = keep[2]
        ctx.tower, ctx.ws, ctx.generation = tower, ws, ws.generation
        if need:
            tower.hold(ws, ctx)
        ctx.keep = keep   # the native handle remembers the EOT-index pointer until backward
        ctx.pshape, ctx.pdtype = prefix.shape, prefix.dtype
        return out

    @staticmethod
    def backward(ctx, grad_out):
        g = ctx.tower.text_backward(grad_out, tuple(ctx.pshape), ctx.ws, ctx.generation)
        ctx.tower.release(ctx.ws)
        return None, None, g.to(ctx.pdtype), None


def vit_prefix_forward(tower, images, prefix, pos_emb=True):
    """CustomVisionTransformer.forward on the native tower.  The train-mode forward (activations saved for the prompt
    gradient) runs only when a gradient can actually be asked for: grad mode on AND the prompt requires grad.  Under
    torch.no_grad() -- validation, test predictions, the pseudolabel passes -- it is the plain inference forward, the same
    arithmetic as the pool encode (autograd's needs_input_grad alone does not see the surrounding no_grad)."""
    if torch.is_grad_enabled() and prefix.requires_grad:
        return VitPrefixFn.apply(tower, images, prefix, pos_emb)
    return tower.vit_forward(images, prefix.detach(), train=False, pos_emb=pos_emb)[0]


def text_prefix_forward(tower, token_ids, prefix, pos_emb=True):
    """CustomTextEncoder.forward on the native tower; see vit_prefix_forward.  pos_emb=False is the reference's enable_pos_emb=False
    branch (models/clip_encoders.py:70-74): the positional embedding is not added (the positions' gradient path is untouched: it is additive)."""
    if torch.is_grad_enabled() and prefix.requires_grad:
        return TextPrefixFn.apply(tower, token_ids, prefix, pos_emb)
    cached = getattr(token_ids, "_grip_seq_len", None)
    out, _, keep = tower.text_forward(token_ids, prefix.detach(), train=False, seq_len=cached, pos_emb=pos_emb)
    token_ids._grip_seq_len = keep[2]
    return out


def cosine_head(img_emb, txt_emb, scale, want_probs=True):
    """normalize -> scale * img @ txt.T -> softmax / argmax, fused (no autograd)."""
    lib = native.lib()
    img = img_emb.detach().contiguous().float()
    txt = txt_emb.detach().contiguous().float()
    n, e = img.shape
    c = txt.shape[0]
    dev = img.device
    logits = torch.empty(n, c, dtype=torch.float32, device=dev)
    probs = torch.empty(n, c, dtype=torch.float32, device=dev) if want_probs else None
    am_l = torch.empty(n, dtype=torch.int32, device=dev)
    am_p = torch.empty(n, dtype=torch.int32, device=dev) if want_probs else None
    scratch = torch.empty(c, e, dtype=torch.float32, device=dev)
    native.check(lib.grip_cosine_head(_ptr(img), _ptr(txt), float(scale), n, c, e, _ptr(logits), _ptr(probs), _ptr(am_l), _ptr(am_p),
                                      _ptr(scratch), _stream()))
    return logits, probs, am_l, am_p


class CosineHeadFn(torch.autograd.Function):
    """logits = scale * normalize(img) @ normalize(txt).T with native forward and backward."""

    @staticmethod
    def forward(ctx, img_emb, txt_emb, scale):
        logits, _, _, _ = cosine_head(img_emb, txt_emb, scale, want_probs=False)
        ctx.save_for_backward(img_emb.detach(), txt_emb.detach())
        ctx.scale = float(scale)
        ctx.need = (img_emb.requires_grad, txt_emb.requires_grad)
        return logits

    @staticmethod
    def backward(ctx, grad_logits):
        img, txt = ctx.saved_tensors
        lib = native.lib()
        img = img.contiguous().float()
        txt = txt.contiguous().float()
        n, e = img.shape
        c = txt.shape[0]
        g = grad_logits.contiguous().float()
        gi = torch.empty_like(img) if ctx.need[0] else None
        gt = torch.empty_like(txt) if ctx.need[1] else None
        native.check(lib.grip_cosine_head_backward(_ptr(img), _ptr(txt), ctx.scale, n, c, e, _ptr(g), _ptr(gi), _ptr(gt), _stream()))
        return gi, gt, None


class WeightedCEFn(torch.autograd.Function):
    """sum_i w_i * CE(logits_i, label_i), native forward+backward (one kernel produces both)."""

    @staticmethod
    def forward(ctx, logits, labels, row_weight):
        lib = native.lib()
        lg = logits.contiguous().float()
        n, c = lg.shape
        lab = labels.to(device=lg.device, dtype=torch.int32).contiguous()
        w = row_weight.to(device=lg.device, dtype=torch.float32).contiguous()
        loss = torch.zeros(1, dtype=torch.float32, device=lg.device)
        grad = torch.empty_like(lg)
        native.check(lib.grip_weighted_ce(_ptr(lg), _ptr(lab), _ptr(w), n, c, _ptr(loss), _ptr(grad), _stream()))
        ctx.save_for_backward(grad)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g, None, None


class UptMixerFn(torch.autograd.Function):
    """UPTModel's prompt mixer (models/prompts_models.py:129-146) on the native kernels (csrc/mixer.hip): forward and the
    gradients of all 22 tensors -- the two prompt embeddings, the four projections and the one-block transformer."""

    @staticmethod
    def forward(ctx, *tensors):
        lib = native.lib()
        ts = [t.detach().contiguous().float() for t in tensors]
        coop, vpt = ts[0].reshape(-1, ts[0].shape[-1]), ts[1].reshape(-1, ts[1].shape[-1])
        P, dt, dv, D = coop.shape[0], coop.shape[1], vpt.shape[1], ts[2].shape[0]
        if vpt.shape[0] != P:
            raise native.GripError(f"UPT mixer: {P} text prompt tokens but {vpt.shape[0]} visual ones (the reference concatenates them along dim 0)")
        dev = coop.device
        nbytes = c_size_t()
        native.check(lib.grip_upt_mixer_workspace(P, dt, dv, D, byref(nbytes)))
        ws = torch.empty(nbytes.value, dtype=torch.uint8, device=dev)
        coop_out = torch.empty(P, dt, dtype=torch.float32, device=dev)
        vpt_out = torch.empty(P, dv, dtype=torch.float32, device=dev)
        # the reference's float16 branch (multimodal_prompt.py:46): fp16 projection Linears / prompt embeddings around the fp32 block
        half = int(tensors[2].dtype == torch.float16)
        m = native.UptMixer(P, dt, dv, D, half, 0, *[t.data_ptr() for t in ts])
        native.check(lib.grip_upt_mixer_forward(byref(m), _ptr(coop_out), _ptr(vpt_out), _ptr(ws), ws.numel(), _stream()))
        ctx.save_for_backward(*ts)
        ctx.ws, ctx.dims, ctx.half = ws, (P, dt, dv, D), half
        ctx.shapes = [t.shape for t in tensors]
        ctx.dtypes = [t.dtype for t in tensors]
        return coop_out, vpt_out

    @staticmethod
    def backward(ctx, d_coop, d_vpt):
        lib = native.lib()
        ts = ctx.saved_tensors
        P, dt, dv, D = ctx.dims
        dev = ts[0].device
        d_coop = torch.zeros(P, dt, device=dev) if d_coop is None else d_coop.contiguous().float()
        d_vpt = torch.zeros(P, dv, device=dev) if d_vpt is None else d_vpt.contiguous().float()
        grads = [torch.empty_like(t) for t in ts]
        m = native.UptMixer(P, dt, dv, D, ctx.half, 0, *[t.data_ptr() for t in ts])
        g = native.UptMixer(P, dt, dv, D, ctx.half, 0, *[t.data_ptr() for t in grads])
        native.check(lib.grip_upt_mixer_backward(byref(m), _ptr(d_coop), _ptr(d_vpt), byref(g), _ptr(ctx.ws), ctx.ws.numel(), _stream()))
        return tuple(gr.reshape(sh).to(dtp) for gr, sh, dtp in zip(grads, ctx.shapes, ctx.dtypes))


def leaderboard_scan(probs, pred, path_rank, k):
    """Host scan (exact, sequential).  probs [n,c] f32 CPU, pred [n] int32 CPU, path_rank [n] int64 CPU."""
    import numpy as np
    lib = native.lib()
    probs = np.ascontiguousarray(probs, dtype=np.float32)
    pred = np.ascontiguousarray(pred, dtype=np.int32)
    rank = np.ascontiguousarray(path_rank, dtype=np.int64)
    n, c = probs.shape
    cap = c * max(1, min(int(k), n))
    out_img = np.empty(cap, dtype=np.int32)
    out_cls = np.empty(cap, dtype=np.int32)
    m = c_int64()
    native.check(lib.grip_leaderboard_scan(c_void_p(probs.ctypes.data), c_void_p(pred.ctypes.data), c_void_p(rank.ctypes.data),
                                           n, c, int(k), c_void_p(out_img.ctypes.data), c_void_p(out_cls.ctypes.data), byref(m)))
    return out_img[: m.value].copy(), out_cls[: m.value].copy()


BOUND_FORMS = {"relative": 0, "odds": 1}


def leaderboard_scan_bounded(probs, pred, path_rank, rel_eps, k, abs_eps=0.0, form="relative", threads=0):
    """grip_leaderboard_scan_bounded: (img, cls, ambiguous) with ambiguous a bool [n] array of the rows the caller has to
    re-encode more accurately before the lists can be trusted (include/grip_amd.h).  rel_eps [n]: per-row bound (0 = final) -- a relative
    bound on every probability (form "relative") or a bound on the spread of the row's logit errors (form "odds": every entry's odds
    p / (1 - p) are known to a factor e^{+-delta});  abs_eps: absolute slack of every non-final row;  threads: worker threads of the scan's
    pre-filter (0 = the library's default)."""
    import numpy as np
    lib = native.lib()
    probs = np.ascontiguousarray(probs, dtype=np.float32)
    pred = np.ascontiguousarray(pred, dtype=np.int32)
    rank = np.ascontiguousarray(path_rank, dtype=np.int64)
    eps = np.ascontiguousarray(rel_eps, dtype=np.float32)
    n, c = probs.shape
    cap = n if int(k) == 10000000 else c * max(1, min(int(k), n))
    out_img = np.empty(max(cap, 1), dtype=np.int32)
    out_cls = np.empty(max(cap, 1), dtype=np.int32)
    amb = np.zeros(max(n, 1), dtype=np.uint8)
    m, na = c_int64(), c_int64()
    native.check(lib.grip_leaderboard_scan_bounded(c_void_p(probs.ctypes.data), c_void_p(pred.ctypes.data), c_void_p(rank.ctypes.data),
                                                   c_void_p(eps.ctypes.data), float(abs_eps), BOUND_FORMS[form], int(threads), n, c, int(k),
                                                   c_void_p(out_img.ctypes.data), c_void_p(out_cls.ctypes.data), byref(m), c_void_p(amb.ctypes.data), byref(na)))
    return out_img[: m.value].copy(), out_cls[: m.value].copy(), amb[:n].astype(bool)


# ------------------------------------------------------------------------------------------ CU-masked side stream (look-ahead encodes)
_MASKED_STREAMS = {}


def set_cu_budget(n_cus):
    """grip_set_cu_budget: CUs the persistent kernels may size their grids to for the launches that follow (0 = the whole chip)."""
    native.check(native.lib().grip_set_cu_budget(int(n_cus)))


def masked_stream(device, quarters=3):
    """(torch stream, n_cus) whose kernels may only run on `quarters` / 4 of the chip's CUs (hipExtStreamCreateWithCUMask), or None when the runtime
    refuses.  Every 32-CU XCD and every 8-CU slice of the CU numbering loses the same share whichever way the driver numbers the CUs (CU i is dropped
    when (i / 8 + i) % 4 >= quarters), so the XCD-aware tile walks of the GEMMs stay balanced.  What it is for: a throughput-bound encode on most of
    the chip next to a latency-bound chain of small kernels on the rest (steps.lookahead_image_features)."""
    import ctypes
    import os
    dev = torch.device(device)
    if dev.index is None:           # "cuda": the CURRENT device, not device 0
        dev = torch.device("cuda", torch.cuda.current_device())
    key = (dev.index, quarters)
    if key in _MASKED_STREAMS:
        return _MASKED_STREAMS[key]
    out = None
    try:
        total = torch.cuda.get_device_properties(dev).multi_processor_count
        words = (total + 31) // 32
        mask = (ctypes.c_uint32 * words)()
        kept = 0
        for i in range(total):
            if (i // 8 + i) % 4 < quarters:
                mask[i // 32] |= 1 << (i % 32)
                kept += 1
        hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
        st = ctypes.c_void_p()
        with torch.cuda.device(dev):
            rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), ctypes.c_uint32(words), mask)
        if rc == 0 and st.value:
            out = (torch.cuda.ExternalStream(st.value, device=dev), kept)
    except Exception:
        out = None
    _MASKED_STREAMS[key] = out
    return out
