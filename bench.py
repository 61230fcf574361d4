#!/usr/bin/env python3
"""bench.py -- images/sec of the CLIP ViT-B/16 pseudolabel + prompt-step loop on N MI355X.

    python bench.py [--gpus N --steps K --warmup W]

N > 1: either launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` (RANK /
WORLD_SIZE in the environment), or plainly as `python bench.py --gpus N`, in which case it re-executes itself under
torch.distributed.run on 127.0.0.1 (one process per GPU, RCCL = backend "nccl" on device tensors).

One "step" is one full pass of the hot path over this rank's pool of synthetic images (BASELINE.json
configs[1]: Flowers102-shaped CoOp textual-prompt SSL, C = 102 classes, 16 prompt tokens, k = 16,
ViT-B/16):
  (i)   encode the pool with the frozen ViT (chunks of --chunk images), all-gather embeddings over ranks;
  (ii)  cosine head + softmax + arg-max against the class text features (encoded once per pass);
  (iii) sequential leaderboard scan, k = 16 (host, exact);
  (iv)  ceil(M / (16 * ranks)) CoOp prompt-tuning steps (text tower fwd+bwd over 102 prompts, frozen
        image tower fwd, head, CE, SGD) over the M selected pseudolabeled images.
Inputs are resident in HBM before the timed region.  Weak scaling: every rank holds --pool images.
Prints ONE JSON line (rank 0) with `roofline` (dominant kernel, live HIP-event timing inside the
library), `cpu_baseline` (the CPU oracle on a bounded sample, rank 0, N = 1 only), `exact` (the fp32 comparison mode:
images/sec and its pair overlap with the f16 lists on the same pool) and `secondary` (the other BASELINE.json
configs: VPT step, UPT step, ViT-L/14@336px encode) -- the last two at N = 1 only, outside the timed region.
"""
import argparse
import ctypes
import json
import math
import os
import socket
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)
import grip_amd  # noqa: E402
from grip_amd import clip, config, dist as gdist, engine, native, pseudolabels as pl, rng, steps  # noqa: E402
from grip_amd.models import (CustomImageEncoder, CustomTextEncoder, ImagePrefixModel, TextPrefixModel, UPTModel)  # noqa: E402

MODEL = "ViT-B/16"
F_IMG = 35.13e9          # algorithmic FLOPs per image, frozen ViT-B/16 forward
F_TXT = 5.96e9           # per class prompt, text tower forward (77 positions)
PEAK_F16_TFLOPS = 2500.0 # MI355X dense f16/bf16 MFMA peak (MI355X_MICROARCH.md)
PEAK_F32_TFLOPS = 157.3  # dense f32 MFMA peak (v_mfma_f32_16x16x4_f32), the exact mode's roof
EPI_NAMES = ["EPI_F32", "EPI_BIAS_F16", "EPI_BIAS_GELU_F16", "EPI_BIAS_RESID", "EPI_F16", "EPI_GELUGRAD_F16", "EPI_F32_SCALE", "EPI_LNFOLD_F16",
             "EPI_LNFOLD_GELU_F16", "EPI_BIAS_RESID_STATS"]
TRAFFIC_FILE = os.path.join("profiles", "r06_traffic.json")
PEAK_CLOCK_MHZ = 2400.0  # the shader clock behind the 2.5 PFLOP/s figure


def clock_marker(label):
    """`##clock_trace <label>` on stderr for tools/clock_trace.py (GRIP_CLOCK_MARKERS=1)."""
    if os.environ.get("GRIP_CLOCK_MARKERS") == "1":
        print(f"##clock_trace {label}", file=sys.stderr, flush=True)


def clock_sampler():
    """The hwmon sampler of tools/clock_trace.py (shader clock / package power at 20 Hz from a background thread), or None when
    the box exposes no readable amdgpu hwmon files."""
    try:
        sys.path.insert(0, os.path.join(REPO, "tools"))
        from clock_trace import ClockSampler
        s = ClockSampler(20.0)
        return s if s.available() else None
    except Exception:
        return None


def text_flops(seq, width=512, layers=12):
    """FLOPs of one class prompt through the text tower when `seq` positions are encoded (QKV + out-proj + MLP = 24 S d^2,
    attention core 4 S^2 d per layer): 5.96 GF at seq = 77; the engine encodes only positions <= the longest EOT (exact)."""
    return layers * (24.0 * seq * width * width + 4.0 * seq * seq * width)


def vit_flops_executed(seq=197, width=768, nominal=F_IMG):
    """FLOPs the engine issues per image at inference: the last block computes K and V for every row but Q, attention, out-proj
    and the MLP only for the CLS row (csrc/tower.hip run_blocks; nothing else of that block's output is ever read):
    20 d^2 (S - 1) + 4 S d (S - 1) fewer than the nominal block."""
    return nominal - (20.0 * width * width * (seq - 1) + 4.0 * seq * width * (seq - 1))


def synth_tokens(C, P, seed=7):
    """[C,77] ids = SOT, P x 343, 3 random ids in [1000, 40000), EOT, 0...."""
    ids = np.zeros((C, 77), dtype=np.int32)
    body = rng.integers(seed, rng.stream_id(f"bench.tokens.{P}"), (C, 3), 1000, 40000)
    for c in range(C):
        row = [config.SOT_TOKEN] + [config.X_TOKEN] * P + list(body[c]) + [config.EOT_TOKEN]
        ids[c, : len(row)] = row
    return torch.from_numpy(ids)


def synth_pool(n, res, device, seed):
    g = torch.Generator(device=device).manual_seed(seed)
    pool = torch.empty(n, 3, res, res, dtype=torch.float32, device=device)
    for s in range(0, n, 2048):
        e = min(s + 2048, n)
        pool[s:e].normal_(generator=g)
    return pool


class Loop:
    def __init__(self, args, device, rank, ws):
        self.args, self.device, self.rank, self.ws = args, device, rank, ws
        self.C, self.P, self.k = args.classes, args.prefix, args.k
        self.m, _ = clip.load(MODEL, device=device)
        self.d = config.get_dims(MODEL)
        self.pool = synth_pool(args.pool, self.d.image_resolution, device, 1234 + rank)
        self.n_total = args.pool * ws
        self.zs_tokens = synth_tokens(self.C, 0).to(device)
        self.classes = [f"class_{i}" for i in range(self.C)]
        self.paths = [f"pool/{i:08d}.jpg" for i in range(self.n_total)]
        self.ranks = pl.path_ranks(self.paths)
        self.class_labels = list(range(self.C))
        prefix = torch.from_numpy(rng.normal(1, rng.stream_id("bench.prefix"), (1, self.P, self.d.transformer_width), 0.0, 0.02)).to(device)
        enc = CustomTextEncoder(self.m, device, torch.float32)
        self.coop_tokens = synth_tokens(self.C, self.P).to(device)
        enc._tok_cache[(self.P, tuple(self.classes))] = self.coop_tokens   # no tokenizer needed
        self.model = TextPrefixModel(prefix, enc, self.classes, device=device)
        self.opt = torch.optim.SGD([self.model.prefix], lr=0.1, weight_decay=0.1)
        self.graphed = steps.GraphedCoopStep(self.model, self.m, self.opt) if args.graph else None
        self.graphed_f = steps.GraphedCoopFeatureStep(self.model, self.m, self.opt) if args.graph else None
        self.twin = self.m.exact_twin() if args.mode == "identical" else None
        self.split = self.m.split_twin() if args.mode == "identical" and os.environ.get("GRIP_SPLIT_TIER", "auto") != "0" else None     # the middle tier (precision 2)
        self.refine_stats = None
        self.stage = {k: 0.0 for k in ("encode_f16", "allgather", "head_scan", "refine_split", "refine_exact", "train")}     # wall seconds per stage, this rank
        self.t_pl = self.t_tr = 0.0
        self.m_selected = 0
        self.train_steps = 0
        self.last_lists = None

    def tick(self):
        """Stage boundary: the device work of the stage is done when the clock is read (the next stage depends on it anyway)."""
        torch.cuda.synchronize()
        return time.perf_counter()

    def identical_pass(self, streams):
        """(i)-(iii) with the index guarantee (pseudolabels.identical_lists on this rank's resident shard): f16 encode of the
        whole pool, head against the EXACT text features, error-bounded scan, exact re-encode of the rows it marks, until the
        scan certifies that its lists are the fp32 scan's."""
        a = self.args
        st = self.stage
        with torch.no_grad():
            t0 = self.tick()
            txt = self.twin.encode_text(self.zs_tokens)
            local = torch.empty(a.pool, self.d.embed_dim, dtype=torch.float32, device=self.device)
            key = ("bench", id(self.pool), a.pool, self.C)
            stream = pl.screen_stream(key)
            self.m.visual.tower.encode_chunks(self.pool, local, 0, a.pool, a.chunk, streams=streams, hilo=stream == "hilo")
            t1 = self.tick()
            emb = gdist.allgather_rows(local, self.n_total, a.pool, tag="pool_embeddings")
            t2 = self.tick()
            scale = self.m.logit_scale.exp().item()
            _, probs, _, am_p = engine.cosine_head(emb, txt, scale)
            probs_h, pred_h = probs.cpu().numpy(), am_p.cpu().numpy()
            lo = self.rank * a.pool
            t_tier = {"exact": 0.0, "split": 0.0}

            def rows_through(tower, tier, chunk):
                # (pseudolabels.tier_rows: the product's tier callback; the tiers of one refinement step run side by side on their own streams, so the
                # per-tier seconds below are host-side waits and may overlap)
                def spent(dt):
                    t_tier[tier] += dt
                return pl.tier_rows(tower, lambda rows: self.pool[torch.from_numpy(rows - lo).to(self.device)], txt, scale, self.n_total, lo, lo + a.pool, chunk, timer=spent)

            img, cls, self.refine_stats = pl.refine_scan(probs_h, pred_h, self.ranks, self.k, rows_through(self.twin.visual.tower, "exact", a.exact_chunk),
                                                         mid_rows=rows_through(self.split.visual.tower, "split", a.exact_chunk) if self.split is not None else None)
            pl.note_screen_bound(key, stream, self.refine_stats)
            t3 = self.tick()
        st["encode_f16"] += t1 - t0
        st["allgather"] += t2 - t1
        st["refine_exact"] += t_tier["exact"]
        st["refine_split"] += t_tier["split"]
        st["head_scan"] += t3 - t2 - t_tier["exact"] - t_tier["split"]
        return img, cls

    def pseudolabel_pass(self, model, streams):
        """(i)-(iii) with `model` (the f16 engine, or the exact one for the comparison block)."""
        a = self.args
        if model is self.m and a.mode == "identical":
            return self.identical_pass(streams)
        trace = os.environ.get("GRIP_BENCH_TRACE") == "1"     # developer: per-stage wall times of one pass (adds synchronisations)
        marks = [("start", time.perf_counter())]

        def mark(name):
            if trace:
                torch.cuda.synchronize()
                marks.append((name, time.perf_counter()))

        with torch.no_grad():
            t0 = self.tick()
            txt = model.encode_text(self.zs_tokens)
            mark("text")
            # every rank encodes its own pool; embeddings are gathered in global (rank-major) order
            local = torch.empty(a.pool, self.d.embed_dim, dtype=torch.float32, device=self.device)
            model.visual.tower.encode_chunks(self.pool, local, 0, a.pool, a.chunk if model is self.m else a.exact_chunk, streams=streams)
            mark("encode")
            t1 = self.tick()
            emb = gdist.allgather_rows(local, self.n_total, a.pool, tag="pool_embeddings")
            t2 = self.tick()
            if model is self.m:
                self.stage["encode_f16"] += t1 - t0
                self.stage["allgather"] += t2 - t1
            logits, probs, am_l, am_p = engine.cosine_head(emb, txt, model.logit_scale.exp().item())
            mark("gather+head")
            probs_h = probs.cpu().numpy()
            pred_h = am_p.cpu().numpy()
            mark("d2h")
        out = engine.leaderboard_scan(probs_h, pred_h, self.ranks, self.k)
        mark("scan")
        if trace and self.rank == 0:
            print("pass stages (ms): " + ", ".join(f"{n} {(t - marks[i][1]) * 1e3:.2f}" for i, (n, t) in enumerate(marks[1:])), file=sys.stderr)
        return out

    def step(self):
        a = self.args
        t0 = time.perf_counter()
        img, cls = self.pseudolabel_pass(self.m, a.streams)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        # (iv) prompt steps over the selected pairs, data-parallel as the product's trainer is (TrainingStrategy._loader -> dist.rank_batches = accelerate's
        # even sharding, textual_prompt.py:131 / :239): the M pairs in list order are cut into batches of `batch`, batch j goes to rank j % ws (tail padded
        # from the start), gradients are averaged.  The pool is resident per rank, so the selected images travel once per pass: every rank contributes the
        # ones of its own shard to one all-gather (dist.allgather_selected; nothing moves at N = 1).
        lo = self.rank * a.pool
        uniq = np.unique(img) if len(img) else np.empty(0, np.int64)
        mine = torch.from_numpy(uniq[(uniq >= lo) & (uniq < lo + a.pool)] - lo).long().to(self.device)
        sel = gdist.allgather_selected(self.pool[mine].flatten(1), uniq, self.n_total, tag="train_images").view(-1, *self.pool.shape[1:]) if len(uniq) else self.pool[:0]
        my_batches = gdist.rank_batches(range(len(img)), a.batch)
        n_steps = len(my_batches)
        at = torch.from_numpy(np.searchsorted(uniq, img)).to(self.device)           # pair -> row of `sel`
        pair_lab = torch.from_numpy(cls).to(self.device)
        flat = torch.tensor([i for b in my_batches for i in b], dtype=torch.long, device=self.device)
        order, labels = at[flat], pair_lab[flat]
        sizes = [len(b) for b in my_batches]          # (one process keeps a ragged last batch, like the trainer's loader)
        starts = np.concatenate([[0], np.cumsum(sizes)]).astype(int)
        w_rows = {n: torch.full((n,), 1.0 / n, device=self.device) for n in set(sizes)}

        def batches():
            for t in range(n_steps):
                sl = slice(starts[t], starts[t + 1])
                yield sel[order[sl]], labels[sl], w_rows[sizes[t]]

        if a.lookahead > 1:     # frozen image tower encoded `lookahead` steps at a time (steps.lookahead_image_features)
            for f, y, w in steps.lookahead_image_features(self.m, batches(), a.lookahead):
                if self.graphed_f is not None:
                    self.graphed_f(f, y, w)
                else:
                    steps.coop_step(self.model, self.m, None, y, w, self.opt, image_features=f)
        else:
            for x, y, w in batches():
                if self.graphed is not None:
                    self.graphed(x, y, w)
                else:
                    steps.coop_step(self.model, self.m, x, y, w, self.opt)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        self.t_pl += t1 - t0
        self.t_tr += t2 - t1
        self.stage["train"] += t2 - t1
        self.m_selected = len(img)
        self.train_steps = n_steps
        self.last_lists = (img, cls)


# ------------------------------------------------------------------------------------------------ GEMM profiler helpers
N_SLOTS = 144   # slot = variant * 16 + epilogue id (csrc/gemm.hip)


def kname(slot):
    v, e = divmod(slot, 16)
    return {0: f"gemm_f32_kernel<{e}>", 7: f"gemm_split_kernel<{e}>", 1: f"gemm_f16_kernel<{e}, 4>", 4: f"gemm_f16_kernel<{e}, 2>", 2: f"gemm_big_kernel<{e}, 256, 256, 4>", 3: f"gemm_big_kernel<{e}, 256, 128, 3>",
            5: f"gemm_k64_kernel<{e}, 8, 8>", 6: f"gemm_k64p_kernel<{e}>", 8: f"gemm_k64_kernel<{e}, 8, 6>"}.get(v, f"gemm?<{e}>") + f" [{EPI_NAMES[e] if e < len(EPI_NAMES) else e}]"


def profile_collect(lib):
    launches = np.zeros(N_SLOTS, dtype=np.int64)
    ms = np.zeros(N_SLOTS, dtype=np.float64)
    fl = np.zeros(N_SLOTS, dtype=np.float64)
    native.check(lib.grip_profile_collect(N_SLOTS, ctypes.c_void_p(launches.ctypes.data), ctypes.c_void_p(ms.ctypes.data), ctypes.c_void_p(fl.ctypes.data)))
    lib.grip_profile_enable(0)
    return launches, ms, fl


def dominant(launches, ms, fl):
    dom = int(np.argmax(ms))
    tf = fl[dom] / (ms[dom] * 1e-3) / 1e12 if ms[dom] > 0 else 0.0
    return dom, tf


def full_kernel_name(slot):
    """The instantiation rocprofv3 names for a profiler slot, template arguments included (csrc/gemm.hip launch_k64p: the persistent
    GEMM carries its epilogue form and the single-block flag): `gemm_k64p_kernel<9, 1, true>`.  None when this host cannot tell."""
    v, e = divmod(slot, 16)
    if v != 6:
        return kname(slot).split(" [")[0]
    env = int(os.environ.get("GRIP_GEMM_EMODE", "421"))
    mode = env if env < 100 else (env // 100 if e == 7 else (env // 10) % 10 if e == 8 else env % 10)
    if e == 9 and mode in (2, 4):
        mode = 1
    if e not in (7, 8, 9):
        return None
    sd = os.environ.get("GRIP_GEMM_SD", "1") != "0"
    return f"gemm_k64p_kernel<{e}, {mode}, true>" if sd else f"gemm_k64p_kernel<{e}, {mode}>"


def pmc_entry(slot):
    """Per-launch HBM bytes (2 x FETCH_SIZE + WRITE_SIZE) and MFMA utilisation of the dominant kernel from the separate rocprofv3 --pmc
    passes of this same command that are committed under profiles/ (PMC counters cannot be collected inside a timed run).  The entry
    is used only when the file describes EXACTLY the instantiation that ran (name and every template argument); otherwise
    (None, None): a traffic figure of another kernel is worse than none."""
    try:
        with open(os.path.join(REPO, TRAFFIC_FILE)) as f:
            kernels = json.load(f)["kernels"]
        t = kernels[full_kernel_name(slot)]
        return t.get("bytes_per_launch"), t.get("mfma_util")
    except Exception:
        return None, None


# ------------------------------------------------------------------------------------------------ CPU baseline
def cpu_model_string():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def cpu_baseline(args):
    """The CPU oracle (oracle/, a port of the reference path: kind "port") on a bounded sample of the same workload
    (BASELINE.md section 3), within a wall-clock budget (--cpu-budget seconds, default 40) so that the default run stays short on
    any host.
    R-mode = the reference loop of utils/clip_pseudolabels.py:31-41: batch 1, the full clip_model(image, text) per image, i.e.
    the image tower at batch 1 plus all C class prompts re-encoded for every image.  Full calls are timed first (at least one, at
    most --cpu-text-reps, a third of the budget), then the image tower alone at batch 1 on further images (at most --cpu-sample,
    another third); the per-image text re-encode -- the same 102 prompts every time, ~95 % of the loop's FLOPs -- is the
    difference of the two means.  R-mode images/sec = 1 / (mean image time + text share).  --cpu-full times the literal loop on
    every sampled image instead.  B-mode (reported beside it) = batch 16 with text features cached.
    Threads: the CPUs this process may actually use (affinity mask and cgroup quota), not the host's logical CPU count."""
    import importlib

    from grip_amd.data.decode import usable_cpus
    try:
        os.sched_setaffinity(0, range(os.cpu_count() or 1))      # undo the GPU-side NUMA pin for the host measurement
    except OSError:
        pass
    threads = max(1, min(torch.get_num_threads(), usable_cpus()))
    torch.set_num_threads(threads)
    oclip = importlib.import_module("oracle.clip")
    t0 = time.perf_counter()
    om, _ = oclip.load(MODEL)
    build_s = time.perf_counter() - t0
    C = args.classes
    tok = synth_tokens(C, 0)
    g = torch.Generator().manual_seed(1234)
    budget = float(args.cpu_budget)
    n_max = max(args.cpu_sample, 1)
    x = torch.randn(max(n_max, 48), 3, 224, 224, generator=g)
    with torch.no_grad():
        om(x[:1], tok[:2])   # warm-up
        t_full, start = [], time.perf_counter()
        reps_max = n_max if args.cpu_full else max(1, min(args.cpu_text_reps, n_max))
        while len(t_full) < reps_max and (not t_full or args.cpu_full or time.perf_counter() - start < budget / 3):
            i = len(t_full)      # the literal loop body: both towers + softmax + argmax
            t0 = time.perf_counter()
            li, _ = om(x[i:i + 1], tok)
            li.softmax(dim=-1).argmax(dim=1)
            t_full.append(time.perf_counter() - t0)
        t_img, start = [], time.perf_counter()
        while len(t_img) < n_max and (len(t_img) < 4 or time.perf_counter() - start < budget / 3):
            i = len(t_img)       # the image tower at batch 1
            t0 = time.perf_counter()
            om.encode_image(x[i:i + 1])
            t_img.append(time.perf_counter() - t0)
        reps, n_r = len(t_full), len(t_img)
        t_text = max(float(np.mean(t_full)) - float(np.mean(t_img[:max(reps, 4)])), 0.0)    # the re-encode share of a full call
        per_image = float(np.mean(t_img)) + t_text if not args.cpu_full else float(np.mean(t_full))
        t0 = time.perf_counter()
        om.encode_text(tok)
        t_txt = time.perf_counter() - t0
        t_b, start = [], time.perf_counter()
        while len(t_b) < 3 and (not t_b or time.perf_counter() - start < budget / 3):
            b = len(t_b)
            t0 = time.perf_counter()
            om.encode_image(x[16 * b:16 * b + 16])
            t_b.append(time.perf_counter() - t0)
    b_ips = 16 * len(t_b) / float(np.sum(t_b))
    return {
        "value": 1.0 / per_image, "unit": "images/sec", "cores": threads, "kind": "port",
        "cpu_model": cpu_model_string(), "cpu_count": os.cpu_count(), "usable_cpus": usable_cpus(),
        "sample": f"R-mode (reference loop: batch 1, {C} class prompts re-encoded per image): image tower timed on {n_r} images "
                  f"({np.mean(t_img) * 1e3:.0f} ms mean), full clip_model(image, text) calls timed on {reps} ({np.mean(t_full):.2f} s mean) "
                  f"and their text share applied to every image; B-mode (batch 16, text cached, {len(t_b)} batches) = {b_ips:.2f} images/sec, "
                  f"one text encode of {C} prompts {t_txt:.2f} s; oracle build {build_s:.0f} s; {threads} torch threads; wall budget {budget:.0f} s",
        "r_mode_images": n_r, "r_mode_full_calls_timed": reps,
        "b_mode_images_per_sec": b_ips, "b_mode_batches": len(t_b),
        # the host CPUs are shared with other tenants (r04 -> r05: R-mode 0.70 -> 2.11, B-mode 30.7 -> 42.6 img/s on the same CPU model and thread count): the
        # spread INSIDE this run, so that a reader can tell a quiet host from a busy one
        "spread": {"r_mode_image_tower_ms_min_median_max": [float(np.min(t_img)) * 1e3, float(np.median(t_img)) * 1e3, float(np.max(t_img)) * 1e3],
                   "r_mode_full_call_s_min_max": [float(np.min(t_full)), float(np.max(t_full))],
                   "b_mode_images_per_sec_per_batch": [16.0 / t for t in t_b]},
    }


# ------------------------------------------------------------------------------------------------ exact + secondary blocks
def exact_block(loop, lib):
    """The fp32 comparison mode on the SAME resident pool, outside the timed region: images/sec of its pseudolabel pass (every row
    through the f32 towers); whether the timed loop's lists ARE its lists (the index guarantee of --mode identical, checked here
    on the full pool); the same pass under a second chunking (rows must not depend on the chunk they are encoded in: lists
    bit-for-bit); and one plain f16 pass for the rate and the overlap of the un-refined f16 lists."""
    a = loop.args
    em = loop.twin if loop.twin is not None else clip.load(MODEL, device=loop.device, exact=True)[0]
    with torch.no_grad():
        em.encode_image(loop.pool[:64])
    torch.cuda.synchronize()
    img_l, cls_l = loop.last_lists
    t0 = time.perf_counter()
    img_e, cls_e = loop.pseudolabel_pass(em, 1)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    keep, a.exact_chunk = a.exact_chunk, a.exact_chunk * 3 // 2
    img_e2, cls_e2 = loop.pseudolabel_pass(em, 1)
    a.exact_chunk = keep
    mode, a.mode = a.mode, "f16"
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    img_h, cls_h = loop.pseudolabel_pass(loop.m, a.streams)
    torch.cuda.synchronize()
    dt_h = time.perf_counter() - t0
    a.mode = mode
    pe, ph = set(zip(img_e.tolist(), cls_e.tolist())), set(zip(img_h.tolist(), cls_h.tolist()))
    f_exec = F_IMG if os.environ.get("GRIP_LAST_BLOCK_FULL", "0") not in ("", "0") else vit_flops_executed()
    return {"exact_images_per_sec": loop.n_total / dt, "pool_images": loop.n_total, "dtype": "f32",
            # (executed FLOPs: since r04 the f32 tower's last block runs for the CLS row only, like the f16 tower's)
            "achieved_tflops": loop.n_total * f_exec / dt / 1e12, "algorithmic_tflops": loop.n_total * F_IMG / dt / 1e12, "peak_tflops": PEAK_F32_TFLOPS,
            "frac": loop.n_total * f_exec / dt / 1e12 / PEAK_F32_TFLOPS,
            "pairs_exact": len(pe),
            "timed_loop_lists_identical_to_exact": bool(np.array_equal(img_l, img_e) and np.array_equal(cls_l, cls_e)),
            "timed_loop_mode": mode,
            "exact_lists_identical_under_second_chunking": bool(np.array_equal(img_e, img_e2) and np.array_equal(cls_e, cls_e2)),
            "chunkings": [keep, keep * 3 // 2],
            "f16_pass_images_per_sec": loop.n_total / dt_h, "pairs_f16": len(ph), "pair_overlap_f16_vs_exact": len(pe & ph) / max(len(pe), 1),
            "f16_lists_identical_to_exact": bool(np.array_equal(img_h, img_e) and np.array_equal(cls_h, cls_e)),
            "note": "f32 weights/activations/attention (v_mfma_f32_16x16x4_f32); list equality of this mode with the fp32 oracle / the reference's own "
                    "outputs is asserted in tests/test_gpu_exact.py, of the identical mode with this mode in tests/test_gpu_identical.py"}


def secondary_block(loop, lib):
    """The other BASELINE.json configs, outside the timed region: configs[2] VPT step (RESISC45-shaped: C = 45, 16 visual prompt
    tokens), configs[3] UPT step (DTD-shaped: C = 47, Pt = Pv = 4), configs[4] ViT-L/14@336px frozen encode (FGVCAircraft:
    the pool encode of GRIP textual; C = 100 text-L prompts).  Each with its dominant GEMM's TFLOP/s (library HIP events)."""
    dev, m, B = loop.device, loop.m, loop.args.batch
    x = loop.pool[:B]
    scale = m.logit_scale.exp().item()
    w = torch.full((B,), 1.0 / B, device=dev)

    def N(name, shape, std=0.02):
        return torch.from_numpy(rng.normal(1, rng.stream_id(name), shape, 0.0, std)).to(dev)

    def timed(fn, n, units, flops):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        lib.grip_profile_enable(1)
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        launches, ms, fl = profile_collect(lib)
        dom, tf = dominant(launches, ms, fl)
        return {"ms": dt * 1e3, "images_per_sec": units / dt, "algorithmic_tflops": flops / dt / 1e12,
                "dominant_kernel": kname(dom), "dominant_kernel_tflops": tf, "dominant_kernel_frac": tf / PEAK_F16_TFLOPS}

    out = {}
    C = 45
    txt = m.encode_text(synth_tokens(C, 0, seed=8).to(dev))
    im = ImagePrefixModel(N("v", (16, 768)), CustomImageEncoder(m.visual), device=dev)
    opt = torch.optim.SGD([im.prefix], lr=0.1, weight_decay=0.1)
    y = torch.randint(0, C, (B,), device=dev, dtype=torch.int32)
    out["vpt_step"] = dict(timed(lambda: steps.vpt_step(im, txt, scale, x, y, w, opt), 20, B, 2 * B * 38.09e9),
                           workload="configs[2] RESISC45-shaped VPT step: B = 16, 16 visual prompt tokens, C = 45, ViT-B/16 fwd + dgrad bwd + SGD")
    gv = steps.GraphedVptStep(im, txt, scale, opt)
    out["vpt_step"]["ms_hip_graph"] = timed(lambda: gv(x, y, w), 20, B, 2 * B * 38.09e9)["ms"]
    C = 47
    classes = [f"class_{i}" for i in range(C)]
    enc = CustomTextEncoder(m, dev, torch.float32)
    enc._tok_cache[(4, tuple(classes))] = synth_tokens(C, 4, seed=9).to(dev)
    um = UPTModel(N("uc", (1, 4, 512)), N("uv", (1, 4, 768)), None, CustomImageEncoder(m.visual), enc, classes, 128, device=dev, dtype=torch.float32)
    opt2 = torch.optim.SGD([p for p in um.parameters() if p.requires_grad], lr=0.01, weight_decay=0.1)
    y2 = torch.randint(0, C, (B,), device=dev, dtype=torch.int32)
    out["upt_step"] = dict(timed(lambda: steps.upt_step(um, scale, x, y2, w, opt2), 20, B, 2 * B * 35.87e9 + 2 * C * F_TXT),
                           workload="configs[3] DTD-shaped UPT step: B = 16, Pt = Pv = 4, C = 47, both towers fwd + bwd, mixer, SGD")
    gu = steps.GraphedUptStep(um, scale, opt2)
    out["upt_step"]["ms_hip_graph"] = timed(lambda: gu(x, y2, w), 20, B, 2 * B * 35.87e9 + 2 * C * F_TXT)["ms"]
    del gv, gu
    del im, um, opt, opt2
    big, _ = clip.load("ViT-L/14@336px", device=dev)
    xl = torch.randn(128, 3, 336, 336, device=dev)
    tokl = synth_tokens(100, 0, seed=10).to(dev)
    with torch.no_grad():
        out["vitl14_336_encode"] = dict(timed(lambda: big.encode_image(xl), 4, 128, 128 * 381.9e9),
                                        workload="configs[4] FGVCAircraft-shaped frozen ViT-L/14@336px encode, chunk 128 (S = 577, 24 layers, d = 1024)")
        t = timed(lambda: big.encode_text(tokl), 5, 100, 100 * 13.30e9)
        out["vitl14_336_encode"]["text_L_100_prompts_ms"] = t["ms"]
    del big, xl
    torch.cuda.empty_cache()
    return out


def refine_summary(rs):
    """What one screen-and-refine pass did (pseudolabels.refine_scan's stats under stable names)."""
    return {"rows_reencoded": rs["rows_refined"], "rows_reencoded_split_f16": rs["rows_mid"], "rows_reencoded_exactly": rs["rows_exact"], "tiers": rs["tiers"],
            "of_rows": rs["rows"], "fraction": rs["rows_refined"] / max(rs["rows"], 1), "nonfinite_screen_rows": rs.get("nonfinite_screen_rows", 0),
            "calibration_rows": rs["calibration_rows"], "rounds": rs["rounds"], "scans": rs["scans"], "rows_per_round": rs["refined_per_round"],
            "tier_calls": rs["tier_calls"], "screen_stream": rs.get("screen_stream"), "screen_stream_next_pass": rs.get("screen_stream_next_pass"), "screen_marked_share": rs.get("screen_marked_share"),
            "bound_form": rs["bound_form"], "bound": rs["eps"], "largest_deviation_seen": rs["max_deviation"], "bound_split_f16": rs["eps_mid"],
            "largest_deviation_seen_split_f16": rs["max_deviation_mid"], "safety": rs["safety"], "safety_split_f16": rs.get("safety_mid"),
            "audit_rows": rs["audit_rows"], "audit_board_rows": rs["audit_board_rows"], "audit_max_deviation": rs["audit_max_deviation"],
            "audit_widened_the_bound": rs["audit_widened"], "audits": rs["audits"], "audit_rows_split_f16": rs.get("audit_mid_rows", 0),
            "audit_max_deviation_split_f16": rs.get("audit_max_deviation_mid", 0.0), "unverified_rows": rs["unverified_rows"],
            "observed_rows": rs["observed_rows"]}


REFINE_NOTE = ("bound_form 'odds': every probability's odds p / (1 - p) are trusted to a factor e^{+-bound} (the form a logit error takes; 'relative': every "
               "probability to a relative bound); bound = safety x the largest such deviation between a tier's probabilities and the better ones that replaced "
               "them, over every row re-encoded so far; audit = hold-out rows re-encoded after certification: audit_max_deviation <= bound or the bound is "
               "widened and the scan repeats; unverified_rows = rows the lists take on trust within the bound")


def structured_pool_block(loop):
    """The identical-mode pass on a pool with class structure (per-image colour cast + low-frequency ramp on top of noise, the recipe of
    grip_amd.data.synthetic generated on the device): the timed pool is i.i.d. noise, on which the random-init tower gives every image the
    same arg-max -- the hardest case for the screen (every image spills to every class).  Reported: rows re-encoded, rounds, pass rate."""
    a = loop.args
    dev = loop.device
    n = a.pool
    g = torch.Generator(device=dev).manual_seed(4242)
    pool = torch.empty(n, 3, loop.d.image_resolution, loop.d.image_resolution, dtype=torch.float32, device=dev)
    ramp = torch.linspace(-1.0, 1.0, loop.d.image_resolution, device=dev).view(1, 1, 1, -1)
    for lo in range(0, n, 2048):
        hi = min(lo + 2048, n)
        x = torch.empty(hi - lo, 3, loop.d.image_resolution, loop.d.image_resolution, device=dev).normal_(generator=g)
        mu = torch.empty(hi - lo, 3, 1, 1, device=dev).normal_(generator=g) * 2.0
        r = torch.empty(hi - lo, 3, 1, 1, device=dev).normal_(generator=g)
        pool[lo:hi] = x * 0.5 + mu + ramp * r
    keep_pool, keep_stats, keep_stage = loop.pool, loop.refine_stats, dict(loop.stage)
    loop.pool = pool
    try:
        loop.identical_pass(a.streams)          # warm-up of nothing new; first pass on this pool
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        img, cls = loop.identical_pass(a.streams)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        rs = loop.refine_stats
        pred_hist = np.bincount(cls, minlength=a.classes)
        return dict(refine_summary(rs), pool_images=n, identical_images_per_sec=n / dt, pairs=int(len(img)), classes_with_a_full_board=int((pred_hist >= a.k).sum()),
                    note="same loop, same towers, structured synthetic pool (seeded on the device; not the fixtures' CPU generator)")
    finally:
        loop.pool, loop.refine_stats = keep_pool, keep_stats
        loop.stage.update(keep_stage)
        del pool
        torch.cuda.empty_cache()


def device_structured_pool(n, res, dev, seed):
    """The structured synthetic pool generated on the device: noise + a per-image colour cast + a low-frequency ramp (the recipe of grip_amd.data.synthetic)."""
    g = torch.Generator(device=dev).manual_seed(seed)
    pool = torch.empty(n, 3, res, res, dtype=torch.float32, device=dev)
    ramp = torch.linspace(-1.0, 1.0, res, device=dev).view(1, 1, 1, -1)
    for lo in range(0, n, 2048):
        hi = min(lo + 2048, n)
        pool[lo:hi] = torch.empty(hi - lo, 3, res, res, device=dev).normal_(generator=g) * 0.5 + torch.empty(hi - lo, 3, 1, 1, device=dev).normal_(generator=g) * 2.0 \
            + ramp * torch.empty(hi - lo, 3, 1, 1, device=dev).normal_(generator=g)
    return pool, g


def peaked_pool_block(loop, kind):
    """The index guarantee OFF the degenerate statistics of the timed pool, at the bench size, outside the timed region.  Both
    kinds run the structured pool against class PROTOTYPES as text features, which gives peaked rows with contested arg-maxes:
      "realistic"  the standard synthetic model; text feature of class c = the unit vector along m + 2 (e_c - m), e_c an anchor image's embedding and
                   m the pool's mean direction: un-centred (cosine 0.68 with the mean direction, as CLIP text features keep a common component), mean top-1 probability
                   ~0.6, every class owns arg-maxes, and an f16 direction error of 1e-3 is a logit error of a few 1e-3 -- the regime of a trained
                   CLIP at logit scale 100 far more than the timed pool's near-uniform softmax is;
      "stress"     `clip.load(..., synthetic="stress")` -- outlier channels (|x| ~ 200) in the vision residual stream and an f16 overflow on about a
                   fifth of the images (weights.stress_state_dict) -- against MEAN-REMOVED prototypes, which amplify the embeddings' direction error
                   ~30x (logit errors of tenths): the adversarial end.
    Reported: identical-mode lists == exact-mode lists, rows per tier, non-finite screen rows, bound and audit, pass rate."""
    from grip_amd import clip as gclip
    a, dev = loop.args, loop.device
    n, C, k = a.pool, a.classes, a.k
    if kind == "stress":
        m, _ = gclip.load("ViT-B/16", device=dev, synthetic="stress")
        twin = m.exact_twin()
    else:
        m, twin = loop.m, loop.twin
    pool, g = device_structured_pool(n, 224, dev, 777)
    paths = [f"pool/{i:08d}.jpg" for i in range(n)]
    labels = list(range(C))
    try:
        with torch.no_grad():
            e32 = torch.empty(n, 512, device=dev)
            t0 = time.perf_counter()
            twin.visual.tower.encode_chunks(pool, e32, 0, n, a.exact_chunk, streams=1)
            torch.cuda.synchronize()
            t_exact = time.perf_counter() - t0
            en = e32 / e32.norm(dim=-1, keepdim=True)
            anchors = torch.randperm(n, generator=g, device=dev)[:C]
            mean = en.mean(0, keepdim=True)
            if kind == "stress":
                txt = (en[anchors] - mean + 0.003 * torch.empty(C, 512, device=dev).normal_(generator=g)).contiguous()
            else:
                txt = mean + 2.0 * (en[anchors] - mean)
                txt = (txt / txt.norm(dim=-1, keepdim=True)).contiguous()
            common = float(((txt / txt.norm(dim=-1, keepdim=True)) @ (mean / mean.norm()).T).mean())
            _, p32, _, a32 = engine.cosine_head(e32, txt, 100.0)
        p32h, a32h = p32.cpu().numpy(), a32.cpu().numpy()
        want = pl.leaderboard(p32h, a32h, paths, labels, k)
        lg = np.log(np.maximum(p32h, 1e-45))
        mid = pl.mid_tower(m, n)
        kw = dict(chunk=a.chunk, exact_chunk=a.exact_chunk, visual_mid=mid, mid_chunk=a.exact_chunk)
        pl.identical_lists(m.visual.tower, twin.visual.tower, pool, txt, 100.0, paths, labels, k, **kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        got = pl.identical_lists(m.visual.tower, twin.visual.tower, pool, txt, 100.0, paths, labels, k, **kw)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        rs = pl.LAST_REFINE_STATS
        return {"pool_images": n, "classes": C, "k": k, "lists_identical_to_exact_mode": (list(got[0]), list(got[1])) == (list(want[0]), list(want[1])),
                "identical_images_per_sec": n / dt, "exact_mode_images_per_sec": n / t_exact,
                "logit_spread_max_minus_median": float(np.mean(lg.max(1) - np.median(lg, 1))), "mean_top_probability": float(p32h.max(1).mean()),
                "distinct_argmax_classes": int(len(np.unique(a32h))), "text_features_cosine_with_the_pool_mean": common,
"refine": refine_summary(rs),
                "model": ("ViT-B/16 synthetic-stress (weights.stress_state_dict): four residual-stream channels at x ~ +200 on every token (LayerNorm gains compensated; fp32 vs fp64 "
                          "oracle 1e-7), last block scaled so that one stream channel of the CLS row leaves the f16 range on ~ 1/5 of the images; text features = mean-removed "
                          "prototypes of the pool's own embeddings") if kind == "stress" else
                         "the timed loop's model; text features = unit blends m + 2 (e_c - m) of anchor embeddings e_c and the pool's mean direction m"}
    finally:
        del pool
        torch.cuda.empty_cache()


def from_files_block(loop, n=1536, chunk=384, procs=None, slots=1):
    """SURVEY.md 8f-2 (the reference's data/dataset.py:56-89 + utils/clip_pseudolabels.py:31-33: PIL open + transform per image on
    the host): images/sec from JPEG FILES to embeddings -- parallel decode on the host (threads, and worker processes around a
    shared-memory segment), one upload + one batched preprocess launch pair per chunk, the f16 ViT-B/16 encode of chunk i running
    while chunk i + 1 decodes.  Files are synthetic JPEGs of ImageNet-like sizes written to a temp directory outside every timed
    region; each configuration is timed over two passes after a warm-up pass."""
    import shutil
    import tempfile
    from concurrent.futures import ThreadPoolExecutor

    from PIL import Image

    from grip_amd.data.decode import default_processes, usable_cpus
    from grip_amd.preprocess import ClipPreprocess
    d = tempfile.mkdtemp(prefix="grip_bench_files_")
    g = np.random.RandomState(0)
    paths = []
    try:
        for i in range(n):      # low-frequency content so that the JPEG sizes are realistic
            h, w = int(g.choice([375, 333, 500, 480])), int(g.choice([500, 400, 640]))
            base = g.randint(0, 256, size=(h // 16 + 1, w // 16 + 1, 3)).astype(np.uint8)
            p = os.path.join(d, f"{i:05d}.jpg")
            Image.fromarray(base).resize((w, h), Image.BICUBIC).save(p, quality=90)
            paths.append(p)
        pre = ClipPreprocess(loop.d.image_resolution, loop.device)
        tower = loop.m.visual.tower
        emb = torch.empty(chunk, loop.d.embed_dim, dtype=torch.float32, device=loop.device)
        spans = [(lo, min(lo + chunk, n)) for lo in range(0, n, chunk)]
        bg = ThreadPoolExecutor(max_workers=1)

        def one_pass(encode, **kw):
            fut = bg.submit(pre.decode_chunk, paths[spans[0][0]:spans[0][1]], **kw)
            for i, (lo, hi) in enumerate(spans):
                h = fut.result()
                if i + 1 < len(spans):
                    fut = bg.submit(pre.decode_chunk, paths[spans[i + 1][0]:spans[i + 1][1]], **kw)
                x = pre.finish_chunk(h)
                if encode:
                    with torch.no_grad():
                        tower.encode_chunks(x, emb, 0, hi - lo, hi - lo, streams=1)
            torch.cuda.synchronize()

        def rate(encode, passes=2, **kw):
            one_pass(encode, **kw)
            t0 = time.perf_counter()
            for _ in range(passes):
                one_pass(encode, **kw)
            return passes * n / (time.perf_counter() - t0)

        procs = default_processes() if procs is None else procs
        out = {"files": n, "chunk": chunk, "usable_cpus": usable_cpus(), "host_cpu_count": os.cpu_count(),
               "threads_8_decode_preprocess_encode": rate(True, workers=8, processes=0)}
        if procs > 0:
            out["decode_processes"] = procs
            out["processes_decode_preprocess"] = rate(False, workers=8, processes=procs)
            out["processes_decode_preprocess_encode"] = rate(True, workers=8, processes=procs)
            out["per_cpu_images_per_sec"] = out["processes_decode_preprocess_encode"] / max(usable_cpus(), 1)
        out["images_per_sec"] = out.get("processes_decode_preprocess_encode", out["threads_8_decode_preprocess_encode"])
        out["note"] = ("JPEG files -> embeddings; decode is host-CPU bound (libjpeg-turbo through Pillow, bit-identical to the reference's transform): "
                       "the rate scales with the CPUs the container may use, the encoder alone runs at `pseudolabel_images_per_sec`")
        pre.close()
        return out
    finally:
        shutil.rmtree(d, ignore_errors=True)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--pool", type=int, default=50000, help="images per GPU (weak scaling)")
    ap.add_argument("--chunk", type=int, default=1320, help="images per encode launch: 1320 x 197 = 260 040 token rows = 1016 M-tiles of 256; every persistent GEMM workgroup walks >= 12 tiles (440: 20.1k img/s, 880-1760: 20.3k)")
    ap.add_argument("--streams", type=int, default=1, choices=(1, 2),
                    help="1: every kernel on one stream, so the per-kernel HIP-event / rocprof durations behind the roofline block are exclusive; "
                         "2: alternate encode chunks on two streams (what pseudolabels.encode_pool does by default; +1 %% images/s since the GEMMs are persistent, per-kernel durations overlap)")
    ap.add_argument("--classes", type=int, default=102)
    ap.add_argument("--prefix", type=int, default=16)
    ap.add_argument("--k", type=int, default=16)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--cpu-sample", type=int, default=64, help="images of the R-mode CPU baseline (BASELINE.md section 3: >= 64)")
    ap.add_argument("--cpu-text-reps", type=int, default=6, help="how many of them time the full clip_model(image, text) call (the per-image text re-encode)")
    ap.add_argument("--cpu-budget", type=float, default=40.0, help="wall-clock seconds the CPU baseline may take (it stops sampling when they are spent)")
    ap.add_argument("--cpu-full", action="store_true", help="time the literal R-mode loop on every sampled image (~6 s each)")
    ap.add_argument("--lookahead", type=int, default=51,
                    help="CoOp steps whose frozen image-tower forward is batched into one encode (steps.lookahead_image_features); 1 = encode inside every step")
    ap.add_argument("--graph", type=int, default=1, choices=(0, 1),
                    help="1: the CoOp step's forward + backward replayed from a HIP graph captured once (steps.GraphedCoopStep); 0: eager launches")
    ap.add_argument("--mode", default="identical", choices=("identical", "f16"),
                    help="identical (default, what utils.pseudolabel_top_k does): the pass returns the fp32 scan's lists -- f16 encode of the pool, "
                         "error-bounded scan, exact (f32) re-encode of the rows it marks; f16: the f16 towers' own lists (boundary items may differ)")
    ap.add_argument("--strong", action="store_true",
                    help="strong scaling: --pool is the TOTAL number of images, split evenly over the ranks (default: weak, --pool images per GPU)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-exact", action="store_true", help="skip the fp32 comparison-mode block")
    ap.add_argument("--exact-chunk", type=int, default=880, help="images per launch of the f32 towers (refinement rounds and the exact block): 220 -> 3 182, 440 -> 3 244, 880 -> 3 282 img/s; rows are bit-identical under any chunking")
    ap.add_argument("--no-secondary", action="store_true", help="skip the VPT / UPT / ViT-L/14@336px block")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        # plain `python bench.py --gpus N`: become N ranks (one per GPU) under torch.distributed.run on the loopback address
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.execv(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
                                  "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:])

    # The bench's synthetic weights sit on the f16 grid, as the weights of every published CLIP checkpoint do (fp16 archives; the reference's CPU path computes in
    # fp32 on those values cast up): the f32 twin then multiplies exactly the numbers the f16 towers hold, as it would with real weights.  GRIP_SYNTHETIC_FP16=0:
    # the un-rounded seeded init of rounds 1-5 (the test fixtures keep it).  Set here, not at import: tests and tools import this module for its helpers.
    os.environ.setdefault("GRIP_SYNTHETIC_FP16", "1")
    rank, ws = gdist.init_from_env()
    if ws != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={ws}")
    total_images = args.pool
    if args.strong:
        args.pool = (args.pool + ws - 1) // ws
    local_rank = gdist.local_device_index()
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    numa = gdist.pin_to_gpu_numa_node(local_rank)        # host threads (scan, launches) next to this rank's GPU
    lib = native.lib()
    on_host = ws > 1 and torch.distributed.get_backend() == "gloo"
    seen = [{"rank": rank, "device": local_rank, "numa_node": numa}]
    if ws > 1:
        gathered = [None] * ws
        torch.distributed.all_gather_object(gathered, seen[0])
        seen = gathered

    loop = Loop(args, device, rank, ws)
    clock_marker("warmup")
    for _ in range(args.warmup):
        loop.step()
    loop.t_pl = loop.t_tr = 0.0
    for k in loop.stage:
        loop.stage[k] = 0.0
    lib.grip_profile_enable(1)
    sampler = clock_sampler() if rank == 0 else None
    gdist.barrier()
    torch.cuda.synchronize()
    clock_marker("timed")
    if sampler is not None:
        sampler.start()
    gdist.trace("marker timed_begin")
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loop.step()
    gdist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    gdist.trace("marker timed_end")
    clocks = sampler.stop() if sampler is not None else None
    clock_marker("after_timed")
    el = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if on_host else device)
    if ws > 1:
        torch.distributed.all_reduce(el, op=torch.distributed.ReduceOp.MAX)
    elapsed = el.item()
    launches, ms, fl = profile_collect(lib)
    # wall seconds per stage of the timed loop on every rank: where a scaling curve bends is read off these
    stage_names = list(loop.stage)
    mine = torch.tensor([loop.stage[k] for k in stage_names], dtype=torch.float64, device="cpu" if on_host else device)
    allst = [torch.zeros_like(mine) for _ in range(ws)]
    if ws > 1:
        torch.distributed.all_gather(allst, mine)
    else:
        allst = [mine]
    allst = torch.stack(allst).cpu().numpy()
    stages = {k: {"max_s": float(allst[:, i].max()), "min_s": float(allst[:, i].min())} for i, k in enumerate(stage_names)}
    gather_bytes = args.steps * loop.n_total * loop.d.embed_dim * 4
    stages["allgather"]["bytes_received_per_rank"] = gather_bytes * (ws - 1) // max(ws, 1)
    stages["allgather"]["achieved_GBps_per_rank"] = (gather_bytes * (ws - 1) / ws / stages["allgather"]["max_s"] / 1e9) if ws > 1 and stages["allgather"]["max_s"] > 0 else None
    f16_loop = None
    if args.mode == "identical":
        # the same loop once more WITHOUT the guarantee (the f16 towers' own lists), outside the timed region: what the index guarantee costs
        keep = (loop.t_pl, loop.t_tr, loop.m_selected, loop.train_steps, loop.last_lists, loop.refine_stats)
        args.mode = "f16"
        loop.t_pl = loop.t_tr = 0.0
        gdist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loop.step()
        gdist.barrier()
        torch.cuda.synchronize()
        el16 = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cpu" if on_host else device)
        if ws > 1:
            torch.distributed.all_reduce(el16, op=torch.distributed.ReduceOp.MAX)
        f16_loop = {"images_per_sec": loop.n_total / el16.item(), "pseudolabel_images_per_sec": loop.n_total / loop.t_pl, "steps": 1,
                    "index_guarantee": "none (f16 lists: boundary items may differ from the fp32 scan's; see exact.pair_overlap_f16_vs_exact)"}
        args.mode = "identical"
        loop.t_pl, loop.t_tr, loop.m_selected, loop.train_steps, loop.last_lists, loop.refine_stats = keep
    if rank != 0:
        if ws > 1:
            gdist.barrier()
        return
    dom, achieved = dominant(launches, ms, fl)
    images = loop.n_total * args.steps
    seq = int(getattr(loop.coop_tokens, "_grip_seq_len", 77) or 77)
    seq_zs = int(loop.zs_tokens.argmax(-1).max().item()) + 1
    train_imgs = loop.train_steps * args.batch * ws * args.steps
    nominal = images * F_IMG + args.steps * args.classes * F_TXT * ws + args.steps * loop.train_steps * ws * (args.batch * F_IMG + 2 * args.classes * F_TXT)
    f_img_x = F_IMG if os.environ.get("GRIP_LAST_BLOCK_FULL", "0") not in ("", "0") else vit_flops_executed()
    executed = images * f_img_x + args.steps * args.classes * text_flops(seq_zs) * ws \
        + args.steps * loop.train_steps * ws * (args.batch * f_img_x + 2 * args.classes * text_flops(seq))
    traffic, mfma_util = pmc_entry(dom)
    peak = PEAK_F32_TFLOPS if dom < 16 else PEAK_F16_TFLOPS       # profiler variant 0 = the exact tower's f32 GEMM
    rs = loop.refine_stats
    if rs is not None:      # the rows the identical pass re-encoded with the f32 tower are work the engine issued on top of the algorithmic count
        executed += args.steps * (rs.get("rows_mid", 0) + rs.get("rows_exact", rs["rows_refined"])) * f_img_x       # (every re-encode of either tier; rows-only last block since r04)
    out = {
        "metric": "images/sec CLIP ViT-B/16 encode+prompt-step",
        "value": images / elapsed,
        "unit": "images/sec",
        "n_gpus": ws, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "strong" if args.strong else "weak", "vs_baseline": None,
        "dtype": "f16", "data": "synthetic",
        "config": {"workload": "Flowers102-shaped CoOp textual-prompt SSL pseudolabel+prompt-step loop, ViT-B/16 (BASELINE.json configs[1])",
                   "pseudolabel_mode": args.mode,
                   "comparable_to_earlier_rounds": "rounds 1-2 timed the f16 lists (no index guarantee): compare their `value` with `f16_mode_loop.images_per_sec` "
                                                   "of this line; `value` here is the identical mode (the default of utils.pseudolabel_top_k since round 3)"
                                                   if args.mode == "identical" else "same path as rounds 1-2",
                   "index_guarantee": ("calibrated bound, audited: every timed pass returns the lists the error-bounded scan CERTIFIES to be the fp32 (exact-mode) "
                                       "scan's provided every row obeys the measured bound of the tier it was left at -- f16 towers screen the pool, the scan marks "
                                       "the rows whose probabilities cannot decide a comparison the lists depend on, more accurate towers (split-f16, then f32) "
                                       "re-encode those, and after certification a hold-out sample of un-refined rows is re-encoded to check the bound "
                                       "(`identical.audit_*`; pseudolabels.refine_scan).  Equality with the exact mode on this very pool is checked in the "
                                       "`exact` block of this run and asserted in tests/test_gpu_identical.py")
                                      if args.mode == "identical" else
                                      "none: f16 lists (boundary items may differ from the fp32 scan's); run with --mode identical for the guarantee",
                   "pool_images_per_gpu": args.pool, "pool_images_total": loop.n_total, "classes": args.classes, "prompt_tokens": args.prefix, "k": args.k,
                   "encode_chunk": args.chunk, "encode_streams": args.streams, "train_batch_per_gpu": args.batch, "parallelism": f"dp{ws}",
                   "selected_pairs": int(loop.m_selected), "prompt_steps_per_pass": int(loop.train_steps),
                   "text_positions_encoded": seq, "prompt_step_hip_graph": bool(args.graph),
                   "train_image_lookahead": f"{args.lookahead} steps: the frozen image tower encodes the batches of {args.lookahead} consecutive prompt steps in one forward "
                                            "(every image is encoded every time a step uses it; nothing is cached)" if args.lookahead > 1 else "1 (encode inside every step)",
                   "last_block_rows_only": f_img_x != F_IMG,
                   "weights": "synthetic seeded init" + (", matrix weights rounded to f16 numbers as in the published fp16 checkpoints (weights.on_f16_grid)"
                                                         if os.environ.get("GRIP_SYNTHETIC_FP16") == "1" else ""),
                   "screen_stream": None if rs is None else f"{os.environ.get('GRIP_SCREEN_STREAM', 'auto')}: the last timed pass screened with the {rs.get('screen_stream')} residual stream",
                   "train_sharding": "the product trainer's sharding (dist.rank_batches = accelerate's even batches): batch j of the selected pairs in list order goes to rank j % N, "
                                     "batch 16 per rank, tail padded from the start; the selected images are all-gathered once per pass (each rank contributes its shard's); "
                                     "prompt gradients are mean-all-reduced every step -- not one global batch split over ranks",
                   "collectives": "RCCL all_gather_into_tensor of [pool, 512] f32 embeddings per pass (+ one of the refined rows per round, + one of the <= k C selected images for the prompt steps) + all_reduce of the 32 KB prompt gradient per step"
                                  if not on_host else "gloo through host memory (GRIP_DIST_BACKEND=gloo)"},
        "ranks_seen": seen,
        "stage_seconds_over_ranks": stages,
        "pseudolabel_images_per_sec": images / loop.t_pl if loop.t_pl else None,
        "identical_images_per_sec": (images / loop.t_pl if loop.t_pl else None) if args.mode == "identical" else None,
        "f16_mode_loop": f16_loop,
        "identical": None if rs is None else dict(refine_summary(rs), note="last timed pass; " + REFINE_NOTE),
        "train_images_per_sec": train_imgs / loop.t_tr if loop.t_tr else None,
        "algorithmic_tflops": nominal / elapsed / 1e12 / ws,
        "executed_tflops": executed / elapsed / 1e12 / ws,
        "flops_note": "per GPU; algorithmic = BASELINE.md section 2 (35.13 GF per image, 77 text positions per prompt); executed = what the engine "
                      f"issues: {f_img_x / 1e9:.2f} GF per frozen image forward (the last block's Q / attention / out-proj / MLP only for the CLS row: no other "
                      f"row of its output is read) and the text tower's {seq_zs} (zero-shot) / {seq} (CoOp) encoded positions (positions after the last "
                      "EOT cannot influence any output), plus every row the refinement tiers re-encode (counted once per re-encode at the same GF; the split-f16 "
                      "tier issues three f16 MFMA products per multiply-add: not counted); results are identical either way",
        "roofline": {
            "bound": "mfma", "kernel": kname(dom), "kernel_instantiation": full_kernel_name(dom),
            "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
            "clock_ghz_sustained": clocks["sclk_mhz_mean"] / 1e3 if clocks else None,
            "frac_at_sustained_clock": achieved / (peak * clocks["sclk_mhz_mean"] / PEAK_CLOCK_MHZ) if clocks else None,
            "clock_power": clocks,
            "clock_note": f"shader clock / package power sampled at 20 Hz from the amdgpu hwmon files over the whole timed region (all kernels, host gaps "
                          f"included); `peak` is quoted at {PEAK_CLOCK_MHZ / 1e3:.1f} GHz, frac_at_sustained_clock scales it to the mean clock measured; "
                          "profiles/r03_clock_power.csv holds the trace with the pure-MFMA zero / random-operand loops beside it",
            "traffic": traffic, "mfma_util_pmc": mfma_util,
            "traffic_source": f"{TRAFFIC_FILE}: separate rocprofv3 --pmc passes of this command (FETCH_SIZE x 2 + WRITE_SIZE; SQ_VALU_MFMA_BUSY_CYCLES), "
                              "read from the committed file, not measured in this run; null when the file does not describe exactly `kernel_instantiation`",
            "launches_timed": int(launches[dom]), "avg_launch_ms": ms[dom] / max(launches[dom], 1),
            "all_gemm": {kname(i): {"launches": int(launches[i]), "ms": round(float(ms[i]), 3),
                                        "tflops": round(float(fl[i] / (ms[i] * 1e-3) / 1e12), 1) if ms[i] > 0 else None}
                         for i in range(N_SLOTS) if launches[i]},
        },
    }
    if ws == 1:
        if not args.no_exact:
            clock_marker("exact")
            out["exact"] = exact_block(loop, lib)
            if args.mode == "identical" and not out["exact"]["timed_loop_lists_identical_to_exact"]:
                # never quote a guarantee the run itself contradicts
                out["config"]["index_guarantee"] = "VIOLATED in this run: the timed loop's lists differ from the exact mode's on this pool (see `exact`)"
                print("bench.py: identical-mode lists DIFFER from the exact mode's lists on this pool", file=sys.stderr, flush=True)
        structured = None
        if not args.no_secondary and args.mode == "identical":
            try:
                structured = structured_pool_block(loop)
            except Exception as e:
                structured = {"error": f"{type(e).__name__}: {e}"}
        if not args.no_secondary:
            del loop.pool
            loop.pool = synth_pool(64, loop.d.image_resolution, device, 99)
            torch.cuda.empty_cache()
            clock_marker("secondary")
            out["secondary"] = secondary_block(loop, lib)
            if structured is not None:
                out["secondary"]["identical_on_structured_pool"] = structured
                # the headline pool is degenerate (i.i.d. noise through random-init towers: one class wins nearly every arg-max); the same pass on a
                # pool WITH class structure re-encodes more rows -- quoted beside the headline, not instead of it (BASELINE.json names synthetic data)
                out["config"]["structured_pool_identical_pass_images_per_sec"] = structured.get("identical_images_per_sec")
                out["config"]["headline_pool_note"] = ("i.i.d. N(0,1) images (SURVEY 8d): a random-init tower gives nearly every image the same arg-max; "
                                                       "`structured_pool_identical_pass_images_per_sec` is the pseudolabel pass (no prompt steps) on a class-structured pool, "
                                                       "to be compared with `identical_images_per_sec`")
            if args.mode == "identical":
                for key, kind in (("identical_on_realistic_pool", "realistic"), ("identical_on_stress_model", "stress")):
                    try:
                        out["secondary"][key] = peaked_pool_block(loop, kind)
                    except Exception as e:
                        out["secondary"][key] = {"error": f"{type(e).__name__}: {e}"}
            try:
                out["secondary"]["from_files"] = from_files_block(loop)
            except Exception as e:      # the input pipeline is a NEXT row: its failure must not take the bench line down
                out["secondary"]["from_files"] = {"error": f"{type(e).__name__}: {e}"}
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args)
    print(json.dumps(out), flush=True)
    if ws > 1:
        gdist.barrier()


if __name__ == "__main__":
    main()
