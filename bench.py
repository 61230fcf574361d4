#!/usr/bin/env python3
"""bench.py -- images/sec of the CLIP ViT-B/16 pseudolabel + prompt-step loop on N MI355X.

    python bench.py [--gpus N --steps K --warmup W]          (N > 1: launched by torch.distributed.run)

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
library) and `cpu_baseline` (the CPU oracle on a bounded sample, rank 0, N = 1 only).
"""
import argparse
import ctypes
import json
import math
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)
import grip_amd  # noqa: E402
from grip_amd import clip, config, dist as gdist, engine, native, pseudolabels as pl, rng, steps  # noqa: E402
from grip_amd.models import CustomTextEncoder, TextPrefixModel  # noqa: E402

MODEL = "ViT-B/16"
F_IMG = 35.13e9          # algorithmic FLOPs per image, frozen ViT-B/16 forward
F_TXT = 5.96e9           # per class prompt, text tower forward
PEAK_F16_TFLOPS = 2500.0 # MI355X dense f16/bf16 MFMA peak (MI355X_MICROARCH.md)
EPI_NAMES = ["EPI_F32", "EPI_BIAS_F16", "EPI_BIAS_GELU_F16", "EPI_BIAS_RESID", "EPI_F16", "EPI_GELUGRAD_F16", "EPI_F32_SCALE"]


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
        self.t_pl = self.t_tr = 0.0
        self.m_selected = 0
        self.train_steps = 0

    def step(self):
        a = self.args
        t0 = time.perf_counter()
        with torch.no_grad():
            txt = self.m.encode_text(self.zs_tokens)
            # every rank encodes its own pool; embeddings are gathered in global (rank-major) order
            local = torch.empty(a.pool, self.d.embed_dim, dtype=torch.float32, device=self.device)
            self.m.visual.tower.encode_chunks(self.pool, local, 0, a.pool, a.chunk, streams=a.streams)
            emb = gdist.allgather_rows(local, self.n_total, a.pool)
            logits, probs, am_l, am_p = engine.cosine_head(emb, txt, self.m.logit_scale.exp().item())
            probs_h = probs.cpu().numpy()
            pred_h = am_p.cpu().numpy()
        img, cls = engine.leaderboard_scan(probs_h, pred_h, self.ranks, self.k)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        # (iv) prompt steps over the selected pairs that live in this rank's shard
        lo = self.rank * a.pool
        mine = (img >= lo) & (img < lo + a.pool)
        my_img = torch.from_numpy(img[mine] - lo).long().to(self.device)
        my_lab = torch.from_numpy(cls[mine]).to(self.device)
        n_steps = math.ceil(len(img) / (a.batch * self.ws)) if len(img) else 0
        for t in range(n_steps):
            if len(my_img):
                idx = (torch.arange(a.batch, device=self.device) + t * a.batch) % len(my_img)
                x, y = self.pool[my_img[idx]], my_lab[idx]
                w = torch.full((a.batch,), 1.0 / a.batch, device=self.device)
            else:   # no selected image lives in this rank's shard: same work, zero weight (it still joins the all-reduce)
                x, y = self.pool[: a.batch], torch.zeros(a.batch, dtype=torch.int32, device=self.device)
                w = torch.zeros(a.batch, device=self.device)
            steps.coop_step(self.model, self.m, x, y, w, self.opt)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        self.t_pl += t1 - t0
        self.t_tr += t2 - t1
        self.m_selected = len(img)
        self.train_steps = n_steps


def hbm_traffic(kernel):
    """HBM bytes per launch of `kernel` from the PMC passes (FETCH_SIZE x2 + WRITE_SIZE, collected in
    separate rocprofv3 --pmc runs of this same script; profiles/r01_traffic.json), or None."""
    try:
        with open(os.path.join(REPO, "profiles", "r01_traffic.json")) as f:
            t = json.load(f)["kernels"]
        return t[kernel.split(" [")[0]]["bytes_per_launch"]
    except Exception:
        return None


def pmc_mfma_util(kernel):
    """SQ_VALU_MFMA_BUSY_CYCLES / (GPU cycles x 1024 SIMDs) of `kernel` from the same PMC profile (fraction of the matrix
    pipes' cycles at the ACTUAL clock that an MFMA was executing), or None."""
    try:
        with open(os.path.join(REPO, "profiles", "r01_traffic.json")) as f:
            return json.load(f)["kernels"][kernel.split(" [")[0]].get("mfma_util")
    except Exception:
        return None


def cpu_baseline(args):
    """The CPU oracle (oracle/, a port of the reference path: kind "port") on a bounded sample.
    R-mode = reference-faithful loop of utils/clip_pseudolabels.py:31-41: batch 1, the full
    clip_model(image, text) per image, i.e. all C class prompts re-encoded for every image.
    B-mode (reported beside it) = batch 16 with text features cached."""
    import importlib
    oclip = importlib.import_module("oracle.clip")
    t0 = time.perf_counter()
    om, _ = oclip.load(MODEL)
    build_s = time.perf_counter() - t0
    C = args.classes
    tok = synth_tokens(C, 0)
    g = torch.Generator().manual_seed(1234)
    n_r = args.cpu_sample
    x = torch.randn(max(n_r, 16), 3, 224, 224, generator=g)
    with torch.no_grad():
        om(x[:1], tok[:2])   # warm-up
        t0 = time.perf_counter()
        for i in range(n_r):
            li, _ = om(x[i:i + 1], tok)
            li.softmax(dim=-1).argmax(dim=1)
        t_r = time.perf_counter() - t0
        t0 = time.perf_counter()
        txt = om.encode_text(tok)
        t_txt = time.perf_counter() - t0
        t0 = time.perf_counter()
        om.encode_image(x[:16])
        t_b = time.perf_counter() - t0
    return {
        "value": n_r / t_r, "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
        "sample": f"R-mode (reference loop: batch 1, {C} class prompts re-encoded per image) on {n_r} images; "
                  f"B-mode (batch 16, text cached) = {16 / t_b:.2f} images/sec, text encode {t_txt:.2f} s; "
                  f"host cpu_count={os.cpu_count()}, oracle build {build_s:.0f} s",
        "b_mode_images_per_sec": 16 / t_b,
    }


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
    ap.add_argument("--cpu-sample", type=int, default=6)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank, ws = gdist.init_from_env()
    if ws != args.gpus and not (ws == 1 and args.gpus == 1):
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={ws}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    local_rank = gdist.local_device_index()
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    lib = native.lib()

    loop = Loop(args, device, rank, ws)
    for _ in range(args.warmup):
        loop.step()
    loop.t_pl = loop.t_tr = 0.0
    lib.grip_profile_enable(1)
    gdist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loop.step()
    gdist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    el = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if (ws > 1 and torch.distributed.get_backend() == "gloo") else device)
    if ws > 1:
        torch.distributed.all_reduce(el, op=torch.distributed.ReduceOp.MAX)
    elapsed = el.item()

    n = 56   # slot = variant * 8 + epilogue id (csrc/gemm.hip)
    launches = np.zeros(n, dtype=np.int64)
    ms = np.zeros(n, dtype=np.float64)
    fl = np.zeros(n, dtype=np.float64)
    native.check(lib.grip_profile_collect(n, ctypes.c_void_p(launches.ctypes.data), ctypes.c_void_p(ms.ctypes.data), ctypes.c_void_p(fl.ctypes.data)))
    lib.grip_profile_enable(0)
    if rank != 0:
        return
    def kname(slot):
        v, e = divmod(slot, 8)
        return {1: f"gemm_f16_kernel<{e}, 4>", 4: f"gemm_f16_kernel<{e}, 2>", 2: f"gemm_big_kernel<{e}, 256, 256, 4>", 3: f"gemm_big_kernel<{e}, 256, 128, 3>",
                5: f"gemm_k64_kernel<{e}, 8>", 6: f"gemm_k64p_kernel<{e}>"}.get(v, f"gemm?<{e}>") + f" [{EPI_NAMES[e]}]"
    dom = int(np.argmax(ms))
    achieved = fl[dom] / (ms[dom] * 1e-3) / 1e12 if ms[dom] > 0 else 0.0
    images = loop.n_total * args.steps
    out = {
        "metric": "images/sec CLIP ViT-B/16 encode+prompt-step",
        "value": images / elapsed,
        "unit": "images/sec",
        "n_gpus": ws, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f16", "data": "synthetic",
        "config": {"workload": "Flowers102-shaped CoOp textual-prompt SSL pseudolabel+prompt-step loop, ViT-B/16 (BASELINE.json configs[1])",
                   "pool_images_per_gpu": args.pool, "classes": args.classes, "prompt_tokens": args.prefix, "k": args.k,
                   "encode_chunk": args.chunk, "encode_streams": args.streams, "train_batch_per_gpu": args.batch, "parallelism": f"dp{ws}",
                   "selected_pairs": int(loop.m_selected), "prompt_steps_per_pass": int(loop.train_steps),
                   "text_positions_encoded": int(getattr(loop.coop_tokens, "_grip_seq_len", 77) or 77)},
        "pseudolabel_images_per_sec": images / loop.t_pl if loop.t_pl else None,
        "train_images_per_sec": (loop.train_steps * args.batch * ws * args.steps) / loop.t_tr if loop.t_tr else None,
        "algorithmic_tflops": (images * F_IMG + args.steps * args.classes * F_TXT
                               + args.steps * loop.train_steps * ws * (args.batch * F_IMG + 2 * args.classes * F_TXT)) / elapsed / 1e12 / ws,
        "roofline": {
            "bound": "mfma", "kernel": kname(dom),
            "achieved": achieved, "peak": PEAK_F16_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_F16_TFLOPS,
            "traffic": hbm_traffic(kname(dom)), "mfma_util_pmc": pmc_mfma_util(kname(dom)),
            "launches_timed": int(launches[dom]), "avg_launch_ms": ms[dom] / max(launches[dom], 1),
            "all_gemm": {kname(i): {"launches": int(launches[i]), "ms": round(float(ms[i]), 3),
                                        "tflops": round(float(fl[i] / (ms[i] * 1e-3) / 1e12), 1) if ms[i] > 0 else None}
                         for i in range(n) if launches[i]},
        },
    }
    if ws == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
