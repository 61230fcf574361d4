"""Developer tool (GPU): the bench pool's probabilities in the f16 and the exact (f32) mode, written to gpurun_out/ so that the
screen-and-refine scan can be developed against real data on the CPU.  Also times the exact encode at several chunk sizes on
gathered (non-contiguous) rows and checks that an exact row does not depend on the chunk it is encoded in.

    python tools/dump_probs.py [--pool 50000] [--structured]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402
import grip_amd  # noqa: E402,F401
from grip_amd import clip, engine  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--pool", type=int, default=50000)
ap.add_argument("--classes", type=int, default=102)
ap.add_argument("--structured", action="store_true")
ap.add_argument("--structured-device", action="store_true", help="bench.py's device-generated structured pool (secondary.identical_on_structured_pool)")
ap.add_argument("--out", default="gpurun_out/probs_dump.npz")
a = ap.parse_args()
dev = torch.device("cuda", 0)
if a.structured_device:
    g = torch.Generator(device=dev).manual_seed(4242)
    pool = torch.empty(a.pool, 3, 224, 224, device=dev)
    ramp = torch.linspace(-1.0, 1.0, 224, device=dev).view(1, 1, 1, -1)
    for lo in range(0, a.pool, 2048):
        hi = min(lo + 2048, a.pool)
        x = torch.empty(hi - lo, 3, 224, 224, device=dev).normal_(generator=g)
        mu = torch.empty(hi - lo, 3, 1, 1, device=dev).normal_(generator=g) * 2.0
        r = torch.empty(hi - lo, 3, 1, 1, device=dev).normal_(generator=g)
        pool[lo:hi] = x * 0.5 + mu + ramp * r
elif a.structured:
    from grip_amd.data.synthetic import structured_images
    pool = torch.empty(a.pool, 3, 224, 224, device=dev)
    for lo in range(0, a.pool, 500):
        pool[lo:lo + 500] = structured_images(1234, lo, min(lo + 500, a.pool), 224).to(dev)
else:
    pool = bench.synth_pool(a.pool, 224, dev, 1234)
tok = bench.synth_tokens(a.classes, 0).to(dev)
res = {}
emb = {}
txt = {}
for exact in (False, True):
    m, _ = clip.load("ViT-B/16", device=dev, exact=exact)
    e = torch.empty(a.pool, 512, device=dev)
    with torch.no_grad():
        txt[exact] = m.encode_text(tok)
        m.visual.tower.encode_chunks(pool, e[:256], 0, 256, 128, streams=1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        m.visual.tower.encode_chunks(pool, e, 0, a.pool, 220 if exact else 1320, streams=1)
        torch.cuda.synchronize()
        print(f"exact={exact}: {a.pool / (time.perf_counter() - t0):.0f} img/s", flush=True)
        if exact:
            # gathered rows at several chunk sizes + chunk invariance
            idx = torch.randperm(a.pool, device=dev)[:2640]
            sub = pool[idx].contiguous()
            for chunk in (64, 110, 220, 440, 880):
                o = torch.empty(sub.shape[0], 512, device=dev)
                m.visual.tower.encode_chunks(sub, o, 0, sub.shape[0], chunk, streams=1)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                m.visual.tower.encode_chunks(sub, o, 0, sub.shape[0], chunk, streams=1)
                torch.cuda.synchronize()
                print(f"   exact gathered chunk {chunk}: {sub.shape[0] / (time.perf_counter() - t0):.0f} img/s; bit-identical to the pool pass: "
                      f"{bool((o == e[idx]).all().item())}", flush=True)
            for nsmall in (1, 7, 33):
                o = torch.empty(nsmall, 512, device=dev)
                m.visual.tower.encode_chunks(sub, o, 0, nsmall, nsmall, streams=1)
                print(f"   exact {nsmall} rows alone bit-identical: {bool((o == e[idx[:nsmall]]).all().item())}", flush=True)
    emb[exact] = e
    scale = m.logit_scale.exp().item()
    del m
    torch.cuda.empty_cache()
for name, (ie, te) in {"f16_f16txt": (False, False), "f16_exacttxt": (False, True), "exact": (True, True)}.items():
    _, p, am_l, am_p = engine.cosine_head(emb[ie], txt[te], scale)
    res["probs_" + name] = p.cpu().numpy()
    res["pred_" + name] = am_p.cpu().numpy()
    res["predl_" + name] = am_l.cpu().numpy()
p16, p32 = res["probs_f16_exacttxt"].astype(np.float64), res["probs_exact"].astype(np.float64)
rel = np.abs(p16 - p32) / p32
print(f"rel dev of p (f16 img, exact txt) vs exact: max {rel.max():.3e} mean {rel.mean():.3e} p99.9 {np.quantile(rel, 0.999):.3e}")
p16b = res["probs_f16_f16txt"].astype(np.float64)
relb = np.abs(p16b - p32) / p32
print(f"rel dev of p (f16 img, f16 txt) vs exact: max {relb.max():.3e} mean {relb.mean():.3e}")
de = (emb[False] - emb[True]).double()
print(f"embedding rel L2 max {(de.norm(dim=1) / emb[True].double().norm(dim=1)).max().item():.3e}")
os.makedirs(os.path.dirname(a.out), exist_ok=True)
np.savez(a.out, probs_f16=res["probs_f16_exacttxt"], probs_exact=res["probs_exact"], pred_f16=res["pred_f16_exacttxt"], pred_exact=res["pred_exact"], predl_exact=res["predl_exact"])
print("wrote", a.out, os.path.getsize(a.out) / 1e6, "MB")
