"""Developer probe (GPU): screen-and-refine lists against the exact mode's on fresh 50 000-image pools (other seeds than the bench / test pool),
several k and class counts.      python tools/identical_check.py [seed ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from grip_amd import clip, engine, pseudolabels as pl  # noqa: E402

seeds = [int(v) for v in sys.argv[1:]] or [4321, 777]
dev = torch.device("cuda", 0)
m, _ = clip.load("ViT-B/16", device=dev)
twin = m.exact_twin()
n = 50000
paths = [f"pool/{(i * 7919) % n:08d}_{i}.jpg" for i in range(n)]      # path order differs from dataset order
ok = True
for seed in seeds:
    pool = bench.synth_pool(n, 224, dev, seed)
    with torch.no_grad():
        e32 = torch.empty(n, 512, device=dev)
        twin.visual.tower.encode_chunks(pool, e32, 0, n, 880, streams=1)
        e16 = pl.encode_pool(m.visual.tower, pool, chunk=1320)
    for C, k in ((102, 16), (45, 16), (10, 64), (102, 1), (47, 300)):
        tok = bench.synth_tokens(C, 0, seed=seed + C).to(dev)
        with torch.no_grad():
            txt = twin.encode_text(tok)
        scale = m.logit_scale.exp().item()
        _, p32, _, a32 = engine.cosine_head(e32, txt, scale)
        want = pl.leaderboard(p32.cpu().numpy(), a32.cpu().numpy(), paths, list(range(C)), k)
        got = pl.identical_lists(m.visual.tower, twin.visual.tower, pool, txt, scale, paths, list(range(C)), k, emb16=e16)
        st = pl.LAST_REFINE_STATS
        same = (list(got[0]), list(got[1])) == (list(want[0]), list(want[1]))
        ok &= same
        print(f"seed {seed} C={C} k={k}: identical={same} pairs={len(want[0])} rows refined {st['rows_refined']} {st['refined_per_round']} eps {st['eps']:.2e}", flush=True)
    del pool
print("ALL IDENTICAL" if ok else "MISMATCH")
