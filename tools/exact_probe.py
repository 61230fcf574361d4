"""Developer probe: how far are the GPU's probabilities (exact f32 mode and default f16 mode) from the committed fp32 oracle
probabilities of the 2 000-image ViT-B/16 sample, and do the pseudolabel lists coincide?"""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import grip_amd  # noqa: E402,F401
from grip_amd import clip, engine, pseudolabels as pl  # noqa: E402
from grip_amd.data.synthetic import pool_paths, structured_images  # noqa: E402
from oracle import leaderboard as LB  # noqa: E402

fx = np.load(os.path.join(REPO, "tests", "golden", sys.argv[1] if len(sys.argv) > 1 else "exact_vitb16_c102.npz"))
o = fx["probs"]
n, C = o.shape
paths = pool_paths(n)
tok = torch.from_numpy(fx["tokens"]).cuda()
o_pred = o.argmax(1)
for exact in (True, False):
    m, _ = clip.load("ViT-B/16", device="cuda", exact=exact)
    emb = torch.empty(n, 512, device="cuda")
    with torch.no_grad():
        for lo in range(0, n, 250):
            emb[lo:lo + 250] = m.encode_image(structured_images(int(fx["seed"]), lo, min(lo + 250, n), 224).cuda())
        txt = m.encode_text(tok)
    _, gp, _, gpred = engine.cosine_head(emb, txt, m.logit_scale.exp().item())
    g, gpred = gp.cpu().numpy(), gpred.cpu().numpy()
    rel = np.abs(g.astype(np.float64) - o) / o
    print(f"exact={exact}: rel err max {rel.max():.3e} mean {rel.mean():.3e}; abs max {np.abs(g - o).max():.3e}; argmax equal {np.mean(gpred == o_pred):.5f}")
    for k in (3, 16, 10000000):
        want = LB.leaderboard_scan(o, o_pred, paths, list(range(C)), k)
        got = pl.leaderboard(g, gpred, paths, list(range(C)), k)
        a, b = set(zip(*want)), set(zip(*got))
        print(f"   k={k}: equal={got == want} pairs {len(a)} common {len(a & b)} margin(rel) {LB.scan_margin(o, o_pred, k):.3e}")
    del m
