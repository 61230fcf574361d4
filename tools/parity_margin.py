"""Developer probe: how close are the native ViT-B/16 outputs to the golden (CPU fp32) vectors?"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import grip_amd  # noqa: E402
from grip_amd import clip, rng  # noqa: E402

g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "golden_vitb16.npz"))
m, _ = clip.load("ViT-B/16", device="cuda")
T = lambda name, shape, std=1.0: torch.from_numpy(rng.normal(100, rng.stream_id(name), shape, 0.0, std))
x = T("g3.x", (2, 3, 224, 224)).cuda()
vp = T("g3.vprefix", (16, 768), 0.02).cuda()
tp = T("g3.tprefix", (1, 16, 512), 0.02).cuda()
outs = {"vision_p0": m.encode_image(x), "vision_p16": m.visual(x, vp), "text_p0": m.encode_text(torch.from_numpy(g["g3.zs_tokens"]).cuda()),
        "text_p16": m.text_tower.text_forward(torch.from_numpy(g["g3.coop_tokens"]).cuda(), tp)[0]}
for k, v in outs.items():
    w = torch.from_numpy(g["g3." + k])
    v = v.cpu()
    cos = torch.nn.functional.cosine_similarity(v, w, dim=-1)
    print(f"{k:12s} 1-cos max {float((1 - cos).max()):.3e}   rel L2 {float((v - w).norm() / w.norm()):.3e}")
