"""Developer probe: the f16 / split-f16 / f32 GPU towers on variants of the stress model (weights.stress_state_dict), each against the CPU oracle's
embeddings of the same 16 images (tests/golden/stress_vitb16.npz, oracle/gen_golden_stress.py) and against each other on 1 024 images.
    python tools/stress_probe.py [variant ...]"""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import grip_amd  # noqa: E402,F401
from grip_amd import config as gcfg, pseudolabels as pl, weights  # noqa: E402
from grip_amd.clip.clip import load_openai_state_dict  # noqa: E402
from grip_amd.clip.model import CLIP  # noqa: E402
from grip_amd.data.synthetic import structured_images  # noqa: E402

VARIANTS = {"stress": {}, "outliers200": {"overflow_gain": (1.0, 1.0)}, "outliers50": {"overflow_gain": (1.0, 1.0), "outlier": 50.0}, "overflow": {"channels": ()}}
dev = torch.device("cuda", 0)
n = 1024
pool = structured_images(77, 0, n, 224).to(dev)
gold = np.load(os.path.join(REPO, "tests", "golden", "stress_vitb16.npz"))
d = gcfg.get_dims("ViT-B/16")
cosf = torch.nn.functional.cosine_similarity
for tag in sys.argv[1:] or list(VARIANTS):
    sd = {k: torch.from_numpy(v) for k, v in weights.stress_state_dict(d, 0, **VARIANTS[tag]).items()}
    emb = {}
    for prec in (0, 2, 1):
        m = CLIP(d, dev, exact=prec, vision_only=prec == 2)
        load_openai_state_dict(m, sd)
        with torch.no_grad():
            e = torch.empty(n, 512, device=dev)
            m.visual.tower.encode_chunks(pool, e, 0, n, 256, streams=1)
        emb[prec] = e
        del m
    o = torch.from_numpy(gold[tag]).to(dev)
    line = f"{tag}:"
    for prec, name in ((1, "f32"), (2, "split"), (0, "f16")):
        fin = torch.isfinite(emb[prec]).all(1)
        c = cosf(emb[prec][:16][fin[:16]], o[fin[:16]], dim=1)
        line += f"  {name}: finite {int(fin.sum())}/{n}, vs oracle 1-cos max {float((1 - c).max()):.2e}"
        if prec != 1:
            q = torch.quantile(1 - cosf(emb[prec][fin], emb[1][fin], dim=1), torch.tensor([0.5, 0.99, 1.0], device=dev)).tolist()
            line += f", vs f32 median {q[0]:.2e} p99 {q[1]:.2e} max {q[2]:.2e};"
    print(line, flush=True)
