"""Developer probe: bench.py's secondary block (VPT / UPT steps eager + HIP graph, ViT-L/14@336px encode) a few times over."""
import json
import os
import sys
import types

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402
from grip_amd import clip, config, native  # noqa: E402

dev = torch.device("cuda", 0)
m, _ = clip.load("ViT-B/16", device=dev)
loop = types.SimpleNamespace(m=m, d=config.get_dims("ViT-B/16"), device=dev, args=types.SimpleNamespace(batch=16), pool=bench.synth_pool(64, 224, dev, 99))
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    out = bench.secondary_block(loop, native.lib())
    print(json.dumps({k: {a: round(b, 3) for a, b in v.items() if isinstance(b, float)} for k, v in out.items()}), flush=True)
