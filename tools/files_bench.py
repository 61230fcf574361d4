"""Developer probe (GPU box): bench.py's secondary.from_files block (JPEG files -> embeddings) over several numbers of decode
processes and chunk sizes.      python tools/files_bench.py [procs,procs,...] [chunk,chunk,...]"""
import json
import os
import sys
import types

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402
from grip_amd import clip, config  # noqa: E402

procs = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [12, 14, 16, 20]
chunks = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [512]
dev = torch.device("cuda", 0)
m, _ = clip.load("ViT-B/16", device=dev)
loop = types.SimpleNamespace(m=m, d=config.get_dims("ViT-B/16"), device=dev)
for c in chunks:
    for p in procs:
        r = bench.from_files_block(loop, n=2048, chunk=c, procs=p)
        print(json.dumps({"procs": p, "chunk": c, **{k: (round(v) if isinstance(v, float) else v) for k, v in r.items() if k not in ("note",)}}), flush=True)
