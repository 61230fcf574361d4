"""Developer probe: images/s of the exact (f32) ViT-B/16 tower on a resident pool slice."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from grip_amd import clip  # noqa: E402

n, chunk = (int(sys.argv[1]) if len(sys.argv) > 1 else 3520), (int(sys.argv[2]) if len(sys.argv) > 2 else 440)
dev = torch.device("cuda", 0)
m, _ = clip.load("ViT-B/16", device=dev, exact=True)
pool = bench.synth_pool(n, 224, dev, 1)
out = torch.empty(n, 512, device=dev)
with torch.no_grad():
    m.visual.tower.encode_chunks(pool, out, 0, chunk, chunk, streams=1)
    torch.cuda.synchronize()
    t = time.perf_counter()
    m.visual.tower.encode_chunks(pool, out, 0, n, chunk, streams=1)
    torch.cuda.synchronize()
print(f"exact encode: {n / (time.perf_counter() - t):.0f} img/s (chunk {chunk}); checksum {out.double().sum().item():.6f}")
