"""Developer probe: images/s of the split-f16 (precision 2) and exact (f32) ViT-B/16 towers on a resident pool slice, and the GEMM rates of both
(in-library HIP-event profiler)."""
import ctypes
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from grip_amd import clip, native  # noqa: E402

n, chunk = (int(sys.argv[1]) if len(sys.argv) > 1 else 2640), (int(sys.argv[2]) if len(sys.argv) > 2 else 880)
dev = torch.device("cuda", 0)
m, _ = clip.load("ViT-B/16", device=dev)
pool = bench.synth_pool(n, 224, dev, 1)
lib = native.lib()
for name, model in (("split", m.split_twin()), ("exact", m.exact_twin()), ("f16", m)):
    out = torch.empty(n, 512, device=dev)
    with torch.no_grad():
        model.visual.tower.encode_chunks(pool, out, 0, min(n, chunk), chunk, streams=1)
        torch.cuda.synchronize()
        lib.grip_profile_enable(1)
        t = time.perf_counter()
        model.visual.tower.encode_chunks(pool, out, 0, n, chunk, streams=1)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
    launches, ms, fl = bench.profile_collect(lib) if hasattr(bench, "profile_collect") else (None, None, None)
    print(f"{name}: {n / dt:.0f} img/s (chunk {chunk}); checksum {out.double().sum().item():.6f}")
    if launches is not None:
        for i in np.flatnonzero(launches):
            print(f"    {bench.kname(int(i))}: {int(launches[i])} launches sampled, {ms[i] / launches[i]:.3f} ms avg, {fl[i] / (ms[i] * 1e-3) / 1e12:.1f} TF/s")
