"""Developer probe: ms per replay of the graphed CoOp step on pre-encoded features (steps.GraphedCoopFeatureStep: text tower forward + backward, head, loss,
SGD) -- what the bench loop's prompt steps cost apart from the look-ahead image encode.  `python tools/coop_graph_bench.py [steps]`."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import grip_amd  # noqa: E402,F401
from grip_amd import clip, rng, steps  # noqa: E402
from grip_amd.models import CustomTextEncoder, TextPrefixModel  # noqa: E402

dev = torch.device("cuda", 0)
m, _ = clip.load("ViT-B/16", device=dev)
C, P, B = 102, 16, 16
classes = [f"class_{i}" for i in range(C)]
enc = CustomTextEncoder(m, dev, torch.float32)
enc._tok_cache[(P, tuple(classes))] = bench.synth_tokens(C, P).to(dev)
tm = TextPrefixModel(torch.from_numpy(rng.normal(1, rng.stream_id("c"), (1, P, 512), 0.0, 0.02)).to(dev), enc, classes, device=dev)
opt = torch.optim.SGD([tm.prefix], lr=0.002, weight_decay=0.1)
g = steps.GraphedCoopFeatureStep(tm, m, opt)
f = torch.randn(B, 512, device=dev)
y = torch.randint(0, C, (B,), device=dev, dtype=torch.int32)
w = torch.full((B,), 1.0 / B, device=dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
for _ in range(10):
    g(f, y, w)
torch.cuda.synchronize()
best = 1e9
for rep in range(5):
    t = time.perf_counter()
    for _ in range(n):
        g(f, y, w)
    torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t) / n)
print(f"graphed CoOp feature step: {best * 1e3:.3f} ms per replay ({B / best:.0f} img/s at B = {B})  env: TRAIN_FOLD={os.environ.get('GRIP_TRAIN_FOLD', '2')} COOP_SPLIT={os.environ.get('GRIP_COOP_SPLIT', 'auto')}")
