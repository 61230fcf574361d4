"""Developer probe: CoOp steps on pre-encoded image features, eager (for rocprofv3 --stats): the text tower's forward + backward,
head, loss, SGD -- what steps.GraphedCoopFeatureStep replays inside the bench loop."""
import os
import sys

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
opt = torch.optim.SGD([tm.prefix], lr=0.1, weight_decay=0.1)
f = torch.randn(B, 512, device=dev)
y = torch.randint(0, C, (B,), device=dev, dtype=torch.int32)
w = torch.full((B,), 1.0 / B, device=dev)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 30):
    steps.coop_step(tm, m, None, y, w, opt, image_features=f)
torch.cuda.synchronize()
