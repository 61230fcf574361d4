"""Developer probe: UPT steps only (for rocprofv3 --kernel-trace); the shapes of bench.py's secondary block (configs[3])."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import grip_amd  # noqa: E402,F401
from grip_amd import clip, steps  # noqa: E402
from grip_amd.models import CustomImageEncoder, CustomTextEncoder, UPTModel  # noqa: E402

dev = torch.device("cuda", 0)
m, _ = clip.load("ViT-B/16", device=dev)
B, C = 16, 47
g = torch.Generator(device=dev).manual_seed(1)
N = lambda shape: torch.randn(*shape, device=dev, generator=g) * 0.02      # noqa: E731
x = torch.randn(B, 3, 224, 224, device=dev, generator=g)
classes = [f"class_{i}" for i in range(C)]
enc = CustomTextEncoder(m, dev, torch.float32)
enc._tok_cache[(4, tuple(classes))] = bench.synth_tokens(C, 4, seed=9).to(dev)
um = UPTModel(N((1, 4, 512)), N((1, 4, 768)), None, CustomImageEncoder(m.visual), enc, classes, 128, device=dev, dtype=torch.float32)
opt = torch.optim.SGD([p for p in um.parameters() if p.requires_grad], lr=0.01, weight_decay=0.1)
y = torch.randint(0, C, (B,), device=dev, dtype=torch.int32)
w = torch.full((B,), 1.0 / B, device=dev)
for _ in range(30):
    steps.upt_step(um, 100.0, x, y, w, opt)
torch.cuda.synchronize()
