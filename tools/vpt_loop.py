"""Developer probe: VPT steps only (for rocprofv3 --stats)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import grip_amd  # noqa: E402
from grip_amd import clip, steps  # noqa: E402
from grip_amd.models import CustomImageEncoder, ImagePrefixModel  # noqa: E402

m, _ = clip.load("ViT-B/16", device="cuda")
B, C = 16, 45
x = torch.randn(B, 3, 224, 224, device="cuda")
txt = m.encode_text(clip.tokenize([f"a photo of a class {i}" for i in range(C)]).cuda())
im = ImagePrefixModel(torch.randn(16, 768, device="cuda") * 0.02, CustomImageEncoder(m.visual), device="cuda")
opt = torch.optim.SGD([im.prefix], lr=0.1)
y = torch.randint(0, C, (B,), device="cuda", dtype=torch.int32)
w = torch.full((B,), 1.0 / B, device="cuda")
for _ in range(30):
    steps.vpt_step(im, txt, 100.0, x, y, w, opt)
torch.cuda.synchronize()
