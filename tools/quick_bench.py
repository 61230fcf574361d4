"""Developer probe (not the judged bench): times the ViT-B/16 encode at a few chunk sizes."""
import sys
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import grip_amd  # noqa: E402
from grip_amd import clip  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "ViT-B/16"
m, _ = clip.load(name, device="cuda")
for B in [int(a) for a in (sys.argv[2:] or ["16", "64", "256", "512"])]:
    x = torch.randn(B, 3, m.visual.input_resolution, m.visual.input_resolution, device="cuda")
    for _ in range(2):
        m.encode_image(x)
    torch.cuda.synchronize()
    t = time.time()
    n = 5
    for _ in range(n):
        m.encode_image(x)
    torch.cuda.synchronize()
    dt = (time.time() - t) / n
    gf = {"ViT-B/16": 35.13, "ViT-L/14@336px": 381.9, "ViT-L/14": 162.0, "ViT-B/32": 8.8}.get(name, 0)
    print(f"{name} B={B}: {dt * 1e3:.2f} ms  {B / dt:.0f} img/s  {B * gf / dt / 1e3:.1f} TFLOP/s", flush=True)
