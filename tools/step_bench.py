"""Developer probe: prompt-step throughput (forward + dgrad backward + SGD) for the three prompt families at the
reference's batch size 16, ViT-B/16: CoOp (C=102, P=16 text), VPT (P=16 visual, C=45), UPT (Pt=Pv=4, C=47)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import grip_amd  # noqa: E402
from grip_amd import clip, rng, steps  # noqa: E402
from grip_amd.models import CustomImageEncoder, CustomTextEncoder, ImagePrefixModel, TextPrefixModel, UPTModel  # noqa: E402

dev = "cuda"
ONLY = sys.argv[1].lower() if len(sys.argv) > 1 else ""      # "coop" | "vpt" | "upt": run just that family
m, _ = clip.load("ViT-B/16", device=dev)
B = 16
x = torch.randn(B, 3, 224, 224, device=dev)
scale = m.logit_scale.exp().item()
w = torch.full((B,), 1.0 / B, device=dev)


def timeit(name, fn, flops, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / n
    print(f"{name:5s}: {dt * 1e3:7.2f} ms/step  {B / dt:7.0f} img/s  {flops / dt / 1e12:6.1f} TFLOP/s (algorithmic)", flush=True)


def N(name, shape, std=0.02):
    return torch.from_numpy(rng.normal(1, rng.stream_id(name), shape, 0.0, std)).to(dev)


def coop():
    C = 102
    classes = [f"class {i}" for i in range(C)]
    tm = TextPrefixModel(N("c", (1, 16, 512)), CustomTextEncoder(m, dev, torch.float32), classes, device=dev)
    opt = torch.optim.SGD([tm.prefix], lr=0.1, weight_decay=0.1)
    y = torch.randint(0, C, (B,), device=dev, dtype=torch.int32)
    timeit("CoOp", lambda: steps.coop_step(tm, m, x, y, w, opt), B * 35.13e9 + 2 * C * 5.96e9)


def vpt():
    C = 45
    txt = m.encode_text(clip.tokenize([f"a photo of a class {i}" for i in range(C)]).to(dev))
    im = ImagePrefixModel(N("v", (16, 768)), CustomImageEncoder(m.visual), device=dev)
    opt = torch.optim.SGD([im.prefix], lr=0.1, weight_decay=0.1)
    y = torch.randint(0, C, (B,), device=dev, dtype=torch.int32)
    timeit("VPT", lambda: steps.vpt_step(im, txt, scale, x, y, w, opt), 2 * B * 38.09e9)


def upt():
    C = 47
    classes = [f"class {i}" for i in range(C)]
    um = UPTModel(N("uc", (1, 4, 512)), N("uv", (1, 4, 768)), None, CustomImageEncoder(m.visual), CustomTextEncoder(m, dev, torch.float32), classes, 128,
                  device=dev, dtype=torch.float32)
    opt = torch.optim.SGD([p for p in um.parameters() if p.requires_grad], lr=0.01, weight_decay=0.1)
    y = torch.randint(0, C, (B,), device=dev, dtype=torch.int32)
    timeit("UPT", lambda: steps.upt_step(um, scale, x, y, w, opt), 2 * B * 35.87e9 + 2 * C * 5.96e9)


for name, fn in (("coop", coop), ("vpt", vpt), ("upt", upt)):
    if ONLY in ("", name):
        fn()
