"""Developer probe (r05): does a CU-masked side stream really run an image encode beside the graphed CoOp steps?  Times, on one box:
encode alone (current stream, full width | masked stream with the budget), steps alone (graph replays), and both at once."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import grip_amd  # noqa: E402,F401
from grip_amd import clip, engine, rng, steps  # noqa: E402
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
x = torch.randn(816, 3, 224, 224, device=dev)
for _ in range(5):
    g(f, y, w)
with torch.no_grad():
    m.encode_image(x)
torch.cuda.synchronize()


def timed(fn, n=3):
    best = 1e9
    for _ in range(n):
        torch.cuda.synchronize()
        t = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t)
    return best * 1e3


def enc_main():
    with torch.no_grad():
        m.encode_image(x)


def steps_only(k=51):
    for _ in range(k):
        g(f, y, w)


print(f"encode 816 images, current stream, full width: {timed(enc_main):.1f} ms")
print(f"51 graphed steps alone: {timed(steps_only):.1f} ms")
for quarters in (3, 2):
    ms = engine.masked_stream(dev, quarters)
    if ms is None:
        print("masked stream refused")
        continue
    side, n_cus = ms

    def enc_side(budget=True):
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            engine.set_cu_budget(n_cus if budget else 0)
            try:
                m.encode_image(x)
            finally:
                engine.set_cu_budget(0)

    def both():
        enc_side()
        steps_only()
        torch.cuda.current_stream().wait_stream(side)

    def both_plain():       # an ordinary side stream, full-width grids
        s2 = both_plain.s
        s2.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s2), torch.no_grad():
            m.encode_image(x)
        steps_only()
        torch.cuda.current_stream().wait_stream(s2)
    both_plain.s = torch.cuda.Stream(device=dev)
    print(f"{quarters}/4 of the chip ({n_cus} CUs): encode alone on the masked stream {timed(enc_side):.1f} ms (grids at full width: {timed(lambda: enc_side(False)):.1f} ms); "
          f"encode + 51 steps at once {timed(both):.1f} ms; with an ordinary side stream {timed(both_plain):.1f} ms")
