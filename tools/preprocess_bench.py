"""Developer probe: images/s of the input pipeline -- JPEG decode on the host + CLIP preprocessing -- three ways:
(a) the reference's way: PIL decode + PIL bicubic resize + crop + normalise on one host thread (data/dataset.py:56-89 runs the
transform per item; 3x per item in training), (b) native per-image kernels (decode on one thread), (c) thread-pool decode +
ONE batched launch per chunk (ClipPreprocess.load_batch).  The encoder's pool rate is printed next to it."""
import os
import sys
import tempfile
import time

import numpy as np
import torch
from PIL import Image

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import grip_amd  # noqa: E402,F401
from grip_amd.preprocess import ClipPreprocess  # noqa: E402
from oracle.preprocess import clip_transform  # noqa: E402

n, workers = int(sys.argv[1]) if len(sys.argv) > 1 else 2048, int(sys.argv[2]) if len(sys.argv) > 2 else 32
d = tempfile.mkdtemp()
g = np.random.RandomState(0)
paths = []
for i in range(n):      # ImageNet-like sizes, natural-image-like low-frequency content so that JPEG sizes are realistic
    h, w = int(g.choice([375, 333, 500, 480])), int(g.choice([500, 400, 640]))
    base = g.randint(0, 256, size=(h // 16 + 1, w // 16 + 1, 3)).astype(np.uint8)
    im = Image.fromarray(base).resize((w, h), Image.BICUBIC)
    p = os.path.join(d, f"{i:05d}.jpg")
    im.save(p, quality=90)
    paths.append(p)
pre = ClipPreprocess(224, "cuda")
pre.load_batch(paths[:8])
torch.cuda.synchronize()
t = time.perf_counter()
for p in paths[:128]:
    clip_transform(Image.open(p).convert("RGB"), 224)
t_ref = (time.perf_counter() - t) / 128
t = time.perf_counter()
for p in paths:
    pre(Image.open(p))
torch.cuda.synchronize()
t_one = (time.perf_counter() - t) / n
from concurrent.futures import ThreadPoolExecutor  # noqa: E402

CHUNK = 512


def pipelined(**kw):
    """images/s of decode_chunk (chunk i+1, background thread) overlapped with finish_chunk (chunk i), as the lazy file pool runs it"""
    bg = ThreadPoolExecutor(max_workers=1)
    spans = [(lo, min(lo + CHUNK, n)) for lo in range(0, n, CHUNK)]
    for rep in range(2):        # first round warms the staging buffers / worker processes
        t = time.perf_counter()
        fut = bg.submit(pre.decode_chunk, paths[spans[0][0]:spans[0][1]], **kw)
        for i in range(len(spans)):
            h = fut.result()
            if i + 1 < len(spans):
                fut = bg.submit(pre.decode_chunk, paths[spans[i + 1][0]:spans[i + 1][1]], **kw)
            pre.finish_chunk(h)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
    return n / dt


res = {}
for w in (1, 8, workers):
    t = time.perf_counter()
    for lo in range(0, n, 256):
        pre.load_batch(paths[lo:lo + 256], workers=w)
    torch.cuda.synchronize()
    res[w] = (time.perf_counter() - t) / n
line = (f"host PIL transform, 1 thread: {1 / t_ref:8.0f} img/s | native per-image kernels, 1 decode thread: {1 / t_one:8.0f} img/s | "
        + " | ".join(f"batched launch, {w} decode threads: {1 / v:8.0f} img/s" for w, v in res.items()) + f"   (host cpu_count {os.cpu_count()})")
print(line)
for w in (8, 32):
    print(f"pipelined (decode one chunk of {CHUNK} ahead), {w:3d} decode threads:   {pipelined(workers=w):8.0f} img/s")
from grip_amd.data.decode import usable_cpus  # noqa: E402
print(f'usable cpus (affinity, cgroup quota): {usable_cpus()}')
for pr in (8, 16, 24, 32):
    if pr <= 2 * usable_cpus():
        print(f"pipelined (decode one chunk of {CHUNK} ahead), {pr:3d} decode processes: {pipelined(processes=pr):8.0f} img/s")
pre.close()
