"""Developer probe, HOST ONLY (no GPU, no torch): images/s of the decode back ends of data/decode.py -- one thread, a thread pool,
worker processes around a shared segment -- on ImageNet-sized synthetic JPEGs.  Usage: python tools/decode_bench.py [n_files] [chunk]"""
import os
import sys
import tempfile
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
from PIL import Image

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import grip_amd  # noqa: E402,F401
from grip_amd.data import decode as D  # noqa: E402

n, chunk = int(sys.argv[1]) if len(sys.argv) > 1 else 1024, int(sys.argv[2]) if len(sys.argv) > 2 else 256
d = tempfile.mkdtemp(prefix="grip_jpeg_")


def make(i):
    g = np.random.RandomState(i)
    h, w = int(g.choice([375, 333, 500, 480])), int(g.choice([500, 400, 640]))
    base = g.randint(0, 256, size=(h // 16 + 1, w // 16 + 1, 3)).astype(np.uint8)
    p = os.path.join(d, f"{i:05d}.jpg")
    Image.fromarray(base).resize((w, h), Image.BICUBIC).save(p, quality=90)
    return p


with ThreadPoolExecutor(max_workers=8) as ex:
    paths = list(ex.map(make, range(n)))
buf = np.zeros(chunk * 1024 * 1024, dtype=np.uint8)
cpus = D.usable_cpus()
print(f"{n} files, chunk {chunk}, usable cpus {cpus}")


def walk(fn):
    best = 0.0
    for _ in range(2):
        t = time.perf_counter()
        for lo in range(0, n, chunk):
            packed, overflow = fn(paths[lo:lo + chunk])
            assert not overflow
        best = max(best, n / (time.perf_counter() - t))
    return best


print(f"one thread:            {walk(lambda p: D.decode_threads(p, buf, None)):8.0f} img/s")
for w in sorted({2, 4, cpus, 2 * cpus}):
    pool = D.make_thread_pool(w)
    print(f"{w:3d} decode threads:    {walk(lambda p: D.decode_threads(p, buf, pool)):8.0f} img/s")
for pr in sorted({2, 4, cpus}):
    dec = D.ProcessDecoder(pr, chunk * 1024 * 1024, slots=1)
    try:
        print(f"{pr:3d} decode processes:  {walk(lambda p: dec.decode(p, 0)):8.0f} img/s")
    finally:
        dec.close()
