"""Developer probe: the small-M GEMMs of the prompt steps (text tower M = 21 classes x 102... = 2 142 rows at d = 512; image
tower M = 16 x 213 = 3 408 rows at d = 768), one shape per line, back-to-back launches.  GRIP_GEMM_RING = 0 / 3 / 4 / unset
selects the two-stage kernel, a ring depth, or the launcher's own choice; variant argument 0 lets the launcher pick the tile.

    GRIP_GEMM_RING=0 python tools/small_gemm_bench.py [variant]
"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import grip_amd  # noqa: E402,F401
from grip_amd import native  # noqa: E402

lib = native.lib()
p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)
s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
ROT = 256 if os.environ.get("ROT", "0") == "1" else 0      # bit 8 of the debug hook's variant: the train-mode launches' row-staggered K walks
variant = (int(sys.argv[1]) if len(sys.argv) > 1 else 0) | ROT
ONLY = os.environ.get("ONLY", "")
SWEEP = os.environ.get("SWEEP", "0") == "1"      # also time every tile shape / split factor per line


def bench(f, reps=40):
    for _ in range(5):
        f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


SHAPES = []
IMAGE_M = int(os.environ.get("IMAGE_M", "3408"))      # 16 x (197 + 16) rows; 3152 = 16 x 197
for tag, M, d in (("text", 2142, 512), ("image", IMAGE_M, 768), ("shared", 425, 512)):
    SHAPES += [(f"{tag} qkv fwd", 1, M, 3 * d, d), (f"{tag} out fwd", 3, M, d, d), (f"{tag} fc fwd", 2, M, 4 * d, d), (f"{tag} proj fwd", 3, M, d, 4 * d),
               (f"{tag} proj dgrad", 5, M, 4 * d, d), (f"{tag} fc dgrad", 0, M, d, 4 * d), (f"{tag} out dgrad", 4, M, d, d), (f"{tag} qkv dgrad", 0, M, d, 3 * d)]

print(f"GRIP_GEMM_RING={os.environ.get('GRIP_GEMM_RING', 'auto')} GRIP_GEMM_KSPLIT={os.environ.get('GRIP_GEMM_KSPLIT', 'auto')} variant={variant}")
total = {"text": 0.0, "image": 0.0, "shared": 0.0}
COLD = int(os.environ.get("COLD", "1"))      # number of operand sets cycled through (>= 12 sets of ~50 MB defeat the 256 MiB Infinity Cache: the in-situ case)
for name, epi, M, N, K in SHAPES:
    if ONLY and ONLY not in name:
        continue
    Mp = (M + 255) // 256 * 256
    sets = []
    for _ in range(COLD):
        sets.append(dict(A=torch.randn(Mp, K, device="cuda").half(), W=(torch.randn(N, K, device="cuda") * K ** -0.5).half(), resid=torch.randn(M, N, device="cuda").half(),
                         aux=torch.randn(M, N, device="cuda").half(), out=torch.empty(8 if epi == 0 else 1, Mp, N, device="cuda", dtype=torch.float32 if epi == 0 else torch.float16)))
    bias = torch.randn(N, device="cuda")
    it = [0]

    def nxt():
        it[0] = (it[0] + 1) % COLD
        return sets[it[0]]
    if epi == 0:
        used = ctypes.c_int(0)

        def run(ks=0, v=variant):
            d = nxt()
            native.check(lib.grip_debug_gemm_splitk(p(d["A"]), p(d["W"]), M, N, K, p(d["out"]), ks, Mp * N, ctypes.addressof(used), Mp, v, s))
        t = bench(run)
        extra = f"ksplit {used.value}"
        if SWEEP:
            for v in (1, 4):
                for ks in (1, 2, 3, 4, 6, 8):
                    if (K // 64) % ks == 0:
                        tt = bench(lambda: run(ks, v | ROT))
                        extra += f" | v{v} ks{ks} {tt:.1f}"
    else:
        def run(v=variant):
            d = nxt()
            native.check(lib.grip_debug_gemm(epi, p(d["A"]), p(d["W"]), M, N, K, p(bias), p(d["resid"]), p(d["aux"]), p(d["out"]), None, 1.0, Mp, v, s))
        t = bench(run)
        extra = ""
        if SWEEP:
            for v in (1, 4, 3, 5, 8):
                if v in (5, 8) and N % 256:
                    continue
                tt = bench(lambda: run(v | ROT))
                extra += f" | v{v} {tt:.1f}"
    total[name.split()[0]] += t
    print(f"{name:18s} epi {epi} M={M} N={N} K={K}: {t:7.2f} us {2.0 * M * N * K / t / 1e6:6.0f} TF/s {extra}", flush=True)
print("sum per block: " + ", ".join(f"{k} {v:.1f} us" for k, v in total.items()))
