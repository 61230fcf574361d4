"""Developer probe: cost of the LayerNorm-carrying epilogues of the persistent GEMM against the plain ones they replaced, same
shapes (the pool-encode chunk), same kernel variant, back-to-back launches (the chip is equally warm for both)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import grip_amd  # noqa: E402
from grip_amd import native  # noqa: E402

lib = native.lib()
p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)
s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
M = int(sys.argv[1]) * 197 if len(sys.argv) > 1 else 1320 * 197
variant = int(sys.argv[2]) if len(sys.argv) > 2 else 6
Mp = (M + 255) // 256 * 256


def bench(f, reps=12):
    for _ in range(3):
        f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


for name, N, K in (("qkv", 2304, 768), ("fc", 3072, 768), ("out", 768, 768), ("proj", 768, 3072)):
    A = torch.randn(Mp, K, device="cuda").half()
    W = (torch.randn(N, K, device="cuda") * K ** -0.5).half()
    bias, cs = torch.randn(N, device="cuda"), torch.randn(N, device="cuda")
    rowstat = torch.rand(Mp, 2, device="cuda") + 0.5
    resid = torch.randn(M, N, device="cuda").half()
    out = torch.empty(M, N, device="cuda", dtype=torch.float16)
    stat = torch.empty(M, N // 64, 2, device="cuda")
    fl = 2.0 * M * N * K
    res = {}
    if name in ("qkv", "fc"):
        plain, fold = (1, 7) if name == "qkv" else (2, 8)
        res["plain"] = bench(lambda: native.check(lib.grip_debug_gemm(plain, p(A), p(W), M, N, K, p(bias), None, None, p(out), None, 1.0, Mp, variant, s)))
        res["lnfold"] = bench(lambda: native.check(lib.grip_debug_gemm_ln(fold, p(A), p(W), M, N, K, p(bias), None, p(out), None, None, p(rowstat), p(cs), Mp, variant, s)))
    else:
        res["plain"] = bench(lambda: native.check(lib.grip_debug_gemm_ln(3, p(A), p(W), M, N, K, p(bias), p(resid), p(out), None, None, None, None, Mp, variant, s)))
        res["stats"] = bench(lambda: native.check(lib.grip_debug_gemm_ln(3, p(A), p(W), M, N, K, p(bias), p(resid), p(out), None, p(stat), None, None, Mp, variant, s)))
    print(f"{name:5s} M={M} N={N} K={K} variant {variant}: " + " | ".join(f"{k} {v:.3f} ms {fl / v / 1e9:.0f} TF/s" for k, v in res.items()), flush=True)
