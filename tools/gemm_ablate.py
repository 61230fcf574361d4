"""Developer probe: timing ablations of the 256x256 GEMM main loop (results are wrong by construction)."""
import ctypes
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import grip_amd  # noqa: E402
from grip_amd import native  # noqa: E402

lib = native.lib()
p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)
M = 43340
for name, N, K, epi in [("qkv", 2304, 768, 1), ("proj", 768, 3072, 0)]:
    Mp = (M + 255) // 256 * 256
    A = torch.randn(Mp, K, device="cuda").half()
    W = (torch.randn(N, K, device="cuda") * K ** -0.5).half()
    bias = torch.randn(N, device="cuda")
    out = torch.empty(M, N, device="cuda", dtype=torch.float32 if epi == 0 else torch.float16)
    s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for variant in (2, 3):
        for abl, label in [(0, "full"), (1, "no global loads"), (2, "no LDS reads"), (3, "MFMA only")]:
            f = lambda: native.check(lib.grip_debug_gemm(epi, p(A), p(W), M, N, K, p(bias), None, None, p(out), None, 1.0, Mp, variant | (abl << 8), s))
            for _ in range(3):
                f()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(20):
                f()
            b.record()
            torch.cuda.synchronize()
            ms = a.elapsed_time(b) / 20
            print(f"{name} variant {variant} {label:16s}: {ms:.3f} ms {2.0 * M * N * K / ms / 1e9:.0f} TF/s", flush=True)
