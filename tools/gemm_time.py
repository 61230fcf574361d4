"""Developer probe: time one GEMM shape / variant (HIP events around `reps` launches)."""
import ctypes
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import grip_amd  # noqa: E402,F401
from grip_amd import native  # noqa: E402

lib = native.lib()
M, N, K, epi, variant, reps = [int(a) for a in sys.argv[1:7]]
Mp = (M + 255) // 256 * 256
p = lambda t: ctypes.c_void_p(t.data_ptr())
A = torch.randn(Mp, K, device="cuda").half()
W = (torch.randn(N, K, device="cuda") * K ** -0.5).half()
bias = torch.randn(N, device="cuda")
resid = torch.randn(M, N, device="cuda").half()
out = torch.empty(M, N, device="cuda", dtype=torch.float32 if epi == 0 else torch.float16)
s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
f = lambda: native.check(lib.grip_debug_gemm(epi, p(A), p(W), M, N, K, p(bias), p(resid), None, p(out), None, 1.0, Mp, variant, s))
for _ in range(3):
    f()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(reps):
    f()
b.record()
torch.cuda.synchronize()
ms = a.elapsed_time(b) / reps
print(f"M={M} N={N} K={K} epi={epi} variant={variant}: {ms:.3f} ms {2.0 * M * N * K / ms / 1e9:.0f} TF/s")
