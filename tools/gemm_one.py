"""Developer probe: run one GEMM shape/variant repeatedly (for rocprofv3 --pmc)."""
import ctypes
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import grip_amd  # noqa: E402
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
for _ in range(reps):
    native.check(lib.grip_debug_gemm(epi, p(A), p(W), M, N, K, p(bias), p(resid), None, p(out), None, 1.0, Mp, variant, s))
torch.cuda.synchronize()
