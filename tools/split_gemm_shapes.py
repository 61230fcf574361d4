"""Developer probe: the split-f16 GEMM on the four shapes of a ViT-B/16 block at a refinement chunk (880 images x 197 rows), one epilogue each so that
rocprofv3's per-kernel statistics separate them:  QKV (epi 1, K = 768, N = 2304), c_fc (epi 2, K = 768, N = 3072), out-proj (epi 3, K = 768, N = 768),
c_proj (epi 0, K = 3072, N = 768).  Weights on the f16 grid (two-pass) unless argv[1] == "3".
    rocprofv3 --kernel-trace --stats --output-format csv -d out -- python tools/split_gemm_shapes.py"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import grip_amd  # noqa: F401,E402
from grip_amd import native  # noqa: E402

lib = native.lib()
three = len(sys.argv) > 1 and sys.argv[1] == "3"
M = 880 * 197
Mp = (M + 255) // 256 * 256
p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)
s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for epi, N, K in ((1, 2304, 768), (2, 3072, 768), (3, 768, 768), (0, 768, 3072)):
    A = torch.randn(Mp, K, device="cuda")
    W = torch.randn(N, K, device="cuda") * K ** -0.5
    if not three:
        W = W.half().float()
    bias = torch.randn(N, device="cuda")
    resid = torch.randn(M, N, device="cuda") if epi == 3 else None
    out = torch.empty(M * N, device="cuda")
    a_s, w_s = torch.empty(Mp * K, device="cuda"), torch.empty(N * K, device="cuda")
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for it in range(6):
        native.check(lib.grip_debug_gemm_split(epi, p(A), p(W), M, N, K, p(bias), p(resid), p(out), p(a_s), p(w_s), Mp, s))
    torch.cuda.synchronize()
    print(f"epi {epi} M {M} N {N} K {K}: last launch formed w_lo = {lib.grip_debug_split_last_wlo()}; 2 M N K = {2 * M * N * K / 1e9:.1f} GF")
