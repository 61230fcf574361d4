"""Developer probe: the 192x256 loader-wave GEMM prototype (variant 9) against the 64-wide kernels (5, 8) and the persistent one (6):
correctness against the f32 product and TFLOP/s at pool-encode and prompt-step row counts."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import grip_amd  # noqa: E402,F401
from grip_amd import native  # noqa: E402

lib = native.lib()
p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)   # noqa: E731
s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def run(M, N, K, epi, variant, reps=10, check=False):
    Mp = (M + 255) // 256 * 256
    if (M + 191) // 192 * 192 > Mp:
        Mp = (M + 191) // 192 * 192
    g = torch.Generator(device="cuda").manual_seed(1)
    A = torch.randn(Mp, K, device="cuda", generator=g).half()
    W = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).half()
    if os.environ.get("ZERO"):      # no data toggling: the clock stays at its maximum (structure of the kernel without the power cap)
        A.zero_()
        W.zero_()
    bias = torch.randn(N, device="cuda", generator=g)
    resid = torch.randn(M, N, device="cuda", generator=g).half()
    out = torch.empty(M, N, device="cuda", dtype=torch.float32 if epi == 0 else torch.float16)
    f = lambda: native.check(lib.grip_debug_gemm(epi, p(A), p(W), M, N, K, p(bias), p(resid), None, p(out), None, 1.0, Mp, variant, s))   # noqa: E731
    for _ in range(3):
        f()
    if check:
        ref = A[:M].float() @ W.float().t()
        if epi in (1, 2, 3):
            ref = ref + bias
        if epi == 2:
            ref = ref * torch.sigmoid(1.702 * ref)
        if epi == 3:
            ref = ref + resid.float()
        err = (out.float() - ref).abs().max().item()
        assert err <= 2e-2 * max(1.0, ref.abs().max().item() / 8), (M, N, K, epi, variant, err)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        f()
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / reps
    return 2.0 * M * N * K / ms / 1e9


for M, N, K, epi in ((400, 256, 128, 0), (1000, 768, 768, 1), (3408, 3072, 768, 2), (3408, 768, 3072, 3), (777, 512, 2048, 3)):
    run(M, N, K, epi, 9, reps=1, check=True)
print("variant 9 correct")
for M in (173360,) if os.environ.get("ZERO") else (173360, 3408):
    for name, N, K, epi in (("qkv", 2304, 768, 1), ("fc", 3072, 768, 2), ("out", 768, 768, 3), ("proj", 768, 3072, 3)):
        print(f"M={M} {name:5s}: " + "  ".join(f"v{v} {run(M, N, K, epi, v):7.0f}" for v in (6, 5, 8, 9)) + " TF/s", flush=True)
