"""Developer probe: TFLOP/s of the GEMM kernels per projection shape (random operands)."""
import ctypes
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import grip_amd  # noqa: E402
from grip_amd import native  # noqa: E402

lib = native.lib()


def p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def run(M, N, K, epi, variant, reps=20):
    Mp = (M + 255) // 256 * 256
    A = torch.randn(Mp, K, device="cuda").half()
    W = (torch.randn(N, K, device="cuda") * K ** -0.5).half()
    if __import__("os").environ.get("ZERO"):
        A.zero_(); W.zero_()
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
    return ms, 2.0 * M * N * K / ms / 1e9


def run_lib(M, N, K, reps=20):
    """Yardstick only (never on the product path): the vendor library GEMM torch dispatches to, same shape, no epilogue."""
    A = torch.randn(M, K, device="cuda").half()
    W = (torch.randn(N, K, device="cuda") * K ** -0.5).half()
    for _ in range(3):
        torch.nn.functional.linear(A, W)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        torch.nn.functional.linear(A, W)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / reps
    return ms, 2.0 * M * N * K / ms / 1e9


arg1 = sys.argv[1] if len(sys.argv) > 1 else "256"
imgs = 0 if arg1.startswith("M=") else int(arg1)
M = imgs * 197
shapes = [("qkv", 2304, 768, 1), ("out", 768, 768, 3), ("fc", 3072, 768, 2), ("proj", 768, 3072, 3), ("f32", 768, 768, 0)]
if arg1.startswith("M=") and not (len(sys.argv) > 2 and sys.argv[2] == "text"):
    M = int(arg1[2:])
if len(sys.argv) > 2 and sys.argv[2] == "text":
    M = imgs * 77
    if arg1.startswith("M="):
        M = int(arg1[2:])
    shapes = [("qkv", 1536, 512, 1), ("out", 512, 512, 3), ("fc", 2048, 512, 2), ("proj", 512, 2048, 3), ("dln", 512, 1536, 0)]
for name, N, K, epi in shapes:
    # variant 3 = 256 x 128 tiles, 4 waves, TWO independent workgroups per CU: the "two 4-wave groups on neighbouring tiles, out of phase" form (one
    # group's epilogue stores under the other's MFMAs) in its natural shape -- the A/B VERDICT r5 #3 asks for against the persistent 256 x 256 kernel (6)
    r = [run(M, N, K, epi, v) for v in (2, 3, 5, 6, 0)]
    print(f"{name:5s} M={M} N={N} K={K}: " + " | ".join(f"{nm} {m:.3f} ms {t:.0f} TF/s" for nm, (m, t) in zip(("ring", "256x128x2wg", "k64", "k64p", "auto"), r))
          + (" | lib(no epilogue) %.3f ms %.0f TF/s" % run_lib(M, N, K) if __import__("os").environ.get("LIB") else ""), flush=True)
