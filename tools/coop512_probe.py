"""Developer probe (r05): would the cooperative split-K of the text tower's c_proj also pay on its K = 512, N = 512 residual GEMM (out-proj: 28 workgroups that each
stage 196 KB through one CU's LDS-DMA path)?  Chains of dependent launches replayed from a HIP graph, hot operands: plain against ksplit 2 / 4."""
import ctypes
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import grip_amd  # noqa: E402,F401
from grip_amd import native  # noqa: E402

dev = torch.device("cuda", 0)
lib = native.lib()
p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
M, N = 425, 512
for K in (512, 2048):
    A = (torch.randn(512, K, device=dev) * 0.1).half()
    W = (torch.randn(N, K, device=dev) * 0.1).half()
    bias = torch.zeros(N, device=dev)
    resid = torch.randn(512, N, device=dev).half()
    out = torch.empty(512, N, dtype=torch.float16, device=dev)
    scratch = torch.zeros(64 * 8 * 4 * 2048, device=dev)
    cnt = torch.zeros(64 * 4, dtype=torch.int32, device=dev)
    ref = None
    for ks in (1, 2, 4, 8):
        if K // 64 // ks < 2:
            continue

        def body():
            native.check(lib.grip_debug_gemm_train(3, p(A), p(W), M, N, K, p(bias), p(resid), p(out), None, None, None, None, None, 0, ks,
                                                   p(scratch) if ks > 1 else None, p(cnt) if ks > 1 else None, 512, st()))
        try:
            body()
            torch.cuda.synchronize()
        except Exception as e:
            print(f"K={K} ksplit {ks}: refused ({str(e)[:80]})")
            continue
        if ref is None:
            ref = out.clone()
        err = float((out.float() - ref.float()).abs().max())
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            body()
        torch.cuda.synchronize()
        with torch.cuda.graph(g):
            for _ in range(200):
                body()
        g.replay()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            t = time.perf_counter()
            for _ in range(10):
                g.replay()
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t) / 10)
        print(f"residual GEMM {M} x {N} x {K}, ksplit {ks}: {best / 200 * 1e6:.2f} us per dependent launch (max |diff| vs plain {err:.3g})", flush=True)
