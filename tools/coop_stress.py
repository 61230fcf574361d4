"""Developer probe: stress of the cooperative split-K GEMM (csrc/gemm.hip, gemm_ringw_kernel): thousands of launches, alone and next to a second stream that keeps
the chip busy with large GEMMs (other workgroup placements, other timing), every result compared BIT FOR BIT with the first one.  A stale partial tile or a lost
ticket shows up as a differing element.  `python tools/coop_stress.py [iterations]`."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import grip_amd  # noqa: E402,F401
from grip_amd import native  # noqa: E402

lib = native.lib()
p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)
n_it = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
bad = 0
for (M, d, K, ks) in [(425, 512, 2048, 4), (425, 512, 2048, 8), (130, 768, 3072, 4), (1000, 256, 1024, 2)]:
    g = torch.Generator(device="cuda").manual_seed(M + K)
    Mp = (M + 255) // 256 * 256
    A = torch.randn(Mp, K, device="cuda", generator=g).half()
    W = (torch.randn(d, K, device="cuda", generator=g) * K ** -0.5).half()
    b = torch.randn(d, device="cuda", generator=g)
    resid = (torch.randn(M, d, device="cuda", generator=g) * 2).half()
    tiles = (M + 63) // 64 * (d // 128)
    scratch = torch.zeros(tiles * ks * 8192, device="cuda")
    cnt = torch.zeros(tiles * 4, dtype=torch.int32, device="cuda")
    parts = d // 64
    side = torch.cuda.Stream()
    big_a = torch.randn(8192, 4096, device="cuda").half()
    big_b = torch.randn(4096, 4096, device="cuda").half()
    ref = ref_stat = None
    for it in range(n_it):
        busy = it >= n_it // 3                      # the last two thirds run next to a busy second stream
        if busy and it % 8 == 0:
            with torch.cuda.stream(side):
                torch.mm(big_a, big_b)
        x = torch.full((Mp, d), float("nan"), device="cuda", dtype=torch.float16)
        stat = torch.full((parts, M, 2), float("nan"), device="cuda")
        s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        native.check(lib.grip_debug_gemm_train(3, p(A), p(W), M, d, K, p(b), p(resid), p(x), None, p(stat), None, None, None, 0, ks, p(scratch), p(cnt), Mp, s))
        if ref is None:
            ref, ref_stat = x[:M].clone(), stat.clone()
            want = A[:M].float() @ W.float().t() + b + resid.float()
            assert (ref.float() - want).abs().max().item() < 0.05
        elif it % 4 == 0 or it > n_it - 50:
            if not (torch.equal(x[:M], ref) and torch.equal(stat, ref_stat)):
                bad += 1
                print(f"MISMATCH M={M} ks={ks} iteration {it}: {int((x[:M] != ref).sum())} elements differ", flush=True)
    torch.cuda.synchronize()
    assert int(cnt.abs().sum()) == 0
    print(f"M={M} d={d} K={K} ks={ks}: {n_it} launches, mismatches so far {bad}", flush=True)
print("coop stress:", "OK" if bad == 0 else f"{bad} MISMATCHES")
sys.exit(1 if bad else 0)
