"""Developer probe (r05): what does a dependent kernel cost inside a HIP-graph replay on this box, as a function of what the kernel is?
Chains of N dependent launches captured once and replayed: (a) a trivial torch elementwise kernel on 1 element, (b) on 1 M elements,
(c) the library's smallest LayerNorm (425 x 512), (d) its 64 x 128 loader-wave GEMM at K = 512 (425 x 512 x 512)."""
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
N = 200


def replay_us(body):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            body()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(N):
            body()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        t = time.perf_counter()
        for _ in range(10):
            g.replay()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t) / 10)
    return best / N * 1e6


a1 = torch.zeros(1, device=dev)
am = torch.zeros(1 << 20, device=dev)
print(f"torch add_ on 1 element:        {replay_us(lambda: a1.add_(1.0)):.2f} us per dependent launch")
print(f"torch add_ on 1 M elements:     {replay_us(lambda: am.add_(1.0)):.2f} us")
x = torch.randn(512, 512, device=dev)
g_ = torch.ones(512, device=dev)
b_ = torch.zeros(512, device=dev)
o = torch.empty(512, 512, dtype=torch.float16, device=dev)
p = lambda t: ctypes.c_void_p(t.data_ptr())
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
print(f"library LayerNorm 425 x 512:    {replay_us(lambda: native.check(lib.grip_debug_layernorm(p(x), p(g_), p(b_), p(o), 425, 512, st()))):.2f} us")
A = torch.randn(512, 512, device=dev).half()
W = torch.randn(512, 512, device=dev).half()
C = torch.empty(512, 512, dtype=torch.float16, device=dev)
C2 = torch.empty(512, 512, dtype=torch.float16, device=dev)


def gemm_pair():
    native.check(lib.grip_debug_gemm(4, p(A), p(W), 425, 512, 512, None, None, None, p(C), None, ctypes.c_float(1.0), 512, 0, st()))
    native.check(lib.grip_debug_gemm(4, p(C), p(W), 425, 512, 512, None, None, None, p(C2), None, ctypes.c_float(1.0), 512, 0, st()))


print(f"library GEMM 425 x 512 x 512:   {replay_us(gemm_pair) / 2:.2f} us  (the same 0.5-MB weight every launch: L2-hot)")
# ... and with weights nobody has touched recently, as in a prompt step: a rotation of 64 matrices (32 MB: past the L2s, inside the Infinity Cache) and of
# 640 (320 MB: past the Infinity Cache)
for n_w, what in ((64, "32 MB of weights in rotation (L2-cold, Infinity-Cache-hot)"), (640, "320 MB in rotation (HBM-cold)")):
    Ws = torch.randn(n_w, 512, 512, device=dev).half()
    state = {"i": 0}

    def gemm_cold():
        i = state["i"]
        state["i"] = (i + 2) % n_w
        native.check(lib.grip_debug_gemm(4, p(A), p(Ws[i]), 425, 512, 512, None, None, None, p(C), None, ctypes.c_float(1.0), 512, 0, st()))
        native.check(lib.grip_debug_gemm(4, p(C), p(Ws[i + 1]), 425, 512, 512, None, None, None, p(C2), None, ctypes.c_float(1.0), 512, 0, st()))
    print(f"library GEMM 425 x 512 x 512:   {replay_us(gemm_cold) / 2:.2f} us  ({what})")

# Producer -> consumer through a kernel boundary: does the consumer find the producer's output in ITS XCD's L2?  A ping-pong chain of elementwise kernels
# over 425 x 512 f16 (0.4 MB) where block b of every launch reads exactly what block b of the previous launch wrote (same XCD: blocks are dealt to XCDs
# round-robin), against chains where it reads what block b + s wrote (another XCD for s not a multiple of 8).
n = 425 * 512
u = torch.zeros(2 * n, device=dev, dtype=torch.float16)
v = torch.zeros(2 * n, device=dev, dtype=torch.float16)


def chain(shift):
    def body():
        torch.add(u[shift: shift + n], 1.0, out=v[:n])
        torch.add(v[shift: shift + n], 1.0, out=u[:n])
    return replay_us(body) / 2


for shift in (0, 1024, 2048, 4096, 3 * 4096, 8 * 4096, 20 * 1024):
    print(f"elementwise ping-pong 425 x 512 f16, consumer reads {shift:6d} elements further on: {chain(shift):.2f} us per dependent launch")

# Does a kernel cost more when the launches before it were OTHER kernels (instruction fetch: 176 different code paths per prompt step)?
Cf = torch.empty(512, 512, dtype=torch.float32, device=dev)
bias = torch.zeros(512, device=dev)


def mixed():
    native.check(lib.grip_debug_gemm(4, p(A), p(W), 425, 512, 512, None, None, None, p(C), None, ctypes.c_float(1.0), 512, 0, st()))            # f16 out
    native.check(lib.grip_debug_layernorm(p(x), p(g_), p(b_), p(o), 425, 512, st()))
    native.check(lib.grip_debug_gemm(0, p(C), p(W), 425, 512, 512, None, None, None, p(Cf), None, ctypes.c_float(1.0), 512, 0, st()))           # f32 out
    native.check(lib.grip_debug_gemm(1, p(A), p(W), 425, 512, 512, p(bias), None, None, p(C2), None, ctypes.c_float(1.0), 512, 0, st()))        # + bias
    native.check(lib.grip_debug_gemm(3, p(A), p(W), 425, 512, 512, p(bias), p(C), None, p(C2), None, ctypes.c_float(1.0), 512, 0, st()))        # + bias + residual


def same4():
    for _ in range(4):
        native.check(lib.grip_debug_gemm(4, p(A), p(W), 425, 512, 512, None, None, None, p(C), None, ctypes.c_float(1.0), 512, 0, st()))
    native.check(lib.grip_debug_layernorm(p(x), p(g_), p(b_), p(o), 425, 512, st()))


print(f"4 GEMMs of one instantiation + 1 LayerNorm per round: {replay_us(same4):.2f} us per round of five launches")
print(f"4 GEMMs of FOUR instantiations + 1 LayerNorm:        {replay_us(mixed):.2f} us per round of five launches")
