import sys, ctypes, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import test_gpu_split as T
native, lib = T._lib()
M, N, K = 256, 256, 64
g = torch.Generator(device="cuda").manual_seed(M * 13 + N + K)
A = torch.randn(M, K, device="cuda", generator=g)
W = torch.randn(N, K, device="cuda", generator=g) * K ** -0.5
bias = torch.randn(N, device="cuda", generator=g)
a_s = torch.empty(M * K, device="cuda"); w_s = torch.empty(N * K, device="cuda")
ref = A.double() @ W.double().t() + bias.double()
h = torch.zeros(M * N, device="cuda")
native.check(lib.grip_debug_gemm_split(2, T._p(A), T._p(W), M, N, K, T._p(bias), None, T._p(h), T._p(a_s), T._p(w_s), M, T._stream()))
got = T._unsplit(h, M, N); want = T.quick_gelu(ref)
err = (got - want).abs()
idx = torch.nonzero(err > 1e-5)
print("bad elements:", idx.shape[0])
v = h.view(torch.float16).reshape(M, N // 32, 2, 32)
for r, c in idx[:12].tolist():
    print(r, c, "pre", ref[r, c].item(), "want", want[r, c].item(), "got", got[r, c].item(), "hi", v[r, c // 32, 0, c % 32].item(), "lo", v[r, c // 32, 1, c % 32].item())
