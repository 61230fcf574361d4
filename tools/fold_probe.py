"""Developer probe: repeatability / location of mismatches of the LayerNorm-folded GEMM epilogues."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import grip_amd  # noqa
from grip_amd import native
lib = native.lib()
_p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)
_s = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
M, d, N2 = 66000, 768, 3072
g = torch.Generator(device="cuda").manual_seed(M + d)
Mp = (M + 255) // 256 * 256
x = (torch.randn(Mp, d, device="cuda", generator=g) * 2 + 0.5).half()
W2 = (torch.randn(N2, d, device="cuda", generator=g) * d ** -0.5).half()
b2 = torch.randn(N2, device="cuda", generator=g)
gamma = 1 + 0.3 * torch.randn(d, device="cuda", generator=g)
beta = 0.2 * torch.randn(d, device="cuda", generator=g)
Wg = torch.empty_like(W2); cs = torch.empty(N2, device="cuda"); bb = torch.empty(N2, device="cuda")
native.check(lib.grip_debug_ln_fold(_p(W2), _p(gamma), _p(beta), _p(b2), _p(Wg), _p(cs), _p(bb), N2, d, None, 0, None, M, d, _s()))
xs = x[:M].float()
rowstat = torch.stack([xs.mean(-1), (xs.var(-1, unbiased=False) + 1e-5).rsqrt()], 1).contiguous()
rowstat = torch.cat([rowstat, torch.zeros(Mp - M, 2, device="cuda")]).contiguous()
want = torch.nn.functional.layer_norm(xs, (d,), gamma, beta, 1e-5) @ W2.float().t() + b2
for variant in (3, 2, 6, 1):
    for epi in (7, 8, 8, 7, 8):
        out = torch.full((M, N2), float("nan"), device="cuda", dtype=torch.float16)
        pre = torch.full((M, N2), float("nan"), device="cuda", dtype=torch.float16)
        native.check(lib.grip_debug_gemm_ln(epi, _p(x), _p(Wg), M, N2, d, _p(bb), None, _p(out), _p(pre) if epi == 8 else None, None, _p(rowstat), _p(cs), Mp, variant, _s()))
        got = pre if epi == 8 else out
        bad = ((got.float() - want).abs() > 4e-3 + 4e-3 * want.abs()) | ~torch.isfinite(got.float())
        idx = bad.nonzero()
        print(f"variant {variant} epi {epi}: {int(bad.sum())} bad", (idx[:6].tolist(), idx[-3:].tolist()) if len(idx) else "", flush=True)
        if len(idx):
            r, c = idx[0].tolist()
            print("   got", got[r, c:c + 8].float().tolist(), "\n   want", want[r, c:c + 8].tolist())
