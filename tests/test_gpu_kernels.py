"""Kernel-level numerics: each HIP kernel against a plain PyTorch fp32 reference of the same op
(inputs rounded to f16 where the kernel consumes f16, so the comparison isolates the kernel)."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _lib():
    import grip_amd  # noqa: F401
    from grip_amd import native
    return native, native.lib()


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def quick_gelu(x):
    return x * torch.sigmoid(1.702 * x)


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (200, 384, 128), (3408, 2304, 768), (77 * 5, 512, 2048), (16, 512, 768),
                                   (50432, 768, 768), (12700, 2304, 768), (25000, 768, 3072), (16500, 3072, 768), (50000, 512, 128)])
@pytest.mark.parametrize("variant", [0, 1, 2, 3, 4, 5, 6, 8])
def test_gemm_epilogues(M, N, K, variant):
    if variant in (2, 5, 6, 8) and N % 256:
        pytest.skip("256x256 tile needs N % 256 == 0")
    if variant in (2, 3, 5, 6, 8) and K < 128:
        pytest.skip("ring needs K >= 128")
    if variant == 8 and (M + 191) // 192 * 192 > (M + 255) // 256 * 256:
        pytest.skip("192-row tiles need A padded to a multiple of 192 rows")
    native, lib = _lib()
    g = torch.Generator(device="cuda").manual_seed(M * 7 + N)
    Mp = (M + 255) // 256 * 256     # rows allocated for A: lets the launcher pick the 256-row tile for large problems
    A = torch.randn(Mp, K, device="cuda", generator=g).half()
    A[M:] = float("nan")  # padding rows must never leak into stored rows
    W = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).half()
    bias = torch.randn(N, device="cuda", generator=g)
    resid = torch.randn(M, N, device="cuda", generator=g)
    aux = torch.randn(M, N, device="cuda", generator=g).half()
    ref = A[:M].float() @ W.float().t()
    tol = dict(rtol=2e-3, atol=2e-3)

    out = torch.full((M, N), 7.0, device="cuda")
    native.check(lib.grip_debug_gemm(0, _p(A), _p(W), M, N, K, None, None, None, _p(out), None, 1.0, Mp, variant, _stream()))
    torch.testing.assert_close(out, ref, rtol=1e-4, atol=1e-4)   # EPI_F32: only f32 summation order differs

    out16 = torch.zeros(M, N, device="cuda", dtype=torch.float16)
    native.check(lib.grip_debug_gemm(1, _p(A), _p(W), M, N, K, _p(bias), None, None, _p(out16), None, 1.0, Mp, variant, _stream()))
    torch.testing.assert_close(out16.float(), ref + bias, **tol)

    pre = torch.zeros(M, N, device="cuda", dtype=torch.float16)
    native.check(lib.grip_debug_gemm(2, _p(A), _p(W), M, N, K, _p(bias), None, None, _p(out16), _p(pre), 1.0, Mp, variant, _stream()))
    torch.testing.assert_close(out16.float(), quick_gelu(ref + bias), **tol)
    torch.testing.assert_close(pre.float(), ref + bias, **tol)

    # residual epilogue: the residual stream is f16 (added in f32, rounded once)
    resid16 = resid.half()
    native.check(lib.grip_debug_gemm(3, _p(A), _p(W), M, N, K, _p(bias), _p(resid16), None, _p(out16), None, 1.0, Mp, variant, _stream()))
    torch.testing.assert_close(out16.float(), ref + bias + resid16.float(), **tol)
    # in place (out aliases resid), as the inference path uses it
    r2 = resid16.clone()
    native.check(lib.grip_debug_gemm(3, _p(A), _p(W), M, N, K, _p(bias), _p(r2), None, _p(r2), None, 1.0, Mp, variant, _stream()))
    torch.testing.assert_close(r2.float(), ref + bias + resid16.float(), **tol)

    native.check(lib.grip_debug_gemm(5, _p(A), _p(W), M, N, K, None, None, _p(aux), _p(out16), None, 1.0, Mp, variant, _stream()))
    x = aux.float()
    s = torch.sigmoid(1.702 * x)
    torch.testing.assert_close(out16.float(), ref * (s * (1 + 1.702 * x * (1 - s))), **tol)


@pytest.mark.parametrize("variant", [2, 3, 5, 6])
def test_gemm256_is_not_transposed(variant):
    """Same transpose check through the 256x256 tile path (large M)."""
    native, lib = _lib()
    M, N, K = 256 * 64, 768, 128
    A = torch.zeros(M, K, device="cuda", dtype=torch.float16)
    idx = torch.arange(M, device="cuda")
    A[idx, idx % K] = 1.0
    W = ((torch.arange(N * K, device="cuda").reshape(N, K) * 7) % 97).half()
    out = torch.zeros(M, N, device="cuda")
    native.check(lib.grip_debug_gemm(0, _p(A), _p(W), M, N, K, None, None, None, _p(out), None, 1.0, M, variant, _stream()))
    torch.testing.assert_close(out, W.float().t()[idx % K])


@pytest.mark.parametrize("M,N,K", [(3408, 768, 3072), (3408, 3072, 768), (2142, 512, 2048), (425, 512, 512), (3152, 2304, 768)])
@pytest.mark.parametrize("variant", [0, 1, 4, 5, 8])
def test_gemm_row_rotation_of_the_train_mode_launches(M, N, K, variant):
    """Train-mode GEMMs let tile row tm start its K walk 2 * tm (split-K: tm) slices further on (csrc/gemm.hip launch_gemm_impl, bit 8 of the
    debug hook's variant): another summation order, the same product -- against the float64 product and against the unrotated launch."""
    if variant in (5, 8) and N % 256:
        pytest.skip("256-column tile")
    native, lib = _lib()
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    Mp = (M + 255) // 256 * 256
    if variant == 8 and (M + 191) // 192 * 192 > Mp:
        pytest.skip("192-row tiles need A padded to a multiple of 192 rows")
    A = torch.randn(Mp, K, device="cuda", generator=g).half()
    A[M:] = float("nan")
    W = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).half()
    bias = torch.randn(N, device="cuda", generator=g)
    resid = torch.randn(M, N, device="cuda", generator=g).half()
    ref = (A[:M].double() @ W.double().t() + bias.double() + resid.double())
    outs = []
    for rot in (0, 256):
        out = torch.zeros(M, N, device="cuda", dtype=torch.float16)
        native.check(lib.grip_debug_gemm(3, _p(A), _p(W), M, N, K, _p(bias), _p(resid), None, _p(out), None, 1.0, Mp, variant | rot, _stream()))
        torch.testing.assert_close(out.double(), ref, rtol=2e-3, atol=4e-3)
        outs.append(out)
    assert (outs[0].float() - outs[1].float()).abs().max().item() <= 4e-3
    if variant in (0, 1, 4):      # split-K partial products (EPI_F32): the partials' sum is what ln_bwd_add consumes
        used = ctypes.c_int(0)
        sums = []
        for rot in (0, 256):
            part = torch.zeros(8, Mp, N, device="cuda")
            native.check(lib.grip_debug_gemm_splitk(_p(A), _p(W), M, N, K, _p(part), 0, Mp * N, ctypes.addressof(used), Mp, variant | rot, _stream()))
            sums.append(part[:max(used.value, 1), :M].sum(0))
            torch.testing.assert_close(sums[-1].double(), A[:M].double() @ W.double().t(), rtol=1e-3, atol=2e-3)


def test_gemm_is_not_transposed():
    """A = I, asymmetric W: catches an output / operand transpose."""
    native, lib = _lib()
    A = torch.eye(128, device="cuda").half()
    W = (torch.arange(128 * 128, device="cuda").reshape(128, 128) % 97).half()
    out = torch.zeros(128, 128, device="cuda")
    native.check(lib.grip_debug_gemm(0, _p(A), _p(W), 128, 128, 128, None, None, None, _p(out), None, 1.0, 128, 0, _stream()))
    torch.testing.assert_close(out, W.float().t())


@pytest.mark.parametrize("B,S,H,causal", [(2, 17, 2, 0), (3, 33, 2, 0), (2, 77, 8, 1), (2, 197, 12, 0), (2, 213, 12, 0), (1, 257, 4, 0), (1, 593, 2, 0),
                                          # >= 1024 (image, head) items: the persistent double-buffered encode kernel, every chunk count it serves
                                          (100, 197, 12, 0), (300, 100, 4, 0), (90, 130, 12, 0), (70, 180, 16, 0), (70, 213, 16, 0), (70, 250, 16, 0),
                                          (64, 273, 16, 0), (257, 288, 4, 0)])
def test_attention(B, S, H, causal):
    native, lib = _lib()
    g = torch.Generator(device="cuda").manual_seed(S)
    D = H * 64
    qkv = (torch.randn(B * S, 3 * D, device="cuda", generator=g) * 1.5).half()
    out = torch.zeros(B * S, D, device="cuda", dtype=torch.float16)
    native.check(lib.grip_debug_attention(_p(qkv), _p(out), B, S, H, causal, _stream()))
    q, k, v = qkv.float().reshape(B, S, 3, H, 64).permute(2, 0, 3, 1, 4)
    s = (q * 0.125) @ k.transpose(-1, -2)
    if causal:
        s = s + torch.full((S, S), float("-inf"), device="cuda").triu(1)
    ref = (s.softmax(-1) @ v).permute(0, 2, 1, 3).reshape(B * S, D)
    torch.testing.assert_close(out.float(), ref, rtol=3e-3, atol=3e-3)


@pytest.mark.parametrize("M,d", [(5, 128), (300, 512), (1000, 768), (64, 1024)])
def test_layernorm(M, d):
    native, lib = _lib()
    g = torch.Generator(device="cuda").manual_seed(d)
    x = torch.randn(M, d, device="cuda", generator=g) * 3 + 1
    gamma = torch.randn(d, device="cuda", generator=g)
    beta = torch.randn(d, device="cuda", generator=g)
    out = torch.zeros(M, d, device="cuda", dtype=torch.float16)
    native.check(lib.grip_debug_layernorm(_p(x), _p(gamma), _p(beta), _p(out), M, d, _stream()))
    ref = torch.nn.functional.layer_norm(x, (d,), gamma, beta, 1e-5)
    torch.testing.assert_close(out.float(), ref, rtol=2e-3, atol=2e-3)


def test_cosine_head_matches_torch():
    import grip_amd  # noqa: F401
    from grip_amd import engine
    g = torch.Generator(device="cuda").manual_seed(5)
    img = torch.randn(1000, 512, device="cuda", generator=g)
    txt = torch.randn(102, 512, device="cuda", generator=g)
    logits, probs, am_l, am_p = engine.cosine_head(img, txt, 100.0)
    i = img / img.norm(dim=-1, keepdim=True)
    t = txt / txt.norm(dim=-1, keepdim=True)
    ref = 100.0 * i @ t.t()
    torch.testing.assert_close(logits, ref, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(probs, ref.softmax(-1), rtol=1e-3, atol=1e-6)
    assert (am_l.long() == logits.argmax(1)).all()      # first-max-wins arg-max of its own logits: exact
    assert (am_p.long() == probs.argmax(1)).all()


@pytest.mark.parametrize("M,N,K,ksplit", [(2142, 512, 2048, 0), (2142, 512, 1536, 0), (3408, 768, 3072, 0), (100, 128, 512, 2), (777, 256, 1536, 8),
                                          (40000, 768, 3072, 0)])
def test_split_k_input_gradient_gemm(M, N, K, ksplit):
    """The EPI_F32 input-gradient products of the prompt steps (few output tiles, long K) are split over K: each workgroup writes
    a partial product; the partials, summed in index order (what ln_bwd_add does), equal the product."""
    import ctypes
    native, lib = _lib()
    g = torch.Generator(device="cuda").manual_seed(M + K)
    Mp = (M + 255) // 256 * 256
    A = torch.randn(Mp, K, device="cuda", generator=g).half()
    A[M:] = float("nan")
    W = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).half()
    used = ctypes.c_int(0)
    out = torch.full((8, Mp, N), float("nan"), device="cuda")
    native.check(lib.grip_debug_gemm_splitk(_p(A), _p(W), M, N, K, _p(out), ksplit, Mp * N, ctypes.addressof(used), Mp, 0, _stream()))
    torch.cuda.synchronize()
    ks = used.value
    assert ks == ksplit or (ksplit == 0 and 1 <= ks <= 8)
    if ksplit == 0 and M < 4000:
        assert ks > 1, "the prompt-step shapes are the ones the split exists for"
    if M >= 40000:
        assert ks == 1
    assert torch.isnan(out[ks:]).all() and torch.isnan(out[:ks, M:]).all(), "wrote outside its partial buffers / rows"
    kp = K // ks
    for p in range(ks):
        ref = A[:M, p * kp:(p + 1) * kp].float() @ W[:, p * kp:(p + 1) * kp].float().t()
        torch.testing.assert_close(out[p, :M], ref, rtol=1e-3, atol=1e-3)
    torch.testing.assert_close(out[:ks, :M].sum(0), A[:M].float() @ W.float().t(), rtol=1e-3, atol=2e-3)


@pytest.mark.parametrize("M,d,N2", [(200, 256, 768), (3408, 768, 2304), (66000, 768, 3072), (35000, 1024, 1024), (77 * 21, 512, 1536)])
@pytest.mark.parametrize("variant", [0, 1, 2, 3, 4, 6])
def test_layernorm_folded_into_gemm(M, d, N2, variant):
    """The LayerNorm-free block structure of the f16 towers: (1) the residual GEMM epilogue emits the per-row partial sums of
    the stream it writes, (2) ln_stats_finalize turns them into (mean, rstd), (3) the consumer GEMM multiplies the RAW stream
    with gamma-scaled weights and applies rstd (acc - mean colsum) + (W beta + b) [+ QuickGELU] in its epilogue -- against
    LayerNorm -> Linear computed by torch in fp32 on the same f16-rounded stream."""
    if variant in (2, 6) and (d % 256 or N2 % 256):
        pytest.skip("256x256 tile needs N % 256 == 0")
    if variant in (2, 3, 6) and M < 1024:
        pytest.skip("large-tile kernels on a tiny problem add nothing")
    native, lib = _lib()
    g = torch.Generator(device="cuda").manual_seed(M + d)
    Mp = (M + 255) // 256 * 256
    A = torch.randn(Mp, d, device="cuda", generator=g).half()
    A[M:] = float("nan")
    W1 = (torch.randn(d, d, device="cuda", generator=g) * d ** -0.5).half()
    b1 = torch.randn(d, device="cuda", generator=g)
    resid = (torch.randn(M, d, device="cuda", generator=g) * 2 + 0.7 * torch.randn(M, 1, device="cuda", generator=g) + 0.5).half()
    # (1) residual epilogue + statistics
    x = torch.full((Mp, d), float("nan"), device="cuda", dtype=torch.float16)
    parts = d // 64
    stat = torch.full((parts, M, 2), float("nan"), device="cuda")         # [d / 64][M]: one (sum, sum of squares) pair per row and 64-column tile
    native.check(lib.grip_debug_gemm_ln(3, _p(A), _p(W1), M, d, d, _p(b1), _p(resid), _p(x), None, _p(stat), None, None, Mp, variant, _stream()))
    ref_x32 = A[:M].float() @ W1.float().t() + b1 + resid.float()
    torch.testing.assert_close(x[:M].float(), ref_x32, rtol=2e-3, atol=2e-3)
    tiles = ref_x32.reshape(M, parts, 64).transpose(0, 1)
    torch.testing.assert_close(stat[..., 0], tiles.sum(-1), rtol=1e-3, atol=2e-2)
    torch.testing.assert_close(stat[..., 1], (tiles ** 2).sum(-1), rtol=1e-3, atol=5e-2)
    # (2) finalize + folded weights
    W2 = (torch.randn(N2, d, device="cuda", generator=g) * d ** -0.5).half()
    b2 = torch.randn(N2, device="cuda", generator=g)
    gamma = 1 + 0.3 * torch.randn(d, device="cuda", generator=g)
    beta = 0.2 * torch.randn(d, device="cuda", generator=g)
    Wg = torch.empty_like(W2)
    cs, bb = torch.empty(N2, device="cuda"), torch.empty(N2, device="cuda")
    rowstat = torch.empty(Mp, 2, device="cuda")
    native.check(lib.grip_debug_ln_fold(_p(W2), _p(gamma), _p(beta), _p(b2), _p(Wg), _p(cs), _p(bb), N2, d, _p(stat), parts, _p(rowstat), M, d, _stream()))
    xs = x[:M].float()
    torch.testing.assert_close(rowstat[:M, 0], xs.mean(-1), rtol=1e-3, atol=1e-3)
    torch.testing.assert_close(rowstat[:M, 1], (xs.var(-1, unbiased=False) + 1e-5).rsqrt(), rtol=2e-3, atol=1e-4)
    torch.testing.assert_close(cs, Wg.float().sum(-1), rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(bb, b2 + W2.float() @ beta, rtol=1e-4, atol=1e-4)
    # (3) the folded consumer GEMM against LayerNorm -> Linear in fp32
    want = torch.nn.functional.layer_norm(xs, (d,), gamma, beta, 1e-5) @ W2.float().t() + b2
    out = torch.full((M, N2), float("nan"), device="cuda", dtype=torch.float16)
    native.check(lib.grip_debug_gemm_ln(7, _p(x), _p(Wg), M, N2, d, _p(bb), None, _p(out), None, None, _p(rowstat), _p(cs), Mp, variant, _stream()))
    torch.testing.assert_close(out.float(), want, rtol=4e-3, atol=4e-3)
    pre = torch.full((M, N2), float("nan"), device="cuda", dtype=torch.float16)
    native.check(lib.grip_debug_gemm_ln(8, _p(x), _p(Wg), M, N2, d, _p(bb), None, _p(out), _p(pre), None, _p(rowstat), _p(cs), Mp, variant, _stream()))
    torch.testing.assert_close(pre.float(), want, rtol=4e-3, atol=4e-3)
    torch.testing.assert_close(out.float(), quick_gelu(want), rtol=4e-3, atol=4e-3)


@pytest.mark.parametrize("K", [3072, 768])
def test_the_96_row_loader_wave_tile_is_the_128_row_one_bit_for_bit(K):
    """r06: the K = 4 d residual GEMM of an image-tower prompt step (M = 3 408 rows, N = 768, K = 3 072) runs on 96-row tiles of the loader-wave ring
    (216 workgroups instead of 162 on 256 CUs).  The tile shape changes which workgroup computes a row, never the row: same K order, same
    statistics tree -- bit-identical to the 128-row launch (debug variant 1) with and without the row statistics, and correct against float64."""
    native, lib = _lib()
    M, N = 3408, 768         # K = 3 072: the c_proj forward; K = 768: the out-proj forward and (plain f16 epilogue) its input gradient
    Mp = (M + 255) // 256 * 256
    g = torch.Generator(device="cuda").manual_seed(96)
    A = torch.randn(Mp, K, device="cuda", generator=g).half()
    A[M:] = float("nan")
    W = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).half()
    bias = torch.randn(N, device="cuda", generator=g)
    resid = torch.randn(M, N, device="cuda", generator=g).half()
    ref = A[:M].double() @ W.double().t() + bias.double() + resid.double()
    outs, stats = {}, {}
    for variant in (0, 1):          # 0 = the launcher's choice (the 96-row form for this shape), 1 = 128 x 128 tiles
        o = torch.zeros(M, N, device="cuda", dtype=torch.float16)
        native.check(lib.grip_debug_gemm(3, _p(A), _p(W), M, N, K, _p(bias), _p(resid), None, _p(o), None, 1.0, Mp, variant, _stream()))
        o2 = torch.zeros(M, N, device="cuda", dtype=torch.float16)
        st = torch.zeros(N // 64, M, 2, device="cuda")
        native.check(lib.grip_debug_gemm_ln(3, _p(A), _p(W), M, N, K, _p(bias), _p(resid), _p(o2), None, _p(st), None, None, Mp, variant, _stream()))
        assert torch.equal(o, o2)
        outs[variant], stats[variant] = o, st
    assert torch.equal(outs[0], outs[1]) and torch.equal(stats[0], stats[1])
    plain = {}
    for variant in (0, 1):          # EPI_F16 (the dgrad GEMMs of the step)
        o = torch.zeros(M, N, device="cuda", dtype=torch.float16)
        native.check(lib.grip_debug_gemm(4, _p(A), _p(W), M, N, K, None, None, None, _p(o), None, 1.0, Mp, variant, _stream()))
        plain[variant] = o
    assert torch.equal(plain[0], plain[1])
    torch.testing.assert_close(plain[0].double(), A[:M].double() @ W.double().t(), rtol=2e-3, atol=2e-3)
    torch.testing.assert_close(outs[0].double(), ref, rtol=2e-3, atol=2e-3)
    # the statistics are (sum, sum of squares) of what the epilogue added into the stream, per 64-column tile
    v = (A[:M].float() @ W.float().t() + bias + resid.float()).reshape(M, N // 64, 64)
    torch.testing.assert_close(stats[0][..., 0].t(), v.sum(-1), rtol=1e-3, atol=2e-2)


@pytest.mark.parametrize("M,N,K", [(425, 512, 512), (425, 1536, 512), (425, 2048, 512), (201, 512, 2048)])
def test_the_32_row_loader_wave_tile_is_the_64_row_one_bit_for_bit(M, N, K):
    """r06: the text tower's GEMMs of a CoOp step (M = 17 + 102 x 4 = 425 rows) run on 32-row tiles of the loader-wave ring where the 64-row tiles fill at
    most half the chip.  Every epilogue -- f32, bias, bias + QuickGELU, residual (with and without the row statistics), plain f16, GELU-gradient, and the two
    LayerNorm-folded ones -- must give the bits of the 64-row launch (debug variant 4)."""
    native, lib = _lib()
    Mp = (M + 255) // 256 * 256
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    A = (torch.randn(Mp, K, device="cuda", generator=g) * 2 + 0.3).half()
    A[M:] = float("nan")
    W = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).half()
    bias = torch.randn(N, device="cuda", generator=g)
    resid = torch.randn(M, N, device="cuda", generator=g).half()
    aux = torch.randn(M, N, device="cuda", generator=g).half()
    colsum = W.float().sum(1).contiguous()
    rowstat = torch.stack([A[:M].float().mean(1), torch.rsqrt(A[:M].float().var(1, unbiased=False) + 1e-5)], 1).contiguous()
    got = {}
    for variant in (0, 4):
        outs = []
        o32 = torch.zeros(M, N, device="cuda")
        native.check(lib.grip_debug_gemm(0, _p(A), _p(W), M, N, K, None, None, None, _p(o32), None, 1.0, Mp, variant, _stream()))
        outs.append(o32)
        for epi, b, r, x in ((1, bias, None, None), (2, bias, None, None), (3, bias, resid, None), (4, None, None, None), (5, None, None, aux)):
            o = torch.zeros(M, N, device="cuda", dtype=torch.float16)
            pre = torch.zeros(M, N, device="cuda", dtype=torch.float16) if epi == 2 else None
            native.check(lib.grip_debug_gemm(epi, _p(A), _p(W), M, N, K, _p(b), _p(r), _p(x), _p(o), _p(pre), 1.0, Mp, variant, _stream()))
            outs += [o] + ([pre] if pre is not None else [])
        st = torch.zeros(N // 64, M, 2, device="cuda")
        o = torch.zeros(M, N, device="cuda", dtype=torch.float16)
        native.check(lib.grip_debug_gemm_ln(3, _p(A), _p(W), M, N, K, _p(bias), _p(resid), _p(o), None, _p(st), None, None, Mp, variant, _stream()))
        outs += [o, st]
        for epi in (7, 8):
            o = torch.zeros(M, N, device="cuda", dtype=torch.float16)
            pre = torch.zeros(M, N, device="cuda", dtype=torch.float16) if epi == 8 else None
            native.check(lib.grip_debug_gemm_ln(epi, _p(A), _p(W), M, N, K, _p(bias), None, _p(o), _p(pre), None, _p(rowstat), _p(colsum), Mp, variant, _stream()))
            outs += [o] + ([pre] if pre is not None else [])
        got[variant] = outs
    for a, b in zip(got[0], got[4]):
        assert torch.equal(a, b)
    torch.testing.assert_close(got[0][0], A[:M].float() @ W.float().t(), rtol=1e-4, atol=1e-3)
