"""The split-f16 tier (dims.precision = 2; csrc/gemm_split.hip): every GEMM operand carried as an f16 hi / lo pair, three f16 MFMA products
per fragment pair, f32 accumulation -- the MIDDLE tier of the screen-and-refine pseudolabel pass (pseudolabels.refine_scan).

Kernel level: the GEMM against a float64 product of the same f32 inputs (relative error <= 1e-5 of the row/column scale, next to what
the f32 MFMA kernel achieves on the same problem).  Tower level: embeddings and probabilities against the exact (f32) twin.  Pass
level: the three-tier identical_lists returns the exact mode's lists with only a fraction of the re-encoded rows on the f32 tower.
The decisions the lists depend on are the reference's fp32 ones (utils/clip_pseudolabels.py:38-41, 73-101); this tier never decides --
it narrows what the f32 tower has to look at."""
import ctypes

import numpy as np
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


LO_SCALE = 1.0       # csrc/common.h GRIP_SPLIT_LO_SCALE: lo' = f16((x - hi) * LO_SCALE)
W_SCALE = 256.0      # csrc/gemm_split.hip SP1_W_SCALE: weight images carry this (exact) factor, the epilogue divides it out


def _unsplit(buf, rows, K):
    """Split layout [rows, K/32, (32 hi | 32 lo')] f16 -> the values hi + lo' / LO_SCALE."""
    v = buf.view(torch.float16).reshape(rows, K // 32, 2, 32).double()
    return (v[:, :, 0] + v[:, :, 1] / LO_SCALE).reshape(rows, K)


def quick_gelu(x):
    return x * torch.sigmoid(1.702 * x)


@pytest.mark.parametrize("M,N,K,scale", [(256, 256, 64, 1.0), (200, 512, 128, 1.0), (3408, 2304, 768, 1.0), (5000, 768, 3072, 1.0), (1000, 3072, 768, 1.0),
                                         (777, 256, 768, 1e-3), (777, 256, 768, 300.0), (300, 256, 256, 1e-5)])
def test_split_gemm_against_float64(M, N, K, scale):
    native, lib = _lib()
    g = torch.Generator(device="cuda").manual_seed(M * 13 + N + K)
    Mp = (M + 255) // 256 * 256
    A = torch.randn(Mp, K, device="cuda", generator=g) * scale
    A[M:] = float("nan")       # padding rows must never leak into stored rows
    # a heavy-tailed column scale, as LayerNorm outputs / MLP hiddens have
    A[:M] *= torch.exp(torch.randn(1, K, device="cuda", generator=g) * 0.7)
    W = torch.randn(N, K, device="cuda", generator=g) * K ** -0.5
    bias = torch.randn(N, device="cuda", generator=g)
    resid = torch.randn(M, N, device="cuda", generator=g)
    a_s = torch.empty(Mp * K, device="cuda")
    w_s = torch.empty(N * K, device="cuda")
    ref = A[:M].double() @ W.double().t()
    # error scale of a dot product: |a| . |w| (what rounding errors are proportional to), per output element
    mag = A[:M].double().abs() @ W.double().abs().t()

    # ... plus the absolute floor of an UNSCALED lo part: below the f16 normal range (|a| < 0.125) it is quantised to 2^-24, i.e. up to 3e-8 per
    # element of a whatever its size -- nothing next to O(1) activations, visible only when a whole operand is tiny (the 1e-3 / 1e-5 cases)
    floor = 3.1e-8 * W.double().abs().sum(1)[None, :]

    def rel(out, want):
        return (((out.double() - want).abs() - floor).clamp_min(0) / mag).max().item()

    out = torch.full((M, N), 7.0, device="cuda")
    native.check(lib.grip_debug_gemm_split(0, _p(A), _p(W), M, N, K, None, None, _p(out), _p(a_s), _p(w_s), Mp, _stream()))
    e_split = rel(out, ref)
    # the operand images are what the layout says
    # (lo is unscaled: below the f16 normal range it is quantised to 2^-24, an absolute error of at most 3e-8 -- the single-accumulator trade)
    torch.testing.assert_close(_unsplit(a_s, Mp, K)[:M], A[:M].double(), rtol=2e-6, atol=3.1e-8)
    torch.testing.assert_close(_unsplit(w_s, N, K) / W_SCALE, W.double(), rtol=2e-6, atol=3.1e-8 / W_SCALE)
    out32 = torch.empty(M, N, device="cuda")
    native.check(lib.grip_debug_gemm(0, _p(A), _p(W), M, N, K, None, None, None, _p(out32), None, 1.0, Mp, 7, _stream()))
    e_f32 = rel(out32, ref)
    e_f16 = rel(A[:M].half().float() @ W.half().float().t(), ref)
    print(f"M={M} N={N} K={K} scale={scale}: max |err| / (|a|.|w|): split {e_split:.2e}, f32 MFMA kernel {e_f32:.2e}, f16 operands {e_f16:.2e}")
    assert e_split <= 4e-7, e_split                      # ~3 x 2^-23 per term, averaged down by the sum
    # relative to the value itself: row-wise in norm, and element-wise wherever the sum did not cancel to below 5 % of its terms
    if scale >= 1.0:
        assert ((out.double() - ref).norm(dim=1) / ref.norm(dim=1)).max().item() <= 1e-6
        assert ((out.double() - ref).abs() / ref.abs().clamp_min(0.05 * mag)).max().item() <= 1e-5
    assert e_split <= 30 * max(e_f32, 1e-8) and (scale < 1.0 or e_split <= e_f16 / 100)

    def worst(got, want, extra):
        """max of (|got - want| - floor) / (error scale of the dot product + the magnitudes of what the epilogue adds)"""
        return (((got.double() - want).abs() - floor).clamp_min(0) / (mag + extra + want.abs())).max().item()

    native.check(lib.grip_debug_gemm_split(1, _p(A), _p(W), M, N, K, _p(bias), None, _p(out), _p(a_s), _p(w_s), Mp, _stream()))
    assert worst(out, ref + bias.double(), bias.double().abs()) <= 5e-7
    native.check(lib.grip_debug_gemm_split(3, _p(A), _p(W), M, N, K, _p(bias), _p(resid), _p(out), _p(a_s), _p(w_s), Mp, _stream()))
    assert worst(out, ref + bias.double() + resid.double(), bias.double().abs() + resid.double().abs()) <= 5e-7
    r2 = resid.clone()             # in place (out aliases resid), as the tower uses it
    native.check(lib.grip_debug_gemm_split(3, _p(A), _p(W), M, N, K, _p(bias), _p(r2), _p(r2), _p(a_s), _p(w_s), Mp, _stream()))
    assert torch.equal(r2, out)
    # QuickGELU epilogue: written in the split layout (it feeds the next split GEMM)
    h = torch.zeros(M * N, device="cuda")
    native.check(lib.grip_debug_gemm_split(2, _p(A), _p(W), M, N, K, _p(bias), None, _p(h), _p(a_s), _p(w_s), Mp, _stream()))
    assert worst(_unsplit(h, M, N), quick_gelu(ref + bias.double()), bias.double().abs() + 0.1) <= 8e-7      # (+ 3e-8 absolute: an unscaled lo again)


def test_split_gemm_is_deterministic_and_row_independent():
    """A row's result does not depend on which rows surround it (the refinement re-encodes gathered, arbitrarily chunked rows)."""
    native, lib = _lib()
    g = torch.Generator(device="cuda").manual_seed(5)
    M, N, K = 1500, 768, 768
    Mp = (M + 255) // 256 * 256
    A = torch.randn(Mp, K, device="cuda", generator=g)
    W = torch.randn(N, K, device="cuda", generator=g) * K ** -0.5
    a_s, w_s = torch.empty(Mp * K, device="cuda"), torch.empty(N * K, device="cuda")
    out = torch.empty(M, N, device="cuda")
    native.check(lib.grip_debug_gemm_split(0, _p(A), _p(W), M, N, K, None, None, _p(out), _p(a_s), _p(w_s), Mp, _stream()))
    out2 = torch.empty(M, N, device="cuda")
    native.check(lib.grip_debug_gemm_split(0, _p(A), _p(W), M, N, K, None, None, _p(out2), _p(a_s), _p(w_s), Mp, _stream()))
    assert torch.equal(out, out2)
    sub = torch.zeros(256, K, device="cuda")
    sub[:100] = A[700:800]
    o3 = torch.empty(100, N, device="cuda")
    native.check(lib.grip_debug_gemm_split(0, _p(sub), _p(W), 100, N, K, None, None, _p(o3), _p(a_s), _p(w_s), 256, _stream()))
    assert torch.equal(o3, out[700:800])


@pytest.mark.parametrize("name,n,res", [("small", 96, 64), ("ViT-B/16", 48, 224), ("ViT-L/14@336px", 6, 336)])      # (S = 577: f32 attention, split output)
def test_split_tower_tracks_the_exact_twin(name, n, res):
    """Embeddings of the split-f16 vision tower against the f32 twin's on the same structured images (with and without a visual prompt):
    two orders of magnitude closer than the f16 tower's, and chunking-independent bit for bit."""
    import grip_amd  # noqa: F401
    from grip_amd import clip, engine, pseudolabels as pl, rng
    from grip_amd.data.synthetic import structured_images
    m, _ = clip.load(name, device="cuda")
    twin, split = m.exact_twin(), m.split_twin()
    assert split is not None and split.precision == 2 and not split.exact
    x = structured_images(77, 0, n, res).cuda()
    d = m.dims
    prefix = torch.from_numpy(rng.normal(3, rng.stream_id("split.prefix"), (4, d.vision_width), 0.0, 0.05)).cuda()
    C = 20
    tok = clip.tokenize([f"a photo of a thing number {i}" for i in range(C)]).cuda()
    with torch.no_grad():
        txt = twin.encode_text(tok)
        for pf in (None, prefix):
            e32 = pl.encode_pool(twin.visual.tower, x, chunk=32, prefix=pf)
            es = pl.encode_pool(split.visual.tower, x, chunk=32, prefix=pf)
            es2 = pl.encode_pool(split.visual.tower, x, chunk=7, prefix=pf)
            e16 = pl.encode_pool(m.visual.tower, x, chunk=32, prefix=pf)
            assert torch.equal(es, es2)
            r_s = ((es - e32).norm(dim=1) / e32.norm(dim=1)).max().item()
            r_16 = ((e16 - e32).norm(dim=1) / e32.norm(dim=1)).max().item()
            _, p32, _, _ = engine.cosine_head(e32, txt, 100.0)
            _, ps, _, _ = engine.cosine_head(es, txt, 100.0)
            _, p16, _, _ = engine.cosine_head(e16, txt, 100.0)
            d_s = ((ps - p32).abs() / p32).max().item()
            d_16 = ((p16 - p32).abs() / p32).max().item()
            print(f"{name} prefix={pf is not None}: embedding rel L2 split {r_s:.2e} / f16 {r_16:.2e}; probability deviation split {d_s:.2e} / f16 {d_16:.2e}")
            assert r_s <= 5e-6 and d_s <= 2e-4 and d_s <= d_16 / 30


def test_three_tier_identical_lists_equal_the_exact_mode():
    """ViT-B/16, 6 000 structured images x 40 classes, k = 8: f16 screen -> split-f16 tier -> f32 tower; the lists are the exact mode's and the
    f32 tower sees the calibration / audit rows plus a small remainder."""
    import grip_amd  # noqa: F401
    from grip_amd import clip, engine, pseudolabels as pl
    from grip_amd.data.synthetic import pool_paths
    m, _ = clip.load("ViT-B/16", device="cuda")
    twin, split = m.exact_twin(), m.split_twin()
    n, C, k = 6000, 40, 8
    from conftest import structured_pool
    x = structured_pool(31, n, 224)
    tok = clip.tokenize([f"a photo of a kind {i}" for i in range(C)]).cuda()
    paths, labels = pool_paths(n), list(range(C))
    with torch.no_grad():
        txt = twin.encode_text(tok)
        e32 = pl.encode_pool(twin.visual.tower, x, chunk=220)
    _, p32, _, a32 = engine.cosine_head(e32, txt, 100.0)
    want = pl.leaderboard(p32.cpu().numpy(), a32.cpu().numpy(), paths, labels, k)
    got3 = pl.identical_lists(m.visual.tower, twin.visual.tower, x, txt, 100.0, paths, labels, k, chunk=440, visual_mid=split.visual.tower)
    st3 = dict(pl.LAST_REFINE_STATS)
    got2 = pl.identical_lists(m.visual.tower, twin.visual.tower, x, txt, 100.0, paths, labels, k, chunk=440)
    st2 = dict(pl.LAST_REFINE_STATS)
    print(f"three tiers: {st3['rows_mid']} split-f16 rows, {st3['rows_exact']} f32 rows (bounds {st3['eps']:.2e} / {st3['eps_mid']:.2e}); two tiers: {st2['rows_exact']} f32 rows")
    assert (list(got3[0]), list(got3[1])) == (list(want[0]), list(want[1]))
    assert (list(got2[0]), list(got2[1])) == (list(want[0]), list(want[1]))
    assert st3["tiers"] == 3 and st3["eps_mid"] < st3["eps"] / 30
    assert st3["rows_exact"] - st3["calibration_rows"] < 0.5 * (st2["rows_exact"] - st2["calibration_rows"]), (st3, st2)


@pytest.mark.parametrize("B,S,H,causal", [(3, 197, 12, 0), (2, 213, 12, 0), (2, 50, 12, 0), (5, 77, 8, 1), (4, 21, 8, 1), (1, 320, 4, 0), (2, 257, 16, 0), (3, 1, 2, 1)])
def test_split_attention_against_float64(B, S, H, causal):
    """attention_split.hip (both products as three f16 MFMAs on hi / lo' pairs, f32 softmax) against float64 attention on the same f32
    projections, next to the f32 vector-ALU kernel writing the same layout; output read back from the split layout."""
    native, lib = _lib()
    g = torch.Generator(device="cuda").manual_seed(S * 31 + H)
    D = H * 64
    qkv = torch.randn(B * S, 3 * D, device="cuda", generator=g)
    qkv[:, :D] *= 2.0          # sharper softmax rows
    q, k, v = (qkv[:, i * D:(i + 1) * D].double().reshape(B, S, H, 64).permute(0, 2, 1, 3) for i in range(3))
    s = q @ k.transpose(-1, -2) / 8.0
    if causal:
        s = s + torch.full((S, S), float("-inf"), device="cuda", dtype=torch.float64).triu(1)
    ref = (s.softmax(-1) @ v).permute(0, 2, 1, 3).reshape(B * S, D)
    errs = {}
    for mfma in (1, 0):
        out = torch.zeros(B * S * D, device="cuda")
        native.check(lib.grip_debug_attention_split(_p(qkv), _p(out), B, S, H, causal, mfma, _stream()))
        got = _unsplit(out, B * S, D)
        errs[mfma] = ((got - ref).abs().max() / ref.abs().max()).item()
    print(f"B={B} S={S} H={H} causal={causal}: max err / max |out|: split MFMA kernel {errs[1]:.2e}, f32 VALU kernel {errs[0]:.2e}")
    assert errs[1] <= 2e-6 and errs[0] <= 2e-6


def test_long_sequences_keep_the_f32_attention_with_split_output():
    """S = 577 (ViT-L/14@336px) is beyond attention_split.hip's four LDS planes: a precision-2 tower then runs the f32 vector-ALU attention and
    only its OUTPUT is written in the split layout (csrc/tower.hip run_blocks)."""
    native, lib = _lib()
    g = torch.Generator(device="cuda").manual_seed(9)
    B, S, H = 1, 577, 16
    D = H * 64
    qkv = torch.randn(B * S, 3 * D, device="cuda", generator=g)
    q, k, v = (qkv[:, i * D:(i + 1) * D].double().reshape(B, S, H, 64).permute(0, 2, 1, 3) for i in range(3))
    ref = ((q @ k.transpose(-1, -2) / 8.0).softmax(-1) @ v).permute(0, 2, 1, 3).reshape(B * S, D)
    out = torch.zeros(B * S * D, device="cuda")
    native.check(lib.grip_debug_attention_split(_p(qkv), _p(out), B, S, H, 0, 0, _stream()))
    assert ((_unsplit(out, B * S, D) - ref).abs().max() / ref.abs().max()).item() <= 2e-6
    with pytest.raises(native.GripError):
        native.check(lib.grip_debug_attention_split(_p(qkv), _p(out), B, S, H, 0, 1, _stream()))


# ------------------------------------------------------------------------------------------ weights that are f16 numbers (r06: GemmArgs::w_exact)
@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (3408, 2304, 768), (5000, 768, 3072), (1000, 3072, 768)])
def test_split_gemm_with_f16_exact_weights_drops_the_w_lo_product(M, N, K):
    """Published CLIP checkpoints hold fp16 weights (the reference's CPU path computes in fp32 on those values cast up, methods/clip_baseline.py:39-41):
    W's lo parts are all zero, the kernel forms a_hi w + a_lo w -- two MFMA passes instead of three -- and the result is as close to the float64
    product as the three-pass form's on general weights.  The same weights pushed off the grid take the three-pass kernel."""
    native, lib = _lib()
    g = torch.Generator(device="cuda").manual_seed(M + 7 * N + K)
    Mp = (M + 255) // 256 * 256
    A = torch.randn(Mp, K, device="cuda", generator=g) * torch.exp(torch.randn(1, K, device="cuda", generator=g) * 0.7)
    W = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).half().float()
    bias = torch.randn(N, device="cuda", generator=g)
    a_s, w_s = torch.empty(Mp * K, device="cuda"), torch.empty(N * K, device="cuda")
    ref = A[:M].double() @ W.double().t()
    mag = A[:M].double().abs() @ W.double().abs().t()
    floor = 3.1e-8 * W.double().abs().sum(1)[None, :]
    out = torch.full((M, N), 7.0, device="cuda")
    native.check(lib.grip_debug_gemm_split(0, _p(A), _p(W), M, N, K, None, None, _p(out), _p(a_s), _p(w_s), Mp, _stream()))
    assert lib.grip_debug_split_last_wlo() == 0                      # the two-pass kernel ran
    v = w_s.view(torch.float16).reshape(N, K // 32, 2, 32)
    assert (v[:, :, 1] == 0).all()                                   # ... because every lo part of W is zero
    err = (((out.double() - ref).abs() - floor).clamp_min(0) / mag).max().item()
    assert err <= 4e-7, err
    assert ((out.double() - ref).norm(dim=1) / ref.norm(dim=1)).max().item() <= 1e-6
    h = torch.zeros(M * N, device="cuda")
    native.check(lib.grip_debug_gemm_split(2, _p(A), _p(W), M, N, K, _p(bias), None, _p(h), _p(a_s), _p(w_s), Mp, _stream()))
    want = quick_gelu(ref + bias.double())
    assert (((_unsplit(h, M, N) - want).abs() - floor).clamp_min(0) / (mag + bias.double().abs() + 0.1 + want.abs())).max().item() <= 8e-7
    W2 = W * (1 + 2.0 ** -13)                                        # off the grid: the lo parts are needed again
    out2 = torch.empty(M, N, device="cuda")
    native.check(lib.grip_debug_gemm_split(0, _p(A), _p(W2), M, N, K, None, None, _p(out2), _p(a_s), _p(w_s), Mp, _stream()))
    assert lib.grip_debug_split_last_wlo() == 1
    ref2 = A[:M].double() @ W2.double().t()
    assert (((out2.double() - ref2).abs() - floor).clamp_min(0) / mag).max().item() <= 4e-7


def test_split_tower_on_an_fp16_checkpoint_tracks_the_exact_twin():
    """clip.load(..., fp16_checkpoint=True): synthetic weights rounded to f16 numbers as a published checkpoint holds them.  The split twin takes
    the two-pass GEMMs (grip_tower_finalize found no non-zero lo part) and stays as close to the f32 twin as on general weights; the f16 tower
    rounds no weight any more and moves closer to the twin than on the un-rounded init."""
    import grip_amd  # noqa: F401
    from grip_amd import clip, pseudolabels as pl
    from grip_amd.data.synthetic import structured_images
    native, lib = _lib()
    x = structured_images(77, 0, 48, 224).cuda()
    devs = {}
    for grid in (False, True):
        m, _ = clip.load("ViT-B/16", device="cuda", fp16_checkpoint=grid)
        twin, split = m.exact_twin(), m.split_twin()
        with torch.no_grad():
            e32 = pl.encode_pool(twin.visual.tower, x, chunk=32)
            es = pl.encode_pool(split.visual.tower, x, chunk=32)
            assert lib.grip_debug_split_last_wlo() == (0 if grid else 1)
            es2 = pl.encode_pool(split.visual.tower, x, chunk=7)
            e16 = pl.encode_pool(m.visual.tower, x, chunk=32)
        assert torch.equal(es, es2)
        r_s = ((es - e32).norm(dim=1) / e32.norm(dim=1)).max().item()
        r_16 = ((e16 - e32).norm(dim=1) / e32.norm(dim=1)).max().item()
        print(f"fp16_checkpoint={grid}: embedding rel L2 split {r_s:.2e} / f16 {r_16:.2e}")
        assert r_s <= 5e-6
        devs[grid] = r_16
        del m, twin, split
        torch.cuda.empty_cache()
    assert devs[True] < devs[False]
