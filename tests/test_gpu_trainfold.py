"""The two GEMM forms the text tower's TRAIN-mode forward uses since r04 (csrc/gemm.hip, gemm_ringw_kernel) and the tower-level behaviour built on them:

  (1) LayerNorm folded into the train-mode QKV / c_fc GEMMs with the producer's partial row sums consumed inside the GEMM (GemmArgs::stat_in) -- no LayerNorm and
      no finalising launch in a prompt step;
  (2) cooperative split-K of the K = 4 d residual GEMM (c_proj): partial tiles through a scratch buffer, the last wave to arrive adds them in split order.

Kernel level: against torch fp32 on the same f16-rounded operands (the tolerances of test_gpu_kernels.py::test_layernorm_folded_into_gemm), bit-equal run to run.
Tower level: the train-mode forward of the text tower against its inference forward and run-to-run bit equality.  The reference computation is
models/clip_encoders.py:43-90 (CustomTextEncoder.forward); its gradient stays pinned by tests/test_gpu_backward.py / test_gpu_trajectory.py on reference-run fixtures."""
import numpy as np
import pytest
import torch

from test_gpu_kernels import _lib, _p, _stream, quick_gelu

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("M,d,N2", [(425, 512, 1536), (425, 512, 2048), (200, 256, 768), (77 * 21, 512, 1536), (64, 768, 2304), (3408, 768, 2304)])
def test_folded_gemm_reads_the_partial_sums(M, d, N2):
    native, lib = _lib()
    g = torch.Generator(device="cuda").manual_seed(M + d)
    Mp = (M + 255) // 256 * 256
    parts = d // 64
    x = (torch.randn(Mp, d, device="cuda", generator=g) * 2 + 0.7 * torch.randn(Mp, 1, device="cuda", generator=g) + 0.5).half()
    xs = x[:M].float()
    tiles = xs.reshape(M, parts, 64).transpose(0, 1)
    stat = torch.stack([tiles.sum(-1), (tiles ** 2).sum(-1)], dim=-1).contiguous()       # [parts][M][2], what the residual epilogue emits
    x[M:] = float("nan")
    W2 = (torch.randn(N2, d, device="cuda", generator=g) * d ** -0.5).half()
    b2 = torch.randn(N2, device="cuda", generator=g)
    gamma = 1 + 0.3 * torch.randn(d, device="cuda", generator=g)
    beta = 0.2 * torch.randn(d, device="cuda", generator=g)
    Wg = torch.empty_like(W2)
    cs, bb = torch.empty(N2, device="cuda"), torch.empty(N2, device="cuda")
    native.check(lib.grip_debug_ln_fold(_p(W2), _p(gamma), _p(beta), _p(b2), _p(Wg), _p(cs), _p(bb), N2, d, None, 0, None, M, d, _stream()))
    want = torch.nn.functional.layer_norm(xs, (d,), gamma, beta, 1e-5) @ W2.float().t() + b2
    rowstat = torch.full((Mp, 2), float("nan"), device="cuda")       # written only when the launcher has to finalise (no loader-wave kernel for the shape)
    outs = []
    for epi in (7, 8):
        out = torch.full((M, N2), float("nan"), device="cuda", dtype=torch.float16)
        pre = torch.full((M, N2), float("nan"), device="cuda", dtype=torch.float16)
        native.check(lib.grip_debug_gemm_train(epi, _p(x), _p(Wg), M, N2, d, _p(bb), None, _p(out), _p(pre) if epi == 8 else None, None, _p(rowstat), _p(cs),
                                               _p(stat), parts, 0, None, None, Mp, _stream()))
        if epi == 7:
            torch.testing.assert_close(out.float(), want, rtol=4e-3, atol=4e-3)
        else:
            torch.testing.assert_close(pre.float(), want, rtol=4e-3, atol=4e-3)
            torch.testing.assert_close(out.float(), quick_gelu(want), rtol=4e-3, atol=4e-3)
        outs.append(out)
    # the same bits as the two-launch form (ln_stats_finalize, then the GEMM on finalised statistics): same arithmetic in the same order
    rs = torch.empty(Mp, 2, device="cuda")
    native.check(lib.grip_debug_ln_fold(_p(W2), _p(gamma), _p(beta), _p(b2), _p(Wg), _p(cs), _p(bb), N2, d, _p(stat), parts, _p(rs), M, d, _stream()))
    out2 = torch.full((M, N2), float("nan"), device="cuda", dtype=torch.float16)
    native.check(lib.grip_debug_gemm_train(7, _p(x), _p(Wg), M, N2, d, _p(bb), None, _p(out2), None, None, _p(rs), _p(cs), None, 0, 0, None, None, Mp, _stream()))
    assert torch.equal(out2, outs[0])


@pytest.mark.parametrize("M,d,K,ks", [(425, 512, 2048, 4), (425, 512, 2048, 2), (425, 512, 2048, 8), (64, 512, 2048, 4), (130, 768, 3072, 4), (1000, 256, 1024, 2), (2142, 512, 2048, 0)])
@pytest.mark.parametrize("stats", [False, True])
def test_cooperative_split_k_residual_gemm(M, d, K, ks, stats):
    native, lib = _lib()
    auto = lib.grip_debug_coop_split(M, d, K)
    if ks == 0:
        assert auto == 1, "7 x 34 tiles are a launch of their own: no split"
        return
    if M == 425 and K == 2048 and ks == 4:
        assert auto == 4                      # the CoOp step's c_proj: 28 tiles x 32 slices -> 4 x 8
    g = torch.Generator(device="cuda").manual_seed(M + K + ks)
    Mp = (M + 255) // 256 * 256
    A = torch.randn(Mp, K, device="cuda", generator=g).half()
    A[M:] = float("nan")
    W = (torch.randn(d, K, device="cuda", generator=g) * K ** -0.5).half()
    b = torch.randn(d, device="cuda", generator=g)
    resid = (torch.randn(M, d, device="cuda", generator=g) * 2).half()
    tiles = (M + 63) // 64 * (d // 128)
    scratch = torch.full((tiles * ks * 8192,), float("nan"), device="cuda")
    cnt = torch.zeros(tiles * 4, dtype=torch.int32, device="cuda")
    parts = d // 64
    want = A[:M].float() @ W.float().t() + b + resid.float()
    outs = []
    for rep in range(3):
        x = torch.full((Mp, d), float("nan"), device="cuda", dtype=torch.float16)
        stat = torch.full((parts, M, 2), float("nan"), device="cuda") if stats else None
        native.check(lib.grip_debug_gemm_train(3, _p(A), _p(W), M, d, K, _p(b), _p(resid), _p(x), None, _p(stat), None, None, None, 0, ks,
                                               _p(scratch), _p(cnt), Mp, _stream()))
        torch.testing.assert_close(x[:M].float(), want, rtol=2e-3, atol=4e-3)
        assert int(cnt.abs().sum()) == 0, "the tickets return to zero"
        if stats:
            t = x[:M].float().reshape(M, parts, 64).transpose(0, 1)
            torch.testing.assert_close(stat[..., 0], t.sum(-1), rtol=1e-3, atol=2e-2)
            torch.testing.assert_close(stat[..., 1], (t ** 2).sum(-1), rtol=1e-3, atol=5e-2)
        outs.append(x[:M].clone())
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]), "split order is fixed: bit-equal run to run"
    # against the unsplit kernel: same products, another summation tree -- f16 rounding of the stored stream is all that may differ
    x1 = torch.full((Mp, d), float("nan"), device="cuda", dtype=torch.float16)
    native.check(lib.grip_debug_gemm_train(3, _p(A), _p(W), M, d, K, _p(b), _p(resid), _p(x1), None, None, None, None, None, 0, 0, None, None, Mp, _stream()))
    torch.testing.assert_close(x1[:M].float(), outs[0].float(), rtol=2e-3, atol=2e-3)


@pytest.mark.parametrize("name,C,P,per_class", [("small", 10, 4, False), ("ViT-B/16", 102, 16, False), ("ViT-B/16", 12, 16, True), ("ViT-B/16", 3, 16, False)])
def test_text_train_forward_equals_inference_forward(name, C, P, per_class):
    """Train mode (folded LayerNorm from partial sums, cooperative c_proj, activations saved) against the inference forward of the same tower: the same function,
    f16 rounding apart; and bit-equal run to run (what a HIP-graph replay of a prompt step relies on)."""
    import grip_amd  # noqa: F401
    from grip_amd import clip, rng
    from grip_amd.engine import TextPrefixFn
    m, _ = clip.load(name, device="cuda")
    d = m.dims.transformer_width
    classes = [f"kind number {i}" if i % 3 else f"a rather longer class name {i}" for i in range(C)]
    tok = clip.tokenize([" ".join(["X"] * P + [c]) for c in classes]).cuda()
    prefix = torch.from_numpy(rng.normal(5, rng.stream_id("tf.prefix"), (C if per_class else 1, P, d), 0.0, 0.02)).cuda()
    with torch.no_grad():
        inf, _, _ = m.text_tower.text_forward(tok, prefix)
    outs, grads = [], []
    for rep in range(3):
        p = prefix.clone().requires_grad_(True)
        out = TextPrefixFn.apply(m.text_tower, tok.clone(), p)
        (out ** 2).sum().backward()
        outs.append(out.detach().clone())
        grads.append(p.grad.clone())
    cos = torch.nn.functional.cosine_similarity(outs[0], inf, dim=-1)
    rel = ((outs[0] - inf).norm() / inf.norm()).item()
    print(f"{name} C={C}: train vs inference forward 1-cos {float((1 - cos).max()):.2e} rel {rel:.2e}")
    assert float((1 - cos).max()) <= 2e-5 and rel <= 5e-3
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    assert torch.equal(grads[0], grads[1]) and torch.equal(grads[0], grads[2])
