"""Determinism (the build's race check): no kernel uses atomics or order-dependent reductions, so two runs of
the same forward / backward give bit-identical results, also across different batch compositions for the per-row outputs."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_forward_backward_are_bit_reproducible():
    import grip_amd  # noqa: F401
    from grip_amd import clip
    from grip_amd.engine import CosineHeadFn, TextPrefixFn, VitPrefixFn, WeightedCEFn
    m, _ = clip.load("small", device="cuda")
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(24, 3, 64, 64, device="cuda", generator=g)
    vp = torch.randn(4, 256, device="cuda", generator=g) * 0.02
    tp = torch.randn(1, 4, 256, device="cuda", generator=g) * 0.02
    tok = clip.tokenize([f"X X X X thing {i}" for i in range(7)]).cuda()
    labels = torch.arange(24, device="cuda", dtype=torch.int32) % 7
    w = torch.full((24,), 1 / 24, device="cuda")

    def run():
        a, b = vp.clone().requires_grad_(True), tp.clone().requires_grad_(True)
        img = VitPrefixFn.apply(m.visual.tower, x, a)
        txt = TextPrefixFn.apply(m.text_tower, tok, b)
        loss = WeightedCEFn.apply(CosineHeadFn.apply(img, txt, 100.0), labels, w)
        loss.backward()
        torch.cuda.synchronize()
        return img.detach().clone(), txt.detach().clone(), loss.detach().clone(), a.grad.clone(), b.grad.clone()

    first = run()
    for _ in range(3):
        for u, v in zip(first, run()):
            assert torch.equal(u, v)
    # Per-image outputs of the INFERENCE forward do not depend on what else is in the batch, bit for bit (the pool encode under any
    # chunking).  The train-mode forward keeps LayerNorm -> GEMM and, since r03, lets a GEMM's tile rows start their K walks at
    # different slices (csrc/gemm.hip, GemmArgs::rot_rows): a row's f32 summation order depends on the tile row it lands in, so a
    # sub-batch agrees with the full batch to accumulation-order accuracy only (as the reference's cuBLAS calls do).
    sub_train = VitPrefixFn.apply(m.visual.tower, x[5:13], vp.clone().requires_grad_(True)).detach()
    cos_t = torch.nn.functional.cosine_similarity(sub_train, first[0][5:13], dim=-1)
    assert (1 - cos_t).max().item() <= 1e-6 and ((sub_train - first[0][5:13]).norm() / first[0][5:13].norm()).item() <= 2e-3
    with torch.no_grad():
        full, sub = m.visual(x, vp), m.visual(x[5:13], vp)
    assert torch.equal(sub, full[5:13])
    cos = torch.nn.functional.cosine_similarity(full, first[0], dim=-1)
    assert (1 - cos).max().item() <= 1e-5        # the two modes agree to f16-operand accuracy


def test_persistent_gemm_epilogue_forms_agree(tmp_path):
    """The epilogue forms of the persistent pool-encode GEMM (GRIP_GEMM_EMODE: 8-byte stores through the LDS slab, 16-byte
    stores through the slab, direct with permuted W fragment rows, fold arithmetic in the fragment layout + f16 slab;
    csrc/gemm.hip) are the same arithmetic in different lane layouts: a ViT-B/16 encode large enough for the persistent kernel
    (>= 512 tiles per GEMM) gives the same embeddings in every mode -- identical up to the order of the f32 row-statistics sums
    and the compiler's contraction choices (1 - cos <= 1e-6, the margin of the f16 engine itself); the c_fc direct form is
    bit-identical to the slab form."""
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = (
        "import sys, torch; sys.path.insert(0, %r); import grip_amd; from grip_amd import clip\n"
        "m, _ = clip.load('ViT-B/16', device='cuda')\n"
        "g = torch.Generator(device='cuda').manual_seed(3)\n"
        "x = torch.randn(704, 3, 224, 224, device='cuda', generator=g)\n"
        "with torch.no_grad(): e = m.encode_image(x)\n"
        "torch.save(e.float().cpu(), sys.argv[1])\n" % repo)
    out, procs = {}, {}
    for mode in ("0", "1", "2", "4", "121", "421"):      # (the six processes share the GPU: most of a run is interpreter + model start-up)
        f = tmp_path / f"emb_{mode}.pt"
        env = dict(os.environ, GRIP_GEMM_EMODE=mode)
        procs[mode] = (subprocess.Popen([sys.executable, "-c", script, str(f)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True), f)
    for mode, (pr, f) in procs.items():
        _, err = pr.communicate(timeout=600)
        assert pr.returncode == 0, err[-2000:]
        out[mode] = torch.load(f)
    ref = out["0"]
    assert torch.isfinite(ref).all() and ref.abs().max() > 0
    for mode in ("1", "2", "4", "121", "421"):
        cos = torch.nn.functional.cosine_similarity(out[mode].double(), ref.double(), dim=1)
        assert (1 - cos).max().item() <= 1e-6, (mode, (1 - cos).max().item())
    # only the residual epilogue's statistics differ between modes 1 and 121 (c_fc direct vs slab is the same f32 arithmetic per element)
    assert torch.equal(out["121"], out["1"])


def test_pool_encode_is_bitwise_independent_of_the_chunking():
    """ViT-B/16 pool encode under chunkings that put the same rows into launches of very different sizes -- served by different
    GEMM tile shapes (persistent 256x256x64 above ~220 images per launch, 128x128 / ring kernels below): every kernel of gemm.hip
    sums a row's K slices in the same rotated order and the row statistics in the same tree, so the embeddings are bit-identical
    (what makes a sharded pseudolabel pass reproduce the single-GPU one whatever the shard sizes are)."""
    import grip_amd  # noqa: F401
    from grip_amd import clip
    m, _ = clip.load("ViT-B/16", device="cuda")
    g = torch.Generator(device="cuda").manual_seed(11)
    x = torch.randn(900, 3, 224, 224, device="cuda", generator=g)
    tower = m.visual.tower
    outs = []
    # 700 -> a 200-image tail, 333 -> a 234-image tail; 40 and 8: launches small enough for the loader-wave ring kernel (gemm_ringw: a launch of at
    # most one workgroup per CU) and the 192-row form, on either side of their thresholds
    for chunk in (900, 450, 700, 333, 40, 8):
        o = torch.empty(900, 512, device="cuda")
        with torch.no_grad():
            tower.encode_chunks(x, o, 0, 900, chunk, streams=1)
        torch.cuda.synchronize()
        outs.append(o.clone())
    for o in outs[1:]:
        assert torch.equal(outs[0], o)


def test_overlapped_lookahead_encode_is_bitwise_the_sequential_one():
    """r05: steps.lookahead_image_features encodes group g + 1 on a CU-masked side stream (three quarters of the chip, persistent grids sized by
    grip_set_cu_budget) while the steps of group g run.  The launch width changes which workgroup computes a tile, never an element's arithmetic:
    overlapped == sequential == one encode per batch, bit for bit -- with concurrent work on the main stream in between, as in a real epoch."""
    import grip_amd  # noqa: F401
    from grip_amd import clip, engine, steps
    m, _ = clip.load("ViT-B/16", device="cuda")
    g = torch.Generator(device="cuda").manual_seed(3)
    batches = [(torch.randn(16, 3, 224, 224, device="cuda", generator=g), i) for i in range(28)]      # 28 batches: groups of 12, 12 and a ragged 4
    with torch.no_grad():
        want = [m.encode_image(x) for x, _ in batches]
    junk = torch.randn(2048, 2048, device="cuda")
    for overlap in (3, 2, 0):
        got = []
        for f, i in steps.lookahead_image_features(m, iter(batches), 12, overlap=overlap):
            for _ in range(4):
                junk = junk @ junk * 1e-3              # the "prompt step" of this batch: main-stream work beside the side stream's encode
            got.append((f.clone(), i))
        torch.cuda.synchronize()
        assert [i for _, i in got] == list(range(28))
        for (f, _), w in zip(got, want):
            assert torch.equal(f, w), overlap
    if engine.masked_stream("cuda", 3) is None:
        pytest.skip("hipExtStreamCreateWithCUMask refused: only the sequential form ran")
