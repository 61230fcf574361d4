"""Backward (input-gradient) parity on the GPU: kernel level against torch autograd in fp32, tower
level against the committed golden gradients (autograd through the reference's own wrappers over
the CPU oracle, oracle/gen_golden.py).  Tolerance for prompt gradients: cosine >= 0.999 and
relative L2 error <= 3e-2 (f16 gradient operands with a dynamic power-of-two loss scale)."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
SEED = 100


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _inputs(name, shape, std=1.0):
    import grip_amd  # noqa: F401
    from grip_amd import rng
    return torch.from_numpy(rng.normal(SEED, rng.stream_id(name), shape, 0.0, std))


def assert_grad_close(got, want, what, cos_tol=1e-3, rel_tol=3e-2):
    got = got.detach().float().cpu().reshape(-1)
    want = torch.as_tensor(np.asarray(want)).float().reshape(-1)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    assert torch.isfinite(got).all(), f"{what}: non-finite gradient"
    cos = torch.nn.functional.cosine_similarity(got, want, dim=0).item()
    rel = ((got - want).norm() / want.norm()).item()
    assert 1 - cos <= cos_tol, f"{what}: 1-cos = {1 - cos:.3e} (rel {rel:.3e})"
    assert rel <= rel_tol, f"{what}: relative L2 error {rel:.3e}"


@pytest.mark.parametrize("B,S,H,causal", [(2, 17, 2, 0), (2, 40, 2, 1), (2, 77, 8, 1), (2, 197, 12, 0), (1, 213, 12, 0), (1, 250, 3, 0), (1, 273, 4, 0), (2, 120, 2, 0), (1, 150, 2, 1),
                                          # S > 288: the block-tiled kernel (attention_bwd_tiled.hip); 577 / 581 / 593 = ViT-L/14@336px without / with 4 / 16 prompt tokens
                                          (1, 289, 2, 0), (2, 300, 3, 0), (1, 448, 2, 0), (1, 449, 1, 0), (2, 577, 16, 0), (1, 581, 4, 0), (1, 593, 16, 0), (1, 608, 2, 0),
                                          (1, 320, 2, 1), (1, 500, 1, 1)])
def test_attention_backward(B, S, H, causal):
    import grip_amd  # noqa: F401
    from grip_amd import native
    lib = native.lib()
    g = torch.Generator(device="cuda").manual_seed(S + 1)
    D = H * 64
    qkv = (torch.randn(B * S, 3 * D, device="cuda", generator=g)).half()
    dout = (torch.randn(B * S, D, device="cuda", generator=g)).half()
    out = torch.zeros(B * S, D, device="cuda", dtype=torch.float16)
    native.check(lib.grip_debug_attention(_p(qkv), _p(out), B, S, H, causal, _stream()))
    dqkv = torch.full((B * S, 3 * D), float("nan"), device="cuda", dtype=torch.float16)
    native.check(lib.grip_debug_attention_bwd(_p(qkv), _p(out), _p(dout), _p(dqkv), B, S, H, causal, _stream()))

    x = qkv.float().clone().requires_grad_(True)
    q, k, v = x.reshape(B, S, 3, H, 64).permute(2, 0, 3, 1, 4)
    s = (q * 0.125) @ k.transpose(-1, -2)
    if causal:
        s = s + torch.full((S, S), float("-inf"), device="cuda").triu(1)
    ref = (s.softmax(-1) @ v).permute(0, 2, 1, 3).reshape(B * S, D)
    ref.backward(dout.float())
    want = x.grad.reshape(B * S, 3, D)
    got = dqkv.float().reshape(B * S, 3, D)
    for i, nm in enumerate("qkv"):
        assert_grad_close(got[:, i], want[:, i].cpu(), f"d{nm}", cos_tol=1e-4, rel_tol=1e-2)


def test_cosine_head_and_ce_backward():
    import grip_amd  # noqa: F401
    from grip_amd.engine import CosineHeadFn, WeightedCEFn
    g = torch.Generator(device="cuda").manual_seed(3)
    img = torch.randn(16, 512, device="cuda", generator=g, requires_grad=True)
    txt = torch.randn(47, 512, device="cuda", generator=g, requires_grad=True)
    labels = torch.randint(0, 47, (16,), device="cuda", generator=g)
    w = torch.rand(16, device="cuda", generator=g)
    w[3] = 0.0
    loss = WeightedCEFn.apply(CosineHeadFn.apply(img, txt, 100.0), labels, w)
    loss.backward()
    gi, gt = img.grad.clone(), txt.grad.clone()
    img.grad = txt.grad = None
    i = img / img.norm(dim=-1, keepdim=True)
    t = txt / txt.norm(dim=-1, keepdim=True)
    ref = (torch.nn.functional.cross_entropy(100.0 * i @ t.t(), labels, reduction="none") * w).sum()
    ref.backward()
    torch.testing.assert_close(loss, ref, rtol=1e-4, atol=1e-4)
    assert_grad_close(gi, img.grad.cpu(), "d img", cos_tol=1e-5, rel_tol=1e-3)
    assert_grad_close(gt, txt.grad.cpu(), "d txt", cos_tol=1e-5, rel_tol=1e-3)


def test_weighted_ce_kernel_reproduces_the_three_reference_fpl_losses(golden_small):
    """grip_weighted_ce on the reference-generated G7 block: g7.{ssl, trzsl, ul} are the FPL losses the reference's own
    define_loss_function bodies returned on g7.logits (semi_supervised_learning/textual_fpl.py:123-165, transductive_zsl/
    textual_fpl.py:117-147, unsupervised_learning/visual_fpl.py:107-122; oracle/gen_golden.py); the kernel with the three
    fpl_row_weights variants must return them, and its gradient must be the autograd gradient of the same expression."""
    import numpy as np

    import grip_amd  # noqa: F401
    from grip_amd.engine import WeightedCEFn
    from grip_amd.steps import fpl_row_weights
    g = golden_small
    labels = torch.tensor([0, 3, 1, 4, 2, 3])
    unl = [True, False, True, True, False, True]
    variants = {"ssl": fpl_row_weights(unl, gamma_seen=4 / 2, gamma_pseudo=1.0),
                "trzsl": fpl_row_weights([int(l) in (3, 4) for l in labels], gamma_seen=1.0, gamma_pseudo=3 / 3),
                "ul": fpl_row_weights([False] * 6)}
    for name, w in variants.items():
        logits = torch.from_numpy(g["g7.logits"]).cuda().requires_grad_(True)
        loss = WeightedCEFn.apply(logits, labels.cuda(), w.cuda())
        loss.backward()
        np.testing.assert_allclose(loss.item(), float(g[f"g7.{name}"]), rtol=2e-6, err_msg=name)
        ref_in = torch.from_numpy(g["g7.logits"]).requires_grad_(True)
        (torch.nn.functional.cross_entropy(ref_in, labels, reduction="none") * w).sum().backward()
        torch.testing.assert_close(logits.grad.cpu(), ref_in.grad, rtol=1e-5, atol=1e-7)


@pytest.fixture(scope="module")
def models():
    import grip_amd  # noqa: F401
    from grip_amd import clip
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = clip.load(name, device="cuda")[0]
        return cache[name]
    return get


@pytest.mark.parametrize("tag,name,n_img,P", [("g1", "tiny", 3, 3), ("g1s", "small", 2, 16)])
def test_golden_prompt_gradients(models, golden_small, tag, name, n_img, P):
    """grad of sum(out^2) w.r.t. the visual / textual prompt, as the reference wrappers produce it."""
    import grip_amd  # noqa: F401
    from grip_amd import config
    from grip_amd.models import CustomImageEncoder, CustomTextEncoder, ImagePrefixModel
    m = models(name)
    d = config.get_dims(name)
    x = _inputs(f"{tag}.x", (n_img, 3, d.image_resolution, d.image_resolution)).cuda()
    vprefix = _inputs(f"{tag}.vprefix", (P, d.vision_width), 0.02).cuda()
    model = ImagePrefixModel(vprefix.clone(), CustomImageEncoder(m.visual), device="cuda")
    out = model(x)
    (out ** 2).sum().backward()
    assert_grad_close(model.prefix.grad, golden_small[f"{tag}.vision_p{P}_grad_prefix"], "visual prompt grad")
    for prm in m.parameters():
        assert prm.grad is None and not prm.requires_grad      # frozen backbone

    tprefix = _inputs(f"{tag}.tprefix", (1, P, d.transformer_width), 0.02).cuda().requires_grad_(True)
    from grip_amd.engine import TextPrefixFn
    ctok = torch.from_numpy(golden_small[f"{tag}.coop_tokens"]).cuda()
    tout = TextPrefixFn.apply(m.text_tower, ctok, tprefix)
    (tout ** 2).sum().backward()
    assert_grad_close(tprefix.grad, golden_small[f"{tag}.text_p{P}_grad_prefix"], "textual prompt grad")


def test_golden_upt_end_to_end(models, golden_small):
    """G4: UPTModel forward (mixer with the f16 round trip, both towers), CE loss, all gradients."""
    import grip_amd  # noqa: F401
    from grip_amd import config, weights
    from grip_amd.engine import CosineHeadFn, WeightedCEFn
    from grip_amd.models import CustomImageEncoder, CustomTextEncoder, UPTModel
    g = golden_small
    m = models("tiny")
    d = config.get_dims("tiny")
    classes = ["forest", "annual crop land", "river"]
    x = _inputs("g4.x", (3, 3, d.image_resolution, d.image_resolution)).cuda()
    coop = _inputs("g4.coop", (1, 4, d.transformer_width), 0.02).cuda()
    vpt = _inputs("g4.vpt", (1, 4, d.vision_width), 0.02).cuda()
    upt = UPTModel(coop, vpt, None, CustomImageEncoder(m.visual), CustomTextEncoder(m, "cuda", torch.float32), classes, 128,
                   device="cuda", dtype=torch.float32)
    mixer = {k: torch.from_numpy(v) for k, v in weights.init_upt_mixer(d.transformer_width, d.vision_width, 128, SEED).items()}
    missing, unexpected = upt.load_state_dict(mixer, strict=False)
    assert not unexpected
    ce, ve = upt.mix()
    torch.testing.assert_close(ce.cpu(), torch.from_numpy(g["g4.mixer_coop"]), rtol=2e-3, atol=2e-3)
    torch.testing.assert_close(ve.cpu(), torch.from_numpy(g["g4.mixer_vpt"]), rtol=2e-3, atol=2e-3)
    t_out, v_out = upt(x, classes)
    from test_gpu_towers import assert_embeddings_close
    assert_embeddings_close(t_out, g["g4.text"], "upt text")
    assert_embeddings_close(v_out, g["g4.vision"], "upt vision")
    logits = CosineHeadFn.apply(v_out, t_out, m.logit_scale.exp().item())
    labels = torch.arange(3, device="cuda") % 3
    loss = WeightedCEFn.apply(logits, labels, torch.full((3,), 1 / 3, device="cuda"))
    assert abs(loss.item() - float(g["g4.loss"])) <= 2e-2 * max(1.0, abs(float(g["g4.loss"])))
    loss.backward()
    for name, p in upt.named_parameters():
        key = f"g4.grad.{name}"
        if key in g.files:
            assert p.grad is not None, name
            assert_grad_close(p.grad, g[key], name, cos_tol=5e-3, rel_tol=8e-2)


def test_vitb16_gradient_vs_oracle(models):
    """Full-size VPT step shape (B=4, P=16) against CPU-oracle autograd computed here."""
    from conftest import oracle_clip
    from oracle import wrappers as W
    import grip_amd  # noqa: F401
    from grip_amd.engine import VitPrefixFn
    om, _ = oracle_clip().load("ViT-B/16")
    m = models("ViT-B/16")
    x = _inputs("b16g.x", (4, 3, 224, 224))
    prefix = _inputs("b16g.p", (16, 768), 0.02)
    w = _inputs("b16g.w", (4, 512))
    pc = prefix.clone().requires_grad_(True)
    (W.vision_forward(om.visual, x, pc) * w).sum().backward()
    pg = prefix.clone().cuda().requires_grad_(True)
    (VitPrefixFn.apply(m.visual.tower, x.cuda(), pg) * w.cuda()).sum().backward()
    assert_grad_close(pg.grad, pc.grad, "ViT-B/16 visual prompt grad")


def _full_size_text_gradient(m, g, tag, dt, cos_tol=1e-3, rel_tol=3e-2):
    """CoOp prompt gradient through the whole text tower against the committed reference gradient, on the path bench.py
    times (EOT-truncated: the tower encodes only positions <= the longest EOT) AND with all 77 positions."""
    from grip_amd.engine import TextPrefixFn
    ctok = torch.from_numpy(g[f"{tag}.coop_tokens"]).cuda()
    want = g[f"{tag}.text_p16_grad_prefix"]
    tower = m.text_tower
    for truncate in (True, False):
        tower.truncate_text_at_eot = truncate
        try:
            tok = ctok.clone()          # a fresh tensor: the cached sequence length lives on the token tensor
            tprefix = _inputs(f"{tag}.tprefix", (1, 16, dt), 0.02).cuda().requires_grad_(True)
            out = TextPrefixFn.apply(tower, tok, tprefix)
            assert (0 < tok._grip_seq_len < 77) == truncate
            (out ** 2).sum().backward()
            assert_grad_close(tprefix.grad, want, f"{tag} textual prompt grad (truncate={truncate})", cos_tol, rel_tol)
        finally:
            tower.truncate_text_at_eot = True


@pytest.mark.parametrize("name,C,P,lens", [("small", 5, 3, (2, 1, 4, 3, 1)), ("ViT-B/16", 102, 16, None), ("ViT-B/16", 2, 4, (1, 6)), ("ViT-B/16", 37, 16, "ragged")])
def test_shared_prefix_layout_equals_plain(models, name, C, P, lens):
    """One shared context (CoOp): positions 0 .. P are the same tokens for every class under a causal mask, so the engine encodes
    them once (GRIP_FWD_SHARED_PREFIX: 1 + P + C * (S - 1 - P) rows instead of C * S).  Same embeddings (train and inference
    forward) and same prompt gradient as the plain C x S layout, ragged class-name lengths included; class-specific contexts, a
    single class, or tokens that differ inside the shared positions keep the plain layout."""
    import grip_amd  # noqa: F401
    from grip_amd import config, native
    from grip_amd.engine import TextPrefixFn, text_prefix_forward
    m = models(name)
    d = config.get_dims(name)
    tower = m.text_tower
    gen = torch.Generator().manual_seed(C * 31 + P)
    if lens is None:
        lens = [3] * C
    elif lens == "ragged":
        lens = [1 + int(v) for v in torch.randint(0, 9, (C,), generator=gen)]
    sot, eot = d.vocab_size - 2, d.vocab_size - 1
    tok = torch.zeros(C, d.context_length, dtype=torch.int32)
    for c, n in enumerate(lens):
        row = [sot] + [7] * P + [int(v) for v in torch.randint(10, d.vocab_size - 2, (n,), generator=gen)] + [eot]
        tok[c, :len(row)] = torch.tensor(row, dtype=torch.int32)
    prefix0 = _inputs(f"shared.{name}.{C}", (1, P, d.transformer_width), 0.02).cuda()
    wout = _inputs(f"shared.w.{name}.{C}", (C, d.embed_dim)).cuda()
    res = {}
    for share in (True, False):
        tower.share_text_prefix = share
        try:
            t = tok.clone().cuda()
            pf = prefix0.clone().requires_grad_(True)
            out = TextPrefixFn.apply(tower, t, pf)
            assert bool(tower.last_text_flags & native.FWD_SHARED_PREFIX) == share
            (out * wout).sum().backward()
            with torch.no_grad():
                inf = text_prefix_forward(tower, t, pf)
            res[share] = (out.detach().clone(), inf.clone(), pf.grad.clone())
        finally:
            tower.share_text_prefix = True
    for i, what in enumerate(("train-mode embeddings", "inference embeddings")):
        a, b = res[True][i], res[False][i]
        cos = torch.nn.functional.cosine_similarity(a, b, dim=1)
        assert (1 - cos).max().item() <= 2e-6, f"{what}: 1-cos {(1 - cos).max().item():.2e}"
        torch.testing.assert_close(a, b, rtol=2e-3, atol=2e-3 * b.abs().max().item())
    assert_grad_close(res[True][2], res[False][2].cpu().numpy(), "prompt gradient, shared vs plain layout", cos_tol=1e-4, rel_tol=1e-2)
    # not shareable: per-class contexts, one class, tokens differing inside the shared positions
    t = tok.clone().cuda()
    TextPrefixFn.apply(tower, t, prefix0.expand(C, P, -1).contiguous().requires_grad_(True))
    assert not tower.last_text_flags & native.FWD_SHARED_PREFIX
    TextPrefixFn.apply(tower, tok[:1].clone().cuda(), prefix0.clone().requires_grad_(True))
    assert not tower.last_text_flags & native.FWD_SHARED_PREFIX
    t2 = tok.clone()
    t2[C - 1, P] = 9
    TextPrefixFn.apply(tower, t2.cuda(), prefix0.clone().requires_grad_(True))
    assert not tower.last_text_flags & native.FWD_SHARED_PREFIX


def test_golden_vitb16_prompt_gradients(models, golden_vitb16):
    """G3 with gradients: the CoOp step bench.py times -- d = 512, 12 layers, P = 16, EOT-truncated --
    and the VPT prompt gradient through the 12-layer ViT-B/16, against autograd through the reference's own wrappers."""
    import grip_amd  # noqa: F401
    from grip_amd.models import CustomImageEncoder, ImagePrefixModel
    m, g = models("ViT-B/16"), golden_vitb16
    _full_size_text_gradient(m, g, "g3", 512)
    x = _inputs("g3.x", (2, 3, 224, 224)).cuda()
    model = ImagePrefixModel(_inputs("g3.vprefix", (16, 768), 0.02).cuda(), CustomImageEncoder(m.visual), device="cuda")
    (model(x) ** 2).sum().backward()
    assert_grad_close(model.prefix.grad, g["g3.vision_p16_grad_prefix"], "g3 visual prompt grad")


def test_golden_vitb32_prompt_gradients(models, golden_vitb32):
    """ViT-B/32 (the reference scripts' default encoder): CoOp gradient through its text tower and the VPT gradient through the
    12-layer ViT at S = 66, against autograd through the reference's own wrappers."""
    import grip_amd  # noqa: F401
    from grip_amd.models import CustomImageEncoder, ImagePrefixModel
    m, g = models("ViT-B/32"), golden_vitb32
    _full_size_text_gradient(m, g, "g6", 512)
    x = _inputs("g6.x", (2, 3, 224, 224)).cuda()
    model = ImagePrefixModel(_inputs("g6.vprefix", (16, 768), 0.02).cuda(), CustomImageEncoder(m.visual), device="cuda")
    (model(x) ** 2).sum().backward()
    assert_grad_close(model.prefix.grad, g["g6.vision_p16_grad_prefix"], "g6 visual prompt grad")


def test_golden_vitb16_upt_end_to_end(models, golden_vitb16):
    """UPT at ViT-B/16 dimensions, Pt = Pv = 4 (BASELINE.json configs[3]): reference UPTModel output, loss and every gradient."""
    import grip_amd  # noqa: F401
    from grip_amd import weights
    from grip_amd.engine import CosineHeadFn, WeightedCEFn
    from grip_amd.models import CustomImageEncoder, CustomTextEncoder, UPTModel
    from test_gpu_towers import assert_embeddings_close
    g, m = golden_vitb16, models("ViT-B/16")
    classes = ["forest", "annual crop land", "river"]
    x = _inputs("g4b.x", (2, 3, 224, 224)).cuda()
    upt = UPTModel(_inputs("g4b.coop", (1, 4, 512), 0.02).cuda(), _inputs("g4b.vpt", (1, 4, 768), 0.02).cuda(), None, CustomImageEncoder(m.visual),
                   CustomTextEncoder(m, "cuda", torch.float32), classes, 128, device="cuda", dtype=torch.float32)
    mixer = {k: torch.from_numpy(v) for k, v in weights.init_upt_mixer(512, 768, 128, SEED).items()}
    assert not upt.load_state_dict(mixer, strict=False)[1]
    t_out, v_out = upt(x, classes)
    assert_embeddings_close(t_out, g["g4b.text"], "B/16 upt text")
    assert_embeddings_close(v_out, g["g4b.vision"], "B/16 upt vision")
    logits = CosineHeadFn.apply(v_out, t_out, m.logit_scale.exp().item())
    loss = WeightedCEFn.apply(logits, torch.arange(2, device="cuda") % 3, torch.full((2,), 0.5, device="cuda"))
    assert abs(loss.item() - float(g["g4b.loss"])) <= 2e-2 * max(1.0, abs(float(g["g4b.loss"])))
    loss.backward()
    seen = 0
    for name, p in upt.named_parameters():
        if f"g4b.grad.{name}" in g.files:
            assert p.grad is not None, name
            assert_grad_close(p.grad, g[f"g4b.grad.{name}"], name, cos_tol=5e-3, rel_tol=8e-2)
            seen += 1
    assert seen >= 10


def test_golden_vitl14_336_text_gradient(models, golden_vitl14):
    """BASELINE.json configs[4] (FGVCAircraft GRIP textual, ViT-L/14@336px): CoOp prompt gradient through the 12-head, 768-wide
    text tower at its real dimensions."""
    _full_size_text_gradient(models("ViT-L/14@336px"), golden_vitl14, "g5", 768)


def test_golden_vitl14_336_visual_prompt_gradient(models, golden_vitl14):
    """VERDICT r1 missing #3: a visual prompt on ViT-L/14@336px (S = 593 > 288: the block-tiled attention backward) -- gradient
    of sum(out^2) w.r.t. the 16 prompt tokens through all 24 layers against autograd through the reference's wrappers."""
    import grip_amd  # noqa: F401
    from grip_amd.models import CustomImageEncoder, ImagePrefixModel
    m, g = models("ViT-L/14@336px"), golden_vitl14
    x = _inputs("g5.x", (2, 3, 336, 336)).cuda()
    model = ImagePrefixModel(_inputs("g5.vprefix", (16, 1024), 0.02).cuda(), CustomImageEncoder(m.visual), device="cuda")
    out = model(x)
    from test_gpu_towers import assert_embeddings_close
    assert_embeddings_close(out, g["g5.vision_p16"], "L/14@336 vision+prefix (train-mode forward)")
    (out ** 2).sum().backward()
    assert_grad_close(model.prefix.grad, g["g5.vision_p16_grad_prefix"], "L/14@336 visual prompt grad")
