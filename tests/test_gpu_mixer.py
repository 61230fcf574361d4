"""UPTModel's prompt mixer (reference models/prompts_models.py:99-146) on the native kernels (csrc/mixer.hip, C ABI
grip_upt_mixer_forward / _backward): outputs and the gradients of all 22 tensors against a float64 autograd restatement of the
reference's own module graph (nn.Linear x 4, clip.model.Transformer(width, 1, 1), the .to(float16) round trip), and the
reference's float16 branch (methods/semi_supervised_learning/multimodal_prompt.py:46: dtype = float16 whenever a GPU is present)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _model(P, dt, dv, D, dtype, seed=0):
    import grip_amd  # noqa: F401
    from grip_amd.models import UPTModel
    torch.manual_seed(seed)
    g = torch.Generator().manual_seed(seed + 1)
    coop = (torch.randn(1, P, dt, generator=g) * 0.02).to(dtype).cuda()
    vpt = (torch.randn(1, P, dv, generator=g) * 0.02).to(dtype).cuda()
    m = UPTModel(coop, vpt, None, None, None, ["a"], D, device="cuda", dtype=dtype)
    with torch.no_grad():     # LayerNorm affine and biases away from their init so that their gradients are exercised
        for n, p in m.named_parameters():
            if n.endswith("bias") or "ln_" in n:
                p.add_((torch.randn(p.shape, generator=g) * 0.1).to(p.dtype).cuda())
    return m


def _reference_mix(m, dd=torch.float64):
    """The reference's mix in float64 on the CPU (same module graph, QuickGELU written out; the fp16 round trip with autograd's
    fp16 gradient)."""
    sd = {k: v.detach().cpu().to(dd).requires_grad_(True) for k, v in m.named_parameters()}
    coop, vpt = sd["coop_embeddings"], sd["vpt_embeddings"]
    lin = lambda x, w, b: x @ sd[w].t() + sd[b]      # noqa: E731
    x = torch.cat((lin(coop, "proj_coop_pre.weight", "proj_coop_pre.bias"), lin(vpt, "proj_vpt_pre.weight", "proj_vpt_pre.bias")), dim=0)   # [2, P, D]
    D = x.shape[-1]
    pre = "transformer.resblocks.0."
    ln = lambda t, w, b: torch.nn.functional.layer_norm(t, (D,), sd[pre + w], sd[pre + b], 1e-5)      # noqa: E731
    y = ln(x, "ln_1.weight", "ln_1.bias")
    q, k, v = lin(y, pre + "attn.in_proj_weight", pre + "attn.in_proj_bias").chunk(3, dim=-1)
    att = torch.einsum("lnd,mnd->nlm", q, k) / D ** 0.5
    o = torch.einsum("nlm,mnd->lnd", att.softmax(-1), v)
    x = x + lin(o, pre + "attn.out_proj.weight", pre + "attn.out_proj.bias")
    h = lin(ln(x, "ln_2.weight", "ln_2.bias"), pre + "mlp.c_fc.weight", pre + "mlp.c_fc.bias")
    x = x + lin(h * torch.sigmoid(1.702 * h), pre + "mlp.c_proj.weight", pre + "mlp.c_proj.bias")
    out = x.to(torch.float16).to(dd)                 # :138-145 (the backward rounds the gradient to fp16 on its way through)
    n = len(coop)
    return lin(out[:n], "proj_coop_post.weight", "proj_coop_post.bias"), lin(out[n:], "proj_vpt_post.weight", "proj_vpt_post.bias"), sd


@pytest.mark.parametrize("P,dt,dv,D", [(4, 512, 768, 128), (4, 128, 128, 128), (16, 768, 1024, 128), (1, 512, 768, 64), (3, 192, 320, 256)])
def test_native_mixer_forward_and_all_gradients(P, dt, dv, D):
    m = _model(P, dt, dv, D, torch.float32)
    assert m._native_mixer_ok()
    ce, ve = m.mix()
    assert ce.shape == (1, P, dt) and ve.shape == (1, P, dv)
    g = torch.Generator().manual_seed(5)
    wc, wv = torch.randn(1, P, dt, generator=g), torch.randn(1, P, dv, generator=g)
    ((ce * wc.cuda()).sum() + (ve * wv.cuda()).sum()).backward()
    rc, rv, sd = _reference_mix(m)
    ((rc * wc.double()).sum() + (rv * wv.double()).sum()).backward()
    # forward: fp32 arithmetic vs float64, through one fp16 rounding (a value next to a rounding boundary may land one fp16 ulp away)
    torch.testing.assert_close(ce.detach().cpu().double(), rc.detach(), rtol=2e-3, atol=2e-3 * rc.detach().abs().max().item())
    torch.testing.assert_close(ve.detach().cpu().double(), rv.detach(), rtol=2e-3, atol=2e-3 * rv.detach().abs().max().item())
    seen = 0
    for name, p in m.named_parameters():
        assert p.grad is not None, name
        want = sd[name].grad
        got = p.grad.detach().cpu().double()
        cos = torch.nn.functional.cosine_similarity(got.reshape(-1), want.reshape(-1), dim=0).item()
        rel = ((got - want).norm() / want.norm().clamp_min(1e-30)).item()
        assert cos >= 1 - 1e-4 and rel <= 1e-2, f"{name}: cos {cos:.6f} rel {rel:.2e}"
        seen += 1
    assert seen == 22


def test_native_mixer_equals_the_framework_path_and_is_reproducible(monkeypatch):
    """GRIP_NATIVE_MIXER=0 runs the same module through ATen / rocBLAS: same outputs and gradients to fp32 accuracy; the native
    path itself is bit-reproducible (no atomics)."""
    m = _model(4, 512, 768, 128, torch.float32, seed=3)
    outs = []
    for native_on in ("1", "1", "0"):
        monkeypatch.setenv("GRIP_NATIVE_MIXER", native_on)
        m.zero_grad(set_to_none=True)
        ce, ve = m.mix()
        (ce.square().sum() + ve.square().sum()).backward()
        outs.append((ce.detach().clone(), ve.detach().clone(), {n: p.grad.clone() for n, p in m.named_parameters()}))
    a, b, t = outs
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and all(torch.equal(a[2][n], b[2][n]) for n in a[2])
    torch.testing.assert_close(a[0], t[0], rtol=2e-3, atol=2e-3 * t[0].abs().max().item())
    for n in a[2]:
        cos = torch.nn.functional.cosine_similarity(a[2][n].reshape(-1), t[2][n].reshape(-1), dim=0).item()
        assert cos >= 1 - 1e-3, (n, cos)


def test_mixer_argument_checks():
    import ctypes

    import grip_amd  # noqa: F401
    from grip_amd import native
    lib = native.lib()
    n = ctypes.c_size_t()
    assert lib.grip_upt_mixer_workspace(4, 512, 768, 128, ctypes.byref(n)) == 0 and n.value > 0
    assert lib.grip_upt_mixer_workspace(40, 512, 768, 128, ctypes.byref(n)) == 1 and b"n_prompt" in lib.grip_last_error()
    assert lib.grip_upt_mixer_workspace(4, 512, 768, 100, ctypes.byref(n)) == 1 and b"dim" in lib.grip_last_error()
    m = _model(4, 512, 768, 128, torch.float32)
    m.vpt_embeddings = torch.nn.Parameter(torch.zeros(1, 3, 768, device="cuda"))
    m.vpt_length = 3
    assert not m._native_mixer_ok()        # unequal prompt counts: the reference's cat fails too; never reaches the kernels


def test_upt_model_float16_branch(monkeypatch):
    """multimodal_prompt.py:46 makes the UPT dtype float16 whenever torch.cuda.is_available() (true on ROCm): float16 prompt
    embeddings and projection layers around the float32 transformer, float16 prompts into both towers, gradients back in
    float16.  Against the float32 model with the same (fp16-representable) parameters."""
    import grip_amd  # noqa: F401
    from grip_amd import clip
    from grip_amd.engine import CosineHeadFn, WeightedCEFn
    from grip_amd.models import CustomImageEncoder, CustomTextEncoder, UPTModel
    cm, _ = clip.load("small", device="cuda")
    classes = ["forest", "river", "sea lake", "highway"]
    g = torch.Generator().manual_seed(2)
    x = torch.randn(6, 3, 64, 64, generator=g).cuda()
    coop = (torch.randn(1, 4, 256, generator=g) * 0.02).half()
    vpt = (torch.randn(1, 4, 256, generator=g) * 0.02).half()
    res = {}
    trainable = lambda mod: [(n, p) for n, p in mod.named_parameters() if p.requires_grad]      # noqa: E731  (the frozen towers hang off the model too)
    for key, dtype, native_mixer in (("f16", torch.float16, "1"), ("f16_framework", torch.float16, "0"), ("f32", torch.float32, "1")):
        monkeypatch.setenv("GRIP_NATIVE_MIXER", native_mixer)
        torch.manual_seed(11)
        um = UPTModel(coop.to(dtype).cuda(), vpt.to(dtype).cuda(), None, CustomImageEncoder(cm.visual), CustomTextEncoder(cm, "cuda", dtype), classes, 128,
                      device="cuda", dtype=dtype)
        if key != "f16":      # the same parameter values as the float16 model
            with torch.no_grad():
                for (n, p), (_, q) in zip(trainable(um), res["f16_params"]):
                    p.copy_(q.to(p.dtype))
        else:
            res["f16_params"] = [(n, p.detach().clone()) for n, p in trainable(um)]
            assert um.proj_coop_pre.weight.dtype == torch.float16 and um.transformer.resblocks[0].ln_1.weight.dtype == torch.float32
        # r04: the float16 branch runs on csrc/mixer.hip too (grip_upt_mixer.half_linears); GRIP_NATIVE_MIXER=0 = the framework's kernels
        assert um._native_mixer_ok() == (native_mixer == "1")
        t_out, v_out = um(x, classes)
        assert t_out.dtype == torch.float32 and v_out.shape == (6, cm.visual.output_dim)
        logits = CosineHeadFn.apply(v_out, t_out, 100.0)
        loss = WeightedCEFn.apply(logits, torch.arange(6, device="cuda") % 4, torch.full((6,), 1 / 6, device="cuda"))
        loss.backward()
        res[key] = (t_out.detach(), v_out.detach(), loss.item(), {n: p.grad.detach().float().clone() for n, p in trainable(um)})
        assert len(res[key][3]) == 22
        for n, p in trainable(um):
            assert p.grad is not None and p.grad.dtype == p.dtype and torch.isfinite(p.grad).all(), n
    cos = torch.nn.functional.cosine_similarity
    h, hf, f = res["f16"], res["f16_framework"], res["f32"]
    # native float16 branch against the framework's float16 branch: the same rounding points, different summation orders inside the fp16 Linears
    for a, b in ((h[0], hf[0]), (h[1], hf[1])):
        assert cos(a, b, dim=-1).min().item() >= 1 - 1e-5
    assert abs(h[2] - hf[2]) <= 2e-3 * max(1.0, abs(hf[2]))
    for n in hf[3]:
        if hf[3][n].norm().item() > 1e-6:
            c = cos(h[3][n].reshape(-1), hf[3][n].reshape(-1), dim=0).item()
            assert c >= 0.995, (n, c)
    # ... and against the float32 model with the same (fp16-representable) parameters
    for a, b in ((h[0], f[0]), (h[1], f[1])):
        assert cos(a, b, dim=-1).min().item() >= 1 - 1e-4
    assert abs(h[2] - f[2]) <= 2e-2 * max(1.0, abs(f[2]))
    for n in f[3]:
        if f[3][n].norm().item() > 1e-6:      # fp16 gradients of tiny magnitude lose bits; compare direction where there is a signal
            c = cos(h[3][n].reshape(-1), f[3][n].reshape(-1), dim=0).item()
            assert c >= 0.98, (n, c)
