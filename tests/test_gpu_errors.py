"""Error behaviour of the C ABI on the GPU box: every misuse returns a status + message (no crash, no C++
exception across the boundary), mirroring the reference's plain Python exceptions with GripError."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def model():
    import grip_amd  # noqa: F401
    from grip_amd import clip
    return clip.load("tiny", device="cuda")[0]


def test_workspace_too_small_and_misaligned(model):
    import grip_amd  # noqa: F401
    from grip_amd import native
    from grip_amd.engine import _ptr, _stream
    t = model.visual.tower
    x = torch.randn(2, 3, 32, 32, device="cuda")
    out = torch.empty(2, 128, device="cuda")
    ws = torch.empty(4096, dtype=torch.uint8, device="cuda")
    rc = t.lib.grip_vit_forward(t.handle, _ptr(x), 0, None, 0, 2, _ptr(out), ctypes.c_void_p(ws.data_ptr()), 4096, 0, None, _stream())
    assert rc == 3 and b"workspace too small" in t.lib.grip_last_error()
    big = t.workspace(2, 0, False)
    p, n = t._aligned(big)
    rc = t.lib.grip_vit_forward(t.handle, _ptr(x), 0, None, 0, 2, _ptr(out), ctypes.c_void_p(p.value + 8), n - 8, 0, None, _stream())
    assert rc == 1 and b"aligned" in t.lib.grip_last_error()
    with pytest.raises(native.GripError):
        native.check(rc)


def test_backward_without_training_forward(model):
    import grip_amd  # noqa: F401
    from grip_amd import native
    t = model.visual.tower
    prefix = torch.randn(3, 128, device="cuda") * 0.02
    x = torch.randn(2, 3, 32, 32, device="cuda")
    _, ws = t.vit_forward(x, prefix, train=False)
    with pytest.raises(native.GripError, match="train-mode forward"):
        t.vit_backward(torch.ones(2, 128, device="cuda"), prefix, ws)


def test_two_forwards_before_backward(model):
    """two grad-enabled forwards of the same shape before one backward (model(aug_1) + model(aug_2)).
    Through the autograd Functions each forward gets its own workspace and both gradients are right; at the C ABI a
    backward that presents the generation of an overwritten forward fails with GRIP_ERR_STATE instead of returning the
    gradients of the wrong forward."""
    import grip_amd  # noqa: F401
    from grip_amd import native
    from grip_amd.engine import VitPrefixFn
    t = model.visual.tower
    g = torch.Generator(device="cuda").manual_seed(5)
    x1 = torch.randn(2, 3, 32, 32, device="cuda", generator=g)
    x2 = torch.randn(2, 3, 32, 32, device="cuda", generator=g)
    p = (torch.randn(3, 128, device="cuda", generator=g) * 0.02).requires_grad_(True)

    def grad_of(x):
        p.grad = None
        (VitPrefixFn.apply(t, x, p) ** 2).sum().backward()
        return p.grad.clone()
    g1, g2 = grad_of(x1), grad_of(x2)
    p.grad = None
    y1 = VitPrefixFn.apply(t, x1, p)
    y2 = VitPrefixFn.apply(t, x2, p)          # same shape: must not overwrite y1's saved activations
    ((y1 ** 2).sum() + (y2 ** 2).sum()).backward()
    torch.testing.assert_close(p.grad, g1 + g2, rtol=1e-4, atol=1e-7)
    # dropped graphs free their workspace again: the pool does not grow without bound
    for _ in range(6):
        VitPrefixFn.apply(t, x1, p)
    assert len(t._ws[(2, 3, True, 0)]) <= 2
    # the raw ABI: forward #a, forward #b on the SAME workspace, backward(#a) -> GRIP_ERR_STATE
    pd = p.detach()
    _, ws = t.vit_forward(x1, pd, train=True)
    gen_a = ws.generation
    _, ws_b = t.vit_forward(x2, pd, train=True)
    assert ws_b is ws and ws.generation != gen_a
    with pytest.raises(native.GripError, match="overwritten"):
        t.vit_backward(torch.ones(2, 128, device="cuda"), pd, ws, gen_a)
    t.vit_backward(torch.ones(2, 128, device="cuda"), pd, ws, ws.generation)     # the live one still works ...
    with pytest.raises(native.GripError, match="already been back-propagated"):
        t.vit_backward(torch.ones(2, 128, device="cuda"), pd, ws, ws.generation)  # ... exactly once, and the second one says why
    # a workspace a captured HIP graph replays into is never handed to an eager forward (its activations would be overwritten)
    from grip_amd import engine
    with engine.pin_workspaces() as pins:
        _, ws_g = t.vit_forward(x1, pd, train=True)
    _, ws_e = t.vit_forward(x1, pd, train=True)
    assert ws_e is not ws_g and t._busy(ws_g)
    engine.release_pins(pins)
    assert not t._busy(ws_g)


def test_bad_arguments_are_rejected(model):
    import grip_amd  # noqa: F401
    from grip_amd import native
    t = model.visual.tower
    x = torch.randn(2, 3, 32, 32, device="cuda")
    with pytest.raises(native.GripError, match="max_prefix"):
        t.vit_forward(x, torch.zeros(100, 128, device="cuda"))            # more prompt tokens than the tower was built for
    tt = model.text_tower
    ids = torch.zeros(3, 77, dtype=torch.int32)
    ids[:, 0], ids[:, 1:7], ids[:, 7] = 49406, 343, 49407
    with pytest.raises(native.GripError, match="prefix_classes"):
        tt.text_forward(ids.cuda(), torch.zeros(2, 4, 128, device="cuda"))  # prefix for 2 classes, 3 prompts
    with pytest.raises(native.GripError, match="not a vision tower"):
        native.check(tt.lib.grip_vit_forward(tt.handle, None, 0, None, 0, 1, None, None, 0, 0, None, None))
    ws = t.workspace(2, 0, False)
    p, n = t._aligned(ws)
    out = torch.empty(2, t.embed_dim, device="cuda")
    with pytest.raises(native.GripError, match="unknown flag bits"):
        native.check(t.lib.grip_vit_forward(t.handle, x.data_ptr(), 0, None, 0, 2, out.data_ptr(), p, n, 16, None, None))
    with pytest.raises(NotImplementedError):
        from grip_amd.models import CustomImageEncoder
        CustomImageEncoder(model.visual)(x, torch.zeros(2, 128, device="cuda"), deep_embds=torch.zeros(1))


def test_out_of_range_token_ids_do_not_fault(model):
    ids = torch.full((2, 77), 10 ** 6, dtype=torch.int32)
    ids[:, 0] = -5
    out = model.encode_text(ids.cuda())
    torch.cuda.synchronize()
    assert torch.isfinite(out).all()


def test_head_and_leaderboard_argument_checks():
    import numpy as np

    import grip_amd  # noqa: F401
    from grip_amd import engine, native
    with pytest.raises(native.GripError, match="unsupported shape"):
        engine.cosine_head(torch.zeros(4, 6, device="cuda"), torch.zeros(3, 6, device="cuda"), 1.0)   # e % 4 != 0
    with pytest.raises(native.GripError, match="out of range"):
        engine.leaderboard_scan(np.ones((2, 3), np.float32) / 3, np.array([0, 7]), np.array([0, 1]), 2)
