"""The no-positional-embedding branches of the two prompt wrappers (G10): CustomTextEncoder.forward(..., enable_pos_emb=False)
(models/clip_encoders.py:70-74) and CustomVisionTransformer.forward(..., pos_emb=False) (:141) -- forward and prompt gradient of the
native towers (GRIP_FWD_NO_POS_EMB, include/grip_amd.h) against the outputs of the REFERENCE's own modules over the CPU fp32 oracle CLIP
(tests/golden/posemb.npz, written by oracle/gen_golden_posemb.py).  Same tolerances as the default branches (test_gpu_towers.py /
test_gpu_backward.py).  The f32 and split-f16 twins run the same embedding kernels and are checked on the forward."""
import os

import numpy as np
import pytest
import torch

from test_gpu_backward import assert_grad_close
from test_gpu_towers import _inputs, assert_embeddings_close

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "posemb.npz")
CLASSES = ["forest", "annual crop land", "river", "sea lake"]


@pytest.mark.parametrize("name,n_img,n_cls,P,tag", [("small", 2, 4, 16, "px.s"), ("ViT-B/16", 2, 3, 16, "px.b")])
def test_wrappers_without_positional_embedding(name, n_img, n_cls, P, tag):
    import grip_amd  # noqa: F401
    from grip_amd import clip, config
    from grip_amd.models import CustomImageEncoder, CustomTextEncoder
    g = np.load(GOLD)
    m, _ = clip.load(name, device="cuda")
    d = config.get_dims(name)
    classes = CLASSES[:n_cls]
    x = _inputs(f"{tag}.x", (n_img, 3, d.image_resolution, d.image_resolution)).cuda()
    vp = _inputs(f"{tag}.vprefix", (P, d.vision_width), 0.02).cuda().requires_grad_(True)
    tp = _inputs(f"{tag}.tprefix", (1, P, d.transformer_width), 0.02).cuda().requires_grad_(True)
    img_enc, txt_enc = CustomImageEncoder(m.visual), CustomTextEncoder(m, "cuda", torch.float32)
    assert np.array_equal(txt_enc._token_ids(P, classes).cpu().numpy(), g[f"{tag}.tokens"])

    v = img_enc.visual(x, vp, pos_emb=False)
    assert_embeddings_close(v, g[f"{tag}.vision"], "vision pos_emb=False")
    (v ** 2).sum().backward()
    assert_grad_close(vp.grad, g[f"{tag}.vision_grad_prefix"], "visual prompt grad, pos_emb=False")

    t = txt_enc(tp, classes, enable_pos_emb=False)
    assert_embeddings_close(t, g[f"{tag}.text"], "text enable_pos_emb=False")
    (t ** 2).sum().backward()
    assert_grad_close(tp.grad, g[f"{tag}.text_grad_prefix"], "textual prompt grad, enable_pos_emb=False")

    with torch.no_grad():
        # the flag changes the result (the default branch is elsewhere) ...
        assert (img_enc.visual(x, vp) - v).abs().max().item() > 1e-2 and (txt_enc(tp, classes) - t).abs().max().item() > 1e-2
        # ... and the inference forward equals the train-mode one
        assert_embeddings_close(img_enc.visual(x, vp, pos_emb=False), v.detach().cpu(), "inference forward")
        # exact (f32) twin: 1e-5 of the reference's fp32
        ex, _ = clip.load(name, device="cuda", exact=True)
        ve = CustomImageEncoder(ex.visual).visual(x, vp.detach(), pos_emb=False).cpu()
        te = CustomTextEncoder(ex, "cuda", torch.float32)(tp.detach(), classes, enable_pos_emb=False).cpu()
        for got, key in ((ve, "vision"), (te, "text")):
            want = torch.from_numpy(g[f"{tag}.{key}"])
            assert ((got - want).norm() / want.norm()).item() <= 2e-5, key
