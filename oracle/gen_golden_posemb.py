"""ORACLE (test infrastructure) -- G10: fixtures for the two no-positional-embedding branches of the wrappers.  Runs ONLY in
the build container (imports the reference from /root/reference, unmodified):

    python oracle/gen_golden_posemb.py            # writes tests/golden/posemb.npz

    models/clip_encoders.py:43 / :70-74    CustomTextEncoder.forward(class_embeddings, classes, enable_pos_emb=False)
    models/clip_encoders.py:123 / :141     CustomVisionTransformer.forward(x, image_prefix, pos_emb=False)

Neither is reached by a shipped config, both are part of the operator interface.  The reference's own modules run on the CPU
fp32 oracle CLIP (as in oracle/gen_golden.py); stored are their outputs and the prompt gradients of sum(out^2); inputs are
regenerated from seeds by the test.  The oracle restatements (oracle/wrappers.py) must reproduce both before anything is written."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as G  # noqa: E402  (sets up sys.path: oracle clip stand-in + the reference's `models`)

CASES = [("small", 2, 4, 16, "px.s"), ("ViT-B/16", 2, 3, 16, "px.b")]
CLASSES = ["forest", "annual crop land", "river", "sea lake"]


def main():
    out = {}
    for name, n_img, n_cls, P, tag in CASES:
        m, _ = G.clip.load(name)
        d = G.grip_amd.config.get_dims(name)
        classes = CLASSES[:n_cls]
        x = G.T(f"{tag}.x", (n_img, 3, d.image_resolution, d.image_resolution))
        vp = G.T(f"{tag}.vprefix", (P, d.vision_width), 0.02).requires_grad_(True)
        tp = G.T(f"{tag}.tprefix", (1, P, d.transformer_width), 0.02).requires_grad_(True)
        v = G.RM.CustomImageEncoder(m.visual).visual(x, vp, pos_emb=False)                 # REFERENCE :123-194
        t = G.RM.CustomTextEncoder(m, "cpu", torch.float32)(tp, classes, enable_pos_emb=False)   # REFERENCE :43-90
        tok = G.clip.tokenize(G.W.coop_prompt_strings(P, classes))
        G.close(G.W.vision_forward(m.visual, x, vp.detach(), pos_emb=False), v.detach(), f"{name} vision pos_emb=False")
        G.close(G.W.text_forward(m, tok, tp.detach(), enable_pos_emb=False), t.detach(), f"{name} text enable_pos_emb=False")
        with torch.no_grad():      # the branch must actually differ from the default one, or the fixture pins nothing
            assert (G.W.vision_forward(m.visual, x, vp.detach()) - v).abs().max() > 1e-2
            assert (G.W.text_forward(m, tok, tp.detach()) - t).abs().max() > 1e-2
        (v ** 2).sum().backward()
        (t ** 2).sum().backward()
        out[f"{tag}.vision"], out[f"{tag}.vision_grad_prefix"] = v.detach().numpy(), vp.grad.numpy()
        out[f"{tag}.text"], out[f"{tag}.text_grad_prefix"] = t.detach().numpy(), tp.grad.numpy()
        out[f"{tag}.tokens"] = tok.numpy().astype(np.int32)
        print(name, "vision", tuple(v.shape), "text", tuple(t.shape))
    np.savez_compressed(os.path.join(G.OUT, "posemb.npz"), **out)
    print("wrote tests/golden/posemb.npz")


if __name__ == "__main__":
    main()
