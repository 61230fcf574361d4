"""ORACLE (test infrastructure) -- fixtures for the STRESS model (grip_amd.weights.stress_state_dict: outlier channels + an image-dependent f16
overflow): the CPU fp32 oracle's image embeddings of the first 16 images of the structured pool (seed 77), for the full recipe ("stress") and for
variants of it that tools/stress_probe.py compares the GPU towers on.  Build container only:
    python oracle/gen_golden_stress.py  ->  tests/golden/stress_vitb16.npz"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO)
import grip_amd  # noqa: E402,F401
from grip_amd import config as gcfg, weights  # noqa: E402
from grip_amd.data.synthetic import structured_images  # noqa: E402
from oracle.clip.model import CLIP  # noqa: E402

VARIANTS = {        # tag -> keyword arguments of weights.stress_state_dict
    "stress": {},
    "outliers200": {"overflow_gain": (1.0, 1.0)},
    "outliers50": {"overflow_gain": (1.0, 1.0), "outlier": 50.0},
    "overflow": {"channels": ()},
}


def build(sd, d):
    m = CLIP(d.embed_dim, d.image_resolution, d.vision_layers, d.vision_width, d.vision_patch_size, d.context_length, d.vocab_size, d.transformer_width,
             d.transformer_heads, d.transformer_layers)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    return m.float().eval()


def main():
    d = gcfg.get_dims("ViT-B/16")
    x = structured_images(77, 0, 16, 224)
    out = {}
    for tag, kw in VARIANTS.items():
        m = build(weights.stress_state_dict(d, 0, **kw), d)
        with torch.no_grad():
            out[tag] = m.encode_image(x).numpy()
            from oracle.clip import model as OM                      # how well conditioned the model is: fp32 against fp64 (CLIP's LayerNorm pins fp32)
            keep = OM.LayerNorm.forward
            OM.LayerNorm.forward = torch.nn.LayerNorm.forward
            try:
                out[tag + ".f64"] = m.double().encode_image(x.double()).float().numpy()
            finally:
                OM.LayerNorm.forward = keep
        c = torch.nn.functional.cosine_similarity(torch.from_numpy(out[tag]), torch.from_numpy(out[tag + ".f64"]), dim=1)
        print(f"{tag}: |e| max {np.abs(out[tag]).max():.3g}; fp32 vs fp64 oracle 1-cos max {float((1 - c).max()):.2e}", flush=True)
    path = os.path.join(REPO, "tests", "golden", "stress_vitb16.npz")
    np.savez_compressed(path, **out)
    print("wrote", path)


if __name__ == "__main__":
    main()
