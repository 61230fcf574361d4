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


# ------------------------------------------------------------------------------------------------------------------ reference-run LISTS (r06)
def lists(n=2048, n_classes=40, seed=77, variant="stress"):
    """`python oracle/gen_golden_stress.py lists` -> tests/golden/stress_vitb16_lists.npz (~6 min on 8 cores);
    `python oracle/gen_golden_stress.py lists realistic` -> tests/golden/realistic_vitb16_lists.npz: the same with the STANDARD synthetic weights and the
    un-centred prototype blend unit(m + 2 (e_c - m)) of bench.py's `identical_on_realistic_pool` (peaked rows at ordinary logit errors).

    The REFERENCE's own utils/clip_pseudolabels.compute_pseudo_labels (:13-117, imported unmodified from /root/reference) driven over `n` structured
    images on the CPU fp32 oracle with the STRESS weights, for k in {3, 16, 10000000}.  The class side is what tests/test_gpu_stress.py uses: the
    mean-removed prototypes of `n_classes` anchor images' own embeddings (peaked rows, contested arg-maxes) -- handed to the reference function as
    the text features of a wrapped model (its encode_text returns them; the image tower, the normalisation, the logit scale, the softmax, the
    arg-max and the scan are the reference's / the oracle's own).  Stored: the text features, the fp32 probabilities the reference compared, its
    lists and the decision margin of each scan.  tests/test_gpu_stress.py compares the exact- and identical-mode lists of the GPU towers with them."""
    import json
    import time
    sys.path.insert(0, HERE)                      # `import clip` -> oracle/clip (the reference imports it)
    sys.path.insert(0, "/root/reference")
    from grip_amd.data.synthetic import pool_paths
    from utils import clip_pseudolabels as RP     # REFERENCE, unmodified
    from oracle import leaderboard as LB
    d = gcfg.get_dims("ViT-B/16")
    om = build(weights.stress_state_dict(d, 0) if variant == "stress" else weights.init_state_dict(d, 0), d)
    paths = pool_paths(n)
    index = {p: i for i, p in enumerate(paths)}
    t0 = time.time()
    emb = torch.empty(n, d.embed_dim)
    with torch.no_grad():
        for lo in range(0, n, 32):
            x = structured_images(seed, lo, min(lo + 32, n), 224)
            for i in range(x.shape[0]):            # batch 1, as the reference loop encodes (:35)
                emb[lo + i] = om.encode_image(x[i:i + 1])[0]
            if lo % 256 == 0:
                print(f"encoded {lo} of {n} images, {time.time() - t0:.0f} s", flush=True)
    g = torch.Generator().manual_seed(seed)
    anchors = torch.randperm(n, generator=g)[:n_classes]
    en = emb / emb.norm(dim=-1, keepdim=True)
    mean = en.mean(0, keepdim=True)
    if variant == "stress":
        txt = (en[anchors] - mean + 0.003 * torch.randn(n_classes, d.embed_dim, generator=g)).contiguous()
    else:
        txt = mean + 2.0 * (en[anchors] - mean)
        txt = (txt / txt.norm(dim=-1, keepdim=True)).contiguous()

    class _FakeImg:
        def __init__(self, idx):
            self.idx = idx

        def convert(self, mode):
            return self

    class _Dataset:
        def __init__(self, p):
            self.filepaths, self.labels = list(p), None

    class _Model:
        """clip_model(image, text): the oracle CLIP.forward with the prototype text features; the image features are the batch-1 embeddings above
        (a pure function of the image: the memo returns what re-encoding would)."""
        logit_scale = om.logit_scale
        seen = {}

        def __call__(self, image, text):
            i = int(image[0, 0, 0, 0].item())
            f = emb[i:i + 1]
            f = f / f.norm(dim=1, keepdim=True)
            t = txt / txt.norm(dim=1, keepdim=True)
            logits = om.logit_scale.exp() * f @ t.t()
            self.seen[i] = logits[0].clone()
            return logits, logits.t()

    model = _Model()
    RP.Image.open = lambda path: _FakeImg(index[path])
    RP.tqdm = lambda it: it
    transform = lambda img: torch.full((3, 1, 1), float(img.idx))          # carries the index to the memo (the image itself was encoded above)
    classnames = [f"kind_{i:03d}" for i in range(n_classes)]
    label_to_idx = {c: 100 + i for i, c in enumerate(classnames)}
    out = {"seed": np.int64(seed), "txt": txt.numpy(), "anchors": anchors.numpy(), "logit_scale": np.float32(om.logit_scale.exp().item())}
    with torch.no_grad():
        for k in (3, 16, 10000000):
            ds = _Dataset(paths)
            RP.compute_pseudo_labels(k, "a photo of a {}", ds, classnames, transform, model, label_to_idx, "cpu", "/tmp/_stress_pl.pickle")
            os.remove("/tmp/_stress_pl.pickle")
            out[f"lists_k{k}"] = json.dumps([list(ds.filepaths), [int(v) for v in ds.labels]])
            print(f"k={k}: {len(ds.filepaths)} pairs", flush=True)
    logits = torch.stack([model.seen[i] for i in range(n)])
    probs = logits.softmax(dim=-1).numpy().astype(np.float32)              # :38 of the reference
    pred = probs.argmax(1)
    for k in (3, 16, 10000000):
        want = json.loads(out[f"lists_k{k}"])
        got = LB.leaderboard_scan(probs, pred, paths, [100 + i for i in range(n_classes)], k)
        assert [list(got[0]), list(got[1])] == want, f"k={k}: oracle scan != reference"
        out[f"margin_k{k}"] = np.float64(LB.scan_margin(probs, pred, k))
        print(f"k={k}: relative decision margin {out[f'margin_k{k}']:.3e}")
    out["probs"] = probs
    print(f"mean top-1 probability {probs.max(1).mean():.3f}, {len(np.unique(pred))} arg-max classes, logit spread "
          f"{np.mean(np.log(np.maximum(probs, 1e-45)).max(1) - np.median(np.log(np.maximum(probs, 1e-45)), 1)):.1f}")
    path = os.path.join(REPO, "tests", "golden", f"{variant}_vitb16_lists.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, f"{time.time() - t0:.0f} s")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "lists":
        lists(variant=sys.argv[2] if len(sys.argv) > 2 else "stress")
    else:
        main()
