"""ORACLE (test infrastructure) -- fixture for the index-exact comparison at ViT-B/16 size.  Runs ONLY in the build
container (imports the reference from /root/reference, unmodified):

    python oracle/gen_golden_exact.py        # ~7 min on 8 cores; writes tests/golden/exact_vitb16_c{10,102}.npz
    python oracle/gen_golden_exact.py 10000  # ~40 min; writes tests/golden/exact_vitb16_c102_n10000.npz (102 classes only)

The reference's own utils/clip_pseudolabels.compute_pseudo_labels (:13-117) is driven over N = 2 000 seeded structured
images (grip_amd.data.synthetic, regenerated from the seed on the GPU box) on the CPU fp32 oracle CLIP (ViT-B/16
dimensions, seeded synthetic weights), once per k in {3, 16, 10000000} and per class set: the 10 EuroSAT class names
(BASELINE.json configs[0]) and 102 synthetic names (the Flowers102-shaped bench workload, configs[1]).  The oracle model
is wrapped so that each set of class prompts is encoded once and every image once: the reference loop calls
clip_model(image, text) per image and per k, re-encoding the same prompts each time -- a pure function of
identical inputs, so the memo returns exactly what the re-computation would (the image tower still runs at batch 1, as in
the loop).  Stored per class set: the fp32 probabilities softmax(logits) the reference compared, the token ids it
tokenised, its output lists, and the relative decision margin of each scan (oracle.leaderboard.scan_margin).
"""
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
REF = "/root/reference"
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)        # `import clip` -> oracle/clip
sys.path.insert(0, REF)

import clip  # noqa: E402  (oracle stand-in)
import grip_amd  # noqa: E402,F401
from grip_amd.data.synthetic import pool_paths, structured_images  # noqa: E402
from utils import clip_pseudolabels as RP  # noqa: E402  (REFERENCE, unmodified)
from oracle import leaderboard as LB  # noqa: E402

N, C, SEED, NAME = (int(sys.argv[1]) if len(sys.argv) > 1 else 2000), 102, 4242, "ViT-B/16"
BIG = N != 2000      # the larger pool: 102 classes only, own file name


class _FakeImg:
    def __init__(self, idx):
        self.idx = idx

    def convert(self, mode):
        return self


class _Dataset:
    def __init__(self, paths):
        self.filepaths = list(paths)
        self.labels = None


class _Memo:
    """clip_model(image, text) of the oracle CLIP with both towers memoised (see module docstring)."""

    def __init__(self, om):
        self.om, self.txt, self.img, self.cur = om, {}, {}, None
        self.logits = {}

    def __call__(self, image, text):
        om = self.om
        key = text.numpy().tobytes()
        if key not in self.txt:
            t = om.encode_text(text)
            self.txt[key] = (t / t.norm(dim=1, keepdim=True), text.clone())
        i = self.cur
        if i not in self.img:
            f = om.encode_image(image)
            self.img[i] = f / f.norm(dim=1, keepdim=True)
        logits = om.logit_scale.exp() * self.img[i] @ self.txt[key][0].t()      # oracle CLIP.forward, verbatim tail
        self.logits[(key, i)] = logits[0].clone()
        self.last_key = key
        return logits, logits.t()


EUROSAT = ["annual_crop_land", "forest", "herbaceous_vegetation_land", "highway_or_road", "industrial_buildings", "pasture_land",
           "permanent_crop_land", "residential_buildings", "river", "sea_or_lake"]


def main():
    torch.manual_seed(0)
    om, _ = clip.load(NAME)
    paths = pool_paths(N)
    index = {p: i for i, p in enumerate(paths)}
    memo = _Memo(om)
    cache = {}

    def transform(img):
        memo.cur = img.idx
        blk = img.idx // 50
        if blk not in cache:
            cache.clear()
            cache[blk] = structured_images(SEED, blk * 50, min(blk * 50 + 50, N), 224)
        return cache[blk][img.idx - blk * 50]

    RP.Image.open = lambda path: _FakeImg(index[path])
    RP.tqdm = lambda it: it
    t0 = time.time()
    sets = (("c10", EUROSAT), ("c102", [f"kind_{i:03d}" for i in range(C)])) if not BIG else ((f"c102_n{N}", [f"kind_{i:03d}" for i in range(C)]),)
    for tag, classnames in sets:
        label_to_idx = {c: i for i, c in enumerate(classnames)}
        out = {"seed": np.int64(SEED)}
        with torch.no_grad():
            for k in (3, 16, 10000000):
                ds = _Dataset(paths)
                RP.compute_pseudo_labels(k, "a photo of a {}", ds, classnames, transform, memo, label_to_idx, "cpu", "/tmp/_exact_pl.pickle")
                os.remove("/tmp/_exact_pl.pickle")
                out[f"lists_k{k}"] = json.dumps([list(ds.filepaths), [int(x) for x in ds.labels]])
                print(f"{tag} k={k}: {len(ds.filepaths)} pairs, {time.time() - t0:.0f} s", flush=True)
        key = memo.last_key
        logits = torch.stack([memo.logits[(key, i)] for i in range(N)])
        probs = logits.softmax(dim=-1).numpy().astype(np.float32)      # :38 of the reference
        pred = probs.argmax(1)
        for k in (3, 16, 10000000):      # the literal oracle scan over these probabilities reproduces the reference's lists
            want = json.loads(out[f"lists_k{k}"])
            got = LB.leaderboard_scan(probs, pred, paths, list(range(len(classnames))), k)
            assert [list(got[0]), list(got[1])] == want, f"{tag} k={k}: oracle scan != reference"
            out[f"margin_k{k}"] = np.float64(LB.scan_margin(probs, pred, k))
            print(f"{tag} k={k}: relative decision margin {out[f'margin_k{k}']:.3e}")
        out["probs"] = probs
        out["tokens"] = memo.txt[key][1].numpy().astype(np.int32)
        np.savez_compressed(os.path.join(REPO, "tests", "golden", f"exact_vitb16_{tag}.npz"), **out)
        print(f"wrote tests/golden/exact_vitb16_{tag}.npz")


if __name__ == "__main__":
    main()
