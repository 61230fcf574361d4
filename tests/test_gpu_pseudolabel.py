"""End-to-end pseudolabel path on the GPU through the reference-named entry points
(utils.pseudolabel_top_k -> compute_pseudo_labels): pool encode + cached text features + fused head +
exact scan, against the CPU oracle doing the reference's per-image loop on the same inputs; cache-file
name and pickle schema; the arg-max-only branch; determinism."""
import os
import pickle
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


class _Pool:
    """Dataset stand-in with the attributes pseudolabel_top_k touches (filepaths, labels) plus a
    pre-decoded tensor pool aligned with filepaths."""

    def __init__(self, images, paths):
        self.images, self.filepaths, self.labels = images, list(paths), None


def _structured_images(n, res, seed):
    import grip_amd  # noqa: F401
    from grip_amd import rng
    x = torch.from_numpy(rng.normal(seed, rng.stream_id("pl.x"), (n, 3, res, res)))
    # per-image colour cast + low-frequency pattern: random-init towers separate such images far
    # better than i.i.d. noise, so score margins are well above f16 rounding
    mu = torch.from_numpy(rng.normal(seed, rng.stream_id("pl.mu"), (n, 3, 1, 1))) * 2.0
    ramp = torch.linspace(-1, 1, res).view(1, 1, 1, res) * torch.from_numpy(rng.normal(seed, rng.stream_id("pl.r"), (n, 3, 1, 1)))
    return x * 0.5 + mu + ramp


def _oracle_lists(name, images, paths, classnames, label_to_idx, k, template):
    """The reference algorithm on the CPU oracle: per-image clip_model(image, text) -> softmax ->
    argmax(probs) -> literal leaderboard (utils/clip_pseudolabels.py:24-112)."""
    from conftest import oracle_logits
    from oracle import leaderboard as LB
    logits = oracle_logits(name, images, template, classnames)
    probs, pred = LB.softmax_argmax(logits.numpy())
    return LB.leaderboard_scan(probs, pred, paths, [label_to_idx[c] for c in classnames], k), probs


@pytest.mark.parametrize("k", [3, 16, 10000000])
def test_pseudolabel_top_k_matches_reference_algorithm(tmp_path, monkeypatch, k):
    import grip_amd  # noqa: F401
    from grip_amd import clip
    from grip_amd.utils import pseudolabel_top_k
    monkeypatch.chdir(tmp_path)
    monkeypatch.setenv("GRIP_PSEUDOLABEL_MODE", "f16")      # the f16 towers' own lists; the default mode (identical) is tests/test_gpu_identical.py
    name, n = "small", 300
    m, transform = clip.load(name, device="cuda")
    images = _structured_images(n, 64, 21)
    paths = [f"/data/EuroSAT/train/{(i * 7919) % 1000:04d}_{i}.jpg" for i in range(n)]
    classnames = ["annual_crop_land", "forest", "herbaceous_vegetation", "highway", "industrial_buildings", "pasture", "river"]
    label_to_idx = {c: i + 10 for i, c in enumerate(classnames)}
    cfg = types.SimpleNamespace(LEARNING_PARADIGM="ul", MODEL="visual_fpl")
    ds = _Pool(images, paths)
    out = pseudolabel_top_k(cfg, "EuroSAT", k, "a photo of a {}", ds, classnames, transform, m, label_to_idx, "cuda", "ViT-B/32", 500)
    assert out is ds
    (want_fp, want_lab), o_probs = _oracle_lists(name, images, paths, classnames, label_to_idx, k, "a photo of a {}")
    # (1) the scan is exact: the literal oracle scan over the probabilities the GPU produced gives the same lists
    from grip_amd import engine, pseudolabels as pl
    from oracle import leaderboard as LB
    with torch.no_grad():
        emb = pl.encode_pool(m.visual.tower, images)
        txt = m.encode_text(clip.tokenize([f"a photo of a {{}}{' '.join(c.split('_'))}" for c in classnames]).cuda())
    _, g_probs, _, g_pred = engine.cosine_head(emb, txt, m.logit_scale.exp().item())
    exact = LB.leaderboard_scan(g_probs.cpu().numpy(), g_pred.cpu().numpy(), paths, [label_to_idx[c] for c in classnames], k)
    assert (ds.filepaths, ds.labels) == exact
    # (2) the probabilities agree with the fp32 oracle to f16-operand accuracy.  Softmax of 100 x cosine: an embedding error of
    # 1e-3 relative L2 (the measured parity margin) moves a logit by up to ~0.1 and a probability by up to ~0.025; the worst
    # seen over kernel revisions is 5.1e-3 on this random-init model.  The binding criteria are (1) and (3).
    assert np.abs(g_probs.cpu().numpy() - o_probs).max() <= 5e-3      # measured 2.4e-3 (r03)
    # (3) end to end against the oracle's own lists.  The scan compares near-tied probabilities with strict '<', so a
    # 4th-digit difference (f16 operands and residual stream vs the fp32 oracle) may swap a boundary item: the lists must
    # agree on at least 90 % of the (image, label) pairs, and the arg-max-only branch on 99 % of the images.
    got_pairs, want_pairs = set(zip(ds.filepaths, ds.labels)), set(zip(want_fp, want_lab))
    overlap = len(got_pairs & want_pairs) / len(want_pairs)
    print(f"k={k}: f16 lists overlap the fp32 oracle's {overlap:.4f}; max |dp| {np.abs(g_probs.cpu().numpy() - o_probs).max():.2e}")
    # measured (r03): k = 3: 20 of 21 pairs (0.952), k = 16: 1.000, label-everything: 1.000.  This is the f16 MODE (no index guarantee); the
    # default mode returns the fp32 lists themselves (tests/test_gpu_identical.py)
    assert overlap >= {3: 0.95, 16: 0.98}.get(k, 0.995), overlap
    want_fp, want_lab = ds.filepaths, ds.labels
    # cache: file name and schema of utils/clip_pseudolabels.py:134 / :114-115
    fn = f"pseudolabels/EuroSAT_ViT-B32_ul_visual_fpl_{k}_pseudolabels_split_500.pickle"
    assert os.path.exists(fn)
    with open(fn, "rb") as f:
        blob = pickle.load(f)
    assert set(blob) == {"filepaths", "labels"} and blob["filepaths"] == want_fp and blob["labels"] == want_lab
    # second call hits the cache and does not need images at all
    ds2 = _Pool(None, paths)
    pseudolabel_top_k(cfg, "EuroSAT", k, "a photo of a {}", ds2, classnames, None, None, label_to_idx, "cuda", "ViT-B/32", 500)
    assert ds2.filepaths == want_fp and ds2.labels == want_lab


def test_probabilities_close_and_pipeline_deterministic():
    import grip_amd  # noqa: F401
    from grip_amd import clip, engine, pseudolabels as pl
    name, n = "small", 257          # ragged last chunk
    m, _ = clip.load(name, device="cuda")
    images = _structured_images(n, 64, 33)
    classes = ["a", "b c", "d", "e f g", "h"]
    tok = clip.tokenize([f"a photo of a {{}}{c}" for c in classes]).cuda()
    runs = []
    for chunk in (64, 100):
        with torch.no_grad():
            emb = pl.encode_pool(m.visual.tower, images, chunk=chunk)
            txt = m.encode_text(tok)
        logits, probs, am_l, am_p = engine.cosine_head(emb, txt, m.logit_scale.exp().item())
        runs.append((emb.cpu(), probs.cpu(), am_p.cpu()))
    # same images in different chunkings give bit-identical rows (no cross-row coupling anywhere)
    assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1])
    from conftest import oracle_clip
    oc = oracle_clip()
    om, _ = oc.load(name)
    with torch.no_grad():
        li, _ = om(images, oc.tokenize([f"a photo of a {{}}{c}" for c in classes]))
    want = li.softmax(-1)
    assert (runs[0][1] - want).abs().max().item() <= 1e-2
    assert (runs[0][2].long() == want.argmax(1)).float().mean().item() >= 0.99


def test_assign_pseudo_labels_style_scan_uses_logit_argmax():
    """assign_pseudo_labels (e.g. methods/transductive_zsl/multimodal_fpl.py:194-285) takes arg-max over
    LOGITS; with saturated softmax rows that differs from arg-max over probs."""
    import grip_amd  # noqa: F401
    from grip_amd import pseudolabels as pl
    from oracle import leaderboard as LB
    img = torch.eye(4, 128, device="cuda") + 0.001 * torch.arange(4, device="cuda").view(4, 1)
    txt = torch.eye(3, 128, device="cuda")
    paths = [f"x{i}" for i in range(4)]
    got = pl.pseudolabel_from_features(img, txt, 100.0, paths, [5, 6, 7], 2, argmax_on="logits")
    i = (img / img.norm(dim=-1, keepdim=True)).cpu()
    t = (txt / txt.norm(dim=-1, keepdim=True)).cpu()
    logits = 100.0 * i @ t.t()
    probs = logits.softmax(-1).numpy()
    want = LB.leaderboard_scan(probs, logits.argmax(1).numpy(), paths, [5, 6, 7], 2)
    assert got == want


def test_pseudolabel_top_k_from_image_files(tmp_path, monkeypatch):
    """The reference's real input path: a dataset of FILE paths opened with PIL and pushed through the transform that
    clip.load returned (here the GPU preprocess), against the oracle fed by the PIL/numpy transform chain."""
    from PIL import Image

    import grip_amd  # noqa: F401
    from grip_amd import clip
    from grip_amd.utils import pseudolabel_top_k
    from test_preprocess import _reference_transform
    monkeypatch.chdir(tmp_path)
    m, preprocess = clip.load("small", device="cuda")
    g = np.random.RandomState(4)
    paths, arrays = [], []
    (tmp_path / "imgs").mkdir()
    for i in range(40):
        h, w = 70 + g.randint(0, 60), 70 + g.randint(0, 90)
        base = g.randint(0, 256, size=(1, 1, 3)) * np.ones((h, w, 1)) * 0.7 + g.randint(0, 80, size=(h, w, 3))
        arr = np.clip(base, 0, 255).astype(np.uint8)
        p = str(tmp_path / "imgs" / f"{i:03d}.png")
        Image.fromarray(arr).save(p)
        paths.append(p)
        arrays.append(arr)
    classnames = ["forest", "river", "highway", "pasture"]
    label_to_idx = {c: i for i, c in enumerate(classnames)}

    class FileDataset:
        def __init__(self, fp):
            self.filepaths, self.labels = list(fp), None
    ds = FileDataset(paths)
    cfg = types.SimpleNamespace(LEARNING_PARADIGM="ul", MODEL="textual_fpl")
    pseudolabel_top_k(cfg, "Files", 10000000, "a photo of a {}", ds, classnames, preprocess, m, label_to_idx, "cuda", "small", 1)
    x = torch.from_numpy(np.stack([_reference_transform(a, 64) for a in arrays]))
    (want_fp, want_lab), _ = _oracle_lists("small", x, paths, classnames, label_to_idx, 10000000, "a photo of a {}")
    assert ds.filepaths == want_fp
    assert np.mean(np.array(ds.labels) == np.array(want_lab)) >= 0.95
