"""`TrainingStrategy.assign_pseudo_labels` against fixtures produced by the REFERENCE's own
assign_pseudo_labels (tests/golden/assign_{small,vitb16}.npz, written by oracle/gen_golden_assign.py, which executes
methods/transductive_zsl/multimodal_fpl.py:194-285, methods/semi_supervised_learning/textual_fpl.py:195-283 and
methods/unsupervised_learning/visual_fpl.py:185-328 unmodified over the reference's own prompt models on the CPU fp32 oracle CLIP).

Same seeded pool, same class names / label ids, same (moved) prompts and mixer weights -> the product must return the same
(filepaths, labels) lists: in the default `identical` mode (f16 screen + refinement) AND in the exact (all-f32) mode, with plain
list equality.  The fixtures were generated with pool / prompt seeds whose decision margin (oracle.leaderboard.scan_margin) is
>= 7e-5 relative, an order of magnitude above the fp32 GPU-vs-CPU deviation of the probabilities (asserted below: <= 2e-5)."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import REPO

pytestmark = pytest.mark.gpu

CLS = {"multi": ("MultimodalFPL", "trzsl"), "text": ("TextualFPL", "ssl"), "image": ("VisualFPL", "ul")}
PROB_TOL = {"small": 5e-5, "vitb16": 5e-5}      # relative, fp32 GPU towers vs the CPU oracle's probabilities (measured: see the printed value)


def _fixture(group):
    return np.load(os.path.join(REPO, "tests", "golden", f"assign_{group}.npz"))


def _strategy(fx, modality, monkeypatch, tmp_path, exact):
    """The product's strategy object set up as the fixture's `self` was: same encoder, classes, label ids, prompts."""
    import grip_amd  # noqa: F401
    from grip_amd import methods, rng, weights
    from grip_amd.data import TensorPoolDataset
    from grip_amd.methods.main import DEFAULTS, Config, synthetic_pool
    meta = json.loads(str(fx[f"{modality}.meta"]))
    monkeypatch.chdir(tmp_path)
    monkeypatch.delenv("GRIP_PSEUDOLABEL_MODE", raising=False)
    if exact:
        monkeypatch.setenv("GRIP_EXACT", "1")
    else:
        monkeypatch.delenv("GRIP_EXACT", raising=False)
    cls_name, paradigm = CLS[modality]
    c = dict(DEFAULTS)
    c.update(OPTIM_SEED=1, VIS_ENCODER=meta["encoder"], DATASET_NAME="Synthetic", SPLIT_SEED=500, DATASET_DIR="", MODEL="x", LEARNING_PARADIGM=paradigm,
             PREFIX_SIZE=meta["P"], TEXT_PREFIX_SIZE=meta["P"], VISION_PREFIX_SIZE=meta["P"], N_PSEUDOSHOTS=meta["k"])
    conf = Config(**c)
    classes, unseen, l2i = meta["classes"], meta["unseen"], meta["label_to_idx"]
    seen = [x for x in classes if x not in unseen] if modality == "multi" else classes
    m = getattr(methods, cls_name)(conf, l2i, "", classes, seen, unseen if modality == "multi" else classes, "cuda")
    assert m.clip_model.exact == exact
    m.define_model(classes)
    with torch.no_grad():
        if modality == "multi":
            m.model.coop_embeddings.copy_(torch.from_numpy(fx["multi.coop"]))
            m.model.vpt_embeddings.copy_(torch.from_numpy(fx["multi.vpt"]))
            d = m.clip_model.dims
            mixer = {k: torch.from_numpy(v) for k, v in weights.init_upt_mixer(d.transformer_width, d.vision_width, 128, meta["prompt_seed"]).items()}
            missing, unexpected = m.model.load_state_dict(mixer, strict=False)
            assert not unexpected and not [k for k in missing if k.startswith(("proj_", "transformer."))], (missing, unexpected)
        else:
            m.model.prefix.copy_(torch.from_numpy(fx[f"{modality}.prefix"]))
    _, files, images, _ = synthetic_pool(meta["n_classes"], meta["n_per_class"], m.clip_model.dims.image_resolution, meta["pool_seed"])
    assert [f"/data/synthetic/train/{f}" for f in files] == meta["paths"]
    data = TensorPoolDataset(meta["paths"], images.cuda(), labels=None, label_map=l2i)
    return m, data, meta


@pytest.mark.parametrize("group", ["small", "vitb16"])
@pytest.mark.parametrize("modality", ["multi", "text", "image"])
@pytest.mark.parametrize("exact", [False, True], ids=["identical", "exact"])
def test_assign_pseudo_labels_returns_the_reference_lists(tmp_path, monkeypatch, group, modality, exact):
    from grip_amd import pseudolabels as pl
    fx = _fixture(group)
    m, data, meta = _strategy(fx, modality, monkeypatch, tmp_path, exact)
    pl.LAST_REFINE_STATS = None
    out = m.assign_pseudo_labels(meta["k"], data)
    want_fp, want_lab = meta["lists"]
    assert out is data and out.label_id is True
    assert (list(out.filepaths), [int(x) for x in out.labels]) == (want_fp, want_lab), \
        f"{group}.{modality} ({'exact' if exact else 'identical'} mode): lists differ from the reference's assign_pseudo_labels"
    if not exact:
        st = pl.LAST_REFINE_STATS
        assert st is not None and st["rows"] == len(meta["paths"])          # the default path is the screen-and-refine one
        print(f"{group}.{modality}: {st['rows_refined']} of {st['rows']} rows refined, bound {st['eps']:.2e}")


@pytest.mark.parametrize("group,modality", [("small", "multi"), ("small", "image"), ("vitb16", "text"), ("vitb16", "multi")])
def test_three_tier_refinement_returns_the_reference_lists(tmp_path, monkeypatch, group, modality):
    """The same fixtures through the THREE-tier path (GRIP_SPLIT_TIER=1 forces the split-f16 middle tier, which pools this small would skip):
    f16 screen -> split-f16 tower -> f32 tower, trained prompts in all three towers; still the reference's lists."""
    from grip_amd import pseudolabels as pl
    monkeypatch.setenv("GRIP_SPLIT_TIER", "1")
    fx = _fixture(group)
    m, data, meta = _strategy(fx, modality, monkeypatch, tmp_path, False)
    out = m.assign_pseudo_labels(meta["k"], data)
    st = pl.LAST_REFINE_STATS
    print(f"{group}.{modality}: {st['rows_mid']} split-f16 rows, {st['rows_exact']} f32 rows, bounds {st['eps']:.2e} / {st['eps_mid']:.2e}")
    assert (list(out.filepaths), [int(x) for x in out.labels]) == tuple(meta["lists"])
    assert st["tiers"] == 3 and st["rows_mid"] > 0 and st["eps_mid"] < st["eps"]


@pytest.mark.parametrize("group", ["small", "vitb16"])
@pytest.mark.parametrize("modality", ["multi", "text", "image"])
def test_trained_prompt_probabilities_match_the_reference(tmp_path, monkeypatch, group, modality):
    """The fp32 probabilities the reference method compared (softmax of logit_scale x cosine, features of the reference's own
    prompt models) against the exact mode's head over `trained_features` -- text tower with the trained prompt / mixer output, image
    tower with the trained visual prompt -- and the f16 towers' features against the reference's to the north-star cosine bound."""
    from grip_amd import engine
    fx = _fixture(group)
    m, data, meta = _strategy(fx, modality, monkeypatch, tmp_path, True)
    target = meta["unseen"]
    img, txt = m.trained_features(data.images, target)
    _, probs, am_l, _ = engine.cosine_head(img, txt, m.scale())
    p, want = probs.cpu().numpy().astype(np.float64), fx[f"{modality}.probs"].astype(np.float64)
    rel = np.abs(p - want) / want
    print(f"{group}.{modality}: exact-mode probabilities vs the reference's: max rel {rel.max():.2e}, mean {rel.mean():.2e}")
    assert rel.max() <= PROB_TOL[group]
    assert np.array_equal(am_l.cpu().numpy(), fx[f"{modality}.pred"])
    cos = torch.nn.functional.cosine_similarity
    assert 1 - cos(img.cpu(), torch.from_numpy(fx[f"{modality}.img_feats"]), dim=1).min().item() <= 1e-6
    assert 1 - cos(txt.cpu(), torch.from_numpy(fx[f"{modality}.txt_feats"]), dim=1).min().item() <= 1e-6
    # default (f16) towers: north_star "within 1e-3 cosine on fp32 embeddings"
    m16, data16, _ = _strategy(fx, modality, monkeypatch, tmp_path, False)
    img16, txt16 = m16.trained_features(data16.images, target)
    assert 1 - cos(img16.float().cpu(), torch.from_numpy(fx[f"{modality}.img_feats"]), dim=1).min().item() <= 1e-4
    assert 1 - cos(txt16.float().cpu(), torch.from_numpy(fx[f"{modality}.txt_feats"]), dim=1).min().item() <= 1e-4


SUB_ULP_MARGIN = 2.0 ** -17      # one ulp of an fp32 logit in [64, 128) (100 x cosine), as a relative step of the probability it produces


class _Sub:
    """One case of an assign_unselected_*.npz file under the key names of the selected fixtures."""

    def __init__(self, fx, modality, seed):
        self.fx, self.pre, self.modality = fx, f"{modality}.{seed}.", modality

    def __getitem__(self, key):
        m, what = key.split(".", 1)
        assert m == self.modality
        return self.fx[self.pre + what]


@pytest.mark.parametrize("group", ["small", "vitb16"])
@pytest.mark.parametrize("modality", ["multi", "text", "image"])
def test_unselected_seeds(tmp_path, monkeypatch, group, modality):
    """the fixtures above were SELECTED for a decision margin >= 7e-5.  These are the next ten (ViT-B/16: four) pool / prompt seeds per modality, whatever
    their margin (oracle/gen_golden_assign.py unselected-<group>: the reference's own assign_pseudo_labels, margins down to 3e-7 relative).  What must
    hold for every one of them: the same number of pairs per class, and lists identical to the reference's whenever its decision margin is above
    the fp32 GPU-vs-CPU deviation of the probabilities (PROB_TOL); below it -- the reference's own outcome then hangs on the last bits of its
    BLAS -- a boundary pair may differ.  The count of list-identical cases is printed: it quantifies the sub-ulp tie rate instead of avoiding it."""
    from grip_amd import pseudolabels as pl
    path = os.path.join(REPO, "tests", "golden", f"assign_unselected_{group}.npz")
    if not os.path.exists(path):
        pytest.skip(f"{path} not generated")
    fx = np.load(path)
    seeds = sorted({int(k.split(".")[1]) for k in fx.files if k.startswith(modality + ".") and k.endswith(".meta")})
    assert len(seeds) == (10 if group == "small" else 4)
    identical = same_set = 0
    report, rows = [], []
    for seed in seeds:
        sub = _Sub(fx, modality, seed)
        m, data, meta = _strategy(sub, modality, monkeypatch, tmp_path, False)
        out = m.assign_pseudo_labels(meta["k"], data)
        got = list(zip(out.filepaths, [int(x) for x in out.labels]))
        want = list(zip(*meta["lists"]))
        margin = float(fx[f"{modality}.{seed}.margin"])
        by_class = lambda pairs: sorted(np.unique([l for _, l in pairs], return_counts=True)[1].tolist())    # noqa: E731
        assert len(got) == len(want) and by_class(got) == by_class(want), (seed, len(got), len(want))
        overlap = len(set(got) & set(want)) / len(want)
        identical += got == want
        same_set += set(got) == set(want)
        report.append(f"{seed}: margin {margin:.1e} {'identical' if got == want else 'same set' if set(got) == set(want) else f'overlap {overlap:.3f}'}")
        rows.append({"seed": seed, "margin": margin, "identical": got == want, "same_pair_set": set(got) == set(want), "overlap": overlap, "pairs": len(want)})
        if margin >= SUB_ULP_MARGIN:
            assert got == want, f"{group}.{modality} seed {seed}: margin {margin:.2e} is above one fp32 ulp of the logits, yet the lists differ"
        else:
            assert overlap >= 0.9, (seed, overlap)
        del m
        torch.cuda.empty_cache()
    print(f"{group}.{modality}: {identical} of {len(seeds)} un-selected seeds list-identical, {same_set} with the same pair set; " + "; ".join(report))
    from conftest import write_report
    write_report(f"unselected_seeds_{group}_{modality}.json", {"group": group, "modality": modality, "seeds": len(seeds), "list_identical": identical,
                                                              "same_pair_set": same_set, "sub_ulp_margin": SUB_ULP_MARGIN, "cases": rows})
    # The allowance, numbered: a seed may differ from the reference's list ONLY when the reference's own decision margin is below one fp32 ulp of
    # the logits it compared (SUB_ULP_MARGIN: its outcome then hangs on the last bit of its BLAS), and at most ONE such seed per (group, modality).
    # Measured r05 / r06: 42 of 42 identical -- the allowance has never been used.
    sub_ulp = sum(r["margin"] < SUB_ULP_MARGIN for r in rows)
    assert identical >= len(seeds) - min(1, sub_ulp), (identical, len(seeds), rows)
