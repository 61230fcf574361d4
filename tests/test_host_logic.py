"""CPU: host-side logic of the product -- tokenizer stand-in, prompt strings, sharding, FPL row
weights, the exact host leaderboard (C++ in libgrip_amd.so) against the oracles and the golden
vectors."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import REPO, oracle_clip


def test_tokenizer_stand_in_is_identical_on_both_sides():
    import grip_amd  # noqa: F401
    from grip_amd import clip
    texts = ["a photo of a {}forest", "X X X X annual crop land", "Sea-Lake 42!", "x"]
    assert (clip.tokenize(texts) == oracle_clip().tokenize(texts)).all()
    t = clip.tokenize(texts)
    assert t.shape == (4, 77) and t[0, 0] == 49406 and (t.argmax(-1) == (t == 49407).int().argmax(-1)).all()
    assert (t[1, 1:5] == 343).all()
    with pytest.raises(RuntimeError):
        clip.tokenize(["word " * 100])


def test_shard_range_covers_the_pool_in_order():
    import grip_amd  # noqa: F401
    from grip_amd import dist
    for n in (0, 1, 7, 50000, 50001):
        for ws in (1, 2, 3, 8):
            got = []
            per0 = None
            for r in range(ws):
                lo, hi, per = dist.shard_range(n, r, ws)
                per0 = per if per0 is None else per0
                assert per == per0 and hi - lo <= per
                got += list(range(lo, hi))
            assert got == list(range(n))


def test_fpl_row_weights_reproduce_the_three_reference_losses(golden_small):
    """sum_i w_i CE_i with grip_amd.steps.fpl_row_weights == the reference FPL losses (golden G7)."""
    import grip_amd  # noqa: F401
    from grip_amd.steps import fpl_row_weights
    g = golden_small
    logits = torch.from_numpy(g["g7.logits"])
    labels = torch.tensor([0, 3, 1, 4, 2, 3])
    ce = torch.nn.functional.cross_entropy(logits, labels, reduction="none")
    unl = [True, False, True, True, False, True]
    w = fpl_row_weights(unl, gamma_seen=4 / 2, gamma_pseudo=1.0)                     # SSL: gamma = |unseen| / |seen|
    np.testing.assert_allclose((w * ce).sum().numpy(), g["g7.ssl"], rtol=1e-6)
    is_unseen = [int(l) in (3, 4) for l in labels]
    w = fpl_row_weights(is_unseen, gamma_seen=1.0, gamma_pseudo=3 / 3)               # TRZSL: CE(seen) + gamma CE(unseen)
    np.testing.assert_allclose((w * ce).sum().numpy(), g["g7.trzsl"], rtol=1e-6)
    w = fpl_row_weights([False] * 6)                                                  # UL: plain mean
    np.testing.assert_allclose((w * ce).sum().numpy(), g["g7.ul"], rtol=1e-6)


def _cases():
    with open(os.path.join(REPO, "tests", "golden", "leaderboard.json")) as f:
        return json.load(f)["cases"]


@pytest.mark.parametrize("case", _cases(), ids=lambda c: c["name"])
def test_product_leaderboard_matches_reference_outputs(case):
    import grip_amd  # noqa: F401
    from grip_amd import pseudolabels as pl
    n = len(case["paths"])
    probs = np.frombuffer(bytes.fromhex(case["probs_f32_hex"]), dtype=np.float32).reshape(n, -1)
    fp, lab = pl.leaderboard(probs, np.array(case["pred"]), case["paths"], case["label_ids"], case["k"])
    assert fp == case["filepaths"] and lab == case["labels"]


@pytest.mark.parametrize("n,c,k,alpha,seed", [(1, 3, 2, 1.0, 0), (64, 2, 1, 0.2, 1), (500, 13, 7, 0.5, 2), (2000, 47, 16, 0.05, 3),
                                               (3000, 10, 3000, 1.0, 4), (1500, 102, 16, 30.0, 5)])
def test_product_leaderboard_matches_oracles_on_random_pools(n, c, k, alpha, seed):
    """Randomised: peaked and flat rows, quantised scores (exact ties), duplicate paths, k >= n."""
    import grip_amd  # noqa: F401
    from grip_amd import pseudolabels as pl
    from oracle import cbind, leaderboard as LB
    g = np.random.RandomState(seed)
    probs = g.dirichlet(np.ones(c) * alpha, size=n).astype(np.float32)
    probs[::2] = np.round(probs[::2] * 16) / 16
    paths = [f"root/{g.randint(0, n // 2 + 1):05d}.jpg" for _ in range(n)]        # duplicates on purpose
    pred = torch.from_numpy(probs).argmax(dim=1).numpy()
    ids = [int(v) for v in g.permutation(c)]
    want = cbind.leaderboard_ref(probs, pred, paths, ids, k)
    got = pl.leaderboard(probs, pred, paths, ids, k)
    assert got == want
    if n <= 600:
        assert got == LB.leaderboard_scan(probs, pred, paths, ids, k)


def test_leaderboard_at_full_size_matches_the_c_oracle():
    """BASELINE.json size (50 000 x 102, k = 16) + properties that hold for any input."""
    import grip_amd  # noqa: F401
    from grip_amd import engine, pseudolabels as pl
    from oracle import cbind
    g = np.random.RandomState(9)
    n, c, k = 50000, 102, 16
    logits = g.randn(n, c).astype(np.float32) * 4
    probs = torch.from_numpy(logits).softmax(-1).numpy()
    pred = probs.argmax(1)
    paths = [f"p/{i:07d}.jpg" for i in range(n)]
    img, cls = engine.leaderboard_scan(probs, pred, pl.path_ranks(paths), k)
    assert len(img) <= c * k
    for j in range(c):
        rows = img[cls == j]
        assert len(rows) <= k and len(set(rows.tolist())) == len(rows)
    assert pl.leaderboard(probs, pred, paths, list(range(c)), k) == cbind.leaderboard_ref(probs, pred, paths, list(range(c)), k)


def test_empty_pool_and_k_all():
    import grip_amd  # noqa: F401
    from grip_amd import pseudolabels as pl
    assert pl.leaderboard(np.zeros((0, 4), np.float32), np.zeros(0, np.int32), [], [0, 1, 2, 3], 5) == ([], [])
    probs = np.array([[0.1, 0.9], [0.6, 0.4]], np.float32)
    assert pl.leaderboard(probs, np.array([1, 0]), ["a", "b"], [7, 9], pl.K_ALL) == (["a", "b"], [9, 7])


def test_reference_prompt_string_quirks():
    """utils/clip_pseudolabels.py:24 concatenates (literal '{}' stays); CoOp prompts are 'X .. X name'."""
    import grip_amd  # noqa: F401
    src = open(os.path.join(REPO, "menghini-neurips23-code_amd", "utils", "clip_pseudolabels.py")).read()
    assert "f\"{template}{' '.join(i.split('_'))}\"" in src
    from grip_amd.models import CustomTextEncoder
    assert CustomTextEncoder.forward.__code__.co_varnames[:4] == ("self", "class_embeddings", "classes", "enable_pos_emb")


def test_result_helpers_follow_the_reference_schema(tmp_path, monkeypatch):
    """evaluate_predictions / store_results (utils/compute_metrics.py:18-103 of the reference): same return tuples and the same
    JSON-lines file."""
    import json
    import types

    import pandas as pd

    import grip_amd  # noqa: F401
    from grip_amd.utils import evaluate_predictions, store_results
    monkeypatch.chdir(tmp_path)
    files = [f"/d/{i}.jpg" for i in range(6)]
    truth = ["a", "a", "b", "b", "c", "c"]
    df = pd.DataFrame({"id": [f"{i}.jpg" for i in range(6)], "class": ["a", "b", "b", "b", "c", "a"]})
    ssl = types.SimpleNamespace(LEARNING_PARADIGM="ssl", MODEL="textual_prompt", LR=0.1)
    acc = evaluate_predictions(ssl, df, files, truth, ["a", "b", "c"], ["a", "b", "c"])
    assert acc == (4 / 6, None, None)
    tz = types.SimpleNamespace(LEARNING_PARADIGM="trzsl", MODEL="grip_textual", LR=0.1)
    ua, sa, hm = evaluate_predictions(tz, df, files, truth, ["c"], ["a", "b"])
    assert (ua, sa) == (0.5, 0.75) and abs(hm - 2 * 0.5 * 0.75 / 1.25) < 1e-12
    store_results(ssl, acc)
    store_results(ssl, acc)
    store_results(tz, (ua, sa, hm))
    lines = [json.loads(l) for l in open("results_model_textual_prompt.json")]
    assert len(lines) == 2 and lines[0]["accuracy"] == 4 / 6 and lines[0]["model"] == "textual_prompt" and lines[0]["config"]["LR"] == 0.1
    z = json.loads(open("results_model_grip_textual.json").read())
    assert set(z) == {"model", "config", "harmonic_mean", "seen_accuracy", "unseen_accuracy"} and z["unseen_accuracy"] == 0.5


def test_fpl_balance_parameter_for_every_paradigm_and_modality():
    """gamma of the FPL loss, 3 modalities x 3 paradigms: ssl |unseen| / |seen| (semi_supervised_learning/
    textual_fpl.py:115, visual_fpl.py:110), trzsl |seen| / |unseen| (transductive_zsl/textual_fpl.py:109, visual_fpl.py:105),
    square root of either for the multimodal strategies (semi_supervised_learning/multimodal_fpl.py:107,
    transductive_zsl/multimodal_fpl.py:104), none for ul (unsupervised_learning/*_fpl.py: plain CE)."""
    import math
    import types

    import grip_amd  # noqa: F401
    from grip_amd.methods.training_strategies import TrainingStrategy
    n_u, n_s = 32, 8     # pseudolabeled vs labeled images
    want = {("ssl", "text"): n_u / n_s, ("ssl", "image"): n_u / n_s, ("ssl", "multi"): math.sqrt(n_u / n_s),
            ("trzsl", "text"): n_s / n_u, ("trzsl", "image"): n_s / n_u, ("trzsl", "multi"): math.sqrt(n_s / n_u),
            ("ul", "text"): 1.0, ("ul", "image"): 1.0, ("ul", "multi"): 1.0}
    for (paradigm, modality), gamma in want.items():
        me = object.__new__(TrainingStrategy)      # no GPU: only the host-side merge logic is exercised
        me.config = types.SimpleNamespace(N_PSEUDOSHOTS=2, validation_seed=0, ratio_train_val=0.8)
        me.paradigm, me.modality, me.label_to_idx = paradigm, modality, {"a": 0, "b": 1}
        train = types.SimpleNamespace(filepaths=[f"s{i}.jpg" for i in range(n_s)], labels=["a"] * n_s, label_id=False)
        unl = types.SimpleNamespace(filepaths=[f"u{i}.jpg" for i in range(n_u)], labels=[1] * n_u)
        me.merge_pseudolabels(train, unl)
        assert me.balance_param == pytest.approx(gamma), (paradigm, modality, me.balance_param)
        assert len(train.filepaths) == (n_u if paradigm == "ul" else n_u + n_s) and train.label_id is True


def test_on_disk_names_and_schemas_of_the_grip_artefacts(tmp_path, monkeypatch):
    """utils/compute_metrics.py:105-171 of the reference: file names (incl. `_opt_{OPTIM_SEED}` in the pseudolabel file,
    :152) and the eight positional UPT files `{base}_{name}.pt|.pickle` (:119-143).  The test writes with the product's
    functions and reads the files back the way the reference's notebooks do (plain pickle.load / torch.load on the
    reference's own f-string names), then the other way round through load_parameters."""
    import pickle
    import types

    import numpy as np
    import torch

    import grip_amd  # noqa: F401
    from grip_amd.utils import load_parameters, save_parameters, save_predictions, save_pseudo_labels
    monkeypatch.chdir(tmp_path)
    c = types.SimpleNamespace(DATASET_NAME="DTD", LEARNING_PARADIGM="trzsl", MODEL="grip_multimodal", VIS_ENCODER="ViT-B/16",
                              OPTIM_SEED=3, SPLIT_SEED=500, MODALITY="multi")
    enc = c.VIS_ENCODER.replace("/", "")
    # pseudolabels (:150-154)
    fn = save_pseudo_labels(["a.jpg", "b.jpg"], [4, 7], c, 2)
    ref_name = f"pseudolabels/{c.DATASET_NAME}_{c.LEARNING_PARADIGM}_{c.MODEL}_{enc}_iter_2_opt_{c.OPTIM_SEED}_spl_{c.SPLIT_SEED}.pickle"
    assert fn == ref_name == "pseudolabels/DTD_trzsl_grip_multimodal_ViT-B16_iter_2_opt_3_spl_500.pickle"
    assert pickle.load(open(ref_name, "rb")) == {"filepaths": ["a.jpg", "b.jpg"], "labels": [4, 7]}
    c2 = types.SimpleNamespace(**{**c.__dict__, "OPTIM_SEED": 4})
    assert save_pseudo_labels([], [], c2, 2) != fn            # another optimisation seed does not overwrite it
    # UPT parameters (:105-143): eight positional pieces, five torch-saved state_dicts + three pickled arrays
    lin = torch.nn.Linear(4, 3)
    obj = [{"w": torch.ones(2)}, lin.state_dict(), lin.state_dict(), lin.state_dict(), lin.state_dict(),
           np.ones((1, 4, 8), np.float32), None, np.zeros((1, 4, 16), np.float32)]
    files = save_parameters(obj, c, iteration=2)
    base = f"trained_prompts/{c.DATASET_NAME}_{c.LEARNING_PARADIGM}_{c.MODEL}_{enc}_iter_2_opt_{c.OPTIM_SEED}_spl_{c.SPLIT_SEED}"
    names = ["transformer", "proj_coop_pre", "proj_coop_post", "proj_vpt_pre", "proj_vpt_post", "coop_embeddings", "deep_vpt", "vpt_embeddings"]
    assert files == [f"{base}_{n}.pt" for n in names[:5]] + [f"{base}_{n}.pickle" for n in names[5:]]
    assert torch.equal(torch.load(f"{base}_proj_vpt_post.pt")["weight"], lin.weight)
    assert pickle.load(open(f"{base}_deep_vpt.pickle", "rb")) is None
    assert pickle.load(open(f"{base}_vpt_embeddings.pickle", "rb")).shape == (1, 4, 16)
    back = load_parameters(c, iteration=2)
    assert len(back) == 8 and torch.equal(back[0]["w"], torch.ones(2)) and back[6] is None and back[5].shape == (1, 4, 8)
    # textual / visual prompt (:144-147) and predictions (:157-171): one pickle each, no iteration tag when iteration is None
    t = types.SimpleNamespace(**{**c.__dict__, "MODEL": "textual_prompt", "MODALITY": "text"})
    fn = save_parameters([np.ones((1, 16, 512), np.float32)], t)
    assert fn == "trained_prompts/DTD_trzsl_textual_prompt_ViT-B16_opt_3_spl_500.pickle"
    assert pickle.load(open(fn, "rb"))[0].shape == (1, 16, 512) and load_parameters(t)[0].shape == (1, 16, 512)
    fn = save_predictions({"images": ["a"], "predictions": ["x"], "labels": ["x"], "logits": torch.zeros(1, 2)}, t, iteration=5)
    assert fn == "evaluation/DTD_trzsl_textual_prompt_ViT-B16_iter_5_opt_3_spl_500.pickle"
    assert set(pickle.load(open(fn, "rb"))) == {"images", "predictions", "labels", "logits"}


def test_bench_quotes_pmc_traffic_only_for_the_exact_instantiation(tmp_path, monkeypatch):
    """bench.py's roofline.traffic / mfma_util_pmc come from a committed PMC summary; they are quoted only when that file names exactly
    the kernel instantiation that ran (a file lookup can go stale against the kernel it describes)."""
    import json as _json

    import bench
    slot = 6 * 16 + 9                                                   # persistent GEMM, EPI_BIAS_RESID_STATS
    monkeypatch.delenv("GRIP_GEMM_EMODE", raising=False)
    monkeypatch.delenv("GRIP_GEMM_SD", raising=False)
    assert bench.full_kernel_name(slot) == "gemm_k64p_kernel<9, 1, true>"
    assert bench.full_kernel_name(6 * 16 + 7) == "gemm_k64p_kernel<7, 4, true>" and bench.full_kernel_name(6 * 16 + 8) == "gemm_k64p_kernel<8, 2, true>"
    monkeypatch.setenv("GRIP_GEMM_EMODE", "111")
    assert bench.full_kernel_name(6 * 16 + 7) == "gemm_k64p_kernel<7, 1, true>"
    monkeypatch.setenv("GRIP_GEMM_SD", "0")
    assert bench.full_kernel_name(slot) == "gemm_k64p_kernel<9, 1>"
    monkeypatch.delenv("GRIP_GEMM_EMODE")
    monkeypatch.delenv("GRIP_GEMM_SD")
    (tmp_path / "profiles").mkdir()
    f = tmp_path / "profiles" / "t.json"
    f.write_text(_json.dumps({"kernels": {"gemm_k64p_kernel<9, 1, true>": {"bytes_per_launch": 2.5e9, "mfma_util": 0.5},
                                          "gemm_k64p_kernel<8, 1, true>": {"bytes_per_launch": 1.0, "mfma_util": 0.1}}}))
    monkeypatch.setattr(bench, "REPO", str(tmp_path))
    monkeypatch.setattr(bench, "TRAFFIC_FILE", os.path.join("profiles", "t.json"))
    assert bench.pmc_entry(slot) == (2.5e9, 0.5)
    assert bench.pmc_entry(6 * 16 + 8) == (None, None)                  # the file holds <8, 1, true>, the run uses <8, 2, true>
    assert bench.pmc_entry(0 * 16 + 3) == (None, None)                  # the f32 GEMM is not in the file


def test_grip_schedule_matches_the_reference_formula():
    """TrainingStrategy._n_pseudoshots against the literal arithmetic of methods/semi_supervised_learning/pseudo_iterative.py:62-75 (first
    iteration) and :113-125 (ALL_UNLABELED growth; identical in the ul / trzsl twins): pseudo-shots per class at GRIP iteration 1 .. num_iter."""
    import math

    import grip_amd  # noqa: F401
    from grip_amd.methods.training_strategies import TrainingStrategy

    def reference_schedule(n_unlabeled, step_quantile, n_unseen):
        num_iter = int(100 / step_quantile)                                   # :63
        num_samples = int(n_unlabeled / num_iter)                             # :64
        n_per_class = int(num_samples / n_unseen)                             # :66
        shots = [n_per_class if n_per_class * n_unseen <= n_unlabeled else math.floor(n_unlabeled / n_unseen)]      # :68-75
        for niter in range(1, num_iter):                                      # the value set at the END of iteration niter is used by iteration niter + 1
            n_per_class = int((niter + 1) * num_samples / n_unseen)           # :114
            shots.append(n_per_class if n_per_class * n_unseen <= n_unlabeled else math.floor(n_unlabeled / n_unseen))   # :116-125
        return num_iter, num_samples, shots

    for n_unlabeled, q, c in [(3120, 10, 47), (50000, 10, 102), (1000, 10, 38), (999, 20, 10), (57, 10, 18), (10, 50, 3), (7, 10, 2), (31500, 10, 45),
                              (8144, 25, 196), (123, 10, 100), (5000, 1, 7)]:
        num_iter, num_samples, want = reference_schedule(n_unlabeled, q, c)
        got = [TrainingStrategy._n_pseudoshots(None, niter, num_samples, n_unlabeled, c) for niter in range(1, num_iter + 1)]
        assert got == want, (n_unlabeled, q, c, got, want)


def test_cooperative_split_factor_choice():
    """Host-side shape logic of the cooperative split-K GEMM (csrc/gemm.hip, gemm_pick_coop_split; no GPU needed): used for a few dozen 64 x 128 tiles walking
    >= 16 K slices (the text tower's c_proj in a prompt step), never for launches that fill the chip on their own."""
    import grip_amd  # noqa: F401
    from grip_amd import native
    lib = native.lib()
    f = lib.grip_debug_coop_split
    assert f(425, 512, 2048) == 4          # CoOp step: 28 tiles x 32 slices -> 4 x 8, 128 workgroups on one XCD each
    assert f(240, 512, 2048) == 4          # UPT text side
    assert f(425, 768, 3072) == 4          # ViT-L/14's text tower
    assert f(2142, 512, 2048) == 1         # plain row layout, 102 x 21 rows: 136 tiles are a launch of their own
    assert f(425, 512, 512) == 1           # 8 slices: nothing to split
    assert f(3408, 768, 3072) == 1         # the image tower's prompt step
    assert f(425, 500, 2048) == 1          # N % 128 != 0


def test_lookahead_image_features_groups_and_order():
    """steps.lookahead_image_features on the host path (no GPU: the sequential form): every batch comes back once, in order, with its own slice of the
    group's features and its pass-through elements, for groups that do and do not divide the batch count and a ragged last batch."""
    import types
    import torch
    import grip_amd  # noqa: F401
    from grip_amd import steps
    calls = []
    model = types.SimpleNamespace(encode_image=lambda x: (calls.append(len(x)), x.reshape(len(x), -1)[:, :3] * 2.0)[1])
    sizes = [4, 4, 4, 4, 4, 4, 3]
    data = [(torch.full((n, 1, 2, 2), float(i)), i, f"b{i}") for i, n in enumerate(sizes)]
    for group in (1, 2, 3, 7, 50):
        calls.clear()
        got = list(steps.lookahead_image_features(model, iter(data), group))
        assert [g[1:] for g in got] == [(i, f"b{i}") for i in range(len(sizes))]
        for (f, i, _), n in zip(got, sizes):
            assert f.shape == (n, 3) and torch.all(f == 2.0 * i)
        assert sum(calls) == sum(sizes) and len(calls) == -(-len(sizes) // group)
    assert list(steps.lookahead_image_features(model, iter([]), 4)) == []


def test_stress_weights_are_well_conditioned_and_keep_gemm_operands_in_f16_range():
    """weights.stress_state_dict (the model of tests/test_gpu_stress.py and bench.py's stress secondary) at the `small` dimensions: the fp32 oracle agrees with
    an fp64 evaluation of the same weights (a stress model fp32 itself cannot evaluate would test nothing -- the first, gain-based recipe was off by 3e-2 in
    cosine at ViT-B/16), the outlier channels really sit near +200 in the residual stream, and every GEMM operand stays inside what the split-f16 tower can
    hold (|w| x 2^8 < 65 504)."""
    import numpy as np
    import torch
    import grip_amd  # noqa: F401
    from grip_amd import config as gcfg, weights
    from grip_amd.data.synthetic import structured_images
    from oracle.clip import model as OM
    d = gcfg.get_dims("small")
    sd = weights.stress_state_dict(d, 0)
    big = max(float(np.abs(v).max()) for k, v in sd.items() if k.startswith("visual.") and k.endswith(("in_proj_weight", "out_proj.weight", "c_fc.weight", "c_proj.weight")))
    assert big * 256 < 65504, big
    m = OM.CLIP(d.embed_dim, d.image_resolution, d.vision_layers, d.vision_width, d.vision_patch_size, d.context_length, d.vocab_size, d.transformer_width,
                d.transformer_heads, d.transformer_layers)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    m = m.float().eval()
    x = structured_images(77, 0, 8, d.image_resolution)
    seen = {}
    h = m.visual.ln_pre.register_forward_hook(lambda mod, i, o: seen.__setitem__("x", o))
    with torch.no_grad():
        e32 = m.encode_image(x)
    h.remove()
    ch = sorted({c % d.vision_width for c in weights.STRESS_OUTLIER_CHANNELS})
    assert float(seen["x"][..., ch].mean()) > 150 and float(seen["x"][..., ch].std()) < 40
    keep = OM.LayerNorm.forward
    OM.LayerNorm.forward = torch.nn.LayerNorm.forward          # (CLIP's LayerNorm pins fp32)
    try:
        with torch.no_grad():
            e64 = m.double().encode_image(x.double()).float()
    finally:
        OM.LayerNorm.forward = keep
    cos = torch.nn.functional.cosine_similarity(e32, e64, dim=1)
    assert float((1 - cos).max()) < 1e-5, float((1 - cos).max())


def test_screen_stream_is_picked_per_pool_from_the_last_pass(monkeypatch):
    """pseudolabels.screen_stream("auto"): the first pass over a pool screens with the compensated stream; a compensated pass that marked little sends the
    next pass over the SAME pool to the plain stream, a plain pass that marked a lot back to the compensated one (two thresholds: the same break-even seen
    from either side); calibration, audit and non-finite rows do not count; explicit settings win; other pools are not affected."""
    import grip_amd  # noqa: F401
    from grip_amd import pseudolabels as pl
    monkeypatch.delenv("GRIP_SCREEN_STREAM", raising=False)
    pl._SCREEN_CHOICE.clear()
    key, other = ("tower", 50000, 102), ("tower", 2000, 10)
    assert pl.screen_stream(key) == "hilo" and pl.screen_stream() == "hilo"
    st = {"rows": 50000, "rows_refined": 3189, "calibration_rows": 256, "audit_rows": 1024, "nonfinite_screen_rows": 0}
    pl.note_screen_bound(key, "hilo", st)                         # 1 909 marked rows = 3.8 % < 4 %: the plain screen is cheaper here (the timed bench pool)
    assert pl.screen_stream(key) == "f16" and st["screen_stream"] == "hilo" and st["screen_stream_next_pass"] == "f16" and abs(st["screen_marked_share"] - 0.03818) < 1e-4
    assert pl.screen_stream(other) == "hilo"
    pl.note_screen_bound(key, "f16", dict(st, rows_refined=3989))     # 5.4 % marked by the plain screen: stays plain (below 6.4 %)
    assert pl.screen_stream(key) == "f16"
    pl.note_screen_bound(key, "f16", dict(st, rows_refined=5586))     # the structured pool: 8.6 % -> compensated
    assert pl.screen_stream(key) == "hilo"
    pl.note_screen_bound(key, "hilo", dict(st, rows_refined=3671))    # ... where 4.8 % stay marked: keeps it
    assert pl.screen_stream(key) == "hilo"
    pl.note_screen_bound(key, "hilo", dict(st, rows_refined=11500, nonfinite_screen_rows=9900))      # overflowed rows are re-encoded either way: 0.6 % marked
    assert pl.screen_stream(key) == "f16"
    pl.note_screen_bound(key, "hilo", {"rows": 0})                    # (an empty pool leaves the choice alone)
    assert pl.screen_stream(key) == "f16"
    monkeypatch.setenv("GRIP_SCREEN_STREAM", "hilo")
    assert pl.screen_stream(key) == "hilo" and pl.screen_stream(other) == "hilo"
    monkeypatch.setenv("GRIP_SCREEN_STREAM", "bogus")
    import pytest
    with pytest.raises(ValueError):
        pl.screen_stream(key)
    pl._SCREEN_CHOICE.clear()


def test_balanced_chunk_keeps_the_launch_count_and_evens_the_sizes():
    """pseudolabels.balanced_chunk: the refinement tiers' rows per launch (rows are chunk-independent: tests/test_gpu_exact.py)."""
    from grip_amd.pseudolabels import balanced_chunk
    for n, chunk in ((1024, 880), (2445, 880), (880, 880), (881, 880), (2, 880), (1, 1), (0, 880), (50000, 1320), (7, 3)):
        c = balanced_chunk(n, chunk)
        assert 1 <= c <= max(chunk, 1)
        launches = -(-n // chunk) if n else 0
        assert (-(-n // c) if n else 0) == launches, (n, chunk, c)                # never more launches than the plain chunking
        if n:
            sizes = [min(c, n - s) for s in range(0, n, c)]
            assert max(sizes) - min(sizes) <= launches, (n, chunk, sizes)          # ... and no short tail
    assert balanced_chunk(1024, 880) == 512 and balanced_chunk(2445, 880) == 815
