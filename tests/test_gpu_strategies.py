"""The training-strategy stand-in and the run_main_* entry points drive the native engine end to end on a
synthetic, class-structured pool (small towers, a few epochs): every MODEL of the reference's dispatch runs,
prompt tuning fits the labeled shots, and GRIP's schedule grows the pseudolabel budget."""
import json
import os
import subprocess
import sys

import pytest
import torch

from conftest import REPO

pytestmark = pytest.mark.gpu


def _conf(**kw):
    import grip_amd  # noqa: F401
    from grip_amd.methods.main import DEFAULTS, Config
    c = dict(DEFAULTS)
    c.update(OPTIM_SEED=1, VIS_ENCODER="small", DATASET_NAME="Synthetic", SPLIT_SEED=500, DATASET_DIR="", EPOCHS=4, WARMUP_EPOCHS=1,
             N_PSEUDOSHOTS=4, STEP_QUANTILE=50, LR=0.05, PREFIX_SIZE=4)
    c.update(kw)
    return Config(**c)


@pytest.mark.parametrize("paradigm,model", [("ssl", "textual_prompt"), ("ssl", "visual_fpl"), ("ul", "visual_fpl"), ("ul", "textual_fpl"),
                                            ("trzsl", "multimodal_fpl"), ("trzsl", "grip_textual"), ("ssl", "iterative_visual_fpl"),
                                            ("trzsl", "multimodal_prompt")])
def test_every_strategy_runs_and_learns(tmp_path, monkeypatch, paradigm, model):
    import grip_amd  # noqa: F401
    from grip_amd.methods.main import workflow
    monkeypatch.chdir(tmp_path)
    conf = _conf(MODEL=model, LEARNING_PARADIGM=paradigm)
    res = workflow(conf, "cuda", n_synth=24, n_classes=6)
    assert 0.0 <= res["test_accuracy"] <= 1.0 and res["n_test"] > 0
    assert res["val_accuracy"] >= 0.0
    if paradigm == "trzsl":
        assert {"seen_accuracy", "unseen_accuracy", "harmonic_mean"} <= set(res)


def test_prompt_tuning_fits_the_training_shots(tmp_path, monkeypatch):
    import grip_amd  # noqa: F401
    from grip_amd.data import TensorPoolDataset
    from grip_amd.methods import VisualPrompt
    from grip_amd.methods.main import synthetic_pool
    monkeypatch.chdir(tmp_path)
    conf = _conf(MODEL="visual_prompt", LEARNING_PARADIGM="ssl", EPOCHS=12, LR=0.2)
    classes, files, images, names = synthetic_pool(4, 8, 64, 3)
    l2i = {c: i for i, c in enumerate(classes)}
    data = TensorPoolDataset(files, images.cuda(), labels=names, label_map=l2i)
    m = VisualPrompt(conf, l2i, classes, classes, classes, "cuda")
    m.define_model(classes)
    loader = m._loader(data, True)
    first = m._train_epoch(loader)[0]
    for _ in range(10):
        last = m._train_epoch(loader)[0]
    assert last < first, (first, last)      # the CE on the shots goes down: gradients reach the prompt and SGD applies them


def test_run_main_entry_point(tmp_path):
    env = dict(os.environ, VIS_ENCODER="small", MODEL="textual_fpl", EPOCHS="2", N_PSEUDOSHOTS="3", PYTHONPATH=REPO)
    out = subprocess.run([sys.executable, os.path.join(REPO, "run_main_ul.py"), "--synthetic", "12", "--classes", "4"], cwd=tmp_path, env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    res = json.loads(out.stdout.strip().splitlines()[-1])
    assert res["paradigm"] == "ul" and res["model"] == "textual_fpl"
    assert os.path.exists(tmp_path / "results" / "results_model_textual_fpl.json")
    # on-disk artefacts with the reference's schemas (utils/compute_metrics.py:105-171, clip_pseudolabels.py:114)
    import glob
    import pickle
    (pp,) = glob.glob(str(tmp_path / "trained_prompts" / "*.pickle"))
    prompts = pickle.load(open(pp, "rb"))
    assert isinstance(prompts, list) and prompts[0].shape == (1, 16, 256)
    (ev,) = glob.glob(str(tmp_path / "evaluation" / "*.pickle"))
    assert set(pickle.load(open(ev, "rb"))) == {"images", "predictions", "labels", "logits"}
    (pl,) = glob.glob(str(tmp_path / "pseudolabels" / "*.pickle"))
    assert set(pickle.load(open(pl, "rb"))) == {"filepaths", "labels"}


def test_frozen_feature_cache_changes_nothing(tmp_path, monkeypatch):
    """Textual strategies cache the frozen tower's image features across epochs: same prompts, bit for bit,
    as re-encoding the images every epoch, and the tower is entered once per image instead of once per image per epoch."""
    import grip_amd  # noqa: F401
    from grip_amd.data import TensorPoolDataset
    from grip_amd.methods import TextualPrompt
    from grip_amd.methods.main import synthetic_pool
    monkeypatch.chdir(tmp_path)
    classes, files, images, names = synthetic_pool(4, 8, 64, 3)
    l2i = {c: i for i, c in enumerate(classes)}
    out = []
    for cache in (True, False):
        conf = _conf(MODEL="textual_prompt", LEARNING_PARADIGM="ssl", EPOCHS=3, LR=0.1, CACHE_FROZEN_FEATURES=cache)
        data = TensorPoolDataset(files, images.cuda(), labels=names, label_map=l2i)
        m = TextualPrompt(conf, l2i, classes, classes, classes, "cuda")
        calls = []
        enc = m.clip_model.encode_image
        monkeypatch.setattr(m.clip_model, "encode_image", lambda x, _e=enc, _c=calls: (_c.append(len(x)), _e(x))[1])
        m.define_model(classes)
        loader = m._loader(data, True)
        losses = [m._train_epoch(loader)[0] for _ in range(3)]
        out.append((losses, m.unwrap_model().prefix.detach().clone(), sum(calls)))
    (l_c, p_c, n_c), (l_r, p_r, n_r) = out
    assert l_c == l_r and torch.equal(p_c, p_r)
    assert n_c == len(files) and n_r == 3 * len(files)


def test_run_main_clip_baseline(tmp_path):
    """run_main_clip.py: zero-shot CLIP (methods/clip_baseline.py) through the same result / evaluation files."""
    env = dict(os.environ, VIS_ENCODER="small", PYTHONPATH=REPO)
    out = subprocess.run([sys.executable, os.path.join(REPO, "run_main_clip.py"), "--synthetic", "12", "--classes", "4"], cwd=tmp_path, env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    res = json.loads(out.stdout.strip().splitlines()[-1])
    assert res["model"] == "clip_baseline" and 0.0 <= res["test_accuracy"] <= 1.0 and res["n_test"] > 0
    line = json.loads(open(tmp_path / "results_model_clip_baseline.json").read().splitlines()[0])
    assert set(line) == {"model", "config", "accuracy"} and abs(line["accuracy"] - res["test_accuracy"]) < 1e-9


@pytest.mark.parametrize("cls_name,paradigm", [("TextualFPL", "ssl"), ("VisualFPL", "ul"), ("MultimodalFPL", "trzsl")])
def test_assign_pseudo_labels_uses_the_sharded_pool_encode(tmp_path, monkeypatch, cls_name, paradigm):
    """GRIP's trained-model pass (the nine assign_pseudo_labels, e.g. transductive_zsl/multimodal_fpl.py:194-285): the pool
    goes through pseudolabels.encode_pool (chunked + sharded, trained visual prompt included; mixer and text tower once per
    call) and gives bit-identical features, hence identical lists, to the per-batch path the training loop uses."""
    import grip_amd  # noqa: F401
    from grip_amd import methods, pseudolabels as pl
    from grip_amd.data import TensorPoolDataset
    from grip_amd.methods.main import synthetic_pool
    monkeypatch.chdir(tmp_path)
    classes, files, images, names = synthetic_pool(5, 9, 64, 11)
    l2i = {c: i for i, c in enumerate(classes)}
    conf = _conf(MODEL="x", LEARNING_PARADIGM=paradigm, TEXT_PREFIX_SIZE=4, VISION_PREFIX_SIZE=4)
    seen, unseen = (classes[:3], classes[3:]) if paradigm == "trzsl" else (classes, classes)
    m = getattr(methods, cls_name)(conf, l2i, "", classes, seen, unseen, "cuda")
    m.define_model(classes)
    with torch.no_grad():    # move the prompts away from their init so a dropped prefix would show
        for p in m.model.parameters():
            if p.requires_grad:
                p.add_(torch.randn_like(p) * 0.05)
    data = TensorPoolDataset(files, images.cuda(), labels=None, label_map=l2i)
    target = unseen if paradigm == "trzsl" else classes
    calls = []
    real = pl.encode_pool
    monkeypatch.setattr(pl, "encode_pool", lambda *a, **k: (calls.append(k.get("prefix")), real(*a, **k))[1])
    img, txt = m.trained_features(data.images, target, chunk=16)
    assert len(calls) == 1 and (calls[0] is None) == (cls_name == "TextualFPL")
    with torch.no_grad():
        per_batch = [m.features(data.images[i:i + 16], target) for i in range(0, len(files), 16)]
    assert torch.equal(img, torch.cat([f[0] for f in per_batch])) and torch.equal(txt, per_batch[0][1])
    want = pl.pseudolabel_from_features(img, txt, m.scale(), list(data.filepaths), [l2i[c] for c in target], 3, argmax_on="logits")
    out = m.assign_pseudo_labels(3, data)
    assert (out.filepaths, out.labels) == want and out.label_id is True and len(want[0]) > 0


def test_graphed_coop_step_equals_eager():
    """steps.GraphedCoopStep (forward + backward replayed from a HIP graph) follows exactly the trajectory of the eager
    coop_step: same losses and the same prompt after several SGD steps on changing batches."""
    import grip_amd  # noqa: F401
    from grip_amd import clip, rng, steps
    from grip_amd.models import CustomTextEncoder, TextPrefixModel
    m, _ = clip.load("small", device="cuda")
    classes = [f"class {i}" for i in range(7)]
    g = torch.Generator(device="cuda").manual_seed(0)
    xs = [torch.randn(8, 3, 64, 64, device="cuda", generator=g) for _ in range(5)]
    ys = [torch.randint(0, 7, (8,), device="cuda", generator=g, dtype=torch.int32) for _ in range(5)]
    w = torch.full((8,), 1 / 8, device="cuda")
    out = []
    for graphed in (False, True):
        p0 = torch.from_numpy(rng.normal(3, rng.stream_id("gr.p"), (1, 4, 256), 0.0, 0.02)).cuda()
        tm = TextPrefixModel(p0, CustomTextEncoder(m, "cuda", torch.float32), classes, device="cuda")
        opt = torch.optim.SGD([tm.prefix], lr=0.1, weight_decay=0.1)
        step = steps.GraphedCoopStep(tm, m, opt) if graphed else (lambda x, y, ww, _tm=tm, _opt=opt: steps.coop_step(_tm, m, x, y, ww, _opt))
        losses = [float(step(x, y, w)) for x, y in zip(xs, ys)]
        losses.append(float(step(xs[0][:5], ys[0][:5], w[:5] * 8 / 5)))       # another batch size: the graphed step falls back to eager
        out.append((losses, tm.prefix.detach().clone()))
    (l_e, p_e), (l_g, p_g) = out
    assert l_e == l_g, (l_e, l_g)
    assert torch.equal(p_e, p_g)


@pytest.mark.parametrize("encoder,model,extra", [("ViT-L/14@336px", "grip_textual", {}), ("ViT-B/16", "grip_multimodal", {"LR": "0.01"})])
def test_baseline_configs_run_end_to_end_at_real_dimensions(tmp_path, encoder, model, extra):
    """BASELINE.json configs[4] (GRIP textual TRZSL on ViT-L/14@336px) and configs[3] (GRIP multimodal / UPT TRZSL on ViT-B/16) through
    the reference-named entry point run_main_trzsl.py at the REAL tower dimensions (tiny synthetic pool, one epoch per GRIP
    iteration): pseudolabels from frozen CLIP, prompt training, re-labelling with the trained prompts through the sharded pool
    encode, evaluation, and the reference's on-disk artefacts."""
    import glob
    import pickle
    env = dict(os.environ, VIS_ENCODER=encoder, MODEL=model, EPOCHS="1", STEP_QUANTILE="50", N_PSEUDOSHOTS="2", BATCH_SIZE="4", PYTHONPATH=REPO, **extra)
    out = subprocess.run([sys.executable, os.path.join(REPO, "run_main_trzsl.py"), "--synthetic", "4", "--classes", "5"], cwd=tmp_path, env=env,
                         capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, out.stderr[-3000:]
    res = json.loads(out.stdout.strip().splitlines()[-1])
    assert res["encoder"] == encoder and res["model"] == model and res["paradigm"] == "trzsl"
    assert {"seen_accuracy", "unseen_accuracy", "harmonic_mean"} <= set(res) and res["n_test"] > 0
    enc = encoder.replace("/", "")
    pls = sorted(glob.glob(str(tmp_path / "pseudolabels" / f"Synthetic_trzsl_{model}_{enc}_iter_*_opt_1_spl_500.pickle")))
    assert len(pls) == 2, os.listdir(tmp_path / "pseudolabels")          # STEP_QUANTILE 50 -> two GRIP iterations, reference file names
    assert set(pickle.load(open(pls[0], "rb"))) == {"filepaths", "labels"}
    prompts = glob.glob(str(tmp_path / "trained_prompts" / f"Synthetic_trzsl_{model}_{enc}_iter_2_*"))
    assert len(prompts) == (8 if model == "grip_multimodal" else 1), prompts


def test_graphed_vpt_and_upt_steps_equal_eager():
    """GraphedVptStep / GraphedUptStep against the eager steps: identical losses and parameters after several SGD steps (the UPT
    graph also holds the torch mixer and both towers on two streams)."""
    import grip_amd  # noqa: F401
    from grip_amd import clip, rng, steps
    from grip_amd.models import CustomImageEncoder, CustomTextEncoder, ImagePrefixModel, UPTModel
    m, _ = clip.load("small", device="cuda")
    classes = [f"class {i}" for i in range(6)]
    g = torch.Generator(device="cuda").manual_seed(1)
    xs = [torch.randn(8, 3, 64, 64, device="cuda", generator=g) for _ in range(4)]
    ys = [torch.randint(0, 6, (8,), device="cuda", generator=g, dtype=torch.int32) for _ in range(4)]
    w = torch.full((8,), 1 / 8, device="cuda")
    txt = m.encode_text(clip.tokenize([f"a photo of a {c}" for c in classes]).cuda())
    N = lambda name, shape: torch.from_numpy(rng.normal(4, rng.stream_id(name), shape, 0.0, 0.02)).cuda()   # noqa: E731
    res = {}
    for graphed in (False, True):
        im = ImagePrefixModel(N("gv.p", (4, 256)), CustomImageEncoder(m.visual), device="cuda")
        opt = torch.optim.SGD([im.prefix], lr=0.1, weight_decay=0.1)
        step = steps.GraphedVptStep(im, txt, 100.0, opt) if graphed else (lambda x, y, ww, _m=im, _o=opt: steps.vpt_step(_m, txt, 100.0, x, y, ww, _o))
        losses = [float(step(x, y, w)) for x, y in zip(xs, ys)]
        torch.manual_seed(7)
        um = UPTModel(N("gu.c", (1, 4, 256)), N("gu.v", (1, 4, 256)), None, CustomImageEncoder(m.visual), CustomTextEncoder(m, "cuda", torch.float32),
                      classes, 128, device="cuda", dtype=torch.float32)
        opt2 = torch.optim.SGD([p for p in um.parameters() if p.requires_grad], lr=0.01, weight_decay=0.1)
        step2 = steps.GraphedUptStep(um, 100.0, opt2) if graphed else (lambda x, y, ww, _m=um, _o=opt2: steps.upt_step(_m, 100.0, x, y, ww, _o))
        losses2 = [float(step2(x, y, w)) for x, y in zip(xs, ys)]
        res[graphed] = (losses, im.prefix.detach().clone(), losses2, um.coop_embeddings.detach().clone(), um.proj_vpt_post.weight.detach().clone())
    e, gr = res[False], res[True]
    assert e[0] == gr[0] and torch.equal(e[1], gr[1])
    assert e[2] == gr[2] and torch.equal(e[3], gr[3]) and torch.equal(e[4], gr[4])


@pytest.mark.parametrize("cls_name,paradigm", [("TextualPrompt", "ssl"), ("VisualPrompt", "ul"), ("MultimodalPrompt", "trzsl"), ("TextualFPL", "trzsl")])
def test_train_epoch_graph_replay_equals_eager_epoch(tmp_path, monkeypatch, cls_name, paradigm):
    """TrainingStrategy._train_epoch (what run_main_* executes) replays the prompt step from a HIP graph, accumulates loss and
    accuracy on the device and encodes the frozen tower several batches ahead; with GRAPH_STEPS False it runs the eager per-batch
    loop.  Same losses, accuracies and trained parameters, bit for bit -- including a ragged last batch (eager fallback inside the
    graphed step) and un-cached frozen features (look-ahead encode)."""
    import grip_amd  # noqa: F401
    from grip_amd import methods
    from grip_amd.data import TensorPoolDataset
    from grip_amd.methods.main import synthetic_pool
    monkeypatch.chdir(tmp_path)
    classes, files, images, names = synthetic_pool(5, 7, 64, 5)          # 35 images: batches of 8 + a ragged 3
    l2i = {c: i for i, c in enumerate(classes)}
    out = []
    for graph in (True, False):
        conf = _conf(MODEL="x", LEARNING_PARADIGM=paradigm, BATCH_SIZE=8, GRAPH_STEPS=graph, CACHE_FROZEN_FEATURES=False, IMAGE_LOOKAHEAD=3,
                     TEXT_PREFIX_SIZE=4, VISION_PREFIX_SIZE=4)
        data = TensorPoolDataset(files, images.cuda(), labels=names, label_map=l2i)
        m = getattr(methods, cls_name)(conf, l2i, classes, classes[:3], classes[3:], "cuda") if not cls_name.endswith("FPL") else \
            getattr(methods, cls_name)(conf, l2i, "", classes, classes[:3], classes[3:], "cuda")
        m.define_model(classes)
        loader = m._loader(data, True)
        stats = [m._train_epoch(loader) for _ in range(3)]
        out.append((stats, [p.detach().clone() for p in m.model.parameters() if p.requires_grad]))
    (s_g, p_g), (s_e, p_e) = out
    assert s_g == s_e, (s_g, s_e)
    assert all(torch.equal(a, b) for a, b in zip(p_g, p_e))


@pytest.mark.parametrize("cls_name,paradigm", [("TextualFPL", "ssl"), ("VisualFPL", "ul"), ("MultimodalFPL", "trzsl")])
def test_assign_pseudo_labels_default_mode_returns_the_fp32_lists(tmp_path, monkeypatch, cls_name, paradigm):
    """The trained-prompt pseudolabel pass of GRIP (the nine assign_pseudo_labels) in the default `identical` mode: the lists are
    those of the all-f32 computation with the SAME prompts (exact twin: f32 text tower with the trained textual prompt / mixer
    output, f32 vision tower with the trained visual prompt, arg-max on the logits), whatever the f16 towers' own lists are."""
    import grip_amd  # noqa: F401
    from grip_amd import methods, pseudolabels as pl
    from grip_amd.data import TensorPoolDataset
    from grip_amd.methods.main import synthetic_pool
    monkeypatch.chdir(tmp_path)
    monkeypatch.delenv("GRIP_PSEUDOLABEL_MODE", raising=False)
    classes, files, images, names = synthetic_pool(5, 40, 64, 17)
    l2i = {c: i for i, c in enumerate(classes)}
    conf = _conf(MODEL="x", LEARNING_PARADIGM=paradigm, TEXT_PREFIX_SIZE=4, VISION_PREFIX_SIZE=4)
    seen, unseen = (classes[:2], classes[2:]) if paradigm == "trzsl" else (classes, classes)
    m = getattr(methods, cls_name)(conf, l2i, "", classes, seen, unseen, "cuda")
    m.define_model(classes)
    with torch.no_grad():
        for p in m.model.parameters():
            if p.requires_grad:
                p.add_(torch.randn_like(p) * 0.05)
    target = unseen if paradigm == "trzsl" else classes
    twin = m.clip_model.exact_twin()
    with torch.no_grad():
        txt, vprompt = m.trained_text_features(target, twin)
        emb = pl.encode_pool(twin.visual.tower, images.cuda(), chunk=32, prefix=vprompt)
    want = pl.pseudolabel_from_features(emb, txt, m.scale(), list(files), [l2i[c] for c in target], 7, argmax_on="logits")
    data = TensorPoolDataset(files, images.cuda(), labels=None, label_map=l2i)
    pl.LAST_REFINE_STATS = None
    out = m.assign_pseudo_labels(7, data)
    assert (out.filepaths, out.labels) == want and pl.LAST_REFINE_STATS is not None and len(want[0]) > 7
    assert pl.LAST_REFINE_STATS["rows_refined"] < len(files)
