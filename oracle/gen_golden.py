"""ORACLE (test infrastructure) -- golden-vector generator.  Runs ONLY in the build container:
imports the reference's own modules UNMODIFIED from /root/reference (models/clip_encoders.py,
models/prompts_models.py, utils/clip_pseudolabels.py, and the FPL loss methods of the strategy
files) on top of the CPU oracle `clip` stand-in (oracle/clip), checks the oracle restatements
(oracle/wrappers.py, oracle/leaderboard.py, oracle/leaderboard_ref.c) against them, and writes
small fixtures (inputs regenerated from seeds; outputs stored) to tests/golden/.

    python oracle/gen_golden.py          # rewrites tests/golden/*

Nothing under tests/, bench.py or smoke() reads /root/reference at run time.
"""
import importlib.util
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
REF = "/root/reference"
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)        # `import clip` -> oracle/clip
sys.path.insert(0, REF)         # `import models`, `import utils` -> the reference

import clip  # noqa: E402  (oracle stand-in)
import grip_amd  # noqa: E402,F401
from grip_amd import rng, weights  # noqa: E402
import models as RM  # noqa: E402  (REFERENCE, unmodified)
from utils import clip_pseudolabels as RP  # noqa: E402  (REFERENCE, unmodified)

sys.path.insert(0, REPO)
from oracle import wrappers as W, leaderboard as LB, cbind  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")
SEED = 100


def T(name, shape, std=1.0, mean=0.0):
    return torch.from_numpy(rng.normal(SEED, rng.stream_id(name), shape, mean, std))


def close(a, b, what, tol=1e-5):
    err = (a - b).abs().max().item()
    assert err <= tol, f"{what}: oracle restatement differs from the reference by {err}"
    return err


# ----------------------------------------------------------------------------- G1 / G2 / G4 / G5
def towers(model_name, n_img, classes, P, tag, out, with_grad=True):
    m, _ = clip.load(model_name)
    d = grip_amd.config.get_dims(model_name)
    R = d.image_resolution
    x = T(f"{tag}.x", (n_img, 3, R, R))
    vprefix = T(f"{tag}.vprefix", (P, d.vision_width), 0.02)
    tprefix = T(f"{tag}.tprefix", (1, P, d.transformer_width), 0.02)

    ref_img = RM.CustomImageEncoder(m.visual)
    ref_txt = RM.CustomTextEncoder(m, "cpu", torch.float32)

    # vision, no prefix (== encode_image) and with prefix
    v0 = m.encode_image(x)
    close(W.vision_forward(m.visual, x, None), v0, "vision P=0")
    vp = vprefix.clone().requires_grad_(with_grad)
    v1 = ref_img(x, vp)
    close(W.vision_forward(m.visual, x, vp.detach()), v1.detach(), "vision prefix")
    out[f"{tag}.vision_p0"] = v0.detach().numpy()
    out[f"{tag}.vision_p{P}"] = v1.detach().numpy()
    if with_grad:
        (v1 ** 2).sum().backward()
        out[f"{tag}.vision_p{P}_grad_prefix"] = vp.grad.numpy()

    # text with prefix (CoOp) and plain encode_text
    tp = tprefix.clone().requires_grad_(with_grad)
    t1 = ref_txt(tp, classes)
    tok = clip.tokenize(W.coop_prompt_strings(P, classes))
    close(W.text_forward(m, tok, tp.detach()), t1.detach(), "text prefix")
    out[f"{tag}.coop_tokens"] = tok.numpy().astype(np.int32)
    out[f"{tag}.text_p{P}"] = t1.detach().numpy()
    if with_grad:
        (t1 ** 2).sum().backward()
        out[f"{tag}.text_p{P}_grad_prefix"] = tp.grad.numpy()
    ztok = clip.tokenize(W.zero_shot_prompt_strings("a photo of a {}", classes))
    t0 = RM.TextEncoder(m)(ztok)
    close(W.text_forward(m, ztok, None), t0, "text plain")
    out[f"{tag}.zs_tokens"] = ztok.numpy().astype(np.int32)
    out[f"{tag}.text_p0"] = t0.detach().numpy()

    # head (G5) on these features, via the full CLIP forward the pseudolabeler calls
    li, _ = m(x, ztok)
    lo, am = W.cosine_head(v0, t0, m.logit_scale.detach())
    close(lo, li, "head", 1e-4)
    out[f"{tag}.zs_logits"] = li.detach().numpy()
    out[f"{tag}.zs_probs"] = li.softmax(dim=-1).detach().numpy()
    return m, d, x


def upt(model_name, n_img, classes, Pm, tag, out):
    """G4: UPTModel end to end (mixer incl. the fp16 round trip, then both towers) + all gradients."""
    m, _ = clip.load(model_name)
    d = grip_amd.config.get_dims(model_name)
    x = T(f"{tag}.x", (n_img, 3, d.image_resolution, d.image_resolution))
    coop = T(f"{tag}.coop", (1, Pm, d.transformer_width), 0.02)
    vpt = T(f"{tag}.vpt", (1, Pm, d.vision_width), 0.02)
    tdim = 128
    mixer = {k: torch.from_numpy(v) for k, v in weights.init_upt_mixer(d.transformer_width, d.vision_width, tdim, SEED).items()}
    ref = RM.UPTModel(coop.clone(), vpt.clone(), None, RM.CustomImageEncoder(m.visual),
                      RM.CustomTextEncoder(m, "cpu", torch.float32), classes, tdim, device="cpu", dtype=torch.float32)
    missing, unexpected = ref.load_state_dict(mixer, strict=False)
    assert not unexpected, unexpected
    t_out, v_out = ref(x, classes)
    ce, ve = W.upt_mixer(mixer, coop, vpt)
    tok = clip.tokenize(W.coop_prompt_strings(Pm, classes))
    close(W.text_forward(m, tok, ce), t_out.detach(), "upt text")
    close(W.vision_forward(m.visual, x, ve), v_out.detach(), "upt vision")
    out[f"{tag}.mixer_coop"] = ce.detach().numpy()
    out[f"{tag}.mixer_vpt"] = ve.detach().numpy()
    out[f"{tag}.text"] = t_out.detach().numpy()
    out[f"{tag}.vision"] = v_out.detach().numpy()
    out[f"{tag}.tokens"] = tok.numpy().astype(np.int32)
    logits, _ = W.cosine_head(v_out, t_out, m.logit_scale.detach())
    labels = torch.arange(n_img) % len(classes)
    loss = torch.nn.functional.cross_entropy(logits, labels)
    loss.backward()
    out[f"{tag}.loss"] = loss.detach().numpy()
    for name, p in ref.named_parameters():
        if p.grad is not None and not name.startswith(("image_encoder", "text_encoder")):
            out[f"{tag}.grad.{name}"] = p.grad.numpy()


# ----------------------------------------------------------------------------- G6
class _FakeImg:
    def __init__(self, idx):
        self.idx = idx

    def convert(self, mode):
        return self


class _FakeDataset:
    def __init__(self, paths):
        self.filepaths = list(paths)
        self.labels = None


def run_reference_leaderboard(P, paths, classnames, label_to_idx, k, tmp="/tmp/_golden_pl.pickle"):
    """Drive utils/clip_pseudolabels.compute_pseudo_labels with a fake model returning log P."""
    P = torch.tensor(np.asarray(P, dtype=np.float32))
    index = {p: i for i, p in enumerate(paths)}
    RP.Image.open = lambda path: _FakeImg(index[path])
    RP.tqdm = lambda it: it

    def transform(img):
        return torch.tensor(float(img.idx))

    def fake_model(img, text):
        i = int(img.item())
        lg = torch.log(P[i])[None]
        return lg, lg.t()

    ds = _FakeDataset(paths)
    RP.compute_pseudo_labels(k, "a photo of a {}", ds, classnames, transform, fake_model, label_to_idx, "cpu", tmp)
    os.remove(tmp)
    # the probabilities the scan actually compared: softmax(log P) in fp32
    probs = torch.log(P).softmax(dim=-1).numpy()
    return ds.filepaths, [int(x) for x in ds.labels], probs


def leaderboard_cases():
    cases = []

    def add(name, P, paths, classnames, ids, k):
        P = np.asarray(P, dtype=np.float32)
        l2i = dict(zip(classnames, ids))
        fp, lab, probs = run_reference_leaderboard(P, paths, classnames, l2i, k)
        pred = probs.argmax(axis=1) if k != LB.K_ALL else torch.from_numpy(probs).argmax(dim=1).numpy()
        pred = torch.argmax(torch.from_numpy(probs), dim=1).numpy()
        o_fp, o_lab = LB.leaderboard_scan(probs, pred, paths, ids, k)
        assert (o_fp, o_lab) == (fp, lab), f"{name}: python oracle != reference"
        if k != LB.K_ALL:
            c_fp, c_lab = cbind.leaderboard_ref(probs, pred, paths, ids, k)
            assert (c_fp, c_lab) == (fp, lab), f"{name}: C oracle != reference"
        cases.append({"name": name, "k": k, "classnames": classnames, "label_ids": ids, "paths": paths,
                      "P": P.tolist(), "probs_f32_hex": probs.astype(np.float32).tobytes().hex(),
                      "pred": [int(x) for x in pred], "filepaths": fp, "labels": lab})

    # KAT from SURVEY.md 8(a): last-appended quirk, spill to both b and c, eviction.
    add("survey_kat", [[.9, .05, .05], [.5, .3, .2], [.7, .2, .1], [.6, .3, .1], [.8, .1, .1], [.1, .8, .1]],
        [f"img{i}" for i in range(6)], ["a", "b", "c"], [10, 11, 12], 3)
    # exact ties -> path-string order (descending), incl. equal-score eviction candidates
    tie = [[.5, .25, .25]] * 3 + [[.6, .2, .2], [.5, .25, .25], [.5, .3, .2], [.25, .5, .25], [.25, .5, .25], [.2, .5, .3]]
    add("ties_paths", tie, ["p/b.jpg", "p/a.jpg", "p/c.jpg", "p/e.jpg", "p/d.jpg", "p/zz.jpg", "p/m.jpg", "p/n.jpg", "p/k.jpg"],
        ["x", "y", "z"], [0, 1, 2], 2)
    # k larger than anything a class receives; k = 1
    g = np.random.RandomState(7)
    Pr = g.dirichlet(np.ones(4) * 0.6, size=12)
    add("k_gt_n", Pr, [f"d/{i:03d}.png" for i in range(12)], ["c0", "c1", "c2", "c3"], [3, 1, 2, 0], 50)
    add("k_eq_1", Pr, [f"d/{i:03d}.png" for i in range(12)], ["c0", "c1", "c2", "c3"], [3, 1, 2, 0], 1)
    # the arg-max-only branch
    add("k_all", Pr, [f"d/{i:03d}.png" for i in range(12)], ["c0", "c1", "c2", "c3"], [3, 1, 2, 0], LB.K_ALL)
    # random medium cases with peaked and flat rows, many spills
    for s, (n, c, k, alpha) in enumerate([(200, 5, 4, 0.3), (300, 7, 16, 1.0), (400, 10, 3, 5.0), (150, 3, 8, 0.1)]):
        g = np.random.RandomState(100 + s)
        Pr = g.dirichlet(np.ones(c) * alpha, size=n).astype(np.float32) + 1e-6
        # quantise a third of the rows so exact score ties occur across images
        Pr[::3] = np.round(Pr[::3] * 8) / 8 + 1e-3
        Pr = Pr / Pr.sum(axis=1, keepdims=True)
        paths = [f"root/train/{g.randint(0, 10**6):06d}_{i}.jpg" for i in range(n)]
        add(f"random_{s}", Pr, paths, [f"class_{j}" for j in range(c)], list(g.permutation(c).astype(int).tolist()), k)
    return cases


# ----------------------------------------------------------------------------- G7
def _load_strategy_file(rel, pkg, base_name):
    """Import one reference strategy file with its (missing) base class stubbed, to reach its loss methods."""
    stub = types.ModuleType(pkg)
    setattr(stub, base_name, type(base_name, (), {}))
    stub.__path__ = []
    sys.modules.setdefault("methods", types.ModuleType("methods"))
    sys.modules[pkg] = stub
    spec = importlib.util.spec_from_file_location("_ref_" + rel.replace("/", "_"), os.path.join(REF, rel))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def fpl_losses(out):
    logits = T("g7.logits", (6, 5), 3.0)
    labels = torch.tensor([0, 3, 1, 4, 2, 3])
    paths = [f"f{i}.jpg" for i in range(6)]
    unl = [True, False, True, True, False, True]
    ce = torch.nn.CrossEntropyLoss()

    ssl = _load_strategy_file("methods/semi_supervised_learning/textual_fpl.py", "methods.semi_supervised_learning", "TextualPrompt")
    me = types.SimpleNamespace(loss_func=ce, check_unlabeled=[p for p, u in zip(paths, unl) if u], balance_param=W.balance_ssl(4, 2))
    me.cross_entropy = lambda *a, **k: ssl.TextualFPL.cross_entropy(me, *a, **k)
    ref = ssl.TextualFPL.define_loss_function(me, logits, labels, paths)
    close(W.fpl_loss_ssl(logits, labels, unl, me.balance_param), ref, "fpl ssl")
    out["g7.ssl"] = ref.numpy()

    tr = _load_strategy_file("methods/transductive_zsl/textual_fpl.py", "methods.transductive_zsl", "TextualPrompt")
    classes = ["a", "b", "c", "d", "e"]
    me = types.SimpleNamespace(loss_func=ce, label_to_idx={c: i for i, c in enumerate(classes)}, seen_classes=classes[:3],
                               unseen_classes=classes[3:], balance_param=W.balance_trzsl(3, 3))
    me.cross_entropy = lambda *a, **k: tr.TextualFPL.cross_entropy(me, *a, **k)
    ref = tr.TextualFPL.define_loss_function(me, logits, labels)
    close(W.fpl_loss_trzsl(logits, labels, [0, 1, 2], [3, 4], me.balance_param), ref, "fpl trzsl")
    out["g7.trzsl"] = ref.numpy()

    ul = _load_strategy_file("methods/unsupervised_learning/visual_fpl.py", "methods.unsupervised_learning", "VisualPrompt")
    me = types.SimpleNamespace(loss_func=ce, classes=classes)
    me.cross_entropy = lambda *a, **k: ul.VisualFPL.cross_entropy(me, *a, **k)
    ref = ul.VisualFPL.define_loss_function(me, logits, labels)
    close(W.fpl_loss_ul(logits, labels), ref, "fpl ul")
    out["g7.ul"] = ref.numpy()
    out["g7.logits"] = logits.numpy()


def main():
    """python oracle/gen_golden.py [small] [vitb16] [vitb32] [vitl14] [leaderboard]   (no argument = all groups)"""
    os.makedirs(OUT, exist_ok=True)
    cbind.build()
    groups = set(sys.argv[1:]) or {"small", "vitb16", "vitb32", "vitl14", "leaderboard"}
    classes = ["forest", "annual crop land", "river", "sea lake", "highway"]

    if "small" in groups:
        out = {}
        towers("tiny", 3, classes, 3, "g1", out)
        towers("small", 2, classes[:4], 16, "g1s", out)
        upt("tiny", 3, classes[:3], 4, "g4", out)
        fpl_losses(out)
        np.savez_compressed(os.path.join(OUT, "golden_small.npz"), **out)

    if "vitb16" in groups:
        # G3: full-size ViT-B/16 + text-B spot check, incl. the prompt gradients of the step bench.py times (CoOp: [1,16,512]
        # through the 12-layer text tower; VPT: [16,768] through the 12-layer ViT) and UPT end to end at these dimensions
        big = {}
        towers("ViT-B/16", 2, classes[:3], 16, "g3", big, with_grad=True)
        upt("ViT-B/16", 2, classes[:3], 4, "g4b", big)
        np.savez_compressed(os.path.join(OUT, "golden_vitb16.npz"), **big)

    if "vitb32" in groups:
        # ViT-B/32: the encoder every shipped script of the reference defaults to (scripts/run_pseudolabels_ssl.sh:4): patch 32
        # (im2col K = 3 072), S = 50 / 66; 2 images, 3 prompts, 16 prompt tokens, forward of both towers + both prompt gradients
        b32 = {}
        towers("ViT-B/32", 2, classes[:3], 16, "g6", b32, with_grad=True)
        np.savez_compressed(os.path.join(OUT, "golden_vitb32.npz"), **b32)

    if "vitl14" in groups:
        # BASELINE.json configs[4]: ViT-L/14@336px (d = 1024, 24 layers, 16 heads, S = 577 / 593, E = 768) + the 12-head 768-wide
        # text tower, 2 images and 3 prompts: forward of both towers without and with 16 prompt tokens, and both prompt gradients
        huge = {}
        towers("ViT-L/14@336px", 2, classes[:3], 16, "g5", huge, with_grad=True)
        np.savez_compressed(os.path.join(OUT, "golden_vitl14_336.npz"), **huge)

    if "leaderboard" in groups:
        with open(os.path.join(OUT, "leaderboard.json"), "w") as f:
            json.dump({"seed": SEED, "cases": leaderboard_cases()}, f)
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
