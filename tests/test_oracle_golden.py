"""CPU: the oracle restatements reproduce the committed golden vectors (which were produced by the
reference's own wrappers over the oracle `clip`, oracle/gen_golden.py).  Guards the oracle against
drift; runs without a GPU and without /root/reference."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import REPO, oracle_clip

SEED = 100


def _inputs(name, shape, std=1.0):
    import grip_amd  # noqa: F401
    from grip_amd import rng
    return torch.from_numpy(rng.normal(SEED, rng.stream_id(name), shape, 0.0, std))


@pytest.fixture(scope="module")
def tiny():
    return oracle_clip().load("tiny")[0]


def test_vision_and_text_wrappers_match_golden(tiny, golden_small):
    from oracle import wrappers as W
    g = golden_small
    x = _inputs("g1.x", (3, 3, 32, 32))
    vp = _inputs("g1.vprefix", (3, 128), 0.02)
    tp = _inputs("g1.tprefix", (1, 3, 128), 0.02)
    with torch.no_grad():
        np.testing.assert_allclose(W.vision_forward(tiny.visual, x, None).numpy(), g["g1.vision_p0"], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(W.vision_forward(tiny.visual, x, vp).numpy(), g["g1.vision_p3"], rtol=1e-5, atol=1e-5)
        tok = torch.from_numpy(g["g1.coop_tokens"])
        np.testing.assert_allclose(W.text_forward(tiny, tok, tp).numpy(), g["g1.text_p3"], rtol=1e-5, atol=1e-5)
        ztok = torch.from_numpy(g["g1.zs_tokens"])
        np.testing.assert_allclose(W.text_forward(tiny, ztok, None).numpy(), g["g1.text_p0"], rtol=1e-5, atol=1e-5)
    vp = vp.requires_grad_(True)
    (W.vision_forward(tiny.visual, x, vp) ** 2).sum().backward()
    np.testing.assert_allclose(vp.grad.numpy(), g["g1.vision_p3_grad_prefix"], rtol=1e-4, atol=1e-4)


def test_wrappers_without_positional_embedding_match_golden():
    """G10 (oracle/gen_golden_posemb.py): the enable_pos_emb=False / pos_emb=False branches, models/clip_encoders.py:70-74 and :141."""
    from oracle import wrappers as W
    g = np.load(os.path.join(REPO, "tests", "golden", "posemb.npz"))
    m = oracle_clip().load("small")[0]
    x = _inputs("px.s.x", (2, 3, 64, 64))
    vp = _inputs("px.s.vprefix", (16, m.visual.conv1.weight.shape[0]), 0.02).requires_grad_(True)
    tp = _inputs("px.s.tprefix", (1, 16, m.token_embedding.weight.shape[1]), 0.02).requires_grad_(True)
    v = W.vision_forward(m.visual, x, vp, pos_emb=False)
    t = W.text_forward(m, torch.from_numpy(g["px.s.tokens"]), tp, enable_pos_emb=False)
    np.testing.assert_allclose(v.detach().numpy(), g["px.s.vision"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(t.detach().numpy(), g["px.s.text"], rtol=1e-5, atol=1e-5)
    (v ** 2).sum().backward()
    (t ** 2).sum().backward()
    np.testing.assert_allclose(vp.grad.numpy(), g["px.s.vision_grad_prefix"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(tp.grad.numpy(), g["px.s.text_grad_prefix"], rtol=1e-4, atol=1e-4)


def test_tokens_in_golden_come_from_the_stand_in_tokenizer(golden_small):
    from oracle import wrappers as W
    classes = ["forest", "annual crop land", "river", "sea lake", "highway"]
    oc = oracle_clip()
    assert (oc.tokenize(W.coop_prompt_strings(3, classes)).numpy() == golden_small["g1.coop_tokens"]).all()
    z = W.zero_shot_prompt_strings("a photo of a {}", classes)
    assert z[0] == "a photo of a {}forest"          # the literal "{}" quirk of utils/clip_pseudolabels.py:24
    assert (oc.tokenize(z).numpy() == golden_small["g1.zs_tokens"]).all()
    assert W.format_prompt_strings("a photo of a {}", ["annual_crop"]) == ["a photo of a annual crop"]


def test_upt_mixer_matches_golden(golden_small):
    import grip_amd  # noqa: F401
    from grip_amd import weights
    from oracle import wrappers as W
    mixer = {k: torch.from_numpy(v) for k, v in weights.init_upt_mixer(128, 128, 128, SEED).items()}
    coop = _inputs("g4.coop", (1, 4, 128), 0.02)
    vpt = _inputs("g4.vpt", (1, 4, 128), 0.02)
    with torch.no_grad():
        ce, ve = W.upt_mixer(mixer, coop, vpt)
    np.testing.assert_allclose(ce.numpy(), golden_small["g4.mixer_coop"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(ve.numpy(), golden_small["g4.mixer_vpt"], rtol=1e-5, atol=1e-5)


def test_fpl_losses_match_golden(golden_small):
    from oracle import wrappers as W
    g = golden_small
    logits = torch.from_numpy(g["g7.logits"])
    labels = torch.tensor([0, 3, 1, 4, 2, 3])
    unl = [True, False, True, True, False, True]
    np.testing.assert_allclose(W.fpl_loss_ssl(logits, labels, unl, W.balance_ssl(4, 2)).numpy(), g["g7.ssl"], rtol=1e-6)
    np.testing.assert_allclose(W.fpl_loss_trzsl(logits, labels, [0, 1, 2], [3, 4], W.balance_trzsl(3, 3)).numpy(), g["g7.trzsl"], rtol=1e-6)
    np.testing.assert_allclose(W.fpl_loss_ul(logits, labels).numpy(), g["g7.ul"], rtol=1e-6)


def _cases():
    with open(os.path.join(REPO, "tests", "golden", "leaderboard.json")) as f:
        return json.load(f)["cases"]


@pytest.mark.parametrize("case", _cases(), ids=lambda c: c["name"])
def test_leaderboard_oracles_match_reference_outputs(case):
    from oracle import cbind, leaderboard as LB
    n = len(case["paths"])
    probs = np.frombuffer(bytes.fromhex(case["probs_f32_hex"]), dtype=np.float32).reshape(n, -1)
    fp, lab = LB.leaderboard_scan(probs, case["pred"], case["paths"], case["label_ids"], case["k"])
    assert fp == case["filepaths"] and lab == case["labels"]
    if case["k"] != LB.K_ALL:
        fp, lab = cbind.leaderboard_ref(probs, case["pred"], case["paths"], case["label_ids"], case["k"])
        assert fp == case["filepaths"] and lab == case["labels"]


def test_survey_known_answer():
    """SURVEY.md 8(a) known-answer vector (produced there by running the reference function)."""
    case = [c for c in _cases() if c["name"] == "survey_kat"][0]
    assert list(zip(case["filepaths"], case["labels"])) == [("img0", 10), ("img4", 10), ("img2", 10), ("img3", 11), ("img5", 11), ("img3", 12)]
