"""The index guarantee on NON-degenerate statistics: the seeded random-init towers give near-uniform softmaxes and well-behaved
activations, so screen-and-refine had only ever met peaked rows, outlier channels and f16 overflows as fabricated probabilities
(tests/test_refine_scan.py).  Here they come out of the REAL towers: `clip.load(..., synthetic="stress")` (weights.stress_state_dict) carries
|x| ~ 200 in four channels of the vision residual stream and overflows the f16 stream on roughly a fifth of the images; the class "text features"
are prototypes of the pool's own embeddings, which gives peaked rows (logit spread > 10) with contested arg-maxes.  Asserted: default (identical)
mode == exact mode, list for list; the non-finite screen rows are counted and refined; bound and audit are reported.

north_star: "identical top-k pseudolabel indices" (reference utils/clip_pseudolabels.py:38-41, 73-101 decides on fp32 values)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


_PROBLEM = {}


def stress_problem(n, n_classes, device, seed=77):
    """(f16 model, f32 twin, pool [n,3,224,224] on the device, paths, prototype text features [C,512], exact embeddings); built once per process
    (the three k cases share it: the model, the pool and its exact encode were 2/3 of each case's 21 s)."""
    key = (n, n_classes, str(device), seed)
    if key not in _PROBLEM:
        _PROBLEM.clear()
        _PROBLEM[key] = _stress_problem(n, n_classes, device, seed)
    return _PROBLEM[key]


def _stress_problem(n, n_classes, device, seed):
    import grip_amd  # noqa: F401
    from grip_amd import clip, pseudolabels as pl
    from grip_amd.data.synthetic import pool_paths
    m, _ = clip.load("ViT-B/16", device=device, synthetic="stress")
    twin = m.exact_twin()
    from conftest import structured_pool
    pool = structured_pool(seed, n, 224, device)
    with torch.no_grad():
        e32 = torch.empty(n, 512, device=device)
        twin.visual.tower.encode_chunks(pool, e32, 0, n, 440, streams=1)
    assert torch.isfinite(e32).all(), "the f32 tower must stay finite on the stress model"
    g = torch.Generator().manual_seed(seed)
    anchors = torch.randperm(n, generator=g)[:n_classes].to(device)
    en = e32 / e32.norm(dim=-1, keepdim=True)
    # class "text features": prototypes of the pool's own embeddings with the pool mean removed (what separates the classes, not what the images share):
    # 100 x cosine against them spreads over tens of logits -- peaked rows, and every prototype wins somewhere
    txt = en[anchors] - en.mean(0, keepdim=True) + 0.003 * torch.randn(n_classes, 512, generator=g).to(device)
    return m, twin, pool, pool_paths(n), txt.contiguous(), e32


@pytest.mark.parametrize("k", [16, 3, 10000000])
def test_identical_equals_exact_on_the_stress_model(k):
    import grip_amd  # noqa: F401
    from grip_amd import engine, pseudolabels as pl
    dev = torch.device("cuda", 0)
    n, C = 6144, 40
    m, twin, pool, paths, txt, e32 = stress_problem(n, C, dev)
    labels = list(range(100, 100 + C))
    _, p32, _, a32 = engine.cosine_head(e32, txt, 100.0)
    p32h, a32h = p32.cpu().numpy(), a32.cpu().numpy()
    lg = np.log(np.maximum(p32h, 1e-45))
    spread = float(np.mean(lg.max(1) - np.median(lg, 1)))
    assert spread >= 10.0, spread                                     # peaked rows: the median class is e^-10 below the winner
    assert len(np.unique(a32h)) >= C // 2                             # ... and the arg-max is contested, not one dominant class
    want = pl.leaderboard(p32h, a32h, paths, labels, k)
    with torch.no_grad():
        e16 = pl.encode_pool(m.visual.tower, pool, chunk=440, screen=True)       # the product's screen (compensated stream by default)
    bad16 = int((~torch.isfinite(e16).all(dim=1)).sum())
    assert 0.03 * n < bad16 < 0.6 * n, bad16                          # the f16 stream really overflows on a share of the images
    for tier in ("mid", "two"):
        mid = pl.mid_tower(m, n) if tier == "mid" else None
        got = pl.identical_lists(m.visual.tower, twin.visual.tower, pool, txt, 100.0, paths, labels, k, emb16=e16, visual_mid=mid)
        st = pl.LAST_REFINE_STATS
        print(f"stress k={k} tiers={st['tiers']}: logit spread {spread:.1f}, {bad16} non-finite f16 rows, {st['rows_refined']} of {n} rows re-encoded "
              f"({st['rows_mid']} split / {st['rows_exact']} f32), bound {st['eps']:.2e} (largest deviation {st['max_deviation']:.2e}), "
              f"audit {st['audit_rows']} rows max {st['audit_max_deviation']:.2e} widened={st['audit_widened']}")
        assert (list(got[0]), list(got[1])) == (list(want[0]), list(want[1])), f"k={k} {tier}: identical-mode lists differ from the exact mode's on the stress model"
        assert st["nonfinite_screen_rows"] == bad16 and np.isfinite(st["eps"]) and st["bound_form"] == "odds"
        # A ceiling on what the screen costs here (VERDICT r5 #1 / ADVICE r5): against mean-removed prototypes the f16 embeddings' direction error is a
        # logit error of tenths.  Under the RELATIVE bound of rounds 3-5 that re-encoded every row (the test could only assert rows_refined <= n); the
        # log-odds form keeps it to the non-finite rows plus the rows near a threshold.  The pool is small (k C board slots = up to 10 % of it are members
        # that must be ordered exactly), so the ceiling is generous; bench.py `secondary.identical_on_stress_model` reports N = 50 000.
        # (simulated from dumped embeddings of this model, tools/delta_probe.py: 0.29 / 0.19 / 0.19 of the finite rows for k = 16 / 3 / label-everything with the
        # compensated stream, 0.48 / 0.32 / 0.36 with the plain one)
        assert st["rows_refined"] <= bad16 + {16: 0.6, 3: 0.5, 10000000: 0.5}[k] * (n - bad16), (st["rows_refined"], bad16, n)
        if tier == "mid":
            assert st["rows_exact"] <= 0.25 * st["rows_refined"] + st["calibration_rows"]


@pytest.mark.parametrize("variant", ["stress", "realistic"])
def test_peaked_pool_lists_equal_the_reference_functions(variant):
    """tests/golden/{stress,realistic}_vitb16_lists.npz (oracle/gen_golden_stress.py lists [realistic]): what the REFERENCE's compute_pseudo_labels
    returned for 2 048 structured images x 40 prototype classes on the CPU fp32 oracle, with the probabilities it compared -- "stress": the STRESS
    weights against mean-removed prototypes (mean top-1 0.75, decision margins 6e-7 .. 2e-3, logit errors of tenths in the f16 screen); "realistic": the
    standard weights against the un-centred blends unit(m + 2 (e_c - m)) of bench.py's realistic pool (peaked rows at ordinary logit errors: the regime
    the log-odds bound was made for).  The GPU's exact mode must reproduce the probabilities and the lists (k = 3, 16, label-everything) up to
    transpositions of scores closer than one fp32 logit ulp, and the default screen-and-refine path -- two and three tiers -- must return exactly
    the exact mode's lists.  (VERDICT r5 #6a: until r06 the stress model was pinned by 16 embeddings only and identical == exact was a property
    test between two HIP paths.)"""
    import json
    import os

    import grip_amd  # noqa: F401
    from grip_amd import clip, engine, pseudolabels as pl
    from grip_amd.data.synthetic import pool_paths
    from test_gpu_exact import assert_lists_identical
    fx = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", f"{variant}_vitb16_lists.npz"))
    o_probs = fx["probs"]
    n, C = o_probs.shape
    dev = torch.device("cuda", 0)
    m, _ = clip.load("ViT-B/16", device=dev, synthetic="stress" if variant == "stress" else "standard")
    twin = m.exact_twin()
    seed = int(fx["seed"])
    from conftest import structured_pool
    pool = structured_pool(seed, n, 224, dev)
    txt = torch.from_numpy(fx["txt"]).to(dev)
    scale = float(fx["logit_scale"])
    paths, labels = pool_paths(n), [100 + i for i in range(C)]
    with torch.no_grad():
        e32 = torch.empty(n, 512, device=dev)
        twin.visual.tower.encode_chunks(pool, e32, 0, n, 256, streams=1)
        e16 = pl.encode_pool(m.visual.tower, pool, chunk=512, screen=True)
    _, p32, _, a32 = engine.cosine_head(e32, txt, scale)
    p32h, a32h = p32.cpu().numpy(), a32.cpu().numpy()
    dev_odds = pl._deviation_odds(p32h, o_probs, 1e-30)
    # (fp32 GPU towers vs the fp32 CPU oracle: ~1e-5 in the logits, which the mean-removed prototypes amplify ~30x: measured 2.9e-4)
    assert dev_odds <= 1e-3, f"exact-mode probabilities are {dev_odds:.2e} (log-odds) from the reference's on the stress model"
    os.environ["GRIP_SPLIT_TIER"] = "1"
    try:
        mid = pl.mid_tower(m, n)
    finally:
        del os.environ["GRIP_SPLIT_TIER"]
    assert mid is not None
    rows = []
    for k in (3, 16, 10000000):
        ref = json.loads(str(fx[f"lists_k{k}"]))
        exact = pl.leaderboard(p32h, a32h, paths, labels, k)
        swapped = assert_lists_identical(exact, (ref[0], ref[1]), o_probs, paths, labels, f"{variant} exact k={k}")
        assert swapped <= 2, swapped
        for tiers, vm in ((2, None), (3, mid)):
            got = pl.identical_lists(m.visual.tower, twin.visual.tower, pool, txt, scale, paths, labels, k, emb16=e16, visual_mid=vm)
            st = pl.LAST_REFINE_STATS
            assert (list(got[0]), list(got[1])) == (list(exact[0]), list(exact[1])), f"{variant} k={k} tiers={tiers}: screen-and-refine differs from the exact mode"
            rows.append({"k": k, "tiers": tiers, "pairs": len(ref[0]), "reference_margin": float(fx[f"margin_k{k}"]), "tie_transpositions_vs_reference": swapped,
                         "rows_reencoded": st["rows_refined"], "nonfinite_screen_rows": st["nonfinite_screen_rows"], "bound_form": st["bound_form"], "bound": st["eps"]})
            print(rows[-1])
    from conftest import write_report
    write_report(f"{variant}_reference_lists.json", {"images": n, "classes": C, "exact_vs_reference_log_odds_deviation": dev_odds, "cases": rows})
