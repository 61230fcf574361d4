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


def stress_problem(n, n_classes, device, seed=77):
    """(f16 model, f32 twin, pool [n,3,224,224] on the device, paths, prototype text features [C,512], exact embeddings)."""
    import grip_amd  # noqa: F401
    from grip_amd import clip, pseudolabels as pl
    from grip_amd.data.synthetic import pool_paths, structured_images
    m, _ = clip.load("ViT-B/16", device=device, synthetic="stress")
    twin = m.exact_twin()
    pool = torch.empty(n, 3, 224, 224, device=device)
    for lo in range(0, n, 1024):
        hi = min(lo + 1024, n)
        pool[lo:hi] = structured_images(seed, lo, hi, 224).to(device)
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
        e16 = pl.encode_pool(m.visual.tower, pool, chunk=440)
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
        assert st["nonfinite_screen_rows"] == bad16 and np.isfinite(st["eps"])
        # (no claim on how many rows the screen saves here: against mean-removed prototypes the f16 embeddings' ~1e-3 direction error becomes a
        # logit error of ~0.3, the measured bound is ~0.7 and nearly every row of so small a pool sits within it of a threshold -- the pass degrades
        # to the exact mode, as it must; bench.py `secondary.identical_on_stress_model` reports the same at N = 50 000)
        assert st["rows_refined"] <= n
