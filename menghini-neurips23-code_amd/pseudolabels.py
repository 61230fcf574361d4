"""Pseudolabel engine: batched pool encode (sharded over ranks) -> cached text features ->
fused head -> exact sequential leaderboard.  This is the compute behind
utils.clip_pseudolabels.{compute_pseudo_labels, pseudolabel_top_k} and the nine
assign_pseudo_labels of the reference; those keep their signatures in grip_amd.utils / grip_amd.methods.

Two algorithmic changes against the reference loop (utils/clip_pseudolabels.py:31-61), neither of
which changes a result: the class prompts are encoded once instead of once per image, and images
go through the tower in chunks instead of one by one.
"""
import numpy as np
import torch

from . import dist as gdist
from . import engine

K_ALL = 10000000   # utils/clip_pseudolabels.py:27


def path_ranks(paths):
    """Dense rank of every path among all paths under Python's string order (the leaderboard breaks
    score ties by comparing the path strings, utils/clip_pseudolabels.py:79-82)."""
    order = {p: i for i, p in enumerate(sorted(set(paths)))}
    return np.fromiter((order[p] for p in paths), dtype=np.int64, count=len(paths))


@torch.no_grad()
def encode_pool(visual_tower, images, chunk=880, prefix=None, out=None):
    """Encode an ordered pool.  `images` is a tensor [N,3,R,R] (any device) or a callable
    (lo, hi) -> tensor for that slice.  With torch.distributed initialised the pool is sharded
    contiguously and the embeddings are all-gathered; returns [N, E] f32 on the device."""
    n = images.shape[0] if torch.is_tensor(images) else images.n
    lo, hi, per = gdist.shard_range(n)
    dev = visual_tower.device
    local = torch.empty(max(hi - lo, 0), visual_tower.embed_dim, dtype=torch.float32, device=dev)
    visual_tower.encode_chunks(images, local, lo, hi, chunk, prefix)
    return gdist.allgather_rows(local, n, per)


def leaderboard(probs, pred, paths, class_labels, k):
    """(filepaths, labels) exactly as utils/clip_pseudolabels.py:103-112 rebuilds them."""
    if k == K_ALL:
        return list(paths), [class_labels[int(j)] for j in pred]
    img, cls = engine.leaderboard_scan(probs, pred, path_ranks(paths), k)
    return [paths[i] for i in img], [class_labels[int(c)] for c in cls]


# ------------------------------------------------------------------------------------------ screen and refine
REFINE_CALIB_ROWS = 256     # rows re-encoded exactly up front to measure the f16 towers' deviation on THIS pool
REFINE_SAFETY = 2.0         # bound = safety x the largest deviation seen on any row re-encoded so far (it only ever grows)


REFINE_ESCALATE_AFTER = 8   # rounds after which whatever is still un-refined is re-encoded in one go (pathological pools only, see refine_scan)


def refine_scan(probs, pred, ranks, k, exact_rows, calib=REFINE_CALIB_ROWS, safety=REFINE_SAFETY, max_rounds=64):
    """Leaderboard lists of the reference's fp32 scan from probabilities of the f16 towers (utils/clip_pseudolabels.py:38-112).

    `probs` [N, C] f32 / `pred` [N] come from the f16 image tower (modified in place); `exact_rows(idx)` returns the exact
    (f32-tower) probabilities and arg-max of the rows `idx` (ascending int64 array).  The f16 rows are trusted only up to a relative
    bound eps = safety x (largest |p16 / p32 - 1| over every row re-encoded so far, starting with `calib` rows -- at most 1/16 of
    the pool, at least 16 -- spread evenly over it); grip_leaderboard_scan_bounded marks every un-refined row that takes part in a comparison the bound
    cannot decide; those rows are re-encoded exactly and the scan repeats until nothing is marked and the bound has not moved.
    The final scan takes, decision by decision, the decisions of the scan over the all-f32 probabilities, so the lists are the
    exact mode's lists (asserted at N = 50 000 in tests/test_gpu_identical.py) at a fraction of its cost.
    Returns (img, cls, stats)."""
    n, c = probs.shape
    refined = np.zeros(n, dtype=bool)
    dev_max = 0.0
    floor = np.float32(1e-30)       # below this a probability has no relative accuracy left to speak of (denormal range)

    def refine(idx):
        nonlocal dev_max
        idx = np.asarray(idx, dtype=np.int64)
        if idx.size == 0:
            return
        p32, a32 = exact_rows(idx)
        p16 = probs[idx]
        ok = (p16 > floor) & (p32 > floor)
        if ok.any():
            dev_max = max(dev_max, float(np.max(np.abs(p16[ok].astype(np.float64) - p32[ok]) / p16[ok])))
        probs[idx] = p32
        pred[idx] = a32
        refined[idx] = True

    stats = {"rows": n, "calibration_rows": 0, "rounds": 0, "scans": 0}
    if n == 0:
        return np.empty(0, np.int32), np.empty(0, np.int32), dict(stats, rows_refined=0, eps=0.0, max_deviation=0.0)
    # calibration rows: `calib`, but at most 1/16 of the pool -- and never fewer than 16 (a bound from one or two rows is no bound)
    refine(np.unique(np.linspace(0, n - 1, min(n, max(16, min(calib, n // 16)))).astype(np.int64)))
    stats["calibration_rows"] = int(refined.sum())
    eps = safety * dev_max
    per_round = []
    while True:
        rel = np.where(refined, np.float32(0), np.float32(eps)).astype(np.float32)
        img, cls, amb = engine.leaderboard_scan_bounded(probs, pred, ranks, rel, k)
        stats["scans"] += 1
        todo = np.flatnonzero(amb & ~refined)
        if todo.size == 0 and safety * dev_max <= eps:
            break
        if stats["rounds"] >= max_rounds:
            raise RuntimeError(f"refine_scan: no fixed point after {max_rounds} rounds ({int(refined.sum())} of {n} rows refined)")
        if stats["rounds"] >= REFINE_ESCALATE_AFTER:
            # Heavily tied scores can keep a board in the reference's unsorted regime, where every comparison has to be certain and a
            # round only advances a few images in dataset order: stop trickling and take the exact tower to everything that is left.
            todo = np.flatnonzero(~refined)
            stats["escalated"] = True
        refine(todo)
        per_round.append(int(todo.size))
        stats["rounds"] += 1
        eps = max(eps, safety * dev_max)
    stats.update(rows_refined=int(refined.sum()), refined_per_round=per_round, eps=float(eps), max_deviation=float(dev_max), safety=float(safety))
    return img, cls, stats


LAST_REFINE_STATS = None     # what the most recent identical_lists call did (rows re-encoded, rounds, bound): logged / reported by bench.py


def mode():
    """How pseudolabel passes compute their lists: "identical" (default: f16 screen + exact refinement, the lists of the fp32
    scan), "f16" (the f16 towers' lists as they are: boundary items may differ from the fp32 scan's), or "exact" is what a
    model loaded with clip.load(..., exact=True) does by itself.  $GRIP_PSEUDOLABEL_MODE."""
    import os
    m = os.environ.get("GRIP_PSEUDOLABEL_MODE", "identical")
    if m not in ("identical", "f16"):
        raise ValueError(f"GRIP_PSEUDOLABEL_MODE={m!r}: expected 'identical' or 'f16'")
    return m


def take_images(images, idx):
    """Rows `idx` (ascending int64 array) of an image pool: a tensor [N,3,R,R] or a lazy pool with .take(idx)."""
    if torch.is_tensor(images):
        return images[torch.as_tensor(idx, device=images.device)]
    return images.take(idx)


@torch.no_grad()
def identical_lists(visual16, visual32, images, txt_exact, scale, paths, class_labels, k, chunk=880, exact_chunk=880, prefix=None,
                    argmax_on="probs", streams=2, emb16=None):
    """(filepaths, labels) of the reference's fp32 pseudolabel scan (utils/clip_pseudolabels.py:24-112) at close to the f16
    towers' throughput: the whole pool goes through the f16 vision tower `visual16` (sharded over ranks, one all-gather), the
    head scores it against the EXACT text features `txt_exact`, and refine_scan re-encodes with the f32 tower `visual32` only the
    rows whose f16 probabilities cannot decide a comparison the lists depend on (each rank re-encodes the marked rows of its own
    shard; one small all-gather per round).  Bit-for-bit the lists of the exact mode (tests/test_gpu_identical.py)."""
    global LAST_REFINE_STATS
    n = len(paths)
    if n == 0:
        LAST_REFINE_STATS = {"rows": 0, "rows_refined": 0, "rounds": 0, "scans": 0, "calibration_rows": 0, "refined_per_round": [], "eps": 0.0, "max_deviation": 0.0,
                             "safety": REFINE_SAFETY, "rows_refined_this_rank": 0}
        return [], []
    emb = emb16 if emb16 is not None else encode_pool(visual16, images, chunk=chunk, prefix=prefix)
    dev = emb.device
    _, probs, am_l, am_p = engine.cosine_head(emb, txt_exact, scale)
    probs_h = probs.cpu().numpy()
    pred_h = (am_p if argmax_on == "probs" else am_l).cpu().numpy()
    lo, hi, _ = gdist.shard_range(n)
    encoded = [0]

    def exact_rows(idx):
        mine = idx[(idx >= lo) & (idx < hi)]
        local = torch.empty(len(mine), visual32.embed_dim, dtype=torch.float32, device=dev)
        if len(mine):
            visual32.encode_chunks(lambda a, b: take_images(images, mine[a:b]), local, 0, len(mine), exact_chunk, prefix, streams=1)
        encoded[0] += len(mine)
        rows = gdist.allgather_selected(local, idx, n)
        _, p, al, ap = engine.cosine_head(rows, txt_exact, scale)
        return p.cpu().numpy(), (ap if argmax_on == "probs" else al).cpu().numpy()

    img, cls, stats = refine_scan(probs_h, pred_h, path_ranks(paths), k, exact_rows)
    stats["rows_refined_this_rank"] = encoded[0]
    LAST_REFINE_STATS = stats
    return [paths[i] for i in img], [class_labels[int(c)] for c in cls]


@torch.no_grad()
def pseudolabel_from_features(img_emb, txt_emb, scale, paths, class_labels, k, argmax_on="probs"):
    """Head + scan.  argmax_on: "probs" (compute_pseudo_labels, :39) or "logits" (assign_pseudo_labels)."""
    logits, probs, am_l, am_p = engine.cosine_head(img_emb, txt_emb, scale)
    pred = (am_p if argmax_on == "probs" else am_l).cpu().numpy()
    return leaderboard(probs.cpu().numpy(), pred, paths, class_labels, k)
