"""Pseudolabel engine: batched pool encode (sharded over ranks) -> cached text features ->
fused head -> exact sequential leaderboard.  This is the compute behind
utils.clip_pseudolabels.{compute_pseudo_labels, pseudolabel_top_k} and the nine
assign_pseudo_labels of the reference; those keep their signatures in grip_amd.utils / grip_amd.methods.

Two algorithmic changes against the reference loop (utils/clip_pseudolabels.py:31-61), neither of
which changes a result: the class prompts are encoded once instead of once per image, and images
go through the tower in chunks instead of one by one.
"""
import os

import numpy as np
import torch

from . import dist as gdist
from . import engine

K_ALL = 10000000   # utils/clip_pseudolabels.py:27


def _path_ranks_literal(paths):
    order = {p: i for i, p in enumerate(sorted(set(paths)))}
    return np.fromiter((order[p] for p in paths), dtype=np.int64, count=len(paths))


def _path_ranks_bytes(paths):
    """The same ranks without a Python-level sort: ASCII paths as a fixed-width byte matrix (code-point order == byte order, a shorter
    string pads with NUL, which sorts first -- as a proper prefix does in Python), common prefix dropped, 8 bytes per big-endian u64
    column; one SIMD argsort when the first column already separates the strings, a column-wise stable sort otherwise."""
    n = len(paths)
    a = np.array(paths, dtype=np.bytes_)        # UnicodeEncodeError for non-ASCII strings: the caller falls back
    w = a.dtype.itemsize
    if w == 0:
        return np.zeros(n, dtype=np.int64)
    b = a.view(np.uint8).reshape(n, w)
    if int(b.min()) == 0 and any("\0" in p for p in paths):
        raise UnicodeEncodeError("ascii", "", 0, 1, "NUL inside a path")      # (padding and content would be confused)
    varies = b.min(axis=0) != b.max(axis=0)
    if not varies.any():
        return np.zeros(n, dtype=np.int64)
    b = b[:, int(np.argmax(varies)):]
    w = b.shape[1]
    c = np.zeros((n, w + (-w) % 8), dtype=np.uint8)
    c[:, :w] = b
    cols = c.view(">u8")
    first = np.ascontiguousarray(cols[:, 0]).astype(np.uint64)
    order = np.argsort(first)
    s0 = first[order]
    if cols.shape[1] > 1 and (s0[1:] == s0[:-1]).any():
        order = np.lexsort(cols.T[::-1])
    srt = cols[order]
    new = np.empty(n, dtype=np.int64)
    new[0] = 0
    np.cumsum((srt[1:] != srt[:-1]).any(axis=1), out=new[1:])
    out = np.empty(n, dtype=np.int64)
    out[order] = new
    return out


_RANK_CACHE = []        # [(tuple(paths), ranks)], most recent first: GRIP re-labels the same pool every iteration


def path_ranks(paths):
    """Dense rank of every path among all paths under Python's string order (the leaderboard breaks
    score ties by comparing the path strings, utils/clip_pseudolabels.py:79-82).  Vectorised (0.1 s at 400 000 paths against
    0.6 s for the literal form, tools/scan_scale.py) and remembered for the last few pools."""
    key = tuple(paths)
    for i, (k, r) in enumerate(_RANK_CACHE):
        if len(k) == len(key) and k == key:          # element-wise; identical string objects compare by pointer
            if i:
                _RANK_CACHE.insert(0, _RANK_CACHE.pop(i))
            return r
    try:
        r = _path_ranks_bytes(paths)
    except (UnicodeEncodeError, TypeError, ValueError):
        r = _path_ranks_literal(paths)
    r.setflags(write=False)
    _RANK_CACHE.insert(0, (key, r))
    del _RANK_CACHE[4:]
    return r


# When the compensated screen pays: it costs +6.6 % encode time -- as much as re-encoding 2.2 % of the pool with the split-f16 tier (0.1 ms against 34 us per image) -- and
# saves the rows its 2.2x tighter bound no longer marks: measured r06 at N = 50 000, 30 % of the plain screen's marked rows on the near-uniform timed pool (2 836 -> 1 999:
# plain wins by 1 %), 44 % on the structured pool (4 472 -> 2 498: compensated wins by 3 %), and the difference between 9.0k and 11.3k img/s on the stress model.
SCREEN_HILO_KEEP = 0.040    # a COMPENSATED pass that still marked more than this share of the pool (calibration, audit and non-finite rows aside) keeps the next pass compensated
SCREEN_HILO_TAKE = 0.064    # a PLAIN pass that marked more than this share sends the next pass to the compensated stream (the same break-even seen from the other side)
_SCREEN_CHOICE = {}         # pool key -> stream the NEXT pass over it screens with (what its last pass measured)


def screen_stream(key=None):
    """Residual stream of the f16 tower when it SCREENS a pool for identical_lists.  "hilo": a compensated pair of f16 numbers per element
    (GRIP_FWD_STREAM_HILO: the 24 roundings of the stream of a ViT-B/16 image no longer accumulate; the embeddings' direction error against the f32
    tower and the measured bound of the screen drop 2.2 - 2.5x, for +6.6 % encode time).  "f16": the plain stream (rounds 1-5).  $GRIP_SCREEN_STREAM =
    "auto" (default) picks per pool: the first pass over a pool screens compensated (the safe side: it never costs more than 6.6 %), every pass records the
    share of the pool its scan marked (note_screen_bound), and the next pass over the same pool -- GRIP re-labels one pool every iteration,
    pseudo_iterative.py:62-125 -- screens plain where that share says the tighter bound does not pay (SCREEN_HILO_KEEP / _TAKE).  The lists do not depend on
    the choice (both screens are certified against the same exact values); only the number of re-encoded rows does.  Train-mode forwards, evaluation and
    the f16 MODE always keep the plain stream."""
    v = os.environ.get("GRIP_SCREEN_STREAM", "auto")
    if v not in ("auto", "hilo", "f16"):
        raise ValueError(f"GRIP_SCREEN_STREAM={v!r}: expected 'auto', 'hilo' or 'f16'")
    return _SCREEN_CHOICE.get(key, "hilo") if v == "auto" else v


def note_screen_bound(key, stream, stats):
    """Record what a pass measured, for screen_stream("auto"): the share of the pool its scan marked decides the next pass's stream.  The statistics are
    identical on every rank, so every rank decides alike."""
    if key is None or not stats or stats.get("rows", 0) == 0:
        return
    marked = max(0, stats["rows_refined"] - stats["calibration_rows"] - stats["audit_rows"] - stats.get("nonfinite_screen_rows", 0)) / stats["rows"]
    _SCREEN_CHOICE[key] = "hilo" if marked > (SCREEN_HILO_KEEP if stream == "hilo" else SCREEN_HILO_TAKE) else "f16"
    stats["screen_stream"] = stream
    stats["screen_marked_share"] = marked
    stats["screen_stream_next_pass"] = _SCREEN_CHOICE[key]


@torch.no_grad()
def encode_pool(visual_tower, images, chunk=880, prefix=None, out=None, screen=False):
    """Encode an ordered pool.  `images` is a tensor [N,3,R,R] (any device) or a callable
    (lo, hi) -> tensor for that slice.  With torch.distributed initialised the pool is sharded
    contiguously and the embeddings are all-gathered; returns [N, E] f32 on the device.
    screen: False, or the stream form ("hilo" / "f16") of a screen-and-refine pass's screen; True = screen_stream() without a pool history."""
    if screen is True:
        screen = screen_stream()
    n = images.shape[0] if torch.is_tensor(images) else images.n
    lo, hi, per = gdist.shard_range(n)
    dev = visual_tower.device
    local = torch.empty(max(hi - lo, 0), visual_tower.embed_dim, dtype=torch.float32, device=dev)
    visual_tower.encode_chunks(images, local, lo, hi, chunk, prefix, hilo=screen == "hilo")
    return gdist.allgather_rows(local, n, per, tag="pool_embeddings")


def leaderboard(probs, pred, paths, class_labels, k):
    """(filepaths, labels) exactly as utils/clip_pseudolabels.py:103-112 rebuilds them."""
    if k == K_ALL:
        return list(paths), [class_labels[int(j)] for j in pred]
    img, cls = engine.leaderboard_scan(probs, pred, path_ranks(paths), k)
    return [paths[i] for i in img], [class_labels[int(c)] for c in cls]


# ------------------------------------------------------------------------------------------ screen and refine
REFINE_CALIB_ROWS = 256     # rows re-encoded exactly up front to measure the cheaper tiers' deviation on THIS pool
REFINE_MIN_SAMPLE = 64      # ... and never fewer than this (calibration and audit alike; pools under 256 rows: a quarter of the pool, at least 16).  Was 16 until r06: the
                            # log-odds bound of the compensated screen is tighter than any bound before it, and 18 rows are a thin sample to take a maximum over
REFINE_SAFETY = 2.0         # bound = safety x the largest deviation seen on any row re-encoded so far (it only ever grows)
REFINE_SAFETY_MID = 4.0     # the same for the middle tier: its bound rests on a quarter of the calibration rows (an f32 row costs 2.5 split-f16 ones), so it is
                            # given twice the margin instead (8 x was measured: 293 instead of 222 f32 rows per pass on the bench pool, +0.025 s) -- at 1e-5-sized deviations the extra band holds a handful of rows
REFINE_ESCALATE_AFTER = 8   # rounds after which whatever is still un-refined moves up a tier in one go (pathological pools only, see refine_scan)
REFINE_AUDIT_ROWS = 256     # un-refined rows re-encoded AFTER the scan certified its lists, to check the bound they were trusted to ($GRIP_REFINE_AUDIT)
REFINE_AUDIT_ROWS_LARGE = 1024   # ... for pools of REFINE_AUDIT_LARGE_POOL rows and more (what the lists take on trust there rests on a four times larger hold-out)
REFINE_AUDIT_LARGE_POOL = 50000
REFINE_AUDIT_MID_ROWS = 32  # rows the MIDDLE tier is trusted on that an audit sends through the exact tower (its deviations are f32-rounding-sized and tightly
                            # distributed; every row the scan sends on to the exact tower adds to the sample; an f32 row costs 2.5 split-f16 ones)
REFINE_MAX_AUDITS = 4       # audits that may each widen the bound before everything left is simply re-encoded
REFINE_ABS_EPS = 1e-30      # absolute slack of an un-refined probability: below this a softmax output has no relative accuracy (denormals, 0)
_EPS_CAP = 9e5              # grip_leaderboard_scan_bounded takes relative bounds below 1e6 (a bound >= 1 already means "anything below")
_DELTA_CAP = 80.0           # ... and log-odds bounds below 700 (e^80 already turns every interval into (0, 1))
_KR = _KU = 2.0 ** -20      # the log-odds form's slack for the f32 evaluation of p and 1 - p (csrc/leaderboard.cpp: kR, kU)


def audit_rows_default(n=0):
    v = os.environ.get("GRIP_REFINE_AUDIT", "")
    return int(v) if v.strip() else (REFINE_AUDIT_ROWS_LARGE if n >= REFINE_AUDIT_LARGE_POOL else REFINE_AUDIT_ROWS)


def bound_form():
    """Form of the per-row bound the screen-and-refine pass works with: "odds" (default) -- every entry's odds p / (1 - p) are known to a factor
    e^{+-delta} (include/grip_amd.h, bound_form 1: the form a logit error takes; tight for the p ~ 0.9+ entries on the board thresholds of a peaked
    pool) -- or "relative" (rounds 3-5: one relative bound on every probability of a tier).  $GRIP_REFINE_BOUND."""
    v = os.environ.get("GRIP_REFINE_BOUND", "odds")
    if v not in engine.BOUND_FORMS:
        raise ValueError(f"GRIP_REFINE_BOUND={v!r}: expected 'odds' or 'relative'")
    return v


def _deviation(approx, better, abs_eps):
    """Largest relative deviation between the approximate probabilities `approx` and the more accurate `better` of the same
    rows, beyond the absolute slack, relative to the SMALLER of the two values (so it bounds |better - approx| / approx, the form
    the scan's intervals use, and |approx / better - 1|, the form include/grip_amd.h states, alike).  Pairs that both lie inside
    the absolute slack are covered by it; a pair with one value at 0 and the other above the slack deviates infinitely."""
    a, b = np.asarray(approx, dtype=np.float64), np.asarray(better, dtype=np.float64)
    big = np.maximum(a, b) > abs_eps
    if not big.any():
        return 0.0
    d = np.maximum(np.abs(a - b)[big] - abs_eps, 0.0)
    m = np.minimum(a, b)[big]
    with np.errstate(divide="ignore", invalid="ignore"):
        r = np.where(d > 0, d / m, 0.0)
    return float(np.max(r))


def odds_interval(p, delta, abs_eps):
    """(lo, hi) of the true probability given an approximate `p` whose row obeys the log-odds bound `delta`: the numpy restatement of RowBound::interval
    (csrc/leaderboard.cpp, bound_form 1) that _deviation_odds inverts (tests hold the two together)."""
    s = np.asarray(p, dtype=np.float64)
    E = np.exp(np.float64(delta))
    q = 1.0 - s
    a_hi, a_lo = s * (1.0 + _KR), s * (1.0 - _KR)
    q_lo, q_hi = np.maximum(q - _KU, 0.0), np.maximum(q, 0.0) + _KU
    with np.errstate(divide="ignore", invalid="ignore"):
        hi = a_hi / (a_hi + q_lo / E) * (1.0 + _KR) + abs_eps
        lo = a_lo / (a_lo + q_hi * E) * (1.0 - _KR) - abs_eps
    return lo, hi


def _deviation_odds(approx, better, abs_eps):
    """Smallest delta >= 0 for which every value of `better` lies inside the log-odds interval of the corresponding value of `approx`
    (odds_interval): the largest deviation of the rows' odds, in the form the scan uses it.  Pairs inside the absolute slack deviate by 0; an approximate
    0 against a better value above the slack deviates infinitely."""
    a, b = np.asarray(approx, dtype=np.float64).ravel(), np.asarray(better, dtype=np.float64).ravel()
    if a.size == 0:
        return 0.0
    q = 1.0 - a
    a_hi, a_lo = a * (1.0 + _KR), a * (1.0 - _KR)
    q_lo, q_hi = np.maximum(q - _KU, 0.0), np.maximum(q, 0.0) + _KU
    with np.errstate(divide="ignore", invalid="ignore"):
        bu = (b - abs_eps) / (1.0 + _KR)                   # upper side: bu <= a_hi / (a_hi + q_lo / E)
        need = bu > a_hi / (a_hi + q_lo)
        e_up = np.where(need, np.where((a_hi > 0) & (bu < 1.0), q_lo * bu / (a_hi * (1.0 - bu)), np.inf), 1.0)
        bl = (b + abs_eps) / (1.0 - _KR)                   # lower side: bl >= a_lo / (a_lo + q_hi E)
        need = bl < a_lo / (a_lo + q_hi)
        e_lo = np.where(need, np.where(bl > 0, a_lo * (1.0 - bl) / (bl * q_hi), np.inf), 1.0)
    e = max(1.0, float(np.max(e_up)), float(np.max(e_lo)))
    return float(np.log(e)) if np.isfinite(e) else float("inf")


def scan_placement():
    """Where the bounded scan of a multi-rank pass runs: "root" (default) -- rank 0 scans with the CPUs of the whole node and broadcasts lists and
    marks -- or "replicated" (every rank scans: what rounds 1-4 did).  The scan is sequential and does not shard; replicated, the g ranks
    of a node run it at the same moment on 1/g of the CPUs each.  $GRIP_SCAN_PLACEMENT."""
    v = os.environ.get("GRIP_SCAN_PLACEMENT", "root")
    if v not in ("root", "replicated"):
        raise ValueError(f"GRIP_SCAN_PLACEMENT={v!r}: expected 'root' or 'replicated'")
    return v


def _usable_cpus():
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except Exception:
        pass
    return n


def scan_bounded(probs, pred, ranks, rel, k, abs_eps, form="relative"):
    """engine.leaderboard_scan_bounded for the current process group: one rank, or placement "replicated": the local scan.  Placement "root": rank 0
    runs it on every CPU the process may use (the native default divides them by $LOCAL_WORLD_SIZE; the count travels as an argument, ABI 8) and the
    other ranks receive (img, cls, ambiguous) in one broadcast of k C + N / 4 words -- they spend no CPU on it and cannot disagree with rank 0
    (tests/test_dist_gloo.py).  The first word of that message is the pair count, or -1 when rank 0's scan failed: every rank then raises together
    instead of waiting in the broadcast for the collective's watchdog."""
    rank, ws = gdist.world()
    if ws == 1 or scan_placement() != "root":
        return engine.leaderboard_scan_bounded(probs, pred, ranks, rel, k, abs_eps, form=form)
    n, c = probs.shape
    cap = n if int(k) == K_ALL else c * max(1, min(int(k), n))
    words = (n + 3) // 4
    buf = np.zeros(1 + 2 * cap + words, dtype=np.int32)
    failure = None
    if rank == 0:
        try:
            threads = 0 if "GRIP_SCAN_THREADS" in os.environ else min(16, _usable_cpus())
            img, cls, amb = engine.leaderboard_scan_bounded(probs, pred, ranks, rel, k, abs_eps, form=form, threads=threads)
            buf[0] = len(img)
            buf[1: 1 + len(img)] = img
            buf[1 + cap: 1 + cap + len(cls)] = cls
            buf[1 + 2 * cap:].view(np.uint8)[:n] = amb
        except (engine.native.GripError, MemoryError) as e:
            failure = e
            buf[0] = -1
    gdist.broadcast_array_(buf)
    m = int(buf[0])
    if m < 0:
        raise failure if failure is not None else engine.native.GripError("bounded leaderboard scan failed on rank 0 (see its log)")
    return buf[1: 1 + m].copy(), buf[1 + cap: 1 + cap + m].copy(), buf[1 + 2 * cap:].view(np.uint8)[:n].astype(bool)


def refine_scan(probs, pred, ranks, k, exact_rows, calib=REFINE_CALIB_ROWS, safety=REFINE_SAFETY, max_rounds=64, mid_rows=None,
                audit=None, abs_eps=REFINE_ABS_EPS, bound=None):
    """Leaderboard lists of the reference's fp32 scan from probabilities of the f16 towers (utils/clip_pseudolabels.py:38-112).

    `probs` [N, C] f32 / `pred` [N] come from the f16 image tower (the SCREEN; modified in place); `exact_rows(idx)` returns the
    exact (f32-tower) probabilities and arg-max of the rows `idx` (ascending int64 array); `mid_rows(idx)`, optional, the same from
    a MIDDLE tier (the split-f16 tower: far more accurate than the screen, several times cheaper than the f32 tower).  A row
    sits at level 0 (screen), 1 (middle) or 2 (exact, final).  Level-0 / level-1 values are trusted only up to a relative bound
    eps[level] = safety x (largest deviation seen so far between a value of that level and the better value that later replaced it),
    plus an absolute slack `abs_eps` for values in the denormal range; the bounds start from `calib` rows -- at most 1/16 of the pool,
    at least REFINE_MIN_SAMPLE = 64 -- spread evenly over it: all of them through the next tier up, every fourth through the exact tower as well when there
    is a middle tier.  grip_leaderboard_scan_bounded marks every non-final row
    that takes part in a comparison its bound cannot decide; marked rows move up one tier and the scan repeats until nothing is
    marked and no bound has moved: the lists are then certified, decision by decision, to be those of the scan over the all-f32
    probabilities PROVIDED every non-final row obeys its bound.

    That proviso is a measured bound, not a theorem about f16 arithmetic, so it is AUDITED: after certification `audit` (default
    256, 1 024 for pools of 50 000 rows and more; $GRIP_REFINE_AUDIT; 0 = off; at most 1/16 of the pool but at least 64) rows still at level 0 -- half of them members of the final boards where there are any, the rest
    drawn uniformly from the pool (seeded: every rank draws the same rows) -- are re-encoded by the next tier; if one of them turns
    out to deviate by MORE than the bound it was trusted to, the bound was understated: it is widened to safety x that deviation, the scan
    repeats under it and is audited again (at most REFINE_MAX_AUDITS times, after which every row left is re-encoded).  stats reports the audit (`audit_rows`, `audit_max_deviation`, `audit_widened`), the rows the final lists
    still take on trust (`unverified_rows`) and the number of rows every bound rests on (`observed_rows`).

    `bound` ("odds" / "relative", default bound_form()): what a tier's bound says.  "relative": every probability of the row is within the relative
    eps of the truth.  "odds" (r06): every entry's odds p / (1 - p) are within a factor e^{+-delta} of the truth -- the same statement for the small
    entries, (1 - p) times tighter for the large ones, and the one an embedding-direction error actually produces (a logit error of
    scale x <de, t_c>, whatever the probability).  Deviations are measured in the same form, entry by entry (_deviation_odds), so calibration,
    growth and audit work unchanged; `eps` / `max_deviation` in the stats are then deltas.
    Returns (img, cls, stats)."""
    n, c = probs.shape
    form = bound_form() if bound is None else bound
    if form not in engine.BOUND_FORMS:
        raise ValueError(f"refine_scan: bound={form!r}")
    deviation = _deviation if form == "relative" else _deviation_odds
    cap = _EPS_CAP if form == "relative" else _DELTA_CAP
    audit = audit_rows_default(n) if audit is None else int(audit)
    if audit > 0:
        audit = min(n, max(min(REFINE_MIN_SAMPLE, max(16, n // 4)), min(audit, n // 16)))       # like the calibration sample: at most 1/16 of the pool, at least 64 (16 .. n / 4 on tiny pools)
    level = np.zeros(n, dtype=np.int8)
    dev = [0.0, 0.0]                # largest deviation seen of a level-0 / level-1 value from the better value that replaced it
    n_mid = n_exact = 0

    def submit(fn, idx):
        """Start a tier's re-encode of rows `idx`: tier callbacks with a `.submit(idx)` (the GPU tiers: work enqueued on the tier's own stream, nothing
        waited for) return a callable that waits and hands back (probs, argmax); plain callbacks run here and now.  Two tiers submitted back to back run side
        by side on the GPU -- the f32 tower's few hundred rows per pass no longer run alone at a third of its rate (r06)."""
        idx = np.asarray(idx, dtype=np.int64)
        if idx.size == 0:
            return None
        stats["tier_calls"] += 1
        if hasattr(fn, "submit"):
            return fn.submit(idx)
        out = fn(idx)
        return lambda: out

    def to_exact(idx, measure=True, pending=None):
        nonlocal n_exact
        idx = np.asarray(idx, dtype=np.int64)
        if idx.size == 0:
            return
        p32, a32 = (pending or submit(exact_rows, idx))()
        for lv in (0, 1):
            sel = (level[idx] == lv) & np.isfinite(probs[idx]).all(axis=1)       # (a non-finite row says nothing about the tier's accuracy)
            if measure and sel.any():
                dev[lv] = max(dev[lv], deviation(probs[idx[sel]], p32[sel], abs_eps))
        probs[idx] = p32
        pred[idx] = a32
        level[idx] = 2
        n_exact += idx.size

    def to_mid(idx, pending=None):
        nonlocal n_mid
        idx = np.asarray(idx, dtype=np.int64)
        if idx.size == 0:
            return
        pm, am = (pending or submit(mid_rows, idx))()
        n_mid += idx.size           # (rows the tier ENCODED: an overflow inside it is paid for all the same)
        bad = ~np.isfinite(pm).all(axis=1)
        if bad.any():       # an overflow inside the cheaper tower (f16 range): those rows go straight to the exact tower
            to_exact(idx[bad], measure=False)
            idx, pm, am = idx[~bad], pm[~bad], am[~bad]
            if idx.size == 0:
                return
        # the middle tier's value is itself only known to eps[1]: a screen value within d of it is within d (1 + eps1) + eps1 of the truth
        # (log-odds form: the two factors multiply, the deltas add)
        fin = np.isfinite(probs[idx]).all(axis=1)      # (a non-finite screen row says nothing about the tier's accuracy: as in to_exact)
        if fin.any():
            d = deviation(probs[idx[fin]], pm[fin], abs_eps)
            dev[0] = max(dev[0], d * (1.0 + eps[1]) + eps[1] if form == "relative" else d + eps[1] + 4 * _KR)
        probs[idx] = pm
        pred[idx] = am
        level[idx] = 1

    def up(idx):
        """Move rows up one tier: level 0 -> middle tier where there is one, everything else -> exact."""
        idx = np.asarray(idx, dtype=np.int64)
        lo = idx[level[idx] == 0] if mid_rows is not None else idx[:0]
        hi = idx[level[idx] == 1] if mid_rows is not None else idx
        h_lo, h_hi = submit(mid_rows, lo), submit(exact_rows, hi)        # both tiers in flight before either is waited for
        to_mid(lo, pending=h_lo)
        to_exact(hi, pending=h_hi)

    def bound(lv):
        return min((safety if lv == 0 else max(safety, REFINE_SAFETY_MID)) * dev[lv], cap)

    stats = {"rows": n, "calibration_rows": 0, "rounds": 0, "scans": 0, "tier_calls": 0, "audits": 0, "audit_rows": 0, "audit_board_rows": 0, "audit_max_deviation": 0.0,
             "audit_widened": False}
    eps = [0.0, 0.0]
    if n == 0:
        return np.empty(0, np.int32), np.empty(0, np.int32), dict(stats, rows_refined=0, rows_mid=0, rows_exact=0, eps=0.0, eps_mid=0.0, max_deviation=0.0,
                                                                    max_deviation_mid=0.0, refined_per_round=[], safety=float(safety), unverified_rows=0,
                                                                    observed_rows=0, tiers=2 + (mid_rows is not None))
    broken = np.flatnonzero(~np.isfinite(probs).all(axis=1))       # rows the screen overflowed on (f16 range): up a tier at once, outside every bound --
    stats["nonfinite_screen_rows"] = int(broken.size)               # and BEFORE the calibration, so that none of them can enter a measured deviation.
    if mid_rows is not None:            # The middle tier keeps an f32 residual stream: it resolves them at 2.5x the exact tower's rate (what it
        to_mid(broken)                  # overflows on itself goes on to the exact tower inside to_mid); r05 sent them to the exact tower directly
    else:
        to_exact(broken, measure=False)
    # calibration rows: `calib`, but at most 1/16 of the pool -- and never fewer than REFINE_MIN_SAMPLE (a bound from a handful of rows is no bound) -- spread evenly over
    # the rows the screen is still trusted on (a row it overflowed on measures nothing; with no such rows: over the pool, as before)
    live = np.flatnonzero(level == 0)
    want = min(live.size, max(min(REFINE_MIN_SAMPLE, max(16, n // 4)), min(calib, n // 16)))
    cal = live[np.unique(np.linspace(0, live.size - 1, want).astype(np.int64))] if want else live
    if cal.size == 0:
        pass                            # every row is final already (an all-non-finite screen): nothing to calibrate, nothing left to trust
    elif mid_rows is not None:
        # With a middle tier the screen's bound is calibrated against IT (every calibration row; its own error is folded in), and the middle
        # tier's bound against the exact tower on every fourth calibration row (at least 16): its deviations are f32-rounding-sized and tightly
        # distributed (three f16 products with f32 accumulation: ~3 x 2^-23 per term), the rows the scan sends on to the exact tower and the
        # audit keep adding to the sample, and an f32 row costs 2.5x a split-f16 one.
        cal_x = cal[:: max(1, len(cal) // max(16, len(cal) // 4))]
        rest = np.setdiff1d(cal, cal_x)
        h_mid, h_ex, h_rest = submit(mid_rows, cal_x), submit(exact_rows, cal_x), submit(mid_rows, rest)     # the three calibration encodes side by side
        pm_x, _ = h_mid()
        n_mid += cal_x.size
        p32_x, a32_x = h_ex()
        n_exact += cal_x.size
        fin = np.isfinite(pm_x).all(axis=1)            # (a middle-tier overflow measures nothing either; the row is exact now anyway)
        if fin.any():
            dev[1] = deviation(pm_x[fin], p32_x[fin], abs_eps)
        eps[1] = bound(1)
        fin = np.isfinite(probs[cal_x]).all(axis=1)
        if fin.any():
            dev[0] = deviation(probs[cal_x[fin]], p32_x[fin], abs_eps)
        probs[cal_x], pred[cal_x], level[cal_x] = p32_x, a32_x, 2
        to_mid(rest, pending=h_rest)
    else:
        to_exact(cal)
    stats["calibration_rows"] = int(cal.size)
    stats["calibration_rows_exact"] = int((level[cal] == 2).sum())
    eps = [bound(0), bound(1)]
    per_round = []
    g = np.random.default_rng(1000003 * n + int(min(k, 1 << 30)))      # the audit's draw: a function of the problem only (identical on every rank)
    while True:
        rel = np.where(level == 2, np.float32(0), np.where(level == 1, np.float32(eps[1]), np.float32(eps[0]))).astype(np.float32)
        img, cls, amb = scan_bounded(probs, pred, ranks, rel, k, abs_eps, form)
        stats["scans"] += 1
        todo = np.flatnonzero(amb & (level < 2))
        moved = bound(0) > eps[0] or bound(1) > eps[1]
        if todo.size == 0 and not moved:
            # certified under the current bounds: audit rows that are still taken on trust
            pending = np.flatnonzero(level == 0)
            if audit <= 0 or pending.size == 0:
                break
            if stats["audits"] >= REFINE_MAX_AUDITS:      # every audit so far widened the bound: it cannot be trusted on this pool
                stats["escalated"] = True
                up(pending)
                per_round.append(int(pending.size))
                stats["rounds"] += 1
                eps = [max(eps[0], bound(0)), max(eps[1], bound(1))]
                continue
            in_board = np.zeros(n, dtype=bool)
            if k != K_ALL:
                in_board[np.asarray(img, dtype=np.int64)] = True
            members = pending[in_board[pending]]
            n_b = min(members.size, audit // 2)
            pick_b = g.choice(members, size=n_b, replace=False) if n_b else np.empty(0, np.int64)
            others = pending[~in_board[pending]] if n_b else pending
            n_o = min(others.size, audit - n_b)
            pick_o = g.choice(others, size=n_o, replace=False) if n_o else np.empty(0, np.int64)
            sample = np.unique(np.concatenate([pick_b, pick_o]).astype(np.int64))
            before = list(dev)
            dev[0] = dev[1] = 0.0
            h_mids = None
            if mid_rows is not None:                        # ... and a few rows the middle tier is trusted on go through the exact tower (beside the screen's sample:
                mids = np.flatnonzero(level == 1)           # drawn before it moves up, submitted first so that the two tiers run side by side)
                sample_mid = np.sort(g.choice(mids, size=min(mids.size, max(min(audit // 8, REFINE_AUDIT_MID_ROWS), 1)), replace=False)) if mids.size else mids
                h_mids = submit(exact_rows, sample_mid)
            up(sample)
            if mid_rows is not None:
                to_exact(sample_mid, pending=h_mids)
                stats["audit_mid_rows"] = stats.get("audit_mid_rows", 0) + int(sample_mid.size)
                stats["audit_max_deviation_mid"] = max(stats.get("audit_max_deviation_mid", 0.0), float(dev[1]))
            seen = list(dev)
            stats["audits"] += 1
            stats["audit_rows"] += int(sample.size)
            stats["audit_board_rows"] += int(n_b)
            stats["audit_max_deviation"] = max(stats["audit_max_deviation"], float(seen[0]))
            if seen[0] <= eps[0] and seen[1] <= eps[1]:
                dev[0], dev[1] = before                     # a hold-out check of the bounds, passed: they stay what the lists were certified under
                break
            # a row beyond the bound it was trusted to: the bound was understated -- widen it (safety x what was just seen) and scan again
            dev[0], dev[1] = max(before[0], seen[0]), max(before[1], seen[1])
            stats["audit_widened"] = True
            eps = [max(eps[0], bound(0)), max(eps[1], bound(1))]
            continue
        if stats["rounds"] >= max_rounds:
            raise RuntimeError(f"refine_scan: no fixed point after {max_rounds} rounds ({int((level > 0).sum())} of {n} rows refined)")
        if stats["rounds"] >= REFINE_ESCALATE_AFTER:
            # Heavily tied scores can keep a board in the reference's unsorted regime, where every comparison has to be certain and a
            # round only advances a few images in dataset order: stop trickling and move everything that is left up a tier.
            todo = np.flatnonzero(level < 2)
            stats["escalated"] = True
        up(todo)
        per_round.append(int(todo.size))
        stats["rounds"] += 1
        eps = [max(eps[0], bound(0)), max(eps[1], bound(1))]
    stats.update(rows_refined=int((level > 0).sum()), rows_mid=int(n_mid), rows_exact=int(n_exact), refined_per_round=per_round, eps=float(eps[0]),
                 eps_mid=float(eps[1]), max_deviation=float(dev[0]), max_deviation_mid=float(dev[1]), safety=float(safety), safety_mid=float(max(safety, REFINE_SAFETY_MID)),
                 unverified_rows=int((level == 0).sum()), observed_rows=int((level > 0).sum()), tiers=2 + (mid_rows is not None), abs_eps=float(abs_eps),
                 bound_form=form)
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


SPLIT_TIER_MIN_ROWS = 4096      # pools below this go straight from the f16 screen to the f32 tower (the split twin costs HBM and a build)


def tier_streams():
    """HIP streams the re-encodes of a refinement tier alternate their chunks on ($GRIP_TIER_STREAMS, default 2 as for the pool encode: the split tower's
    bandwidth-bound kernels of one chunk run next to the GEMMs of the other, refine_split 0.81 -> 0.79 s per two passes; rows do not depend on it)."""
    return 1 if os.environ.get("GRIP_TIER_STREAMS", "2") == "1" else 2


def mid_tower(clip_model, n_rows):
    """The vision tower of `clip_model`'s split-f16 twin when the middle tier pays for a pool of `n_rows` rows, else None.
    $GRIP_SPLIT_TIER: "auto" (default: pools of >= SPLIT_TIER_MIN_ROWS rows), "1" always, "0" never."""
    import os
    want = os.environ.get("GRIP_SPLIT_TIER", "auto")
    if want == "0" or (want == "auto" and n_rows < SPLIT_TIER_MIN_ROWS) or not hasattr(clip_model, "split_twin"):
        return None
    if clip_model.dims.vision_width % 256:      # the split GEMM's 256 x 256 tiles (grip_tower_create refuses other widths at precision 2)
        return None
    try:
        twin = clip_model.split_twin()
    except engine.native.GripError as e:        # the tier is an optimisation: without it the screen's rows go straight to the f32 tower
        import logging
        logging.getLogger(__name__).warning("split-f16 middle tier unavailable (%s): refining with the f32 tower only", e)
        return None
    return None if twin is None else twin.visual.tower


def take_images(images, idx):
    """Rows `idx` (ascending int64 array) of an image pool: a tensor [N,3,R,R] or a lazy pool with .take(idx)."""
    if torch.is_tensor(images):
        return images[torch.as_tensor(idx, device=images.device)]
    return images.take(idx)


def balanced_chunk(n_rows, chunk):
    """Rows per launch when `n_rows` rows go through a tower at most `chunk` at a time: the smallest size that keeps the number of launches
    (1 024 audit rows at 880 -> 2 x 512, not 880 + 144: the tail launch left the 256 x 256 GEMM tiles of the refinement towers a chip's worth of
    workgroups short).  Rows do not depend on the chunking."""
    parts = max(1, -(-int(n_rows) // max(1, int(chunk))))
    return max(1, -(-int(n_rows) // parts))


def tier_rows(tower, fetch, txt, scale, n, lo, hi, chunk, prefix=None, argmax_on="probs", on_rows=None, timer=None):
    """A refinement tier as refine_scan's callback: rows(idx) -> (probs [len(idx), C], arg-max) of the global rows `idx` (ascending) re-encoded by `tower`.
    Each rank encodes the rows of its own shard [lo, hi) -- `fetch(global_rows)` returns their images -- and one padded all-gather assembles the rest.
    rows.submit(idx) only ENQUEUES the work, on the tier's own HIP stream, and returns the function that waits for it: two tiers submitted back to back
    (refine_scan does that wherever their row sets are independent) share the GPU instead of taking turns at small-batch efficiency."""
    dev = tower.device
    side = torch.cuda.Stream(device=dev)

    def submit(idx):
        import time
        t0 = time.perf_counter()
        mine = idx[(idx >= lo) & (idx < hi)]
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            local = torch.empty(len(mine), tower.embed_dim, dtype=torch.float32, device=dev)
            if len(mine):
                tower.encode_chunks(lambda a, b: fetch(mine[a:b]), local, 0, len(mine), balanced_chunk(len(mine), chunk), prefix, streams=tier_streams())
            if on_rows is not None:
                on_rows(len(mine))
            got = gdist.allgather_selected(local, idx, n, tag="refined_rows")
            _, p, al, ap = engine.cosine_head(got, txt, scale)
            am = ap if argmax_on == "probs" else al
        if timer is not None:
            timer(time.perf_counter() - t0)

        def result():
            t1 = time.perf_counter()
            side.synchronize()
            out = p.cpu().numpy(), am.cpu().numpy()
            if timer is not None:
                timer(time.perf_counter() - t1)
            return out
        return result

    def rows(idx):
        return submit(idx)()
    rows.submit = submit
    return rows


@torch.no_grad()
def identical_lists(visual16, visual32, images, txt_exact, scale, paths, class_labels, k, chunk=880, exact_chunk=880, prefix=None,
                    argmax_on="probs", streams=2, emb16=None, visual_mid=None, mid_chunk=880):
    """(filepaths, labels) of the reference's fp32 pseudolabel scan (utils/clip_pseudolabels.py:24-112) at close to the f16
    towers' throughput: the whole pool goes through the f16 vision tower `visual16` (sharded over ranks, one all-gather), the
    head scores it against the EXACT text features `txt_exact`, and refine_scan re-encodes only the rows whose probabilities cannot
    decide a comparison the lists depend on -- with the split-f16 tower `visual_mid` (precision 2) where there is one, and with the
    f32 tower `visual32` what that tier cannot decide either (each rank re-encodes the marked rows of its own shard; one small
    all-gather per round and tier).  The lists are the exact mode's lists provided every row obeys the measured bound of the tier it
    was left at; the bound is calibrated on this pool, audited on a hold-out sample after certification and reported in
    LAST_REFINE_STATS (asserted equal to the exact mode at N = 50 000 in tests/test_gpu_identical.py)."""
    global LAST_REFINE_STATS
    n = len(paths)
    if n == 0:
        _, _, LAST_REFINE_STATS = refine_scan(np.empty((0, max(len(class_labels), 1)), np.float32), np.empty(0, np.int32), np.empty(0, np.int64), k, None)
        LAST_REFINE_STATS["rows_refined_this_rank"] = 0
        return [], []
    key = (id(visual16), n, len(class_labels))        # the pool as far as the screen's choice of stream goes (same tower, same size, same class count)
    stream = screen_stream(key) if emb16 is None else None
    emb = emb16 if emb16 is not None else encode_pool(visual16, images, chunk=chunk, prefix=prefix, screen=stream)
    dev = emb.device
    _, probs, am_l, am_p = engine.cosine_head(emb, txt_exact, scale)
    probs_h = probs.cpu().numpy()
    pred_h = (am_p if argmax_on == "probs" else am_l).cpu().numpy()
    lo, hi, _ = gdist.shard_range(n)
    encoded = {"exact": 0, "mid": 0}

    def rows_through(tower, tier, tier_chunk):
        def count(m):
            encoded[tier] += m
        return tier_rows(tower, lambda rows: take_images(images, rows), txt_exact, scale, n, lo, hi, tier_chunk, prefix, argmax_on, on_rows=count)

    img, cls, stats = refine_scan(probs_h, pred_h, path_ranks(paths), k, rows_through(visual32, "exact", exact_chunk),
                                  mid_rows=rows_through(visual_mid, "mid", mid_chunk) if visual_mid is not None else None)
    stats["rows_refined_this_rank"] = encoded["exact"] + encoded["mid"]
    stats["rows_exact_this_rank"], stats["rows_mid_this_rank"] = encoded["exact"], encoded["mid"]
    if stream is not None:
        note_screen_bound(key, stream, stats)
    LAST_REFINE_STATS = stats
    return [paths[i] for i in img], [class_labels[int(c)] for c in cls]


@torch.no_grad()
def pseudolabel_from_features(img_emb, txt_emb, scale, paths, class_labels, k, argmax_on="probs"):
    """Head + scan.  argmax_on: "probs" (compute_pseudo_labels, :39) or "logits" (assign_pseudo_labels)."""
    logits, probs, am_l, am_p = engine.cosine_head(img_emb, txt_emb, scale)
    pred = (am_p if argmax_on == "probs" else am_l).cpu().numpy()
    return leaderboard(probs.cpu().numpy(), pred, paths, class_labels, k)
