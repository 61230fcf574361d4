"""Host-side cost of a pseudolabel pass as the pool grows (CPU only; multi-GPU readiness without the hardware):
under weak scaling every rank runs the replicated leaderboard scan over ALL N_total rows (the scan is sequential
and does not shard), so its cost grows with the number of GPUs while the per-rank encode stays put.

    python tools/scan_scale.py [--classes 102] [--k 16] [--sizes 50000,100000,200000,400000]

Times, per pool size (N x C f32 probabilities shaped like the bench pool: one dominant class, near-tied columns, f16-like noise):
`path_ranks` (first call / cached), the plain scan (grip_leaderboard_scan), the bounded scan with the f16 tier's bound
(grip_leaderboard_scan_bounded, first round: nothing refined) and with everything final."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import grip_amd  # noqa: E402,F401
from grip_amd import engine, pseudolabels as pl  # noqa: E402


def pool(n, c, seed=0):
    r = np.random.default_rng(seed)
    lg = (r.standard_normal((n, c), dtype=np.float32) * 0.045 + r.standard_normal((1, c), dtype=np.float32) * 0.3)
    lg[:, 1] += 2.0
    z = np.exp(lg - lg.max(1, keepdims=True))
    p = (z / z.sum(1, keepdims=True)).astype(np.float32)
    p16 = (p * (1.0 + np.clip(r.standard_normal((n, c), dtype=np.float32), -5, 5) * np.float32(2.3e-3))).astype(np.float32)
    return p, p16


def best(f, reps=3):
    ts = []
    for _ in range(reps):
        t = time.perf_counter()
        f()
        ts.append(time.perf_counter() - t)
    return min(ts)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--classes", type=int, default=102)
    ap.add_argument("--k", type=int, default=16)
    ap.add_argument("--sizes", default="50000,100000,200000,400000")
    a = ap.parse_args()
    print(f"C = {a.classes}, k = {a.k}, {os.cpu_count()} CPUs; seconds (best of 3)")
    print(f"{'N':>8} {'path_ranks':>11} {'(cached)':>9} {'plain scan':>11} {'bounded eps=2.4e-2':>19} {'bounded eps=0':>14} {'marked':>7}")
    for n in [int(x) for x in a.sizes.split(",")]:
        p, p16 = pool(n, a.classes)
        paths = [f"/data/pool/train/{(i * 7919) % 100000:05d}_{i}.jpg" for i in range(n)]

        def first():
            pl._RANK_CACHE.clear()
            pl.path_ranks(paths)
        t_rank = best(first)
        ranks = pl.path_ranks(paths)
        t_cached = best(lambda: pl.path_ranks(list(paths)))
        pred, pred16 = p.argmax(1).astype(np.int32), p16.argmax(1).astype(np.int32)
        t_plain = best(lambda: engine.leaderboard_scan(p, pred, ranks, a.k))
        rel = np.full(n, 2.4e-2, np.float32)
        t_b = best(lambda: engine.leaderboard_scan_bounded(p16, pred16, ranks, rel, a.k, 1e-30))
        marked = int(engine.leaderboard_scan_bounded(p16, pred16, ranks, rel, a.k, 1e-30)[2].sum())
        zero = np.zeros(n, np.float32)
        t_z = best(lambda: engine.leaderboard_scan_bounded(p, pred, ranks, zero, a.k, 1e-30))
        print(f"{n:8d} {t_rank:11.3f} {t_cached:9.3f} {t_plain:11.3f} {t_b:19.3f} {t_z:14.3f} {marked:7d}", flush=True)


if __name__ == "__main__":
    main()
