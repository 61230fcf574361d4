"""CPU, world_size 2 over gloo: the N > 1 plumbing -- contiguous pool sharding + all-gather of
embeddings in dataset order (with a ragged last shard), and the prompt-gradient mean all-reduce."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import REPO


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, ws, port, n, e, ret):
    import sys
    sys.path.insert(0, REPO)
    import grip_amd  # noqa: F401
    from grip_amd import dist as gdist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(ws), LOCAL_RANK=str(rank))
    assert gdist.init_from_env(backend="gloo") == (rank, ws)
    full = torch.arange(n * e, dtype=torch.float32).reshape(n, e)
    lo, hi, per = gdist.shard_range(n)
    out = gdist.allgather_rows(full[lo:hi].clone(), n, per)
    ok = torch.equal(out, full)
    # a scattered selection (the rows a screen-and-refine scan asks for): every rank contributes the rows of its own shard
    import numpy as np
    for sel in ([0, n - 1], list(range(0, n, 3)), [n // 2], list(range(n)), []):
        idx = np.array(sorted(set(i for i in sel if 0 <= i < n)), dtype=np.int64)
        mine = idx[(idx >= lo) & (idx < hi)]
        got = gdist.allgather_selected(full[torch.from_numpy(mine)].clone(), idx, n)
        ok = ok and torch.equal(got, full[torch.from_numpy(idx)])
    # the trainer's evaluation loops: every rank scores ITS batches of the dataset order (dist.rank_batches = accelerate's even sharding), the rows
    # come back in dataset order with the padded duplicates dropped (accelerator.gather + drop_duplicates, textual_prompt.py:285-294)
    for bs in (4, 3, 16):
        mine = [i for b in gdist.rank_batches(range(n), bs) for i in b]
        got = gdist.gather_in_dataset_order(full[torch.tensor(mine, dtype=torch.long)].clone(), n, bs)
        ok = ok and torch.equal(got, full)
    t = torch.tensor([1.0 + rank, 10.0], dtype=torch.float64)
    gdist.allreduce_sum_(t)
    ok = ok and t.tolist() == [3.0, 20.0]
    g1 = torch.full((3, 4), float(rank + 1))
    g2 = torch.full((5,), float(10 * (rank + 1)))
    gdist.allreduce_mean_([g1, g2])
    ok = ok and torch.allclose(g1, torch.full((3, 4), 1.5)) and torch.allclose(g2, torch.full((5,), 15.0))
    gdist.barrier()
    ret[rank] = bool(ok)
    dist.destroy_process_group()


@pytest.mark.parametrize("n", [10, 7, 1])
def test_allgather_and_allreduce_world_size_2(n):
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, 6, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert ret[0] and ret[1]


@pytest.mark.parametrize("n,bs,ws", [(70, 16, 2), (64, 16, 2), (65, 16, 4), (10, 16, 2), (33, 16, 2), (48, 16, 4), (1, 16, 8), (1632, 16, 8), (17, 4, 3), (5, 1, 2)])
def test_rank_batches_are_accelerates_even_shards(n, bs, ws):
    """dist.rank_batches restates what `accelerator.prepare(DataLoader)` gives each process of the reference (accelerate's BatchSamplerShard with its
    defaults split_batches=False, even_batches=True: methods_config/accelerate_config.yml + e.g. textual_prompt.py:239) -- checked against the
    installed accelerate itself, on a shuffled order."""
    import sys
    sys.path.insert(0, REPO)
    import grip_amd  # noqa: F401
    from grip_amd import dist as gdist
    from accelerate.data_loader import BatchSamplerShard
    from torch.utils.data import BatchSampler
    order = torch.randperm(n, generator=torch.Generator().manual_seed(n * 31 + bs)).tolist()
    for r in range(ws):
        want = [list(b) for b in BatchSamplerShard(BatchSampler(order, bs, False), ws, r)]
        got = gdist.rank_batches(order, bs, r, ws)
        assert got == want, (r, got, want)
        assert all(len(b) == bs for b in got)
    assert gdist.rank_batches(order, bs, 0, 1) == [order[i: i + bs] for i in range(0, n, bs)]          # one process: the ragged tail stays
    assert gdist.rank_batches([], bs, 0, ws) == []


def _scan_worker(rank, ws, port, placement, ret):
    import sys
    sys.path.insert(0, REPO)
    import numpy as np
    import grip_amd  # noqa: F401
    from grip_amd import dist as gdist, engine, pseudolabels as pl
    from oracle import cbind
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(ws), LOCAL_RANK=str(rank), LOCAL_WORLD_SIZE=str(ws),
                      GRIP_SCAN_PLACEMENT=placement)
    gdist.init_from_env(backend="gloo")
    r = np.random.RandomState(7)
    n, c, k = 6000, 12, 5
    lg = (r.randn(n, c) * 0.4).astype(np.float32)
    z = np.exp(lg - lg.max(1, keepdims=True))
    p32 = (z / z.sum(1, keepdims=True)).astype(np.float32)
    p16 = (p32.astype(np.float64) * (1 + np.clip(r.randn(n, c), -5, 5) * 2e-3)).astype(np.float32)
    a32, a16 = p32.argmax(1).astype(np.int32), p16.argmax(1).astype(np.int32)
    paths = [f"p/{(i * 7919) % 100000:05d}_{i}.jpg" for i in range(n)]
    calls = []
    real = engine.leaderboard_scan_bounded
    engine.leaderboard_scan_bounded = lambda *a, **kw: (calls.append(1), real(*a, **kw))[1]
    img, cls, st = pl.refine_scan(p16.copy(), a16.copy(), pl.path_ranks(paths), k, lambda idx: (p32[idx], a32[idx]))
    want = cbind.leaderboard_ref(p32, a32, paths, list(range(c)), k)
    ok = ([paths[i] for i in img], [int(j) for j in cls]) == want
    ret[rank] = (bool(ok), len(calls), st["scans"], os.environ.get("GRIP_SCAN_THREADS"))
    gdist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("placement", ["root", "replicated"])
def test_bounded_scan_runs_on_rank_0_and_is_broadcast(placement):
    """under N ranks the sequential bounded scan runs on rank 0 only (with the node's CPUs) and every rank ends with rank 0's lists and
    marks; GRIP_SCAN_PLACEMENT=replicated keeps the per-rank scans.  Either way the refined lists are the oracle's on every rank."""
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_scan_worker, args=(r, 2, port, placement, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    (ok0, calls0, scans0, env0), (ok1, calls1, scans1, env1) = ret[0], ret[1]
    assert ok0 and ok1 and scans0 == scans1 >= 2 and env0 is None and env1 is None
    assert calls0 == scans0 and calls1 == (0 if placement == "root" else scans1)


def _failing_scan_worker(rank, ws, port, trace, ret):
    import sys
    sys.path.insert(0, REPO)
    import numpy as np
    import grip_amd  # noqa: F401
    from grip_amd import dist as gdist, engine, pseudolabels as pl
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(ws), LOCAL_RANK=str(rank), LOCAL_WORLD_SIZE=str(ws),
                      GRIP_SCAN_PLACEMENT="root", GRIP_COMM_TRACE=trace)
    gdist.init_from_env(backend="gloo")
    r = np.random.RandomState(3)
    n, c = 500, 6
    p = r.dirichlet(np.ones(c), size=n).astype(np.float32)
    a = p.argmax(1).astype(np.int32)
    ranks = pl.path_ranks([f"p/{i:05d}.jpg" for i in range(n)])
    rel = np.full(n, 1e-2, np.float32)
    img, cls, amb = pl.scan_bounded(p, a, ranks, rel, 4, 1e-30, "odds")            # a healthy scan first: one broadcast on both ranks
    healthy = len(img) > 0 and amb.dtype == bool
    if rank == 0:
        def boom(*args, **kw):
            raise engine.native.GripError("bounded leaderboard: out of memory (injected)")
        engine.leaderboard_scan_bounded = boom
    try:
        pl.scan_bounded(p, a, ranks, rel, 4, 1e-30, "odds")
        raised = None
    except engine.native.GripError as e:
        raised = str(e)
    gdist.barrier()                                                                  # (nobody is left waiting inside the broadcast)
    ret[rank] = (healthy, raised)
    dist.destroy_process_group()


def test_a_failing_root_scan_raises_on_every_rank_instead_of_hanging(tmp_path):
    """with the scan placed on rank 0, a failure there used to leave the other ranks inside the broadcast until the collective's watchdog fired.
    The broadcast now starts with a status word: rank 0 re-raises its own error, every other rank raises a GripError that points at rank 0 -- and
    $GRIP_COMM_TRACE shows the two broadcasts (the healthy scan's and the failed one's) on both ranks."""
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    trace = str(tmp_path / "comm.log")
    procs = [ctx.Process(target=_failing_scan_worker, args=(r, 2, port, trace, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert ret[0][0] and ret[1][0]
    assert ret[0][1] is not None and "injected" in ret[0][1]
    assert ret[1][1] is not None and "rank 0" in ret[1][1]
    lines = open(trace).read().splitlines()
    for r in (0, 1):
        assert sum(l.startswith(f"rank{r} broadcast ") for l in lines) == 2, lines
