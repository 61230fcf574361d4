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
