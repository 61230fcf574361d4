"""The N > 1 path on the GPU box (which has ONE GPU): two ranks share cuda:0 and talk over gloo
(GRIP_SINGLE_DEVICE=1, GRIP_DIST_BACKEND=gloo).  Checks that the sharded pool encode + all-gather returns the
embeddings of a single-process run in dataset order (ragged last shard included), that both ranks build the
same pseudolabel lists, and that bench.py's multi-rank branch runs end to end and prints one JSON line."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

from conftest import REPO

pytestmark = pytest.mark.gpu

WORKER = r'''
import os, sys, pickle, torch
sys.path.insert(0, os.environ["GRIP_REPO"])
import grip_amd
from grip_amd import clip, dist as gdist, pseudolabels as pl, rng
rank, ws = gdist.init_from_env()
m, _ = clip.load("small", device="cuda")
n = 101
x = torch.from_numpy(rng.normal(5, rng.stream_id("dist.x"), (n, 3, 64, 64))) + 2 * torch.from_numpy(rng.normal(5, rng.stream_id("dist.mu"), (n, 3, 1, 1)))
with torch.no_grad():
    emb = pl.encode_pool(m.visual.tower, x, chunk=16)
    # the trained-model pass of GRIP (assign_pseudo_labels): the same sharded encode with a visual prompt
    prefix = torch.from_numpy(rng.normal(5, rng.stream_id("dist.prefix"), (1, 4, 256), 0.0, 0.05)).cuda()
    emb_p = pl.encode_pool(m.visual.tower, x, chunk=16, prefix=prefix)
    txt = m.encode_text(clip.tokenize(["a", "b c", "d e f", "g"]).cuda())
lists = pl.pseudolabel_from_features(emb, txt, 100.0, [f"p{i:03d}" for i in range(n)], [0, 1, 2, 3], 5)
g = torch.ones(3, device="cuda") * (rank + 1)
gdist.allreduce_mean_([g])
with open(os.environ["GRIP_OUT"] + f".{rank}", "wb") as f:
    pickle.dump({"emb": emb.cpu(), "emb_p": emb_p.cpu(), "lists": lists, "g": g.cpu(), "ws": ws}, f)
gdist.barrier()
'''


def _port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _env(tmp_path):
    return dict(os.environ, GRIP_SINGLE_DEVICE="1", GRIP_DIST_BACKEND="gloo", GRIP_REPO=REPO, GRIP_OUT=str(tmp_path / "out"), PYTHONPATH=REPO,
                HSA_ENABLE_IPC_MODE_LEGACY="0")


def test_sharded_encode_allgather_matches_single_process(tmp_path):
    import pickle
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = _env(tmp_path)
    one = subprocess.run([sys.executable, str(script)], env=dict(env, GRIP_OUT=str(tmp_path / "single")), capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(_port()), str(script)], env=env, capture_output=True, text=True, timeout=900)
    assert two.returncode == 0, two.stderr[-3000:]
    ref = pickle.load(open(str(tmp_path / "single") + ".0", "rb"))
    r0 = pickle.load(open(str(tmp_path / "out") + ".0", "rb"))
    r1 = pickle.load(open(str(tmp_path / "out") + ".1", "rb"))
    assert ref["ws"] == 1 and r0["ws"] == 2
    assert torch.equal(r0["emb"], ref["emb"]) and torch.equal(r1["emb"], ref["emb"])     # rows are independent of the chunking
    assert torch.equal(r0["emb_p"], ref["emb_p"]) and torch.equal(r1["emb_p"], ref["emb_p"]) and not torch.equal(ref["emb_p"], ref["emb"])
    assert r0["lists"] == ref["lists"] and r1["lists"] == ref["lists"]
    assert torch.allclose(r0["g"], torch.full((3,), 1.5)) and torch.allclose(r1["g"], torch.full((3,), 1.5))


def test_bench_two_ranks(tmp_path):
    env = _env(tmp_path)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(_port()), os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                          "--pool", "1320", "--chunk", "220"], env=env, capture_output=True, text=True, timeout=1200, cwd=tmp_path)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["selected_pairs"] > 0 and "cpu_baseline" not in d
    assert d["roofline"]["bound"] == "mfma" and d["roofline"]["achieved"] > 0


def test_native_rccl_communicator_single_rank():
    """The C ABI's RCCL entry points (grip_comm_*, grip_allgather_embeddings, grip_allreduce_mean) execute on this box's one
    GPU: a 1-rank communicator is created through RCCL, the all-gather returns the local rows, the mean all-reduce is the
    identity.  (More than one rank per GPU is refused by RCCL itself; the N > 1 data path is covered over gloo above and its
    RCCL form is measured by the driver's multi-GPU bench.)"""
    import grip_amd  # noqa: F401
    from grip_amd import dist as gdist
    torch.cuda.set_device(0)
    c = gdist.NativeComm()
    try:
        assert (c.rank, c.ws) == (0, 1)
        local = torch.randn(37, 512, device="cuda")
        out = c.allgather(local, 37)
        g = torch.randn(8192, device="cuda")
        want = g.clone()
        c.allreduce_mean_(g)
        torch.cuda.synchronize()
        assert torch.equal(out, local) and torch.equal(g, want)
    finally:
        c.close()
