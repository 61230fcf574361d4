"""is the c_fc -> c_proj hidden round trip (3.2 GB per layer and 1 320-image chunk) what the pool-encode GEMMs wait for?
Timing ablations on the developer build `make -C menghini-neurips23-code_amd/csrc ablate` (libgrip_ablate.so: gemm.hip with -DGRIP_ABLATE):
    mask 1  the c_fc epilogue (gemm_k64p_kernel<8, 2, true>) computes everything and issues NO global store
    mask 2  the K = 3 072 residual GEMM (c_proj) reads its A operand from a 20-panel window (31 MB: Infinity-Cache resident) instead of the 1.6 GB hidden
    mask 4  the c_fc stores are issued, but into a 31-MB window of the hidden buffer (same instructions, no HBM write-back pressure): separates the
            store ISSUE from the store TRAFFIC
    mask 3  both = what a fused c_fc -> QuickGELU -> c_proj kernel would save in memory traffic, at zero cost for the fusion itself
Same process, same box, same resident pool, modes interleaved (0 1 2 3 0 1 2 3 ...); results are wrong by design under a mask, timing is not.
The hidden buffer keeps the real values of the warm-up pass, so c_proj multiplies realistic data (power / clock as in production).
    GRIP_LIB=menghini-neurips23-code_amd/libgrip_ablate.so python tools/mlp_ablation.py [rounds]"""
import ctypes
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
os.environ.setdefault("GRIP_LIB", os.path.join(REPO, "menghini-neurips23-code_amd", "libgrip_ablate.so"))
import bench  # noqa: E402
import grip_amd  # noqa: E402,F401
from grip_amd import clip, native  # noqa: E402

dev = torch.device("cuda", 0)
lib = native.lib()
ablate = ctypes.CDLL(native.LIB_PATH).grip_debug_ablate
m, _ = clip.load("ViT-B/16", device=dev)
chunk, n = 1320, 1320 * 12
pool = bench.synth_pool(n, 224, dev, 1234)
out = torch.empty(n, 512, device=dev)
tower = m.visual.tower
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4


def one(mask):
    ablate(mask)
    torch.cuda.synchronize()
    lib.grip_profile_enable(1)
    sampler = bench.clock_sampler()
    if sampler is not None:
        sampler.start()
    t = time.perf_counter()
    with torch.no_grad():
        tower.encode_chunks(pool, out, 0, n, chunk, streams=1)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    clk = sampler.stop() if sampler is not None else None
    launches, ms, fl = bench.profile_collect(lib)
    ablate(0)
    per = {}
    for slot in np.flatnonzero(launches):
        name = bench.kname(int(slot))
        if "k64p" in name:
            per[name.split(" [")[1].rstrip("]")] = (fl[slot] / (ms[slot] * 1e-3) / 1e12, ms[slot] / launches[slot])
    return n / dt, per, clk


with torch.no_grad():
    tower.encode_chunks(pool, out, 0, n, chunk, streams=1)       # warm-up, un-ablated: the hidden buffer holds real values from here on
torch.cuda.synchronize()
MASKS = (0, 1, 2, 3, 4, 6)
acc = {k: [] for k in MASKS}
for r in range(rounds):
    for mask in MASKS:
        ips, per, clk = one(mask)
        acc[mask].append((ips, per, clk))
names = {0: "baseline", 1: "(a) c_fc without its store", 2: "(b) c_proj A from a 31-MB window", 3: "(c) both", 4: "(d) c_fc stores into a 31-MB window", 6: "(e) = (b) + (d)"}
base = np.mean([a[0] for a in acc[0]])
for mask in MASKS:
    ips = np.mean([a[0] for a in acc[mask]])
    line = f"{names[mask]:34s} encode {ips:8.0f} img/s ({(ips / base - 1) * 100:+5.1f} %)"
    for epi in ("EPI_LNFOLD_F16", "EPI_LNFOLD_GELU_F16", "EPI_BIAS_RESID_STATS"):
        tf = np.mean([a[1][epi][0] for a in acc[mask] if epi in a[1]])
        us = np.mean([a[1][epi][1] for a in acc[mask] if epi in a[1]]) * 1e3
        line += f" | {epi[4:]:18s} {tf:6.0f} TF/s {us:7.1f} us"
    clks = [a[2] for a in acc[mask] if a[2]]
    if clks:
        try:
            line += f" | sclk {np.mean([c['sclk_mhz_mean'] for c in clks]):.0f} MHz {np.mean([c['power_w_mean'] for c in clks]):.0f} W"
        except Exception:
            pass
    print(line, flush=True)
