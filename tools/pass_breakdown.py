import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, numpy as np
import bench
import argparse
a = argparse.Namespace(pool=50000, chunk=1320, streams=1, classes=102, prefix=16, k=16, batch=16)
import grip_amd
from grip_amd import engine, dist as gdist
loop = bench.Loop(a, torch.device("cuda:0"), 0, 1)
loop.step()
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.no_grad():
        txt = loop.m.encode_text(loop.zs_tokens)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        local = torch.empty(a.pool, loop.d.embed_dim, dtype=torch.float32, device="cuda")
        loop.m.visual.tower.encode_chunks(loop.pool, local, 0, a.pool, a.chunk, streams=1)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        logits, probs, am_l, am_p = engine.cosine_head(local, txt, loop.m.logit_scale.exp().item())
        torch.cuda.synchronize(); t3 = time.perf_counter()
        ph = probs.cpu().numpy(); pr = am_p.cpu().numpy()
        t4 = time.perf_counter()
    img, cls = engine.leaderboard_scan(ph, pr, loop.ranks, a.k)
    t5 = time.perf_counter()
    print(f"text {1e3*(t1-t0):.1f} ms | encode {1e3*(t2-t1):.1f} | head {1e3*(t3-t2):.1f} | d2h {1e3*(t4-t3):.1f} | scan {1e3*(t5-t4):.1f} | total {1e3*(t5-t0):.1f}")
