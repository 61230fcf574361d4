"""Developer tool (GPU): where the f16 screen's deviation from the f32 towers comes from, and what it does to the screen-and-refine bound.

  (1) real towers: direction error of the f16 / split-f16 embeddings against the f32 twin's on the bench pool, the structured pool and the stress
      model; the same with every GEMM-operand weight on the f16 grid (what every published CLIP checkpoint holds: the reference's CPU path is
      clip.load(..., "cpu") = those fp16 weights cast up, so the f16 towers round NO weight there) -- how much of the deviation is weight rounding;
  (2) a torch emulation of the f16 tower's arithmetic (f16 operands, f32 accumulation, LayerNorm folded into the consumer GEMM) with the residual
      stream kept in f16 (the product) or in f32 (the pre-r01.e form): how much of the deviation the 24 roundings of the stream are (VERDICT r5 #1b);
  (3) geometry of the synthetic embeddings (how far the images spread around their mean direction) -- what a "realistic" synthetic text-feature set
      can be built from -- and the logit / log-odds deviations against candidate text features;
  (4) dumps e32 and (e16 - e32) of N rows per model to gpurun_out/ for offline work on the scan.

    python tools/delta_probe.py [--rows 8192] [--emu-rows 512]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402
import grip_amd  # noqa: E402,F401
from grip_amd import weights as W, config  # noqa: E402
from grip_amd.clip.clip import load_openai_state_dict  # noqa: E402
from grip_amd.clip.model import CLIP  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=8192)
ap.add_argument("--emu-rows", type=int, default=512)
ap.add_argument("--classes", type=int, default=102)
ap.add_argument("--out", default="gpurun_out/delta_probe")
a = ap.parse_args()
dev = torch.device("cuda", 0)
os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
report = {}


def structured(n, seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    pool = torch.empty(n, 3, 224, 224, device=dev)
    ramp = torch.linspace(-1.0, 1.0, 224, device=dev).view(1, 1, 1, -1)
    for lo in range(0, n, 2048):
        hi = min(lo + 2048, n)
        x = torch.empty(hi - lo, 3, 224, 224, device=dev).normal_(generator=g)
        mu = torch.empty(hi - lo, 3, 1, 1, device=dev).normal_(generator=g) * 2.0
        r = torch.empty(hi - lo, 3, 1, 1, device=dev).normal_(generator=g)
        pool[lo:hi] = x * 0.5 + mu + ramp * r
    return pool


def unit(e):
    return e / e.norm(dim=-1, keepdim=True)


on_grid = W.on_f16_grid      # every matrix weight rounded to an f16 number (kept f32): what a published fp16 checkpoint holds


def build(sd, precision):
    d = config.get_dims("ViT-B/16")
    m = CLIP(d, dev, exact=precision, vision_only=precision == 2)
    load_openai_state_dict(m, {k: torch.from_numpy(v) for k, v in sd.items()})
    return m


def encode(m, pool, chunk):
    e = torch.empty(pool.shape[0], 512, device=dev)
    with torch.no_grad():
        m.visual.tower.encode_chunks(pool, e, 0, pool.shape[0], chunk, streams=1)
    torch.cuda.synchronize()
    return e


def dir_err(e, ref):
    d = (unit(e) - unit(ref)).norm(dim=-1)
    d = d[torch.isfinite(d)]
    return {"rms": float(d.pow(2).mean().sqrt()), "max": float(d.max()), "p99": float(d.quantile(0.99)), "finite_rows": int(d.numel())}


def logit_dev(e, ref, txt, scale=100.0):
    """Spread over the classes of the logit error, and the per-entry log-odds deviation, of e against ref for text features txt."""
    t = unit(txt)
    l, lr = scale * unit(e) @ t.T, scale * unit(ref) @ t.T
    ok = torch.isfinite(l).all(dim=1)
    dl = (l - lr)[ok]
    p, pr = torch.softmax(l[ok].double(), 1), torch.softmax(lr[ok].double(), 1)
    lo = (torch.log(p) - torch.log1p(-p)) - (torch.log(pr) - torch.log1p(-pr))
    lo = lo[torch.isfinite(lo)]
    rel = ((p - pr).abs() / torch.minimum(p, pr).clamp_min(1e-300))
    rel = rel[(torch.maximum(p, pr) > 1e-30)]
    spread = dl.max(1).values - dl.min(1).values
    return {"logit_err_rms": float(dl.pow(2).mean().sqrt()), "logit_err_absmax": float(dl.abs().max()), "spread_max": float(spread.max()), "spread_median": float(spread.median()),
            "log_odds_dev_max": float(lo.abs().max()), "log_odds_dev_p999": float(lo.abs().quantile(0.999)) if lo.numel() < 16_000_000 else None,
            "relative_dev_max": float(rel.max()), "mean_top1": float(pr.max(1).values.mean()), "distinct_argmax": int(pr.argmax(1).unique().numel())}


def text_sets(e32, tok_model, C):
    """Candidate text-feature sets for a pool with exact embeddings e32."""
    en = unit(e32)
    g = torch.Generator(device=dev).manual_seed(777)
    anchors = torch.randperm(en.shape[0], generator=g, device=dev)[:C]
    mean = en.mean(0, keepdim=True)
    out = {}
    if tok_model is not None:
        with torch.no_grad():
            out["zero_shot_text_tower"] = tok_model.encode_text(bench.synth_tokens(C, 0).to(dev)).float()
    out["prototypes_uncentred"] = en[anchors].clone()
    out["prototypes_mean_removed"] = en[anchors] - mean
    for gamma in (2.0, 4.0, 8.0, 16.0):
        out[f"prototypes_blend_gamma{gamma:g}"] = mean + gamma * (en[anchors] - mean)
    return out


# ---------------------------------------------------------------------------------------------------------------- torch emulation of the f16 tower
def emulate_vit(sd, x, stream_f32):
    """ViT-B/16 forward with the product's rounding points: f16 GEMM operands, f32 accumulation (an f32 matmul of f16-rounded operands), LayerNorm folded
    into the QKV / c_fc GEMMs (raw stream as the A operand, W' = f16(gamma * W)), f16 qkv / probabilities / attention output / MLP hidden, every
    residual add in f32 -- and the stream rounded to f16 after it (stream_f32 = False: the product) or kept in f32 (True)."""
    h16 = lambda t: t.half().float()
    P = lambda k: torch.from_numpy(sd["visual." + k]).to(dev)
    B = x.shape[0]
    w = h16(P("conv1.weight").reshape(768, -1))
    patches = torch.nn.functional.unfold(x, 16, stride=16).transpose(1, 2)          # [B, 196, 768]
    t = h16(patches) @ w.T
    s = torch.cat([P("class_embedding").expand(B, 1, 768), t], 1) + P("positional_embedding")
    mu, var = s.mean(-1, keepdim=True), s.var(-1, unbiased=False, keepdim=True)
    s = (s - mu) / torch.sqrt(var + 1e-5) * P("ln_pre.weight") + P("ln_pre.bias")
    rnd = (lambda t: t) if stream_f32 else h16
    s = rnd(s)
    for l in range(12):
        b = f"transformer.resblocks.{l}."

        def folded(s, ln, wk, bk):
            g, beta, Wm, bias = P(b + ln + ".weight"), P(b + ln + ".bias"), P(b + wk), P(b + bk)
            Wg = h16(Wm * g)
            mu, var = s.mean(-1, keepdim=True), s.var(-1, unbiased=False, keepdim=True)
            acc = h16(s) @ Wg.T
            return (acc - mu * Wg.sum(1)) / torch.sqrt(var + 1e-5) + (Wm @ beta + bias)
        qkv = h16(folded(s, "ln_1", "attn.in_proj_weight", "attn.in_proj_bias"))
        q, k, v = [t.view(B, -1, 12, 64).transpose(1, 2) for t in qkv.split(768, -1)]
        p = h16(torch.softmax(q @ k.transpose(-1, -2) / 8.0, -1))
        o = h16((p @ v).transpose(1, 2).reshape(B, -1, 768))
        s = rnd(s + o @ h16(P(b + "attn.out_proj.weight")).T + P(b + "attn.out_proj.bias"))
        hdn = folded(s, "ln_2", "mlp.c_fc.weight", "mlp.c_fc.bias")
        hdn = h16(hdn * torch.sigmoid(1.702 * hdn))
        s = rnd(s + hdn @ h16(P(b + "mlp.c_proj.weight")).T + P(b + "mlp.c_proj.bias"))
    c = s[:, 0]
    mu, var = c.mean(-1, keepdim=True), c.var(-1, unbiased=False, keepdim=True)
    c = (c - mu) / torch.sqrt(var + 1e-5) * P("ln_post.weight") + P("ln_post.bias")
    return h16(c) @ h16(P("proj"))


def main():
    d = config.get_dims("ViT-B/16")
    pools = {"noise": bench.synth_pool(a.rows, 224, dev, 1234), "structured": structured(a.rows, 4242)}
    dumps = {}
    for variant in ("standard", "stress"):
        sd = W.stress_state_dict(d, 0) if variant == "stress" else W.init_state_dict(d, 0)
        for grid in (False, True):
            sdv = on_grid(sd) if grid else sd
            t0 = time.time()
            m16, m32 = build(sdv, 0), build(sdv, 1)
            msp = build(sdv, 2)
            key = variant + ("+f16grid" if grid else "")
            report[key] = {}
            for pname, pool in pools.items():
                if variant == "stress" and pname == "noise":
                    continue
                e32, e16, esp = encode(m32, pool, 220), encode(m16, pool, 1320), encode(msp, pool, 440)
                r = {"f16_vs_f32": dir_err(e16, e32), "split_vs_f32": dir_err(esp, e32)}
                en = unit(e32)
                mean = en.mean(0, keepdim=True)
                r["geometry"] = {"norm_of_mean_direction": float(mean.norm()), "rms_distance_from_mean": float((en - mean).norm(dim=-1).pow(2).mean().sqrt()),
                                 "top_singular_values_of_centred": [float(v) for v in torch.linalg.svdvals((en - mean)[:4096])[:8]]}
                r["text_sets"] = {name: {"f16": logit_dev(e16, e32, t), "split": logit_dev(esp, e32, t)} for name, t in text_sets(e32, m32 if pname != "x" else None, a.classes).items()}
                report[key][pname] = r
                if not grid and pname == "structured":
                    dumps[f"{variant}_e32"] = e32.cpu().numpy()
                    dumps[f"{variant}_d16"] = (e16 - e32).cpu().numpy().astype(np.float16)
                print(key, pname, json.dumps(r["f16_vs_f32"]), f"{time.time() - t0:.0f}s", flush=True)
            if variant == "standard":
                # the emulation against the same f32 twin (first emu-rows of the structured pool), on the grid and off it
                x = pools["structured"][: a.emu_rows]
                ref = encode(m32, x, 128)
                real = encode(m16, x, a.emu_rows)
                with torch.no_grad():
                    emu = {}
                    for name, f32s in (("f16_stream", False), ("f32_stream", True)):
                        emu[name] = torch.cat([emulate_vit(sdv, x[i:i + 64], f32s) for i in range(0, a.emu_rows, 64)])
                txt = text_sets(ref, m32, a.classes)
                report[key]["emulation"] = {"real_f16_tower": dir_err(real, ref), "emulated_f16_stream": dir_err(emu["f16_stream"], ref), "emulated_f32_stream": dir_err(emu["f32_stream"], ref),
                                            "emulated_f16_vs_real_f16": dir_err(emu["f16_stream"], real),
                                            "logit_dev_zero_shot": {n: logit_dev(e, ref, txt["zero_shot_text_tower"]) for n, e in (("real", real), ("f16_stream", emu["f16_stream"]), ("f32_stream", emu["f32_stream"]))}}
                print(key, "emulation", json.dumps(report[key]["emulation"])[:600], flush=True)
            del m16, m32, msp
            torch.cuda.empty_cache()
    np.savez_compressed(a.out + "_emb.npz", **dumps)
    with open(a.out + ".json", "w") as f:
        json.dump(report, f, indent=1)
    print("wrote", a.out + ".json")


main()
