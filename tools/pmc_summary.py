"""Fold the per-kernel PMC CSVs written by tools/collect_profiles.sh into profiles/r<NN>_traffic.json (what bench.py
reports as roofline.traffic / roofline.mfma_util).  Usage: python tools/pmc_summary.py gpurun_out/prof_<tag> <tag> [round=03] [traffic=1]"""
import csv
import json
import os
import shutil
import sys

src, tag = sys.argv[1], sys.argv[2]
RND = sys.argv[3] if len(sys.argv) > 3 else "03"
WRITE_TRAFFIC = (sys.argv[4] if len(sys.argv) > 4 else "1") == "1"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_XCD, N_SIMD = 8, 1024


def short(name):
    name = name.replace("void ", "")
    return name.split("(")[0]


per = {}
for f in ("pmc_FETCH_SIZE.csv", "pmc_WRITE_SIZE.csv", "pmc_SQ_VALU_MFMA_BUSY_CYCLES.csv"):
    for r in csv.DictReader(open(os.path.join(src, f))):
        per.setdefault(short(r["Kernel_Name"]), {})[r["Counter_Name"]] = (int(r["Launches"]), float(r["Average_per_launch"]))
kernels = {}
for k, c in per.items():
    if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
        continue
    fetch = 2.0 * c["FETCH_SIZE"][1] * 1024.0          # gfx950: FETCH_SIZE tallies 128-B requests at 64 B (MI355X_MICROARCH.md, HBM)
    write = c["WRITE_SIZE"][1] * 1024.0
    e = {"launches": c["FETCH_SIZE"][0], "fetch_bytes": fetch, "write_bytes": write, "bytes_per_launch": fetch + write}
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c and c["GRBM_GUI_ACTIVE"][1] > 0:
        cycles = c["GRBM_GUI_ACTIVE"][1] / N_XCD       # the counter is summed over the 8 XCDs
        e["gpu_cycles_per_launch"] = cycles
        e["mfma_util"] = c["SQ_VALU_MFMA_BUSY_CYCLES"][1] / (cycles * N_SIMD)
    kernels[k] = e
out = {
    "_how": "tools/collect_profiles.sh: three separate rocprofv3 --kernel-trace --pmc passes (FETCH_SIZE | WRITE_SIZE | "
            "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES) of `python bench.py --no-cpu-baseline --no-secondary --no-exact --graph 0 --lookahead 1 --pool 13200 --steps 1 "
            "--warmup 0` (default encode chunk 1320: ten full chunks; default identical mode: the exact re-encodes of the marked rows are in the pass too); per-kernel averages over launches. FETCH_SIZE (KiB) is doubled per "
            "MI355X_MICROARCH.md (gfx950 tallies a wide coalesced read stream at half its bytes); mfma_util = "
            "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs), i.e. against the ACTUAL shader clock "
            "(~2.05 GHz under this load, not the 2.4 GHz behind the 2.5 PF/s peak)",
    "source": f"profiles/r{RND}_{tag}_pmc_*.csv",
    "kernels": kernels,
}
if WRITE_TRAFFIC:
    json.dump(out, open(os.path.join(REPO, "profiles", f"r{RND}_traffic.json"), "w"), indent=1)
else:
    json.dump(out, open(os.path.join(REPO, "profiles", f"r{RND}_{tag}_traffic.json"), "w"), indent=1)
for f, d in (("kernel_stats.csv", "bench_kernel_stats.csv"), ("bench_line.json", "bench_line.json"),
             ("bench_line_full.json", "bench_line_full.json"), ("pmc_FETCH_SIZE.csv", "pmc_FETCH_SIZE.csv"),
             ("pmc_WRITE_SIZE.csv", "pmc_WRITE_SIZE.csv"), ("pmc_SQ_VALU_MFMA_BUSY_CYCLES.csv", "pmc_MFMA_BUSY.csv")):
    if os.path.exists(os.path.join(src, f)):
        shutil.copy(os.path.join(src, f), os.path.join(REPO, "profiles", f"r{RND}_{tag}_{d}"))
for k in sorted(kernels, key=lambda k: -kernels[k]["bytes_per_launch"] * kernels[k]["launches"])[:8]:
    e = kernels[k]
    print(f"{k[:60]:60s} {e['bytes_per_launch'] / 1e6:8.1f} MB/launch  mfma_util {e.get('mfma_util', 0):.3f}")
