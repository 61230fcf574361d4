#!/bin/bash
# Run on the GPU box: shader clock / package power traces (tools/clock_trace.py, 20 Hz hwmon samples) of
#   (1) the pure-MFMA loop with zero and with random f16 operands (tools/micro/mfma_peak.hip),
#   (2) one f16 bench pass + CoOp steps (the kernels the roofline figure is about),
#   (3) the default (identical-mode) bench loop.
# Writes gpurun_out/clock/*.csv + summary.json; copy what is to be judged into profiles/.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/clock
mkdir -p $OUT
cd $R
python tools/clock_trace.py --probe > $OUT/probe.txt 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/mfma_peak.hip -o /tmp/mfma_peak || exit 1
python tools/clock_trace.py --out $OUT/mfma_peak.csv -- /tmp/mfma_peak 4 > $OUT/mfma_peak.json 2> $OUT/mfma_peak.err
python tools/clock_trace.py --out $OUT/bench_f16.csv -- python bench.py --mode f16 --no-exact --no-secondary --no-cpu-baseline --steps 3 > $OUT/bench_f16.json 2> $OUT/bench_f16.err
python tools/clock_trace.py --out $OUT/bench_identical.csv -- python bench.py --no-exact --no-secondary --no-cpu-baseline --steps 2 > $OUT/bench_identical.json 2> $OUT/bench_identical.err
tail -n 40 $OUT/probe.txt $OUT/mfma_peak.json
grep -h "TFLOP/s" $OUT/mfma_peak.json $OUT/mfma_peak.err 2>/dev/null
python - <<'PY'
import json, re, sys
for name in ("bench_f16", "bench_identical"):
    txt = open(f"gpurun_out/clock/{name}.json").read()
    try:
        tail = txt[txt.rindex('{\n "csv"'):]
        print(name, json.dumps(json.loads(tail)["by_label"]))
    except Exception as e:
        print(name, "no summary", e)
    m = re.search(r'^\{"metric".*$', txt, re.M)
    if m:
        d = json.loads(m.group(0))
        r = d["roofline"]
        print("  value", d["value"], "achieved", r["achieved"], "frac", r["frac"], "clock", r["clock_ghz_sustained"], "frac@clock", r["frac_at_sustained_clock"])
PY
