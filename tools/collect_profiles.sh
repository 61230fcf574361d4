#!/bin/bash
# Run on the GPU box (gpurun): rocprofv3 kernel stats of the default bench + three separate PMC passes on a short run.
# Usage: bash tools/collect_profiles.sh <tag>   -> gpurun_out/prof_<tag>/
set -u
TAG=${1:-x}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
if [ "${SKIP_STATS:-0}" != "1" ]; then
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o r -- python $R/bench.py --no-cpu-baseline --no-secondary --no-exact > $OUT/bench_line.json 2> $OUT/bench_stderr.log
cp $(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
fi
# (the PMC passes time ONE pass over a fresh pool, which the auto policy screens with the compensated stream; the timed passes of the default bench screen the
# noise pool with the plain stream -- so the counters are collected with the plain stream unless the caller says otherwise: r06)
export GRIP_SCREEN_STREAM=${GRIP_SCREEN_STREAM:-f16}
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES"; do
  N=$(echo $C | cut -d' ' -f1)
  # (--lookahead 1: the prompt steps encode their own 16 images, so every persistent-GEMM launch in the pass is a full pool chunk and the
  # per-kernel averages are per-launch figures of the bench's dominant launch, not a mix with the 208-image look-ahead encodes.)
  # (--graph 0: rocprofv3 counter collection aborts on HIP-graph replays -- "incomplete dispatches" -- and then hangs; the encode
  # kernels the counters are wanted for are launched eagerly either way.  timeout: never let a profiler hang eat the box.)
  timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/prof_$N -o r -- python $R/bench.py --no-cpu-baseline --no-secondary --no-exact --graph 0 --lookahead 1 --pool 13200 --steps 1 --warmup 0 > /dev/null 2>> $OUT/bench_stderr.log
  python3 - "$(find /tmp/prof_$N -name '*counter_collection.csv' | head -1)" "$OUT/pmc_$N.csv" <<'PY'
import csv, sys, collections
src, dst = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(src)):
    k = (r["Kernel_Name"], r["Counter_Name"])
    acc[k][0] += 1
    acc[k][1] += float(r["Counter_Value"])
with open(dst, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Kernel_Name", "Counter_Name", "Launches", "Sum", "Average_per_launch"])
    for (k, c), (n, s) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        w.writerow([k, c, n, s, s / n])
PY
done
unset GRIP_SCREEN_STREAM
if [ "${FULL_LINE:-1}" = "1" ]; then timeout 900 python $R/bench.py > $OUT/bench_line_full.json 2>> $OUT/bench_stderr.log; fi
ls -la $OUT
