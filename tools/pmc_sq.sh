#!/bin/bash
# Run on the GPU box: SQ wave-time counters of the pool-encode kernels (persistent GEMMs, attention) in three separate --pmc passes
# of a short bench (ten full chunks, eager launches).  Usage: bash tools/pmc_sq.sh <tag>  -> gpurun_out/sq_<tag>.csv
# Counter meaning (MI355X_MICROARCH.md): SQ_WAVE_CYCLES = SQ_WAIT_ANY (parked on s_waitcnt / barrier) + SQ_WAIT_INST_ANY (issue stalls)
# + SQ_ACTIVE_INST_ANY, all in quad-cycles summed over waves; SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = share of LDS cycles lost to conflicts.
set -u
TAG=${1:-x}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/sq_$TAG.csv
cd /tmp && export TMPDIR=/tmp
echo "Kernel_Name,Counter_Name,Launches,Average_per_launch" > $OUT
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA"; do
    d=/tmp/sq_$(echo $grp | md5sum | cut -c1-6)
    timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $d -o r -- python $R/bench.py --no-cpu-baseline --no-secondary --no-exact --graph 0 --lookahead 1 --pool 13200 --steps 1 --warmup 0 > /dev/null 2>&1
    python3 - "$(find $d -name '*counter_collection.csv' | head -1)" >> $OUT <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "gemm_k64p_kernel" in n or "attn_fwd_pipe" in n:
        k = (n.replace("void ", "").split("(")[0], r["Counter_Name"])
        acc[k][0] += 1
        acc[k][1] += float(r["Counter_Value"])
for (k, c), (n, s) in sorted(acc.items()):
    print(f'"{k}",{c},{n},{s / n:.1f}')
PY
done
cat $OUT
