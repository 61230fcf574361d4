#!/bin/bash
# Run on the GPU box: SQ wave-time counters of the VPT step's kernels (tools/vpt_loop.py, eager), three separate --pmc passes.
# Usage: bash tools/pmc_sq_vpt.sh <tag>  -> gpurun_out/sq_vpt_<tag>.csv
set -u
TAG=${1:-x}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/sq_vpt_$TAG.csv
cd /tmp && export TMPDIR=/tmp
echo "Kernel_Name,Counter_Name,Launches,Average_per_launch" > $OUT
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM"; do
    d=/tmp/sqv_$(echo $grp | md5sum | cut -c1-6)
    timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $d -o r -- python $R/tools/vpt_loop.py > /dev/null 2>&1
    python3 - "$(find $d -name '*counter_collection.csv' | head -1)" >> $OUT <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "attn_bwd" in n or "gemm_f16_kernel<0" in n or "ln_bwd_add" in n or "gemm_ring_kernel<3" in n:
        k = (n.replace("void ", "").split("(")[0], r["Counter_Name"])
        acc[k][0] += 1
        acc[k][1] += float(r["Counter_Value"])
for (k, c), (n, s) in sorted(acc.items()):
    print(f'"{k}",{c},{n},{s / n:.1f}')
PY
done
cat $OUT
