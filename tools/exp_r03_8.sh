#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd $R
{
echo "== cold, rotated, sweep"; ONLY=image COLD=16 ROT=1 SWEEP=1 python tools/small_gemm_bench.py 0 2>&1 | grep "^image"
echo "== cold, rotated, sweep (M = 3216: the UPT image side)"; IMAGE_M=3216 ONLY=image COLD=16 ROT=1 SWEEP=1 python tools/small_gemm_bench.py 0 2>&1 | grep "^image"
} > $R/gpurun_out/exp8.log 2>&1
cat $R/gpurun_out/exp8.log
