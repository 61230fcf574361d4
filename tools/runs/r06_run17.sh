#!/bin/bash
# split GEMM per shape (rocprofv3 kernel stats): how far the K = 768 shapes sit below the K = 3072 one (per-tile prologue / epilogue share)
mkdir -p gpurun_out/r06
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sg -o sg -- python $GRAFT_REPO_ROOT/tools/split_gemm_shapes.py > $GRAFT_REPO_ROOT/gpurun_out/r06/run17.log 2>&1
f=$(find /tmp/sg -name "*kernel_stats.csv" | head -1)
grep -i "split" "$f" >> $GRAFT_REPO_ROOT/gpurun_out/r06/run17.log
