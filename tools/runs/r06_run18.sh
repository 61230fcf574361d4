#!/bin/bash
# the driver's own bench invocation (BENCH_r05.json: --gpus 1 --steps 20 --warmup 5) on the last tree, wall-clocked
mkdir -p gpurun_out/r06
s=$(date +%s)
timeout 1200 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06/run18_bench.json 2> gpurun_out/r06/run18_bench.err
echo "rc=$? wall=$(( $(date +%s) - s )) s" > gpurun_out/r06/run18.log
