#!/bin/bash
# split GEMM wave layout A/B (4 x 2 "wide" vs 2 x 4), two-pass and three-pass forms
mkdir -p gpurun_out/r06
{
timeout 600 python -m pytest tests/test_gpu_split.py -x -q -m gpu 2>&1 | tail -3
for rep in 1 2; do
  for w in 1 0; do
    echo "== GRIP_SPLIT_WIDE=$w GRIP_SYNTHETIC_FP16=1"
    GRIP_SPLIT_WIDE=$w GRIP_SYNTHETIC_FP16=1 timeout 300 python tools/split_rate.py 2640 880 2>&1 | grep -A3 "^split"
  done
done
for w in 1 0; do
  echo "== GRIP_SPLIT_WIDE=$w (weights off the f16 grid: three-pass)"
  GRIP_SPLIT_WIDE=$w timeout 300 python tools/split_rate.py 2640 880 2>&1 | grep -A3 "^split"
done
} > gpurun_out/r06/run15.log 2>&1
