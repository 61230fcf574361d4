#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|rror|FAILED" | tail -15 > gpurun_out/exp7.log 2>&1
cat gpurun_out/exp7.log
