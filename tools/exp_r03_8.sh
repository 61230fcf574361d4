#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd $R
{
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_backward.py tests/test_gpu_towers.py tests/test_gpu_strategies.py -q -m gpu -x 2>&1 | grep -E "passed|failed|rror|FAILED|assert|ERROR" | tail -5
for rep in 1 2; do
echo "== prev"; GRIP_LIB=$R/menghini-neurips23-code_amd/libgrip_prev.so python tools/secondary_probe.py 1 2>&1 | grep vpt_step | cut -c1-420
echo "== new"; python tools/secondary_probe.py 1 2>&1 | grep vpt_step | cut -c1-420
done
} > $R/gpurun_out/exp8.log 2>&1
cat $R/gpurun_out/exp8.log
