#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd $R
{
python -m pytest tests/test_gpu_strategies.py tests/test_gpu_mixer.py tests/test_gpu_backward.py -q -m gpu -x 2>&1 | grep -E "passed|failed|rror|FAILED|assert|ERROR" | tail -5
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python tools/secondary_probe.py 3 2>&1 | grep vpt_step | cut -c1-420
} > $R/gpurun_out/exp8.log 2>&1
cat $R/gpurun_out/exp8.log
