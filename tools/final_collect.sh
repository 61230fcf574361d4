#!/bin/bash
# Final collection of a round (TAG=r03e): whole GPU suite, rocprofv3 kernel stats + PMC passes + the default bench line, in-situ kernel tables of the prompt steps
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd $R
python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|rror|FAILED|ERROR" | tail -8 > $R/gpurun_out/gputest_${TAG:-r03e}.log
bash tools/collect_profiles.sh ${TAG:-r03e} > $R/gpurun_out/collect_${TAG:-r03e}.log 2>&1
cd /tmp && export TMPDIR=/tmp
for step in vpt upt; do
  rm -rf /tmp/st_$step
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_$step -o r -- python $R/tools/${step}_loop.py > /dev/null 2>&1
  cp $(find /tmp/st_$step -name '*kernel_stats.csv' | head -1) $R/gpurun_out/${step}_step_kernel_stats_${TAG:-r03e}.csv
done
cat $R/gpurun_out/gputest_${TAG:-r03e}.log; tail -3 $R/gpurun_out/collect_${TAG:-r03e}.log
