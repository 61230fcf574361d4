#!/bin/bash
# Final collection of a round (TAG=r05a): whole GPU suite, rocprofv3 kernel stats + PMC passes + the default bench line, in-situ kernel tables of the three
# prompt steps (eager loops under rocprofv3), the graphed CoOp step (replay time + per-kernel table of one replay), SQ counters of the VPT step's kernels
R=${GRAFT_REPO_ROOT:-$(pwd)}
T=${TAG:-r05a}
mkdir -p $R/gpurun_out
cd $R
python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|rror|FAILED|ERROR" | tail -8 > $R/gpurun_out/gputest_$T.log
bash tools/collect_profiles.sh $T > $R/gpurun_out/collect_$T.log 2>&1
cd /tmp && export TMPDIR=/tmp
for step in vpt upt coop_feature; do
  rm -rf /tmp/st_$step
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_$step -o r -- python $R/tools/${step}_loop.py > /dev/null 2>&1
  cp $(find /tmp/st_$step -name '*kernel_stats.csv' | head -1) $R/gpurun_out/${step}_step_kernel_stats_$T.csv
done
cd $R
python tools/coop_graph_bench.py 2>&1 | tail -n 1 > $R/gpurun_out/coop_graph_bench_$T.txt
bash tools/coop_graph_replay.sh > $R/gpurun_out/coop_graph_replay_$T.txt 2>&1
cp $R/gpurun_out/coop_graph_sequence.txt $R/gpurun_out/coop_graph_sequence_$T.txt 2>/dev/null
bash tools/pmc_sq_vpt.sh $T > /dev/null 2>&1
cat $R/gpurun_out/gputest_$T.log; tail -3 $R/gpurun_out/collect_$T.log; cat $R/gpurun_out/coop_graph_bench_$T.txt
