set -x
R=$(pwd)
mkdir -p gpurun_out/r06
python -m pytest tests/test_gpu_hilo.py -q -s 2>&1 | grep -E "direction error|screen stream|passed|failed" > gpurun_out/r06/hilo_numbers.txt; cat gpurun_out/r06/hilo_numbers.txt
GRIP_SCREEN_STREAM=f16 SKIP_STATS=1 FULL_LINE=0 bash tools/collect_profiles.sh r06f > gpurun_out/collect_r06f.log 2>&1; tail -3 gpurun_out/collect_r06f.log
cd $R
python tools/pmc_summary.py gpurun_out/prof_r06f r06f 06 1 > gpurun_out/r06/pmc_summary_f.txt 2>&1; tail -8 gpurun_out/r06/pmc_summary_f.txt
cp profiles/r06_traffic.json gpurun_out/r06/r06_traffic_f16.json
python bench.py > gpurun_out/r06/bench_final.json 2> gpurun_out/r06/bench_final.err; echo "bench rc $?"
