set -x
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_hilo.py tests/test_gpu_split.py tests/test_gpu_stress.py tests/test_gpu_dist.py::test_bench_two_ranks tests/test_gpu_determinism.py -x -q --durations=15 > gpurun_out/r06/gpu_tests_b.log 2>&1
echo "tests rc $?"; tail -25 gpurun_out/r06/gpu_tests_b.log
python bench.py > gpurun_out/r06/bench_b.json 2> gpurun_out/r06/bench_b.err
echo "bench rc $?"
GRIP_SCREEN_STREAM=f16 python bench.py --no-secondary --no-exact --no-cpu-baseline > gpurun_out/r06/bench_b_f16stream.json 2> gpurun_out/r06/bench_b_f16stream.err
GRIP_SPLIT_WLO=1 python bench.py --no-secondary --no-exact --no-cpu-baseline > gpurun_out/r06/bench_b_wlo.json 2> gpurun_out/r06/bench_b_wlo.err
python tools/gemm_bench.py 1320 > gpurun_out/r06/gemm_bench_1320.txt 2>&1
tail -30 gpurun_out/r06/gemm_bench_1320.txt
