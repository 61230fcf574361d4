set -x
mkdir -p gpurun_out/r06
timeout 180 tools/micro/edge_cost 4000 > gpurun_out/r06/edge_cost.txt 2>&1; echo "edge rc $?"; cat gpurun_out/r06/edge_cost.txt
timeout 1200 python -m pytest tests/test_gpu_hilo.py tests/test_gpu_split.py tests/test_gpu_stress.py tests/test_gpu_dist.py::test_bench_two_ranks tests/test_gpu_determinism.py tests/test_gpu_identical.py -q --durations=15 > gpurun_out/r06/gpu_tests_c.log 2>&1
echo "tests rc $?"; tail -40 gpurun_out/r06/gpu_tests_c.log | cut -c1-250
python bench.py > gpurun_out/r06/bench_c.json 2> gpurun_out/r06/bench_c.err
echo "bench rc $?"
