set -x
mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests/test_gpu_hilo.py tests/test_gpu_stress.py tests/test_gpu_identical.py tests/test_gpu_split.py tests/test_gpu_dist.py tests/test_gpu_assign.py tests/test_gpu_strategies.py -q --durations=12 > gpurun_out/r06/gpu_tests_d.log 2>&1
echo "tests rc $?"; tail -30 gpurun_out/r06/gpu_tests_d.log | cut -c1-250
python bench.py > gpurun_out/r06/bench_d.json 2> gpurun_out/r06/bench_d.err
echo "bench rc $?"
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06/smoke_d.log 2>&1; echo "smoke rc $?"; tail -2 gpurun_out/r06/smoke_d.log
