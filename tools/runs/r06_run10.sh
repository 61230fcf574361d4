set -x
mkdir -p gpurun_out/r06
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06/smoke_f.log 2>&1; echo "smoke rc $?"
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r06/gpu_tests_f.log 2>&1
echo "tests rc $?"; tail -3 gpurun_out/r06/gpu_tests_f.log | cut -c1-200
python bench.py > gpurun_out/r06/bench_final2.json 2> gpurun_out/r06/bench_final2.err; echo "bench rc $?"
