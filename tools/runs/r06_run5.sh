set -x
mkdir -p gpurun_out/r06
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06/smoke_e.log 2>&1; echo "smoke rc $?"; tail -2 gpurun_out/r06/smoke_e.log | cut -c1-400
timeout 1500 python -m pytest tests -m gpu -q --durations=40 > gpurun_out/r06/gpu_tests_e.log 2>&1
echo "tests rc $?"; tail -60 gpurun_out/r06/gpu_tests_e.log | cut -c1-200
