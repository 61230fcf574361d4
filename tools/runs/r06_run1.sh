set -x
mkdir -p gpurun_out/r06
python tools/delta_probe.py --rows 8192 --emu-rows 512 --out gpurun_out/r06/delta_probe > gpurun_out/r06/delta_probe.log 2>&1
echo "probe rc $?"
python bench.py --steps 1 --warmup 1 > gpurun_out/r06/bench_a.json 2> gpurun_out/r06/bench_a.err
echo "bench rc $?"
timeout 1500 python -m pytest tests -m gpu -x -q --durations=60 > gpurun_out/r06/gpu_tests_a.log 2>&1
echo "tests rc $?"
tail -5 gpurun_out/r06/gpu_tests_a.log
