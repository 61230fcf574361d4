mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_gpu_trainfold.py tests/test_gpu_backward.py tests/test_gpu_kernels.py tests/test_gpu_trajectory.py tests/test_gpu_errors.py -q -m gpu -x > gpurun_out/t_all.log 2>&1; tail -n 4 gpurun_out/t_all.log
python tools/coop_graph_bench.py 2>&1 | tail -n 1
bash tools/run14_r04.sh 2>&1 | head -16
