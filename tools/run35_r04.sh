mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_trainfold.py tests/test_gpu_determinism.py tests/test_gpu_towers.py tests/test_gpu_backward.py tests/test_gpu_identical.py tests/test_gpu_trajectory.py -q -m gpu -x > gpurun_out/t_sub.log 2>&1
grep -E "passed|failed|rror" gpurun_out/t_sub.log | tail -n 3
