mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_trainfold.py tests/test_gpu_determinism.py -q -m gpu -x > gpurun_out/t_sub.log 2>&1
grep -E "passed|failed|rror" gpurun_out/t_sub.log | tail -n 3
GRIP_TRAIN_FOLD=2 timeout 2400 python -m pytest tests/test_gpu_backward.py tests/test_gpu_towers.py tests/test_gpu_trajectory.py tests/test_gpu_strategies.py tests/test_gpu_mixer.py tests/test_gpu_dist.py -q -m gpu -x > gpurun_out/t_f2.log 2>&1
grep -E "passed|failed|rror" gpurun_out/t_f2.log | tail -n 3
