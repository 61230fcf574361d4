mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_gpu_identical.py tests/test_gpu_assign.py tests/test_gpu_split.py tests/test_gpu_dist.py tests/test_gpu_pseudolabel.py tests/test_gpu_strategies.py tests/test_gpu_determinism.py -q -m gpu -x > gpurun_out/t_sub.log 2>&1
grep -E "passed|failed|rror" gpurun_out/t_sub.log | tail -n 4
