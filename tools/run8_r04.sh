mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_split.py tests/test_gpu_trajectory.py -q -m gpu -s > gpurun_out/t_split.log 2>&1
grep -E "steps, loss|passed|failed|prefix=|three tiers" gpurun_out/t_split.log | cut -c1-260
python tools/split_rate.py 2640 880 2>&1 | grep -v amdgpu.ids | head -5
