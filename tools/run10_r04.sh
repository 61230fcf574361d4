mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_mixer.py tests/test_gpu_assign.py tests/test_gpu_split.py tests/test_gpu_trajectory.py -q -m gpu -s > gpurun_out/t_mix.log 2>&1
grep -E "passed|failed|^FAILED|rows, bounds|ViT-L" gpurun_out/t_mix.log | cut -c1-220
grep -E "^E  " gpurun_out/t_mix.log | head -10
timeout 600 python tools/upt_loop.py > /dev/null 2>&1; echo upt_loop rc $?
