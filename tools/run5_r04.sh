mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_split.py -x -q -m gpu -s > gpurun_out/t_split.log 2>&1
grep -v "^$" gpurun_out/t_split.log | grep -v amdgpu.ids | tail -n 30
python tools/split_rate.py 2640 880 2>&1 | grep -v amdgpu.ids | head -12
