mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_split.py -q -m gpu -x > gpurun_out/t_split.log 2>&1
tail -n 3 gpurun_out/t_split.log
python tools/split_rate.py 2640 880 2>&1 | grep -v amdgpu.ids | head -5
