mkdir -p gpurun_out
timeout 3000 python -m pytest tests -q -m gpu --deselect tests/test_gpu_kernels.py > gpurun_out/t_all.log 2>&1
tail -n 15 gpurun_out/t_all.log
timeout 1200 python -m pytest tests/test_gpu_trajectory.py -q -m gpu -s > gpurun_out/t_traj.log 2>&1
grep -v "^$" gpurun_out/t_traj.log | grep -v amdgpu | tail -n 15
