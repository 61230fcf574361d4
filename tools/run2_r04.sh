mkdir -p gpurun_out
tools/micro/mfma_denorm > gpurun_out/mfma_denorm.txt 2>&1
timeout 1500 python -m pytest tests/test_gpu_split.py -x -q -m gpu -s > gpurun_out/t_split.log 2>&1
timeout 1500 python -m pytest tests/test_gpu_assign.py -q -m gpu -s > gpurun_out/t_assign.log 2>&1
tail -n 25 gpurun_out/t_split.log
tail -n 12 gpurun_out/t_assign.log
cat gpurun_out/mfma_denorm.txt
