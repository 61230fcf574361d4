mkdir -p gpurun_out
for S in 197 201 213 160; do python tools/attn_one.py 1320 $S 12 0 20 2>&1 | grep attention; done
python tools/attn_one.py 64 197 12 0 50 2>&1 | grep attention
GRIP_ATTN_PIPE=0 python tools/attn_one.py 1320 197 12 0 20 2>&1 | grep attention
timeout 1200 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "attention" > gpurun_out/t_att.log 2>&1; tail -n 3 gpurun_out/t_att.log
timeout 1200 python -m pytest tests/test_gpu_determinism.py tests/test_gpu_towers.py -q -m gpu > gpurun_out/t_det.log 2>&1; tail -n 3 gpurun_out/t_det.log
