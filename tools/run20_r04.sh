mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_determinism.py tests/test_gpu_towers.py tests/test_gpu_identical.py -q -m gpu -x 2>&1 | tail -n 2
python tools/attn_one.py 1320 197 12 0 50
python tools/attn_one.py 1320 197 12 0 50
GRIP_ATTN_PIPE=0 python tools/attn_one.py 1320 197 12 0 50
python tools/attn_one.py 64 577 16 0 50
