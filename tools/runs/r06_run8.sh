mkdir -p gpurun_out/r06
timeout 600 python -m pytest tests/test_gpu_split.py tests/test_gpu_stress.py -q 2>&1 | tail -8 | cut -c1-250
{
echo "# split-f16 tower on fp16-checkpoint weights (two-pass GEMMs): W lo halves staged (GRIP_SPLIT_WDMA_FULL=1) vs skipped (default)"
for rep in 1 2; do
GRIP_SYNTHETIC_FP16=1 GRIP_SPLIT_WDMA_FULL=1 python tools/split_rate.py 2640 880 2>/dev/null | grep -E "^split|gemm_split|variant 7|k=7" | head -6
GRIP_SYNTHETIC_FP16=1 python tools/split_rate.py 2640 880 2>/dev/null | grep -E "^split|gemm_split|variant 7|k=7" | head -6
done
} > gpurun_out/r06/split_wdma_ab.txt 2>&1
cat gpurun_out/r06/split_wdma_ab.txt
GRIP_SYNTHETIC_FP16=1 python tools/split_rate.py 2640 880 2>&1 | tail -25 | cut -c1-200
