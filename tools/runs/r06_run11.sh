mkdir -p gpurun_out/r06
{
echo "# 96-row loader-wave tile also for the SHORT-walk M = 3408 GEMMs (out-proj forward, its dgrad): GRIP_GEMM_R96=1 (long walks only, default) vs 2"
for rep in 1 2; do
GRIP_GEMM_R96=1 python tools/step_bench.py vpt 2>/dev/null | tail -1
GRIP_GEMM_R96=2 python tools/step_bench.py vpt 2>/dev/null | tail -1
GRIP_GEMM_R96=1 python tools/step_bench.py upt 2>/dev/null | tail -1
GRIP_GEMM_R96=2 python tools/step_bench.py upt 2>/dev/null | tail -1
done
} > gpurun_out/r06/r96_short_ab.txt 2>&1
cat gpurun_out/r06/r96_short_ab.txt
GRIP_GEMM_R96=2 timeout 600 python -m pytest tests/test_gpu_backward.py tests/test_gpu_trajectory.py -q -k "vitb16 or ViT-B" 2>&1 | tail -3 | cut -c1-200
