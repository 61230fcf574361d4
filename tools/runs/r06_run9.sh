mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_backward.py tests/test_gpu_trajectory.py tests/test_gpu_strategies.py -q -k "96_row or row_rotation or gemm_epilogues or vitb16 or ViT-B or golden or trajectory or graph" 2>&1 | tail -6 | cut -c1-220
{
echo "# VPT / UPT prompt steps (tools/step_bench.py, eager, 20 steps each): 96-row loader-wave tile for the K = 4 d residual GEMM (default) vs GRIP_GEMM_R96=0"
for rep in 1 2; do
GRIP_GEMM_R96=0 python tools/step_bench.py vpt 2>/dev/null | tail -2
python tools/step_bench.py vpt 2>/dev/null | tail -2
GRIP_GEMM_R96=0 python tools/step_bench.py upt 2>/dev/null | tail -2
python tools/step_bench.py upt 2>/dev/null | tail -2
done
} > gpurun_out/r06/r96_ab.txt 2>&1
cat gpurun_out/r06/r96_ab.txt
