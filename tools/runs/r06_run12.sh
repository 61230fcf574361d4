mkdir -p gpurun_out/r06
{
echo "# graphed CoOp feature step (tools/coop_graph_bench.py) and eager UPT step: 32-row loader-wave tiles for the text tower's GEMMs whose 64-row tiles fill <= half the chip (GRIP_GEMM_R32=1) vs default"
for rep in 1 2 3; do
GRIP_GEMM_R32=0 python tools/coop_graph_bench.py 2>/dev/null | tail -1
GRIP_GEMM_R32=1 python tools/coop_graph_bench.py 2>/dev/null | tail -1
done
GRIP_GEMM_R32=0 python tools/step_bench.py upt 2>/dev/null | tail -1
GRIP_GEMM_R32=1 python tools/step_bench.py upt 2>/dev/null | tail -1
} > gpurun_out/r06/r32_ab.txt 2>&1
cat gpurun_out/r06/r32_ab.txt
GRIP_GEMM_R32=1 timeout 600 python -m pytest tests/test_gpu_backward.py tests/test_gpu_trainfold.py tests/test_gpu_trajectory.py -q -k "text or coop or Text" 2>&1 | tail -3 | cut -c1-200
