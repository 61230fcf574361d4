mkdir -p gpurun_out/r06
{
echo "# after the split rule (gemm_pick_ksplit counts the 32-row tiles): default vs forced 3 / 8"
for rep in 1 2; do
python tools/coop_graph_bench.py 2>/dev/null | tail -1
GRIP_GEMM_KSPLIT=3 python tools/coop_graph_bench.py 2>/dev/null | tail -1
GRIP_GEMM_KSPLIT=8 python tools/coop_graph_bench.py 2>/dev/null | tail -1
done
python tools/step_bench.py upt 2>/dev/null | tail -1
python tools/step_bench.py vpt 2>/dev/null | tail -1
} > gpurun_out/r06/ksplit_ab2.txt 2>&1
cat gpurun_out/r06/ksplit_ab2.txt
timeout 600 python -m pytest tests/test_gpu_backward.py tests/test_gpu_trainfold.py tests/test_gpu_trajectory.py tests/test_gpu_kernels.py -q -k "text or coop or Text or splitk or 32_row" 2>&1 | tail -3 | cut -c1-200
