mkdir -p gpurun_out/r06
{
echo "# graphed CoOp feature step: split factor of the text tower's input-gradient GEMMs (default: gemm_pick_ksplit = 8 at K = 1536 / 2048) forced to 4 / 2 (32-row tiles then apply)"
for rep in 1 2; do
python tools/coop_graph_bench.py 2>/dev/null | tail -1
GRIP_GEMM_KSPLIT=4 python tools/coop_graph_bench.py 2>/dev/null | tail -1
GRIP_GEMM_KSPLIT=2 python tools/coop_graph_bench.py 2>/dev/null | tail -1
done
} > gpurun_out/r06/ksplit_ab.txt 2>&1
cat gpurun_out/r06/ksplit_ab.txt
