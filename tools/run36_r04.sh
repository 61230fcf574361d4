mkdir -p gpurun_out
for cfg in "X=0" "HIP_FORCE_DEV_KERNARG=1" "HIP_FORCE_DEV_KERNARG=0" "GPU_MAX_HW_QUEUES=2" "HSA_ENABLE_SDMA=0" "AMD_DIRECT_DISPATCH=0" "DEBUG_HIP_GRAPH_DOT_PRINT=0 HIP_GRAPH_SEGMENT_SCHEDULING=0"; do
  env $cfg python tools/coop_graph_bench.py 300 2>&1 | tail -n 1 | sed "s/^/[$cfg] /"
done | tee gpurun_out/launch_knobs.txt
