mkdir -p gpurun_out/r06
{
echo "# attention forward of the pool encode (attn_fwd_pipe_kernel, 1 320 images x 12 heads per launch) against the number of 16-row query tiles per (image, head):"
echo "# 16 waves = 4 per SIMD; 12 tiles = 3 per SIMD, the 13th lands on one SIMD as its 4th wave.  tools/attn_one.py, 20 launches each."
for S in 160 176 192 193 197 208 209 224; do python tools/attn_one.py 1320 $S 12 0 20 2>/dev/null | tail -1; done
echo "# S = 197 without the output stores (GRIP_ATTN_DBG=1):"
GRIP_ATTN_DBG=1 python tools/attn_one.py 1320 197 12 0 20 2>/dev/null | tail -1
echo "# S = 197 on the plain (non-pipelined, two co-resident workgroups) kernel (GRIP_ATTN_PIPE=0):"
GRIP_ATTN_PIPE=0 python tools/attn_one.py 1320 197 12 0 20 2>/dev/null | tail -1
} > gpurun_out/r06/attention_tiles.txt 2>&1
cat gpurun_out/r06/attention_tiles.txt
