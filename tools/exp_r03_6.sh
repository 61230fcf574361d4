# same-box A/B: persistent GEMM with the DMA pieces of a SIMD's two waves issued by one of them (libgrip_asym.so, -DGRIP_ASYM=1) vs default
cd /root/repo
B="python bench.py --mode f16 --no-exact --no-cpu-baseline --no-secondary --steps 2"
show() { python - "$1" <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])
g=d["roofline"]["all_gemm"]
def tf(k):
    return next((v["tflops"] for n,v in g.items() if n.startswith(k)), None)
print(sys.argv[1].split("/")[-1], "value %.0f pl %.0f | resid %s qkv %s cfc %s" % (d["value"], d["pseudolabel_images_per_sec"], tf("gemm_k64p_kernel<9>"), tf("gemm_k64p_kernel<7>"), tf("gemm_k64p_kernel<8>")))
PY
}
GRIP_LIB=/root/repo/menghini-neurips23-code_amd/libgrip_asym.so python -m pytest tests/test_gpu_kernels.py tests/test_gpu_determinism.py tests/test_gpu_towers.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error" 
for rep in 1 2; do
$B > gpurun_out/asym_off_$rep.json 2>gpurun_out/asym.err; show gpurun_out/asym_off_$rep.json
GRIP_LIB=/root/repo/menghini-neurips23-code_amd/libgrip_asym.so $B > gpurun_out/asym_on_$rep.json 2>gpurun_out/asym.err; show gpurun_out/asym_on_$rep.json
done
