# same-box A/B: default build (GEMM s_setprio in sub-step 1 of the LN-folded instantiations + attention score-MFMA priority) vs the GEMM without it
cd /root/repo
B="python bench.py --mode f16 --no-exact --no-cpu-baseline --steps 2"
show() { python - "$1" <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])
g=d["roofline"]["all_gemm"]
def tf(k):
    return next((v["tflops"] for n,v in g.items() if n.startswith(k)), None)
s=d.get("secondary",{})
print(sys.argv[1].split("/")[-1], "value %.0f pl %.0f | resid %s qkv %s cfc %s | vitl %s vpt %s" % (d["value"], d["pseudolabel_images_per_sec"], tf("gemm_k64p_kernel<9>"), tf("gemm_k64p_kernel<7>"), tf("gemm_k64p_kernel<8>"), s.get("vitl14_336_encode",{}).get("images_per_sec"), s.get("vpt_step",{}).get("ms_hip_graph")))
PY
}
for rep in 1 2; do
GRIP_LIB=/root/repo/menghini-neurips23-code_amd/libgrip_sp0.so $B > gpurun_out/ab_off_$rep.json 2>/dev/null; show gpurun_out/ab_off_$rep.json
$B > gpurun_out/ab_on_$rep.json 2>/dev/null; show gpurun_out/ab_on_$rep.json
done
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_determinism.py tests/test_gpu_towers.py -x -q -m gpu 2>&1 | grep -E "passed|failed" 
