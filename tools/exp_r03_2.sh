# A/B of the s_setprio experiment (persistent GEMM): same box, f16 bench loop, alternate builds through GRIP_LIB
cd /root/repo
B="python bench.py --mode f16 --no-exact --no-secondary --no-cpu-baseline --steps 2"
show() { python - "$1" <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])
g=d["roofline"]["all_gemm"]
def tf(k): 
    return next((v["tflops"] for n,v in g.items() if n.startswith(k)), None)
print(sys.argv[1].split("/")[-1], "value %.0f pl %.0f | resid %s qkv %s cfc %s clk %s" % (d["value"], d["pseudolabel_images_per_sec"], tf("gemm_k64p_kernel<9>"), tf("gemm_k64p_kernel<7>"), tf("gemm_k64p_kernel<8>"), d["roofline"]["clock_ghz_sustained"]))
PY
}
for rep in 1 2; do
$B > gpurun_out/sp0_$rep.json 2>/dev/null; show gpurun_out/sp0_$rep.json
for v in 1 2 3; do GRIP_LIB=/root/repo/menghini-neurips23-code_amd/libgrip_sp$v.so $B > gpurun_out/sp${v}_$rep.json 2>/dev/null; show gpurun_out/sp${v}_$rep.json; done
done
