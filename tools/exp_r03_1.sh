cd /root/repo
B="python bench.py --mode f16 --no-exact --no-secondary --no-cpu-baseline --steps 2"
show() { python - "$1" <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])
g=d["roofline"]["all_gemm"]
def tf(k): 
    return next((v["tflops"] for n,v in g.items() if n.startswith(k)), None)
print(sys.argv[1].split("/")[-1], "value %.0f pl %.0f train %.0f | resid %s qkv %s cfc %s" % (d["value"], d["pseudolabel_images_per_sec"], d["train_images_per_sec"], tf("gemm_k64p_kernel<9>"), tf("gemm_k64p_kernel<7>"), tf("gemm_k64p_kernel<8>")))
PY
}
for la in 13 26 51 102; do $B --lookahead $la > gpurun_out/la_$la.json 2>/dev/null; show gpurun_out/la_$la.json; done
for cg in 2 3 4; do GRIP_GEMM_COLGROUP=$cg $B > gpurun_out/cg_$cg.json 2>/dev/null; show gpurun_out/cg_$cg.json; done
python tools/files_bench.py 12,14,16,20,24 512 2>/dev/null | cut -c1-400
python tools/files_bench.py 16 256,1024 2>/dev/null | cut -c1-400
