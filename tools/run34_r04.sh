mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
for lib in "" "$R/menghini-neurips23-code_amd/libgrip_amd_r04d.so" "" "$R/menghini-neurips23-code_amd/libgrip_amd_r04d.so"; do
  GRIP_LIB=$lib timeout 600 python bench.py --no-cpu-baseline --no-secondary --no-exact --steps 2 --warmup 1 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('lib=$lib'.split('/')[-1], round(d['value']), {k:round(v['max_s'],3) for k,v in d['stage_seconds_over_ranks'].items() if k!='allgather'}, {k.split(' ')[0][-12:]:v['tflops'] for k,v in r['all_gemm'].items() if 'k64p' in k}, r['clock_ghz_sustained'])
"
done | tee gpurun_out/lib_ab.txt
