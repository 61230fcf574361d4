mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
for lib in "" "$R/menghini-neurips23-code_amd/libgrip_amd_pf3.so" "$R/menghini-neurips23-code_amd/libgrip_amd_pf5.so" ""; do
  GRIP_LIB=$lib timeout 600 python bench.py --no-cpu-baseline --no-secondary --steps 2 --warmup 1 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('lib=$lib'.split('/')[-1], round(d['value']), round(d['pseudolabel_images_per_sec']), {k:round(v['max_s'],3) for k,v in d['stage_seconds_over_ranks'].items() if k!='allgather'}, round(r['achieved']), {k.split(' ')[0][-12:]:v['tflops'] for k,v in r['all_gemm'].items() if 'k64p' in k}, r['clock_ghz_sustained'], d['exact'].get('timed_loop_lists_identical_to_exact'))
"
done | tee gpurun_out/pf_ab.txt
