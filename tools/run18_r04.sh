mkdir -p gpurun_out
for cfg in "X=0" "GRIP_GEMM_COLGROUP=2" "GRIP_GEMM_COLGROUP=4" "X=1"; do
  env $cfg timeout 600 python bench.py --no-cpu-baseline --no-secondary --no-exact --steps 2 --warmup 1 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$cfg', round(d['value']), round(d['pseudolabel_images_per_sec']), {k:round(v['max_s'],3) for k,v in d['stage_seconds_over_ranks'].items() if k!='allgather'}, round(r['achieved']), r['clock_ghz_sustained'], round(r['clock_power']['power_w_mean']))
"
done | tee gpurun_out/colgroup_ab.txt
