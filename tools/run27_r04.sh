mkdir -p gpurun_out
for cfg in "--lookahead 51" "--lookahead 102" "--lookahead 34" "--chunk 1980" "--chunk 2640"; do
  timeout 600 python bench.py --no-cpu-baseline --no-secondary --no-exact --steps 2 --warmup 1 $cfg 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$cfg', round(d['value']), {k:round(v['max_s'],3) for k,v in d['stage_seconds_over_ranks'].items() if k!='allgather'}, round(d['train_images_per_sec']))
"
done | tee gpurun_out/knobs_ab.txt
