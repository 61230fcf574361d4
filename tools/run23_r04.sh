mkdir -p gpurun_out
for cfg in "GRIP_TIER_STREAMS=1" "GRIP_TIER_STREAMS=2" "GRIP_TIER_STREAMS=2 EXCH=440" "GRIP_TIER_STREAMS=1 EXCH=440" "GRIP_TIER_STREAMS=2 EXCH=660"; do
  ex=880; for kv in $cfg; do case $kv in EXCH=*) ex=${kv#EXCH=};; esac; done
  env $cfg timeout 600 python bench.py --no-cpu-baseline --no-secondary --no-exact --steps 2 --warmup 1 --exact-chunk $ex 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$cfg', round(d['value']), {k:round(v['max_s'],3) for k,v in d['stage_seconds_over_ranks'].items() if k!='allgather'})
"
done | tee gpurun_out/tier_streams_ab.txt
