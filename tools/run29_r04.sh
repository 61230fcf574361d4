mkdir -p gpurun_out
timeout 900 python bench.py --steps 12 --warmup 1 --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value']), d['ms_per_step'], d['steps'], {k:round(v['max_s']/d['steps'],4) for k,v in d['stage_seconds_over_ranks'].items() if k!='allgather'}, d['exact'].get('timed_loop_lists_identical_to_exact'), d['identical']['audit_widened_the_bound'], d['identical']['audits'])
"
rocm-smi --showmeminfo vram 2>/dev/null | grep -i "used" | head -2
