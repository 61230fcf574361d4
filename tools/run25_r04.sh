mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_gpu_towers.py tests/test_gpu_exact.py tests/test_gpu_split.py tests/test_gpu_identical.py tests/test_gpu_assign.py tests/test_gpu_posemb.py tests/test_gpu_determinism.py -q -m gpu -x > gpurun_out/t_sub.log 2>&1
grep -E "passed|failed|rror|assert" gpurun_out/t_sub.log | tail -n 8
timeout 900 python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value']), d['ms_per_step'], {k:round(v['max_s'],3) for k,v in d['stage_seconds_over_ranks'].items() if k!='allgather'}, d['exact'].get('timed_loop_lists_identical_to_exact'), d['exact'].get('images_per_sec'))
"
