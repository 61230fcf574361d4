mkdir -p gpurun_out
python tools/split_rate.py 2640 880 > gpurun_out/split_rate.txt 2>&1
python tools/split_rate.py 880 220 >> gpurun_out/split_rate.txt 2>&1
timeout 900 python bench.py --steps 3 --warmup 1 --no-secondary --no-cpu-baseline > gpurun_out/bench_r04_1.json 2> gpurun_out/bench_r04_1.err
grep -v "^$" gpurun_out/split_rate.txt | grep -v amdgpu.ids | tail -n 60
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_r04_1.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], {k:v['max_s'] for k,v in d['stage_seconds_over_ranks'].items()})
print({k:d['identical'][k] for k in ('rows_reencoded','rows_reencoded_split_f16','rows_reencoded_exactly','relative_bound','relative_bound_split_f16','audit_max_deviation','rounds','rows_per_round')}, d['exact']['timed_loop_lists_identical_to_exact'])
PY
