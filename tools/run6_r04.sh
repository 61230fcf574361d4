mkdir -p gpurun_out
timeout 900 python bench.py --steps 3 --warmup 1 --no-secondary --no-cpu-baseline > gpurun_out/bench_r04_2.json 2> gpurun_out/bench_r04_2.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_r04_2.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], {k:round(v['max_s'],3) for k,v in d['stage_seconds_over_ranks'].items()})
print({k:d['identical'][k] for k in ('rows_reencoded','rows_reencoded_split_f16','rows_reencoded_exactly','relative_bound','relative_bound_split_f16','audit_max_deviation','rounds','rows_per_round')}, d['exact']['timed_loop_lists_identical_to_exact'])
PY
timeout 3000 python -m pytest tests -x -q -m gpu > gpurun_out/t_all.log 2>&1
tail -n 8 gpurun_out/t_all.log
