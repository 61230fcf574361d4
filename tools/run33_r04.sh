mkdir -p gpurun_out
python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|rror|FAILED|ERROR" | tail -6
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -n 1
timeout 900 python bench.py > gpurun_out/bench_r04_final.json 2> gpurun_out/bench_r04_final.err
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_r04_final.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["train_images_per_sec"], d["secondary"]["vpt_step"]["ms_hip_graph"], d["secondary"]["upt_step"]["ms_hip_graph"], d["exact"].get("timed_loop_lists_identical_to_exact"))
PY
cd /tmp && export TMPDIR=/tmp
for step in vpt upt; do
  rm -rf /tmp/st_$step
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_$step -o r -- python $GRAFT_REPO_ROOT/tools/${step}_loop.py > /dev/null 2>&1
  cp $(find /tmp/st_$step -name '*kernel_stats.csv' | head -1) $GRAFT_REPO_ROOT/gpurun_out/${step}_step_kernel_stats_r04e.csv
done
