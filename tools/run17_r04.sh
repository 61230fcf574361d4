mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_backward.py tests/test_gpu_trainfold.py tests/test_gpu_kernels.py -q -m gpu -x 2>&1 | tail -n 2
cd /tmp && export TMPDIR=/tmp
for step in vpt coop_feature; do
  rm -rf /tmp/st_$step
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_$step -o r -- python $GRAFT_REPO_ROOT/tools/${step}_loop.py > /dev/null 2>&1
  cp $(find /tmp/st_$step -name '*kernel_stats.csv' | head -1) $GRAFT_REPO_ROOT/gpurun_out/${step}_step_kernel_stats_r04c.csv
done
python3 - <<'PY'
import csv, os
for step in ("vpt", "coop_feature"):
    rows = list(csv.DictReader(open(os.environ["GRAFT_REPO_ROOT"] + f"/gpurun_out/{step}_step_kernel_stats_r04c.csv")))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print(step, "kernel ms per step", tot / 30 / 1e6, "launches per step", sum(int(r["Calls"]) for r in rows) / 30)
    for r in rows[:8]:
        print("   ", r["Name"][:80], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), r["Percentage"])
PY
