mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_gpu_trainfold.py tests/test_gpu_backward.py tests/test_gpu_trajectory.py -q -m gpu -x > gpurun_out/t_all.log 2>&1; tail -n 12 gpurun_out/t_all.log
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/st_coop
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_coop -o r -- python $GRAFT_REPO_ROOT/tools/coop_feature_loop.py 30 > /dev/null 2>&1
cp $(find /tmp/st_coop -name '*kernel_stats.csv' | head -1) $GRAFT_REPO_ROOT/gpurun_out/coop_step_kernel_stats_r04b.csv
cp $(find /tmp/st_coop -name '*kernel_trace.csv' | head -1) /tmp/coop_trace.csv
python3 - <<'PY'
import csv, os
rows = list(csv.DictReader(open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/coop_step_kernel_stats_r04b.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("coop kernel ms per step", tot / 30 / 1e6, "launches per step", sum(int(r["Calls"]) for r in rows) / 30)
for r in rows[:40]:
    print("   ", r["Name"][:90], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), r["Percentage"])
# one step's launch sequence (the last step) with durations and gaps
tr = list(csv.DictReader(open("/tmp/coop_trace.csv")))
tr.sort(key=lambda r: int(r["Start_Timestamp"]))
n = len(tr) // 30
last = tr[-n:]
t0 = int(last[0]["Start_Timestamp"])
print("last step: launches", n, "span us", (int(last[-1]["End_Timestamp"]) - t0) / 1e3)
prev = t0
with open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/coop_step_sequence_r04b.txt", "w") as f:
    for r in last:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        f.write(f"{(s - t0) / 1e3:9.1f} gap {(s - prev) / 1e3:6.1f} dur {(e - s) / 1e3:6.1f} grid {r.get('Grid_Size_X', r.get('Grid_Size', '?')):>8} wg {r.get('Workgroup_Size_X', r.get('Workgroup_Size', '?')):>5} {r['Kernel_Name'][:100]}\n")
        prev = e
PY
