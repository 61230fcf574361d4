mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
for m in 1 2; do
  rm -rf /tmp/st_vpt$m
  GRIP_TRAIN_FOLD=$m timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_vpt$m -o r -- python $GRAFT_REPO_ROOT/tools/vpt_loop.py > /dev/null 2>&1
  python3 - $m <<'PY'
import csv, sys, glob
m = sys.argv[1]
f = glob.glob(f"/tmp/st_vpt{m}/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("GRIP_TRAIN_FOLD=" + m, "kernel ms per step", round(tot / 30 / 1e6, 3), "launches", round(sum(int(r["Calls"]) for r in rows) / 30, 1))
for r in rows[:14]:
    print("   ", r["Name"][:84], round(int(r["Calls"]) / 30, 1), round(float(r["AverageNs"]) / 1e3, 1))
PY
done
