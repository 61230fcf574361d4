mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/st_coopg
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_coopg -o r -- python $GRAFT_REPO_ROOT/tools/coop_graph_bench.py 40 > /dev/null 2>&1
cp $(find /tmp/st_coopg -name '*kernel_trace.csv' | head -1) /tmp/coopg_trace.csv
python3 - <<'PY'
import csv, os, collections
tr = list(csv.DictReader(open("/tmp/coopg_trace.csv")))
tr.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(tr) if "text_embed" in r["Kernel_Name"]]
a, b = idx[-2], idx[-1]
step = tr[a:b]
t0 = int(step[0]["Start_Timestamp"])
print("one replay: launches", len(step), "span us", (int(tr[b]["Start_Timestamp"]) - t0) / 1e3, "kernel us", sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in step) / 1e3)
agg = collections.OrderedDict()
prev = t0
gaps = 0
with open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/coop_graph_sequence.txt", "w") as f:
    for r in step:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        f.write(f"{(s - t0) / 1e3:9.1f} gap {(s - prev) / 1e3:6.1f} dur {(e - s) / 1e3:6.1f} {r['Kernel_Name'][:110]}\n")
        gaps += max(0, s - prev)
        k = r["Kernel_Name"][:70]
        c = agg.setdefault(k, [0, 0])
        c[0] += 1; c[1] += e - s
        prev = e
print("gaps us", gaps / 1e3)
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:24]:
    print(f"   {n:3d} x {t / n / 1e3:6.1f} us = {t / 1e3:7.1f}  {k}")
PY
