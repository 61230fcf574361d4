#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd $R
python -m pytest tests/test_gpu_towers.py tests/test_gpu_backward.py tests/test_gpu_strategies.py tests/test_gpu_determinism.py tests/test_gpu_kernels.py -q -m gpu -x 2>&1 | grep -E "passed|failed|rror|FAILED|assert|ERROR" | tail -15 > $R/gpurun_out/exp8.log
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/up; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/up -o r -- python $R/tools/upt_loop.py > /tmp/upt.out 2>&1; tail -3 /tmp/upt.out >> $R/gpurun_out/exp8.log
python3 - /tmp/up >> $R/gpurun_out/exp8.log <<'PY'
import csv, sys, collections, glob
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    k = (n.replace("void ", "").split("(")[0][:60], r.get("Grid_Size_X", "?"), r.get("Workgroup_Size_X", "?"))
    acc[k][0] += 1
    acc[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = 0
for k, (n, s) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print(k, n, round(s / n, 1), round(s / 30, 1))
    tot += s / 30
print("total per step", round(tot, 1))
PY
head -70 $R/gpurun_out/exp8.log
