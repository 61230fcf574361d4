#!/bin/bash
# Run on the GPU box: builds tools/micro/fetch_calib.hip and reports FETCH_SIZE per launch against the known byte count.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/fetch_calib
mkdir -p $OUT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 $R/tools/micro/fetch_calib.hip -o /tmp/fetch_calib || exit 1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/prof_calib -o r -- /tmp/fetch_calib > $OUT/run.log 2>&1
python3 - "$(find /tmp/prof_calib -name '*counter_collection.csv' | head -1)" <<'PY' | tee $OUT/summary.txt
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] == "FETCH_SIZE":
        acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
known = 4096 * 256 * 768 * 2
for k, v in acc.items():
    avg = sum(v) / len(v)
    print(f"{k:28s} launches {len(v)}  FETCH_SIZE {avg:.0f} KiB/launch = {avg * 1024 / known:.3f} x the {known / 1e9:.2f} GB actually read")
PY
tail -1 $OUT/run.log
