#!/bin/bash
# per-kernel, per-grid-size times of the VPT step in situ (rocprofv3 --kernel-trace), previous commit's library vs this tree (+ env variants)
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
summ() { python3 - "$1" "$2" <<'PY'
import csv, sys, collections, glob
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if "gemm" in n or "ln_bwd_add" in n:
        k = (n.replace("void ", "").split("(")[0], r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", "?"), r.get("LDS_Block_Size", r.get("LDS_Block_Size_v", "?")))
        acc[k][0] += 1
        acc[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = 0
for k, (n, s) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print(sys.argv[2], k, n, round(s / n, 1), round(s / 30, 1))
    tot += s / 30
print(sys.argv[2], "GEMM total per step", round(tot, 1))
PY
}
go() { tag=$1; shift; rm -rf /tmp/vp_$tag; env "$@" timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/vp_$tag -o r -- python $R/tools/vpt_loop.py > /dev/null 2>&1; summ /tmp/vp_$tag $tag; }
{
go base X=1
go k64w GRIP_K64W=1
} > $R/gpurun_out/exp5.log 2>&1
grep "total\|k64" $R/gpurun_out/exp5.log
