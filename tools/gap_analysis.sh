cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/gt -o r -- python $GRAFT_REPO_ROOT/tools/pass_breakdown.py > /dev/null 2>&1
python3 - <<'PY'
import csv,glob,collections
f=glob.glob("/tmp/gt/**/*kernel_trace.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
ks=sorted(((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"]) for r in rows))
# take the last 40% of the trace (the final pass)
t0=ks[0][0]; t1=ks[-1][1]
cut=t0+int((t1-t0)*0.72)
ks=[k for k in ks if k[0]>=cut]
busy=sum(e-s for s,e,_ in ks); span=ks[-1][1]-ks[0][0]
gaps=collections.defaultdict(lambda:[0,0])
for (s0,e0,n0),(s1,e1,n1) in zip(ks,ks[1:]):
    g=s1-e0
    key=(n0.split("(")[0][-40:], n1.split("(")[0][-40:])
    gaps[key][0]+=1; gaps[key][1]+=max(g,0)
print("kernels",len(ks),"span ms",span/1e6,"busy ms",busy/1e6,"idle ms",(span-busy)/1e6)
for k,(n,t) in sorted(gaps.items(), key=lambda kv:-kv[1][1])[:14]:
    print(f"{k[0]:>40s} -> {k[1]:<40s} n={n:5d} total {t/1e6:7.2f} ms avg {t/n/1e3:6.1f} us")
PY
