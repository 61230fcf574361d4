cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for grp in "SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_DATA_FIFO_FULL SQ_INSTS_VALU"; do
  for m in 0 8; do
    d=/tmp/pmc_${m}_$(echo $grp | md5sum | cut -c1-6)
    GRIP_ATTN_PIPE=$m rocprofv3 --pmc $grp --output-format csv -d $d -- python $R/tools/attn_one.py 440 197 12 0 5 > /dev/null 2>&1
    f=$(find $d -name "*counter_collection.csv" | head -1)
    python3 - "$f" "$m" <<'PY'
import csv,sys,collections
f,m=sys.argv[1],sys.argv[2]
acc=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if 'attn_fwd' in r['Kernel_Name']:
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
print('pipe='+m, {k: round(sum(v)/len(v)) for k,v in acc.items()}, 'launches', {k:len(v) for k,v in acc.items()})
PY
  done
done
