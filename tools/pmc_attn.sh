cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for grp in "SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU" "SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU_TRANS_F32"; do
    d=/tmp/pmc_$(echo $grp | md5sum | cut -c1-6)
    rocprofv3 --pmc $grp --output-format csv -d $d -- python $R/tools/attn_one.py 440 197 12 0 5 > /dev/null 2>&1
    f=$(find $d -name "*counter_collection.csv" | head -1)
    python3 - "$f" <<'PY'
import csv,sys,collections
acc=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if 'attn_fwd' in r['Kernel_Name']:
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
print({k: round(sum(v)/len(v)) for k,v in acc.items()})
PY
done
