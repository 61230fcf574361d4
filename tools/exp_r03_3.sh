# A/B of s_setprio placements in the attention forward (GRIP_ATTN_PRIO builds through GRIP_LIB) + the new GEMM default
cd /root/repo
L=/root/repo/menghini-neurips23-code_amd
for rep in 1 2; do
for v in 0 1 2 4 5 7; do
  if [ $v = 0 ]; then unset GRIP_LIB; else export GRIP_LIB=$L/libgrip_ap$v.so; fi
  echo "prio $v: $(python tools/attn_one.py 1320 197 12 0 60 2>/dev/null) | $(python tools/attn_one.py 128 577 16 0 60 2>/dev/null) | $(python tools/attn_one.py 16 213 12 0 200 2>/dev/null)"
done
done
unset GRIP_LIB
python bench.py --mode f16 --no-exact --no-secondary --no-cpu-baseline --steps 2 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); g=d['roofline']['all_gemm']
print('default build: value %.0f pl %.0f' % (d['value'], d['pseudolabel_images_per_sec']), {k.split(' ')[0]: v['tflops'] for k, v in g.items() if 'k64p' in k})"
