mkdir -p gpurun_out
timeout 1500 python bench.py > gpurun_out/bench_r04_c.json 2> gpurun_out/bench_r04_c.err; tail -n 3 gpurun_out/bench_r04_c.err
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_r04_c.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["config"].get("train_images_per_sec"), d["config"].get("pseudolabel_images_per_sec"))
s = d.get("secondary", d["config"].get("secondary", {}))
for k in ("coop_step", "vpt_step", "upt_step"):
    if k in s: print(k, s[k])
print({k: v for k, v in d["config"].items() if "per_sec" in k})
PY
