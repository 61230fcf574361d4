mkdir -p gpurun_out
timeout 900 python bench.py > gpurun_out/bench_r04_d_full.json 2> gpurun_out/bench_r04_d_full.err
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_r04_d_full.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["algorithmic_tflops"], d["executed_tflops"], d["exact"]["exact_images_per_sec"], d["exact"]["achieved_tflops"], d["exact"]["frac"], d["exact"]["timed_loop_lists_identical_to_exact"])
PY
