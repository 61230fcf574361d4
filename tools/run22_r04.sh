mkdir -p gpurun_out
timeout 1200 python tools/coop_stress.py 4000 2>&1 | tail -n 8 | tee gpurun_out/coop_stress.txt
