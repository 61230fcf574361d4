mkdir -p gpurun_out
tools/micro/mfma_denorm > gpurun_out/mfma_denorm.txt 2>&1
timeout 1500 python -m pytest tests/test_gpu_assign.py -k small -x -q -m gpu > gpurun_out/t_assign.log 2>&1
timeout 1500 python -m pytest tests/test_gpu_identical.py tests/test_gpu_strategies.py tests/test_gpu_dist.py tests/test_gpu_pseudolabel.py -x -q -m gpu > gpurun_out/t_ident.log 2>&1
timeout 900 python bench.py --steps 3 --warmup 1 --no-secondary --no-cpu-baseline > gpurun_out/bench_r04_0.json 2> gpurun_out/bench_r04_0.err
tail -3 gpurun_out/t_assign.log gpurun_out/t_ident.log
cat gpurun_out/mfma_denorm.txt
