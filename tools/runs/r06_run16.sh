#!/bin/bash
# last tree (4 x 2 split GEMM waves, balanced tier chunks): bench line, then smoke + whole GPU suite
mkdir -p gpurun_out/r06
timeout 900 python bench.py > gpurun_out/r06/run16_bench.json 2> gpurun_out/r06/run16_bench.err
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r06/run16_smoke.log 2>&1
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r06/run16_gputest.log 2>&1
tail -3 gpurun_out/r06/run16_gputest.log
