#!/bin/bash
set -e
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_flow.py -x -q -k "three_wave or two_iteration" > gpurun_out/w3_tests.log 2>&1 || { tail -30 gpurun_out/w3_tests.log; exit 1; }
tail -2 gpurun_out/w3_tests.log
timeout -k 10 300 python tools/fi2_bench.py 2>&1 | tee gpurun_out/fi2_bench.log
for v in 0 1 900 400; do
OFC_FLOW_W3=$v timeout -k 10 200 python bench.py --no-cpu --no-extras --steps 5 | python tools/brief.py "w3=$v"
done
