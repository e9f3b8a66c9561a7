#!/bin/bash
# first GPU contact of the two-iteration kernel: parity tests, then timing, then the bench with and without it
set -e
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_flow.py -x -q -k "two_iteration or engine_with" > gpurun_out/fi2_tests.log 2>&1 || { tail -30 gpurun_out/fi2_tests.log; exit 1; }
tail -3 gpurun_out/fi2_tests.log
timeout -k 10 200 python tools/fi2_bench.py 2>&1 | tee gpurun_out/fi2_bench.log
OFC_FLOW_FUSE2=0 timeout -k 10 200 python bench.py --no-cpu --steps 5 > gpurun_out/bench_f0.log 2>&1; python tools/brief.py fuse2=0 < gpurun_out/bench_f0.log
OFC_FLOW_FUSE2=1 timeout -k 10 200 python bench.py --no-cpu --steps 5 > gpurun_out/bench_f1.log 2>&1; python tools/brief.py fuse2=1 < gpurun_out/bench_f1.log
OFC_FLOW_FUSE2=400 timeout -k 10 200 python bench.py --no-cpu --steps 5 > gpurun_out/bench_f400.log 2>&1; python tools/brief.py fuse2=400 < gpurun_out/bench_f400.log
