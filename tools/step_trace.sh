#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/step_trace
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/step_trace -- python3 bench.py --no-cpu --no-extras --steps 3 --engines ${1:-2} > gpurun_out/step_trace.log 2>&1
grep '^{' gpurun_out/step_trace.log | python3 tools/brief.py traced
python3 tools/trace_timeline.py gpurun_out/step_trace
cp $(find gpurun_out/step_trace -name "*kernel_stats.csv" | head -1) gpurun_out/step_kernel_stats.csv
