#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/shard_trace
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/shard_trace -- python3 bench.py --no-cpu --no-extras --frames 39 --steps 10 --warmup 2 --batch ${1:-19} --engines ${2:-2} > gpurun_out/shard_trace.log 2>&1
tail -2 gpurun_out/shard_trace.log | cut -c1-300
python3 tools/trace_timeline.py gpurun_out/shard_trace
