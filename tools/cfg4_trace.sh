#!/bin/bash
# rocprofv3 kernel + memory-copy trace of the configs[4] streaming run: does the upload of batch b+1 overlap the flow of batch b?
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/cfg4_trace; rm -rf $out
timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $out -- python3 bench.py --workload cfg4 --steps 1 --warmup 1 --frames4k 96 > $out.log 2>&1 || { tail -5 $out.log; exit 1; }
grep '^{' $out.log | cut -c1-400
python3 tools/cfg4_overlap.py $out | tee gpurun_out/cfg4_overlap.txt
cp $(find $out -name "*kernel_stats.csv" | head -1) gpurun_out/cfg4_kernel_stats.csv
