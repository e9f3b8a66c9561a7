#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/fi2pmc
for pass in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=nt_$(echo $pass | tr ' ' '_' | cut -c1-40)
  timeout -k 10 180 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d gpurun_out/fi2pmc/$tag -- python3 tools/fi2_pmc.py > gpurun_out/fi2pmc/$tag.log 2>&1 || { echo "pass $tag failed"; tail -5 gpurun_out/fi2pmc/$tag.log; }
done
python3 tools/pmc_summary.py gpurun_out/fi2pmc | grep -A3 "nt_\|k_flow_iter" | grep -v "^--"
timeout -k 10 200 python tools/fi2_bench.py 2>&1 | head -3
