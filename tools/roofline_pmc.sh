#!/bin/bash
# usage: tools/roofline_pmc.sh <tag>   -- PMC passes (each its own run, --kernel-trace only beside --pmc) + a --stats run over
# tools/roofline_pmc.py; summaries into profiles/<tag>_*, traffic JSON into profiles/${tag}_pmc.json
tag=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/rpmc_$tag
rm -rf $out; mkdir -p $out profiles
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM"; do
  t=$(echo $pass | tr ' ' '_' | cut -c1-32)
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $out/$t -- python3 tools/roofline_pmc.py > $out/$t.log 2>&1 || { echo "pass $t failed"; tail -5 $out/$t.log; exit 1; }
done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 tools/roofline_pmc.py 20 10 10 > $out/stats.log 2>&1 || { tail -5 $out/stats.log; exit 1; }
python3 tools/pmc_summary.py $out > profiles/${tag}_roofline_pmc_summary.txt
cp $(find $out/stats -name "*kernel_stats.csv" | head -1) profiles/${tag}_roofline_kernel_stats.csv
python3 tools/make_pmc_json.py $out $tag
cat profiles/${tag}_pmc.json
