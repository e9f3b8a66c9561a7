#!/bin/bash
# PMC passes (each its own run, --kernel-trace only beside --pmc) over tools/fi2_pmc.py
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/fi2pmc
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM" "SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS"; do
  tag=$(echo $pass | tr ' ' '_' | cut -c1-40)
  timeout -k 10 180 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d gpurun_out/fi2pmc/$tag -- python3 tools/fi2_pmc.py > gpurun_out/fi2pmc/$tag.log 2>&1 || { echo "pass $tag failed"; tail -5 gpurun_out/fi2pmc/$tag.log; }
done
python3 tools/pmc_summary.py gpurun_out/fi2pmc | grep -v "k_polyexp\|k_synth\|rocclr" > gpurun_out/fi2pmc_summary.txt
cat gpurun_out/fi2pmc_summary.txt
