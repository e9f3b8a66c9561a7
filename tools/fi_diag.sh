#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/fidiag; rm -rf $out; mkdir -p $out
i=0
for pass in "TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" "TCP_GATE_EN1_sum TCP_GATE_EN2_sum" "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCP_TA_ADDR_STALL_CYCLES_sum" "TCC_BUSY_avr TCC_TAG_STALL_sum" "TCC_REQ_sum TCC_READ_sum" "TD_TD_BUSY_sum TD_TC_STALL_sum" "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_LEVEL_sum" "SQ_BUSY_CU_CYCLES SQ_LEVEL_WAVES SQ_WAVES SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  timeout -k 10 120 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $out/p$i -- python3 tools/fi_diag.py > $out/p$i.log 2>&1 || { echo "pass $i ($pass) failed"; tail -3 $out/p$i.log; }
done
python3 tools/pmc_summary.py $out | grep -v "^==" | grep -A40 "k_flow_iter" | grep -v "k_polyexp\|synth" > gpurun_out/fidiag_summary.txt
cat gpurun_out/fidiag_summary.txt
