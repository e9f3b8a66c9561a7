#!/bin/bash
# why does k_level_dec<8,9> run at ~0.7 waves per SIMD?  SPI resource-stall and instruction-cache counters, one pass each
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/ld_diag; rm -rf $out; mkdir -p $out
for pass in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVES" \
            "SPI_RA_REQ_NO_ALLOC_CSN SPI_RA_RES_STALL_CSN SPI_RA_LDS_CU_FULL_CSN SPI_RA_VGPR_SIMD_FULL_CSN SPI_RA_WAVE_SIMD_FULL_CSN SPI_RA_BAR_CU_FULL_CSN" \
            "SPI_RA_TGLIM_CU_FULL_CSN SPI_RA_WVLIM_STALL_CSN SPI_RA_TMP_STALL_CSN SPI_RA_SGPR_SIMD_FULL_CSN GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES" \
            "OccupancyPercent MeanOccupancyPerCU"; do
  t=$(echo $pass | tr ' ' '_' | cut -c1-24)
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $out/$t -- python3 bench.py --frames 66 --batch 65 --steps 1 --warmup 1 --no-cpu --no-extras --engines 1 > $out/$t.log 2>&1 || { echo "pass $t failed"; tail -3 $out/$t.log; }
done
python3 tools/pmc_summary.py $out | grep -A8 "==\|k_level_dec<8\|k_level_dec<4" | grep -v "^--" | cut -c1-120
