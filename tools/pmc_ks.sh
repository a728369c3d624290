#!/bin/bash
# Counters of the wide-batch key-switch kernel (one 65 536-gate launch per --pmc pass).  bash tools/pmc_ks.sh <tag>  -> gpurun_out/<tag>_ks_pmc.txt
tag=${1:-r06b}; cd "$(dirname "$0")/.." && export TMPDIR=/tmp
out=gpurun_out/${tag}_ks_pmc.txt; : > $out
for grp in "GRBM_GUI_ACTIVE" "SQ_INSTS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAVES" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" "FETCH_SIZE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  d=/tmp/pmc_ks; rm -rf $d
  timeout 300 rocprofv3 --pmc $grp -d $d -o pmc -- python bench.py --params ${PARAMS:-128bit} --steps 1 --warmup 0 --cpu-sample 0 > /tmp/pmc_ks.log 2>&1
  db=$(find $d -name "*.db" | head -1)
  [ -n "$db" ] && python tools/rocprof_summary.py $db --pmc | grep "keyswitch_lut" | cut -c1-110 >> $out
done
cat $out
