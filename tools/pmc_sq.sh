#!/bin/bash
# SQ / LDS counters of one small blind-rotate launch (8192 gates), one --pmc pass per counter group.
# Run on the GPU box:  bash tools/pmc_sq.sh <tag>   -> gpurun_out/<tag>_pmc_sq.txt
tag=${1:-pmc}
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
out=gpurun_out/${tag}_pmc_sq.txt
: > $out
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" \
           "SQ_INST_CYCLES_VMEM_RD SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_SMEM" \
           "SQ_WAIT_INST_VMEM SQ_WAIT_INST_SCA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU" \
           "SQ_INSTS SQ_INSTS_BRANCH SQ_INSTS_WAVE32 SQ_WAVES" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
           "TA_BUSY_avr TA_TA_BUSY_sum TD_TD_BUSY_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    d=/tmp/pmc_$tag_$i
    rm -rf $d
    timeout 300 rocprofv3 --pmc $grp -d $d -o pmc -- python bench.py --gates ${GATES:-8192} --steps 1 --warmup 0 --cpu-sample 0 > /tmp/pmc_$i.log 2>&1
    db=$(find $d -name "*.db" | head -1)
    if [ -n "$db" ]; then python tools/rocprof_summary.py $db --pmc | grep "blind_rotate" >> $out; else echo "# group failed: $grp" >> $out; tail -3 /tmp/pmc_$i.log >> $out; fi
done
cat $out
