#!/bin/bash
# Lean counter set of one blind-rotate launch: clock, issue / wait split, instruction mix.  One --pmc pass per group.
#   bash tools/pmc_lean.sh <tag>   (env: GATES, IYK_HIP_NTT / IYK_HIP_ROT_KERNEL)  -> gpurun_out/<tag>_pmc.txt
tag=${1:-pmc}
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
out=gpurun_out/${tag}_pmc.txt
: > $out
i=0
for grp in "GRBM_GUI_ACTIVE" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM"; do
    i=$((i+1))
    d=/tmp/pmcl_${tag}_$i
    rm -rf $d
    timeout 300 rocprofv3 --pmc $grp -d $d -o pmc -- python bench.py --gates ${GATES:-22528} --steps 1 --warmup 0 --cpu-sample 0 > /tmp/pmcl_$i.log 2>&1
    db=$(find $d -name "*.db" | head -1)
    if [ -n "$db" ]; then python tools/rocprof_summary.py $db --pmc | grep "blind_rotate" | cut -c1-120 >> $out; else echo "# group failed: $grp" >> $out; tail -3 /tmp/pmcl_$i.log >> $out; fi
done
cat $out
