#!/bin/bash
# r03 round B: SQ / LDS counters of t16 and w32 on the same box, 22528 rotations (8 full t16 rounds = 11 full w32 rounds)
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
T=${1:-r03b}
for k in t16 w32; do
  IYK_HIP_TP_KERNEL=$k GATES=22528 bash tools/pmc_sq.sh ${T}_$k > /dev/null 2>&1
done
tail -n 40 gpurun_out/${T}_t16_pmc_sq.txt
