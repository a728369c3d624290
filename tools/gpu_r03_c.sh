#!/bin/bash
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
T=${1:-r03c}
for k in t16 w32; do
  IYK_HIP_TP_KERNEL=$k bash tools/pmc_lean.sh ${T}_$k > /dev/null 2>&1
done
