#!/bin/bash
# Same-box A/B of the key-switch kernels on wide batches: IYK_HIP_KS_KERNEL = 1 (a wave per 16 gates, compare-and-branch decode) against
# 2 (pre-added rows selected by address, kernels.hpp: keyswitch_lut_kernel).  Columns: gates/s, blind-rotate ms, key-switch ms,
# word_check (full batch only), decrypt_check.   bash tools/ab_ks_lut.sh [gates ...]   -> gpurun_out/r06b_ks_lut_ab.txt
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
out=gpurun_out/r06b_ks_lut_ab.txt; : > $out
for g in ${@:-65536}; do for rep in 1 2; do for k in 1 2; do for P in 128bit 80bit; do
echo "gates=$g ks=$k $P $(IYK_HIP_KS_KERNEL=$k timeout 300 python bench.py --params $P --gates $g --steps 3 --warmup 1 --cpu-sample 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value']), round(r['avg_launch_ms'],2), round(r['keyswitch_avg_launch_ms'],3), d['config'].get('word_check'), d['config']['decrypt_check'])")" >> $out
done; done; done; done
sort -s -k1,1 -k3,3 -k2,2 $out
