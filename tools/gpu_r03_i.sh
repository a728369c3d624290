#!/bin/bash
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
T=${1:-r03i}
out=gpurun_out/${T}_80bit_ab.txt
: > $out
for rep in 1 2; do for k in w32 t16; do
  echo "$k $(IYK_HIP_TP_KERNEL=$k timeout 300 python bench.py --params 80bit --steps 2 --warmup 1 --cpu-sample 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['roofline']['avg_launch_ms'],2), round(d['roofline']['keyswitch_avg_launch_ms'],2), d['config']['decrypt_check'])")" >> $out
done; done
cat $out
