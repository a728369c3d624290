#!/bin/bash
# timing-only experiments on the w32 kernel (results are wrong by construction in the variants): where do its stalls come from?
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
T=${1:-r03g}
out=gpurun_out/${T}_w32_stall_exp.txt
: > $out
for v in base nokeys base nokeys; do
  cp iyokan_amd/lib/variant_$v.so iyokan_amd/lib/libiyokan_hip.so
  echo "$v $(timeout 300 python bench.py --steps 2 --warmup 1 --cpu-sample 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['roofline']['avg_launch_ms'],2), d['config']['decrypt_check'])")" >> $out
done
cp iyokan_amd/lib/variant_base.so iyokan_amd/lib/libiyokan_hip.so
cat $out
