#!/bin/bash
# same-box A/B of w32 variants: iyokan_amd/lib/variant_<name>.so, names in $VARIANTS
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
T=${1:-r03h}
out=gpurun_out/${T}_w32_ab.txt
: > $out
cp iyokan_amd/lib/libiyokan_hip.so /tmp/keep.so
for rep in 1 2; do
for v in $VARIANTS; do
  cp iyokan_amd/lib/variant_$v.so iyokan_amd/lib/libiyokan_hip.so
  echo "$v $(timeout 300 python bench.py --steps 2 --warmup 1 --cpu-sample 0 ${BENCH_ARGS} 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['roofline']['avg_launch_ms'],2), d['config']['decrypt_check'])")" >> $out
done
done
cp /tmp/keep.so iyokan_amd/lib/libiyokan_hip.so
cat $out
