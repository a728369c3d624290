#!/bin/bash
# same-box A/B of key-switch variants: iyokan_amd/lib/variant_<name>.so, names in $VARIANTS; both parameter sets
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
T=${1:-r03ks}
out=gpurun_out/${T}_ks_ab.txt
: > $out
cp iyokan_amd/lib/libiyokan_hip.so /tmp/keep.so
for rep in 1 2; do
for v in $VARIANTS; do
  cp iyokan_amd/lib/variant_$v.so iyokan_amd/lib/libiyokan_hip.so
  for ps in 128bit 80bit; do
  echo "$v $ps $(timeout 300 python bench.py --params $ps --steps 2 --warmup 1 --cpu-sample 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['roofline']['avg_launch_ms'],2), round(d['roofline']['keyswitch_avg_launch_ms'],2), d['config']['decrypt_check'])")" >> $out
  done
done
done
cp iyokan_amd/lib/variant_${LAST:-base}.so iyokan_amd/lib/libiyokan_hip.so
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_80bit.py -x -q -m gpu 2>&1 | tail -2 >> $out
cp /tmp/keep.so iyokan_amd/lib/libiyokan_hip.so
cat $out
