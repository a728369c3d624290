#!/bin/bash
# same-box A/B of narrow-frontier kernel variants: iyokan_amd/lib/variant_<name>.so, names in $VARIANTS
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
T=${1:-r03l}
out=gpurun_out/${T}_lat3_ab.txt
: > $out
cp iyokan_amd/lib/libiyokan_hip.so /tmp/keep.so
last=""
for rep in 1 2; do
for v in $VARIANTS; do
  cp iyokan_amd/lib/variant_$v.so iyokan_amd/lib/libiyokan_hip.so
  echo "== $v (rep $rep)" >> $out
  KERNELS="lat3" bash tools/sweep_rot.sh 32 256 1024 >> $out 2>&1
  last=$v
done
done
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_80bit.py -x -q -m gpu -k "rotation_kernels_agree or adversarial or 80bit_gates" 2>&1 | tail -2 | sed "s/^/[$last] /" >> $out
cp /tmp/keep.so iyokan_amd/lib/libiyokan_hip.so
cat $out
