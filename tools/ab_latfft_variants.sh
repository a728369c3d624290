#!/bin/bash
# same-box A/B of narrow-frontier FFT-kernel variants (blind_rotate_fft_lat_kernel): iyokan_amd/lib/variant_<name>.so, names in $VARIANTS
#   VARIANTS="a b" bash tools/ab_latfft_variants.sh <tag>  ->  gpurun_out/<tag>_latfft_ab.txt ; the last variant then runs the parity tests
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
T=${1:-r05}
out=gpurun_out/${T}_latfft_ab.txt
: > $out
cp iyokan_amd/lib/libiyokan_hip.so /tmp/keep.so
ms() { IYK_HIP_ROT_KERNEL=latfft timeout 200 python bench.py --params $1 --gates $2 --steps 10 --warmup 2 --cpu-sample 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['roofline']['avg_launch_ms'],4), d['config']['decrypt_check'])"; }
last=""
for rep in 1 2; do
for v in $VARIANTS; do
  cp iyokan_amd/lib/variant_$v.so iyokan_amd/lib/libiyokan_hip.so
  for g in ${GATES:-16 64 256 512}; do echo "$v 128bit gates=$g rot_ms $(ms 128bit $g)" >> $out; done
  echo "$v 80bit gates=256 rot_ms $(ms 80bit 256)" >> $out
  last=$v
done
done
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_80bit.py -x -q -m gpu 2>&1 | tail -2 | sed "s/^/[$last] /" >> $out
cp /tmp/keep.so iyokan_amd/lib/libiyokan_hip.so
cat $out
