#!/bin/bash
# same-box A/B of the narrow-frontier kernel: r02's lat3 (inverse on 4 waves) vs r03's (inverse on 8 waves, keys fetched in the forward phase)
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
T=${1:-r03f}
out=gpurun_out/${T}_lat3_ab.txt
: > $out
for rep in 1 2; do
for v in r02lat3 new; do
  cp iyokan_amd/lib/variant_$v.so iyokan_amd/lib/libiyokan_hip.so
  echo "== $v (rep $rep)" >> $out
  KERNELS="lat3" bash tools/sweep_rot.sh 32 256 512 1024 1280 >> $out 2>&1
  if [ $rep = 1 ]; then for net in cahp-ruby cahp-system mux-ram; do timeout 600 python tools/bench_netlist.py --net $net 2>/dev/null | tail -1 | cut -c1-230 >> $out; done; fi
done
done
cp iyokan_amd/lib/variant_new.so iyokan_amd/lib/libiyokan_hip.so
cat $out
