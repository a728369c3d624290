#!/bin/bash
# A/B of library variants on the narrow-frontier path: blind-rotate launch time at a few batch sizes (workgroup-per-rotation
# kernel 3) for libiyokan_hip.so and every iyokan_amd/lib/variant_*.so, then a netlist clock for each
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && export TMPDIR=/tmp
L=iyokan_amd/lib
cp $L/libiyokan_hip.so /tmp/base.so
run() {
  echo "== $1"
  KERNELS=3 bash tools/sweep_kernels.sh ${SIZES:-32 256 512}
  if [ -n "$NET" ]; then timeout 600 python tools/bench_netlist.py --net $NET --clocks 4 2>/dev/null | tail -1 | cut -c1-160; fi
}
run base
for v in $L/variant_*.so; do
  [ -f "$v" ] || continue
  cp $v $L/libiyokan_hip.so
  run $(basename $v .so)
done
cp /tmp/base.so $L/libiyokan_hip.so
