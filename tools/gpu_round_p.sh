#!/bin/bash
# A/B of whole-library build variants: iyokan_amd/lib/variant_*.so are swapped in for libiyokan_hip.so one at a time
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && export TMPDIR=/tmp
L=iyokan_amd/lib
cp $L/libiyokan_hip.so /tmp/base.so
run() {
  tag=$1
  timeout 600 python bench.py $BENCH_ARGS --cpu-sample 0 --steps 3 --warmup 1 > gpurun_out/r02p_bench_$tag.json 2> gpurun_out/r02p_bench_$tag.err
  tail -1 gpurun_out/r02p_bench_$tag.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$tag:', round(d['value'],1), round(d['ms_per_step'],2), 'br', round(d['roofline']['avg_launch_ms'],2), 'ks', round(d['roofline'].get('keyswitch_avg_launch_ms'),2))"
  if [ -n "$LAT" ]; then timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1; fi
}
run base
for v in $L/variant_*.so; do
  [ -f "$v" ] || continue
  cp $v $L/libiyokan_hip.so
  run $(basename $v .so)
done
cp /tmp/base.so $L/libiyokan_hip.so
if [ -n "$PYTEST" ]; then timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_80bit.py -m gpu -x -q 2>&1 | tail -3; fi
