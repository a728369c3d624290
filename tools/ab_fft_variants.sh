#!/bin/bash
# Same-box A/B of FFT-kernel variants.  Since round 6 the knobs live in tools/experiments/kernels_fft_r05_knobs.hpp (round 5's kernels with
# every alternative), NOT in the product header: a variant is that file + its -D flags, and its build id is the hash of sources + flags
# (tools/src_hash.py) with the suffix "+x" — never the product's.
# Build (here, no GPU):  bash tools/ab_fft_variants.sh build "name:-DFLAG ..." ...
#   -> iyokan_amd/lib/variant_<name>.so ; then on the GPU box:  VARIANTS="a b" bash tools/ab_fft_variants.sh run <tag>
# Variants whose name starts with t_ are TIMING-ONLY (wrong results by construction): the decrypt check is expected to fail.
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
if [ "$1" = build ]; then
  shift
  for spec in "$@"; do
    name=${spec%%:*}; flags=${spec#*:}
    X="-DIYK_EXPERIMENT_KERNELS_FFT=\"../../tools/experiments/kernels_fft_r05_knobs.hpp\""
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared "$X" $flags \
      -DIYK_BUILD_ID="\"$(python3 tools/src_hash.py "$X" $flags)\"" \
      -o iyokan_amd/lib/variant_$name.so iyokan_amd/csrc/iyokan_hip.hip &
  done
  wait; ls -la iyokan_amd/lib/variant_*.so; exit 0
fi
T=${2:-r04}
out=gpurun_out/${T}_fft_ab.txt
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
