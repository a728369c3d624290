#!/bin/bash
# Builds variants of the PRODUCT FFT kernel header that differ only in tuned constants (sed on a copy; the product file is untouched):
#   bash tools/ab_prio_sweep.sh build "name:s/old/new/;s/old2/new2/" ...   ->  iyokan_amd/lib/variant_<name>.so
# Each variant is an experiment build (-DIYK_EXPERIMENT_KERNELS_FFT: build id + "+x").  Run them with tools/ab_same_box.sh.
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
[ "$1" = build ] || { echo "usage: $0 build name:sedscript ..."; exit 1; }
shift
n=0
for spec in "$@"; do
  name=${spec%%:*}; script=${spec#*:}
  d=/tmp/iyk_variant_$name; mkdir -p $d
  sed -e "$script" iyokan_amd/csrc/kernels_fft.hpp > $d/kernels_fft_variant.hpp
  if cmp -s $d/kernels_fft_variant.hpp iyokan_amd/csrc/kernels_fft.hpp && [ "$script" != "" ]; then echo "variant $name: sed script changed nothing" >&2; fi
  X="-DIYK_EXPERIMENT_KERNELS_FFT=\"$d/kernels_fft_variant.hpp\""
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Iiyokan_amd/csrc "$X" \
    -DIYK_BUILD_ID="\"$(python3 tools/src_hash.py "-DVARIANT=$name:$script")\"" -o iyokan_amd/lib/variant_$name.so iyokan_amd/csrc/iyokan_hip.hip 2>/dev/null &
  n=$((n+1)); [ $((n % 4)) -eq 0 ] && wait
done
wait; ls -la iyokan_amd/lib/variant_*.so
