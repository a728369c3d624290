#!/bin/bash
# Round 6 experiment: a ONE-OFF start delay per workgroup of the throughput kernel, so that the 256 CUs do not run the same phase
# of a CMUX step at the same moment (all of them start together and execute identical instruction streams: their FMA-dense MAC
# phases coincide).  Question: does the clock the chip sustains under its power limit depend on that?  The experiment header is the
# PRODUCT header + the block below, generated at build time into tools/experiments/_gen_kernels_fft_stagger.hpp (git-ignored).
#   bash tools/ab_stagger.sh build        (here)      VARIANTS="base sx sc sxc" bash tools/ab_same_box.sh r06b_stagger   (GPU box)
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
gen=tools/experiments/_gen_kernels_fft_stagger.hpp
python3 - "$gen" <<'P'
import sys
src = open("iyokan_amd/csrc/kernels_fft.hpp").read()
anchor = "    const fft::Keys keys(bk_fft, bk_bytes, lane0);\n    double worst = 0.0;\n"
assert src.count(anchor) == 1
block = '''#ifdef IYK_FFT_STAGGER_UNIT
    if (blockIdx.x < 256u) {   // first resident generation only: later workgroups inherit the offset of the CU they land on
        const unsigned grp = IYK_FFT_STAGGER_BY == 0 ? blockIdx.x & 7u : IYK_FFT_STAGGER_BY == 1 ? (blockIdx.x >> 3) & 7u : (blockIdx.x * 5u) & 63u;
        for (unsigned k = 0; k < grp * IYK_FFT_STAGGER_UNIT; ++k) __builtin_amdgcn_s_sleep(16);   // ~1 k cycles each
    }
#endif
'''
open(sys.argv[1], "w").write(src.replace(anchor, block + anchor))
P
build() {  # name flags...
  name=$1; shift
  X="-DIYK_EXPERIMENT_KERNELS_FFT=\"../../$gen\""
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Iiyokan_amd/csrc "$X" "$@" \
    -DIYK_BUILD_ID="\"$(python3 tools/src_hash.py "$X" "$@")\"" -o iyokan_amd/lib/variant_$name.so iyokan_amd/csrc/iyokan_hip.hip
}
if [ "$1" = build ]; then
  cp iyokan_amd/lib/libiyokan_hip.so iyokan_amd/lib/variant_base.so
  # a step of one wave is ~44 k cycles; units of ~1 k cycles per group index
  build sx  -DIYK_FFT_STAGGER_BY=0 -DIYK_FFT_STAGGER_UNIT=5 &    # by XCD (blockIdx % 8): eight offsets over one step
  build sc  -DIYK_FFT_STAGGER_BY=1 -DIYK_FFT_STAGGER_UNIT=5 &    # by CU position inside the XCD
  build sxc -DIYK_FFT_STAGGER_BY=2 -DIYK_FFT_STAGGER_UNIT=1 &    # 64 offsets, both
  wait; ls -la iyokan_amd/lib/variant_*.so
fi
if [ "$1" = build2 ]; then   # wider offsets: CUs several steps (key rows) apart
  build sxc4  -DIYK_FFT_STAGGER_BY=2 -DIYK_FFT_STAGGER_UNIT=4 &
  build sxc16 -DIYK_FFT_STAGGER_BY=2 -DIYK_FFT_STAGGER_UNIT=16 &
  build sx40  -DIYK_FFT_STAGGER_BY=0 -DIYK_FFT_STAGGER_UNIT=40 &
  wait; ls -la iyokan_amd/lib/variant_*.so
fi
