#!/bin/bash
# Round 6 experiment (companion of the throughput kernel's start offset): the same one-off offset in the NARROW-frontier kernel, one
# rotation per CU, whose 1 .. 256 workgroups read the same key lines from L2 at the same moments.
#   bash tools/ab_stagger_lat.sh build ; on the GPU box: bash tools/ab_stagger_lat.sh run <tag>
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
gen=tools/experiments/_gen_kernels_fft_stagger_lat.hpp
python3 - "$gen" <<'P'
import sys
src = open("iyokan_amd/csrc/kernels_fft.hpp").read()
anchor = "    double worst = 0.0;\n    __syncthreads();\n"
assert src.count(anchor) == 1
block = '''#ifdef IYK_LAT_STAGGER_UNIT
    {
        const unsigned grp = IYK_LAT_STAGGER_BY == 0 ? blockIdx.x & 7u : IYK_LAT_STAGGER_BY == 1 ? (blockIdx.x >> 3) & 7u : (blockIdx.x * 5u) & 63u;
        for (unsigned k = 0; k < grp * IYK_LAT_STAGGER_UNIT; ++k) __builtin_amdgcn_s_sleep(4);   // ~256 cycles each
    }
#endif
'''
open(sys.argv[1], "w").write(src.replace(anchor, anchor + block))
P
build() {
  name=$1; shift
  X="-DIYK_EXPERIMENT_KERNELS_FFT=\"../../$gen\""
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Iiyokan_amd/csrc "$X" "$@" \
    -DIYK_BUILD_ID="\"$(python3 tools/src_hash.py "$X" "$@")\"" -o iyokan_amd/lib/variant_$name.so iyokan_amd/csrc/iyokan_hip.hip
}
if [ "$1" = build ]; then
  cp iyokan_amd/lib/libiyokan_hip.so iyokan_amd/lib/variant_base.so
  build lx4  -DIYK_LAT_STAGGER_BY=0 -DIYK_LAT_STAGGER_UNIT=4 &     # XCD k: k x 1 k cycles (a step is 9.2 k)
  build lc4  -DIYK_LAT_STAGGER_BY=1 -DIYK_LAT_STAGGER_UNIT=4 &     # CU position inside the XCD
  build lxc1 -DIYK_LAT_STAGGER_BY=2 -DIYK_LAT_STAGGER_UNIT=1 &     # 64 offsets of 256 cycles
  build lxc4 -DIYK_LAT_STAGGER_BY=2 -DIYK_LAT_STAGGER_UNIT=4 &     # 64 offsets of 1 k cycles (up to 7 steps)
  wait; ls iyokan_amd/lib/variant_l*.so; exit 0
fi
T=${2:-r06b_lat_stagger}
out=gpurun_out/${T}_ab.txt; : > $out
cp iyokan_amd/lib/libiyokan_hip.so /tmp/keep.so
ms() { timeout 200 python bench.py --gates $1 --steps 6 --warmup 2 --cpu-sample 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['roofline']['avg_launch_ms'],4), d['config']['decrypt_check'])"; }
for rep in 1 2; do for v in ${VARIANTS:-base lx4 lc4 lxc1 lxc4}; do
  cp iyokan_amd/lib/variant_$v.so iyokan_amd/lib/libiyokan_hip.so
  for g in ${WIDTHS:-16 64 256}; do echo "$v gates=$g $(ms $g)" >> $out; done
done; done
cp /tmp/keep.so iyokan_amd/lib/libiyokan_hip.so
sort -s -k2,2 -k1,1 $out
