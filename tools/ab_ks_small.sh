#!/bin/bash
# Narrow-frontier key switch, same box: the round-4 form (every wave its own gates: IYK_HIP_KS_SHARED_MAX=0) against the shared-gates
# form (kernels.hpp keyswitch_wave_kernel<.., SHARED>) at several slicings (IYK_HIP_KS_SHARED_WG = workgroups a launch is cut into).
#   bash tools/ab_ks_small.sh <tag>   ->  gpurun_out/<tag>_ks_small_ab.txt   (bench.py --gates G: keyswitch_avg_launch_ms, decrypt check)
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
T=${1:-r05}
out=gpurun_out/${T}_ks_small_ab.txt
: > $out
one() {  # label, params, gates
  echo "$1 $2 gates=$3 $(timeout 300 python bench.py --params $2 --gates $3 --steps 20 --warmup 3 --cpu-sample 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ks_ms', round(d['roofline']['keyswitch_avg_launch_ms'],4), 'rot_ms', round(d['roofline']['avg_launch_ms'],4), 'ok', d['config']['decrypt_check'])")" >> $out
}
for rep in 1 2; do
for g in ${GATES:-8 16 64 256 1024 4096}; do
  IYK_HIP_KS_SHARED_MAX=0 one r04form 128bit $g
  for wg in ${WGS:-128 256 512 1024}; do
    IYK_HIP_KS_SHARED_WG=$wg one shared_wg$wg 128bit $g
  done
done
done
for g in 16 64 256 1024; do
  IYK_HIP_KS_SHARED_MAX=0 one r04form 80bit $g
  one shared_default 80bit $g
done
cat $out
