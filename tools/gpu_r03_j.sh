#!/bin/bash
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
T=${1:-r03j}
cp iyokan_amd/lib/variant_idx.so iyokan_amd/lib/libiyokan_hip.so
timeout 900 python -m pytest tests/test_gpu_80bit.py tests/test_gpu_parity.py -x -q -m gpu -k "80bit or rotation_kernels_agree or adversarial" > gpurun_out/${T}_parity.txt 2>&1
tail -3 gpurun_out/${T}_parity.txt
BENCH_ARGS="--params 80bit" VARIANTS="base idx" bash tools/gpu_r03_h.sh ${T}_80
VARIANTS="base idx" bash tools/gpu_r03_h.sh ${T}_128
