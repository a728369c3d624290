#!/bin/bash
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
T=${1:-r03e}
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_80bit.py -x -q -m gpu -k "rotation_kernels_agree or adversarial or 80bit_gates or two_levels or mid_size or dispatch_split or mux_batch" > gpurun_out/${T}_parity.txt 2>&1
tail -4 gpurun_out/${T}_parity.txt
KERNELS="lat3" bash tools/sweep_rot.sh 64 256 512 1024 1280 > gpurun_out/${T}_sweep_lat3.txt 2>&1
cat gpurun_out/${T}_sweep_lat3.txt
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DIYK_LAT3_TRACE=300 -o /tmp/lat3_trace tools/ubench/lat3_trace.hip > /tmp/tr.log 2>&1 && /tmp/lat3_trace > gpurun_out/${T}_lat3_trace.txt 2>&1
tail -30 gpurun_out/${T}_lat3_trace.txt
