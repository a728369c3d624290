#!/bin/bash
# GPU session D/E: latency kernel 3 (v2: frequency-split MAC, 8 waves; v3: + pass split over SIMDs, key prefetch after barrier 2): trace, sweep, parity, netlist clocks
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
T=r02i
./tools/ubench/lat3_trace 64 > gpurun_out/${T}_lat3_trace.txt 2>&1
KERNELS="1 3" bash tools/sweep_kernels.sh 32 256 512 > gpurun_out/${T}_sweep.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_80bit.py -m gpu -x -q 2>&1 | tail -3 > gpurun_out/${T}_pytest.txt
IYK_HIP_LATENCY_KERNEL=3 GATES=256 bash tools/pmc_sq.sh ${T}_lat3 > /dev/null 2>&1
for net in cahp-ruby mux-ram cahp-system; do
  timeout 600 python tools/bench_netlist.py --net $net 2>/dev/null | tail -1 >> gpurun_out/${T}_netlist.txt
done
cat gpurun_out/${T}_lat3_trace.txt gpurun_out/${T}_sweep.txt gpurun_out/${T}_pytest.txt gpurun_out/${T}_netlist.txt
