#!/bin/bash
# GPU session A of round 2: parity tests, kernel sweep incl. latency kernel 3, headline bench, netlist clocks.
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
T=r02a
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > gpurun_out/${T}_pytest.txt
KERNELS="0 1 3" bash tools/sweep_kernels.sh 32 256 512 1024 2048 > gpurun_out/${T}_sweep.txt 2>&1
timeout 600 python bench.py 2>gpurun_out/${T}_bench.err | tail -1 > gpurun_out/${T}_bench.json
for net in cahp-ruby mux-ram cahp-system; do
  for k in 1 3; do
    echo -n "lat_default=$k " >> gpurun_out/${T}_netlist.txt
    IYK_HIP_LATENCY_DEFAULT=$k timeout 600 python tools/bench_netlist.py --net $net 2>/dev/null | tail -1 >> gpurun_out/${T}_netlist.txt
  done
done
cat gpurun_out/${T}_pytest.txt gpurun_out/${T}_sweep.txt gpurun_out/${T}_netlist.txt
cut -c1-600 gpurun_out/${T}_bench.json
