#!/bin/bash
# GPU session G: sweep of the workgroup-per-rotation kernels up to the dispatch limits, full GPU suite, netlist clocks
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
T=r02g
KERNELS="0 1 3" bash tools/sweep_kernels.sh 256 512 768 1024 1100 1536 > gpurun_out/${T}_sweep.txt 2>&1
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/${T}_pytest.txt
for net in cahp-ruby mux-ram cahp-system; do
  timeout 600 python tools/bench_netlist.py --net $net 2>/dev/null | tail -1 >> gpurun_out/${T}_netlist.txt
done
cat gpurun_out/${T}_sweep.txt gpurun_out/${T}_pytest.txt gpurun_out/${T}_netlist.txt
