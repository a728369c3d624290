#!/bin/bash
# GPU session R: refresh of the secondary numbers with the final r02 kernels — 80-bit bench line, three netlist clock latencies,
# C++ frontend on the CAHP system
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
T=r02r
timeout 600 python bench.py --params 80bit --cpu-sample 0 2>/dev/null | tail -1 > gpurun_out/${T}_bench_80bit.json
: > gpurun_out/${T}_bench_netlist.txt
for net in cahp-ruby mux-ram cahp-system; do
  timeout 900 python tools/bench_netlist.py --net $net --clocks 4 2>/dev/null | tail -1 >> gpurun_out/${T}_bench_netlist.txt
done
R=tests/golden/reftest
./iyokan_amd/host/test0_hip --hip-run $R/config-toml/cahp-ruby-mux.toml $R/in/test09.in -c 7 --expect $R/out/test09-ruby.out > gpurun_out/${T}_cpp_cahp.txt 2>&1
cat gpurun_out/${T}_bench_netlist.txt gpurun_out/${T}_cpp_cahp.txt; cut -c1-400 gpurun_out/${T}_bench_80bit.json
