#!/bin/bash
# GPU session L: kernel trace of the CAHP-ruby core clock (where the non-rotation time of narrow levels goes)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && export TMPDIR=/tmp
T=r02l
rm -rf /tmp/prof_nl && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_nl -o nl -- python tools/bench_netlist.py --net cahp-ruby --clocks 3 > /tmp/nl.log 2>&1
tail -2 /tmp/nl.log > gpurun_out/${T}_netlist_trace.txt
python tools/rocprof_summary.py $(find /tmp/prof_nl -name "*.db" | head -1) >> gpurun_out/${T}_netlist_trace.txt
cat gpurun_out/${T}_netlist_trace.txt
