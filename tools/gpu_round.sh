#!/bin/bash
# One GPU call at round end (through gpurun): the whole GPU test suite, the three profile rounds (128-bit, 80-bit, 80-bit direct),
# the netlist benches with both level plans, and the bench lines with the counters just measured.  bash tools/gpu_round.sh <tag>
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
T=${1:-r06}
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/${T}_gputests.txt 2>&1
tail -3 gpurun_out/${T}_gputests.txt
bash tools/profile_round.sh ${T} > gpurun_out/${T}_profile_round.log 2>&1
PARAMS=80bit bash tools/profile_round.sh ${T}_80bit > gpurun_out/${T}_80bit_profile_round.log 2>&1
for net in cahp-ruby cahp-system mux-ram; do timeout 600 python tools/bench_netlist.py --net $net --plan asap 2>/dev/null | tail -1; done > gpurun_out/${T}_bench_netlist.txt
for net in cahp-ruby cahp-system mux-ram; do timeout 600 python tools/bench_netlist.py --net $net 2>/dev/null | tail -1; done > gpurun_out/${T}_bench_netlist_balanced.txt
cat gpurun_out/${T}_bench_netlist.txt | cut -c1-200
# what the narrow levels of a real netlist cost, kernel by kernel (one clock of the CAHP core = 41 levels)
rm -rf /tmp/prof_nl && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_nl -o nl -- python tools/bench_netlist.py --net cahp-ruby --clocks 2 > /tmp/nl.log 2>&1
python tools/rocprof_summary.py $(find /tmp/prof_nl -name "*.db" | head -1) > gpurun_out/${T}_netlist_kernel_trace.txt 2>&1
# inputs of tools/scale_model.py (calibrated cost tables of both parameter sets, key-switch and fixed per-level costs)
timeout 600 python tools/scale_model.py --measure gpurun_out/${T}_model_inputs.json 2>&1 | tail -1 | cut -c1-300
python -c "
import json
for t in ('${T}','${T}_80bit'):
    d=json.load(open('gpurun_out/%s_bench.json'%t)); print(t, round(d['value']), d['roofline']['bound'], round(d['roofline']['frac'],3), d['roofline']['fp64']['frac'] if d['roofline'].get('fp64') else None, d['roofline']['issue'].get('useful_frac'), d['cpu_baseline']['value'] if d.get('cpu_baseline') else None)
"
head -8 gpurun_out/${T}_kernel_trace.txt
# bench lines WITH the counters just measured (bench.py reads profiles/: copy first)
cp gpurun_out/${T}_counters.json profiles/r06_counters.json; cp gpurun_out/${T}_80bit_counters.json profiles/r06_counters_80bit.json
python bench.py 2>/dev/null | tail -1 > gpurun_out/${T}_bench_final.json
python bench.py --params 80bit 2>/dev/null | tail -1 > gpurun_out/${T}_80bit_bench_final.json
python -c "
import json
for t in ('${T}','${T}_80bit'):
    d=json.load(open('gpurun_out/%s_bench_final.json'%t)); r=d['roofline']; print(t, round(d['value']), r['bound'], round(r['frac'],3), r['binding'], r['issue'].get('frac'), r['issue'].get('useful_frac'), r['issue'].get('counters_dropped'))
"
