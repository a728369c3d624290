#!/bin/bash
# Same-box A/B of the level plans: --plan nospread (round 5's candidates) against balanced (with round 6's capped candidates), then the
# three benchmark netlists with the default plan.   -> gpurun_out/r06b_plan_ab.txt, gpurun_out/r06b_bench_netlist_balanced.txt
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
: > gpurun_out/r06b_plan_ab.txt
for rep in 1 2; do for plan in nospread balanced; do for net in cahp-ruby cahp-system; do
echo "$plan $net $(timeout 280 python tools/bench_netlist.py --net $net --plan $plan --clocks 3 --burst 4 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['s_per_clock'],5), round(d['s_per_clock_back_to_back'],5), round(d['model_s_per_clock'],5), d['outputs_match_plaintext'])")" >> gpurun_out/r06b_plan_ab.txt
done; done; done
cat gpurun_out/r06b_plan_ab.txt
for net in cahp-ruby cahp-system mux-ram; do timeout 600 python tools/bench_netlist.py --net $net 2>/dev/null | tail -1; done > gpurun_out/r06b_bench_netlist_balanced.txt
cut -c1-250 gpurun_out/r06b_bench_netlist_balanced.txt
