#!/bin/bash
# Same-box A/B of two whole libraries on the netlist benches: iyokan_amd/lib/variant_old.so against variant_new.so (build them first: e.g.
# `git archive <rev> iyokan_amd/csrc include tools/src_hash.py | tar -x -C /tmp/old` + hipcc there for the old one, a copy of libiyokan_hip.so for
# the new one).  Two repetitions, mux-ram and the CAHP system, single clocks and a back-to-back burst.  -> gpurun_out/r06b_nl_ab.txt
cd /root/repo
cp iyokan_amd/lib/libiyokan_hip.so /tmp/keep.so
for rep in 1 2; do for v in old new; do
  cp iyokan_amd/lib/variant_$v.so iyokan_amd/lib/libiyokan_hip.so
  for net in mux-ram cahp-system; do echo "$v $net $(timeout 280 python tools/bench_netlist.py --net $net --burst 4 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['s_per_clock'],5), round(d['s_per_clock_back_to_back'],5), round(d['model_s_per_clock'],5), d['outputs_match_plaintext'])")"; done
done; done > gpurun_out/r06b_nl_ab.txt
cp /tmp/keep.so iyokan_amd/lib/libiyokan_hip.so
cat gpurun_out/r06b_nl_ab.txt
