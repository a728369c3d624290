#!/bin/bash
# GPU session M: full GPU suite (incl. two ranks on one GPU, MUX straddle, doHIP from files), smoke, default bench
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
T=${T:-r02m}
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/${T}_pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 > gpurun_out/${T}_smoke.txt
timeout 900 python bench.py 2>gpurun_out/${T}_bench.err | tail -1 > gpurun_out/${T}_bench.json
cat gpurun_out/${T}_pytest.txt gpurun_out/${T}_smoke.txt; python -c "
import json; d=json.load(open('gpurun_out/${T}_bench.json')); print(d['value'], d['ms_per_step'], d['roofline'].get('valu'), d['cpu_baseline'])"
