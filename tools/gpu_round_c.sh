#!/bin/bash
# GPU session C: full GPU test suite (C++ frontend, CMUX pieces, replicas), smoke, then the round profile (bench + kernel trace + PMC)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
T=r02c
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -30 > gpurun_out/${T}_pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 > gpurun_out/${T}_smoke.txt
bash tools/profile_round.sh r02 > gpurun_out/${T}_profile.log 2>&1
cat gpurun_out/${T}_pytest.txt gpurun_out/${T}_smoke.txt
tail -15 gpurun_out/${T}_profile.log
