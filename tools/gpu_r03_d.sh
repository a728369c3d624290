#!/bin/bash
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
T=${1:-r03d}
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/${T}_gputests.txt 2>&1
tail -15 gpurun_out/${T}_gputests.txt
timeout 600 python bench.py --gpus 1 --spawn --steps 2 --warmup 1 --cpu-sample 0 2>gpurun_out/${T}_spawn.err | tail -1 > gpurun_out/${T}_spawn.json
python -c "
import json
d=json.load(open('gpurun_out/${T}_spawn.json'))
print('spawn:', round(d['value']), d['n_gpus'], d['rccl_world_size'], d['per_rank_ms_per_step'], d['config']['decrypt_check'])
"
tail -3 gpurun_out/${T}_spawn.err
