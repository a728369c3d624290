#!/bin/bash
# r03 round A: parity of the new t16 kernel + A/B bench t16 vs w32 (same box)
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
mkdir -p gpurun_out
T=${1:-r03a}
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "rotation_kernels_agree or adversarial or all_gate_kinds" > gpurun_out/${T}_parity.txt 2>&1
tail -5 gpurun_out/${T}_parity.txt
for k in t16 w32 t16 w32; do
  IYK_HIP_TP_KERNEL=$k timeout 300 python bench.py --steps 3 --warmup 1 --cpu-sample 0 2>/dev/null | tail -1 > gpurun_out/${T}_bench_$k.json
  python - <<PY
import json
d=json.load(open("gpurun_out/${T}_bench_$k.json"))
print("$k", round(d["value"]), "gates/s", "br ms", round(d["roofline"]["avg_launch_ms"],2), "ks ms", round(d["roofline"]["keyswitch_avg_launch_ms"],2), d["config"]["decrypt_check"])
PY
done
