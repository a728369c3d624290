#!/bin/bash
# Batch-size sweep of the rotation kernels (forced through IYK_HIP_ROT_KERNEL): average blind-rotate launch ms per batch.
#   bash tools/sweep_rot.sh [sizes...]      KERNELS="lat3 w32 fft" selects kernels
cd "$(dirname "$0")/.."
sizes=${@:-"32 128 256 512 768 1024 1280 1536 2048"}
kernels=${KERNELS:-"lat3 w32"}
ms() { IYK_HIP_ROT_KERNEL=$1 timeout 200 python bench.py --gates $2 --steps 3 --warmup 1 --cpu-sample 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['roofline']['avg_launch_ms'],2))"; }
for g in $sizes; do
  line="gates=$g"
  for k in $kernels; do line="$line ${k}_ms=$(ms $k $g)"; done
  echo "$line"
done
