#!/bin/bash
# Batch-size sweep of the blind-rotate kernels (forced through IYK_HIP_LATENCY_KERNEL): average blind-rotate launch
# time per batch.  Run on the GPU box: bash tools/sweep_kernels.sh [sizes...]   (KERNELS="0 1 3" selects kernels)
cd "$(dirname "$0")/.."
sizes=${@:-"32 128 256 512 768 1024 1536 2048 4096"}
kernels=${KERNELS:-"0 1 2 3"}
ms() { IYK_HIP_LATENCY_KERNEL=$1 timeout 200 python bench.py --gates $2 --steps 3 --warmup 1 --cpu-sample 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['roofline']['avg_launch_ms'],2))"; }
names=(throughput_kernel_ms latency1_kernel_ms latency2_kernel_ms latency3_kernel_ms)
for g in $sizes; do
  line="gates=$g"
  for k in $kernels; do line="$line ${names[$k]}=$(ms $k $g)"; done
  echo "$line"
done
