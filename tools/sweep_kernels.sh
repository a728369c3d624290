#!/bin/bash
# Batch-size sweep of the three blind-rotate kernels (forced through IYK_HIP_LATENCY_KERNEL): average
# blind-rotate launch time per batch.  Run on the GPU box: bash tools/sweep_kernels.sh [sizes...]
cd "$(dirname "$0")/.."
sizes=${@:-"32 128 256 384 512 768 1024 1536 2048 3072 4096 8192 16384"}
ms() { IYK_HIP_LATENCY_KERNEL=$1 timeout 200 python bench.py --gates $2 --steps 3 --warmup 1 --cpu-sample 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['roofline']['avg_launch_ms'],2))"; }
for g in $sizes; do
  echo "gates=$g throughput_kernel_ms=$(ms 0 $g) latency_kernel_ms=$(ms 1 $g) latency2_kernel_ms=$(ms 2 $g)"
done
