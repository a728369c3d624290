for g in 256 512 768 1024 1536 2048 3072 4096 8192 16384; do
  a=$(IYK_HIP_LATENCY_KERNEL=0 timeout 200 python bench.py --gates $g --steps 3 --warmup 1 --cpu-sample 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['roofline']['avg_launch_ms'],2))")
  b=$(IYK_HIP_LATENCY_KERNEL=1 timeout 200 python bench.py --gates $g --steps 3 --warmup 1 --cpu-sample 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['roofline']['avg_launch_ms'],2))")
  echo "gates=$g throughput_kernel_ms=$a latency_kernel_ms=$b"
done
