#!/bin/bash
# round-2 GPU session N: LDS-shared key switch — parity of the key-switch kernels, then A/B bench
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_80bit.py -m gpu -x -q > gpurun_out/r02n_pytest.txt 2>&1
tail -5 gpurun_out/r02n_pytest.txt
for k in ${KS_KERNELS:-1 2 3 0}; do
  IYK_HIP_KS_KERNEL=$k timeout 600 python bench.py --cpu-sample 0 --steps 3 --warmup 1 > gpurun_out/r02n_bench_ks$k.json 2> gpurun_out/r02n_bench_ks$k.err
  tail -1 gpurun_out/r02n_bench_ks$k.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ks kernel $k:', d['value'], d['ms_per_step'], d['roofline'].get('keyswitch_avg_launch_ms'))"
done
