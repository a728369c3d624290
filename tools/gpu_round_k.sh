#!/bin/bash
# GPU session K: full GPU suite, 80-bit bench line, C++ frontend timing on the CAHP system
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
T=r02k
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/${T}_pytest.txt
timeout 600 python bench.py --params 80bit --cpu-sample 0 2>/dev/null | tail -1 > gpurun_out/${T}_bench_80bit.json
R=tests/golden/reftest
./iyokan_amd/host/test0_hip --hip-run $R/config-toml/cahp-ruby-mux.toml $R/in/test09.in -c 7 --expect $R/out/test09-ruby.out > gpurun_out/${T}_cpp_cahp.txt 2>&1
./iyokan_amd/host/test0_hip --hip-run $R/config-toml/cahp-ruby-mux.toml $R/in/test09.in -c 7 --expect $R/out/test09-ruby.out --gpus 2 >> gpurun_out/${T}_cpp_cahp.txt 2>&1
cat gpurun_out/${T}_pytest.txt gpurun_out/${T}_cpp_cahp.txt; cut -c1-300 gpurun_out/${T}_bench_80bit.json
