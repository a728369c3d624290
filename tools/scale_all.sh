#!/bin/bash
# The whole scaling table in ONE command, for the first lease with more than one GPU:
#   bash tools/scale_all.sh [outfile]          # default gpurun_out/scale_all.jsonl
#   - bench.py --gpus {1,2,4,8}: 65 536 flat NANDs, 128-bit set, then the 80-bit set (configs #2 / #5; strong line + weak leg)
#   - tools/bench_netlist.py --gpus {1,8}: config #3 (mux-ram-8-16-16) and config #4 (CAHP system), 10 clocks each
#   - before anything else, on a box with >= 2 GPUs: the distinct-device replica-exchange and multi-GPU init tests (one JSON line)
# One JSON line per run, in that order.  bench.py / bench_netlist.py refuse (status 3, no JSON) to report an N-GPU number
# from fewer than N devices; this script then stops with status 3 after writing what it has — a partial table is labelled
# by its last line {"refused": ...}, never silently padded.  The reference takes its GPU count the same way
# (/root/reference/src/main.cpp:147-148).  GPUS="1 2" restricts the counts (e.g. a 2-GPU box).
#   DRY_RUN=1 bash tools/scale_all.sh      # print the exact commands, in order, and run nothing (no GPU needed; tests/test_scale_model.py)
# When the table exists, compare it with the model's prediction: python tools/scale_model.py --check <outfile>
# (profiles/r06_scale_model.json: flat batches >= 0.97 efficiency at N = 8, config #4 <= 0.115 s per clock at N = 8).
cd "$(dirname "$0")/.." && export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
out=${1:-gpurun_out/scale_all.jsonl}
mkdir -p "$(dirname "$out")"; : > "$out"
GPUS=${GPUS:-"1 2 4 8"}
NET_GPUS=${NET_GPUS:-"1 8"}
run() {   # run <label> <cmd...>
  local label=$1; shift
  if [ -n "$DRY_RUN" ]; then echo "[$label] $*"; return 0; fi
  local line rc
  "$@" > /tmp/scale_all.out 2>/tmp/scale_all.err; rc=$?
  line=$(tail -1 /tmp/scale_all.out)
  if [ "$rc" -ne 0 ] || [ -z "$line" ]; then
    echo "{\"refused\": \"$label\", \"status\": $rc, \"stderr\": \"$(tail -1 /tmp/scale_all.err | tr -d '"\\' | cut -c1-200)\"}" >> "$out"
    echo "scale_all: $label failed with status $rc" >&2
    [ "$rc" -eq 3 ] && exit 3
    exit 1
  fi
  echo "$line" >> "$out"
  echo "$label: $(echo "$line" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d.get('value') or d.get('s_per_clock'))")"
}
# FIRST, where the box has more than one GPU: the in-process replica exchange between DISTINCT devices (hipMemcpyPeerAsync over
# xGMI, iyk_hip_arena_sync_slots_multi) and the concurrent multi-GPU init — the two paths a 1-GPU box can only run aliased
if [ -n "$DRY_RUN" ]; then
  echo "[distinct-device tests, only where >= 2 GPUs are visible] python -m pytest tests/test_gpu_zz_debug.py -q -m gpu -k 'fan_out or multi_gpu_init'"
  ndev=0
else
  ndev=$(python -c "import torch; print(torch.cuda.device_count())" 2>/dev/null || echo 0)
fi
if [ -n "$DRY_RUN" ]; then :
elif [ "${ndev:-0}" -ge 2 ]; then
  timeout 900 python -m pytest tests/test_gpu_zz_debug.py -q -m gpu -k "fan_out or multi_gpu_init" > /tmp/scale_all.peer 2>&1; rc=$?
  echo "{\"distinct_device_tests\": \"$(tail -1 /tmp/scale_all.peer | tr -d '"\\' | cut -c1-120)\", \"status\": $rc, \"devices\": $ndev}" >> "$out"
  echo "distinct-device tests: $(tail -1 /tmp/scale_all.peer)"
else
  echo "{\"distinct_device_tests\": \"skipped: $ndev device(s) visible\"}" >> "$out"
fi
for params in 128bit 80bit; do
  for n in $GPUS; do
    run "bench $params x$n" python bench.py --gpus $n --params $params --steps ${STEPS:-4} --warmup 1 --cpu-sample 0
  done
done
for net in mux-ram cahp-system; do
  for n in $NET_GPUS; do
    run "netlist $net x$n" python tools/bench_netlist.py --net $net --gpus $n --clocks ${CLOCKS:-10}
  done
done
[ -n "$DRY_RUN" ] && { echo "scale_all: dry run, nothing executed"; exit 0; }
echo "scale_all: table complete -> $out"
