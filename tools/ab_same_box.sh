#!/bin/bash
# Same-box A/B of whole libraries at BOTH parameter sets: iyokan_amd/lib/variant_<name>.so, names in $VARIANTS, REPS repetitions
# (default 3), interleaved.  Every line carries the bench's word_check (all 65 536 outputs of the timed step against the oracle's
# committed digests).   VARIANTS="base new" bash tools/ab_same_box.sh <tag>   ->  gpurun_out/<tag>_ab.txt
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
T=${1:-r06}
out=gpurun_out/${T}_ab.txt; mkdir -p gpurun_out; : > $out
cp iyokan_amd/lib/libiyokan_hip.so /tmp/keep.so
line() { timeout 300 python bench.py $1 --steps ${STEPS:-3} --warmup 1 --cpu-sample 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['roofline']['avg_launch_ms'],2), d['config']['word_check'], d.get('build_id') or d['config'].get('build_id'))"; }
for rep in $(seq 1 ${REPS:-3}); do for v in $VARIANTS; do
  cp iyokan_amd/lib/variant_$v.so iyokan_amd/lib/libiyokan_hip.so
  echo "$v 128bit $(line)" >> $out
  echo "$v 80bit $(line '--params 80bit')" >> $out
done; done
cp /tmp/keep.so iyokan_amd/lib/libiyokan_hip.so
python - "$out" <<'P'
import collections, sys
r = collections.defaultdict(list)
for l in open(sys.argv[1]):
    f = l.split()
    if len(f) >= 5: r[(f[1], f[0])].append(int(f[2]))
for k in sorted(r): print(k, r[k], round(sum(r[k]) / len(r[k])))
P
