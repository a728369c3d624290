#!/bin/bash
# Same-box A/B of throughput-kernel variants at BOTH parameter sets, three repetitions each, with a per-variant summary
# (profiles/r05_prio_ab.txt was made with it).  Build the variants first:  bash tools/ab_fft_variants.sh build "name:-DFLAG ..." ...
# then on the GPU box:  VARIANTS="base a b" bash tools/ab_prio.sh   ->  gpurun_out/r05z_fft_ab.txt (edit the tag below per batch)
export TMPDIR=/tmp
out=gpurun_out/r05z_fft_ab.txt; : > $out
cp iyokan_amd/lib/libiyokan_hip.so /tmp/keep.so
for rep in 1 2 3; do for v in $VARIANTS; do
  cp iyokan_amd/lib/variant_$v.so iyokan_amd/lib/libiyokan_hip.so
  echo "$v $(timeout 300 python bench.py --steps 3 --warmup 1 --cpu-sample 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['roofline']['avg_launch_ms'],2), d['config']['word_check'])")" >> $out
  echo "$v 80bit $(timeout 300 python bench.py --params 80bit --steps 3 --warmup 1 --cpu-sample 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['roofline']['avg_launch_ms'],2), d['config']['word_check'])")" >> $out
done; done
cp /tmp/keep.so iyokan_amd/lib/libiyokan_hip.so
python - <<'P'
import collections
r=collections.defaultdict(list)
for l in open('/root/repo/gpurun_out/r05z_fft_ab.txt'):
    f=l.split()
    k=(f[0],'80' if f[1]=='80bit' else '128'); r[k].append(int(f[-3]))
for k in sorted(r, key=lambda k:(k[1],k[0])): print(k, r[k], round(sum(r[k])/len(r[k])))
P
