#!/bin/bash
# Sustained shader clock of the rotation kernel at a given batch width: GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / kernel
# duration, one rocprofv3 --pmc + --kernel-trace run per width.   bash tools/clock_probe.sh <tag> [widths...]
tag=${1:-clock}; shift
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
out=gpurun_out/${tag}_clock_probe.txt
: > $out
for g in ${@:-2 16 64 256 2048}; do
    d=/tmp/clk_${tag}_$g; rm -rf $d
    timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -d $d -o clk -- python bench.py --gates $g --steps 3 --warmup 1 --cpu-sample 0 > /tmp/clk_$g.log 2>&1
    db=$(find $d -name "*.db" | head -1)
    [ -n "$db" ] && python - "$db" $g >> $out <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
rows = cur.execute("select name, dispatch_id, start, end from kernels where name like '%blind_rotate%' order by start").fetchall()
cnt = dict(cur.execute("select dispatch_id, sum(value) from counters_collection where counter_name='GRBM_GUI_ACTIVE' group by dispatch_id").fetchall())
for name, did, s, e in rows:
    c = cnt.get(did)
    if c: print(f"gates={sys.argv[2]:>5} {name.split('<')[0][-28:]:28s} {(e - s) / 1e6:8.3f} ms  GRBM_GUI_ACTIVE/8 = {c / 8:12.0f}  -> {c / 8 / (e - s):.3f} GHz")
PY
done
cat $out
