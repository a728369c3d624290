#!/usr/bin/env python3
"""Per-dispatch view of a rocprofv3 --kernel-trace run (rocpd sqlite): every kernel with its start, duration and the idle
gap since the previous kernel's end, plus totals per clock-sized window.  Answers "how much of a netlist clock is the GPU
idle between the kernels of consecutive levels" (profiles/r06_level_gaps.txt).

usage: tools/trace_gaps.py <results.db> [--csv out.csv]
"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    names = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    view = "kernels" if "kernels" in names else None
    if view is None:
        print("# no `kernels` view; objects:", names)
        return
    cols = [d[0] for d in cur.execute(f"select * from {view} limit 1").description]
    print("# columns:", cols)
    pick = lambda *c: next(x for x in c if x in cols)
    q = f"select {pick('name','kernel_name')}, {pick('start','start_timestamp')}, {pick('end','end_timestamp')} from {view} order by 2"
    rows = cur.execute(q).fetchall()
    if "--csv" in sys.argv:
        with open(sys.argv[sys.argv.index("--csv") + 1], "w") as f:
            for n, s, e in rows:
                f.write(f"{n[:60].replace(',', ';')},{s},{e}\n")
    prev_end = None
    busy = gap_total = 0
    hist = {}
    for n, s, e in rows:
        gap = 0 if prev_end is None else max(0, s - prev_end)
        key = n.split("<")[0].split("(")[0][-40:]
        h = hist.setdefault(key, [0, 0, 0])
        h[0] += 1
        h[1] += e - s
        h[2] += gap if gap < 1_000_000 else 0   # gaps above 1 ms are host phases (set-up, checks), not level hand-overs
        prev_end = max(prev_end or 0, e)
    print(f"{'calls':>6} {'avg_us':>10} {'avg_gap_before_us':>18}  kernel")
    for k, (c, d, g) in sorted(hist.items(), key=lambda kv: -kv[1][1]):
        print(f"{c:6d} {d / c / 1e3:10.2f} {g / c / 1e3:18.2f}  {k}")


if __name__ == "__main__":
    main()
