#!/usr/bin/env python3
"""Summarise a rocprofv3 (ROCm 7.2, rocpd sqlite output) run into a small text file for profiles/.

usage: tools/rocprof_summary.py <results.db> [--pmc] > profiles/<name>.txt
Kernel-trace runs: per-kernel calls / total / average duration (the `--stats` view).
PMC runs: per-kernel mean counter values.
"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    print(f"# source: {sys.argv[1]}")
    try:
        rows = cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
        print("# kernel-trace stats (durations in ns as stored by rocprofv3)")
        print(f"{'calls':>6} {'total_ns':>16} {'avg_ns':>16} {'pct':>7}  kernel")
        for name, calls, total, avg, pct in rows:
            print(f"{calls:6d} {total*1e3:16.0f} {avg*1e3:16.0f} {pct:7.3f}  {name[:110]}")
    except sqlite3.Error as e:
        print("# no kernel stats:", e)
    if "--pmc" in sys.argv:
        try:
            rows = cur.execute(
                "select kernel_name, counter_name, sum(value), count(distinct dispatch_id) from counters_collection "
                "group by kernel_name, counter_name").fetchall()
            rows = [(k, c, v / max(n, 1), n) for k, c, v, n in rows]
            print("# PMC counters: value summed over all SEs/XCDs, mean per dispatch")
            for kname, pname, val, cnt in rows:
                print(f"{pname:28s} {val:20.1f}  n={cnt:<4d} {kname[:80]}")
        except sqlite3.Error as e:
            print("# no pmc data:", e)


if __name__ == "__main__":
    main()
