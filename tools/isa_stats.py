#!/usr/bin/env python3
"""Instruction mix of one kernel in a hipcc -save-temps .s file: python tools/isa_stats.py file.s <name substring> [...]"""
import re
import sys
from collections import Counter


def main():
    path, pats = sys.argv[1], sys.argv[2:]
    lines = open(path).read().split("\n")
    starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
    for si, s in enumerate(starts):
        name = lines[s][:-1]
        if not all(p in name for p in pats):
            continue
        end = starts[si + 1] if si + 1 < len(starts) else len(lines)
        c = Counter()
        for l in lines[s:end]:
            l = l.strip()
            m = re.match(r"^([a-z_0-9]+)\s", l)
            if not m or l.startswith("."):
                continue
            op = m.group(1)
            if op.startswith("v_"):
                c["valu"] += 1
                if "f64" in op:
                    c["valu_f64"] += 1
                if "permlane" in op:
                    c["permlane"] += 1
                if op.startswith("v_mov") or op.startswith("v_accvgpr"):
                    c["v_mov/acc"] += 1
            elif op.startswith("ds_"):
                c["lds"] += 1
            elif op.startswith("s_waitcnt"):
                c["waitcnt"] += 1
            elif op.startswith("s_barrier"):
                c["barrier"] += 1
            elif op.startswith("s_load") or op.startswith("s_buffer"):
                c["smem"] += 1
            elif op.startswith("s_"):
                c["salu"] += 1
            elif op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_"):
                c["vmem"] += 1
            elif op.startswith("scratch_"):
                c["scratch"] += 1
        print(name[:110])
        print("   ", dict(c))


if __name__ == "__main__":
    main()
