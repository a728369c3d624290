#!/usr/bin/env python3
"""Per-basic-block instruction mix of one kernel in a hipcc --save-temps .s file, with branch targets, so that the
per-step dynamic count of a loop nest can be read off:  python tools/isa_blocks.py file.s <name substring> [...]"""
import re
import sys
from collections import Counter


def classify(op):
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_load") or op.startswith("s_buffer"):
        return "smem"
    if op.startswith("s_cbranch") or op.startswith("s_branch"):
        return "br"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("scratch_"):
        return "scratch"
    if op.startswith(("global_", "buffer_", "flat_")):
        return "vmem"
    return "other"


def main():
    path, pats = sys.argv[1], sys.argv[2:]
    lines = open(path).read().split("\n")
    starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
    for si, s in enumerate(starts):
        name = lines[s][:-1]
        if not all(p in name for p in pats):
            continue
        end = starts[si + 1] if si + 1 < len(starts) else len(lines)
        print(name[:140])
        blocks, cur, label = [], Counter(), "entry"
        targets = []
        for l in lines[s + 1:end]:
            m = re.match(r"^(\.LBB\w+):", l)
            if m:
                blocks.append((label, cur, targets))
                cur, label, targets = Counter(), m.group(1), []
                continue
            l = l.strip()
            m = re.match(r"^([a-z_0-9]+)(\s|$)", l)
            if not m or l.startswith(".") or l.startswith(";"):
                continue
            op = m.group(1)
            k = classify(op)
            cur[k] += 1
            if "f64" in op:
                cur["f64"] += 1
            if "permlane" in op:
                cur["swap"] += 1
            if op.startswith("v_mov") or op.startswith("v_accvgpr"):
                cur["mov"] += 1
            if k == "br":
                t = re.search(r"(\.LBB\w+)", l)
                targets.append((op.replace("s_cbranch_", "").replace("s_branch", "jmp"), t.group(1) if t else "?"))
            if op == "s_endpgm":
                break
        blocks.append((label, cur, targets))
        for label, c, t in blocks:
            tot = sum(v for k, v in c.items() if k not in ("f64", "swap", "mov"))
            if tot == 0:
                continue
            print(f"  {label:12s} n={tot:5d} " + " ".join(f"{k}={c[k]}" for k in ("valu", "f64", "swap", "mov", "lds", "vmem", "scratch", "salu", "smem", "wait") if c[k])
                  + ("   -> " + ", ".join(f"{a}:{b}" for a, b in t) if t else ""))


if __name__ == "__main__":
    main()
