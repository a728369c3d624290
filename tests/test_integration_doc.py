"""INTEGRATION.md's symbol map names a line for every replacement class / function: each must still be where the table says
(the table is what a maintainer navigates by; files move under edits)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
UP = os.path.join(ROOT, "integration", "upstream")


def _rows():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    start = text.index("## Symbol map")
    end = text.index("\n## ", start + 5)
    for line in text[start:end].splitlines():
        if line.startswith("| `") or line.startswith("| test0"):
            cells = [c.strip() for c in line.strip("|").split("|")]
            if len(cells) == 2:
                yield cells[1]


def test_symbol_map_lines_point_at_the_named_symbols():
    checked = 0
    lines = {}
    for cell in _rows():
        current_file = "iyokan_hip.hpp"   # the table's convention: a bare `:N` before any file name means the plugin header
        name = None
        # walk the cell left to right: `Name` tokens, then a location that is either `file.ext:N` or a bare `:N` (same file as before)
        for m in re.finditer(r"`([^`]+)`", cell):
            tok = m.group(1)
            loc = re.fullmatch(r"(?:([\w.]+\.(?:hpp|cpp)))?:(\d+)(?:-(\d+))?", tok)
            if loc is None:
                name = tok
                continue
            if loc.group(1):
                current_file = loc.group(1)
            if current_file is None or name is None:
                continue
            path = os.path.join(UP, current_file)
            assert os.path.exists(path), f"{current_file} named in INTEGRATION.md does not exist under integration/upstream/"
            src = lines.setdefault(path, open(path).read().splitlines())
            lo, hi = int(loc.group(2)), int(loc.group(3) or loc.group(2))
            if not loc.group(1) and hi > len(src):
                name = None
                continue   # a bare `:N` beyond the file: a citation of the REFERENCE's lines inside the prose (e.g. "of `:754-830`")
            assert hi <= len(src), f"{current_file}:{hi} is past the end of the file"
            # the first identifier of the cell's name (e.g. `TaskHIPGateBootstrapped<Op, NumInputs>` -> TaskHIPGateBootstrapped;
            # `HIP2TFHEppBridge`); for lists ("A, B") any of them within the quoted span (+- 2 lines of slack for attributes / templates)
            idents = re.findall(r"[A-Za-z_]\w+", name)
            window = "\n".join(src[max(0, lo - 3): hi + 2])
            assert any(i in window for i in idents if len(i) > 3), f"none of {idents} near {current_file}:{lo}-{hi}"
            checked += 1
            name = None
    assert checked >= 30, f"only {checked} locations parsed from the symbol map"
