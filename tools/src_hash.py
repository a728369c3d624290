#!/usr/bin/env python3
"""Build id of libiyokan_hip.so: the first 16 hex digits of the SHA-256 over the device/host sources that go into it
(iyokan_amd/csrc/*.hpp, iyokan_hip.hip, include/*.h; names and contents, sorted) AND over the sorted -D / -U / -include flags the
build passes to hipcc:  python tools/src_hash.py [flag ...].  The product is built with no such flag; an A/B build
(tools/ab_*.sh: knobs of tools/experiments/) passes its flags here, so its id differs from the product's even when the sources
are the same — a timing-only variant can no longer carry the product's id.  __graft_entry__.build() and
the Makefile compile the id in (-DIYK_BUILD_ID), iyk_hip_build_id() returns it (with the suffix "+x" whenever an experiment
header replaced a product one), tools/profile_round.sh stamps it into the counter file and bench.py refuses to price a live
duration with an instruction count measured on a different build."""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def source_files():
    csrc = os.path.join(ROOT, "iyokan_amd", "csrc")
    inc = os.path.join(ROOT, "include")
    files = [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith(".hpp") or f == "iyokan_hip.hip"]
    files += [os.path.join(inc, f) for f in os.listdir(inc) if f.endswith(".h")]
    return sorted(files)


def build_id(flags=()):
    h = hashlib.sha256()
    for f in source_files():
        h.update(os.path.relpath(f, ROOT).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
        h.update(b"\0")
    # compiler flags that change what is compiled: every -D / -U / -include (an experiment header is hashed by content, too)
    seen = sorted(set(a for a in flags if a.startswith(("-D", "-U", "-include"))))
    if seen:
        h.update(b"flags\0")
        for a in seen:
            h.update(a.encode() + b"\0")
            path = a.split("=", 1)[1].strip("'\"") if a.startswith("-DIYK_EXPERIMENT_") and "=" in a else None
            if path:
                full = os.path.normpath(os.path.join(ROOT, "iyokan_amd", "csrc", path))
                if os.path.exists(full):
                    with open(full, "rb") as fh:
                        h.update(fh.read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    sys.stdout.write(build_id(sys.argv[1:]))
