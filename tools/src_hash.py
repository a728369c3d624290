#!/usr/bin/env python3
"""Build id of libiyokan_hip.so: the first 16 hex digits of the SHA-256 over the device/host sources that go into it
(iyokan_amd/csrc/*.hpp, iyokan_hip.hip, include/*.h; names and contents, sorted).  __graft_entry__.build() and the Makefile
compile it in (-DIYK_BUILD_ID), iyk_hip_build_id() returns it, tools/profile_round.sh stamps it into the counter file and
bench.py refuses to price a live duration with an instruction count measured on a different build."""
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


def build_id():
    h = hashlib.sha256()
    for f in source_files():
        h.update(os.path.relpath(f, ROOT).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
        h.update(b"\0")
    return h.hexdigest()[:16]


if __name__ == "__main__":
    sys.stdout.write(build_id())
