"""The exact device math (goldilocks.hpp / ntt32.hpp) compiled for the host and self-checked:
field ops vs __int128, every power-of-two twiddle, two-pass NTT vs its defining sum, and the
NTT product vs schoolbook uint32 negacyclic multiplication (bit-exact); the FP64 field (fp50.hpp):
mulmod/norm vs 128-bit integers at the largest lazy magnitudes, worst-case external products."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_selftest(tmp_path):
    exe = tmp_path / "host_selftest"
    src = os.path.join(ROOT, "iyokan_amd", "csrc", "host_selftest.cpp")
    subprocess.run(["g++", "-O2", "-march=x86-64-v3", "-ffp-contract=off", "-std=c++17", "-o", str(exe), src], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    assert "ALL OK" in out
