"""CPU plumbing (BASELINE config #1 shape): the C++ engine + plaintext backend self-test builds and passes."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_engine_plain_backend():
    host = os.path.join(ROOT, "iyokan_amd", "host")
    subprocess.run(["make", "-C", host], check=True, capture_output=True)
    out = subprocess.run([os.path.join(host, "test0_hip")], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "plain: all tests passed" in out.stdout
