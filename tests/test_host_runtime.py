"""CPU plumbing (BASELINE config #1 shape): the C++ engine + plaintext backend self-test builds and passes,
including the circuits read from Iyokan-L1 / Yosys JSON by the C++ readers (host/readers.hpp)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_engine_plain_backend():
    host = os.path.join(ROOT, "iyokan_amd", "host")
    subprocess.run(["make", "-C", host], check=True, capture_output=True)
    fixtures = os.path.join(ROOT, "tests", "golden", "reftest")   # the JSON circuits of the reference's test0
    out = subprocess.run([os.path.join(host, "test0_hip"), "--fixtures", fixtures], capture_output=True, text=True,
                         timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "plain: all tests passed" in out.stdout


def test_reader_rejects_bad_input(tmp_path):
    """Error convention of the readers = the reference's error::die: message on stderr, exit status 1."""
    host = os.path.join(ROOT, "iyokan_amd", "host")
    bad = tmp_path / "iyokanl1-json"
    bad.mkdir()
    for name in ("pass-4bit", "and-4bit", "and-4_2bit", "mux-4bit", "addr-4bit", "register-4bit", "counter-4bit"):
        (bad / f"{name}-iyokanl1.json").write_text('{"ports": [], "cells": [{"type": "FOO", "id": 1}]}')
    out = subprocess.run([os.path.join(host, "test0_hip"), "--fixtures", str(tmp_path)], capture_output=True, text=True,
                         timeout=120)
    assert out.returncode == 1
    assert "Invalid type: FOO" in out.stdout + out.stderr
