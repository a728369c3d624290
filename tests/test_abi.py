"""The C-ABI library loads on a GPU-less host and exports every symbol include/iyokan_hip.h
declares; calls that need a device fail with a status code, never crash, never fall back."""
import ctypes
import os
import re

import numpy as np
import pytest

from iyokan_amd import hip
from iyokan_amd.params import params_128bit

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_are_exported():
    hdr = open(os.path.join(ROOT, "include", "iyokan_hip.h")).read()
    declared = set(re.findall(r"\b(iyk_hip_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(hip.EXPORTS)
    L = hip.lib()
    for sym in declared:
        assert getattr(L, sym) is not None


def test_uninitialised_calls_report_state_error():
    L = hip.lib()
    if L.iyk_hip_is_initialized():
        pytest.skip("library already initialised in this process")
    h = ctypes.c_void_p()
    assert L.iyk_hip_stream_create(0, ctypes.byref(h)) == -2
    assert b"not initialised" in L.iyk_hip_last_error()
    assert L.iyk_hip_cleanup() == -2
    p = params_128bit()
    p.N = 512
    z = np.zeros(4, dtype=np.uint32).ctypes.data_as(ctypes.POINTER(ctypes.c_uint32))
    assert L.iyk_hip_init(1, None, ctypes.byref(p), z, z) == -1  # unsupported N rejected before any HIP call


def test_params_mirror_matches_header():
    p = params_128bit()
    assert (p.n, p.N, p.k, p.l, p.Bgbit, p.t, p.basebit, p.mu) == (636, 1024, 1, 3, 6, 7, 2, 1 << 29)
    # SURVEY.md §8(d) contract figure
    assert p.gate_algorithmic_bytes(1, 2) == 80_793_052
    assert p.bk_words * 8 == 62_521_344
