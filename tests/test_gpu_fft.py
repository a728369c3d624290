"""GPU: the complex-FFT rotation kernels (csrc/fft512.hpp, kernels_fft.hpp) — exact by a rounding bound, not by a field.

Every comparison is word for word against the oracle's exact integer arithmetic.  A kernel is forced (IYK_HIP_ROT_KERNEL =
fft: a wave per rotation; latfft: a workgroup per rotation) wherever the size-based dispatch would pick the other one."""
import os

import numpy as np
import pytest

from iyokan_amd import client
from iyokan_amd.params import OPS

pytestmark = pytest.mark.gpu


def _run_batch(hip, host, ops, in0, in1, in2, out):
    st = hip.Stream(0)
    arena = hip.Arena(host.shape[0])
    st.upload(arena, 0, host)
    st.gate_batch(arena, ops, in0, in1, in2, out)
    st.sync()
    got = st.download(arena, 0, host.shape[0])
    arena.free()
    st.destroy()
    return got


@pytest.mark.parametrize("kernel", ["fft", "latfft"])
@pytest.mark.parametrize("which", ["128", "80"])
def test_fft_kernel_all_kinds_and_adversarial_rows(which, kernel, request, monkeypatch):
    """All gate kinds on fresh encryptions + the rows no encryption produces, 700 gates (partial workgroups, idle waves),
    with IYK_HIP_DEBUG=1: the kernel's own record of max |z - rint(z)| must stay below 2^-10 — a trip-wire far inside what DESIGN.md §2b
    PROVES for any key and digits (< 2^-8.5 / 2^-5.1 at the 128- / 80-bit set, against the ½ that rint needs); real keys give ~2^-20."""
    import oracle_lib
    from iyokan_amd import hip

    keys = request.getfixturevalue("keys" + which)
    orc = request.getfixturevalue("oracle" + which)
    monkeypatch.setenv("IYK_HIP_NTT", "fft")
    monkeypatch.setenv("IYK_HIP_ROT_KERNEL", kernel)     # fft: a wave per rotation; latfft: a workgroup of 8 waves per rotation
    monkeypatch.setenv("IYK_HIP_DEBUG", "1")
    monkeypatch.delenv("IYK_HIP_KS_KERNEL", raising=False)
    p = keys.params
    rows = oracle_lib.adversarial_rows(p.n)
    nin, G = 64, 700
    rng = np.random.default_rng(31)
    bits = rng.integers(0, 2, size=nin).astype(np.uint8)
    host = np.zeros((nin + len(rows) + G, p.n + 1), dtype=np.uint32)
    host[:nin] = client.encrypt_bits(keys, bits, seed=41)
    host[nin:nin + len(rows)] = rows
    nsrc = nin + len(rows)
    kinds = rng.choice(["AND", "NAND", "ANDNOT", "OR", "NOR", "ORNOT", "XOR", "XNOR", "MUX"], size=G)
    ops = np.array([OPS[k] for k in kinds], dtype=np.int32)
    in0, in1, in2 = (rng.integers(0, nsrc, size=G).astype(np.int32) for _ in range(3))
    in2 = np.where(ops == OPS["MUX"], in2, -1).astype(np.int32)
    out = np.arange(nsrc, nsrc + G, dtype=np.int32)
    hip.initialize(keys, device_ids=(0,))
    try:
        assert hip.ntt_path() == "fft" and hip.decomposition_levels() == p.l
        got = _run_batch(hip, host, ops, in0, in1, in2, out)
        err = hip.fft_round_error(0)
    finally:
        hip.cleanup()
    ref = host.copy()
    orc.gate_batch(ops, in0, in1, in2, out, ref, nthreads=os.cpu_count() or 1)
    assert np.array_equal(got, ref)
    assert 0.0 < err < 2.0 ** -10, err


@pytest.mark.parametrize("kernel", ["fft", "latfft"])
@pytest.mark.parametrize("which", ["128", "80"])
def test_fft_kernel_worst_case_key_and_digits(which, kernel, request, monkeypatch):
    """The rounding bound's extremes on the device: 'bootstrapping keys' whose every word has both 16-bit halves at -2^15
    (0x80008000), all aligned / randomly mixed with +(2^15 - 1) halves / alternating, driven by rows that put every digit
    at its extreme from the first step — the norms ||d||_2 ||k||_2 of the bound are attained.  Still the oracle's words
    (exact integer arithmetic on the same bogus key), and the recorded distance from an integer stays below 2^-10."""
    import oracle_lib
    from iyokan_amd import hip

    keys = request.getfixturevalue("keys" + which)
    monkeypatch.setenv("IYK_HIP_NTT", "fft")
    monkeypatch.setenv("IYK_HIP_ROT_KERNEL", kernel)
    monkeypatch.setenv("IYK_HIP_DEBUG", "1")
    p = keys.params
    rng = np.random.default_rng(5)
    rows = oracle_lib.adversarial_rows(p.n)
    extra = np.zeros((2, p.n + 1), dtype=np.uint32)
    extra[0, 0] = 0x7FE00000          # abar_0 = 1023: (X^1023 - 1) tv = -2 mu on 1023 coefficients
    extra[0, 1:8] = 0x33300000
    extra[1, :] = 0x7FE00000
    src = np.concatenate([rows, extra])
    nsrc = len(src)
    ops = [OPS["OR"]] * nsrc + [OPS["MUX"]] * 4          # OR(x, x) = 2x + mu keeps the rows' structure
    in0 = list(range(nsrc)) + [0, 3, 6, 9]
    in1 = list(range(nsrc)) + [1, 4, 7, 10]
    in2 = [-1] * nsrc + [2, 5, 8, 0]
    out = list(range(nsrc, nsrc + len(ops)))
    host = np.zeros((nsrc + len(ops), p.n + 1), dtype=np.uint32)
    host[:nsrc] = src
    worst = 0.0
    for kind in range(3):
        if kind == 0:
            bkw = np.full(p.bk_words, 0x80008000, dtype=np.uint32)
        elif kind == 1:
            bkw = np.where(rng.integers(0, 2, p.bk_words) == 1, 0x80008000, 0x7FFF7FFF).astype(np.uint32)
        else:
            bkw = np.where(np.arange(p.bk_words) % 2 == 1, 0x80008000, 0x7FFF7FFF).astype(np.uint32)
        bad = client.KeySet(p, keys.s0, keys.s1, bkw, keys.ksk)
        orc = oracle_lib.Oracle(bad)
        hip.initialize(bad, device_ids=(0,))
        try:
            got = _run_batch(hip, host, ops, in0, in1, in2, out)
            worst = max(worst, hip.fft_round_error(0))
        finally:
            hip.cleanup()
        ref = host.copy()
        orc.gate_batch(ops, in0, in1, in2, out, ref, nthreads=os.cpu_count() or 1)
        orc.close()
        assert np.array_equal(got, ref), kind
    assert 0.0 < worst < 2.0 ** -10, worst


def test_fft_and_field_paths_agree_on_a_round_plus_remainder(keys128, monkeypatch):
    """One full round + a remainder in the default configuration (both FFT kernels: a wave per rotation for the round, a
    workgroup per rotation for the remainder) against the same batch with IYK_HIP_NTT=fp (the two field kernels): identical
    arenas — two different exact products, four kernels."""
    from iyokan_amd import hip

    monkeypatch.delenv("IYK_HIP_ROT_KERNEL", raising=False)
    monkeypatch.delenv("IYK_HIP_DEBUG", raising=False)
    p = keys128.params
    rng = np.random.default_rng(77)
    nin = 256
    bits = rng.integers(0, 2, size=nin).astype(np.uint8)
    enc = client.encrypt_bits(keys128, bits, seed=78)
    got = {}
    for path in ("fft", "fp"):
        monkeypatch.setenv("IYK_HIP_NTT", path)
        hip.initialize(keys128, device_ids=(0,))
        try:
            G = hip.rotation_round() + 300
            ia = rng.integers(0, nin, size=G).astype(np.int32) if path == "fft" else ia
            ib = rng.integers(0, nin, size=G).astype(np.int32) if path == "fft" else ib
            host = np.zeros((nin + G, p.n + 1), dtype=np.uint32)
            host[:nin] = enc
            got[path] = _run_batch(hip, host, np.full(G, OPS["NAND"], dtype=np.int32), ia, ib,
                                   np.full(G, -1, dtype=np.int32), np.arange(nin, nin + G, dtype=np.int32))
        finally:
            hip.cleanup()
    assert np.array_equal(got["fft"], got["fp"])
    assert np.array_equal(client.decrypt_bits(keys128, got["fft"][nin:]), 1 - (bits[ia] & bits[ib]))


@pytest.mark.parametrize("which", ["128", "80"])
def test_both_fft_kernels_give_identical_arenas_over_several_passes(which, request, monkeypatch):
    """1 100 mixed gates (MUX included: 2 rotations) forced through the workgroup-per-rotation kernel — five passes of one rotation
    per CU, the last one partial, every transform in the decimation halves of csrc/fft256.hpp — and through the wave-per-rotation
    kernel (three radix-8 passes): two different FFT networks, one exact product — identical arenas, and the oracle's bits."""
    from iyokan_amd import hip

    keys = request.getfixturevalue("keys" + which)
    p = keys.params
    rng = np.random.default_rng(2024)
    nin, G = 128, 1100
    bits = rng.integers(0, 2, size=nin).astype(np.uint8)
    kinds = rng.choice(["AND", "NAND", "OR", "XOR", "XNOR", "MUX", "ORNOT", "ANDNOT"], size=G)
    ops = np.array([OPS[k] for k in kinds], dtype=np.int32)
    in0, in1, in2 = (rng.integers(0, nin, size=G).astype(np.int32) for _ in range(3))
    in2 = np.where(ops == OPS["MUX"], in2, -1).astype(np.int32)
    out = np.arange(nin, nin + G, dtype=np.int32)
    host = np.zeros((nin + G, p.n + 1), dtype=np.uint32)
    host[:nin] = client.encrypt_bits(keys, bits, seed=99)
    monkeypatch.setenv("IYK_HIP_NTT", "fft")
    monkeypatch.delenv("IYK_HIP_DEBUG", raising=False)
    got = {}
    hip.initialize(keys, device_ids=(0,))
    try:
        for kernel in ("latfft", "fft"):
            monkeypatch.setenv("IYK_HIP_ROT_KERNEL", kernel)
            got[kernel] = _run_batch(hip, host, ops, in0, in1, in2, out)
    finally:
        hip.cleanup()
    assert np.array_equal(got["latfft"], got["fft"])
    a, b, s = bits[in0], bits[in1], bits[np.where(in2 >= 0, in2, 0)]
    want = {"AND": a & b, "NAND": 1 - (a & b), "OR": a | b, "XOR": a ^ b, "XNOR": 1 - (a ^ b), "ORNOT": a | (1 - b),
            "ANDNOT": a & (1 - b), "MUX": np.where(s == 1, b, a)}
    expect = np.array([want[k][g] for g, k in enumerate(kinds)], dtype=np.uint8)
    assert np.array_equal(client.decrypt_bits(keys, got["fft"][nin:]), expect)
