"""Lane-by-lane CPU execution of the HIP kernel's phase functions (csrc/emul.cpp runs the
same blind_rotate_core.hpp the kernel does) must equal the oracle bit for bit."""
import ctypes
import os

import numpy as np
import pytest

from iyokan_amd import client

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
u32p = ctypes.POINTER(ctypes.c_uint32)
u64p = ctypes.POINTER(ctypes.c_uint64)


def _emul():
    return ctypes.CDLL(os.path.join(ROOT, "iyokan_amd", "lib", "libiyk_emul.so"))


@pytest.mark.parametrize("which", ["128", "80"])
def test_emulated_blind_rotate_bit_exact(which, request):
    keys = request.getfixturevalue("keys" + which)
    orc = request.getfixturevalue("oracle" + which)
    p = keys.params
    em = _emul()
    bkntt = np.zeros(p.bk_words, dtype=np.uint64)
    assert em.iyk_emul_bk_ntt(ctypes.byref(p), keys.bk.ctypes.data_as(u32p), bkntt.ctypes.data_as(u64p)) == 0
    for seed, (a, b) in enumerate([(1, 1), (0, 1)]):
        ca = client.encrypt_bits(keys, [a], seed=50 + seed)[0]
        cb = client.encrypt_bits(keys, [b], seed=60 + seed)[0]
        lin = (np.uint32(0) - ca - cb).astype(np.uint32)
        lin[-1] = np.uint32((int(lin[-1]) + p.mu) & 0xFFFFFFFF)
        ref = orc.bootstrap_lvl1(lin)
        got = np.zeros(p.N + 1, dtype=np.uint32)
        assert em.iyk_emul_blind_rotate(ctypes.byref(p), lin.ctypes.data_as(u32p),
                                        bkntt.ctypes.data_as(u64p), got.ctypes.data_as(u32p)) == 0
        assert np.array_equal(ref, got)


@pytest.mark.parametrize("which", ["128", "80"])
def test_emulated_fp64_path_bit_exact(which, request):
    """FP64-field kernel (fp50.hpp, p = 3 * 2^48 + 1097729): lane-by-lane emulation == oracle, and the
    lazily-reduced magnitudes stay well inside the exact-integer range of a double (< 8 p = 2^53).
    80-bit set: every 10-bit gadget digit split into two 5-bit halves (4 virtual levels)."""
    keys128 = request.getfixturevalue("keys" + which)
    oracle128 = request.getfixturevalue("oracle" + which)
    p = keys128.params
    em = _emul()
    em.iyk_emul_fp_max_magnitude.restype = ctypes.c_double
    dp = ctypes.POINTER(ctypes.c_double)
    bk = np.zeros(p.bk_words * (2 if p.l == 2 else 1), dtype=np.float64)
    assert em.iyk_emul_bk_ntt_fp(ctypes.byref(p), keys128.bk.ctypes.data_as(u32p), bk.ctypes.data_as(dp)) == 0
    assert np.abs(bk).max() <= 844424931229697 / 2 + 1
    for seed, (a, b) in enumerate([(1, 1), (0, 1), (1, 0)]):
        ca = client.encrypt_bits(keys128, [a], seed=70 + seed)[0]
        cb = client.encrypt_bits(keys128, [b], seed=80 + seed)[0]
        lin = (np.uint32(0) - ca - cb).astype(np.uint32)
        lin[-1] = np.uint32((int(lin[-1]) + p.mu) & 0xFFFFFFFF)
        got = np.zeros(p.N + 1, dtype=np.uint32)
        assert em.iyk_emul_blind_rotate_fp(ctypes.byref(p), lin.ctypes.data_as(u32p), bk.ctypes.data_as(dp),
                                           got.ctypes.data_as(u32p)) == 0
        ref = oracle128.bootstrap_lvl1(lin)
        assert np.array_equal(ref, got)
        # wave-per-(polynomial, level) low-latency kernel: 16 points per lane, permlane32 exchanges, shared sums
        got3 = np.zeros(p.N + 1, dtype=np.uint32)
        assert em.iyk_emul_blind_rotate_fp_lat3(ctypes.byref(p), lin.ctypes.data_as(u32p), bk.ctypes.data_as(dp),
                                                got3.ctypes.data_as(u32p)) == 0
        assert np.array_equal(ref, got3)
    assert em.iyk_emul_fp_max_magnitude() < 0.95 * 2.0 ** 53 / 844424931229697


@pytest.mark.parametrize("which", ["128", "80"])
def test_emulated_kernels_on_adversarial_rows(which, request):
    """The lane-by-lane emulations of the wave-per-rotation and the workgroup-per-rotation FP64 kernels on rows no
    encryption produces (all-ones, rounding threshold, one non-zero coefficient, uniform words): oracle words, and the
    lazily-reduced magnitudes still inside the exact range."""
    import oracle_lib

    keys = request.getfixturevalue("keys" + which)
    orc = request.getfixturevalue("oracle" + which)
    p = keys.params
    em = _emul()
    em.iyk_emul_fp_max_magnitude.restype = ctypes.c_double
    dp = ctypes.POINTER(ctypes.c_double)
    bk = np.zeros(p.bk_words * (2 if p.l == 2 else 1), dtype=np.float64)
    assert em.iyk_emul_bk_ntt_fp(ctypes.byref(p), keys.bk.ctypes.data_as(u32p), bk.ctypes.data_as(dp)) == 0
    rows = oracle_lib.adversarial_rows(p.n)
    for r in (0, 3, 6, 7):
        lin = np.ascontiguousarray(rows[r])
        ref = orc.bootstrap_lvl1(lin)
        for fn in (em.iyk_emul_blind_rotate_fp, em.iyk_emul_blind_rotate_fp_lat3):
            got = np.zeros(p.N + 1, dtype=np.uint32)
            assert fn(ctypes.byref(p), lin.ctypes.data_as(u32p), bk.ctypes.data_as(dp), got.ctypes.data_as(u32p)) == 0
            assert np.array_equal(ref, got), (r, fn)
    assert em.iyk_emul_fp_max_magnitude() < 0.95 * 2.0 ** 53 / 844424931229697


def test_emulated_direct_decomposition_80bit(keys80, oracle80):
    """Decomp<2, 10, 1> (IYK_HIP_DECOMP=direct): the 80-bit set's 10-bit digits as they are — 2 levels, half the key
    stream.  Exact iff every integer sum stays below p/2, which real key rows give with probability 1 - 2e-17 per gate
    in the worst case over digits (blind_rotate_fp.hpp): both kernels' emulations == oracle on encrypted inputs
    and on the adversarial LWE rows (the rows are adversarial, the key is a real one)."""
    import oracle_lib

    p = keys80.params
    em = _emul()
    em.iyk_emul_fp_max_magnitude.restype = ctypes.c_double
    dp = ctypes.POINTER(ctypes.c_double)
    em.iyk_emul_set_direct(1)
    try:
        bk = np.zeros(p.bk_words, dtype=np.float64)   # no virtual levels: one double per key word
        assert em.iyk_emul_bk_ntt_fp(ctypes.byref(p), keys80.bk.ctypes.data_as(u32p), bk.ctypes.data_as(dp)) == 0
        lins = []
        for seed, (a, b) in enumerate([(1, 1), (0, 1)]):
            ca = client.encrypt_bits(keys80, [a], seed=170 + seed)[0]
            cb = client.encrypt_bits(keys80, [b], seed=180 + seed)[0]
            lin = (np.uint32(0) - ca - cb).astype(np.uint32)
            lin[-1] = np.uint32((int(lin[-1]) + p.mu) & 0xFFFFFFFF)
            lins.append(lin)
        rows = oracle_lib.adversarial_rows(p.n)
        lins += [np.ascontiguousarray(rows[r]) for r in (0, 6)]
        for lin in lins:
            ref = oracle80.bootstrap_lvl1(lin)
            for fn in (em.iyk_emul_blind_rotate_fp, em.iyk_emul_blind_rotate_fp_lat3):
                got = np.zeros(p.N + 1, dtype=np.uint32)
                assert fn(ctypes.byref(p), lin.ctypes.data_as(u32p), bk.ctypes.data_as(dp), got.ctypes.data_as(u32p)) == 0
                assert np.array_equal(ref, got), fn
        assert em.iyk_emul_fp_max_magnitude() < 0.95 * 2.0 ** 53 / 844424931229697
    finally:
        em.iyk_emul_set_direct(0)


def test_direct_decomposition_boundary_is_where_the_bound_says(keys80):
    """The other side of the direct decomposition's contract: it is exact IFF every integer sum stays below p/2.  A
    'bootstrapping key' no key generation produces — every word 2^31 - 1, all aligned — drives a sum to 1023 * 256 *
    2^31 = 2^49 > p/2 = 2^48.58 in the first CMUX step: the direct decomposition then differs from the oracle, the
    default (split digits, |sum| <= 2^48 whatever the key) still equals it word for word."""
    import oracle_lib

    p = keys80.params
    em = _emul()
    dp = ctypes.POINTER(ctypes.c_double)
    bad = client.KeySet(p, keys80.s0, keys80.s1, np.full(p.bk_words, 0x7FFFFFFF, dtype=np.uint32), keys80.ksk)
    orc = oracle_lib.Oracle(bad)
    lin = np.zeros(p.n + 1, dtype=np.uint32)
    lin[0] = 0x7FE00000      # abar_0 = 1023: (X^1023 - 1) * testvector = -2 mu on 1023 coefficients, digit -256 at level 1, and
                             # coefficient 1023 of its product with the all-max row is 1023 * 256 * (2^31 - 1): no wrap-around sign
    ref = orc.bootstrap_lvl1(lin)
    got = {}
    for direct in (0, 1):
        em.iyk_emul_set_direct(direct)
        try:
            bk = np.zeros(p.bk_words * (1 if direct else 2), dtype=np.float64)
            assert em.iyk_emul_bk_ntt_fp(ctypes.byref(p), bad.bk.ctypes.data_as(u32p), bk.ctypes.data_as(dp)) == 0
            out = np.zeros(p.N + 1, dtype=np.uint32)
            assert em.iyk_emul_blind_rotate_fp(ctypes.byref(p), lin.ctypes.data_as(u32p), bk.ctypes.data_as(dp),
                                               out.ctypes.data_as(u32p)) == 0
            got[direct] = out
        finally:
            em.iyk_emul_set_direct(0)
    orc.close()
    assert np.array_equal(got[0], ref)
    assert not np.array_equal(got[1], ref)


# ---- complex-FFT path (csrc/fft512.hpp, blind_rotate_fft.hpp, kernels_fft.hpp) ----------------------------------------
def _fft_keys(em, keys):
    p = keys.params
    dp = ctypes.POINTER(ctypes.c_double)
    bk = np.zeros(p.bk_words * 2, dtype=np.float64)      # cplx [polys][2 halves][512] = 2 doubles per key word
    assert em.iyk_emul_bk_fft(ctypes.byref(p), keys.bk.ctypes.data_as(u32p), bk.ctypes.data_as(dp)) == 0
    return bk


@pytest.mark.parametrize("which", ["128", "80"])
def test_emulated_fft_path_bit_exact(which, request):
    """The complex-FFT kernel (key words split into signed 16-bit halves, 512-point FP64 FFT, rint): lane-by-lane emulation
    == oracle word for word on fresh encryptions AND on the adversarial rows, both parameter sets (the 80-bit set with its
    10-bit digits as they are — no digit split), and every inverse-transform output within 2^-10 of an integer (a trip-wire
    far inside the < 2^-5.6 that DESIGN.md §2b proves for any key and digits; observed: ~2^-20)."""
    import oracle_lib

    keys = request.getfixturevalue("keys" + which)
    orc = request.getfixturevalue("oracle" + which)
    p = keys.params
    em = _emul()
    em.iyk_emul_fft_round_error.restype = ctypes.c_double
    dp = ctypes.POINTER(ctypes.c_double)
    bk = _fft_keys(em, keys)
    em.iyk_emul_fft_round_error(1)
    lins = []
    for seed, (a, b) in enumerate([(1, 1), (0, 1)]):
        ca = client.encrypt_bits(keys, [a], seed=270 + seed)[0]
        cb = client.encrypt_bits(keys, [b], seed=280 + seed)[0]
        lin = (np.uint32(0) - ca - cb).astype(np.uint32)
        lin[-1] = np.uint32((int(lin[-1]) + p.mu) & 0xFFFFFFFF)
        lins.append(lin)
    rows = oracle_lib.adversarial_rows(p.n)
    lins += [np.ascontiguousarray(rows[r]) for r in (0, 3, 6, 7)]
    for lin in lins:
        ref = orc.bootstrap_lvl1(lin)
        # wave-per-rotation kernel, then the workgroup-per-rotation kernel (8 waves, doubled accumulator, frequency-split MAC)
        for fn in (em.iyk_emul_blind_rotate_fft, em.iyk_emul_blind_rotate_fft_lat):
            got = np.zeros(p.N + 1, dtype=np.uint32)
            assert fn(ctypes.byref(p), lin.ctypes.data_as(u32p), bk.ctypes.data_as(dp), got.ctypes.data_as(u32p)) == 0
            assert np.array_equal(ref, got), fn
    assert 0.0 < em.iyk_emul_fft_round_error(0) < 2.0 ** -10


@pytest.mark.parametrize("which", ["128", "80"])
def test_emulated_fft_path_worst_case_magnitudes(which, request):
    """The rounding bound's worst case: a 'bootstrapping key' no key generation produces whose every word has BOTH 16-bit
    halves at -2^15 (0x80008000), or alternates between that and +(2^15 - 1) in both halves, and an LWE row that drives every
    digit of the first steps to its extreme — the norms ||d|| ||k|| the bound is stated in are attained.  The FFT path still
    equals the oracle (exact integer arithmetic) word for word, and the distance from an integer stays below 2^-10."""
    import oracle_lib

    keys = request.getfixturevalue("keys" + which)
    p = keys.params
    em = _emul()
    em.iyk_emul_fft_round_error.restype = ctypes.c_double
    dp = ctypes.POINTER(ctypes.c_double)
    rng = np.random.default_rng(5)
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
        bk = _fft_keys(em, bad)
        em.iyk_emul_fft_round_error(1)
        rows = oracle_lib.adversarial_rows(p.n)
        lin0 = np.zeros(p.n + 1, dtype=np.uint32)
        lin0[0] = 0x7FE00000          # abar_0 = 1023: (X^1023 - 1) tv = -2 mu on 1023 coefficients: extreme digits at once
        lin0[1:8] = 0x33300000
        for lin in (lin0, np.ascontiguousarray(rows[6]), np.ascontiguousarray(rows[7])):
            ref = orc.bootstrap_lvl1(lin)
            # both networks: three radix-8 passes (wave per rotation), and the radix-4 halves of the workgroup-per-rotation kernel
            for fn in (em.iyk_emul_blind_rotate_fft, em.iyk_emul_blind_rotate_fft_lat):
                got = np.zeros(p.N + 1, dtype=np.uint32)
                assert fn(ctypes.byref(p), lin.ctypes.data_as(u32p), bk.ctypes.data_as(dp), got.ctypes.data_as(u32p)) == 0
                assert np.array_equal(ref, got), (kind, fn)
        worst = max(worst, em.iyk_emul_fft_round_error(0))
        orc.close()
    assert worst < 2.0 ** -10, worst


def test_half_transforms_equal_the_full_transform():
    """csrc/fft256.hpp (the narrow-frontier kernel's hand-off-free halves: 4 points per lane, four radix-4 passes) against
    csrc/fft512.hpp on the same random integer input, lane by lane through the kernels' own exchange slots: forward
    F_0 +- W^k' F_1 = the [k2][lane''] spectrum, inverse even / odd coefficients = the full inverse's; to rounding (both are
    exact networks with <= 9 butterfly and <= 6 multiplicative layers)."""
    em = _emul()
    em.iyk_emul_fft256_selftest.restype = ctypes.c_double
    for seed in range(8):
        assert em.iyk_emul_fft256_selftest(seed) < 4e-15


def test_fft256_design_model():
    """tools/fft256_model.py: the half transforms' index algebra in numpy against the O(n^2) definition, and every LDS
    exchange of both directions conflict-free under the ds_write_b128 / ds_read_b128 lane grouping of the CDNA4 guide."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "fft256_model.py")], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = out.stdout.strip().splitlines()
    assert float(lines[0].split()[-1]) < 1e-7 and float(lines[1].split()[2]) < 1e-6     # naive O(n^2) sums of magnitude ~10^3
    ways = [l for l in lines if "write-way" in l]
    assert len(ways) == 6 and all(l.endswith("write-way 1 read-way 1") for l in ways)
