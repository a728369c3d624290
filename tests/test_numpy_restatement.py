"""Formula-level pins of the oracle: (a) the noise of bootstrapped outputs matches the CGGI
prediction computed from the parameters; (b) a third restatement of the whole gate bootstrap in plain numpy
(tests/numpy_tfhe.py: mod-switch, gadget decomposition, CMUX, extraction, key-switch digit extraction, written from
SURVEY.md section 8's formulas, exact FFT products) equals the C oracle word for word on adversarial rows and on fresh
encryptions, both parameter sets.  The GPU twins of (a) live in tests/test_gpu_parity.py / test_gpu_80bit.py."""
import os

import numpy as np
import pytest

import numpy_tfhe as T
import oracle_lib
from iyokan_amd import client
from iyokan_amd.params import OPS


def test_fft_product_is_the_exact_negacyclic_product(keys128):
    """external_product's float64 FFT route == direct integer convolution, on extreme digits and real key words."""
    p = keys128.params
    rng = np.random.default_rng(3)
    kf = T.KeyFFT(keys128.bk, p)
    digits = rng.integers(-32, 32, size=((p.k + 1) * p.l, p.N)).astype(np.int64)
    digits[0] = -32                                      # all digits at the negative extreme
    digits[1] = 31
    got = T.external_product(digits, kf.step(5), p.N)
    bk5 = keys128.bk.reshape(p.n, (p.k + 1) * p.l, p.k + 1, p.N)[5]
    for c in range(2):
        want = np.zeros(p.N, dtype=np.uint64)
        for r in (0, 1, 4):                              # O(N^2) in Python: three rows are enough to catch an index slip
            want = (want + T.negacyclic_direct(digits[r], bk5[r, c], p.N)) & T.MASK32
        sub = digits.copy()
        sub[[2, 3, 5]] = 0
        assert np.array_equal(T.external_product(sub, kf.step(5), p.N)[c], want)
    assert got.shape == (2, p.N)


@pytest.mark.parametrize("which", ["128", "80"])
def test_numpy_restatement_equals_the_oracle(which, request):
    keys = request.getfixturevalue("keys" + which)
    orc = request.getfixturevalue("oracle" + which)
    p = keys.params
    kf = T.KeyFFT(keys.bk, p)
    rows = oracle_lib.adversarial_rows(p.n)
    cases = [rows[r] for r in ((0, 3, 4, 6, 7) if which == "128" else (2, 6, 8))]
    ca, cb = client.encrypt_bits(keys, [1, 0], seed=31)
    cases.append(T.linear("NAND", ca, cb, p))
    cases.append(T.linear("XOR", ca, cb, p))
    for lin in cases:
        lin = np.ascontiguousarray(lin, dtype=np.uint32)
        out, t1 = T.bootstrap(lin, p, kf, keys.ksk)
        ref1 = orc.bootstrap_lvl1(lin)
        assert np.array_equal(t1, ref1)
        assert np.array_equal(out, orc.keyswitch(ref1))
    # the gate as a whole, through the oracle's own linear step
    assert np.array_equal(T.bootstrap(T.linear("NAND", ca, cb, p), p, kf, keys.ksk)[0], orc.gate(OPS["NAND"], ca, cb))
    assert np.array_equal(T.bootstrap(T.linear("ORNOT", ca, cb, p), p, kf, keys.ksk)[0], orc.gate(OPS["ORNOT"], ca, cb))


@pytest.mark.parametrize("which", ["128", "80"])
def test_key_switch_through_the_pair_table_equals_the_oracle(which, request):
    """Round 6's key switch of wide batches adds PRE-ADDED rows — one per pair of 2-bit digits, from a table built from the
    key-switching key (csrc/kernels.hpp: ks_lut_build_kernel / keyswitch_lut_kernel).  Its numpy restatement against the oracle's
    IdentityKeySwitch on lvl1 samples with every digit pattern: uniform words (all 16 pair values, all four values of t = 7's single
    last digit), all-zero digits (a' = -prec: nothing subtracted), all-ones digits, and words one below / at a digit boundary."""
    keys = request.getfixturevalue("keys" + which)
    orc = request.getfixturevalue("oracle" + which)
    p = keys.params
    table = T.keyswitch_pair_table(keys.ksk, p)
    assert table.shape == (p.N, 16 * (p.t // 2) + 4 * (p.t % 2), p.n + 1)
    assert not table[:, 0].any() and not table[:, 16].any()                          # row 0 of every stage: no digit, no row
    rows = keys.ksk.reshape(p.N, p.t, 3, p.n + 1)
    assert np.array_equal(table[:, 4 * 2 + 3], rows[:, 0, 1] + rows[:, 1, 2])      # stage 0, v_h = 2, v_l = 3
    prec = 1 << (32 - (1 + 2 * p.t))
    rng = np.random.default_rng(2026)
    cases = [rng.integers(0, 1 << 32, size=p.N + 1, dtype=np.uint64).astype(np.uint32) for _ in range(3)]
    zero = np.full(p.N + 1, (1 << 32) - prec, dtype=np.uint64).astype(np.uint32)    # a' + prec = 0: every digit 0
    ones = np.full(p.N + 1, (1 << 32) - prec - 1, dtype=np.uint64).astype(np.uint32)  # a' + prec = 2^32 - 1: every digit 3
    edge = rng.integers(0, 1 << 32, size=p.N + 1, dtype=np.uint64)
    edge[::2] = ((edge[::2] >> (32 - 2 * p.t)) << (32 - 2 * p.t)) - prec             # exactly on a boundary of the last digit
    edge[1::2] = ((edge[1::2] >> (32 - 2 * p.t)) << (32 - 2 * p.t)) - prec - 1       # one below it
    cases += [zero, ones, (edge & 0xFFFFFFFF).astype(np.uint32)]
    for t1 in cases:
        t1 = np.ascontiguousarray(t1, dtype=np.uint32)
        want = orc.keyswitch(t1)
        assert np.array_equal(T.keyswitch(t1, keys.ksk, p), want)
        assert np.array_equal(T.keyswitch_by_table(t1, table, p), want)
    assert np.array_equal(T.keyswitch_by_table(zero, table, p)[:p.n], np.zeros(p.n, dtype=np.uint32))


def test_formula_pieces_on_threshold_words():
    """mod-switch and decomposition at their rounding thresholds (the words adversarial_rows is built from)."""
    N = 1024
    lin = np.array([0x000FFFFF, 0x00100000, 0xFFF00000, 0xFFEFFFFF, 0x7FE00000, 0x40000000], dtype=np.uint32)
    abar, bbar = T.modswitch(lin, N)
    assert list(abar) == [0, 1, 0, 2047, 1023] and bbar == (2 * N - 512) % (2 * N)
    # 128-bit set: l = 3, Bgbit = 6: digits recompose to the value rounded to 18 bits, each in [-32, 32)
    rng = np.random.default_rng(1)
    v = np.concatenate([rng.integers(0, 2**32, size=1000, dtype=np.uint64),
                        np.array([0, 1 << 13, (1 << 13) - 1, 0xFFFFFFFF, 0x80000000, 0x7FFFFFFF, 0x82082000, 0x7DF7DFFF], dtype=np.uint64)])
    for l, bg in ((3, 6), (2, 10)):
        d = T.decompose(v, l, bg)
        assert d.min() >= -(1 << (bg - 1)) and d.max() < (1 << (bg - 1))
        rec = sum(d[j].astype(np.int64) << (32 - (j + 1) * bg) for j in range(l)) & T.MASK32
        err = ((rec - v.astype(np.int64) + (1 << 31)) & T.MASK32) - (1 << 31)
        assert np.abs(err).max() <= 1 << (32 - l * bg - 1)


@pytest.mark.parametrize("which,ngates", [("128", 1536), ("80", 2048)])
def test_oracle_output_noise_matches_cggi(which, ngates, request):
    """Noise KAT on the oracle: the phase errors of bootstrapped NAND outputs have the mean (a key-dependent bias of the
    key switch, about one sigma) and the variance CGGI's analysis predicts for this key (the estimate of a variance from n
    samples has relative standard deviation sqrt(2 / n) = 3.6 % at n = 1536: 15 % is 4 sigma)."""
    keys = request.getfixturevalue("keys" + which)
    orc = request.getfixturevalue("oracle" + which)
    p = keys.params
    rng = np.random.default_rng(77)
    nin = 256
    bits = rng.integers(0, 2, size=nin).astype(np.uint8)
    ia, ib = rng.integers(0, nin, size=ngates), rng.integers(0, nin, size=ngates)
    arena = np.zeros((nin + ngates, p.n + 1), dtype=np.uint32)
    arena[:nin] = client.encrypt_bits(keys, bits, seed=99)
    mode = "fp" if orc.has_fp() else "goldilocks"
    orc.gate_batch([OPS["NAND"]] * ngates, ia, ib, [-1] * ngates, np.arange(nin, nin + ngates), arena,
                   nthreads=os.cpu_count() or 1, mode=mode)
    want = 1 - (bits[ia] & bits[ib])
    assert np.array_equal(client.decrypt_bits(keys, arena[nin:]), want)
    T.check_noise_against_cggi(keys, arena[nin:], want, rel_tol=0.15)
    # fresh encryptions carry alpha0 only: the same estimator must see that, too (guards the estimator itself)
    e = T.phase_errors(arena[:nin], bits, keys.s0, p.mu)
    assert abs(e.var() / p.alpha0 ** 2 - 1.0) < 0.3
