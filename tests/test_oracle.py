"""Oracle pinning (CPU): the reference's decrypt-level truth tables, the schoolbook anchor,
and noise margins.  Ciphertext-level parity with TFHEpp is unpinned (see oracle header)."""
import json
import os

import numpy as np
import pytest

import oracle_lib
from iyokan_amd import client
from iyokan_amd.params import OPS, PLAIN

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "truth_tables.json")))
BINOPS = ["AND", "NAND", "ANDNOT", "OR", "NOR", "ORNOT", "XOR", "XNOR"]


def test_selfchecks():
    L = oracle_lib.lib()
    assert L.orc_selfcheck_field(123, 2_000_000) == 0
    assert L.orc_selfcheck_product(1024, 32, 5) == 0     # Bg/2 = 32 (128-bit set)
    assert L.orc_selfcheck_product(1024, 512, 6) == 0    # Bg/2 = 512 (80-bit set)
    assert L.orc_selfcheck_product(256, 512, 7) == 0


def test_plain_semantics_match_reference_tables():
    for op in BINOPS:
        assert [PLAIN[op](a, b) for a, b in GOLD["binary_inputs"]] == GOLD[op]
    assert all(PLAIN["MUX"](a, b, s) == o for a, b, s, o in GOLD["MUX"])
    assert all(PLAIN["NOT"](a) == o for a, o in GOLD["NOT"])


@pytest.mark.parametrize("op", BINOPS)
def test_binary_gate_truth_table(op, keys128, oracle128):
    for (a, b), want in zip(GOLD["binary_inputs"], GOLD[op]):
        ca = client.encrypt_bits(keys128, [a], seed=100 + a)[0]
        cb = client.encrypt_bits(keys128, [b], seed=200 + b)[0]
        out = oracle128.gate(OPS[op], ca, cb)
        assert int(client.decrypt_bits(keys128, out)[0]) == want
        # comfortable margin: |phase| within 1/16 of +-1/8
        ph = np.int32(client.phases(keys128, out)[0])
        assert abs(abs(int(ph)) - (1 << 29)) < (1 << 28)


def test_mux_not_const_truth_tables(keys128, oracle128):
    for a, b, s, want in GOLD["MUX"]:
        ca, cb, cs = (client.encrypt_bits(keys128, [v], seed=300 + i)[0] for i, v in enumerate((a, b, s)))
        out = oracle128.gate(OPS["MUX"], ca, cb, cs)
        assert int(client.decrypt_bits(keys128, out)[0]) == want
    for a, want in GOLD["NOT"]:
        ca = client.encrypt_bits(keys128, [a], seed=400)[0]
        assert int(client.decrypt_bits(keys128, oracle128.gate(OPS["NOT"], ca))[0]) == want
    assert int(client.decrypt_bits(keys128, oracle128.gate(OPS["CONSTONE"]))[0]) == 1
    assert int(client.decrypt_bits(keys128, oracle128.gate(OPS["CONSTZERO"]))[0]) == 0
    assert np.array_equal(oracle128.gate(OPS["CONSTONE"]), client.trivial(keys128.params, 1))


def test_trivial_inputs_like_reference_gpu_tests(keys128, oracle128):
    # /root/reference/src/test0.cpp:702-710 feeds the GPU backend trivial ciphertexts
    t0, t1 = client.trivial(keys128.params, 0), client.trivial(keys128.params, 1)
    for (a, b), want in zip(GOLD["binary_inputs"], GOLD["NAND"]):
        out = oracle128.gate(OPS["NAND"], t1 if a else t0, t1 if b else t0)
        assert int(client.decrypt_bits(keys128, out)[0]) == want


@pytest.mark.slow
def test_ntt_path_equals_schoolbook_path(keys128, oracle128):
    """Whole-gate anchor: exact-NTT blind rotation == uint32 schoolbook blind rotation."""
    ca = client.encrypt_bits(keys128, [1], seed=11)[0]
    cb = client.encrypt_bits(keys128, [0], seed=12)[0]
    lin = (np.uint32(0) - ca - cb).astype(np.uint32)
    lin[-1] = np.uint32((int(lin[-1]) + keys128.params.mu) & 0xFFFFFFFF)
    a = oracle128.bootstrap_lvl1(lin, schoolbook=False)
    b = oracle128.bootstrap_lvl1(lin, schoolbook=True)
    assert np.array_equal(a, b)


def test_80bit_set(keys80, oracle80):
    for (a, b), want in zip(GOLD["binary_inputs"], GOLD["XOR"]):
        ca = client.encrypt_bits(keys80, [a], seed=1)[0]
        cb = client.encrypt_bits(keys80, [b], seed=2)[0]
        assert int(client.decrypt_bits(keys80, oracle80.gate(OPS["XOR"], ca, cb))[0]) == want


def test_gate_batch_addressing(keys128, oracle128):
    bits = np.array([0, 1, 1, 0], dtype=np.uint8)
    p = keys128.params
    arena = np.zeros((8, p.n + 1), dtype=np.uint32)
    arena[:4] = client.encrypt_bits(keys128, bits, seed=9)
    oracle128.gate_batch([OPS["NAND"], OPS["MUX"], OPS["NOT"], OPS["COPY"]], [0, 0, 3, 1], [1, 1, -1, -1],
                         [-1, 2, -1, -1], [4, 5, 6, 7], arena, nthreads=4)
    assert list(client.decrypt_bits(keys128, arena[4:])) == [1, 1, 1, 1]


def test_fp_restatement_equals_goldilocks_restatement(keys128, oracle128, keys80, oracle80):
    """oracle/tfhe_oracle_fp.c (FP64-field products, the faster CPU baseline of bench.py) against
    oracle/tfhe_oracle.c (Goldilocks products): same words for every gate kind on fresh and on bootstrapped
    inputs; unavailable (falls back) where its exactness bound fails (80-bit set)."""
    import os

    import numpy as np

    from iyokan_amd import client
    from iyokan_amd.params import OPS

    assert oracle128.has_fp() and not oracle80.has_fp()
    p = keys128.params
    rng = np.random.default_rng(77)
    nin = 12
    bits = rng.integers(0, 2, size=nin).astype(np.uint8)
    kinds = ["AND", "NAND", "ANDNOT", "OR", "NOR", "ORNOT", "XOR", "XNOR", "MUX", "MUX", "NOT"]
    ops = [OPS[k] for k in kinds]
    in0 = [int(v) for v in rng.integers(0, nin, size=len(kinds))]
    in1 = [int(v) if k != "NOT" else -1 for v, k in zip(rng.integers(0, nin, size=len(kinds)), kinds)]
    in2 = [int(v) if k == "MUX" else -1 for v, k in zip(rng.integers(0, nin, size=len(kinds)), kinds)]
    out = list(range(nin, nin + len(kinds)))
    a = np.zeros((nin + 2 * len(kinds), p.n + 1), dtype=np.uint32)
    a[:nin] = client.encrypt_bits(keys128, bits, seed=123)
    b = a.copy()
    nt = os.cpu_count() or 1
    oracle128.gate_batch(ops, in0, in1, in2, out, a, nthreads=nt, mode="goldilocks")
    oracle128.gate_batch(ops, in0, in1, in2, out, b, nthreads=nt, mode="fp")
    assert np.array_equal(a, b)
    # second level: inputs are bootstrapped ciphertexts
    out2 = [o + len(kinds) for o in out]
    in0b = [nin + (i * 5) % len(kinds) for i in range(len(kinds))]
    in1b = [nin + (i * 3 + 1) % len(kinds) if k != "NOT" else -1 for i, k in enumerate(kinds)]
    in2b = [nin + (i * 7 + 2) % len(kinds) if k == "MUX" else -1 for i, k in enumerate(kinds)]
    oracle128.gate_batch(ops, in0b, in1b, in2b, out2, a, nthreads=nt, mode="goldilocks")
    oracle128.gate_batch(ops, in0b, in1b, in2b, out2, b, nthreads=nt, mode="fp")
    assert np.array_equal(a, b)


@pytest.mark.parametrize("which", ["128bit", "80bit"])
def test_fft_restatement_equals_goldilocks_restatement(which, keys128, oracle128, keys80, oracle80):
    """oracle/tfhe_oracle_fft.c — the GPU's arithmetic on the CPU (key split into signed 16-bit halves, folded complex FP64
    transform, two roundings recombined; bench.py's third cpu_baseline restatement) — against oracle/tfhe_oracle.c: the same words
    for every gate kind on fresh encryptions, on bootstrapped inputs and on rows no encryption produces, BOTH parameter sets;
    and its measured rounding distance stays far from the 1/4 at which it aborts."""
    import os

    keys, orc = (keys128, oracle128) if which == "128bit" else (keys80, oracle80)
    assert orc.has_fft()
    p = keys.params
    rng = np.random.default_rng(78)
    adv = oracle_lib.adversarial_rows(p.n)
    nfresh = 12
    nin = nfresh + adv.shape[0]
    kinds = ["AND", "NAND", "ANDNOT", "OR", "NOR", "ORNOT", "XOR", "XNOR", "MUX", "MUX", "NOT", "NAND", "XOR", "MUX", "NAND", "MUX"]
    ops = [OPS[k] for k in kinds]
    in0 = [int(v) for v in rng.integers(0, nin, size=len(kinds))]
    in1 = [int(v) if k != "NOT" else -1 for v, k in zip(rng.integers(0, nin, size=len(kinds)), kinds)]
    in2 = [int(v) if k == "MUX" else -1 for v, k in zip(rng.integers(0, nin, size=len(kinds)), kinds)]
    in0[-4:] = [nfresh, nfresh + 2, nfresh + 4, nfresh + 6]       # the adversarial rows are certainly among the operands
    in1[-4:] = [nfresh + 1, nfresh + 3, nfresh + 5, nfresh + 7]
    out = list(range(nin, nin + len(kinds)))
    a = np.zeros((nin + 2 * len(kinds), p.n + 1), dtype=np.uint32)
    a[:nfresh] = client.encrypt_bits(keys, rng.integers(0, 2, size=nfresh).astype(np.uint8), seed=124)
    a[nfresh:nin] = adv
    b = a.copy()
    nt = os.cpu_count() or 1
    orc.gate_batch(ops, in0, in1, in2, out, a, nthreads=nt, mode="goldilocks")
    orc.gate_batch(ops, in0, in1, in2, out, b, nthreads=nt, mode="fft")
    assert np.array_equal(a, b)
    out2 = [o + len(kinds) for o in out]                            # second level: bootstrapped inputs
    in0b = [nin + (i * 5) % len(kinds) for i in range(len(kinds))]
    in1b = [nin + (i * 3 + 1) % len(kinds) if k != "NOT" else -1 for i, k in enumerate(kinds)]
    in2b = [nin + (i * 7 + 2) % len(kinds) if k == "MUX" else -1 for i, k in enumerate(kinds)]
    orc.gate_batch(ops, in0b, in1b, in2b, out2, a, nthreads=nt, mode="goldilocks")
    orc.gate_batch(ops, in0b, in1b, in2b, out2, b, nthreads=nt, mode="fft")
    assert np.array_equal(a, b)
    d = orc.fft_rounding_distance()
    print(f"fft restatement, {which}: largest rounding distance {d:.3g}")
    assert 0.0 <= d < 1.0 / 64


def test_adversarial_rows_same_words_in_both_restatements(keys128, oracle128):
    """Rows no encryption produces (oracle_lib.adversarial_rows): the Goldilocks and the FP64-field restatements
    must still agree word for word — the pipeline is a deterministic map on u32 vectors, valid ciphertext or not."""
    p = keys128.params
    rows = oracle_lib.adversarial_rows(p.n)
    nin = rows.shape[0]
    kinds = ["NAND", "XOR", "MUX", "ANDNOT", "MUX", "XNOR", "OR", "NAND", "MUX"]
    ops = [OPS[k] for k in kinds]
    in0 = list(range(nin))
    in1 = [(i + 1) % nin for i in range(nin)]
    in2 = [(i + 4) % nin if k == "MUX" else -1 for i, k in enumerate(kinds)]
    out = list(range(nin, 2 * nin))
    a = np.zeros((2 * nin, p.n + 1), dtype=np.uint32)
    a[:nin] = rows
    b = a.copy()
    nt = os.cpu_count() or 1
    oracle128.gate_batch(ops, in0, in1, in2, out, a, nthreads=nt, mode="goldilocks")
    oracle128.gate_batch(ops, in0, in1, in2, out, b, nthreads=nt, mode="fp")
    assert np.array_equal(a, b)
    assert len({a[nin + i].tobytes() for i in range(nin)}) == nin      # nine different outputs: nothing degenerate


@pytest.mark.parametrize("which", ["128", "80"])
def test_tfhepp_algorithm_timing_mode_is_decrypt_equal(which, keys128, keys80, oracle128, oracle80):
    """Mode 4 of the oracle library (bench.py's fourth cpu_baseline entry, round 6): TFHEpp's ALGORITHM — unsplit key, inexact FP64
    products, two rotations per call.  It is a TIMING figure, not a parity reference: the claim checked here is the only one made for
    it — every binary gate kind decrypts to its truth table on fresh encryptions, both parameter sets, odd batch sizes included (the
    last gate is paired with itself) — plus that the exact modes are NOT affected by it having run."""
    keys, orc = (keys128, oracle128) if which == "128" else (keys80, oracle80)
    p = keys.params
    rng = np.random.default_rng(64 + int(which))
    kinds = ["AND", "NAND", "ANDNOT", "OR", "NOR", "ORNOT", "XOR", "XNOR", "NAND"]        # 9: odd on purpose
    nin = 8
    bits = rng.integers(0, 2, size=nin).astype(np.uint8)
    in0 = rng.integers(0, nin, size=len(kinds)).astype(np.int32)
    in1 = rng.integers(0, nin, size=len(kinds)).astype(np.int32)
    ops = np.array([OPS[k] for k in kinds], dtype=np.int32)
    out = np.arange(nin, nin + len(kinds), dtype=np.int32)
    arena = np.zeros((nin + len(kinds), p.n + 1), dtype=np.uint32)
    arena[:nin] = client.encrypt_bits(keys, bits, seed=65)
    exact = orc.gate_batch(ops, in0, in1, [-1] * len(kinds), out, arena.copy(), nthreads=os.cpu_count() or 1)
    fast = orc.gate_batch(ops, in0, in1, [-1] * len(kinds), out, arena.copy(), nthreads=os.cpu_count() or 1,
                          mode="tfhepp_algorithm_inexact")
    want = [PLAIN[k](int(bits[a]), int(bits[b])) for k, a, b in zip(kinds, in0, in1)]
    assert list(client.decrypt_bits(keys, fast[nin:])) == want
    assert list(client.decrypt_bits(keys, exact[nin:])) == want
    assert np.array_equal(fast[:nin], arena[:nin])
    # inexact means: the words MAY differ from the exact ones in their lowest bits — never by more than a rounding each over the
    # n steps of a rotation plus what the key switch makes of it; far inside the noise (the decrypts above) either way
    again = orc.gate_batch(ops, in0, in1, [-1] * len(kinds), out, arena.copy(), nthreads=os.cpu_count() or 1)
    assert np.array_equal(again, exact)
