"""GPU parity proper: the HIP path, called through the C ABI, against the oracle on the same
seeded inputs (bit-exact), against the committed golden digests, and through size-independent
properties at the full BASELINE size."""
import hashlib
import json
import os
import sys

import numpy as np
import pytest

from iyokan_amd import client
from iyokan_amd.params import OPS, PLAIN

pytestmark = pytest.mark.gpu
GOLD_DIR = os.path.join(os.path.dirname(__file__), "golden")
TT = json.load(open(os.path.join(GOLD_DIR, "truth_tables.json")))
BINOPS = ["AND", "NAND", "ANDNOT", "OR", "NOR", "ORNOT", "XOR", "XNOR"]


@pytest.fixture(scope="module")
def gpu128(keys128):
    from iyokan_amd import hip

    hip.initialize(keys128, device_ids=(0,))
    st = hip.Stream(0)
    yield hip, st
    st.destroy()
    hip.cleanup()


def _run(hip, st, arena_host, ops, in0, in1, in2, out):
    arena = hip.Arena(arena_host.shape[0])
    st.upload(arena, 0, arena_host)
    st.gate_batch(arena, ops, in0, in1, in2, out)
    st.sync()
    got = st.download(arena, 0, arena_host.shape[0])
    arena.free()
    return got


def test_all_gate_kinds_bit_exact_vs_oracle(gpu128, keys128, oracle128):
    hip, st = gpu128
    p = keys128.params
    rng = np.random.default_rng(3)
    nin = 64
    bits = rng.integers(0, 2, size=nin).astype(np.uint8)
    kinds = BINOPS * 3 + ["MUX"] * 6 + ["NOT", "COPY", "CONSTONE", "CONSTZERO"]
    ops, in0, in1, in2, out, want = [], [], [], [], [], []
    for g, kind in enumerate(kinds):
        a, b, s = (int(v) for v in rng.integers(0, nin, size=3))
        ops.append(OPS[kind])
        out.append(nin + g)
        if kind in BINOPS:
            in0.append(a); in1.append(b); in2.append(-1); want.append(PLAIN[kind](int(bits[a]), int(bits[b])))
        elif kind == "MUX":
            in0.append(a); in1.append(b); in2.append(s); want.append(PLAIN["MUX"](int(bits[a]), int(bits[b]), int(bits[s])))
        elif kind in ("NOT", "COPY"):
            in0.append(a); in1.append(-1); in2.append(-1); want.append(PLAIN[kind](int(bits[a])))
        else:
            in0.append(-1); in1.append(-1); in2.append(-1); want.append(PLAIN[kind]())
    host = np.zeros((nin + len(kinds), p.n + 1), dtype=np.uint32)
    host[:nin] = client.encrypt_bits(keys128, bits, seed=2)
    got = _run(hip, st, host, ops, in0, in1, in2, out)
    ref = host.copy()
    oracle128.gate_batch(ops, in0, in1, in2, out, ref, nthreads=os.cpu_count() or 1)
    assert np.array_equal(got[:nin], host[:nin])             # inputs untouched
    assert np.array_equal(got[nin:], ref[nin:])              # bit-exact vs oracle
    assert list(client.decrypt_bits(keys128, got[nin:])) == want


def test_two_levels_of_random_gates_bit_exact_vs_oracle(gpu128, keys128, oracle128):
    """600 random binary / MUX gates on fresh encryptions, then 600 more on the OUTPUTS of the first level
    (bootstrapped ciphertexts have a different noise shape than fresh ones): every ciphertext word must
    equal the oracle's.  Crosses the dispatch boundaries (600 + MUX rotations > 256: one-wave-per-level
    kernel; nothing here fills a 2048 round)."""
    hip, st = gpu128
    p = keys128.params
    rng = np.random.default_rng(1234)
    nin, ng = 96, 600
    bits = rng.integers(0, 2, size=nin).astype(np.uint8)
    host = np.zeros((nin + 2 * ng, p.n + 1), dtype=np.uint32)
    host[:nin] = client.encrypt_bits(keys128, bits, seed=4242)
    kinds = BINOPS + ["MUX", "MUX"]
    ref = host.copy()
    got = host.copy()
    lo, hi = 0, nin
    for level in range(2):
        ops = np.array([OPS[kinds[k]] for k in rng.integers(0, len(kinds), size=ng)], dtype=np.int32)
        in0, in1, in2 = (rng.integers(lo, hi, size=ng).astype(np.int32) for _ in range(3))
        in2 = np.where(ops == OPS["MUX"], in2, -1).astype(np.int32)
        out = np.arange(nin + level * ng, nin + (level + 1) * ng, dtype=np.int32)
        got = _run(hip, st, got, ops, in0, in1, in2, out)
        oracle128.gate_batch(ops, in0, in1, in2, out, ref, nthreads=os.cpu_count() or 1)
        lo, hi = nin + level * ng, nin + (level + 1) * ng
    assert np.array_equal(got, ref)


def test_reference_truth_tables_on_gpu(gpu128, keys128):
    """test0-style known answers (/root/reference/src/test0.cpp:86-94,130-136) on fresh encryptions."""
    hip, st = gpu128
    p = keys128.params
    host = np.zeros((2 + 32 + 8 + 2, p.n + 1), dtype=np.uint32)
    host[:2] = client.encrypt_bits(keys128, [0, 1], seed=5)
    ops, in0, in1, in2, out, want = [], [], [], [], [], []
    slot = 2
    for kind in BINOPS:
        for (a, b), w in zip(TT["binary_inputs"], TT[kind]):
            ops.append(OPS[kind]); in0.append(a); in1.append(b); in2.append(-1); out.append(slot); want.append(w); slot += 1
    for a, b, s, w in TT["MUX"]:
        ops.append(OPS["MUX"]); in0.append(a); in1.append(b); in2.append(s); out.append(slot); want.append(w); slot += 1
    for a, w in TT["NOT"]:
        ops.append(OPS["NOT"]); in0.append(a); in1.append(-1); in2.append(-1); out.append(slot); want.append(w); slot += 1
    got = _run(hip, st, host, ops, in0, in1, in2, out)
    assert list(client.decrypt_bits(keys128, got[2:])) == want


def test_trivial_inputs(gpu128, keys128, oracle128):
    """The reference's own GPU tests use trivial ciphertexts (test0.cpp:702-710): every abar is 0."""
    hip, st = gpu128
    p = keys128.params
    host = np.zeros((6, p.n + 1), dtype=np.uint32)
    host[0], host[1] = client.trivial(p, 0), client.trivial(p, 1)
    ops = [OPS["NAND"]] * 4
    in0, in1 = [0, 0, 1, 1], [0, 1, 0, 1]
    got = _run(hip, st, host, ops, in0, in1, [-1] * 4, [2, 3, 4, 5])
    ref = host.copy()
    oracle128.gate_batch(ops, in0, in1, [-1] * 4, [2, 3, 4, 5], ref, nthreads=4)
    assert np.array_equal(got, ref)
    assert list(client.decrypt_bits(keys128, got[2:])) == TT["NAND"]


def test_golden_digests(gpu128, keys128):
    """Committed fixture (tests/golden/make_vectors.py, generated with the oracle in the build
    container): sha256 of each output ciphertext for seeded keys/inputs."""
    hip, st = gpu128
    gold = json.load(open(os.path.join(GOLD_DIR, "gate_vectors_128.json")))
    p = keys128.params
    assert gold["key_seed"] == 1 and gold["params"]["n"] == p.n
    bits = np.array(gold["input_bits"], dtype=np.uint8)
    host = np.zeros((len(bits) + len(gold["gates"]), p.n + 1), dtype=np.uint32)
    host[: len(bits)] = client.encrypt_bits(keys128, bits, seed=gold["data_seed"])
    assert hashlib.sha256(host[: len(bits)].tobytes()).hexdigest() == gold["inputs_sha256"]
    g = gold["gates"]
    got = _run(hip, st, host, [OPS[x["op"]] for x in g], [x["in0"] for x in g], [x["in1"] for x in g],
               [x["in2"] for x in g], [x["out"] for x in g])
    for x in g:
        assert hashlib.sha256(got[x["out"]].tobytes()).hexdigest() == x["sha256"], x
        assert int(client.decrypt_bits(keys128, got[x["out"]])[0]) == x["bit"]


def test_host_pointer_gate_and_stream_query(gpu128, keys128, oracle128):
    """cufhe::Nand(out, in0, in1, st) shape + StreamQuery polling."""
    hip, st = gpu128
    ca, cb = client.encrypt_bits(keys128, [1, 0], seed=77)
    got = st.gate_host("NAND", ca, cb)
    assert st.query() is True
    assert np.array_equal(got, oracle128.gate(OPS["NAND"], ca, cb))
    st2 = hip.Stream(0)
    got2 = st2.gate_host("MUX", ca, cb, ca)
    assert np.array_equal(got2, oracle128.gate(OPS["MUX"], ca, cb, ca))
    st2.destroy()


def _gate_host_async(hip, st, op, a, b, out):
    """iyk_hip_gate_host without the wrapper's sync: the gate is parked (coalescing) or enqueued; `out` is written at idle."""
    import ctypes

    u32p = ctypes.POINTER(ctypes.c_uint32)
    ptr = lambda x: x.ctypes.data_as(u32p) if x is not None else None
    rc = hip.lib().iyk_hip_gate_host(st.h, OPS[op], ptr(a), ptr(b), None, ptr(out))
    assert rc == 0, hip.lib().iyk_hip_last_error().decode()


def test_coalesced_gates_survive_every_collection_order(gpu128, keys128, oracle128):
    """iyk_hip_gate_host parks gates and sends them as one batch (include/iyokan_hip.h): whatever the callers do next must
    hand every stream its own result.  Orders exercised: a stream SYNCHRONISED while the previous batch's results are still
    uncollected by their streams (they are then delivered eagerly), a stream DESTROYED with its gate still parked, a stream
    re-used for a second gate before it was ever polled, and collection in reverse order."""
    hip, _ = gpu128
    p = keys128.params
    rng = np.random.default_rng(31)
    cts = client.encrypt_bits(keys128, rng.integers(0, 2, size=16).astype(np.uint8), seed=131)
    want = lambda op, i, j: oracle128.gate(OPS[op], cts[i], cts[j])
    sts = [hip.Stream(0) for _ in range(6)]
    outs = [np.zeros(p.n + 1, dtype=np.uint32) for _ in range(8)]
    # first batch: streams 0 .. 2 park; two polls of stream 0 flush it; nobody collects
    for k in range(3):
        _gate_host_async(hip, sts[k], "NAND", cts[k], cts[k + 1], outs[k])
    assert sts[0].query() is False
    sts[0].query()
    # second batch parks behind it: stream 3 and 4; stream 4 is synchronised while 0 .. 2 have not collected
    _gate_host_async(hip, sts[3], "XOR", cts[3], cts[4], outs[3])
    _gate_host_async(hip, sts[4], "OR", cts[4], cts[5], outs[4])
    sts[4].sync()
    assert np.array_equal(outs[4], want("OR", 4, 5))
    for k in (2, 1, 0):   # reverse order; results were delivered when stream 4 drained the first batch
        assert sts[k].query() is True
        assert np.array_equal(outs[k], want("NAND", k, k + 1))
    sts[3].sync()
    assert np.array_equal(outs[3], want("XOR", 3, 4))
    # a stream destroyed with its gate parked still writes its result; a stream given a second gate before any poll finishes the first
    _gate_host_async(hip, sts[5], "AND", cts[5], cts[6], outs[5])
    _gate_host_async(hip, sts[0], "NOR", cts[6], cts[7], outs[6])
    _gate_host_async(hip, sts[0], "XNOR", cts[7], cts[8], outs[7])     # finishes NOR first
    assert np.array_equal(outs[6], want("NOR", 6, 7))
    sts[5].destroy()
    assert np.array_equal(outs[5], want("AND", 5, 6))
    sts[0].sync()
    assert np.array_equal(outs[7], want("XNOR", 7, 8))
    for s in sts[:5]:
        s.destroy()


def test_coalesced_gates_from_two_host_threads(gpu128, keys128, oracle128):
    """Two host threads, 40 one-gate streams each, polling round-robin as upstream's workers do: the coalescer is shared by all
    streams of a GPU and takes one lock; every result equals the oracle's."""
    import threading

    hip, _ = gpu128
    p = keys128.params
    rng = np.random.default_rng(32)
    nct = 24
    cts = client.encrypt_bits(keys128, rng.integers(0, 2, size=nct).astype(np.uint8), seed=132)
    per_thread, gates_each = 40, 120
    jobs = [[(["NAND", "XOR", "ANDNOT", "OR"][int(rng.integers(0, 4))], int(rng.integers(0, nct)), int(rng.integers(0, nct)))
             for _ in range(gates_each)] for _ in range(2)]
    results = [[None] * gates_each for _ in range(2)]
    errors = []

    def worker(t):
        try:
            sts = [hip.Stream(0) for _ in range(per_thread)]
            bufs = [np.zeros(p.n + 1, dtype=np.uint32) for _ in range(per_thread)]
            busy = [-1] * per_thread
            nxt = done = 0
            while done < gates_each:
                for w in range(per_thread):
                    if busy[w] < 0 and nxt < gates_each:
                        op, i, j = jobs[t][nxt]
                        _gate_host_async(hip, sts[w], op, cts[i], cts[j], bufs[w])
                        busy[w] = nxt
                        nxt += 1
                    if busy[w] >= 0 and sts[w].query():
                        results[t][busy[w]] = bufs[w].copy()
                        busy[w] = -1
                        done += 1
            for s in sts:
                s.destroy()
        except Exception as e:   # noqa: BLE001 - reported below
            errors.append(repr(e))

    th = [threading.Thread(target=worker, args=(t,)) for t in range(2)]
    for x in th:
        x.start()
    for x in th:
        x.join(timeout=300)
    assert not errors, errors
    for t in range(2):
        for g in (0, 1, gates_each // 2, gates_each - 1):
            op, i, j = jobs[t][g]
            assert np.array_equal(results[t][g], oracle128.gate(OPS[op], cts[i], cts[j])), (t, g)
    # all of them against a batch on the GPU (word for word; the oracle checked the sample above)
    st = hip.Stream(0)
    for t in range(2):
        host = np.zeros((nct + gates_each, p.n + 1), dtype=np.uint32)
        host[:nct] = cts
        ops = [OPS[o] for o, _, _ in jobs[t]]
        got = _run(hip, st, host, ops, [i for _, i, _ in jobs[t]], [j for _, _, j in jobs[t]], [-1] * gates_each,
                   list(range(nct, nct + gates_each)))
        assert np.array_equal(got[nct:], np.stack(results[t]))
    st.destroy()


def test_ragged_and_empty_batches(gpu128, keys128, oracle128):
    hip, st = gpu128
    p = keys128.params
    bits = np.array([1, 0, 1], dtype=np.uint8)
    host = np.zeros((3 + 5, p.n + 1), dtype=np.uint32)
    host[:3] = client.encrypt_bits(keys128, bits, seed=31)
    arena = hip.Arena(8)
    st.upload(arena, 0, host)
    st.gate_batch(arena, [], [], [], [], [])            # empty batch is a no-op
    for count in (1, 3, 5):                             # odd sizes: last workgroup is partly idle
        ops = [OPS["XOR"]] * count
        in0 = [i % 3 for i in range(count)]
        in1 = [(i + 1) % 3 for i in range(count)]
        out = [3 + i for i in range(count)]
        st.gate_batch(arena, ops, in0, in1, [-1] * count, out)
        st.sync()
        got = st.download(arena, 0, 8)
        ref = host.copy()
        oracle128.gate_batch(ops, in0, in1, [-1] * count, out, ref, nthreads=4)
        assert np.array_equal(got[3:3 + count], ref[3:3 + count])
    arena.free()
    with pytest.raises(hip.IykHipError):
        st.gate_batch(hip.Arena(1), [OPS["NAND"]], [0], [-1], [-1], [0])  # binary gate without in1


def test_chained_levels_stay_on_device(gpu128, keys128):
    """Depth-4 NAND chain on the arena without host round trips; decrypts like the plaintext."""
    hip, st = gpu128
    p = keys128.params
    rng = np.random.default_rng(11)
    width = 16
    bits = rng.integers(0, 2, size=2 * width).astype(np.uint8)
    arena = hip.Arena(2 * width + 4 * width)
    st.upload(arena, 0, client.encrypt_bits(keys128, bits, seed=4))
    plain = list(map(int, bits))
    prev = list(range(2 * width))
    base = 2 * width
    for level in range(4):
        in0 = [prev[(2 * i) % len(prev)] for i in range(width)]
        in1 = [prev[(2 * i + 1 + level) % len(prev)] for i in range(width)]
        out = [base + i for i in range(width)]
        st.gate_batch(arena, [OPS["NAND"]] * width, in0, in1, [-1] * width, out)
        plain += [1 - (plain[a] & plain[b]) for a, b in zip(in0, in1)]
        prev, base = out, base + width
    st.sync()
    got = st.download(arena, 0, 6 * width)
    assert list(client.decrypt_bits(keys128, got)) == plain
    arena.free()


def test_full_size_flat_nand_property(gpu128, keys128, oracle128):
    """BASELINE config #2 shape (65 536 independent NANDs, fresh encryptions): every output decrypts to NAND of the plaintexts,
    EVERY output TLWE has the digest the oracle produced for it in the build container (tests/golden/fullsize_nand_128.bin,
    made by tests/golden/make_fullsize_digests.py on this very workload: word-level parity on all 65 536 outputs), and a
    64-gate sample is compared with the oracle run here."""
    hip, st = gpu128
    p = keys128.params
    G = 65536
    rng = np.random.default_rng(2)
    nin = 4096                                # inputs are re-used across gates to bound host keygen time
    bits = rng.integers(0, 2, size=nin).astype(np.uint8)
    ia = rng.integers(0, nin, size=G).astype(np.int32)
    ib = rng.integers(0, nin, size=G).astype(np.int32)
    arena = hip.Arena(nin + G)
    enc = client.encrypt_bits(keys128, bits, seed=2)
    st.upload(arena, 0, enc)
    out = np.arange(nin, nin + G, dtype=np.int32)
    st.gate_batch(arena, np.full(G, OPS["NAND"], dtype=np.int32), ia, ib, np.full(G, -1, dtype=np.int32), out)
    st.sync()
    got = st.download(arena, nin, G)
    arena.free()
    want = 1 - (bits[ia] & bits[ib])
    assert np.array_equal(client.decrypt_bits(keys128, got), want)
    # noise KAT over all 65 536 outputs: mean and variance of the phase error as CGGI predicts for this key (5 % = 9 sigma
    # of the variance estimate; tests/numpy_tfhe.py)
    import numpy_tfhe

    numpy_tfhe.check_noise_against_cggi(keys128, got, want, rel_tol=0.05)
    _check_fullsize_digests("128", got)
    sample = rng.choice(G, size=64, replace=False)
    ref = np.zeros((nin + 64, p.n + 1), dtype=np.uint32)
    ref[:nin] = enc
    oracle128.gate_batch([OPS["NAND"]] * 64, ia[sample], ib[sample], [-1] * 64,
                         list(range(nin, nin + 64)), ref, nthreads=os.cpu_count() or 1)
    assert np.array_equal(got[sample], ref[nin:])


def _check_fullsize_digests(name, got):
    """every output row against the committed oracle digests (first 8 bytes of sha256 per TLWE) + the checksum of checksums"""
    import hashlib
    import json

    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    sys.path.insert(0, golden)
    import make_fullsize_digests as mk

    want = np.fromfile(os.path.join(golden, f"fullsize_nand_{name}.bin"), dtype=np.uint8).reshape(-1, 8)
    meta = json.load(open(os.path.join(golden, f"fullsize_nand_{name}.json")))
    assert len(want) == len(got) == meta["gates"] and hashlib.sha256(want.tobytes()).hexdigest() == meta["sha256_of_digests"]
    mine = mk.digests(got)
    bad = np.nonzero((mine != want).any(axis=1))[0]
    assert len(bad) == 0, f"{len(bad)} of {len(got)} outputs differ from the oracle's words, first at gate {bad[:5]}"


def test_all_rotation_kernels_agree(gpu128, keys128, oracle128):
    """The two wave-per-rotation kernels (fft: complex FFT on 16-bit key halves, 8 points per lane; w32: FP64 field, 32 points
    per lane) and the workgroup-per-rotation kernel (lat3, FP64 field); IYK_HIP_ROT_KERNEL=fft/w32/lat3 forces one, the default
    picks by batch size.  All must produce identical ciphertexts, equal to the oracle."""
    hip, st = gpu128
    p = keys128.params
    rng = np.random.default_rng(41)
    nin, ng = 48, 70
    bits = rng.integers(0, 2, size=nin).astype(np.uint8)
    ops = rng.choice([OPS["NAND"], OPS["XNOR"], OPS["MUX"], OPS["ORNOT"]], size=ng).astype(np.int32)
    in0, in1, in2 = (rng.integers(0, nin, size=ng).astype(np.int32) for _ in range(3))
    in2 = np.where(ops == OPS["MUX"], in2, -1).astype(np.int32)
    out = np.arange(nin, nin + ng, dtype=np.int32)
    host = np.zeros((nin + ng, p.n + 1), dtype=np.uint32)
    host[:nin] = client.encrypt_bits(keys128, bits, seed=77)
    results = {}
    old = os.environ.get("IYK_HIP_ROT_KERNEL")
    try:
        for mode in ("fft", "latfft", "w32", "lat3"):
            os.environ["IYK_HIP_ROT_KERNEL"] = mode
            results[mode] = _run(hip, st, host, ops, in0, in1, in2, out)
    finally:
        if old is None:
            os.environ.pop("IYK_HIP_ROT_KERNEL", None)
        else:
            os.environ["IYK_HIP_ROT_KERNEL"] = old
    ref = host.copy()
    oracle128.gate_batch(ops, in0, in1, in2, out, ref, nthreads=os.cpu_count() or 1)
    for mode in ("fft", "latfft", "w32", "lat3"):
        assert np.array_equal(results[mode], ref), mode


@pytest.mark.parametrize("ng", [1, 10, 64, 65, 300])
def test_key_switch_kernels_agree(gpu128, keys128, oracle128, ng, monkeypatch):
    """The two key-switch kernels (IYK_HIP_KS_KERNEL=0: 16 gates per workgroup, 3 words per thread; 1, the default:
    16 gates and whole rows per wave, 64 gates per workgroup) subtract the same KSK rows mod 2^32: word-for-word
    equal, at batch sizes below, at and above a workgroup's 64 gates, MUX included (two rotations summed before
    the switch)."""
    hip, st = gpu128
    p = keys128.params
    rng = np.random.default_rng(100 + ng)
    nin = 32
    bits = rng.integers(0, 2, size=nin).astype(np.uint8)
    ops = rng.choice([OPS["NAND"], OPS["XOR"], OPS["MUX"], OPS["ANDNOT"]], size=ng).astype(np.int32)
    in0, in1, in2 = (rng.integers(0, nin, size=ng).astype(np.int32) for _ in range(3))
    in2 = np.where(ops == OPS["MUX"], in2, -1).astype(np.int32)
    out = np.arange(nin, nin + ng, dtype=np.int32)
    host = np.zeros((nin + ng, p.n + 1), dtype=np.uint32)
    host[:nin] = client.encrypt_bits(keys128, bits, seed=78)
    results = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("IYK_HIP_KS_KERNEL", mode)
        results[mode] = _run(hip, st, host, ops, in0, in1, in2, out)
    monkeypatch.delenv("IYK_HIP_KS_KERNEL")
    assert np.array_equal(results["0"], results["1"])
    sample = rng.choice(ng, size=min(ng, 24), replace=False)
    ref = host.copy()
    oracle128.gate_batch(ops[sample], in0[sample], in1[sample], in2[sample], out[sample], ref, nthreads=os.cpu_count() or 1)
    assert np.array_equal(results["1"][out[sample]], ref[out[sample]])


@pytest.mark.parametrize("ng", [4097, 4224, 9000])
def test_key_switch_table_kernel_agrees(gpu128, keys128, oracle128, ng, monkeypatch):
    """Round 6: batches wider than 4 096 gates take keyswitch_lut_kernel — digits in pairs, a table of pre-added KSK rows, the
    row selected by address (IYK_HIP_KS_KERNEL=2, the default).  It subtracts the same rows as the wave kernel (1) in another
    order: word-for-word equal at one gate above the threshold, at a whole number of 128-gate workgroups + a ragged one, and at
    a batch the i range is sliced for; MUX included; a sample against the oracle; the resident key bytes grow by the table."""
    hip, st = gpu128
    p = keys128.params
    rng = np.random.default_rng(700 + ng)
    nin = 32
    bits = rng.integers(0, 2, size=nin).astype(np.uint8)
    ops = rng.choice([OPS["NAND"], OPS["XOR"], OPS["MUX"], OPS["ORNOT"]], size=ng).astype(np.int32)
    in0, in1, in2 = (rng.integers(0, nin, size=ng).astype(np.int32) for _ in range(3))
    in2 = np.where(ops == OPS["MUX"], in2, -1).astype(np.int32)
    out = np.arange(nin, nin + ng, dtype=np.int32)
    host = np.zeros((nin + ng, p.n + 1), dtype=np.uint32)
    host[:nin] = client.encrypt_bits(keys128, bits, seed=81)
    results = {}
    for mode in ("1", "2"):
        monkeypatch.setenv("IYK_HIP_KS_KERNEL", mode)
        results[mode] = _run(hip, st, host, ops, in0, in1, in2, out)
    monkeypatch.delenv("IYK_HIP_KS_KERNEL")
    results["default"] = _run(hip, st, host, ops, in0, in1, in2, out)
    assert np.array_equal(results["1"], results["2"]) and np.array_equal(results["2"], results["default"])
    assert hip.resident_key_bytes() > 300e6     # keys 180 MB + the table's 136 MB
    sample = rng.choice(ng, size=24, replace=False)
    ref = host.copy()
    oracle128.gate_batch(ops[sample], in0[sample], in1[sample], in2[sample], out[sample], ref, nthreads=os.cpu_count() or 1)
    assert np.array_equal(results["2"][out[sample]], ref[out[sample]])


@pytest.mark.parametrize("ng", [1, 15, 16, 17, 100, 1000, 4096, 4097])
def test_key_switch_narrow_frontier_form_agrees(gpu128, keys128, ng, monkeypatch):
    """Round 5: batches of up to 4 096 gates take keyswitch_wave_kernel's SHARED form (a workgroup's four waves on the same 16
    gates, partial sums reduced in LDS before the atomics); IYK_HIP_KS_SHARED_MAX=0 forces round 4's form.  Integer additions
    commute: the same words, at batch sizes around a workgroup's 16 gates and either side of the threshold (4 097 runs the wide
    form in both settings: the control)."""
    hip, st = gpu128
    p = keys128.params
    rng = np.random.default_rng(300 + ng)
    nin = 32
    bits = rng.integers(0, 2, size=nin).astype(np.uint8)
    ops = rng.choice([OPS["NAND"], OPS["XOR"], OPS["MUX"], OPS["ORNOT"]], size=ng).astype(np.int32)
    in0, in1, in2 = (rng.integers(0, nin, size=ng).astype(np.int32) for _ in range(3))
    in2 = np.where(ops == OPS["MUX"], in2, -1).astype(np.int32)
    out = np.arange(nin, nin + ng, dtype=np.int32)
    host = np.zeros((nin + ng, p.n + 1), dtype=np.uint32)
    host[:nin] = client.encrypt_bits(keys128, bits, seed=79)
    shared = _run(hip, st, host, ops, in0, in1, in2, out)
    monkeypatch.setenv("IYK_HIP_KS_SHARED_MAX", "0")
    wide = _run(hip, st, host, ops, in0, in1, in2, out)
    monkeypatch.setenv("IYK_HIP_KS_SHARED_MAX", "4096")
    monkeypatch.setenv("IYK_HIP_KS_SHARED_WG", "64")           # another slicing of the i range: still the same sums
    sliced = _run(hip, st, host, ops, in0, in1, in2, out)
    assert np.array_equal(shared, wide) and np.array_equal(shared, sliced)


def test_adversarial_rows_bit_exact_on_every_kernel(gpu128, keys128, oracle128, monkeypatch):
    """Rows no encryption produces (all-ones, sign bit, mod-switch rounding threshold either side, one non-zero
    coefficient, uniform words; oracle_lib.adversarial_rows): every rotation kernel and both key-switch kernels must
    map them to the oracle's words — digit extremes, exponent wrap-around and skipped CMUX steps, deterministically."""
    import oracle_lib

    hip, st = gpu128
    p = keys128.params
    rows = oracle_lib.adversarial_rows(p.n)
    nin = rows.shape[0]
    kinds = ["NAND", "XOR", "MUX", "ANDNOT", "MUX", "XNOR", "OR", "NAND", "MUX"]
    ops = np.array([OPS[k] for k in kinds], dtype=np.int32)
    in0 = np.arange(nin, dtype=np.int32)
    in1 = ((in0 + 1) % nin).astype(np.int32)
    in2 = np.array([(i + 4) % nin if k == "MUX" else -1 for i, k in enumerate(kinds)], dtype=np.int32)
    out = np.arange(nin, 2 * nin, dtype=np.int32)
    host = np.zeros((2 * nin, p.n + 1), dtype=np.uint32)
    host[:nin] = rows
    ref = host.copy()
    oracle128.gate_batch(ops, in0, in1, in2, out, ref, nthreads=os.cpu_count() or 1)
    for lat in ("fft", "latfft", "w32", "lat3"):
        for ks in ("0", "1"):
            monkeypatch.setenv("IYK_HIP_ROT_KERNEL", lat)
            monkeypatch.setenv("IYK_HIP_KS_KERNEL", ks)
            got = _run(hip, st, host, ops, in0, in1, in2, out)
            assert np.array_equal(got, ref), (lat, ks)
    monkeypatch.delenv("IYK_HIP_ROT_KERNEL")
    monkeypatch.delenv("IYK_HIP_KS_KERNEL")


def test_in_place_outputs(gpu128, keys128, oracle128):
    """A gate may write its result over one of its own inputs (the reference's tasks own separate buffers, a
    device arena invites reuse): every input of a batch is consumed before any output is written, so the result
    equals the one computed into a fresh slot."""
    hip, st = gpu128
    p = keys128.params
    rng = np.random.default_rng(77)
    n = 40
    bits = rng.integers(0, 2, size=2 * n).astype(np.uint8)
    host = np.zeros((3 * n, p.n + 1), dtype=np.uint32)
    host[: 2 * n] = client.encrypt_bits(keys128, bits, seed=555)
    ops = rng.choice([OPS["NAND"], OPS["XOR"], OPS["ORNOT"], OPS["NOT"]], size=n).astype(np.int32)
    in0 = np.arange(n, dtype=np.int32)
    in1 = np.where(ops == OPS["NOT"], -1, np.arange(n, 2 * n)).astype(np.int32)
    in2 = np.full(n, -1, dtype=np.int32)
    fresh = _run(hip, st, host, ops, in0, in1, in2, np.arange(2 * n, 3 * n, dtype=np.int32))
    inplace = _run(hip, st, host, ops, in0, in1, in2, in0)          # out slot = first input slot
    assert np.array_equal(inplace[:n], fresh[2 * n:])
    assert np.array_equal(inplace[n: 2 * n], host[n: 2 * n])
    ref = host.copy()
    oracle128.gate_batch(ops, in0, in1, in2, np.arange(2 * n, 3 * n, dtype=np.int32), ref, nthreads=os.cpu_count() or 1)
    assert np.array_equal(fresh[2 * n:], ref[2 * n:])


def test_repeated_batches_are_bit_identical(gpu128, keys128):
    """Determinism: the same 2500-gate batch run three times (key-switch partial sums are combined by
    integer atomics, whose order varies) gives identical ciphertext words every time."""
    hip, st = gpu128
    rng = np.random.default_rng(99)
    nin, ng = 256, 2500
    bits = rng.integers(0, 2, size=nin).astype(np.uint8)
    host = np.zeros((nin + ng, keys128.params.n + 1), dtype=np.uint32)
    host[:nin] = client.encrypt_bits(keys128, bits, seed=31337)
    ops = rng.choice([OPS["NAND"], OPS["OR"], OPS["MUX"]], size=ng).astype(np.int32)
    in0, in1, in2 = (rng.integers(0, nin, size=ng).astype(np.int32) for _ in range(3))
    in2 = np.where(ops == OPS["MUX"], in2, -1).astype(np.int32)
    out = np.arange(nin, nin + ng, dtype=np.int32)
    runs = [_run(hip, st, host, ops, in0, in1, in2, out) for _ in range(3)]
    assert np.array_equal(runs[0], runs[1]) and np.array_equal(runs[0], runs[2])


@pytest.mark.parametrize("rem", [100, 300, 1200, 1400])
def test_mid_size_batch_uses_both_kernels(gpu128, keys128, rem):
    """round + rem rotations (round = hip.rotation_round(): one job per resident wave): full round on the wave-per-rotation
    kernel, a remainder of up to 1280 on the workgroup-per-rotation kernel (one to five passes of 256), a larger one as a
    second wave-per-rotation round; every output must decrypt correctly (size-independent property) and inputs stay
    untouched."""
    hip, st = gpu128
    rng = np.random.default_rng(43)
    nin, ng = 512, hip.rotation_round() + rem
    bits = rng.integers(0, 2, size=nin).astype(np.uint8)
    ia = rng.integers(0, nin, size=ng).astype(np.int32)
    ib = rng.integers(0, nin, size=ng).astype(np.int32)
    enc = client.encrypt_bits(keys128, bits, seed=78)
    host = np.zeros((nin + ng, keys128.params.n + 1), dtype=np.uint32)
    host[:nin] = enc
    got = _run(hip, st, host, np.full(ng, OPS["XOR"], dtype=np.int32), ia, ib, np.full(ng, -1, dtype=np.int32),
               np.arange(nin, nin + ng, dtype=np.int32))
    assert np.array_equal(got[:nin], enc)
    assert np.array_equal(client.decrypt_bits(keys128, got[nin:]), bits[ia] ^ bits[ib])


def test_dispatch_split_bit_exact_vs_oracle(gpu128, keys128, oracle128):
    """round + 150 rotations: the full round runs on the wave-per-rotation kernel, the remainder on the
    workgroup-per-rotation kernel with a non-zero job offset (`first`).  64 gates from EACH side of the split
    are compared word for word with the oracle (the other tests of the split only decrypt)."""
    hip, st = gpu128
    p = keys128.params
    rng = np.random.default_rng(4321)
    full = hip.rotation_round()
    nin, ng = 300, full + 150
    bits = rng.integers(0, 2, size=nin).astype(np.uint8)
    ia = rng.integers(0, nin, size=ng).astype(np.int32)
    ib = rng.integers(0, nin, size=ng).astype(np.int32)
    ops = rng.choice([OPS["NAND"], OPS["ANDNOT"], OPS["XNOR"]], size=ng).astype(np.int32)
    host = np.zeros((nin + ng, p.n + 1), dtype=np.uint32)
    host[:nin] = client.encrypt_bits(keys128, bits, seed=79)
    got = _run(hip, st, host, ops, ia, ib, np.full(ng, -1, dtype=np.int32), np.arange(nin, nin + ng, dtype=np.int32))
    sample = np.concatenate([rng.choice(full, size=64, replace=False), full + rng.choice(150, size=64, replace=False)])
    ref = np.zeros((nin + len(sample), p.n + 1), dtype=np.uint32)
    ref[:nin] = host[:nin]
    oracle128.gate_batch(ops[sample], ia[sample], ib[sample], [-1] * len(sample),
                         list(range(nin, nin + len(sample))), ref, nthreads=os.cpu_count() or 1)
    assert np.array_equal(got[nin + sample], ref[nin:])


def test_mux_batch_straddles_the_dispatch_split(gpu128, keys128, oracle128):
    """round / 2 + 376 MUX gates = round + 752 rotations: a full round on the wave-per-rotation kernel, 752 on the
    workgroup-per-rotation kernel (three passes of <= 256); an empty batch is a no-op.  96 gates around the split and
    32 random ones are compared word for word with the oracle, all of them by decryption."""
    hip, st = gpu128
    p = keys128.params
    rng = np.random.default_rng(2468)
    full = hip.rotation_round()
    nin, ng = 200, full // 2 + 376
    bits = rng.integers(0, 2, size=nin).astype(np.uint8)
    a, b, s_ = (rng.integers(0, nin, size=ng).astype(np.int32) for _ in range(3))
    host = np.zeros((nin + ng, p.n + 1), dtype=np.uint32)
    host[:nin] = client.encrypt_bits(keys128, bits, seed=80)
    arena = hip.Arena(host.shape[0])
    st.upload(arena, 0, host)
    st.gate_batch(arena, [], [], [], [], [])                                   # count == 0
    st.gate_batch(arena, np.full(ng, OPS["MUX"], dtype=np.int32), a, b, s_, np.arange(nin, nin + ng, dtype=np.int32))
    st.sync()
    got = st.download(arena, 0, host.shape[0])
    arena.free()
    assert np.array_equal(got[:nin], host[:nin])
    assert np.array_equal(client.decrypt_bits(keys128, got[nin:]), np.where(bits[s_] == 1, bits[b], bits[a]))
    sample = np.concatenate([np.arange(full // 2 - 48, full // 2 + 48), rng.choice(ng, size=32, replace=False)])   # gate full / 2 = first of the remainder
    ref = np.zeros((nin + len(sample), p.n + 1), dtype=np.uint32)
    ref[:nin] = host[:nin]
    oracle128.gate_batch([OPS["MUX"]] * len(sample), a[sample], b[sample], s_[sample], list(range(nin, nin + len(sample))), ref,
                         nthreads=os.cpu_count() or 1, mode="fp")
    assert np.array_equal(got[nin + sample], ref[nin:])
