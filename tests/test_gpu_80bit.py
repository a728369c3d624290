"""GPU parity at the 80-bit parameter set (BASELINE config #5 shape: l = 2, Bgbit = 10, n = 500, t = 8)."""
import os

import numpy as np
import pytest

from iyokan_amd import client
from iyokan_amd.params import OPS, PLAIN

pytestmark = pytest.mark.gpu


NTT_ENV = {"fft": "fft", "fp50": "fp", "goldilocks": "goldilocks"}


@pytest.mark.parametrize("path,kernel,ks", [("fft", "fft", None), ("fft", "latfft", None), ("fft", None, None), ("fp50", "w32", None),
                                            ("fp50", "lat3", None), ("goldilocks", None, None), ("fp50", None, "0")])
def test_80bit_gates_bit_exact(path, kernel, ks, keys80, oracle80, monkeypatch):
    """All three exact paths at the 80-bit set: complex FFT on 16-bit key halves with the 10-bit digits as they are (the
    default; the FFT kernel forced, and the size-based dispatch), FP64 field with split digits (IYK_HIP_NTT=fp; each of its
    two rotation kernels forced in turn) and the 64-bit integer field (IYK_HIP_NTT=goldilocks); the last case
    forces the workgroup-per-16-gates key switch (the default is the wave-per-16-gates one, t = 8 / 4 chunks of 128 words)."""
    from iyokan_amd import hip

    if ks is None:
        monkeypatch.delenv("IYK_HIP_KS_KERNEL", raising=False)
    else:
        monkeypatch.setenv("IYK_HIP_KS_KERNEL", ks)
    if kernel is None:
        monkeypatch.delenv("IYK_HIP_ROT_KERNEL", raising=False)
    else:
        monkeypatch.setenv("IYK_HIP_ROT_KERNEL", kernel)
    old = os.environ.get("IYK_HIP_NTT")
    os.environ["IYK_HIP_NTT"] = NTT_ENV[path]
    hip.initialize(keys80, device_ids=(0,))
    if old is None:
        os.environ.pop("IYK_HIP_NTT", None)
    else:
        os.environ["IYK_HIP_NTT"] = old
    try:
        assert hip.ntt_path() == path
        st = hip.Stream(0)
        p = keys80.params
        rng = np.random.default_rng(5)
        bits = rng.integers(0, 2, size=16).astype(np.uint8)
        kinds = ["AND", "NAND", "ANDNOT", "OR", "NOR", "ORNOT", "XOR", "XNOR", "MUX", "MUX", "NOT"]
        ops, in0, in1, in2, out, want = [], [], [], [], [], []
        for g, k in enumerate(kinds):
            a, b, s = g % 16, (3 * g + 5) % 16, (7 * g + 2) % 16
            ops.append(OPS[k]); out.append(16 + g); in0.append(a)
            in1.append(b if k != "NOT" else -1); in2.append(s if k == "MUX" else -1)
            want.append(PLAIN[k](int(bits[a])) if k == "NOT" else PLAIN[k](int(bits[a]), int(bits[b]), int(bits[s]))
                        if k == "MUX" else PLAIN[k](int(bits[a]), int(bits[b])))
        host = np.zeros((16 + len(kinds), p.n + 1), dtype=np.uint32)
        host[:16] = client.encrypt_bits(keys80, bits, seed=3)
        arena = hip.Arena(host.shape[0])
        st.upload(arena, 0, host)
        st.gate_batch(arena, ops, in0, in1, in2, out)
        st.sync()
        got = st.download(arena, 0, host.shape[0])
        arena.free()
        st.destroy()
    finally:
        hip.cleanup()
    ref = host.copy()
    oracle80.gate_batch(ops, in0, in1, in2, out, ref, nthreads=os.cpu_count() or 1)
    assert np.array_equal(got, ref)
    assert list(client.decrypt_bits(keys80, got[16:])) == want


@pytest.mark.parametrize("ng", [5, 16, 300, 4096])
def test_80bit_key_switch_forms_agree(ng, keys80, oracle80, monkeypatch):
    """The key switch's narrow-frontier form (batches of <= 4 096 gates: shared gates, LDS reduction) against round 4's form at
    the 80-bit set (t = 8, four chunks of 128 words per row): the same words, and the oracle's on a sample."""
    from iyokan_amd import hip

    monkeypatch.delenv("IYK_HIP_ROT_KERNEL", raising=False)
    monkeypatch.delenv("IYK_HIP_KS_KERNEL", raising=False)
    hip.initialize(keys80, device_ids=(0,))
    try:
        st = hip.Stream(0)
        p = keys80.params
        rng = np.random.default_rng(500 + ng)
        nin = 24
        bits = rng.integers(0, 2, size=nin).astype(np.uint8)
        ops = rng.choice([OPS["NAND"], OPS["XNOR"], OPS["MUX"]], size=ng).astype(np.int32)
        in0, in1, in2 = (rng.integers(0, nin, size=ng).astype(np.int32) for _ in range(3))
        in2 = np.where(ops == OPS["MUX"], in2, -1).astype(np.int32)
        out = np.arange(nin, nin + ng, dtype=np.int32)
        host = np.zeros((nin + ng, p.n + 1), dtype=np.uint32)
        host[:nin] = client.encrypt_bits(keys80, bits, seed=4)
        got = {}
        for label, smax in (("shared", None), ("wide", "0")):
            if smax is None:
                monkeypatch.delenv("IYK_HIP_KS_SHARED_MAX", raising=False)
            else:
                monkeypatch.setenv("IYK_HIP_KS_SHARED_MAX", smax)
            arena = hip.Arena(host.shape[0])
            st.upload(arena, 0, host)
            st.gate_batch(arena, ops, in0, in1, in2, out)
            st.sync()
            got[label] = st.download(arena, 0, host.shape[0])
            arena.free()
        st.destroy()
    finally:
        hip.cleanup()
    assert np.array_equal(got["shared"], got["wide"])
    sample = rng.choice(ng, size=min(ng, 16), replace=False)
    ref = host.copy()
    oracle80.gate_batch(ops[sample], in0[sample], in1[sample], in2[sample], out[sample], ref, nthreads=os.cpu_count() or 1)
    assert np.array_equal(got["shared"][out[sample]], ref[out[sample]])


@pytest.mark.parametrize("ng", [4097, 6000])
def test_80bit_key_switch_table_kernel_agrees(ng, keys80, oracle80, monkeypatch):
    """The table key switch (round 6, batches wider than 4 096 gates) at the 80-bit set: t = 8 = four digit pairs, KSK rows of
    504 words against table rows of 512.  Same words as the wave kernel, and the oracle's on a sample."""
    from iyokan_amd import hip

    monkeypatch.delenv("IYK_HIP_ROT_KERNEL", raising=False)
    hip.initialize(keys80, device_ids=(0,))
    try:
        st = hip.Stream(0)
        p = keys80.params
        rng = np.random.default_rng(900 + ng)
        nin = 24
        bits = rng.integers(0, 2, size=nin).astype(np.uint8)
        ops = rng.choice([OPS["NAND"], OPS["XNOR"], OPS["MUX"]], size=ng).astype(np.int32)
        in0, in1, in2 = (rng.integers(0, nin, size=ng).astype(np.int32) for _ in range(3))
        in2 = np.where(ops == OPS["MUX"], in2, -1).astype(np.int32)
        out = np.arange(nin, nin + ng, dtype=np.int32)
        host = np.zeros((nin + ng, p.n + 1), dtype=np.uint32)
        host[:nin] = client.encrypt_bits(keys80, bits, seed=6)
        got = {}
        for mode in ("1", "2"):
            monkeypatch.setenv("IYK_HIP_KS_KERNEL", mode)
            arena = hip.Arena(host.shape[0])
            st.upload(arena, 0, host)
            st.gate_batch(arena, ops, in0, in1, in2, out)
            st.sync()
            got[mode] = st.download(arena, 0, host.shape[0])
            arena.free()
        st.destroy()
    finally:
        hip.cleanup()
    assert np.array_equal(got["1"], got["2"])
    sample = rng.choice(ng, size=16, replace=False)
    ref = host.copy()
    oracle80.gate_batch(ops[sample], in0[sample], in1[sample], in2[sample], out[sample], ref, nthreads=os.cpu_count() or 1)
    assert np.array_equal(got["2"][out[sample]], ref[out[sample]])


@pytest.mark.parametrize("path", ["fft", "fp50", "goldilocks"])
def test_80bit_adversarial_rows(path, keys80, oracle80, monkeypatch):
    """Rows no encryption produces (oracle_lib.adversarial_rows) at the 80-bit set, split-digit FP64 field and
    Goldilocks integers: oracle words (digit extremes of the 10-bit decomposition and of its 5-bit halves)."""
    import oracle_lib
    from iyokan_amd import hip

    monkeypatch.delenv("IYK_HIP_ROT_KERNEL", raising=False)
    if path == "fft":
        monkeypatch.setenv("IYK_HIP_ROT_KERNEL", "fft")   # nine gates would go to the narrow-frontier kernel otherwise
    monkeypatch.delenv("IYK_HIP_KS_KERNEL", raising=False)
    monkeypatch.setenv("IYK_HIP_NTT", NTT_ENV[path])
    p = keys80.params
    rows = oracle_lib.adversarial_rows(p.n)
    nin = rows.shape[0]
    kinds = ["NAND", "XOR", "MUX", "ANDNOT", "MUX", "XNOR", "OR", "NAND", "MUX"]
    ops = [OPS[k] for k in kinds]
    in0 = list(range(nin))
    in1 = [(i + 1) % nin for i in range(nin)]
    in2 = [(i + 4) % nin if k == "MUX" else -1 for i, k in enumerate(kinds)]
    out = list(range(nin, 2 * nin))
    host = np.zeros((2 * nin, p.n + 1), dtype=np.uint32)
    host[:nin] = rows
    hip.initialize(keys80, device_ids=(0,))
    try:
        assert hip.ntt_path() == path
        st = hip.Stream(0)
        arena = hip.Arena(host.shape[0])
        st.upload(arena, 0, host)
        st.gate_batch(arena, ops, in0, in1, in2, out)
        st.sync()
        got = st.download(arena, 0, host.shape[0])
        arena.free()
        st.destroy()
    finally:
        hip.cleanup()
    ref = host.copy()
    oracle80.gate_batch(ops, in0, in1, in2, out, ref, nthreads=os.cpu_count() or 1)
    assert np.array_equal(got, ref)


def test_80bit_full_size_flat_nand_property(keys80, oracle80):
    """BASELINE config #5 shape at full size: 65 536 independent NANDs at the 80-bit set; every output decrypts to
    the NAND of its plaintexts, EVERY output has the oracle's digest (tests/golden/fullsize_nand_80.bin), and a 32-gate sample
    is bit-equal to the oracle run here."""
    from iyokan_amd import hip

    hip.initialize(keys80, device_ids=(0,))
    try:
        st = hip.Stream(0)
        p = keys80.params
        G, nin = 65536, 2048
        rng = np.random.default_rng(12)
        bits = rng.integers(0, 2, size=nin).astype(np.uint8)
        ia = rng.integers(0, nin, size=G).astype(np.int32)
        ib = rng.integers(0, nin, size=G).astype(np.int32)
        enc = client.encrypt_bits(keys80, bits, seed=6)
        arena = hip.Arena(nin + G)
        st.upload(arena, 0, enc)
        st.gate_batch(arena, np.full(G, OPS["NAND"], dtype=np.int32), ia, ib, np.full(G, -1, dtype=np.int32),
                      np.arange(nin, nin + G, dtype=np.int32))
        st.sync()
        got = st.download(arena, nin, G)
        arena.free()
        st.destroy()
    finally:
        hip.cleanup()
    assert np.array_equal(client.decrypt_bits(keys80, got), 1 - (bits[ia] & bits[ib]))
    import numpy_tfhe

    numpy_tfhe.check_noise_against_cggi(keys80, got, 1 - (bits[ia] & bits[ib]), rel_tol=0.05)   # noise KAT, 65 536 outputs
    from test_gpu_parity import _check_fullsize_digests

    _check_fullsize_digests("80", got)       # every one of the 65 536 outputs against the oracle's committed digest
    sample = rng.choice(G, size=32, replace=False)
    ref = np.zeros((nin + 32, p.n + 1), dtype=np.uint32)
    ref[:nin] = enc
    oracle80.gate_batch([OPS["NAND"]] * 32, ia[sample], ib[sample], [-1] * 32, list(range(nin, nin + 32)), ref,
                        nthreads=os.cpu_count() or 1)
    assert np.array_equal(got[sample], ref[nin:])


@pytest.mark.parametrize("kernel", ["w32", "lat3", None])
def test_80bit_direct_decomposition(kernel, keys80, oracle80, monkeypatch):
    """IYK_HIP_DECOMP=direct (opt-in, include/iyokan_hip.h: iyk_hip_decomposition_levels): the 80-bit set's 10-bit digits
    as they are, 2 levels instead of 4 virtual ones.  Same ciphertexts as the oracle, word for word, on every rotation
    kernel — 512 NANDs, the adversarial LWE rows and a TRLWE-mode batch; an integer sum reaching p/2 would show here as a
    mismatch (probability <= 2e-17 per gate)."""
    import oracle_lib
    from iyokan_amd import hip

    if kernel is None:
        monkeypatch.delenv("IYK_HIP_ROT_KERNEL", raising=False)
    else:
        monkeypatch.setenv("IYK_HIP_ROT_KERNEL", kernel)
    monkeypatch.delenv("IYK_HIP_NTT", raising=False)
    monkeypatch.setenv("IYK_HIP_DECOMP", "direct")
    hip.initialize(keys80, device_ids=(0,))
    try:
        assert hip.ntt_path() == "fp50" and hip.decomposition_levels() == 2
        st = hip.Stream(0)
        p = keys80.params
        G, nin = 512, 64
        rng = np.random.default_rng(21)
        bits = rng.integers(0, 2, size=nin).astype(np.uint8)
        ia = rng.integers(0, nin, size=G).astype(np.int32)
        ib = rng.integers(0, nin, size=G).astype(np.int32)
        rows = oracle_lib.adversarial_rows(p.n)
        host = np.zeros((nin + len(rows) + G + len(rows), p.n + 1), dtype=np.uint32)
        host[:nin] = client.encrypt_bits(keys80, bits, seed=9)
        host[nin:nin + len(rows)] = rows
        o0 = nin + len(rows)
        ops = [OPS["NAND"]] * G + [OPS["AND"]] * len(rows)
        in0 = list(ia) + list(range(nin, nin + len(rows)))
        in1 = list(ib) + list(range(nin, nin + len(rows)))
        out = list(range(o0, o0 + G + len(rows)))
        arena = hip.Arena(host.shape[0])
        st.upload(arena, 0, host)
        st.gate_batch(arena, ops, in0, in1, [-1] * len(ops), out)
        st.sync()
        got = st.download(arena, 0, host.shape[0])
        arena.free()
        st.destroy()
    finally:
        hip.cleanup()
    ref = host.copy()
    oracle80.gate_batch(ops, in0, in1, [-1] * len(ops), out, ref, nthreads=os.cpu_count() or 1)
    assert np.array_equal(got, ref)
    assert np.array_equal(client.decrypt_bits(keys80, got[o0:o0 + G]), 1 - (bits[ia] & bits[ib]))


def test_80bit_direct_full_size_and_rejections(keys80, keys128, oracle80, monkeypatch):
    """65 536 NANDs with the direct decomposition: all decrypt, the noise statistics are the scheme's, a 32-gate sample
    equals the oracle; the option is refused for the 128-bit set, together with the integer path, and when misspelt."""
    from iyokan_amd import hip

    monkeypatch.delenv("IYK_HIP_ROT_KERNEL", raising=False)
    monkeypatch.delenv("IYK_HIP_NTT", raising=False)
    monkeypatch.setenv("IYK_HIP_DECOMP", "direct")
    with pytest.raises(hip.IykHipError, match="IYK_HIP_DECOMP=direct applies"):
        hip.initialize(keys128, device_ids=(0,))
    monkeypatch.setenv("IYK_HIP_NTT", "goldilocks")
    with pytest.raises(hip.IykHipError, match="IYK_HIP_DECOMP=direct applies"):
        hip.initialize(keys80, device_ids=(0,))
    monkeypatch.delenv("IYK_HIP_NTT", raising=False)
    monkeypatch.setenv("IYK_HIP_DECOMP", "fast")
    with pytest.raises(hip.IykHipError, match="must be 'split' or 'direct'"):
        hip.initialize(keys80, device_ids=(0,))
    monkeypatch.setenv("IYK_HIP_DECOMP", "direct")
    hip.initialize(keys80, device_ids=(0,))
    try:
        st = hip.Stream(0)
        p = keys80.params
        G, nin = 65536, 2048
        rng = np.random.default_rng(13)
        bits = rng.integers(0, 2, size=nin).astype(np.uint8)
        ia = rng.integers(0, nin, size=G).astype(np.int32)
        ib = rng.integers(0, nin, size=G).astype(np.int32)
        enc = client.encrypt_bits(keys80, bits, seed=8)
        arena = hip.Arena(nin + G)
        st.upload(arena, 0, enc)
        st.gate_batch(arena, np.full(G, OPS["NAND"], dtype=np.int32), ia, ib, np.full(G, -1, dtype=np.int32),
                      np.arange(nin, nin + G, dtype=np.int32))
        st.sync()
        got = st.download(arena, nin, G)
        arena.free()
        st.destroy()
    finally:
        hip.cleanup()
    assert np.array_equal(client.decrypt_bits(keys80, got), 1 - (bits[ia] & bits[ib]))
    import numpy_tfhe

    numpy_tfhe.check_noise_against_cggi(keys80, got, 1 - (bits[ia] & bits[ib]), rel_tol=0.05)
    sample = rng.choice(G, size=32, replace=False)
    ref = np.zeros((nin + 32, p.n + 1), dtype=np.uint32)
    ref[:nin] = enc
    oracle80.gate_batch([OPS["NAND"]] * 32, ia[sample], ib[sample], [-1] * 32, list(range(nin, nin + 32)), ref,
                        nthreads=os.cpu_count() or 1)
    assert np.array_equal(got[sample], ref[nin:])
