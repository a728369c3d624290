"""The Z_p-field form of the bootstrapping key is built lazily on the FFT path (round 6)."""
import os

import numpy as np
import pytest

from iyokan_amd import client
from iyokan_amd.params import OPS

pytestmark = pytest.mark.gpu


def test_field_key_is_built_on_first_use_only(keys128, oracle128, monkeypatch):
    """Round 6: on the default (FFT) path iyk_hip_init no longer builds or keeps the Z_p-field form of the
    bootstrapping key (62.5 MB per GPU at the 128-bit set, read only by the cross-check kernels).  Resident keys: spectra + padded
    KSK + tables right after init; + the field key after the first batch that forces a field kernel — whose results still equal
    the oracle word for word, and the default dispatch's."""
    from iyokan_amd import hip

    for var in ("IYK_HIP_ROT_KERNEL", "IYK_HIP_NTT", "IYK_HIP_LATENCY_KERNEL"):
        monkeypatch.delenv(var, raising=False)
    p = keys128.params
    hip.initialize(keys128, device_ids=(0,))
    try:
        assert hip.ntt_path() == "fft"
        bk_words = p.n * 2 * p.l * 2 * p.N
        spectra = bk_words * 16                                  # two 16-bit halves x 8 bytes per word
        field = bk_words * 8
        ksk = p.N * p.t * 3 * ((p.n + 1 + 3) & ~3) * 4
        before = hip.resident_key_bytes()
        assert before == spectra + ksk + 2 * p.N * 8, (before, spectra, ksk)
        assert 175e6 < before < 185e6                            # 179.8 MB (round 5: 242.3 MB)
        st = hip.Stream(0)
        rng = np.random.default_rng(606)
        nin, ng = 16, 20
        bits = rng.integers(0, 2, size=nin).astype(np.uint8)
        ops = rng.choice([OPS["NAND"], OPS["XOR"], OPS["MUX"]], size=ng).astype(np.int32)
        in0, in1, in2 = (rng.integers(0, nin, size=ng).astype(np.int32) for _ in range(3))
        in2 = np.where(ops == OPS["MUX"], in2, -1).astype(np.int32)
        out = np.arange(nin, nin + ng, dtype=np.int32)
        host = np.zeros((nin + ng, p.n + 1), dtype=np.uint32)
        host[:nin] = client.encrypt_bits(keys128, bits, seed=607)

        def run():
            arena = hip.Arena(nin + ng)
            st.upload(arena, 0, host)
            st.gate_batch(arena, ops, in0, in1, in2, out)
            st.sync()
            got = st.download(arena, 0, nin + ng)
            arena.free()
            return got

        default = run()
        assert hip.resident_key_bytes() == before                # the default dispatch never asks for the field key
        results = {}
        for kernel in ("lat3", "w32"):
            monkeypatch.setenv("IYK_HIP_ROT_KERNEL", kernel)
            results[kernel] = run()
            assert hip.resident_key_bytes() == before + field    # built once, by the first of the two
        monkeypatch.delenv("IYK_HIP_ROT_KERNEL")
        ref = host.copy()
        oracle128.gate_batch(ops, in0, in1, in2, out, ref, nthreads=os.cpu_count() or 1)
        assert np.array_equal(default, ref)
        for kernel, got in results.items():
            assert np.array_equal(got, ref), kernel
        st.destroy()
    finally:
        hip.cleanup()
