"""GPU: the 64-bit integer (Goldilocks) field path forced for the 128-bit set (IYK_HIP_NTT=goldilocks),
so both exact-arithmetic paths are checked bit for bit against the oracle on the same inputs."""
import os

import numpy as np
import pytest

from iyokan_amd import client
from iyokan_amd.params import OPS

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("path", ["goldilocks", "fp50", "fft"])
def test_both_field_paths_bit_exact(path, keys128, oracle128, monkeypatch):
    from iyokan_amd import hip

    old = os.environ.get("IYK_HIP_NTT")
    os.environ["IYK_HIP_NTT"] = {"fft": "fft", "fp50": "fp", "goldilocks": "goldilocks"}[path]
    if path == "fft":
        monkeypatch.setenv("IYK_HIP_ROT_KERNEL", "fft")    # 48 rotations: the size-based dispatch would take the narrow-frontier kernel
    else:
        monkeypatch.delenv("IYK_HIP_ROT_KERNEL", raising=False)
    try:
        hip.initialize(keys128, device_ids=(0,))
        assert hip.ntt_path() == path
        st = hip.Stream(0)
        p = keys128.params
        rng = np.random.default_rng(17)
        nin, ng = 32, 40
        bits = rng.integers(0, 2, size=nin).astype(np.uint8)
        kinds = rng.choice(["NAND", "XOR", "MUX", "ORNOT", "AND"], size=ng)
        in0, in1, in2 = (rng.integers(0, nin, size=ng).astype(np.int32) for _ in range(3))
        ops = np.array([OPS[k] for k in kinds], dtype=np.int32)
        in2 = np.where(ops == OPS["MUX"], in2, -1).astype(np.int32)
        out = np.arange(nin, nin + ng, dtype=np.int32)
        host = np.zeros((nin + ng, p.n + 1), dtype=np.uint32)
        host[:nin] = client.encrypt_bits(keys128, bits, seed=23)
        arena = hip.Arena(nin + ng)
        st.upload(arena, 0, host)
        st.gate_batch(arena, ops, in0, in1, in2, out)
        st.sync()
        got = st.download(arena, 0, nin + ng)
        arena.free()
        st.destroy()
        hip.cleanup()
    finally:
        if old is None:
            os.environ.pop("IYK_HIP_NTT", None)
        else:
            os.environ["IYK_HIP_NTT"] = old
    ref = host.copy()
    oracle128.gate_batch(ops, in0, in1, in2, out, ref, nthreads=os.cpu_count() or 1)
    assert np.array_equal(got, ref)
