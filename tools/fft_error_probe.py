"""Largest |z - rint(z)| the FFT rotation kernels produce (IYK_HIP_DEBUG=1, iyk_hip_fft_round_error): fresh encryptions under a
real key, and the adversarial rows under a 'key' whose every 16-bit half is -2^15.  python tools/fft_error_probe.py (GPU)."""
import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
os.environ["IYK_HIP_NTT"] = "fft"; os.environ["IYK_HIP_DEBUG"] = "1"
from iyokan_amd import client, hip
from iyokan_amd.params import OPS, params_128bit, params_80bit
import oracle_lib
def run(keys, kernel, rows):
    os.environ["IYK_HIP_ROT_KERNEL"] = kernel
    p = keys.params
    n = len(rows)
    host = np.zeros((2 * n, p.n + 1), dtype=np.uint32); host[:n] = rows
    hip.initialize(keys, device_ids=(0,))
    st = hip.Stream(0); ar = hip.Arena(2 * n); st.upload(ar, 0, host)
    idx = np.arange(n, dtype=np.int32)
    st.gate_batch(ar, np.full(n, OPS["OR"], dtype=np.int32), idx, idx, np.full(n, -1, dtype=np.int32), idx + n); st.sync()
    e = hip.fft_round_error(0); ar.free(); st.destroy(); hip.cleanup(); return e
for name, P in (("128", params_128bit()), ("80", params_80bit())):
    keys = client.keygen(P, seed=7)
    p = keys.params
    fresh = client.encrypt_bits(keys, np.random.default_rng(1).integers(0, 2, 64).astype(np.uint8), seed=3)
    adv = oracle_lib.adversarial_rows(p.n)
    bad = client.KeySet(p, keys.s0, keys.s1, np.full(p.bk_words, 0x80008000, dtype=np.uint32), keys.ksk)
    for kernel in ("fft", "latfft"):
        e1 = run(keys, kernel, fresh); e2 = run(bad, kernel, adv)
        print(name, kernel, "real keys 2^%.1f" % math.log2(e1), "worst-case keys 2^%.1f" % math.log2(e2))
