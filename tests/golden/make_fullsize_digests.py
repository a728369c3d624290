"""Generates tests/golden/fullsize_nand_{128,80}.bin with the CPU oracle (run in the build container, ~5 min per set on 8 cores):
    python tests/golden/make_fullsize_digests.py [128|80 ...]
BASELINE configs #2 and #5 at FULL size: 65 536 NAND gates on fresh encryptions (key seed 1; data seeds below, the ones
tests/test_gpu_parity.py and tests/test_gpu_80bit.py use).  Fixture = data only: per output TLWE the first 8 bytes of its
sha256, 65 536 x 8 bytes per set, plus a json with the seeds and the sha256 over all digests.  The GPU tests recompute the
digest of EVERY output and compare (VERDICT r03 next #3a: word-level parity at full size, not a 64-gate sample)."""
import hashlib
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

import oracle_lib  # noqa: E402
from iyokan_amd import client  # noqa: E402
from iyokan_amd.params import OPS, params_128bit, params_80bit  # noqa: E402

SETS = {   # name: (params, rng seed of bits / operand choice, number of input ciphertexts, encryption seed)
    "128": (params_128bit, 2, 4096, 2),
    "80": (params_80bit, 12, 2048, 6),
    # round 5: SURVEY 8(d) config 2's own input shape — 2 x 65 536 FRESH encryptions, gate g =
    # NAND(in[g], in[G + g]), key seed 1, data seed 2 — which is exactly what bench.py times on rank 0 (bits from
    # default_rng(1000), encryptions with seed 2): `bench.py` compares the words of its TIMED step with these digests
    "fresh_128": (params_128bit, 1000, 2 * 65536, 2),
    "fresh_80": (params_80bit, 1000, 2 * 65536, 2),
}
G = 65536


def workload(name):
    """the exact inputs of the full-size GPU tests / of bench.py: (keys, bits, ia, ib, enc)"""
    mk, rs, nin, es = SETS[name]
    keys = client.keygen(mk(), seed=1)
    rng = np.random.default_rng(rs)
    bits = rng.integers(0, 2, size=nin).astype(np.uint8)
    if name.startswith("fresh"):
        ia = np.arange(G, dtype=np.int32)
        ib = ia + G
    else:
        ia = rng.integers(0, nin, size=G).astype(np.int32)
        ib = rng.integers(0, nin, size=G).astype(np.int32)
    return keys, bits, ia, ib, client.encrypt_bits(keys, bits, seed=es)


def digests(rows):
    """first 8 bytes of sha256 of every row: (len(rows), 8) uint8"""
    return np.frombuffer(b"".join(hashlib.sha256(r.tobytes()).digest()[:8] for r in rows), dtype=np.uint8).reshape(-1, 8)


def main():
    for name in (sys.argv[1:] or list(SETS)):
        keys, bits, ia, ib, enc = workload(name)
        p = keys.params
        nin = len(bits)
        orc = oracle_lib.Oracle(keys)
        mode = "fp" if orc.has_fp() else "goldilocks"       # both are exact restatements (tests pin them equal); fp is 4x faster
        out = np.zeros((G, 8), dtype=np.uint8)
        t0 = time.time()
        CH = 4096
        for lo in range(0, G, CH):
            if name.startswith("fresh"):     # the chunk's own operands only: [in0 of the chunk | in1 of the chunk | outputs]
                arena = np.zeros((3 * CH, p.n + 1), dtype=np.uint32)
                arena[:CH] = enc[ia[lo:lo + CH]]
                arena[CH:2 * CH] = enc[ib[lo:lo + CH]]
                a_idx, b_idx, first_out = np.arange(CH, dtype=np.int32), np.arange(CH, 2 * CH, dtype=np.int32), 2 * CH
            else:
                arena = np.zeros((nin + CH, p.n + 1), dtype=np.uint32)
                arena[:nin] = enc
                a_idx, b_idx, first_out = ia[lo:lo + CH], ib[lo:lo + CH], nin
            orc.gate_batch([OPS["NAND"]] * CH, a_idx, b_idx, [-1] * CH, list(range(first_out, first_out + CH)), arena,
                           nthreads=os.cpu_count() or 1, mode=mode)
            assert np.array_equal(client.decrypt_bits(keys, arena[first_out:]), 1 - (bits[ia[lo:lo + CH]] & bits[ib[lo:lo + CH]]))
            out[lo:lo + CH] = digests(arena[first_out:])
            print(name, lo + CH, f"{time.time() - t0:.0f}s", flush=True)
        orc.close()
        out.tofile(os.path.join(HERE, f"fullsize_nand_{name}.bin"))
        meta = {"generator": "tests/golden/make_fullsize_digests.py (oracle/, mode " + mode + ")", "params": p.as_dict(),
                "gates": G, "op": "NAND", "key_seed": 1, "rng_seed": SETS[name][1], "inputs": nin, "enc_seed": SETS[name][3],
                "digest": "first 8 bytes of sha256(output TLWE words, little endian)",
                "sha256_of_digests": hashlib.sha256(out.tobytes()).hexdigest()}
        json.dump(meta, open(os.path.join(HERE, f"fullsize_nand_{name}.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
