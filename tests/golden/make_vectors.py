"""Generates tests/golden/gate_vectors_128.json with the CPU oracle (run in the build container):
    python tests/golden/make_vectors.py
Fixture = data only: seeds, plaintext bits, gate list, sha256 of each expected output TLWE."""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

import oracle_lib  # noqa: E402
from iyokan_amd import client  # noqa: E402
from iyokan_amd.params import OPS, params_128bit  # noqa: E402

p = params_128bit()
keys = client.keygen(p, seed=1)
bits = [0, 1, 1, 0, 1, 0, 0, 1]
enc = client.encrypt_bits(keys, bits, seed=2)
kinds = ["AND", "NAND", "ANDNOT", "OR", "NOR", "ORNOT", "XOR", "XNOR", "MUX", "MUX", "NOT", "COPY", "CONSTONE", "CONSTZERO"]
gates = []
for g, kind in enumerate(kinds):
    a, b, s = g % 8, (3 * g + 1) % 8, (5 * g + 2) % 8
    nin = 3 if kind == "MUX" else 2 if OPS[kind] < 8 else 1 if kind in ("NOT", "COPY") else 0
    gates.append({"op": kind, "in0": a if nin >= 1 else -1, "in1": b if nin >= 2 else -1,
                  "in2": s if nin >= 3 else -1, "out": len(bits) + g})
arena = np.zeros((len(bits) + len(gates), p.n + 1), dtype=np.uint32)
arena[: len(bits)] = enc
orc = oracle_lib.Oracle(keys)
orc.gate_batch([OPS[x["op"]] for x in gates], [x["in0"] for x in gates], [x["in1"] for x in gates],
               [x["in2"] for x in gates], [x["out"] for x in gates], arena, nthreads=os.cpu_count() or 1)
for x in gates:
    x["sha256"] = hashlib.sha256(arena[x["out"]].tobytes()).hexdigest()
    x["bit"] = int(client.decrypt_bits(keys, arena[x["out"]])[0])
doc = {"generator": "tests/golden/make_vectors.py (oracle/tfhe_oracle.c, NTT path)", "params": p.as_dict(),
       "key_seed": 1, "data_seed": 2, "input_bits": bits,
       "inputs_sha256": hashlib.sha256(enc.tobytes()).hexdigest(), "gates": gates}
json.dump(doc, open(os.path.join(HERE, "gate_vectors_128.json"), "w"), indent=1)
print("wrote", len(gates), "vectors")
