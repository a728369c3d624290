"""The device-facing half of the upstream-flavour plugin, executed on the GPU and compared with the oracle word for word.

`integration/upstream/iyokan_hip_device.hpp` (HIPStream, HIPFrontierBatch, pinned staging) is the code the plugin's tasks and
workers call; upstream's engine around it can only be type-checked here (tests/test_upstream_flavour.py), but this half needs
nothing of upstream's and runs as it is through `integration/upstream/hip_flavour_harness.cpp`:

* `--batch`: every gate of a frontier through ONE HIPFrontierBatch — HIPBatchWorker::update() for a flat frontier;
* `--per-gate W`: the reference's harness shape, W one-gate workers (stream + result ciphertext) polled round-robin from one
  host thread with `iyk_hip_gate_host` / `iyk_hip_stream_query` — `processAllGates(net, 240)` of
  /root/reference/src/test0.cpp:696-700 and the frontend's 800 workers (/root/reference/src/iyokan_cufhe.cpp:259)
  (VERDICT r04 "What's missing" #3 / "Next round" #4).

Every gate kind, fresh encryptions, all outputs equal to the oracle's; the rates go to the test log (and, from
tools/gpu_round.sh, to profiles/r05_per_gate.txt).
"""
import json
import os
import subprocess

import numpy as np
import pytest

from iyokan_amd import client
from iyokan_amd.params import OPS, params_80bit, params_128bit

import oracle_lib

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KINDS = ["AND", "NAND", "ANDNOT", "OR", "NOR", "ORNOT", "XOR", "XNOR", "MUX", "NOT"]
ARITY = {"MUX": 3, "NOT": 1}


def _harness(bits):
    path = os.path.join(ROOT, "integration", "upstream", f"hip_flavour_harness_{bits}")
    if not os.path.exists(path):
        raise RuntimeError(f"{path} is missing — build it with __graft_entry__.build()")
    return path


def _case(tmp_path, params, count, seed):
    keys = client.keygen(params, seed=seed)
    rng = np.random.default_rng(seed + 100)
    names = [KINDS[i % len(KINDS)] for i in range(count)]
    rng.shuffle(names)
    ops = np.array([OPS[k] for k in names], dtype=np.int32)
    bits = rng.integers(0, 2, size=3 * count).astype(np.uint8)
    operands = client.encrypt_bits(keys, bits, seed=seed + 200)          # [3 count][n + 1]: gate g reads rows 3g .. 3g + 2
    files = {}
    for name, arr in (("bk", keys.bk), ("ksk", keys.ksk), ("ops", ops), ("operands", operands)):
        files[name] = str(tmp_path / f"{name}.bin")
        np.ascontiguousarray(arr).tofile(files[name])
    # the oracle on the same operands: one arena, gate g writes slot 3 count + g
    n1 = params.n + 1
    arena = np.zeros((4 * count, n1), dtype=np.uint32)
    arena[: 3 * count] = operands
    in0 = [3 * g for g in range(count)]
    in1 = [3 * g + 1 if ARITY.get(names[g], 2) >= 2 else -1 for g in range(count)]
    in2 = [3 * g + 2 if ARITY.get(names[g], 2) >= 3 else -1 for g in range(count)]
    out = [3 * count + g for g in range(count)]
    orc = oracle_lib.Oracle(keys)
    orc.gate_batch(list(ops), in0, in1, in2, out, arena, nthreads=os.cpu_count() or 1)
    return files, arena[3 * count:], params


def _run(files, params, tmp_path, mode, env=None):
    bits = 128 if params.n == 636 else 80
    out = str(tmp_path / ("out_" + "_".join(mode).replace("-", "") + ".bin"))
    r = subprocess.run([_harness(bits), str(params.n), files["bk"], files["ksk"], files["ops"], files["operands"], out] + mode,
                       capture_output=True, text=True, timeout=600, env=dict(os.environ, **(env or {})))
    assert r.returncode == 0, r.stderr[-2000:]
    stats = json.loads(r.stdout.strip().splitlines()[-1])
    got = np.fromfile(out, dtype=np.uint32).reshape(-1, params.n + 1)
    return got, stats


@pytest.mark.parametrize("make_params", [params_128bit, params_80bit], ids=["128bit", "80bit"])
def test_frontier_batch_equals_oracle(tmp_path, make_params):
    files, want, params = _case(tmp_path, make_params(), count=300, seed=11)
    got, stats = _run(files, params, tmp_path, ["--batch"])
    assert np.array_equal(got, want)
    print("upstream flavour, batch:", json.dumps(stats))


@pytest.mark.parametrize("workers", [240, 800])
def test_per_gate_workers_equal_oracle(tmp_path, workers):
    """The reference's one-gate-per-stream shape at test0's and the frontend's worker counts; 1 000 gates, so that every worker
    is used and most of the 240 are used several times (a stream's pinned mirror is recycled: iyk_hip_gate_host)."""
    files, want, params = _case(tmp_path, params_128bit(), count=1000, seed=12)
    got, stats = _run(files, params, tmp_path, ["--per-gate", str(workers)])
    assert np.array_equal(got, want)
    assert stats["gates"] == 1000
    print(f"upstream flavour, per gate, {workers} workers:", json.dumps(stats))


@pytest.mark.parametrize("env, label", [({"IYK_HIP_COALESCE_MAX": "7"}, "flush every 7 parked gates"),
                                        ({"IYK_HIP_COALESCE": "0"}, "one launch sequence per gate on the worker's stream")],
                         ids=["max7", "off"])
def test_per_gate_workers_other_coalescing_settings(tmp_path, env, label):
    """The same one-gate workers with the coalescer forced to flush by COUNT (7 gates: dozens of small batches, both buffer sides in
    use all the time) and with coalescing OFF (the reference's literal launch pattern): the same words as the oracle either way."""
    files, want, params = _case(tmp_path, params_128bit(), count=150, seed=13)
    got, stats = _run(files, params, tmp_path, ["--per-gate", "48"], env=env)
    assert np.array_equal(got, want)
    print(f"upstream flavour, per gate, 48 workers, {label}:", json.dumps(stats))


def test_per_gate_workers_80bit(tmp_path):
    files, want, params = _case(tmp_path, params_80bit(), count=400, seed=14)
    got, stats = _run(files, params, tmp_path, ["--per-gate", "240"])
    assert np.array_equal(got, want)
    print("upstream flavour, per gate, 240 workers, 80-bit set:", json.dumps(stats))
