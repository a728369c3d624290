#!/usr/bin/env python3
"""bench.py — TFHE gate bootstraps/sec on MI355X (BASELINE.json metric).

Workload (BASELINE.json configs[1], SURVEY.md §8d "Config 2"): a flat DAG of independent
HomNAND gates at the 128-bit parameter set, inputs are FRESH encryptions (never trivial
ciphertexts, which skip every CMUX).  One "step" = one pass of the hot path over the whole
batch: iyk_hip_gate_batch -> {blind_rotate kernel, keyswitch kernel}, inputs and keys already
resident in HBM.  With N > 1 (one process per GPU, launched by torch.distributed.run) the batch
is sharded by replication of the per-GPU work: every rank processes its own `--gates` gates,
there is no data-path collective for a flat DAG ("scaling": "weak"); the key material is
generated on rank 0 and broadcast once over RCCL.

Prints ONE JSON line on rank 0 (see README of the contract in DESIGN.md §Measurement).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_BYTES_PER_S = 8.0e12  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--gates", type=int, default=65536, help="gates per step per GPU")
    ap.add_argument("--params", default="128bit", choices=["128bit", "80bit"])
    ap.add_argument("--op", default="NAND", choices=["AND", "NAND", "ANDNOT", "OR", "NOR", "ORNOT", "XOR", "XNOR"],
                    help="binary gate of the flat batch (BASELINE config #2 is NAND)")
    ap.add_argument("--cpu-sample", type=int, default=-1,
                    help="gates timed on the CPU oracle for cpu_baseline (-1: sized for ~15 s, 0: skip)")
    ap.add_argument("--traffic-bytes", type=float, default=None,
                    help="measured HBM bytes per blind_rotate launch from a separate rocprofv3 --pmc pass")
    return ap.parse_args()


def traffic_bytes(args, gates):
    """HBM bytes per blind_rotate launch measured by a separate rocprofv3 --pmc pass (committed under
    profiles/), or None when no measurement matches this workload."""
    if args.traffic_bytes is not None:
        return args.traffic_bytes
    path = os.path.join(ROOT, "profiles", "r01_traffic.json")
    try:
        t = json.load(open(path))
        w = t["workload"]
        if w["gates_per_launch"] == gates and w["params"] == args.params and w["op"] == args.op:
            return t["traffic_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        pass
    return None


def baseline_metric():
    """The headline metric's name, verbatim from BASELINE.json when it is at hand."""
    try:
        with open(os.path.join(ROOT, "BASELINE.json")) as f:
            return json.load(f)["metric"]
    except (OSError, KeyError, ValueError):
        return "TFHE gate bootstraps/sec (128-bit params)"


def broadcast_keys(keys, dist, device, rank):
    """Key material is generated on rank 0 only and broadcast once (RCCL over xGMI on the GPU box; the
    reference instead replicates keys by a per-device cudaMemcpy loop inside cufhe::Initialize).  Works
    on any torch device so the same code is exercised by the gloo CPU test (tests/test_bench_dist.py)."""
    import torch

    for name in ("s0", "s1", "bk", "ksk"):
        host = getattr(keys, name)
        t = torch.from_numpy(host.view(np.int32)).to(device)
        dist.broadcast(t, src=0)
        if rank != 0:
            setattr(keys, name, t.cpu().numpy().view(np.uint32).copy())
        del t
    return keys


def empty_keys(params):
    from iyokan_amd import client

    return client.KeySet(params, np.zeros(params.n, np.uint32), np.zeros(params.N, np.uint32),
                         np.zeros(params.bk_words, np.uint32), np.zeros(params.ksk_words, np.uint32))


def cpu_baseline(keys, params, op_code, sample, data_seed, budget_s=15.0):
    """Oracle (kind 'port') timed on this host's cores on a bounded sample of the same workload.

    A first probe chunk (one gate per thread) sizes the sample so the whole leg takes ~budget_s.
    """
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    from iyokan_amd import client

    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    threads = max(1, min(cores, 64))
    orc = oracle_lib.Oracle(keys)

    def run(count, seed):
        rng = np.random.default_rng(seed)
        bits = rng.integers(0, 2, size=2 * count).astype(np.uint8)
        arena = np.zeros((3 * count, params.n + 1), dtype=np.uint32)
        arena[: 2 * count] = client.encrypt_bits(keys, bits, seed=seed)
        idx = np.arange(count, dtype=np.int32)
        t0 = time.perf_counter()
        orc.gate_batch(np.full(count, op_code, dtype=np.int32), idx, idx + count, np.full(count, -1, dtype=np.int32),
                       idx + 2 * count, arena, nthreads=threads)
        dt = time.perf_counter() - t0
        dec = client.decrypt_bits(keys, arena[2 * count:])
        assert np.array_equal(dec, 1 - (bits[:count] & bits[count:])), "oracle decrypt mismatch"
        return dt

    probe_dt = run(threads, data_seed)
    if sample < 0:
        sample = int(max(threads, min(64 * threads, threads * (budget_s - probe_dt) / max(probe_dt, 1e-3))))
        sample -= sample % threads
        sample = max(sample, threads)
    dt = run(sample, data_seed + 1)
    orc.close()
    return {"value": sample / dt, "unit": "gates/s", "cores": threads, "kind": "port",
            "sample": f"{sample} NAND gates of the same workload in {dt:.1f} s (own exact-NTT CPU restatement "
                      f"oracle/tfhe_oracle.c, OpenMP over gates, {threads} threads of {cores} visible cores); not TFHEpp"}


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist

    from iyokan_amd import client, hip
    from iyokan_amd.params import OPS, params_by_name

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}; using WORLD_SIZE", file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback for the product path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=dev)

    params = params_by_name(args.params)
    op_code = OPS[args.op]
    G = args.gates

    # ---- keys: rank 0 generates, RCCL broadcast (north_star: "bootstrapping key broadcast once") ----
    keys = client.keygen(params, seed=1) if rank == 0 else empty_keys(params)
    if world > 1:
        keys = broadcast_keys(keys, dist, dev, rank)
    hip.initialize(keys, device_ids=(local_rank,))

    # ---- synthetic inputs: 2G fresh encryptions per rank (distinct data seed per rank) ----
    rng = np.random.default_rng(1000 + rank)
    bits = rng.integers(0, 2, size=2 * G).astype(np.uint8)
    enc = client.encrypt_bits(keys, bits, seed=2 + rank)
    arena_t = torch.zeros((3 * G, params.n + 1), dtype=torch.int32, device=dev)
    arena_t[: 2 * G].copy_(torch.from_numpy(enc.view(np.int32)))
    del enc
    tstream = torch.cuda.Stream(device=dev)
    torch.cuda.synchronize()
    st = hip.Stream(0, hip_stream=tstream.cuda_stream)
    arena = hip.Arena.from_torch(arena_t)
    idx = np.arange(G, dtype=np.int32)
    ops = np.full(G, op_code, dtype=np.int32)
    in0, in1, in2, out = idx, idx + G, np.full(G, -1, dtype=np.int32), idx + 2 * G

    def step():
        st.gate_batch(arena, ops, in0, in1, in2, out)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    st.timing_log_begin()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    nb, br_ms, ks_ms = st.timing_log_end()
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- correctness of what was just timed: decrypt every output of the last step ----
    got = arena_t[2 * G:].cpu().numpy().view(np.uint32)
    from iyokan_amd.params import PLAIN

    want = np.array([PLAIN[args.op](int(a), int(b)) for a, b in zip(bits[:G], bits[G:])], dtype=np.uint8)
    decrypt_ok = bool(np.array_equal(client.decrypt_bits(keys, got), want))

    if rank == 0:
        fp_path = hip.ntt_path() == "fp50"
        gates_total = G * args.steps * world
        value = gates_total / elapsed
        b_gate = params.gate_algorithmic_bytes(rotations=1, inputs=2)
        # dominant kernel = blind_rotate: its share of B_gate is the BK stream + its own I/O
        br_bytes_per_gate = params.n * params.trgsw_rows * (params.k + 1) * params.N * 8 \
            + 2 * (params.n + 1) * 4 + (params.N + 1) * 4
        br_avg_s = (br_ms / max(nb, 1)) * 1e-3
        achieved = br_bytes_per_gate * G / br_avg_s if br_avg_s > 0 else 0.0
        line = {
            "metric": baseline_metric() if args.params == "128bit" else "TFHE gate bootstraps/sec (80-bit params)",
            "value": value,
            "unit": "gates/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": ("u32 torus; NTT in f64 mod p = 3*2^48+1097729 (exact FMA arithmetic)" if fp_path
                      else "u32 torus; NTT in u64 mod 2^64-2^32+1"),
            "data": "synthetic",
            "config": {
                "workload": f"{G} independent Hom{args.op} gates per GPU per step (flat DAG), {args.params} params, "
                            "fresh encryptions, keys+ciphertexts resident in HBM",
                "params": {k: v for k, v in params.as_dict().items() if k not in ("alpha0", "alpha1")},
                "gates_per_step_per_gpu": G,
                "parallelism": f"frontier sharded over {world} GPU(s), no data-path collective (flat DAG)",
                "decrypt_check": decrypt_ok,
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "blind_rotate_fp_kernel" if fp_path else "blind_rotate_kernel",
                "achieved": achieved / 1e9,
                "peak": HBM_PEAK_BYTES_PER_S / 1e9,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_BYTES_PER_S,
                "traffic": traffic_bytes(args, G),
                "algorithmic_bytes_per_launch": br_bytes_per_gate * G,
                "avg_launch_ms": br_avg_s * 1e3,
                "keyswitch_avg_launch_ms": ks_ms / max(nb, 1),
                "gate_bytes": b_gate,
                "gate_frac": value / world * b_gate / HBM_PEAK_BYTES_PER_S,
            },
        }
        if world == 1 and args.cpu_sample != 0:
            line["cpu_baseline"] = cpu_baseline(keys, params, op_code, args.cpu_sample, data_seed=99)
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)

    st.destroy()
    del arena, arena_t
    hip.cleanup()
    if world > 1:
        dist.destroy_process_group()
    if not decrypt_ok:
        raise SystemExit("decrypt check failed")


if __name__ == "__main__":
    main()
