#!/usr/bin/env python3
"""bench.py — TFHE gate bootstraps/sec on MI355X (BASELINE.json metric).

Workload (BASELINE.json configs[1], SURVEY.md §8d "Config 2"): a flat DAG of 65 536 independent
HomNAND gates at the 128-bit parameter set, inputs are FRESH encryptions (never trivial
ciphertexts, which skip every CMUX).  One "step" = one pass of the hot path over the whole
batch: iyk_hip_gate_batch -> {blind_rotate kernel, keyswitch kernel}, inputs and keys already
resident in HBM.  With N > 1 — one process per GPU: either the driver launches the ranks with
torch.distributed.run, or `python bench.py --gpus N` alone re-executes itself under it (launcher(): 127.0.0.1
rendezvous on a free port) — the SAME 65 536-gate batch is sharded over the ranks (8 192 gates per GPU at N = 8), with no
data-path collective for a flat DAG: "scaling": "strong".  The weak-scaling figure (every rank its own
65 536 gates) is measured right after and reported as the extra field "weak".  The key material is
generated on rank 0 and broadcast once over RCCL.

The `roofline` object keeps the contract's HBM figure (algorithmic bytes / kernel time / 8 TB/s) and says
what the counters say beside it: the kernel is bound by FP64 VALU issue, `roofline.valu` prices the measured
instruction count (SQ_INSTS_VALU pass committed under profiles/) against 4 cycles per wave-instruction per
SIMD, and `traffic_over_algorithmic` shows the key stream is served by L2.

A run never reports fewer GPUs than it was asked for: `--gpus N` with fewer than N visible devices, or with a
WORLD_SIZE that is not N, exits non-zero before anything is timed (the reference takes --num-gpu the same way:
/root/reference/src/main.cpp:147-148 -> iyokan_cufhe.cpp:530-536).

Prints ONE JSON line on rank 0 (see README of the contract in DESIGN.md §Measurement).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_BYTES_PER_S = 8.0e12  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec
# VALU issue ceiling (same guide): 256 CUs x 4 SIMDs; a wave64 FP64 (or 32-bit integer) instruction occupies a
# SIMD's 16 lanes for 4 cycles; peak engine clock 2.4 GHz -> 1.667 ns per wave-instruction per SIMD
N_SIMDS = 256 * 4
VALU_PEAK_NS = 4.0 / 2.4


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--gates", type=int, default=65536, help="gates per step, whole job (sharded over the GPUs)")
    ap.add_argument("--no-weak", action="store_true", help="skip the extra weak-scaling measurement at N > 1")
    ap.add_argument("--params", default="128bit", choices=["128bit", "80bit"])
    ap.add_argument("--decomp", default=None, choices=["split", "direct"],
                    help="80-bit set: gadget decomposition on the FP64 path (IYK_HIP_DECOMP; default split = exact "
                         "unconditionally, direct = 10-bit digits as they are, include/iyokan_hip.h)")
    ap.add_argument("--op", default="NAND", choices=["AND", "NAND", "ANDNOT", "OR", "NOR", "ORNOT", "XOR", "XNOR"],
                    help="binary gate of the flat batch (BASELINE config #2 is NAND)")
    ap.add_argument("--cpu-sample", type=int, default=-1,
                    help="gates timed on the CPU oracle for cpu_baseline (-1: sized for ~15 s, 0: skip)")
    ap.add_argument("--spawn", action="store_true",
                    help="re-execute under torch.distributed.run even for --gpus 1 (exercises the RCCL path on one GPU)")
    ap.add_argument("--traffic-bytes", type=float, default=None,
                    help="measured HBM bytes per blind_rotate launch from a separate rocprofv3 --pmc pass")
    return ap.parse_args()


COUNTER_FILES = ("r06_counters.json", "r06_counters_80bit.json", "r05_counters.json", "r05_counters_80bit.json", "r04_counters.json", "r04_counters_80bit.json", "r03_counters.json", "r03_counters_80bit.json", "r03_counters_80bit_direct.json", "r02_counters.json",
                 "r01_traffic.json")  # newest first
DEFAULT_LEVELS = {"128bit": 3, "80bit": 4}   # (r04 files always carry the field: the FFT path runs the 80-bit set at 2)   # iyk_hip_decomposition_levels of counter files older than the field


def counters(args, gates, build_id, levels):
    """PMC results of the dominant kernel for this workload (separate rocprofv3 --pmc passes, committed under
    profiles/ by tools/profile_round.sh): HBM traffic, VALU instruction count, busy cycles.

    Returns (counters or None, why).  A file is used only when it was measured on the SAME BUILD as the library that is
    loaded now (its `build_id` equals iyk_hip_build_id(): a hash of the kernel sources, tools/src_hash.py) and on the same
    workload (gates per launch, parameter set, gate kind): an instruction count of another kernel build must never be
    divided into a live duration."""
    why = "no counter file under profiles/ for this workload"
    for name in COUNTER_FILES:
        try:
            t = json.load(open(os.path.join(ROOT, "profiles", name)))
            w = t["workload"]
            if not (w["gates_per_launch"] == gates and w["params"] == args.params and w["op"] == args.op
                    and w.get("decomposition_levels", DEFAULT_LEVELS[args.params]) == levels):
                continue
            if t.get("build_id") != build_id:
                why = (f"profiles/{name} was measured on build {t.get('build_id', 'unstamped')}, the loaded library is "
                       f"{build_id}: counters dropped, re-run tools/profile_round.sh")
                continue
            t["_file"] = "profiles/" + name
            return t, None
        except (OSError, KeyError, ValueError):
            continue
    return None, why


FP64_VECTOR_PEAK_FLOPS = 78.6e12   # MI355X vector FP64 (MI355X_MICROARCH.md): 256 CUs x 4 SIMDs x 16 lanes x 2 flop x 2.4 GHz


def fp64_flops_per_step(params, path):
    """FP64 floating-point operations ONE CMUX step of ONE rotation performs on the FFT path, counted from the algorithm as the
    kernel computes it (fused multiply-add = 2, add or multiply = 1), per 64-lane wave = per rotation:
      forward transform (csrc/fft512.hpp, three twisted DFT8 passes of Linzer-Feig butterflies): 3 passes x 12 butterflies x 6
          instructions, all FMAs except the 8 of pass 1's level 1 whose tangent is 1 (adds): (216 x 2 - 8) per lane = 424
      MAC of a row: 8 frequencies x 4 key spectra x 4 FMAs per lane = 256
      inverse transform (three DFT8 of 48 adds + 2 rotations (2 adds + 2 multiplies), conj T2: 7, conj T1: 8 and the uniform twist: 6
          complex products of 2 multiplies + 2 FMAs, + one rotation): 3 x 56 + 7 x 6 + 8 x 6 + 6 x 6 + 4 = 298; rounding: 16 adds
    per step: rows x (424 + 256) + 4 x (298 + 16) per lane, x 64 lanes.  Integer work (digits, rotated difference, recombination) and
    conversions are not counted.  None on the field paths (their arithmetic is exact FMA / integer modular work, not flops)."""
    if path != "fft":
        return None
    rows = params.trgsw_rows
    return 64 * (rows * (424 + 256) + 4 * (298 + 16))


def useful_valu_per_step(params, path):
    """VALU instructions one CMUX step of one rotation NEEDS at the kernel's stated per-operation costs:
    FFT path (round 5): per forward transform 3 twisted DFT8 x 72 + 16 x (bit-field + convert) = 248; per row 8 x 4
    complex MACs x 4 FMAs = 128; per inverse transform 256, + 3 per coefficient to round, shift and combine; 7 per
    coefficient for (X^a - 1) acc.  Field paths: radix-2 butterflies of 8 (fp50) / ~30 (Goldilocks) instructions."""
    rows, N = params.trgsw_rows, params.N            # (k+1) l digit polynomials
    if path == "fft":
        return rows * (248 + 128) + 4 * 256 + 3 * 2 * 16 + 7 * 2 * 16
    per_bfly, per_mac = (8, 7) if path == "fp50" else (30, 30)
    lanes = 64
    bflies = (rows + params.k + 1) * (N // 2) * 10
    return (bflies * per_bfly + (params.k + 1) * rows * N * per_mac) / lanes


def key_stream_bytes(params, path):
    """bytes of bootstrapping key one rotation reads: 8 per key word on the field paths, 2 halves x 8 per word (complex
    spectra of N/2 points x 16 bytes) on the FFT path"""
    words = params.n * params.trgsw_rows * (params.k + 1) * params.N
    return words * (16 if path == "fft" else 8)


def shard(total, world, rank):
    """Contiguous block of a `total`-gate flat batch owned by `rank` (counts differ by at most one)."""
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, base + (1 if rank < extra else 0)


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


def cpu_quota():
    """CPUs the cgroup lets this process use (cpu.max), or None: sched_getaffinity can list 256 CPUs on a box whose
    container may only burn a few of them."""
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        return None if quota == "max" else float(quota) / float(period)
    except (OSError, ValueError):
        return None


def cpu_baseline(keys, params, op_code, sample, data_seed, budget_s=24.0):
    """The oracle (kind 'port') timed on this host's cores on a bounded sample of the same workload.

    All three exact restatements are timed (oracle/tfhe_oracle_fft.c: the GPU's own algorithm — key split into signed 16-bit
    halves, folded complex FP64 transform, AVX2 across transforms; oracle/tfhe_oracle_fp.c: FP64-field products, AVX2 loops;
    oracle/tfhe_oracle.c: Goldilocks 128-bit products) and the FASTEST one is the reported value.  A fourth entry is listed beside
    them and never becomes `value`: TFHEpp's ALGORITHM (round 6) — unsplit key, one inexact FP64 transform per polynomial, i.e. the
    amount of work the reference's CPU path really does per gate; decrypt-equal, not word-equal, to everything else here.  Thread count: ALL visible cores is tried
    first, then halvings of it — the fastest wins and every attempt is listed (round 2 measured 256 threads slower
    than 64 on the driver's box: a cgroup quota or SMT siblings, cpu_quota says which).  Probe chunks (one gate per
    thread) size the final sample so the whole leg takes ~budget_s."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    from iyokan_amd import client

    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    orc = oracle_lib.Oracle(keys)

    def run(count, seed, mode, threads):
        rng = np.random.default_rng(seed)
        bits = rng.integers(0, 2, size=2 * count).astype(np.uint8)
        arena = np.zeros((3 * count, params.n + 1), dtype=np.uint32)
        arena[: 2 * count] = client.encrypt_bits(keys, bits, seed=seed)
        idx = np.arange(count, dtype=np.int32)
        t0 = time.perf_counter()
        orc.gate_batch(np.full(count, op_code, dtype=np.int32), idx, idx + count, np.full(count, -1, dtype=np.int32),
                       idx + 2 * count, arena, nthreads=threads, mode=mode)
        dt = time.perf_counter() - t0
        dec = client.decrypt_bits(keys, arena[2 * count:])
        assert np.array_equal(dec, 1 - (bits[:count] & bits[count:])), "oracle decrypt mismatch"
        return dt

    fast = "fft" if orc.has_fft() else "fp" if orc.has_fp() else "goldilocks"
    # thread-count probe on the faster restatement: one untimed warm-up pass (thread start-up, page faults), then
    # four gates per thread
    tried, t = {}, cores
    while t >= 1 and len(tried) < 5:
        run(t, data_seed, fast, t)
        tried[t] = 4 * t / run(4 * t, data_seed, fast, t)
        t //= 2
    threads = max(tried, key=tried.get)
    # cores that can actually be busy at once: the cgroup's quota when it is below the thread count (32 threads under a quota of 16
    # are 16 cores' worth of work; the per-thread figures below are per BUSY core, or they would double on such a box)
    quota = cpu_quota()
    busy = min(float(threads), quota) if quota else float(threads)
    modes = (["fft"] if orc.has_fft() else []) + (["fp"] if orc.has_fp() else []) + ["goldilocks"]
    results = {}
    for mode in modes:
        share = (budget_s * 0.6) / len(modes)
        rate0 = tried[threads] if mode == fast else threads / run(threads, data_seed, mode, threads)
        n = sample
        if n < 0:
            n = int(max(threads, min(64 * threads, rate0 * share)))
            n -= n % threads
            n = max(n, threads)
        dt = run(n, data_seed + 1, mode, threads)
        results[mode] = (n / dt, n, dt)
    # TFHEpp's algorithm (inexact; two gates per call), same harness and thread count; labelled, never the headline value
    tfhepp_like = None
    try:
        mode = "tfhepp_algorithm_inexact"
        share = budget_s * 0.25
        rate0 = threads / run(threads, data_seed, mode, threads)       # also builds the unsplit key spectra, untimed below
        n = sample if sample >= 0 else int(max(2 * threads, min(64 * threads, rate0 * share)))
        n -= n % (2 * threads)
        n = max(n, 2 * threads)
        dt = run(n, data_seed + 1, mode, threads)
        tfhepp_like = {"gates_per_s": n / dt, "sample_gates": n, "seconds": dt, "ms_per_gate_per_thread": busy / (n / dt) * 1e3,
                       "what": "TFHEpp's ALGORITHM on this repository's transform code (oracle/tfhe_oracle_fft.c, last section): unsplit "
                               "32-bit key, (k+1) l forward + (k+1) inverse FP64 transforms per CMUX step, INEXACT products — decrypt-equal, "
                               "NOT word-equal to the oracle or the GPU; not TFHEpp's code (spqlios), which is not in this container"}
    except Exception as e:   # the baseline must never take the bench line down
        tfhepp_like = {"error": repr(e)}
    orc.close()
    best = max(results, key=lambda m: results[m][0])
    rate, n, dt = results[best]
    names = {"fft": "oracle/tfhe_oracle_fft.c (the GPU's algorithm: split-key folded complex FP64 transform, AVX2 across transforms)",
             "fp": "oracle/tfhe_oracle_fp.c (FP64-field products)", "goldilocks": "oracle/tfhe_oracle.c (Goldilocks products)"}
    return {"value": rate, "unit": "gates/s", "cores": threads, "kind": "port",
            "sample": f"{n} NAND gates of the same workload in {dt:.1f} s on {threads} threads ({cores} CPUs visible, fastest of "
                      f"the thread counts tried), own exact CPU restatement {names[best]}, OpenMP over gates; not TFHEpp",
            "restatements": dict({m: {"gates_per_s": r[0], "sample_gates": r[1], "seconds": r[2]} for m, r in results.items()},
                                 tfhepp_algorithm_inexact=tfhepp_like),
            "threads_tried_gates_per_s": {str(k): v for k, v in tried.items()},
            "visible_cpus": cores, "cpu_quota": cpu_quota(),
            "busy_cores": busy, "ms_per_gate_per_thread": busy / rate * 1e3}


def word_check(params_name, op, first_gate, rows):
    """The output ciphertexts of the TIMED step against the oracle's committed digests (tests/golden/fullsize_nand_fresh_*.bin:
    first 8 bytes of sha256 of every output TLWE of this very workload — key seed 1, bits default_rng(1000), encryption seed 2,
    gate g = NAND(in[g], in[65536 + g]) — made by tests/golden/make_fullsize_digests.py with the CPU oracle in the build
    container).  rows = this rank's block, starting at gate first_gate.  Returns True / False, or None when the workload is not
    the committed one (another gate kind or size)."""
    import hashlib

    path = os.path.join(ROOT, "tests", "golden", f"fullsize_nand_fresh_{'128' if params_name == '128bit' else '80'}.bin")
    if op != "NAND" or not os.path.exists(path):
        return None
    want = np.fromfile(path, dtype=np.uint8).reshape(-1, 8)
    if first_gate + len(rows) > len(want):
        return None
    got = np.frombuffer(b"".join(hashlib.sha256(r.tobytes()).digest()[:8] for r in rows), dtype=np.uint8).reshape(-1, 8)
    return bool(np.array_equal(got, want[first_gate: first_gate + len(rows)]))


EXIT_BAD_WORLD = 3   # --gpus does not match the ranks that exist / the devices that are visible


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launcher(args, script=None, argv=None, visible_gpus=None):
    """Make sure this process is one of exactly args.gpus ranks, one per GPU.

    * Under a launcher (RANK / WORLD_SIZE in the environment): WORLD_SIZE must equal --gpus, else exit EXIT_BAD_WORLD.
    * Stand-alone with --gpus N > 1 (or --spawn): check that N devices are visible, then re-execute this script under
      `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free>` and
      exit with its status — so `python bench.py --gpus 8` is a complete 8-GPU run, never a silent 1-GPU one.
    * Stand-alone with --gpus 1: nothing to do.
    Returns (rank, world, local_rank, distributed)."""
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None:
        if args.gpus < 1:
            print(f"bench: --gpus {args.gpus} is not a GPU count", file=sys.stderr)
            sys.exit(EXIT_BAD_WORLD)
        if args.gpus == 1 and not getattr(args, "spawn", False):
            return 0, 1, 0, False
        if visible_gpus is None:
            import torch

            visible_gpus = torch.cuda.device_count()
        if visible_gpus < args.gpus:
            print(f"bench: --gpus {args.gpus} but only {visible_gpus} GPU(s) visible; refusing to run on fewer", file=sys.stderr)
            sys.exit(EXIT_BAD_WORLD)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
               script or os.path.abspath(sys.argv[0])] + list(sys.argv[1:] if argv is None else argv)
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL across processes needs it on this driver
        sys.exit(subprocess.call(cmd, env=env))
    world = int(env_world)
    rank = int(os.environ.get("RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks; refusing to report a "
                  f"{args.gpus}-GPU number from {world} rank(s)", file=sys.stderr)
        sys.exit(EXIT_BAD_WORLD)
    return rank, world, int(os.environ.get("LOCAL_RANK", "0")), True


def main():
    args = parse_args()
    rank, world, local_rank, distributed = launcher(args)
    import torch
    import torch.distributed as dist

    from iyokan_amd import client, hip
    from iyokan_amd.params import OPS, PLAIN, params_by_name

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback for the product path)")
    if torch.cuda.device_count() <= local_rank:
        print(f"bench: rank {rank} wants cuda:{local_rank} but {torch.cuda.device_count()} device(s) are visible", file=sys.stderr)
        sys.exit(EXIT_BAD_WORLD)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if distributed:   # also with one rank (--spawn): the RCCL communicator, broadcast and reductions run for real
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=dev)
        if dist.get_world_size() != args.gpus:
            print(f"bench: RCCL sees {dist.get_world_size()} rank(s), --gpus {args.gpus}", file=sys.stderr)
            sys.exit(EXIT_BAD_WORLD)

    params = params_by_name(args.params)
    op_code = OPS[args.op]
    G_total = args.gates
    _, G_mine = shard(G_total, world, rank)      # strong scaling: this rank's block of the one batch
    G_alloc = G_total if (world > 1 and not args.no_weak) else G_mine   # the weak leg runs G_total gates per rank

    # ---- keys: rank 0 generates, RCCL broadcast (north_star: "bootstrapping key broadcast once") ----
    keys = client.keygen(params, seed=1) if rank == 0 else empty_keys(params)
    if distributed:
        keys = broadcast_keys(keys, dist, dev, rank)
    if args.decomp:
        os.environ["IYK_HIP_DECOMP"] = args.decomp
    hip.initialize(keys, device_ids=(local_rank,))

    # ---- synthetic inputs: fresh encryptions, distinct per rank (seeded); layout [in0 | in1 | out] ----
    # (rank 0's are SURVEY 8(d) config 2's: data seed 2 — the workload the committed oracle digests were made on; at N > 1 every
    # rank times its own block of its own 65 536 fresh gates: same work per gate, no shared inputs to distribute)
    rng = np.random.default_rng(1000 + rank)
    bits = rng.integers(0, 2, size=2 * G_alloc).astype(np.uint8)
    enc = client.encrypt_bits(keys, bits, seed=2 + rank)
    arena_t = torch.zeros((3 * G_alloc, params.n + 1), dtype=torch.int32, device=dev)
    arena_t[: 2 * G_alloc].copy_(torch.from_numpy(enc.view(np.int32)))
    del enc
    tstream = torch.cuda.Stream(device=dev)
    torch.cuda.synchronize()
    st = hip.Stream(0, hip_stream=tstream.cuda_stream)
    arena = hip.Arena.from_torch(arena_t)

    def batch(count):
        idx = np.arange(count, dtype=np.int32)
        return (np.full(count, op_code, dtype=np.int32), idx, idx + G_alloc, np.full(count, -1, dtype=np.int32),
                idx + 2 * G_alloc)

    def fence():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(count, steps, warmup):
        """W untimed + exactly K timed steps of `count` gates on this rank, barrier + synchronize on both sides,
        MAX over ranks; also the summed kernel times from HIP events on the launch stream."""
        desc = batch(count)
        for _ in range(warmup):
            st.gate_batch(arena, *desc)
        fence()
        st.timing_log_begin()
        t0 = time.perf_counter()
        for _ in range(steps):
            st.gate_batch(arena, *desc)
        fence()
        elapsed = time.perf_counter() - t0
        nb, br_ms, ks_ms = st.timing_log_end()
        per_rank = [elapsed]
        if distributed:
            mine = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            every = torch.zeros(world, dtype=torch.float64, device=dev)
            dist.all_gather_into_tensor(every, mine)
            dist.all_reduce(mine, op=dist.ReduceOp.MAX)
            elapsed = float(mine.item())
            per_rank = [float(v) for v in every.cpu()]
        return elapsed, nb, br_ms, ks_ms, per_rank

    elapsed, nb, br_ms, ks_ms, per_rank = timed(G_mine, args.steps, args.warmup)

    # ---- correctness of what was just timed: decrypt every output of the last step ----
    got = arena_t[2 * G_alloc: 2 * G_alloc + G_mine].cpu().numpy().view(np.uint32)
    want = np.array([PLAIN[args.op](int(a), int(b)) for a, b in zip(bits[:G_mine], bits[G_alloc: G_alloc + G_mine])],
                    dtype=np.uint8)
    decrypt_ok = bool(np.array_equal(client.decrypt_bits(keys, got), want))
    # ... and, on rank 0 (whose inputs are the committed workload's), the WORDS of the timed step against the oracle's digests
    words_ok = word_check(args.params, args.op, 0, got) if (rank == 0 and G_alloc == 65536 and G_total == 65536) else None

    weak = None
    if world > 1 and not args.no_weak:
        w_elapsed, _, _, _, _ = timed(G_total, args.steps, 1)
        weak = {"value": G_total * world * args.steps / w_elapsed, "unit": "gates/s",
                "gates_per_step_per_gpu": G_total, "ms_per_step": w_elapsed / args.steps * 1e3, "scaling": "weak"}

    if rank == 0:
        path = hip.ntt_path()                         # "fft" (default), "fp50", "goldilocks"
        value = G_total * args.steps / elapsed
        b_gate = params.gate_algorithmic_bytes(rotations=1, inputs=2)
        # dominant kernel = blind_rotate: its share of B_gate is the BK stream at 8 bytes per key word + its own I/O
        br_bytes_per_gate = params.n * params.trgsw_rows * (params.k + 1) * params.N * 8 \
            + 2 * (params.n + 1) * 4 + (params.N + 1) * 4
        br_avg_s = (br_ms / max(nb, 1)) * 1e-3
        achieved = br_bytes_per_gate * G_mine / br_avg_s if br_avg_s > 0 else 0.0
        pmc, pmc_why = counters(args, G_mine, hip.build_id(), hip.decomposition_levels())
        traffic = args.traffic_bytes if args.traffic_bytes is not None else (pmc or {}).get("traffic_bytes_per_launch")
        t_over_a = (traffic / (br_bytes_per_gate * G_mine)) if traffic else None
        steps_per_launch = G_mine * params.n                  # one wave per rotation, n CMUX steps each
        capacity = (br_avg_s * 1e9 / VALU_PEAK_NS) * N_SIMDS if br_avg_s > 0 else 0.0   # issue slots of the launch (one per 4 cycles per SIMD at 2.4 GHz)
        # USEFUL work: the instructions the algorithm needs per CMUX step and wave at the kernel's stated costs (DESIGN.md
        # section 4.1) — no waits, spills, address arithmetic or table reads — against the same issue capacity
        useful = useful_valu_per_step(params, path)
        issue = {
            "note": "what bounds this kernel: a SIMD issues one instruction of ANY kind per ~4 cycles (DESIGN.md section 8); "
                    "capacity = 1024 SIMDs x launch time / (4 cycles at 2.4 GHz)",
            "useful_valu_per_step_per_wave": useful,
            "useful_frac": (useful * steps_per_launch / capacity) if capacity else None,
        }
        if pmc and pmc.get("valu_insts_per_launch") and capacity:
            issue.update({
                "valu_insts_per_step_per_wave": pmc["valu_insts_per_launch"] / steps_per_launch,
                "valu_frac": pmc["valu_insts_per_launch"] / capacity,
                "source": pmc["_file"], "build_id": pmc.get("build_id"),
            })
            if pmc.get("all_insts_per_launch"):
                issue["all_insts_per_step_per_wave"] = pmc["all_insts_per_launch"] / steps_per_launch
                issue["frac"] = pmc["all_insts_per_launch"] / capacity     # issued instructions of any kind / capacity
            for k in ("lds_insts_per_launch", "vmem_rd_insts_per_launch", "salu_insts_per_launch", "smem_insts_per_launch",
                      # (sustained_clock_ghz stays in the counter file: it is the clock of the box and pass that MEASURED it, not of the
                      # box running this bench, which has no way to read its own clock without rocprof — VERDICT r05 #9)
                      "l1_requests_per_launch", "l1_to_l2_requests_per_launch", "l2_hit_rate"):
                if pmc.get(k) is not None:
                    issue[k] = pmc[k]
        else:
            issue["counters_dropped"] = pmc_why
        kernel = {"fft": "blind_rotate_fft_kernel", "fp50": "blind_rotate_fp_kernel", "goldilocks": "blind_rotate_kernel"}[path]
        # --- the three prices of the same launch --------------------------------------------------------------------------------
        # (1) hbm_contract: SURVEY 8(d)'s figure — algorithmic key bytes / launch time vs 8 TB/s.  It cannot bind: every wave of a
        #     launch walks the same key rows in lock-step, the stream is served by L1 / L2 (traffic_over_algorithmic << 1), values
        #     above 1 happen.  Kept because the contract asks for it; `binding` says what it is worth.
        hbm_contract = {
            "achieved": achieved / 1e9, "peak": HBM_PEAK_BYTES_PER_S / 1e9, "unit": "GB/s", "frac": achieved / HBM_PEAK_BYTES_PER_S,
            "binding": (t_over_a >= 0.5) if t_over_a is not None else None,
            "algorithmic_bytes_per_launch": br_bytes_per_gate * G_mine,
            "key_bytes_streamed_per_rotation": key_stream_bytes(params, path),
            "traffic_over_algorithmic": t_over_a,
            "measured_hbm_GBps": (traffic / br_avg_s / 1e9) if (traffic and br_avg_s > 0) else None,
            "gate_bytes": b_gate, "gate_frac": value / world * b_gate / HBM_PEAK_BYTES_PER_S,
        }
        # (2) fp64: flops counted from the algorithm (fp64_flops_per_step's docstring is the formula) / launch time vs the part's
        #     vector FP64 peak: a hardware-spec'd compute fraction anyone can recompute from the launch time alone
        fl = fp64_flops_per_step(params, path)
        fp64 = None
        if fl and br_avg_s > 0:
            flops = fl * steps_per_launch
            fp64 = {"flops_per_step_per_rotation": fl, "flops_per_launch": flops, "achieved": flops / br_avg_s / 1e12,
                    "peak": FP64_VECTOR_PEAK_FLOPS / 1e12, "unit": "TFLOP/s", "frac": flops / br_avg_s / FP64_VECTOR_PEAK_FLOPS,
                    "formula": "64 lanes x (rows x (424 forward + 256 MAC) + 4 x (298 inverse + 16 rounding)) x n steps x rotations; "
                               "FMA = 2 (bench.py: fp64_flops_per_step)"}
        # (3) issue: instructions of ANY kind the launch issued (SQ_INSTS of a committed --pmc pass on this build) against one per
        #     4 cycles per SIMD at 2.4 GHz — the measured bound of this kernel.  The HEADLINE frac when the
        #     counter file matches the loaded build; otherwise the headline falls back to (2), which needs no counters.
        if issue.get("frac") is not None:
            insts = pmc["all_insts_per_launch"]
            head = {"bound": "valu-issue", "frac_kind": "issue: SQ_INSTS (all instruction kinds) per launch / (1024 SIMDs x launch time / "
                    "4 cycles at 2.4 GHz); counters from " + pmc["_file"],
                    "achieved": insts / br_avg_s / 1e9, "peak": N_SIMDS / VALU_PEAK_NS, "unit": "G wave-instructions/s",
                    "frac": issue["frac"]}
        elif fp64:
            head = {"bound": "fp64-valu", "frac_kind": "fp64: algorithm-counted FP64 flops / launch time vs 78.6 TFLOP/s vector FP64 peak "
                    "(no counter file matches this build: " + str(pmc_why) + ")",
                    "achieved": fp64["achieved"], "peak": fp64["peak"], "unit": "TFLOP/s", "frac": fp64["frac"]}
        else:
            head = {"bound": "hbm", "frac_kind": "contract: SURVEY 8(d) algorithmic bytes / launch time vs 8 TB/s",
                    "achieved": hbm_contract["achieved"], "peak": hbm_contract["peak"], "unit": "GB/s", "frac": hbm_contract["frac"]}
        roofline = dict(head)
        roofline.update({
            "kernel": kernel,
            "contract_frac": hbm_contract["frac"],   # SURVEY 8(d)'s "% of HBM roofline", non-binding (see hbm_contract)
            "binding": hbm_contract["binding"],
            "traffic": traffic,                        # HBM bytes per launch: rocprofv3 2 * FETCH_SIZE + WRITE_SIZE (gfx950 correction)
            "avg_launch_ms": br_avg_s * 1e3,
            "keyswitch_avg_launch_ms": ks_ms / max(nb, 1),
            "hbm_contract": hbm_contract,
            "fp64": fp64,
            "issue": issue,
            "build_id": hip.build_id(),
        })
        line = {
            "metric": baseline_metric() if args.params == "128bit" else "TFHE gate bootstraps/sec (80-bit params)",
            "value": value,
            "unit": "gates/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "per_rank_ms_per_step": [v / args.steps * 1e3 for v in per_rank],
            "rccl_world_size": dist.get_world_size() if distributed else None,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": {"fft": "u32 torus; products exact through a 512-point complex f64 FFT on signed 16-bit key halves (proven rounding error < 2^-5, rint)",
                      "fp50": "u32 torus; NTT in f64 mod p = 3*2^48+1097729 (exact FMA arithmetic)",
                      "goldilocks": "u32 torus; NTT in u64 mod 2^64-2^32+1"}[path],
            "data": "synthetic",
            "config": {
                "workload": f"{G_total} independent Hom{args.op} gates per step (flat DAG), {args.params} params, "
                            f"fresh encryptions, keys+ciphertexts resident in HBM; sharded {G_mine} per GPU",
                "params": {k: v for k, v in params.as_dict().items() if k not in ("alpha0", "alpha1")},
                "decomposition_levels": hip.decomposition_levels(),
                "gates_per_step": G_total,
                "gates_per_step_per_gpu": G_mine,
                "parallelism": f"frontier sharded over {world} GPU(s), no data-path collective (flat DAG)",
                "decrypt_check": decrypt_ok,
                "word_check": words_ok,
                "word_check_note": "sha256 digests of the timed step's output ciphertexts (rank 0's block) == the CPU oracle's committed "
                                   "digests for this workload (tests/golden/fullsize_nand_fresh_*.bin); null = not the committed workload",
            },
            "roofline": roofline,
            "weak": weak,
        }
        if world == 1 and args.cpu_sample != 0:
            line["cpu_baseline"] = cpu_baseline(keys, params, op_code, args.cpu_sample, data_seed=99)
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)

    st.destroy()
    del arena, arena_t
    hip.cleanup()
    if distributed:
        dist.destroy_process_group()
    if not decrypt_ok:
        raise SystemExit("decrypt check failed")
    if rank == 0 and words_ok is False:
        raise SystemExit("word check failed: the timed step's ciphertexts differ from the oracle's")


if __name__ == "__main__":
    main()
