#!/usr/bin/env python3
"""Clock-latency benchmark of a whole netlist through the frontier-sharded executor
(BASELINE configs #3 / #4 shape).  `python tools/bench_netlist.py --net mux-ram --gpus N`: with N > 1 (or --spawn) it
re-executes itself under torch.distributed.run, one rank per GPU (bench.launcher: refuses to run when fewer than N
devices are visible or the launcher's WORLD_SIZE is not N).  Keys are generated on rank 0 and broadcast once over RCCL
(the north_star's "bootstrapping key broadcast once"); inputs are fresh encryptions; every output bit is checked against
the plaintext simulator after the timed clocks.  Prints one JSON line on rank 0."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

NETS = {
    "mux-ram": ("mux-ram-8-16-16.min.json", "l1", "test08.in"),      # config #3
    "cahp-ruby": ("cahp-ruby-core-yosys.json", "yosys", None),        # the core of config #4 (no ROM/RAM wiring)
    "counter": ("counter-4bit-iyokanl1.json", "l1", None),
    "cahp-system": ("cahp-ruby-mux.toml", "blueprint", None),        # config #4: core + MUX ROM + MUX RAM
}


def broadcast_cost_table(table, dist, device):
    """rank 0's iyk_level_cost (a dict of numbers + build id) to every rank, as one float64 tensor"""
    import torch

    keys = ["round", "pass", "max_passes", "calibrated", "round_ms"]
    vec = [float(table[k]) for k in keys] + [float(v) for v in table["pass_ms"]]
    t = torch.tensor(vec, dtype=torch.float64, device=device)
    dist.broadcast(t, src=0)
    vals = [float(v) for v in t.cpu()]
    out = dict(table)
    for k, v in zip(keys, vals):
        out[k] = bool(round(v)) if k == "calibrated" else (int(round(v)) if k != "round_ms" else v)
    out["pass_ms"] = vals[len(keys):]
    return out


def plan_fingerprint(plan):
    """63 bits of sha256 over the plan's levels (node ids per level) and slot assignment"""
    import hashlib

    h = hashlib.sha256()
    for level in plan.levels:   # {"boot": node ids, "ew": node ids, "base": first slot, "B": gates per rank}
        h.update(np.asarray(level["boot"], dtype=np.int64).tobytes() + b"|" + np.asarray(level["ew"], dtype=np.int64).tobytes())
        h.update(np.asarray([level["base"], level["B"]], dtype=np.int64).tobytes())
    h.update(np.asarray(plan.slot, dtype=np.int64).tobytes())
    return int.from_bytes(h.digest()[:8], "little") >> 1


def assert_same_plan(plan, dist, device):
    import torch

    mine = torch.tensor([plan_fingerprint(plan)], dtype=torch.int64, device=device)
    lo, hi = mine.clone(), mine.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    if int(lo.item()) != int(hi.item()):
        raise SystemExit("bench_netlist: the ranks derived different level plans — refusing to run the level exchange")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--net", default="mux-ram", choices=sorted(NETS))
    ap.add_argument("--clocks", type=int, default=2)
    ap.add_argument("--burst", type=int, default=4,
                    help="clocks run back to back after the per-clock ones (tick + run, inputs held, ONE sync at the end): the shape of the "
                         "reference's clock loop, /root/reference/src/iyokan_cufhe.cpp:754-802; 0 = skip")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--plan", default="balanced", choices=["balanced", "asap", "nospread"],
                    help="level plan: gates with slack placed where a level's step-shaped cost is lowest (frontier.plan_levels, "
                         "the default) or every gate at its earliest level")
    ap.add_argument("--spawn", action="store_true", help="go through torch.distributed.run even for one GPU (RCCL with one rank)")
    args = ap.parse_args()
    import bench

    rank, world, local, distributed = bench.launcher(args, script=os.path.abspath(__file__))
    import torch
    import torch.distributed as dist

    from iyokan_amd import client, hip
    from iyokan_amd import netlist as N
    from iyokan_amd.frontier import FrontierExecutor, FrontierPlan, HipBackend, level_rotations, make_level_cost, with_sub_pass_shape
    from iyokan_amd.params import params_128bit
    from netlist_util import gold, drive_cycle, input_streams, load_packet

    if torch.cuda.device_count() <= local:
        print(f"bench_netlist: rank {rank} wants cuda:{local} but {torch.cuda.device_count()} device(s) are visible", file=sys.stderr)
        sys.exit(bench.EXIT_BAD_WORLD)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    fname, kind, pkt = NETS[args.net]
    if kind == "blueprint":
        from iyokan_amd.system import load_blueprint

        nl = load_blueprint(gold(fname)).nl
    else:
        nl = (N.load_iyokanl1_json if kind == "l1" else N.load_yosys_json)(gold(fname))
    streams = input_streams(load_packet(gold(pkt))) if pkt else {}
    p = params_128bit()
    keys = client.keygen(p, seed=1) if rank == 0 else bench.empty_keys(p)
    if distributed:
        keys = bench.broadcast_keys(keys, dist, dev, rank)   # rank 0's key material, once, over RCCL
    hip.initialize(keys, device_ids=(local,))
    # The level plan (cuts, slot layout, per-level widths) is a function of the cost table, and every rank must derive the SAME
    # plan: the in-place all_gather_into_tensor assumes identical offsets and sizes everywhere.  Each rank calibrates its own GPU
    # (its dispatch threshold follows its own measurement), but the PLAN is made from rank 0's table, broadcast — noisy per-rank
    # floats picking different cuts would hang RCCL or corrupt ciphertexts.  A hash of the plan is compared
    # across the ranks before anything runs.
    table = hip.calibrate(0)
    if distributed:
        table = broadcast_cost_table(table, dist, dev)
    level_cost = make_level_cost(table)
    plan = FrontierPlan(nl, world, balance=args.plan != "asap", cost=level_cost, spread=args.plan != "nospread")   # nospread: without round 6's capped candidates (A/B)
    if distributed:
        assert_same_plan(plan, dist, dev)
    be = HipBackend(plan.num_slots, p, dev)
    ex = FrontierExecutor(plan, be, rank, world, dist if distributed else None)   # one rank too: the level exchange runs on RCCL
    if distributed and world == 1:
        dist.barrier()   # one-rank RCCL communicator (--spawn): at least one collective on it
    sim = N.PlainSimulator(nl)
    zero = client.trivial(p, 0)
    rng = np.random.default_rng(5)
    seed = [1000]

    def set_enc(port, bit, v):
        seed[0] += 1
        ex.set_input(port, bit, client.encrypt_bits(keys, [v], seed=seed[0])[0])   # same seed on every rank

    def drive(c):
        if streams:
            drive_cycle(set_enc, nl, streams, c)
            drive_cycle(sim.set_input, nl, streams, c)
        else:
            for (port, bit) in sorted(nl.inputs):
                v = int(rng.integers(0, 2)) if port != "reset" else int(c == 0)
                set_enc(port, bit, v)
                sim.set_input(port, bit, v)

    times = []
    for c in range(args.clocks + 1):          # clock 0 is the warm-up
        ex.tick(); sim.tick()
        if c == 0:
            be.write_many([plan.slot[i] for i in plan.dffs], np.tile(zero, (len(plan.dffs), 1)))
            be.write_many([plan.slot[i] for i in plan.sources], np.tile(zero, (len(plan.sources), 1)))
        drive(c)
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ex.run()
        ex.sync()
        if distributed:
            dist.barrier()
        times.append(time.perf_counter() - t0)
        sim.evaluate()
    # The clocks above are timed ONE AT A TIME with the host re-encrypting and uploading every input in between (tens of small
    # synchronous copies): the GPU idles, drops its clock, and the first ~10 levels of the next run() pay for the ramp
    # (profiles/r06_level_gaps.txt: 3.08, 3.12, 2.93 ... 2.52 ms for equal 256-rotation levels).  Upstream's clock loop sets its
    # inputs once and then runs clock after clock; the burst below is that shape — tick + run, `burst` times, nothing waited for
    # in between, the latch inside the timed region — checked against the simulator like the clocks before it.
    burst_s = None
    if args.burst > 0:
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.burst):
            ex.tick()
            ex.run()
        ex.sync()
        if distributed:
            dist.barrier()
        burst_s = (time.perf_counter() - t0) / args.burst
        for _ in range(args.burst):
            sim.tick()
            sim.evaluate()
    outs = sorted(nl.outputs)
    got = client.decrypt_bits(keys, be.read_many([plan.slot[nl.outputs[k]] for k in outs]))
    ok = list(got) == [sim.get_output(*k) for k in outs]
    if rank == 0:
        rot = nl.rotations()
        best = min(times[1:])
        print(json.dumps({"net": args.net, "n_gpus": world, "rccl_world_size": dist.get_world_size() if distributed else None, "levels": len(plan.levels), "rotations_per_clock": rot,
                          "s_per_clock": best, "s_per_clock_back_to_back": burst_s, "burst": args.burst, "rotations_per_s": rot / best, "collectives_per_clock": ex.collectives // (args.clocks + 1 + args.burst),
                          "outputs_match_plaintext": ok, "ntt_path": hip.ntt_path(), "plan": args.plan, "cost_table": level_cost.table,
                          "model_s_per_clock": sum(with_sub_pass_shape(level_cost)(r) for r in level_rotations(nl, [L["boot"] for L in plan.levels], world)) / 1e3}))
    be.close()
    hip.cleanup()
    if distributed:
        dist.destroy_process_group()
    if not ok:
        raise SystemExit("output mismatch")


if __name__ == "__main__":
    main()
