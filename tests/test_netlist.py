"""Netlist ingestion + plaintext evaluation against the reference's own request/result fixtures,
and the frontier executor's sharding logic on a bit backend (single rank and gloo world_size 2)."""
import os

import numpy as np
import pytest

from iyokan_amd import netlist as N
from iyokan_amd.frontier import FrontierExecutor, FrontierPlan, PlainBitBackend
from netlist_util import gold, drive_cycle, input_streams, load_packet, run_plain


def test_counter_yosys_matches_test13():
    nl = N.load_yosys_json(gold("counter-4bit-yosys.json"))
    want = load_packet(gold("test13.out"))
    sim = run_plain(nl, {}, want["cycles"], use_reset=True)
    assert N.bytes_from_bits([sim.get_output("io_out", b) for b in range(4)]) == want["bits"][0]["bytes"]


def test_counter_l1_sequence_like_test0():
    """/root/reference/src/test0.cpp:403-431: 16 clocks count 0..15."""
    nl = N.load_iyokanl1_json(gold("counter-4bit-iyokanl1.json"))
    sim = N.PlainSimulator(nl)
    sim.set_input("reset", 0, 1)
    sim.evaluate()
    sim.set_input("reset", 0, 0)
    for clk in range(16):
        sim.tick()
        sim.evaluate()
        assert sim.get_port("io_out") == clk


@pytest.mark.parametrize("loader,name", [(N.load_yosys_json, "addr-4bit-yosys.json"),
                                         (N.load_iyokanl1_json, "addr-4bit-iyokanl1.json")])
def test_adder_matches_test04(loader, name):
    nl = loader(gold(name))
    req, want = load_packet(gold("test04.in")), load_packet(gold("test04.out"))
    streams = input_streams(req)
    port_a = "io_inA" if ("io_inA", 0) in nl.inputs else "A"
    port_b = "io_inB" if ("io_inB", 0) in nl.inputs else "B"
    streams = {port_a: streams["A"], port_b: streams["B"]}
    sim = run_plain(nl, streams, want["cycles"], use_reset=False)
    out_port = next(p for (p, b) in nl.outputs)
    width = nl.port_width(nl.outputs, out_port)
    assert N.bytes_from_bits([sim.get_output(out_port, b) for b in range(width)])[0] & 0xF == want["bits"][0]["bytes"][0]


def test_mux_ram_8_16_16_matches_test08():
    """BASELINE config #3 netlist, full 8-clock run of test.rb's `mux-ram-8-16-16` case."""
    nl = N.load_iyokanl1_json(gold("mux-ram-8-16-16.min.json"))
    c = nl.counts()
    assert c["MUX"] == 5872 and c["DFF"] == 4096 and nl.rotations() == 18985
    assert len(nl.levelise()) == 14
    req, want = load_packet(gold("test08.in")), load_packet(gold("test08.out"))
    sim = run_plain(nl, input_streams(req), want["cycles"], use_reset=False)
    rdata = N.bytes_from_bits([sim.get_output("rdata", b) for b in range(16)])
    assert rdata == want["bits"][0]["bytes"]
    assert N.bytes_from_bits(sim.ram_image(4096)) == want["ram"][0]["bytes"]


def test_cahp_ruby_statistics():
    nl = N.load_yosys_json(gold("cahp-ruby-core-yosys.json"))
    assert nl.rotations() == 4281 and len(nl.levelise()) == 41 and nl.counts()["DFF"] == 483


def _frontier_vs_sim(nl, streams, cycles, world, rank, dist):
    plan = FrontierPlan(nl, world)
    ex = FrontierExecutor(plan, PlainBitBackend(plan.num_slots), rank, world, dist)
    sim = N.PlainSimulator(nl)
    for c in range(cycles):
        ex.tick(); sim.tick()
        drive_cycle(ex.set_input, nl, streams, c)
        drive_cycle(sim.set_input, nl, streams, c)
        ex.run(); sim.evaluate()
        for (port, bit) in nl.outputs:
            assert ex.get_output(port, bit) == sim.get_output(port, bit), (c, port, bit)
    for i in plan.dffs:
        assert ex.get_node(i) == int(sim.val[i])
    return ex


def test_frontier_single_rank_matches_simulator():
    nl = N.load_iyokanl1_json(gold("mux-ram-8-16-16.min.json"))
    req = load_packet(gold("test08.in"))
    _frontier_vs_sim(nl, input_streams(req), 3, 1, 0, None)


def _gloo_worker(rank, world, port, q):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        nl = N.load_iyokanl1_json(gold("mux-ram-8-16-16.min.json"))
        req = load_packet(gold("test08.in"))
        ex = _frontier_vs_sim(nl, input_streams(req), 2, world, rank, dist)
        q.put((rank, ex.collectives))
    finally:
        dist.destroy_process_group()


def test_frontier_two_ranks_gloo():
    """world_size 2 on CPU: every rank evaluates only its share of each level, the per-level
    all_gather makes both arenas identical, and both match the plaintext simulator."""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    got = sorted(q.get(timeout=10) for _ in range(2))
    assert got[0][1] == got[1][1] == 2 * 14      # one collective per level per clock


def _gloo_worker8(rank, world, port, q):
    import hashlib

    import torch.distributed as dist

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        nl = N.load_iyokanl1_json(gold("mux-ram-8-16-16.min.json"))
        req = load_packet(gold("test08.in"))
        ex = _frontier_vs_sim(nl, input_streams(req), 2, world, rank, dist)
        # the WHOLE replicated arena, pad slots of ragged levels included: every rank must hold the same bytes
        q.put((rank, ex.collectives, hashlib.sha256(ex.be.arena.numpy().tobytes()).hexdigest()))
    finally:
        dist.destroy_process_group()


def test_frontier_eight_ranks_gloo():
    """The 8-GPU shape rehearsed where it can be: world_size 8 over gloo on config #3's netlist.  Every rank
    evaluates its own block [base + r B, base + (r + 1) B) of each level, ONE in-place all-gather per level makes the arenas
    identical — byte for byte, the pad slots of ragged levels included — every output and register equals the plaintext simulator
    on every rank, and collectives == levels per clock.  The netlist has levels whose width is not a multiple of 8 (ragged last
    ranks: exactly what a first 8-rank run would trip on)."""
    import torch.multiprocessing as mp

    world = 8
    nl = N.load_iyokanl1_json(gold("mux-ram-8-16-16.min.json"))
    plan = FrontierPlan(nl, world)
    widths = [len(L["boot"]) for L in plan.levels]
    assert any(w % world for w in widths), widths
    ragged = [L for L in plan.levels if len({len(d[0]) for d in L["rank_desc"]}) > 1]
    assert ragged, "config #3 must exercise ranks with unequal shares"
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_gloo_worker8, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=600)
        assert p.exitcode == 0
    got = sorted(q.get(timeout=10) for _ in range(world))
    assert [g[0] for g in got] == list(range(world))
    assert {g[1] for g in got} == {2 * len(plan.levels)}      # one collective per level per clock, on every rank
    assert len({g[2] for g in got}) == 1                      # identical arenas


def test_plan_shards_are_balanced():
    nl = N.load_iyokanl1_json(gold("mux-ram-8-16-16.min.json"))
    plan = FrontierPlan(nl, 8)
    for L in plan.levels:
        sizes = [len(d[0]) for d in L["rank_desc"]]
        assert max(sizes) - min(sizes) <= 1
        mux = [int((d[0] == 8).sum()) for d in L["rank_desc"]]
        assert max(mux) - min(mux) <= 1
    # every gate appears exactly once across ranks
    seen = sorted(int(o) for L in plan.levels for d in L["rank_desc"] for o in d[4])
    assert len(seen) == len(set(seen)) == sum(len(L["boot"]) for L in plan.levels)


@pytest.mark.parametrize("name,kind", [("cahp-ruby-core-yosys.json", "yosys"), ("mux-ram-8-16-16.min.json", "l1"),
                                       ("cahp-ruby-mux.toml", "blueprint"), ("counter-4bit-iyokanl1.json", "l1")])
@pytest.mark.parametrize("world", [1, 2, 8])
def test_balanced_levels_are_a_valid_cheaper_schedule(name, kind, world):
    """frontier.plan_levels: gates with slack moved to the level where the kernels' step-shaped cost is lowest.  The
    schedule is valid (every node exactly once; every input produced in an EARLIER level, so that no batch reads a slot it
    writes), keeps the number of levels — the critical path and the number of level-boundary exchanges — and never costs
    more than the earliest-level schedule by the model it optimises."""
    from iyokan_amd import frontier as F

    if kind == "blueprint":
        from iyokan_amd.system import load_blueprint

        nl = load_blueprint(gold(name)).nl
    else:
        nl = (N.load_iyokanl1_json if kind == "l1" else N.load_yosys_json)(gold(name))
    asap = nl.levelise()
    for levels in (F.balanced_levels(nl, world), F.balanced_levels(nl, world, quanta=(256,)), F.beam_levels(nl, world), F.plan_levels(nl, world)):
        assert len(levels) == len(asap)
        where = {}
        for k, lv in enumerate(levels):
            for i in lv:
                assert i not in where
                where[i] = k
        assert sorted(where) == sorted(i for lv in asap for i in lv)
        root = nl.roots()
        for i, k in where.items():
            for j in nl.ins[i]:
                j = root[j]
                if nl.kinds[j] not in ("INPUT", "DFF"):
                    assert where[j] < k, (i, j)
    cost = lambda lv: sum(F.mi355x_level_cost(r) for r in F.level_rotations(nl, lv, world))
    assert cost(F.plan_levels(nl, world)) <= min(cost(asap), cost(F.balanced_levels(nl, world))) + 1e-9


@pytest.mark.parametrize("world", [1, 8])
def test_spread_plan_of_a_depth_bound_netlist(world):
    """Round 6: every level of the CAHP core costs one pass of the narrow-frontier kernel whatever it holds, and a FULL pass is the
    dear one (all CUs busy: the power limit; frontier.SUB_PASS_SHAPE).  capped_levels spreads the rotations over all 41 levels — a
    valid schedule (every gate once, after its inputs, depth unchanged, gates at their ALAP level never deferred) — and plan_levels
    prefers it: no level above half a pass per rank where round 5's plan had eleven full ones; spread=False is round 5's plan."""
    from iyokan_amd import frontier as F

    nl = N.load_yosys_json(gold("cahp-ruby-core-yosys.json"))
    asap = nl.levelise()
    for cap in (64, 128, 200):
        lv = F.capped_levels(nl, world, cap)
        assert len(lv) == len(asap) and sorted(i for l in lv for i in l) == sorted(i for l in asap for i in l)
        where = {i: k for k, l in enumerate(lv) for i in l}
        root = nl.roots()
        for i, k in where.items():
            for j in nl.ins[i]:
                j = root[j]
                assert nl.kinds[j] in ("INPUT", "DFF") or where[j] < k, (i, j)
    cost = F.mi355x_level_cost
    price = F.with_sub_pass_shape(cost)
    big, small = cost.quanta
    assert price(small) == cost(small) and price(small // 4) < price(small // 2) < price(small) and price(small + 1) == cost(small + 1)
    new, old = F.plan_levels(nl, world), F.plan_levels(nl, world, spread=False)
    total = lambda lv: sum(price(r) for r in F.level_rotations(nl, lv, world))
    assert total(new) <= total(old)
    if world == 1:
        assert max(F.level_rotations(nl, new, 1)) <= small // 2 and sum(1 for r in F.level_rotations(nl, old, 1) if r == small) >= 8
        assert total(new) < 0.99 * total(old)      # measured: 0.1073 -> 0.1057 s per clock, same box
    # a netlist whose WIDTH sets the clock keeps its plan of whole rounds
    ram = N.load_iyokanl1_json(gold("mux-ram-8-16-16.min.json"))
    assert F.level_rotations(ram, F.plan_levels(ram, 1), 1) == F.level_rotations(ram, F.plan_levels(ram, 1, spread=False), 1)


def test_balanced_plan_gains_on_the_benchmark_netlists():
    """What the planner is for (model milliseconds per clock on one GPU, profiles/r03_bench_netlist*.txt has the measured
    ones): config #4's system, config #3's RAM and the CAHP core lose an eighth of their clock."""
    from iyokan_amd import frontier as F
    from iyokan_amd.system import load_blueprint

    for nl, least in ((load_blueprint(gold("cahp-ruby-mux.toml")).nl, 0.12), (N.load_iyokanl1_json(gold("mux-ram-8-16-16.min.json")), 0.12),
                      (N.load_yosys_json(gold("cahp-ruby-core-yosys.json")), 0.12)):
        cost = lambda lv: sum(F.mi355x_level_cost(r) for r in F.level_rotations(nl, lv, 1))
        assert cost(F.plan_levels(nl, 1)) <= (1 - least) * cost(nl.levelise())


def test_planner_follows_the_library_cost_table():
    """the level-cost figures live in the library (include/iyokan_hip.h: iyk_level_cost), the planner
    holds none.  (1) The default cost IS the library's compiled-in table, stamped with the build it describes; (2) a PERTURBED
    table — another GPU: 96 CUs, other milliseconds, another cross-over — yields a valid plan that is cheapest under THAT
    table and cuts levels at ITS steps, not at 256 / 2048."""
    from iyokan_amd import frontier as F
    from iyokan_amd import hip

    t = hip.level_cost_defaults()
    assert t["round"] == 8 * t["pass"] and 0 < t["max_passes"] <= 8 and t["build_id"] == hip.build_id() and not t["calibrated"]
    assert F.mi355x_level_cost.table == t and F.mi355x_level_cost.quanta == (t["round"], t["pass"])
    assert F.mi355x_level_cost(t["round"]) == pytest.approx(t["round_ms"])
    assert F.mi355x_level_cost(t["pass"] + 1) == pytest.approx(t["pass_ms"][1])
    assert F.mi355x_level_cost(t["round"] + t["max_passes"] * t["pass"] + 1) == pytest.approx(2 * t["round_ms"])

    odd = dict(t, round=768, max_passes=2, round_ms=9.0, pass_ms=[4.0, 7.5, 11.0, 14.5, 18.0, 21.5, 25.0, 28.5])
    odd["pass"] = 96
    cost = F.make_level_cost(odd)
    nl = N.load_yosys_json(gold("cahp-ruby-core-yosys.json"))
    asap = nl.levelise()
    plan = F.plan_levels(nl, 1, cost)
    assert len(plan) == len(asap) and sorted(i for lv in plan for i in lv) == sorted(i for lv in asap for i in lv)
    where = {i: k for k, lv in enumerate(plan) for i in lv}
    root = nl.roots()
    for i, k in where.items():
        for j in nl.ins[i]:
            j = root[j]
            assert nl.kinds[j] in ("INPUT", "DFF") or where[j] < k
    total = lambda lv, c: sum(c(r) for r in F.level_rotations(nl, lv, 1))
    assert total(plan, cost) < total(asap, cost)                       # it optimised the table it was given ...
    default_plan = F.plan_levels(nl, 1)
    assert total(plan, cost) <= total(default_plan, cost) + 1e-9       # ... better than the plan made for the default table
    rots = F.level_rotations(nl, plan, 1)
    assert sum(1 for r in rots if r and r % 96 == 0) > sum(1 for r in F.level_rotations(nl, default_plan, 1) if r and r % 96 == 0)
