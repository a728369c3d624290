"""BASELINE config #4 system (CAHP-ruby core + MUX ROM 7x32 + MUX RAM 8x16x16) assembled from the
reference's blueprint and run in plaintext against its fixture test09-ruby (test.rb:387-424)."""
import os

import numpy as np

from iyokan_amd import netlist as N
from iyokan_amd.frontier import FrontierExecutor, FrontierPlan, PlainBitBackend
from iyokan_amd.system import load_blueprint, make_rom_with_mux
from netlist_util import gold, load_packet


def load_cahp():
    return load_blueprint(gold("cahp-ruby-mux.toml"))


def packet_memories(sysm, req):
    """(node id -> bit) for ROM cells and RAM cells from the request packet's [[rom]] / [[ram]] images."""
    init = {}
    for kind, table in (("rom", sysm.rom), ("ram", sysm.ram)):
        for entry in req.get(kind, []):
            cells = table[entry["name"]]
            bits = N.bits_from_bytes(entry["bytes"], entry["size"])
            for idx, nid in cells.items():
                init[nid] = bits[idx] if idx < len(bits) else 0
    return init


def run_system_plain(sysm, req, cycles):
    nl = sysm.nl
    sim = N.PlainSimulator(nl)
    mem = packet_memories(sysm, req)
    rom_nodes = {nid for cells in sysm.rom.values() for nid in cells.values()}
    for nid, v in mem.items():
        if nid in rom_nodes:
            sim.val[nid] = v
    sim.set_input("reset", 0, 1)
    sim.evaluate()
    for c in range(cycles):
        sim.tick()
        if c == 0:
            sim.set_input("reset", 0, 0)
            for nid, v in mem.items():            # setInitialRAM happens after the first tick
                if nid not in rom_nodes:
                    sim.val[nid] = v
        sim.evaluate()
    return sim


def test_rom_generator_shape():
    rom = make_rom_with_mux(7, 32)
    assert rom.counts()["MUX"] == 4064 and len(rom.rom) == 4096 and len(rom.levelise()) == 7


import pytest


@pytest.mark.parametrize("core,rot", [("ruby", 4281), ("pearl", 3662)])
def test_cahp_system_matches_test09(core, rot):
    sysm = load_blueprint(gold(f"cahp-{core}-mux.toml"))
    nl = sysm.nl
    assert nl.rotations() == rot + 8128 + 18985     # core + ROM + RAM
    req = load_packet(gold("test09.in"))
    want = load_packet(gold(f"test09-{core}.out"))
    sim = run_system_plain(sysm, req, want["cycles"])
    for entry in want["bits"]:
        name, size = entry["name"], entry["size"]
        got = N.bytes_from_bits([sim.get_output(name, b) for b in range(size)])
        assert got == entry["bytes"], name
    ram_bits = [sim.node_value(sysm.ram["ram"][i]) for i in range(4096)]
    assert N.bytes_from_bits(ram_bits) == want["ram"][0]["bytes"]


def test_cahp_system_frontier_plan_matches_simulator():
    """The merged system through the sharded executor (bit backend, 3 ranks' worth of plan on 1 rank
    is not meaningful, so world = 1) equals the simulator for 3 clocks."""
    sysm = load_cahp()
    nl = sysm.nl
    req = load_packet(gold("test09.in"))
    mem = packet_memories(sysm, req)
    plan = FrontierPlan(nl, 1)
    ex = FrontierExecutor(plan, PlainBitBackend(plan.num_slots))
    sim = N.PlainSimulator(nl)
    rom_nodes = {nid for cells in sysm.rom.values() for nid in cells.values()}
    for nid, v in mem.items():
        if nid in rom_nodes:
            ex.set_node(nid, v); sim.val[nid] = v
    ex.set_input("reset", 0, 1); sim.set_input("reset", 0, 1)
    ex.run(); sim.evaluate()
    for c in range(3):
        ex.tick(); sim.tick()
        if c == 0:
            ex.set_input("reset", 0, 0); sim.set_input("reset", 0, 0)
            for nid, v in mem.items():
                if nid not in rom_nodes:
                    ex.set_node(nid, v); sim.val[nid] = v
        ex.run(); sim.evaluate()
        for key in nl.outputs:
            assert ex.get_output(*key) == sim.get_output(*key), (c, key)
    assert len(plan.levels) >= 41
