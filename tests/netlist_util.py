"""Test helpers: run a netlist through the reference's request/result packet protocol
(/root/reference/src/iyokan_plain.cpp:452-548: optional reset cycle; per cycle tick -> inputs -> run)."""
import os

import tomli

from iyokan_amd import netlist as N

REFTEST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reftest")


def gold(name):
    """Path of a fixture file by base name; tests/golden/reftest mirrors the layout of the reference's test/ tree."""
    for sub in ("", "config-toml", "yosys-json", "iyokanl1-json", "in", "out"):
        p = os.path.join(REFTEST, sub, name)
        if os.path.exists(p):
            return p
    raise FileNotFoundError(name)


def load_packet(path):
    with open(path, "rb") as f:
        return tomli.load(f)


def input_streams(req):
    return {b["name"]: N.bits_from_bytes(b["bytes"], b["size"]) for b in req.get("bits", [])}


def drive_cycle(setter, nl, streams, cycle):
    """setCircularInputs: port bit i at cycle c takes stream bit (width*c + i) mod len."""
    for (port, bit) in nl.inputs:
        if port in streams:
            width = nl.port_width(nl.inputs, port)
            s = streams[port]
            setter(port, bit, s[(width * cycle + bit) % len(s)])


def run_plain(nl, streams, cycles, use_reset):
    sim = N.PlainSimulator(nl)
    if use_reset:
        sim.set_input("reset", 0, 1)
        sim.evaluate()
    for c in range(cycles):
        sim.tick()
        if c == 0 and use_reset:
            sim.set_input("reset", 0, 0)
        drive_cycle(sim.set_input, nl, streams, c)
        sim.evaluate()
    return sim
