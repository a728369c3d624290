"""Run a blueprint system (or a bare netlist) through the reference's clocking protocol.

Mirrors the frontends' `go()` (/root/reference/src/iyokan_plain.cpp:453-555 and, identically for the
GPU backend, /root/reference/src/iyokan_cufhe.cpp:754-832):

    ROM images are in place before anything runs;
    if the system has an @reset input (and reset is not skipped): reset = 1, run the combinational logic;
    for cycle in 0 .. N-1  (N < 0: until @finflag reads 1):
        tick                               # every DFF / RAM cell latches its input
        cycle 0 only: reset = 0, then the request packet's RAM images are written (setInitialRAM)
        @inputs take their bit stream, circularly:  bit (width * cycle + i) mod len   (setCircularInputs)
        run the combinational logic
    result packet = every @output, every RAM image, and the number of cycles run.

The engine behind it is anything with `set_nodes / get_nodes / run / tick` working on BITS: `PlainEngine` (numpy evaluator) or `CipherEngine` (a `FrontierExecutor` whose
values are TLWE ciphertexts, wrapped with an encrypt / decrypt pair — the GPU path).
"""
import numpy as np

from . import netlist as N
from .packet import PlainPacket


class PlainEngine:
    def __init__(self, nl):
        self.nl = nl
        self.sim = N.PlainSimulator(nl)

    def set_nodes(self, nids, bits):
        for nid, v in zip(nids, bits):
            self.sim.val[nid] = v

    def get_nodes(self, nids):
        return [self.sim.node_value(i) for i in nids]

    def run(self):
        self.sim.evaluate()

    def tick(self):
        self.sim.tick()


class CipherEngine:
    """A FrontierExecutor (ciphertext slots) seen as a bit engine: `encrypt(bits) -> rows`,
    `decrypt(rows) -> bits`.  State cells start as trivial 0 like TaskCUFHEGateDFF's constructor
    (/root/reference/src/iyokan_cufhe.hpp:108-133)."""

    def __init__(self, executor, encrypt, decrypt, zero_row):
        self.ex, self.encrypt, self.decrypt = executor, encrypt, decrypt
        self.nl = executor.plan.nl
        plan = executor.plan
        cells = list(plan.dffs) + list(plan.sources)   # state cells and not-yet-driven inputs read as 0
        if cells:
            executor.be.write_many([plan.slot[i] for i in cells], np.tile(zero_row, (len(cells), 1)))

    def set_nodes(self, nids, bits):
        if len(nids):
            slot = self.ex.plan.slot
            self.ex.be.write_many([slot[i] for i in nids], self.encrypt([int(b) for b in bits]))

    def get_nodes(self, nids):
        if not len(nids):
            return []
        slot = self.ex.plan.slot
        return [int(b) for b in self.decrypt(self.ex.be.read_many([slot[i] for i in nids]))]

    def run(self):
        self.ex.run()

    def tick(self):
        self.ex.tick()


def _at_width(system, nl, name):
    widths = getattr(system, "at_widths", None)
    if widths and name in widths:
        return widths[name]
    return nl.port_width(nl.inputs, name)


def run_packet(system, request, cycles=None, engine=None, skip_reset=False, on_cycle=None):
    """Run `system` (a `System` from `load_blueprint`, or a bare `Netlist`) on the request packet.
    `cycles`: None -> the packet's own `cycles`, else -1 -> until @finflag.  Returns the result packet."""
    nl = getattr(system, "nl", system)
    roms = getattr(system, "rom", {})
    rams = getattr(system, "ram", {})
    if engine is None:
        engine = PlainEngine(nl)
    if cycles is None:
        cycles = request.cycles if request.cycles is not None else -1

    def set_input(port, bit, v):
        engine.set_nodes([nl.inputs[(port, bit)]], [v])

    def get_output(port, bit):
        return engine.get_nodes([nl.outputs[(port, bit)]])[0]

    for name, cells in roms.items():          # ROM contents exist from the start
        image = request.rom.get(name)
        if image is not None:
            order = sorted(cells)
            engine.set_nodes([cells[i] for i in order], [image[i] if i < len(image) else 0 for i in order])

    if "reset" in request.bits:
        raise ValueError("@reset cannot be set by user's input")
    has_reset = ("reset", 0) in nl.inputs
    negate_reset = False
    if has_reset and not skip_reset:
        set_input("reset", 0, 1)
        engine.run()
        negate_reset = True

    has_finflag = ("finflag", 0) in nl.outputs
    if cycles < 0 and not has_finflag:
        raise ValueError("the number of cycles is unspecified and the system has no @finflag")
    done = 0
    while cycles < 0 or done < cycles:
        engine.tick()
        if done == 0:
            if negate_reset:
                set_input("reset", 0, 0)
            for name, cells in rams.items():
                image = request.ram.get(name)
                if image is None:
                    continue
                if len(image) != len(cells):
                    raise ValueError("Invalid request packet: wrong length of RAM")
                order = sorted(cells)
                engine.set_nodes([cells[i] for i in order], [image[i] for i in order])
        nids, vals = [], []
        for (port, bit), nid in nl.inputs.items():
            stream = request.bits.get(port)
            if stream:
                width = _at_width(system, nl, port)
                nids.append(nid)
                vals.append(stream[(width * done + bit) % len(stream)])
        engine.set_nodes(nids, vals)
        engine.run()
        done += 1
        if on_cycle is not None:
            on_cycle(done, engine)
        if cycles < 0 and get_output("finflag", 0) == 1:
            break
    return result_packet(system, engine, done)


def result_packet(system, engine, cycles):
    """makeResPacket (/root/reference/src/iyokan_plain.cpp:174-224): every @output port, every RAM image."""
    nl = getattr(system, "nl", system)
    res = PlainPacket(cycles=cycles)
    keys = sorted(nl.outputs)
    vals = engine.get_nodes([nl.outputs[k] for k in keys])
    for name in sorted({p for (p, _) in keys}):
        res.bits[name] = [0] * nl.port_width(nl.outputs, name)
    for (name, b), v in zip(keys, vals):
        res.bits[name][b] = int(v)
    for name, cells in getattr(system, "ram", {}).items():
        order = sorted(cells)
        res.ram[name] = [int(v) for v in engine.get_nodes([cells[i] for i in order])]
    return res
