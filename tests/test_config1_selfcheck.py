"""BASELINE config #1: the reference's test0 self-check net (src/test0.cpp: the circuits read from
Iyokan-L1 JSON :157-334, the counter :403-431) evaluated ENCRYPTED on the CPU — oracle as the gate
evaluator behind the same executor / runner the GPU path uses — with the reference's known answers.
Plumbing only: no GPU, no timing."""
import numpy as np
import pytest

from iyokan_amd import client
from iyokan_amd import netlist as N
from iyokan_amd.frontier import FrontierExecutor, FrontierPlan
from iyokan_amd.runner import CipherEngine
from netlist_util import gold
from oracle_lib import OracleBackend


class Bench:
    """One netlist under encryption: set / run / tick / get on integers."""

    def __init__(self, nl, keys, oracle):
        self.nl, self.keys = nl, keys
        plan = FrontierPlan(nl, 1)
        self.seed = 9000
        self.eng = CipherEngine(FrontierExecutor(plan, OracleBackend(plan.num_slots, oracle)), self._enc,
                                lambda rows: client.decrypt_bits(keys, rows), client.trivial(keys.params, 0))

    def _enc(self, bits):
        self.seed += 1
        return client.encrypt_bits(self.keys, bits, seed=self.seed)

    def set(self, port, value):
        bits = [(p, b) for (p, b) in self.nl.inputs if p == port]
        self.eng.set_nodes([self.nl.inputs[k] for k in bits], [(value >> b) & 1 for (_, b) in bits])

    def get(self, port):
        bits = sorted((p, b) for (p, b) in self.nl.outputs if p == port)
        vals = self.eng.get_nodes([self.nl.outputs[k] for k in bits])
        return sum(v << b for v, (_, b) in zip(vals, bits))

    def run(self):
        self.eng.run()

    def tick(self):
        self.eng.tick()


def l1(name):
    return N.load_iyokanl1_json(gold(f"{name}-iyokanl1.json"))


def test_pass_and_and42(keys128, oracle128):
    b = Bench(l1("pass-4bit"), keys128, oracle128)
    b.set("io_in", 0b0110); b.run()
    assert b.get("io_out") == 0b0110
    b = Bench(l1("and-4bit"), keys128, oracle128)
    b.set("io_inA", 0b1100); b.set("io_inB", 0b1010); b.run()
    assert b.get("io_out") == 0b1000
    b = Bench(l1("and-4_2bit"), keys128, oracle128)
    b.set("io_inA", 0b1101); b.set("io_inB", 0b1111); b.run()
    assert b.get("io_out") == 0b01


def test_mux_4bit(keys128, oracle128):
    b = Bench(l1("mux-4bit"), keys128, oracle128)
    b.set("io_inA", 0b1100); b.set("io_inB", 0b1010)
    b.set("io_sel", 0); b.run()
    assert b.get("io_out") == 0b1100
    b.tick()
    b.set("io_sel", 1); b.run()
    assert b.get("io_out") == 0b1010


@pytest.mark.parametrize("loader,fname", [(N.load_iyokanl1_json, "addr-4bit-iyokanl1.json"),
                                          (N.load_yosys_json, "addr-4bit-yosys.json")])
def test_addr_4bit(loader, fname, keys128, oracle128):
    b = Bench(loader(gold(fname)), keys128, oracle128)
    b.set("io_inA", 0b1100); b.set("io_inB", 0b1010); b.run()
    assert b.get("io_out") == 0b0110


def test_register_4bit(keys128, oracle128):
    b = Bench(l1("register-4bit"), keys128, oracle128)
    b.set("io_in", 0b1100)
    # outputs are read after `run` (the executor's OUTPUT wires alias their driver's slot, so after a tick
    # they already show the latched value; the reference's WIRE tasks show it after the next run)
    b.set("reset", 1); b.run(); b.tick()
    b.set("reset", 0); b.run()
    assert b.get("io_out") == 0
    b.tick(); b.run()
    assert b.get("io_out") == 0b1100


def test_counter_4bit(keys128, oracle128):
    b = Bench(l1("counter-4bit"), keys128, oracle128)
    b.set("reset", 1); b.run()
    b.set("reset", 0)
    for clk in range(5):
        b.tick(); b.run()
        assert b.get("io_out") == clk
