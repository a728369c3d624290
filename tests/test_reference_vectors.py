"""The reference's own integration vectors (test.rb:385-540: blueprint, request packet, expected result
packet, number of cycles), run through this package's blueprint loader + clocking protocol + plaintext
evaluator.  These pin the CALLER side of the hot path (netlist / blueprint ingestion, reset and RAM
initialisation protocol, circular input streams, result packets); the encrypted GPU runs in
test_gpu_netlist.py use the same loader and runner and are compared against this evaluator.

ncycles < 0 means "until @finflag" (the reference runs its plain cases that way unless
`set_plain_ncycles` is given)."""
import pytest

from iyokan_amd.packet import PlainPacket
from iyokan_amd.runner import run_packet
from iyokan_amd.system import load_blueprint
from netlist_util import gold

# (test.rb name, blueprint, request, expected, ncycles for the plain run)
CASES = [
    ("cahp-diamond-00", "cahp-diamond.toml", "test00.in", "test00-diamond.out", -1),
    ("cahp-ruby-09", "cahp-ruby.toml", "test09.in", "test09-ruby.out", -1),
    ("cahp-pearl-09", "cahp-pearl.toml", "test09.in", "test09-pearl.out", -1),
    ("cahp-diamond-mux-00", "cahp-diamond-mux.toml", "test00.in", "test00-diamond.out", -1),
    ("cahp-ruby-mux-09", "cahp-ruby-mux.toml", "test09.in", "test09-ruby.out", -1),
    ("cahp-pearl-mux-09", "cahp-pearl-mux.toml", "test09.in", "test09-pearl.out", -1),
    ("cahp-diamond-01", "cahp-diamond.toml", "test01.in", "test01-diamond.out", -1),
    ("cahp-ruby-10", "cahp-ruby.toml", "test10.in", "test10-ruby.out", -1),
    ("cahp-pearl-10", "cahp-pearl.toml", "test10.in", "test10-pearl.out", -1),
    ("cahp-diamond-mux-01", "cahp-diamond-mux.toml", "test01.in", "test01-diamond.out", -1),
    ("cahp-ruby-mux-10", "cahp-ruby-mux.toml", "test10.in", "test10-ruby.out", -1),
    ("cahp-pearl-mux-10", "cahp-pearl-mux.toml", "test10.in", "test10-pearl.out", -1),
    ("const-4bit-22", "const-4bit.toml", "test22.in", "test22.out", 1),
    ("addr-4bit-04", "addr-4bit.toml", "test04.in", "test04.out", 1),
    ("pass-addr-pass-4bit-04", "pass-addr-pass-4bit.toml", "test04.in", "test04.out", 1),
    ("addr-register-4bit-16", "addr-register-4bit.toml", "test16.in", "test16.out", 3),
    ("div-8bit-05", "div-8bit.toml", "test05.in", "test05.out", 1),
    ("ram-addr8bit-06", "ram-addr8bit.toml", "test06.in", "test06.out", 16),
    ("ram-addr9bit-07", "ram-addr9bit.toml", "test07.in", "test07.out", 16),
    ("mux-ram-addr8bit-06", "mux-ram-addr8bit.toml", "test06.in", "test06.out", 16),
    ("mux-ram-addr9bit-07", "mux-ram-addr9bit.toml", "test07.in", "test07.out", 16),
    ("ram-8-16-16-08", "ram-8-16-16.toml", "test08.in", "test08.out", 8),
    ("mux-ram-8-16-16-08", "mux-ram-8-16-16.toml", "test08.in", "test08.out", 8),
    ("rom-7-32-12", "rom-7-32.toml", "test12.in", "test12.out", 1),
    ("rom-4-8-15", "rom-4-8.toml", "test15.in", "test15.out", 1),
    ("counter-4bit-13", "counter-4bit.toml", "test13.in", "test13.out", 3),
    ("cahp-ruby-14", "cahp-ruby.toml", "test14.in", "test14.out", 20),
    ("cahp-ruby-iyokanl1-09", "cahp-ruby-iyokanl1.toml", "test09.in", "test09-ruby.out", -1),
    ("dff-reset-23", "dff-reset.toml", "test23.in", "test23.out", 1),
]
# Not runnable from the reference tree itself: cahp-emerald (its Yosys netlist is not in test/yosys-json),
# big-mult-21 (netlist not in the tree), cahp-ruby-mux-1KiB-11 (2.7 MB netlist left out of the fixtures),
# register-init-18/19 (commented out upstream: $_SDFF_ cells are rejected, see test_netlist.py).


@pytest.mark.parametrize("name,blueprint,req,want,ncycles", CASES, ids=[c[0] for c in CASES])
def test_reference_vector(name, blueprint, req, want, ncycles):
    sysm = load_blueprint(gold(blueprint))
    got = run_packet(sysm, PlainPacket.load(gold(req)), cycles=ncycles)
    expected = PlainPacket.load(gold(want))
    assert got.same_content(expected), got.diff(expected)


def test_chained_runs_through_convert_plain():
    """plain-addr-addr-4bit-20 (test.rb:484-509): the result of one run, rewired by `convert-plain`
    (bits.A = a.out, bits.B = a.out), is the request of the next."""
    sysm = load_blueprint(gold("addr-4bit.toml"))
    first = run_packet(sysm, PlainPacket.load(gold("test20.in")), cycles=1)
    second_req = PlainPacket.convert({"a": first}, ["bits.A = a.out", "bits.B = a.out"])
    got = run_packet(sysm, second_req, cycles=1)
    expected = PlainPacket.load(gold("test20.out"))
    assert got.same_content(expected), got.diff(expected)


def test_convert_plain_fixture():
    """iyokan-packet convert-plain (test.rb:170-193): entries picked from three packets equal test17.in."""
    a, b, c = (PlainPacket.load(gold(f)) for f in ("test00.in", "test08.out", "test03.in"))
    got = PlainPacket.convert({"a": a, "b": b, "c": c},
                              ["rom.foo = a.rom", "ram.bar = a.ramB", "bits.baz = b.rdata", "ram.hoge = b.target",
                               "bits.piyo = c.hoge"])
    expected = PlainPacket.load(gold("test17.in"))
    assert got.same_content(expected), got.diff(expected)


@pytest.mark.parametrize("fname", ["test00.in", "test00-diamond.out", "test03.in"])
def test_packet_toml_round_trip(fname, tmp_path):
    """toml2packet -> packet2toml keeps the content (test.rb:145-166, without the enc / dec legs)."""
    pkt = PlainPacket.load(gold(fname))
    p = tmp_path / "pkt.toml"
    p.write_text(pkt.to_toml())
    assert PlainPacket.load(str(p)).same_content(pkt)


def test_toml2packet_known_answer():
    """test_method_toml2packet (test.rb:132-142)."""
    pkt = PlainPacket.load(gold("test03.in"))
    assert pkt.cycles is None and pkt.ram == {} and pkt.rom == {}
    assert pkt.bits == {"hoge": [1, 0, 1], "piyo": [0, 0, 0]}


def test_reset_port_cannot_be_driven():
    sysm = load_blueprint(gold("counter-4bit.toml"))
    with pytest.raises(ValueError, match="@reset cannot be set"):
        run_packet(sysm, PlainPacket(bits={"reset": [1]}), cycles=1)


@pytest.mark.parametrize("blueprint,req,want,ncycles", [
    ("mux-ram-addr8bit.toml", "test06.in", "test06.out", 16),
    ("cahp-pearl-mux.toml", "test09.in", "test09-pearl.out", -1),
])
def test_frontier_executor_through_runner(blueprint, req, want, ncycles):
    """The level-synchronous executor the GPU path uses, driven by the same runner, with bits standing in
    for ciphertexts (PlainBitBackend): the plumbing of test_gpu_netlist.py::test_reference_vectors_encrypted."""
    import numpy as np

    from iyokan_amd.frontier import FrontierExecutor, FrontierPlan, PlainBitBackend
    from iyokan_amd.runner import CipherEngine

    sysm = load_blueprint(gold(blueprint))
    plan = FrontierPlan(sysm.nl, 1)
    eng = CipherEngine(FrontierExecutor(plan, PlainBitBackend(plan.num_slots)),
                       lambda bits: np.array(bits, dtype=np.uint8).reshape(-1, 1),
                       lambda rows: np.asarray(rows).reshape(-1), np.zeros(1, dtype=np.uint8))
    got = run_packet(sysm, PlainPacket.load(gold(req)), cycles=ncycles, engine=eng)
    expected = PlainPacket.load(gold(want))
    assert got.same_content(expected), got.diff(expected)
