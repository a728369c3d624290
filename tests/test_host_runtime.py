"""CPU plumbing (BASELINE config #1 shape): the C++ engine + plaintext backend self-test builds and passes,
including the circuits read from Iyokan-L1 / Yosys JSON by the C++ readers (host/readers.hpp)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_engine_plain_backend():
    host = os.path.join(ROOT, "iyokan_amd", "host")
    subprocess.run(["make", "-C", host], check=True, capture_output=True)
    fixtures = os.path.join(ROOT, "tests", "golden", "reftest")   # the JSON circuits of the reference's test0
    out = subprocess.run([os.path.join(host, "test0_hip"), "--fixtures", fixtures], capture_output=True, text=True,
                         timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "plain: all tests passed" in out.stdout


def test_reader_rejects_bad_input(tmp_path):
    """Error convention of the readers = the reference's error::die: message on stderr, exit status 1."""
    host = os.path.join(ROOT, "iyokan_amd", "host")
    bad = tmp_path / "iyokanl1-json"
    bad.mkdir()
    for name in ("pass-4bit", "and-4bit", "and-4_2bit", "mux-4bit", "addr-4bit", "register-4bit", "counter-4bit"):
        (bad / f"{name}-iyokanl1.json").write_text('{"ports": [], "cells": [{"type": "FOO", "id": 1}]}')
    out = subprocess.run([os.path.join(host, "test0_hip"), "--fixtures", str(tmp_path)], capture_output=True, text=True,
                         timeout=120)
    assert out.returncode == 1
    assert "Invalid type: FOO" in out.stdout + out.stderr


# ---- the C++ frontend (host/toml.hpp, packet.hpp, blueprint.hpp, plain_frontend.hpp) on the reference's vectors --------
import sys

import numpy as np
import pytest

from iyokan_amd import client

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from test_reference_vectors import CASES  # noqa: E402  (the table restating test.rb:385-540)


def _exe():
    host = os.path.join(ROOT, "iyokan_amd", "host")
    subprocess.run(["make", "-C", host], check=True, capture_output=True)
    return os.path.join(host, "test0_hip")


@pytest.mark.parametrize("name,blueprint,req,want,ncycles", CASES, ids=[c[0] for c in CASES])
def test_cpp_plain_frontend_reference_vector(name, blueprint, req, want, ncycles, tmp_path):
    """Blueprint (TOML) -> one network, request packet (TOML), reset / RAM / circular-input protocol, result packet:
    all in C++, compared with the reference's expected packet the way its test driver does."""
    from iyokan_amd.packet import PlainPacket
    from netlist_util import gold

    out = subprocess.run([_exe(), "--plain-run", gold(blueprint), gold(req), "-c", str(ncycles)], capture_output=True,
                         text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    p = tmp_path / "res.toml"
    p.write_text(out.stdout)
    got, expected = PlainPacket.load(str(p)), PlainPacket.load(gold(want))
    assert got.same_content(expected), got.diff(expected)


def test_cpp_frontend_error_convention():
    from netlist_util import gold

    out = subprocess.run([_exe(), "--plain-run", gold("counter-4bit.toml"), gold("test03.in")], capture_output=True, text=True,
                         timeout=60)
    assert out.returncode == 1 and "the number of cycles is unspecified" in out.stderr
    out = subprocess.run([_exe(), "--plain-run", "/nonexistent.toml", gold("test03.in"), "-c", "1"], capture_output=True,
                         text=True, timeout=60)
    assert out.returncode == 1 and "Can't open the file" in out.stderr


def test_cereal_portable_binary_archives(tmp_path):
    """The packet wire format (/root/reference/src/packet.hpp:287-344), both implementations: C++ self-test (round trips
    + a hand-assembled archive), then the archive the C++ side wrote is read by the Python reader and re-written
    byte-identically (entries sorted by name on both sides)."""
    from iyokan_amd.packet import PlainPacket
    from netlist_util import gold

    arc = tmp_path / "test09.bin"
    out = subprocess.run([_exe(), "--packet-selftest", gold("test09.in"), str(arc)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "packet self-test ok" in out.stdout, out.stdout + out.stderr
    data = arc.read_bytes()
    pkt = PlainPacket.from_archive(data)
    assert pkt.same_content(PlainPacket.load(gold("test09.in")))
    resorted = PlainPacket(bits=dict(sorted(pkt.bits.items())), ram=dict(sorted(pkt.ram.items())),
                           rom=dict(sorted(pkt.rom.items())), cycles=pkt.cycles)
    assert resorted.to_archive() == data
    hand = bytes([1] + [0] * 8 + [0] * 8 + [1] + [0] * 7 + [2] + [0] * 7 + list(b"ab") + [3] + [0] * 7 + [1, 0, 1] + [0, 5, 0, 0, 0])
    h = PlainPacket.from_archive(hand)
    assert h.bits == {"ab": [1, 0, 1]} and h.cycles == 5 and h.ram == {} and h.rom == {}
    assert h.to_archive() == hand
    for bad in (hand[:-1], hand + b"\0", bytes([2]) + hand[1:], hand[:17] + bytes([255] * 8) + hand[25:]):
        with pytest.raises(ValueError, match="Invalid archive"):
            PlainPacket.from_archive(bad)


def test_tfhe_packet_archive_round_trip(keys80):
    """PlainPacket::encrypt fills BOTH forms of every memory image (/root/reference/src/packet.hpp:225-262): TLWE lvl0 rows and
    TRLWE lvl1 — `ram`: one ciphertext per bit, `rom`: N bits per ciphertext; TFHEPacket::decrypt prefers the TRLWE maps, so a
    ROM comes back padded to a multiple of N (as upstream); without the TRLWE maps the sizes are exact."""
    from iyokan_amd.packet import PlainPacket, TFHEPacket

    p = keys80.params
    plain = PlainPacket(bits={"x": [1, 0, 1, 1], "reset": [0]}, ram={"ram": [0, 1] * 8}, rom={"rom": [1, 0, 0, 1, 1]}, cycles=3)
    enc = TFHEPacket.encrypt(keys80, plain, seed=40)
    data = enc.to_archive()
    back = TFHEPacket.from_archive(p, data)
    assert back.to_archive() == data
    assert set(back.ramInTLWE) == {"ram"} and back.ramInTLWE["ram"].shape == (16, p.n + 1)
    assert back.ram["ram"].shape == (16, 2 * p.N) and back.rom["rom"].shape == (1, 2 * p.N)
    dec = back.decrypt(keys80)
    assert dec.bits == plain.bits and dec.ram == plain.ram and dec.cycles == 3
    assert len(dec.rom["rom"]) == p.N and dec.rom["rom"][:5] == plain.rom["rom"]
    tl = TFHEPacket.encrypt(keys80, plain, seed=40, trlwe=False)
    assert tl.ram == {} and tl.rom == {} and TFHEPacket.from_archive(p, tl.to_archive()).decrypt(keys80).same_content(plain)
    # the TRLWE form alone (what an upstream CMUX-memory run returns for a RAM): still decrypts
    only = TFHEPacket(p, cycles=1)
    only.ram["ram"] = enc.ram["ram"]
    assert TFHEPacket.from_archive(p, only.to_archive()).decrypt(keys80).ram == plain.ram


def test_tfhe_packet_hand_assembled_archive_both_readers(keys128, tmp_path):
    """a reader pinned only by its own writer proves nothing about the FORMAT.  This archive is put
    together here, field by field, from cereal's PortableBinary rules with struct.pack — endianness byte; per unordered_map a
    u64 entry count, per entry a u64 string length + bytes and a u64 element count + the raw std::array words; the
    std::optional<int> as a bool byte + i32 — in serialize() order ar(ram, ramInTLWE, rom, romInTLWE, bits, numCycles)
    (/root/reference/src/packet.hpp:208-223), without TFHEPacket.to_archive.  Both readers (Python, C++) must accept it and
    decrypt the same bits; the Python writer must reproduce it byte for byte."""
    import struct

    from iyokan_amd.packet import TFHEPacket

    p = keys128.params
    u64 = lambda v: struct.pack("<Q", v)
    ram_bits, rom_bits, x_bits = [1, 0, 1], [1, 1, 0, 1, 0, 0, 1, 0], [0, 1, 1, 0, 1]
    ram_t = client.encrypt_ram_trlwe(keys128, ram_bits, seed=1)          # (3, 2N) words
    rom_t = client.encrypt_rom_trlwe(keys128, rom_bits, seed=2)          # (1, 2N)
    ram_l = client.encrypt_bits(keys128, ram_bits, seed=3)               # (3, n + 1)
    x_l = client.encrypt_bits(keys128, x_bits, seed=4)
    words = lambda a: np.ascontiguousarray(a, dtype="<u4").tobytes()
    hand = b"\x01"                                                       # little endian
    hand += u64(1) + u64(4) + b"mem0" + u64(3) + words(ram_t)            # ram: {"mem0": 3 x TRLWELvl1}
    hand += u64(1) + u64(4) + b"mem0" + u64(3) + words(ram_l)            # ramInTLWE
    hand += u64(1) + u64(3) + b"rom" + u64(1) + words(rom_t)             # rom: 8 bits in one TRLWE
    hand += u64(0)                                                       # romInTLWE: empty
    hand += u64(1) + u64(1) + b"x" + u64(5) + words(x_l)                 # bits
    hand += b"\x00" + struct.pack("<i", 7)                               # numCycles = 7 (cereal: bool nullopt, then the value)
    pkt = TFHEPacket.from_archive(p, hand)
    assert pkt.cycles == 7 and pkt.to_archive() == hand
    dec = pkt.decrypt(keys128)
    assert dec.ram == {"mem0": ram_bits} and dec.bits == {"x": x_bits} and dec.rom["rom"][:8] == rom_bits
    for cut in (hand[:-1], hand + b"\0", hand[:9] + u64(1 << 40) + hand[17:]):
        with pytest.raises(ValueError, match="Invalid archive"):
            TFHEPacket.from_archive(p, cut)
    # the C++ reader on the same bytes, with a secret-key archive assembled the same way (KeyArchive: 8 u32 parameters, two
    # f64, then s0, s1, bk, ksk as u64-counted u32 vectors; bk / ksk empty in a secret-key archive)
    sk = b"\x01" + struct.pack("<8I", p.n, p.N, p.k, p.l, p.Bgbit, p.t, p.basebit, p.mu) + struct.pack("<2d", p.alpha0, p.alpha1)
    sk += u64(p.n) + words(keys128.s0) + u64(p.N) + words(keys128.s1) + u64(0) + u64(0)
    (tmp_path / "sk.bin").write_bytes(sk)
    (tmp_path / "pkt.bin").write_bytes(hand)
    out = subprocess.run([_exe(), "--tfhe-packet-read", str(tmp_path / "sk.bin"), str(tmp_path / "pkt.bin")],
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = out.stdout.strip().splitlines()
    assert lines[0] == "ok ram=1 ramInTLWE=1 rom=1 romInTLWE=0 bits=1 cycles=7"
    assert "ram mem0 101" in lines and "bits x 01101" in lines
    assert [l for l in lines if l.startswith("rom rom ")][0].split()[2].startswith("11010010")


def test_hostile_size_tags_are_refused_before_allocation(tmp_path):
    """a size tag is checked against the bytes the stream still holds BEFORE anything is allocated — a
    truncated or hostile archive ends in die("Invalid archive: ..."), not in a 4 GiB vector, bad_alloc or std::terminate."""
    import struct

    def u64(v):
        return struct.pack("<Q", v)

    good = b"\x01" + u64(0) + u64(0) + u64(1) + u64(1) + b"x" + u64(2) + b"\x01\x00" + b"\x01"   # bits = {x: [1, 0]}, no cycles
    cases = {
        "good": (good, 0, "ok 0 0 1"),
        "huge bit vector": (b"\x01" + u64(1) + u64(3) + b"abc" + u64(1 << 31), 1, "size tag exceeds the remaining bytes"),
        "huge key": (b"\x01" + u64(1) + u64((1 << 20) - 1), 1, "size tag exceeds the remaining bytes"),
        "million entries": (b"\x01" + u64(1 << 20), 1, "size tag exceeds the remaining bytes"),
        "trailing": (good + b"\x00", 1, "trailing bytes"),
        "truncated": (good[:-3], 1, "Invalid archive"),
    }
    for name, (data, rc, text) in cases.items():
        f = tmp_path / "a.bin"
        f.write_bytes(data)
        out = subprocess.run([_exe(), "--packet-read", str(f)], capture_output=True, text=True, timeout=60)
        assert out.returncode == rc, (name, out.returncode, out.stdout, out.stderr)
        assert text in out.stdout + out.stderr, (name, out.stdout, out.stderr)


def _plan_graph_file(nl, tmp_path):
    from iyokan_amd import frontier as F

    depth, order, succ, npred, alap, rot = F._slack_graph(nl)
    lines = [f"{nl.num_nodes} {depth} 6"]
    placed = set(order)
    for i in range(nl.num_nodes):
        if i in placed:
            lines.append(" ".join(map(str, [rot[i], alap[i] - 1, npred[i], len(succ[i])] + succ[i])))
        else:
            lines.append("0 0 0 0")
    f = tmp_path / "graph.txt"
    f.write_text("\n".join(lines) + "\n")
    return f


@pytest.mark.parametrize("world", [1, 8])
def test_cpp_spread_plan_equals_the_python_one(world, tmp_path):
    """Round 6's capped list schedule and the choice among the candidates, C++ (host/iyokan_hip.hpp: cappedLevels, planBest — what
    planFrontiers runs) against Python (frontier.capped_levels, plan_levels) on the CAHP core: the same frontier for every node at a
    cap of 128 rotations per GPU, the same price, and planBest spreads the core exactly as plan_levels does."""
    from iyokan_amd import frontier as F
    from iyokan_amd import netlist as N
    from netlist_util import gold

    nl = N.load_yosys_json(gold("cahp-ruby-core-yosys.json"))
    f = _plan_graph_file(nl, tmp_path)
    price = F.with_sub_pass_shape(F.mi355x_level_cost)
    total = lambda lv: sum(price(r) for r in F.level_rotations(nl, lv, world))

    def run(*extra):
        out = subprocess.run([_exe(), "--plan-graph", str(f), "--gpus", str(world), *extra], capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stdout[-500:] + out.stderr[-2000:]
        rows = out.stdout.split()
        return float(rows[1]), [int(x) for x in rows[2:]]

    ms, rnd = run("--capped", "128")
    levels = F.capped_levels(nl, world, 128)
    for k, lv in enumerate(levels):
        for i in lv:
            assert rnd[i] == k, (i, k, rnd[i])
    assert ms == pytest.approx(total(levels), rel=1e-9)
    ms, rnd = run("--plan-best")
    best = F.plan_levels(nl, world)
    assert ms <= total(best) * (1 + 1e-9)                    # C++ offers fewer candidates than Python, never a dearer choice here
    if world == 1:
        for k, lv in enumerate(best):
            for i in lv:
                assert rnd[i] == k, (i, k, rnd[i])


@pytest.mark.parametrize("name,kind", [("cahp-ruby-core-yosys.json", "yosys"), ("mux-ram-8-16-16.min.json", "l1"),
                                       ("cahp-ruby-mux.toml", "blueprint")])
@pytest.mark.parametrize("world", [1, 2])
@pytest.mark.parametrize("tie", ["id", "fanout"])
def test_cpp_planner_equals_the_python_planner(name, kind, world, tie, tmp_path):
    """host/iyokan_hip.hpp planLevels (what planFrontiers runs; partial schedules kept as deltas) against
    frontier.beam_levels on the benchmark netlists: the same DAG in the same numbering, the same compiled-in cost table (no
    GPU, so both ask the library for its defaults), the same search order — the same frontier for every node, the same
    milliseconds."""
    from iyokan_amd import frontier as F
    from iyokan_amd import netlist as N
    from netlist_util import gold

    if kind == "blueprint":
        from iyokan_amd.system import load_blueprint

        nl = load_blueprint(gold(name)).nl
    else:
        nl = (N.load_iyokanl1_json if kind == "l1" else N.load_yosys_json)(gold(name))
    depth, order, succ, npred, alap, rot = F._slack_graph(nl)
    lines = [f"{nl.num_nodes} {depth} 6"]          # nodes without a level (sources) are listed as already-placed nothing
    placed = set(order)
    for i in range(nl.num_nodes):
        if i in placed:
            lines.append(" ".join(map(str, [rot[i], alap[i] - 1, npred[i], len(succ[i])] + succ[i])))
        else:
            lines.append("0 0 0 0")
    f = tmp_path / "graph.txt"
    f.write_text("\n".join(lines) + "\n")
    out = subprocess.run([_exe(), "--plan-graph", str(f), "--gpus", str(world)] + (["--tie-fanout"] if tie == "fanout" else []),
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-500:] + out.stderr[-2000:]
    rows = out.stdout.split()
    assert rows[0] == "ms"
    cpp_ms, cpp_round = float(rows[1]), [int(x) for x in rows[2:]]
    levels = F.beam_levels(nl, world, tie=tie)
    assert len(cpp_round) == nl.num_nodes
    for k, lv in enumerate(levels):
        for i in lv:
            assert cpp_round[i] == k, (i, k, cpp_round[i])
    assert cpp_ms == pytest.approx(sum(F.mi355x_level_cost(r) for r in F.level_rotations(nl, levels, world)), rel=1e-9)
