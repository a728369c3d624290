"""GPU: netlists through the frontier executor (device-resident arena, level-synchronous batches)
and the C++ host runtime self-test; decrypted results must equal the plaintext simulator."""
import os
import subprocess

import numpy as np
import pytest

from iyokan_amd import client
from iyokan_amd import netlist as N
from iyokan_amd.frontier import FrontierExecutor, FrontierPlan, HipBackend
from netlist_util import gold, drive_cycle, input_streams, load_packet

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def gpu(keys128):
    from iyokan_amd import hip

    hip.initialize(keys128, device_ids=(0,))
    yield hip
    hip.cleanup()


def _make(nl, keys, seed=100):
    import torch

    plan = FrontierPlan(nl, 1)
    be = HipBackend(plan.num_slots, keys.params, torch.device("cuda", 0))
    ex = FrontierExecutor(plan, be)
    zero = client.trivial(keys.params, 0)

    def init_state():   # DFF / RAM cells = trivial 0 (TaskCUFHEGateDFF ctor, setInitialRAM)
        be.write_many([plan.slot[i] for i in plan.dffs], np.tile(zero, (len(plan.dffs), 1)))

    init_state()
    ex.init_state = init_state
    state = {"seed": seed}

    def set_enc(port, bit, v):
        state["seed"] += 1
        ex.set_input(port, bit, client.encrypt_bits(keys, [v], seed=state["seed"])[0])

    return plan, be, ex, set_enc


def test_counter_4bit_on_gpu(gpu, keys128):
    nl = N.load_iyokanl1_json(gold("counter-4bit-iyokanl1.json"))
    plan, be, ex, set_enc = _make(nl, keys128)
    set_enc("reset", 0, 1)
    ex.run()
    set_enc("reset", 0, 0)
    for clk in range(6):
        ex.tick()
        ex.run()
        got = client.decrypt_bits(keys128, np.stack([ex.get_output("io_out", b) for b in range(4)]))
        assert sum(int(v) << b for b, v in enumerate(got)) == clk
    be.close()


def test_mux_ram_config3_two_clocks(gpu, keys128):
    """BASELINE config #3: mux-ram-8-16-16, inputs of test08 (fresh encryptions), RAM zero-initialised;
    clock 0 (18 985 blind rotations in 14 level batches) then a clock edge and clock 1."""
    nl = N.load_iyokanl1_json(gold("mux-ram-8-16-16.min.json"))
    streams = input_streams(load_packet(gold("test08.in")))
    plan, be, ex, set_enc = _make(nl, keys128)
    sim = N.PlainSimulator(nl)
    for c in range(2):
        ex.tick(); sim.tick()
        if c == 0:
            ex.init_state()   # the reference sets the initial RAM AFTER the first tick (iyokan_plain.cpp:509-511)
        drive_cycle(set_enc, nl, streams, c)
        drive_cycle(sim.set_input, nl, streams, c)
        ex.run(); sim.evaluate()
        outs = sorted(nl.outputs)
        got = client.decrypt_bits(keys128, be.read_many([plan.slot[nl.outputs[k]] for k in outs]))
        want = [sim.get_output(*k) for k in outs]
        assert list(got) == want, f"clock {c}"
    ex.tick(); sim.tick()
    ram = client.decrypt_bits(keys128, be.read_many([plan.slot[nl.ram[i]] for i in range(4096)]))
    assert list(ram) == sim.ram_image(4096)
    be.close()


def test_cahp_system_config4_on_gpu(gpu, keys128):
    """BASELINE config #4: CAHP-ruby core + MUX ROM + MUX RAM from the reference blueprint, program and RAM
    image of test09.in encrypted, TEN clocks after the reset cycle on the GPU.  After 7 clocks the outputs must be the
    reference's test09-ruby.out (@reg_x0 = 42, @finflag = 1); after 10, every output port AND every register / memory
    cell must equal the plaintext evaluator run in lockstep under the same protocol."""
    import torch

    from test_system import load_cahp, packet_memories

    sysm = load_cahp()
    nl = sysm.nl
    req = load_packet(gold("test09.in"))
    want = load_packet(gold("test09-ruby.out"))
    mem = packet_memories(sysm, req)
    rom_nodes = {nid for cells in sysm.rom.values() for nid in cells.values()}
    plan = FrontierPlan(nl, 1)
    be = HipBackend(plan.num_slots, keys128.params, torch.device("cuda", 0))
    ex = FrontierExecutor(plan, be)

    def write_bits(nodes, bits, seed):
        be.write_many([plan.slot[i] for i in nodes], client.encrypt_bits(keys128, bits, seed=seed))

    sim = N.PlainSimulator(nl)

    def write_plain(nodes, bits):
        for i, b in zip(nodes, bits):
            sim.val[i] = b

    srcs = plan.sources
    write_bits(srcs, [mem.get(i, 0) for i in srcs], seed=500)           # ROM image + every other input = enc(0)
    write_plain(srcs, [mem.get(i, 0) for i in srcs])
    write_bits(plan.dffs, [0] * len(plan.dffs), seed=501)
    write_plain(plan.dffs, [0] * len(plan.dffs))
    ex.set_input("reset", 0, client.encrypt_bits(keys128, [1], seed=502)[0])
    sim.set_input("reset", 0, 1)
    ex.run()
    sim.evaluate()
    assert want["cycles"] == 7
    for c in range(10):
        ex.tick()
        sim.tick()
        if c == 0:
            ex.set_input("reset", 0, client.encrypt_bits(keys128, [0], seed=503)[0])
            sim.set_input("reset", 0, 0)
            ram_nodes = [i for i in mem if i not in rom_nodes]
            write_bits(ram_nodes, [mem[i] for i in ram_nodes], seed=504)  # setInitialRAM after the first tick
            write_plain(ram_nodes, [mem[i] for i in ram_nodes])
        ex.run()
        sim.evaluate()
        if c + 1 == want["cycles"]:   # the reference's own known answer
            for entry in want["bits"]:
                keys_ = [(entry["name"], b) for b in range(entry["size"])]
                got = client.decrypt_bits(keys128, be.read_many([plan.slot[nl.outputs[k]] for k in keys_]))
                assert N.bytes_from_bits(list(got)) == entry["bytes"], entry["name"]
    outs = sorted(nl.outputs)
    got = client.decrypt_bits(keys128, be.read_many([plan.slot[nl.outputs[k]] for k in outs]))
    assert list(got) == [sim.get_output(*k) for k in outs]
    state = sorted(set(plan.dffs) | set(plan.sources))                    # registers, RAM / ROM cells, inputs
    got = client.decrypt_bits(keys128, be.read_many([plan.slot[i] for i in state]))
    assert list(got) == [int(sim.val[i]) for i in state]
    be.close()


@pytest.mark.parametrize("blueprint,req,want,ncycles", [
    ("counter-4bit.toml", "test13.in", "test13.out", 3),
    ("addr-register-4bit.toml", "test16.in", "test16.out", 3),
    ("div-8bit.toml", "test05.in", "test05.out", 1),
    ("rom-4-8.toml", "test15.in", "test15.out", 1),
    ("dff-reset.toml", "test23.in", "test23.out", 1),
    ("mux-ram-addr8bit.toml", "test06.in", "test06.out", 16),
    ("cahp-pearl-mux.toml", "test09.in", "test09-pearl.out", -1),
    ("cahp-diamond.toml", "test00.in", "test00-diamond.out", -1),      # two RAMs; rom / ram builtins lowered to MUX forms
    ("ram-addr9bit.toml", "test07.in", "test07.out", 16),              # TOGND-widened @addr stream
])
def test_reference_vectors_encrypted(gpu, keys128, blueprint, req, want, ncycles):
    """The reference's cufhe-* cases (test.rb:314-346): request encrypted bit by bit, the blueprint run on
    the GPU through the same runner as the plaintext cases (runner.run_packet + CipherEngine), the result
    decrypted and compared with the expected packet."""
    import torch

    from iyokan_amd.packet import PlainPacket
    from iyokan_amd.runner import CipherEngine, run_packet
    from iyokan_amd.system import load_blueprint

    sysm = load_blueprint(gold(blueprint))
    plan = FrontierPlan(sysm.nl, 1)
    be = HipBackend(plan.num_slots, keys128.params, torch.device("cuda", 0))
    seed = {"v": 7000}

    def encrypt(bits):
        seed["v"] += 1
        return client.encrypt_bits(keys128, bits, seed=seed["v"])

    eng = CipherEngine(FrontierExecutor(plan, be), encrypt, lambda rows: client.decrypt_bits(keys128, rows),
                       client.trivial(keys128.params, 0))
    got = run_packet(sysm, PlainPacket.load(gold(req)), cycles=ncycles, engine=eng)
    expected = PlainPacket.load(gold(want))
    be.close()
    assert got.same_content(expected), got.diff(expected)


def test_cpp_host_runtime_on_gpu(gpu, keys128):
    """test0-shaped self-check of engine.hpp + iyokan_hip.hpp (batching HIPWorker, device arena)."""
    gpu.cleanup()   # the C++ binary initialises the library in its own process
    try:
        exe = os.path.join(ROOT, "iyokan_amd", "host", "test0_hip")
        fixtures = os.path.join(ROOT, "tests", "golden", "reftest")
        out = subprocess.run([exe, "--hip", "--fixtures", fixtures], capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stdout + out.stderr
        assert "ALL OK" in out.stdout
    finally:
        gpu.initialize(keys128, device_ids=(0,))   # module fixture teardown expects an initialised library


def _cpp(gpu, keys128, args, timeout=900):
    """Run the C++ host binary in its own process (it initialises the library itself)."""
    gpu.cleanup()
    try:
        exe = os.path.join(ROOT, "iyokan_amd", "host", "test0_hip")
        out = subprocess.run([exe] + args, capture_output=True, text=True, timeout=timeout)
        assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
        return out.stdout
    finally:
        gpu.initialize(keys128, device_ids=(0,))   # module fixture teardown expects an initialised library


def test_cpp_hip_frontend_cahp_system(gpu, keys128):
    """the whole of config #4's system — CAHP-ruby core + MUX ROM + MUX RAM from the reference's
    blueprint cahp-ruby-mux.toml, program test09.in — entirely in C++ on the GPU: TOML blueprint, encrypted request
    packet, reset cycle, 7 clocks, result packet decrypted and equal to the reference's test09-ruby.out."""
    from netlist_util import gold

    out = _cpp(gpu, keys128, ["--hip-run", gold("cahp-ruby-mux.toml"), gold("test09.in"), "-c", "7", "--expect",
                              gold("test09-ruby.out")])
    assert "result packet equals" in out


def test_cpp_hip_frontend_two_replicas_and_snapshot(gpu, keys128):
    """In-process multi-GPU path of the C++ adapter (two arena replicas, frontier dealt between them, device-to-device
    exchange at level boundaries) and snapshot / resume, on reference vectors: mux-ram-addr8bit (16 clocks, RAM
    contents in the result packet) cut at clock 5, and the 4-bit counter."""
    from netlist_util import gold

    out = _cpp(gpu, keys128, ["--hip-run", gold("mux-ram-addr8bit.toml"), gold("test06.in"), "-c", "16", "--expect",
                              gold("test06.out"), "--gpus", "2", "--snapshot-at", "5"])
    assert "result packet equals" in out and "2 GPU replica(s)" in out
    out = _cpp(gpu, keys128, ["--hip-run", gold("counter-4bit.toml"), gold("test13.in"), "-c", "3", "--expect",
                              gold("test13.out"), "--gpus", "2"])
    assert "result packet equals" in out


def test_cpp_host_runtime_two_replicas(gpu, keys128):
    out = _cpp(gpu, keys128, ["--hip", "--gpus", "2", "--fixtures", os.path.join(ROOT, "tests", "golden", "reftest")])
    assert "ALL OK" in out and "CMUX-memory tasks ok" in out


def test_cpp_do_hip_from_files(gpu, keys128, tmp_path):
    """doHIP(opt) with everything in files, the shape of `iyokan-packet genkey / genevalkey / enc` -> `iyokan tfhe
    --enable-gpu -c N` -> `iyokan-packet dec`: key archives (OS-entropy keys), cereal request packet, result packet;
    then the same run cut by a snapshot after 1 clock and resumed for the remaining 2."""
    from iyokan_amd.packet import PlainPacket
    from netlist_util import gold

    sk, ek, req, res = (str(tmp_path / n) for n in ("sk.bin", "ek.bin", "req.bin", "res.bin"))
    _cpp(gpu, keys128, ["--genkey", sk, ek])
    _cpp(gpu, keys128, ["--enc", sk, gold("test13.in"), req])
    _cpp(gpu, keys128, ["--do-hip", gold("counter-4bit.toml"), "--bkey", ek, "--in", req, "--out", res, "-c", "3"])
    out = _cpp(gpu, keys128, ["--dec", sk, res])
    (tmp_path / "res.toml").write_text(out)
    expected = PlainPacket.load(gold("test13.out"))
    got = PlainPacket.load(str(tmp_path / "res.toml"))
    assert got.same_content(expected), got.diff(expected)
    snap, res2 = str(tmp_path / "snap.bin"), str(tmp_path / "res2.bin")
    _cpp(gpu, keys128, ["--do-hip", gold("counter-4bit.toml"), "--bkey", ek, "--in", req, "--out", res2, "-c", "1", "--snapshot", snap])
    _cpp(gpu, keys128, ["--do-hip", gold("counter-4bit.toml"), "--resume", snap, "--out", res2, "-c", "2", "--skip-reset"])
    out = _cpp(gpu, keys128, ["--dec", sk, res2])
    (tmp_path / "res2.toml").write_text(out)
    got2 = PlainPacket.load(str(tmp_path / "res2.toml"))
    assert got2.same_content(expected), got2.diff(expected)


def test_cpp_do_hip_from_imported_tfhepp_archives(gpu, keys128, tmp_path):
    """The deployment path of SURVEY 8(f3): key files in TFHEpp's OWN archive format (as `iyokan-packet genkey /
    genevalkey` store them; here assembled by cereal's rules around this session's key material, extra members and all)
    -> `test0_hip --import-tfhepp` (blob search + cryptographic verification) -> `doHIP` on the GPU with the imported
    evaluation key -> result decrypted with the imported secret key == the reference's expected packet."""
    import numpy as np

    from iyokan_amd import tfhepp_keys as K
    from iyokan_amd.packet import PlainPacket
    from netlist_util import gold

    rng = np.random.default_rng(11)
    p = keys128.params
    fft_like = K._ptr(2) + rng.standard_normal(1 << 14).astype("<f8").tobytes()
    ek_tf, sk_tf = tmp_path / "ek.tfhepp", tmp_path / "sk.tfhepp"
    ek_tf.write_bytes(K.write_eval_key_like(p, keys128.bk, keys128.ksk, rng.bytes(93), [K.NULL], [fft_like, K.NULL], [K.NULL]))
    sk_tf.write_bytes(K.write_secret_key_like(p, keys128.s0, keys128.s1, tail=rng.bytes(2048 * 8 + 40)))
    sk, ek, req, res = (str(tmp_path / n) for n in ("sk.bin", "ek.bin", "req.bin", "res.bin"))
    out = _cpp(gpu, keys128, ["--import-tfhepp", str(sk_tf), str(ek_tf), sk, ek])
    assert "imported 128bit keys, verified" in out
    _cpp(gpu, keys128, ["--enc", sk, gold("test13.in"), req])
    _cpp(gpu, keys128, ["--do-hip", gold("counter-4bit.toml"), "--bkey", ek, "--in", req, "--out", res, "-c", "3"])
    (tmp_path / "res.toml").write_text(_cpp(gpu, keys128, ["--dec", sk, res]))
    expected = PlainPacket.load(gold("test13.out"))
    got = PlainPacket.load(str(tmp_path / "res.toml"))
    assert got.same_content(expected), got.diff(expected)


def _two_rank_worker(rank, world, port, q):
    """One of two processes sharing GPU 0: the frontier-sharded executor with real ciphertexts and kernels."""
    import hashlib

    import torch
    import torch.distributed as dist

    from iyokan_amd import hip
    from iyokan_amd.params import params_128bit

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        p = params_128bit()
        keys = client.keygen(p, seed=1)                       # same seed: identical keys on both ranks
        hip.initialize(keys, device_ids=(0,))
        nl = N.load_iyokanl1_json(gold("counter-4bit-iyokanl1.json"))
        plan = FrontierPlan(nl, world)
        be = HipBackend(plan.num_slots, p, torch.device("cuda", 0))
        ex = FrontierExecutor(plan, be, rank, world, dist)
        sim = N.PlainSimulator(nl)
        zero = client.trivial(p, 0)
        be.write_many([plan.slot[i] for i in plan.dffs + plan.sources], np.tile(zero, (len(plan.dffs) + len(plan.sources), 1)))
        seed = 500
        outs = []
        for c in range(4):
            ex.tick(); sim.tick()
            for (port_, bit) in sorted(nl.inputs):
                v = int(c == 0) if port_ == "reset" else 1
                seed += 1
                ex.set_input(port_, bit, client.encrypt_bits(keys, [v], seed=seed)[0])   # same seed on both ranks
                sim.set_input(port_, bit, v)
            ex.run(); ex.sync(); sim.evaluate()
            keys_out = sorted(nl.outputs)
            got = client.decrypt_bits(keys, be.read_many([plan.slot[nl.outputs[k]] for k in keys_out]))
            outs.append((list(map(int, got)), [sim.get_output(*k) for k in keys_out]))
        digest = hashlib.sha256(be.arena.cpu().numpy().tobytes()).hexdigest()
        q.put((rank, outs, digest, ex.collectives))
        be.close()
        hip.cleanup()
    finally:
        dist.destroy_process_group()


def test_frontier_two_ranks_one_gpu(gpu, keys128):
    """the sharded executor had only ever run on a bit backend.  Two processes share GPU 0 (gloo,
    host-staged all-gather: RCCL refuses two ranks on one device); each evaluates its share of every level of the 4-bit
    counter with the HIP kernels on real ciphertexts; after 4 clocks both arenas are bit-identical and decrypt like the
    plaintext simulator."""
    import torch.multiprocessing as mp

    gpu.cleanup()
    try:
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = 29900 + (os.getpid() % 2000)
        procs = [ctx.Process(target=_two_rank_worker, args=(r, 2, port, q)) for r in range(2)]
        for p_ in procs:
            p_.start()
        for p_ in procs:
            p_.join(timeout=600)
            assert p_.exitcode == 0
        got = sorted(q.get(timeout=10) for _ in range(2))
    finally:
        gpu.initialize(keys128, device_ids=(0,))
    assert got[0][2] == got[1][2]                                   # identical arenas
    for rank_out in got:
        for dec, want in rank_out[1]:
            assert dec == want
    assert got[0][3] == got[1][3] > 0


def _twin_run(nl, keys, oracle, drive, clocks, after_first_tick=None):
    """The same netlist, plan and encryptions through the HIP backend and through the CPU oracle (OracleBackend, test
    infrastructure), clock by clock; returns both arenas."""
    import torch

    from oracle_lib import OracleBackend

    plan = FrontierPlan(nl, 1)
    hb = HipBackend(plan.num_slots, keys.params, torch.device("cuda", 0))
    ob = OracleBackend(plan.num_slots, oracle)
    exs = [FrontierExecutor(plan, hb), FrontierExecutor(plan, ob)]
    zero = client.trivial(keys.params, 0)
    for be in (hb, ob):
        be.write_many([plan.slot[i] for i in plan.dffs], np.tile(zero, (len(plan.dffs), 1)))
    seed = [7000]

    def set_both(port, bit, v):
        seed[0] += 1
        row = client.encrypt_bits(keys, [v], seed=seed[0])[0]
        for ex in exs:
            ex.set_input(port, bit, row)

    for c in range(clocks):
        drive(set_both, c)
        for ex in exs:
            ex.run()
        if c + 1 < clocks:
            for ex in exs:
                ex.tick()
            if c == 0 and after_first_tick:
                after_first_tick(plan, (hb, ob))
    everything = list(range(plan.num_slots))
    got, ref = hb.read_many(everything), ob.read_many(everything)
    hb.close()
    return plan, got, ref


def test_config3_arena_equals_oracle_word_for_word(gpu, keys128, oracle128):
    """VERDICT r03 next #3b: one clock of BASELINE config #3 (mux-ram-8-16-16, inputs of test08 cycle 0: 18 985 blind rotations
    in 14 level batches, round + remainder dispatch, both rotation kernels) through the HIP backend and through the oracle on
    IDENTICAL encryptions: the whole ciphertext arena — every wire of the netlist, not just the decrypted outputs — is equal
    word for word.  A slot-aliasing or planner bug that still decrypts shows here."""
    nl = N.load_iyokanl1_json(gold("mux-ram-8-16-16.min.json"))
    streams = input_streams(load_packet(gold("test08.in")))
    plan, got, ref = _twin_run(nl, keys128, oracle128, lambda set_enc, c: drive_cycle(set_enc, nl, streams, c), clocks=1)
    bad = np.nonzero((got != ref).any(axis=1))[0]
    assert len(bad) == 0, f"{len(bad)} of {plan.num_slots} arena slots differ, first {bad[:8]}"
    assert (got != 0).any(axis=1).sum() > 13000          # the comparison is not over an empty arena


def test_cahp_core_arena_equals_oracle_word_for_word(gpu, keys128, oracle128):
    """The CAHP-ruby core (4 281 rotations, 41 levels per clock: the narrow-frontier kernel's territory): reset cycle + two
    clocks, random encrypted inputs, HIP backend vs oracle on identical encryptions — arenas equal word for word after the
    third evaluation (registers latched twice in between)."""
    nl = N.load_yosys_json(gold("cahp-ruby-core-yosys.json"))
    rng = np.random.default_rng(5)

    def drive(set_enc, c):
        for (port, bit) in sorted(nl.inputs):
            set_enc(port, bit, int(rng.integers(0, 2)) if port != "reset" else int(c == 0))

    plan, got, ref = _twin_run(nl, keys128, oracle128, drive, clocks=3)
    bad = np.nonzero((got != ref).any(axis=1))[0]
    assert len(bad) == 0, f"{len(bad)} of {plan.num_slots} arena slots differ, first {bad[:8]}"
    assert (got != 0).any(axis=1).sum() > 3000
