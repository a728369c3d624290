"""GPU: remaining C-ABI entry points and error behaviour (status codes, never a crash or a fallback)."""
import ctypes

import numpy as np
import pytest

from iyokan_amd import client
from iyokan_amd.params import OPS

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu(keys128):
    from iyokan_amd import hip

    hip.initialize(keys128, device_ids=(0,))
    yield hip
    hip.cleanup()


def test_arena_roundtrip_and_copy_ops(gpu, keys128):
    st = gpu.Stream(0)
    p = keys128.params
    rng = np.random.default_rng(1)
    host = rng.integers(0, 2**32, size=(5, p.n + 1), dtype=np.uint64).astype(np.uint32)
    arena = gpu.Arena(12)
    st.upload(arena, 2, host)
    assert np.array_equal(st.download(arena, 2, 5), host)
    # COPY / NOT / CONST are exact elementwise maps on arbitrary words
    st.gate_batch(arena, [OPS["COPY"], OPS["NOT"], OPS["CONSTONE"], OPS["CONSTZERO"]], [2, 3, -1, -1],
                  [-1] * 4, [-1] * 4, [8, 9, 10, 11])
    st.sync()
    got = st.download(arena, 8, 4)
    assert np.array_equal(got[0], host[0])
    assert np.array_equal(got[1], (np.uint32(0) - host[1]).astype(np.uint32))
    assert np.array_equal(got[2], client.trivial(p, 1)) and np.array_equal(got[3], client.trivial(p, 0))
    arena.free()
    st.destroy()


def test_blind_rotate_batch_matches_oracle_lvl1(gpu, keys128, oracle128):
    """iyk_hip_blind_rotate_batch: rotation + sample-extract only, checked against the oracle's lvl1 TLWE."""
    import torch

    st = gpu.Stream(0)
    p = keys128.params
    cts = client.encrypt_bits(keys128, [1, 0, 1], seed=90)
    arena = gpu.Arena(3)
    st.upload(arena, 0, cts)
    out = torch.zeros((2, p.N + 1), dtype=torch.int32, device="cuda")
    mu = np.uint32(p.mu)
    st.blind_rotate_batch(arena, [0, 2], [1, -1], [-1, 1], [-1, 0], [mu, np.uint32(0)], out.data_ptr())
    st.sync()
    got = out.cpu().numpy().view(np.uint32)
    lin0 = (np.uint32(0) - cts[0] - cts[1]).astype(np.uint32)
    lin0[-1] = np.uint32((int(lin0[-1]) + p.mu) & 0xFFFFFFFF)
    assert np.array_equal(got[0], oracle128.bootstrap_lvl1(lin0))
    assert np.array_equal(got[1], oracle128.bootstrap_lvl1(cts[2]))
    br_ms, ks_ms = st.last_batch_timing()
    assert br_ms > 0 and ks_ms == 0
    arena.free()
    st.destroy()


def test_trlwe_bootstrap_then_extract_keyswitch(gpu, keys128, oracle128):
    """cufhe::GateBootstrappingTLWE2TRLWElvl01NTT + cufhe::SampleExtractAndKeySwitch shapes: the two halves
    of a gate, composed, must equal the fused gate and the oracle's intermediate TRLWE."""
    import ctypes as C

    import oracle_lib
    import torch

    st = gpu.Stream(0)
    p = keys128.params
    cts = client.encrypt_bits(keys128, [1, 1], seed=95)
    arena = gpu.Arena(4)
    st.upload(arena, 0, cts)
    trlwe = torch.zeros((1, 2 * p.N), dtype=torch.int32, device="cuda")
    st.bootstrap_trlwe_batch(arena, [0], [1], [-1], [-1], [np.uint32(p.mu)], trlwe.data_ptr())
    st.sample_extract_keyswitch_batch(trlwe.data_ptr(), [0], [2], arena, trlwe_slots=1)
    st.gate_batch(arena, [OPS["NAND"]], [0], [1], [-1], [3])
    st.sync()
    got = st.download(arena, 2, 2)
    assert np.array_equal(got[0], got[1])
    assert np.array_equal(got[0], oracle128.gate(OPS["NAND"], cts[0], cts[1]))
    lin = (np.uint32(0) - cts[0] - cts[1]).astype(np.uint32)
    lin[-1] = np.uint32((int(lin[-1]) + p.mu) & 0xFFFFFFFF)
    acc = np.zeros(2 * p.N, dtype=np.uint32)
    u32p = C.POINTER(C.c_uint32)
    oracle_lib.lib().orc_blind_rotate(oracle128.ctx, lin.ctypes.data_as(u32p), acc.ctypes.data_as(u32p), 0)
    assert np.array_equal(trlwe.cpu().numpy().view(np.uint32)[0], acc)
    arena.free()
    st.destroy()


def test_two_streams_and_wrapped_torch_stream(gpu, keys128, oracle128):
    import torch

    p = keys128.params
    cts = client.encrypt_bits(keys128, [1, 1, 0, 1], seed=91)
    s1 = gpu.Stream(0)
    ts = torch.cuda.Stream()
    s2 = gpu.Stream(0, hip_stream=ts.cuda_stream)
    a1, a2 = gpu.Arena(3), gpu.Arena(3)
    s1.upload(a1, 0, cts[:2])
    s2.upload(a2, 0, cts[2:])
    s1.gate_batch(a1, [OPS["XOR"]], [0], [1], [-1], [2])
    s2.gate_batch(a2, [OPS["ANDNOT"]], [0], [1], [-1], [2])
    ts.synchronize()
    s1.sync()
    assert s1.query() and s2.query()
    assert np.array_equal(s1.download(a1, 2, 1)[0], oracle128.gate(OPS["XOR"], cts[0], cts[1]))
    assert np.array_equal(s2.download(a2, 2, 1)[0], oracle128.gate(OPS["ANDNOT"], cts[2], cts[3]))
    for a in (a1, a2):
        a.free()
    s1.destroy()
    s2.destroy()


def test_error_codes(gpu, keys128):
    L = gpu.lib()
    st = gpu.Stream(0)
    arena = gpu.Arena(4)
    i32 = lambda *v: np.array(v, dtype=np.int32)
    with pytest.raises(gpu.IykHipError, match="unknown gate op"):
        st.gate_batch(arena, i32(99), i32(0), i32(1), i32(-1), i32(2))
    with pytest.raises(gpu.IykHipError, match="MUX needs three input slots"):
        st.gate_batch(arena, i32(OPS["MUX"]), i32(0), i32(1), i32(-1), i32(2))
    with pytest.raises(gpu.IykHipError, match="output slot outside the arena"):
        st.gate_batch(arena, i32(OPS["NAND"]), i32(0), i32(1), i32(-1), i32(-1))
    # every index is checked against the arena size the caller states: no out-of-bounds device access
    with pytest.raises(gpu.IykHipError, match="output slot outside the arena"):
        st.gate_batch(arena, i32(OPS["NAND"]), i32(0), i32(1), i32(-1), i32(4))
    with pytest.raises(gpu.IykHipError, match="two input slots inside the arena"):
        st.gate_batch(arena, i32(OPS["NAND"]), i32(0), i32(4), i32(-1), i32(2))
    with pytest.raises(gpu.IykHipError, match="NOT/COPY needs one input slot"):
        st.gate_batch(arena, i32(OPS["NOT"]), i32(17), i32(-1), i32(-1), i32(2))
    with pytest.raises(gpu.IykHipError, match="outside the buffer"):
        st.download(arena, 3, 2)
    with pytest.raises(gpu.IykHipError, match="slot index outside the arena"):
        st.download_slots(arena, [0, 4])
    with pytest.raises(gpu.IykHipError, match="input slot outside the arena"):
        import torch
        out = torch.zeros((1, keys128.params.N + 1), dtype=torch.int32, device="cuda")
        st.blind_rotate_batch(arena, [-1], [-1], [1], [0], [np.uint32(0)], out.data_ptr())
    assert L.iyk_hip_gate_batch(None, ctypes.c_void_p(arena.ptr), 4, 1, None, None, None, None, None) == -1
    assert L.iyk_hip_init(1, None, ctypes.byref(keys128.params), None, None) == -2       # already initialised
    assert L.iyk_hip_cleanup() == -2 and b"streams still alive" in L.iyk_hip_last_error()
    h = ctypes.c_void_p()
    assert L.iyk_hip_stream_create(7, ctypes.byref(h)) == -1                               # gpu_index out of range
    assert gpu.resident_key_bytes() > 100e6 and gpu.ntt_path() in ("fft", "fp50", "goldilocks")
    assert L.iyk_hip_stream_gpu(st.h) == 0
    arena.free()
    st.destroy()


def test_gate_host_on_an_adopted_stream_is_complete_when_that_stream_is(gpu, keys128, oracle128):
    """a caller that adopted its own hipStream_t (iyk_hip_stream_wrap) and synchronises it natively must find the result
    of iyk_hip_gate_host in `out` — with the coalescer (the default) the gate used to be parked and never launched until a
    library-side poll, and without it the result sat in the pinned mirror.  Now such a stream bypasses both: H2D, kernels, D2H into
    `out`, all on the adopted stream.  Checked with the torch stream's own synchronize and NO iyk_hip_stream_query / _sync."""
    import ctypes

    import torch

    p = keys128.params
    cts = client.encrypt_bits(keys128, [1, 0, 1], seed=191)
    ts = torch.cuda.Stream()
    st = gpu.Stream(0, hip_stream=ts.cuda_stream)
    u32p = ctypes.POINTER(ctypes.c_uint32)
    L = gpu.lib()
    for op, ins in (("NAND", (0, 1)), ("MUX", (0, 1, 2)), ("NOT", (2,))):
        out = np.full(p.n + 1, 0xDEADBEEF, dtype=np.uint32)
        args = [np.ascontiguousarray(cts[i]) for i in ins] + [None] * (3 - len(ins))
        ptr = [a.ctypes.data_as(u32p) if a is not None else None for a in args]
        assert L.iyk_hip_gate_host(st.h, OPS[op], ptr[0], ptr[1], ptr[2], out.ctypes.data_as(u32p)) == 0
        ts.synchronize()                                   # the CALLER's synchronisation only
        want = oracle128.gate(OPS[op], *[cts[i] for i in ins])
        assert np.array_equal(out, want), op
    # gate after gate WITHOUT synchronising in between: nothing of the library's may be shared between two calls in flight
    outs = [np.full(p.n + 1, 0xDEADBEEF, dtype=np.uint32) for _ in range(4)]
    pairs = [(0, 1), (1, 2), (2, 0), (1, 1)]
    keep = []
    for out, (a, b) in zip(outs, pairs):
        xa, xb = np.ascontiguousarray(cts[a]), np.ascontiguousarray(cts[b])
        keep += [xa, xb]
        assert L.iyk_hip_gate_host(st.h, OPS["XNOR"], xa.ctypes.data_as(u32p), xb.ctypes.data_as(u32p), None, out.ctypes.data_as(u32p)) == 0
    ts.synchronize()
    for out, (a, b) in zip(outs, pairs):
        assert np.array_equal(out, oracle128.gate(OPS["XNOR"], cts[a], cts[b])), (a, b)
    st.destroy()


def test_gate_host_from_two_host_threads(gpu, keys128, oracle128):
    """the parked gates of a GPU are shared state behind a lock that one host thread used to hold for a whole batch while it
    waited.  Two host threads, each with its own streams (a stream is used by one thread at a time: the contract), park gates, poll and
    block concurrently — one mostly through iyk_hip_stream_sync (the blocking path that now drops the lock around its wait), the other
    by polling.  Every result must be the oracle's, nothing may deadlock, and cleanup-relevant state must be consistent afterwards
    (the module's fixture tears the library down after this test)."""
    import ctypes
    import threading

    p = keys128.params
    L = gpu.lib()
    u32p = ctypes.POINTER(ctypes.c_uint32)
    rng = np.random.default_rng(2026)
    bits = rng.integers(0, 2, size=8).astype(np.uint8)
    cts = client.encrypt_bits(keys128, bits, seed=2027)
    kinds = ["NAND", "XOR", "ORNOT", "AND"]
    want = {(k, a, b): oracle128.gate(OPS[k], cts[a], cts[b]) for k in kinds for a in range(8) for b in range(8) if (a + 3 * b) % 5 == 0}
    jobs = sorted(want)
    errors = []

    def worker(tid, blocking):
        try:
            streams = [gpu.Stream(0) for _ in range(6)]
            outs = [np.zeros(p.n + 1, dtype=np.uint32) for _ in streams]
            for rnd in range(6):
                mine = [jobs[(tid * 7 + rnd * len(streams) + i) % len(jobs)] for i in range(len(streams))]
                for st, out, (k, a, b) in zip(streams, outs, mine):
                    out[:] = 0xDEADBEEF
                    rc = L.iyk_hip_gate_host(st.h, OPS[k], cts[a].ctypes.data_as(u32p), cts[b].ctypes.data_as(u32p), None,
                                             out.ctypes.data_as(u32p))
                    assert rc == 0, L.iyk_hip_last_error()
                if blocking:
                    for st in streams:
                        st.sync()
                else:
                    pending = set(range(len(streams)))
                    spins = 0
                    while pending:
                        for i in list(pending):
                            if streams[i].query():
                                pending.discard(i)
                        spins += 1
                        assert spins < 5_000_000, "a parked gate never came back"
                for out, job in zip(outs, mine):
                    assert np.array_equal(out, want[job]), (tid, rnd, job)
            for st in streams:
                st.destroy()
        except BaseException as e:   # noqa: BLE001 - reported on the main thread
            errors.append((tid, repr(e)))

    threads = [threading.Thread(target=worker, args=(0, True)), threading.Thread(target=worker, args=(1, False))]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
        assert not t.is_alive(), "deadlock between two host threads driving iyk_hip_gate_host"
    assert not errors, errors


def test_bulk_slot_io_and_arena_copy(gpu, keys128):
    """upload_slots / download_slots (Mem::set/get of many cells in one transfer) and the device-side copy."""
    st = gpu.Stream(0)
    p = keys128.params
    rng = np.random.default_rng(5)
    host = rng.integers(0, 2**32, size=(300, p.n + 1), dtype=np.uint64).astype(np.uint32)
    arena = gpu.Arena(1000)
    slots = rng.permutation(1000)[:300]
    st.upload_slots(arena, slots, host)
    assert np.array_equal(st.download_slots(arena, slots), host)
    order = rng.permutation(300)
    assert np.array_equal(st.download_slots(arena, slots[order]), host[order])
    lo = int(np.argmin(slots))
    assert np.array_equal(st.download(arena, int(slots[lo]), 1)[0], host[lo])
    other = gpu.Arena(400)
    st.arena_copy(other, 100, arena, 0, 300)
    assert np.array_equal(st.download(other, 100, 300), st.download(arena, 0, 300))
    for a in (arena, other):
        a.free()
    st.destroy()


def test_timing_log_covers_rotation_only_calls(gpu, keys128):
    """after timing_log_begin the rotation-only entry points must still work (they create their own events)."""
    import torch

    st = gpu.Stream(0)
    p = keys128.params
    cts = client.encrypt_bits(keys128, [1, 0], seed=93)
    arena = gpu.Arena(3)
    st.upload(arena, 0, cts)
    out = torch.zeros((1, p.N + 1), dtype=torch.int32, device="cuda")
    st.timing_log_begin()
    st.blind_rotate_batch(arena, [0], [1], [-1], [-1], [np.uint32(p.mu)], out.data_ptr())
    st.gate_batch(arena, [OPS["NAND"]], [0], [1], [-1], [2])
    nb, br_ms, ks_ms = st.timing_log_end()
    assert nb == 2 and br_ms > 0 and ks_ms > 0
    st.gate_batch(arena, [OPS["NAND"]], [0], [1], [-1], [2])   # standing events were re-created
    st.sync()
    assert st.last_batch_timing()[0] > 0
    arena.free()
    st.destroy()


@pytest.mark.parametrize("kernel", [None, "fft", "latfft", "w32", "lat3"])
def test_cmux_memory_entry_points_256_jobs(gpu, keys128, oracle128, kernel, monkeypatch):
    """Every rotation kernel (default dispatch, then each one forced) through the TRLWE output mode and the output
    indirection.  VERDICT r01 item 6: the two GPU pieces of the CMUX memories composed on 300 jobs —
    GateBootstrappingTLWE2TRLWElvl01NTT into scattered TRLWE cells (trlwe_out indirection), then
    SampleExtractAndKeySwitch of those cells into arena slots — against the oracle: every TLWE word of every job, and
    the TRLWE words of a sample of cells."""
    import ctypes as C
    import os

    import oracle_lib
    import torch

    if kernel is None:
        monkeypatch.delenv("IYK_HIP_ROT_KERNEL", raising=False)
    else:
        monkeypatch.setenv("IYK_HIP_ROT_KERNEL", kernel)
    st = gpu.Stream(0)
    p = keys128.params
    rng = np.random.default_rng(66)
    jobs, cells = 300, 512
    bits = rng.integers(0, 2, size=jobs).astype(np.uint8)
    cts = client.encrypt_bits(keys128, bits, seed=96)
    arena = gpu.Arena(2 * jobs)
    st.upload(arena, 0, cts)
    L = gpu.lib()
    d_trlwe = C.c_void_p()
    assert L.iyk_hip_trlwe_alloc(0, cells, C.byref(d_trlwe)) == 0
    cell = rng.permutation(cells)[:jobs].astype(np.int32)
    none = np.full(jobs, -1, dtype=np.int32)
    st.bootstrap_trlwe_batch(arena, np.arange(jobs), none, np.ones(jobs), np.zeros(jobs), np.zeros(jobs, dtype=np.uint32),
                             d_trlwe.value, trlwe_slots=cells, trlwe_out=cell)
    st.sample_extract_keyswitch_batch(d_trlwe.value, cell, np.arange(jobs, 2 * jobs), arena, trlwe_slots=cells)
    st.sync()
    got = st.download(arena, jobs, jobs)
    # TLWE words: blind rotation of the input as it is + sample extract + key switch = the oracle's pieces composed
    u32p = C.POINTER(C.c_uint32)
    sample = rng.choice(jobs, size=24, replace=False)
    host = np.zeros((2, 2 * p.N), dtype=np.uint32)
    for j in sample:
        acc = np.zeros(2 * p.N, dtype=np.uint32)
        oracle_lib.lib().orc_blind_rotate(oracle128.ctx, cts[j].ctypes.data_as(u32p), acc.ctypes.data_as(u32p), 2)
        row = np.zeros(2 * p.N, dtype=np.uint32)
        assert L.iyk_hip_trlwe_download(st.h, d_trlwe, cells, int(cell[j]), 1, row.ctypes.data_as(u32p)) == 0
        st.sync()
        assert np.array_equal(row, acc)
        t1 = np.zeros(p.N + 1, dtype=np.uint32)
        oracle_lib.lib().orc_sample_extract0(oracle128.ctx, acc.ctypes.data_as(u32p), t1.ctypes.data_as(u32p))
        assert np.array_equal(got[j], oracle128.keyswitch(t1))
    assert np.array_equal(client.decrypt_bits(keys128, got), bits)     # bootstrapped identity, all 300
    # TRLWE upload / download round trip and bounds
    img = rng.integers(0, 2**32, size=(3, 2 * p.N), dtype=np.uint64).astype(np.uint32)
    assert L.iyk_hip_trlwe_upload(st.h, d_trlwe, cells, 7, 3, img.ctypes.data_as(u32p)) == 0
    back = np.zeros_like(img)
    assert L.iyk_hip_trlwe_download(st.h, d_trlwe, cells, 7, 3, back.ctypes.data_as(u32p)) == 0
    st.sync()
    assert np.array_equal(img, back)
    assert L.iyk_hip_trlwe_download(st.h, d_trlwe, cells, cells - 1, 2, back.ctypes.data_as(u32p)) == -1
    with pytest.raises(gpu.IykHipError, match="output index outside the buffer"):
        st.bootstrap_trlwe_batch(arena, [0], [-1], [1], [0], [np.uint32(0)], d_trlwe.value, trlwe_slots=cells, trlwe_out=[cells])
    assert L.iyk_hip_trlwe_free(0, d_trlwe) == 0
    arena.free()
    st.destroy()


@pytest.mark.parametrize("ks", ["0", "1"])
def test_extract_keyswitch_on_arbitrary_trlwe_words(gpu, keys128, oracle128, ks, monkeypatch):
    """SampleExtractAndKeySwitch on TRLWE cells no bootstrap produces — all-ones (every key-switch digit 3), the sign
    bit, just below the rounding offset (every digit 0: no row subtracted), alternating extremes, uniform words — with
    each key-switch kernel: the oracle's sample-extract + key-switch words."""
    import ctypes as C

    import oracle_lib

    monkeypatch.setenv("IYK_HIP_KS_KERNEL", ks)
    st = gpu.Stream(0)
    p = keys128.params
    rng = np.random.default_rng(67)
    cells = 70                                            # more than one workgroup of 64 gates
    img = rng.integers(0, 2**32, size=(cells, 2 * p.N), dtype=np.uint64).astype(np.uint32)
    img[0] = 0xFFFFFFFF
    img[1] = 0x80000000
    img[2] = (1 << (32 - 1 - 2 * p.t)) - 1                # a + prec stays below the first digit: nothing to subtract
    img[3] = np.where(np.arange(2 * p.N) % 2 == 0, 0, 0xFFFFFFFF)
    img[4] = 0
    L = gpu.lib()
    u32p = C.POINTER(C.c_uint32)
    d_trlwe = C.c_void_p()
    assert L.iyk_hip_trlwe_alloc(0, cells, C.byref(d_trlwe)) == 0
    assert L.iyk_hip_trlwe_upload(st.h, d_trlwe, cells, 0, cells, img.ctypes.data_as(u32p)) == 0
    arena = gpu.Arena(cells)
    order = rng.permutation(cells).astype(np.int32)       # cell order[j] -> slot j
    st.sample_extract_keyswitch_batch(d_trlwe.value, order, np.arange(cells), arena, trlwe_slots=cells)
    st.sync()
    got = st.download(arena, 0, cells)
    for j in range(cells):
        t1 = np.zeros(p.N + 1, dtype=np.uint32)
        acc = np.ascontiguousarray(img[order[j]])
        oracle_lib.lib().orc_sample_extract0(oracle128.ctx, acc.ctypes.data_as(u32p), t1.ctypes.data_as(u32p))
        assert np.array_equal(got[j], oracle128.keyswitch(t1)), j
    assert L.iyk_hip_trlwe_free(0, d_trlwe) == 0
    arena.free()
    st.destroy()


def test_zz_calibrated_cost_table_and_dispatch_threshold(gpu, keys128, oracle128):
    """iyk_hip_calibrate: the level-cost table of include/iyokan_hip.h measured on THIS GPU (one full round of the
    wave-per-rotation kernel, 1 .. 8 passes of the narrow-frontier kernel), stamped with the build; the dispatch's
    narrow-frontier threshold follows the measured cross-over, and a batch either side of it still equals the oracle.
    (Runs last in this module: it changes the dispatch threshold of the shared library instance.)"""
    import os

    before = gpu.level_cost_table(0)
    assert not before["calibrated"] and before["round"] == gpu.rotation_round() == 8 * before["pass"]
    t = gpu.calibrate(0)
    assert t["calibrated"] and t["build_id"] == gpu.build_id() and t["round"] == before["round"]
    assert 5.0 < t["round_ms"] < 60.0 and 1.0 < t["pass_ms"][0] < 10.0
    assert all(b > a for a, b in zip(t["pass_ms"], t["pass_ms"][1:]))
    mp = t["max_passes"]
    assert 1 <= mp <= 8 and t["pass_ms"][mp - 1] < t["round_ms"] and (mp == 8 or t["pass_ms"][mp] >= t["round_ms"])
    L = gpu.lib()
    assert abs(L.iyk_hip_level_cost_ms(0, t["round"] + 1) - (t["round_ms"] + t["pass_ms"][0])) < 1e-3
    # one rotation beyond the threshold: one more (partial) round — unless the threshold is the whole round (8 passes), where
    # "beyond" is a full round plus one pass
    beyond = t["round_ms"] if mp < 8 else t["round_ms"] + t["pass_ms"][0]
    assert abs(L.iyk_hip_level_cost_ms(0, mp * t["pass"] + 1) - beyond) < 1e-3
    print("calibrated table:", t)
    st = gpu.Stream(0)
    p = keys128.params
    rng = np.random.default_rng(314)
    nin = 128
    bits = rng.integers(0, 2, size=nin).astype(np.uint8)
    for ng in (mp * t["pass"], mp * t["pass"] + 1):       # last size of the narrow-frontier kernel, first of the partial round
        ia, ib = (rng.integers(0, nin, size=ng).astype(np.int32) for _ in range(2))
        host = np.zeros((nin + ng, p.n + 1), dtype=np.uint32)
        host[:nin] = client.encrypt_bits(keys128, bits, seed=315)
        arena = gpu.Arena(host.shape[0])
        st.upload(arena, 0, host)
        st.gate_batch(arena, np.full(ng, OPS["NAND"], dtype=np.int32), ia, ib, np.full(ng, -1, dtype=np.int32),
                      np.arange(nin, nin + ng, dtype=np.int32))
        st.sync()
        got = st.download(arena, nin, ng)
        arena.free()
        assert np.array_equal(client.decrypt_bits(keys128, got), 1 - (bits[ia] & bits[ib]))
        sample = rng.choice(ng, size=16, replace=False)
        ref = np.zeros((nin + 16, p.n + 1), dtype=np.uint32)
        ref[:nin] = host[:nin]
        oracle128.gate_batch([OPS["NAND"]] * 16, ia[sample], ib[sample], [-1] * 16, list(range(nin, nin + 16)), ref,
                             nthreads=os.cpu_count() or 1)
        assert np.array_equal(got[sample], ref[nin:])
    st.destroy()

