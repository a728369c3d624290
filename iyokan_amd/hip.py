"""ctypes binding of libiyokan_hip.so — the C ABI declared in include/iyokan_hip.h.

Host-side mirror of the cuFHE surface Iyokan's GPU worker uses
(/root/reference/src/iyokan_cufhe.hpp:8-32,249-261; /root/reference/src/iyokan_cufhe.cpp:530-536,721):
`initialize()` = cufhe::SetGPUNum + cufhe::Initialize, `Stream` = CUFHEStream,
`Stream.gate_batch` = N x cufhe::Nand/And/.../Mux, `Stream.query` = cufhe::StreamQuery,
`cleanup()` = cufhe::CleanUp.  There is no CPU fallback: every call goes to the HIP library
and raises IykHipError if it is missing or reports an error.
"""
import ctypes
import os

import numpy as np

from .params import IykParams, OPS

_LIB = None
_u32p = ctypes.POINTER(ctypes.c_uint32)
_i32p = ctypes.POINTER(ctypes.c_int32)
_vp = ctypes.c_void_p

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libiyokan_hip.so")

# every symbol include/iyokan_hip.h declares (tests check the built library exports them all)
EXPORTS = [
    "iyk_hip_init", "iyk_hip_cleanup", "iyk_hip_is_initialized", "iyk_hip_num_gpus", "iyk_hip_get_params",
    "iyk_hip_last_error", "iyk_hip_stream_create", "iyk_hip_stream_wrap", "iyk_hip_stream_destroy",
    "iyk_hip_stream_query", "iyk_hip_stream_sync", "iyk_hip_arena_alloc", "iyk_hip_arena_free",
    "iyk_hip_arena_upload", "iyk_hip_arena_download", "iyk_hip_gate_batch", "iyk_hip_gate_host",
    "iyk_hip_blind_rotate_batch", "iyk_hip_last_batch_timing", "iyk_hip_resident_key_bytes",
    "iyk_hip_timing_log_begin", "iyk_hip_timing_log_end", "iyk_hip_ntt_path", "iyk_hip_decomposition_levels", "iyk_hip_fft_round_error", "iyk_hip_level_cost_defaults", "iyk_hip_level_cost_table",
    "iyk_hip_level_cost_ms", "iyk_hip_calibrate",
    "iyk_hip_bootstrap_trlwe_batch", "iyk_hip_sample_extract_keyswitch_batch",
    "iyk_hip_stream_gpu", "iyk_hip_arena_upload_slots", "iyk_hip_arena_download_slots", "iyk_hip_arena_copy",
    "iyk_hip_arena_sync_slots", "iyk_hip_trlwe_alloc", "iyk_hip_trlwe_free", "iyk_hip_trlwe_upload",
    "iyk_hip_trlwe_download", "iyk_hip_rotation_round", "iyk_hip_arena_sync_slots_multi", "iyk_hip_peer_access",
    "iyk_hip_build_id", "iyk_hip_host_alloc", "iyk_hip_host_free", "iyk_hip_init_profile",
]


class IykHipError(RuntimeError):
    pass


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise IykHipError(f"{LIB_PATH} is missing — build it with __graft_entry__.build(); there is no CPU fallback")
        # Load order matters: PyTorch ships its own libamdhip64; if our library pulled in the system
        # HIP runtime first, torch.cuda would later find "No HIP GPUs".  Importing torch first makes
        # both share one runtime (torch is only plumbing here: device tensors, streams, RCCL).
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = ctypes.CDLL(LIB_PATH)
        L.iyk_hip_last_error.restype = ctypes.c_char_p
        L.iyk_hip_init.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(IykParams), _u32p, _u32p]
        L.iyk_hip_get_params.argtypes = [ctypes.POINTER(IykParams)]
        L.iyk_hip_stream_create.argtypes = [ctypes.c_int, ctypes.POINTER(_vp)]
        L.iyk_hip_stream_wrap.argtypes = [ctypes.c_int, _vp, ctypes.POINTER(_vp)]
        for f in ("iyk_hip_stream_destroy", "iyk_hip_stream_query", "iyk_hip_stream_sync", "iyk_hip_stream_gpu"):
            getattr(L, f).argtypes = [_vp]
        u64 = ctypes.c_uint64
        L.iyk_hip_arena_alloc.argtypes = [ctypes.c_int, u64, ctypes.POINTER(_vp)]
        L.iyk_hip_arena_free.argtypes = [ctypes.c_int, _vp]
        L.iyk_hip_arena_upload.argtypes = [_vp, _vp, u64, u64, u64, _u32p]
        L.iyk_hip_arena_download.argtypes = [_vp, _vp, u64, u64, u64, _u32p]
        L.iyk_hip_arena_upload_slots.argtypes = [_vp, _vp, u64, u64, _i32p, _u32p]
        L.iyk_hip_arena_download_slots.argtypes = [_vp, _vp, u64, u64, _i32p, _u32p]
        L.iyk_hip_arena_copy.argtypes = [_vp, _vp, u64, u64, _vp, u64, u64, u64]
        L.iyk_hip_arena_sync_slots.argtypes = [_vp, _vp, u64, _vp, _vp, u64, u64, _i32p]
        L.iyk_hip_arena_sync_slots_multi.argtypes = [_vp, _vp, u64, ctypes.c_int, ctypes.POINTER(_vp), ctypes.POINTER(_vp),
                                                     ctypes.POINTER(u64), u64, _i32p]
        L.iyk_hip_peer_access.argtypes = [ctypes.c_int, ctypes.c_int]
        L.iyk_hip_trlwe_alloc.argtypes = [ctypes.c_int, u64, ctypes.POINTER(_vp)]
        L.iyk_hip_trlwe_free.argtypes = [ctypes.c_int, _vp]
        L.iyk_hip_trlwe_upload.argtypes = [_vp, _vp, u64, u64, u64, _u32p]
        L.iyk_hip_trlwe_download.argtypes = [_vp, _vp, u64, u64, u64, _u32p]
        L.iyk_hip_gate_batch.argtypes = [_vp, _vp, u64, u64, _i32p, _i32p, _i32p, _i32p, _i32p]
        L.iyk_hip_gate_host.argtypes = [_vp, ctypes.c_int, _u32p, _u32p, _u32p, _u32p]
        L.iyk_hip_blind_rotate_batch.argtypes = [_vp, _vp, u64, u64, _i32p, _i32p, _i32p, _i32p, _u32p, _vp]
        L.iyk_hip_bootstrap_trlwe_batch.argtypes = [_vp, _vp, u64, u64, _i32p, _i32p, _i32p, _i32p, _u32p, _vp, u64, _i32p]
        L.iyk_hip_sample_extract_keyswitch_batch.argtypes = [_vp, _vp, u64, u64, _i32p, _i32p, _vp, u64]
        L.iyk_hip_last_batch_timing.argtypes = [_vp, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float)]
        L.iyk_hip_resident_key_bytes.argtypes = [ctypes.POINTER(ctypes.c_uint64)]
        L.iyk_hip_level_cost_ms.restype = ctypes.c_double
        L.iyk_hip_level_cost_ms.argtypes = [ctypes.c_int, ctypes.c_int]
        L.iyk_hip_fft_round_error.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
        L.iyk_hip_timing_log_begin.argtypes = [_vp]
        L.iyk_hip_timing_log_end.argtypes = [_vp, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_double),
                                             ctypes.POINTER(ctypes.c_double)]
        _LIB = L
    return _LIB


def _check(rc, what):
    if rc < 0:
        raise IykHipError(f"{what} failed ({rc}): {lib().iyk_hip_last_error().decode()}")
    return rc


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def initialize(keys, device_ids=(0,)):
    """cufhe::SetGPUNum(len(device_ids)) + cufhe::Initialize(ek)."""
    ids = (ctypes.c_int * len(device_ids))(*device_ids)
    bk = np.ascontiguousarray(keys.bk, dtype=np.uint32)
    ksk = np.ascontiguousarray(keys.ksk, dtype=np.uint32)
    _check(lib().iyk_hip_init(len(device_ids), ids, ctypes.byref(keys.params),
                              bk.ctypes.data_as(_u32p), ksk.ctypes.data_as(_u32p)), "iyk_hip_init")


def cleanup():
    _check(lib().iyk_hip_cleanup(), "iyk_hip_cleanup")


def is_initialized():
    return bool(lib().iyk_hip_is_initialized())


def resident_key_bytes():
    v = ctypes.c_uint64()
    _check(lib().iyk_hip_resident_key_bytes(ctypes.byref(v)), "iyk_hip_resident_key_bytes")
    return v.value


def ntt_path():
    """'fft' (complex FP64 FFT on 16-bit key halves, exact by a rounding bound: the default), 'fp50' (FP64 FMA field) or
    'goldilocks' (64-bit integer field)."""
    return {2: "fft", 1: "fp50", 0: "goldilocks"}[_check(lib().iyk_hip_ntt_path(), "iyk_hip_ntt_path")]


def fft_round_error(gpu=0):
    """IYK_HIP_DEBUG=1 at init: largest distance from an integer of any inverse-transform output of the FFT kernel so far."""
    v = ctypes.c_double()
    _check(lib().iyk_hip_fft_round_error(int(gpu), ctypes.byref(v)), "iyk_hip_fft_round_error")
    return v.value


def decomposition_levels():
    """Digit polynomials per accumulator polynomial and CMUX step (include/iyokan_hip.h: 3 for the 128-bit set; 80-bit set
    4 by default, 2 with IYK_HIP_DECOMP=direct at init)."""
    return _check(lib().iyk_hip_decomposition_levels(), "iyk_hip_decomposition_levels")


def build_id():
    """Source hash the loaded libiyokan_hip.so was built from (tools/src_hash.py), or 'unknown'."""
    f = lib().iyk_hip_build_id
    f.restype = ctypes.c_char_p
    return f().decode()


def peer_access(gpu_a, gpu_b):
    """True when GPU a reads GPU b's memory directly (peer access enabled at init), False when copies are host-staged."""
    return _check(lib().iyk_hip_peer_access(int(gpu_a), int(gpu_b)), "iyk_hip_peer_access") == 1


class IykLevelCost(ctypes.Structure):
    """include/iyokan_hip.h: iyk_level_cost — what a level of r rotations costs one GPU (the library's only copy)."""
    _fields_ = [("round", ctypes.c_int32), ("pass_", ctypes.c_int32), ("max_passes", ctypes.c_int32),
                ("calibrated", ctypes.c_int32), ("round_ms", ctypes.c_float), ("pass_ms", ctypes.c_float * 8),
                ("build_id", ctypes.c_char * 20)]

    def as_dict(self):
        return {"round": self.round, "pass": self.pass_, "max_passes": self.max_passes, "calibrated": bool(self.calibrated),
                "round_ms": float(self.round_ms), "pass_ms": [float(v) for v in self.pass_ms], "build_id": self.build_id.decode()}


def level_cost_defaults():
    """The compiled-in MI355X cost table of the loaded library (no GPU, no initialisation needed)."""
    c = IykLevelCost()
    _check(lib().iyk_hip_level_cost_defaults(ctypes.byref(c)), "iyk_hip_level_cost_defaults")
    return c.as_dict()


def level_cost_table(gpu=0):
    """GPU `gpu`'s cost table: its CU count, measured milliseconds once calibrate(gpu) has run."""
    c = IykLevelCost()
    _check(lib().iyk_hip_level_cost_table(int(gpu), ctypes.byref(c)), "iyk_hip_level_cost_table")
    return c.as_dict()


def calibrate(gpu=0):
    """~0.15 s self-calibration of the cost table (and of the dispatch's narrow-frontier threshold) on GPU `gpu`;
    gpu = -1: every GPU concurrently (returns GPU 0's table)."""
    _check(lib().iyk_hip_calibrate(int(gpu)), "iyk_hip_calibrate")
    return level_cost_table(max(int(gpu), 0))


def init_profile():
    """Steps of the last iyk_hip_init as [(name, gpu or None, milliseconds)]: alloc g, pin, enqueue g, wait g."""
    L = lib()
    L.iyk_hip_init_profile.restype = ctypes.c_char_p
    out = []
    for item in L.iyk_hip_init_profile().decode().split(";"):
        parts = item.split()
        if len(parts) == 3:
            out.append((parts[0], int(parts[1]), float(parts[2])))
        elif len(parts) == 2:
            out.append((parts[0], None, float(parts[1])))
    return out


def rotation_round(gpu=0):
    """Rotations in one full round of the default wave-per-rotation kernel on GPU `gpu` (resident waves x CUs)."""
    return _check(lib().iyk_hip_rotation_round(int(gpu)), "iyk_hip_rotation_round")


def current_params():
    p = IykParams()
    _check(lib().iyk_hip_get_params(ctypes.byref(p)), "iyk_hip_get_params")
    return p


class Arena:
    """Device-resident array of TLWE lvl0 ciphertexts, addressed by slot index."""

    def __init__(self, slots, gpu_index=0, device_ptr=None, owner=None):
        self.slots, self.gpu_index = int(slots), gpu_index
        self.n1 = current_params().n + 1
        self._owner = owner  # e.g. the torch tensor backing device_ptr
        if device_ptr is None:
            ptr = _vp()
            _check(lib().iyk_hip_arena_alloc(gpu_index, self.slots, ctypes.byref(ptr)), "iyk_hip_arena_alloc")
            self.ptr, self._owned = ptr.value, True
        else:
            self.ptr, self._owned = int(device_ptr), False

    @classmethod
    def from_torch(cls, tensor, gpu_index=0):
        """Wrap a contiguous int32/uint32-sized CUDA tensor of shape (slots, n+1)."""
        assert tensor.is_cuda and tensor.is_contiguous() and tensor.element_size() == 4
        return cls(tensor.shape[0], gpu_index, tensor.data_ptr(), owner=tensor)

    def free(self):
        if self._owned and self.ptr:
            _check(lib().iyk_hip_arena_free(self.gpu_index, self.ptr), "iyk_hip_arena_free")
        self.ptr = None

    def __del__(self):
        try:
            if is_initialized():
                self.free()
        except Exception:
            pass


class Stream:
    """CUFHEStream equivalent (/root/reference/src/iyokan_cufhe.hpp:8-27)."""

    def __init__(self, gpu_index=0, hip_stream=None):
        h = _vp()
        if hip_stream is None:
            _check(lib().iyk_hip_stream_create(gpu_index, ctypes.byref(h)), "iyk_hip_stream_create")
        else:
            _check(lib().iyk_hip_stream_wrap(gpu_index, _vp(int(hip_stream)), ctypes.byref(h)), "iyk_hip_stream_wrap")
        self.h, self.gpu_index = h, gpu_index

    def destroy(self):
        if self.h:
            _check(lib().iyk_hip_stream_destroy(self.h), "iyk_hip_stream_destroy")
            self.h = None

    def query(self):
        """cufhe::StreamQuery: True once everything enqueued so far has finished."""
        return bool(_check(lib().iyk_hip_stream_query(self.h), "iyk_hip_stream_query"))

    def sync(self):
        _check(lib().iyk_hip_stream_sync(self.h), "iyk_hip_stream_sync")

    def upload(self, arena, first_slot, host):
        host = np.ascontiguousarray(host, dtype=np.uint32).reshape(-1, arena.n1)
        _check(lib().iyk_hip_arena_upload(self.h, arena.ptr, arena.slots, first_slot, host.shape[0],
                                          host.ctypes.data_as(_u32p)), "iyk_hip_arena_upload")
        self.sync()  # host buffer is pageable: finish before it can be garbage collected

    def download(self, arena, first_slot, count):
        out = np.zeros((count, arena.n1), dtype=np.uint32)
        _check(lib().iyk_hip_arena_download(self.h, arena.ptr, arena.slots, first_slot, count,
                                            out.ctypes.data_as(_u32p)), "iyk_hip_arena_download")
        self.sync()
        return out

    def upload_slots(self, arena, slots, host):
        """Mem::set of many cells in one transfer: host row j -> arena slot slots[j]."""
        slots = _i32(slots)
        host = np.ascontiguousarray(host, dtype=np.uint32).reshape(len(slots), arena.n1)
        _check(lib().iyk_hip_arena_upload_slots(self.h, arena.ptr, arena.slots, len(slots), slots.ctypes.data_as(_i32p),
                                                host.ctypes.data_as(_u32p)), "iyk_hip_arena_upload_slots")

    def download_slots(self, arena, slots):
        """Mem::get of many cells in one transfer."""
        slots = _i32(slots)
        out = np.zeros((len(slots), arena.n1), dtype=np.uint32)
        _check(lib().iyk_hip_arena_download_slots(self.h, arena.ptr, arena.slots, len(slots),
                                                  slots.ctypes.data_as(_i32p), out.ctypes.data_as(_u32p)),
               "iyk_hip_arena_download_slots")
        self.sync()
        return out

    def arena_copy(self, dst, dst_first, src, src_first, count):
        _check(lib().iyk_hip_arena_copy(self.h, dst.ptr, dst.slots, dst_first, src.ptr, src.slots, src_first, count),
               "iyk_hip_arena_copy")

    def sync_slots_to(self, src_arena, dst_stream, dst_arena, slots):
        """Level-boundary exchange between in-process GPU replicas: listed slots of src_arena (this stream's GPU)
        -> same slots of dst_arena (dst_stream's GPU), ordered by events on both streams."""
        slots = _i32(slots)
        _check(lib().iyk_hip_arena_sync_slots(self.h, src_arena.ptr, src_arena.slots, dst_stream.h, dst_arena.ptr,
                                              dst_arena.slots, len(slots), slots.ctypes.data_as(_i32p)),
               "iyk_hip_arena_sync_slots")

    def sync_slots_to_many(self, src_arena, dst_streams, dst_arenas, slots):
        """The same with ONE gather on this stream's GPU and several destination replicas (iyk_hip_arena_sync_slots_multi)."""
        slots = _i32(slots)
        n = len(dst_streams)
        assert n == len(dst_arenas)
        sts = (_vp * n)(*[s.h for s in dst_streams])
        ptrs = (_vp * n)(*[a.ptr for a in dst_arenas])
        caps = (ctypes.c_uint64 * n)(*[a.slots for a in dst_arenas])
        _check(lib().iyk_hip_arena_sync_slots_multi(self.h, src_arena.ptr, src_arena.slots, n, sts, ptrs, caps, len(slots),
                                                    slots.ctypes.data_as(_i32p)), "iyk_hip_arena_sync_slots_multi")

    def gate_batch(self, arena, ops, in0, in1, in2, out):
        """`len(ops)` independent gates on arena slots; asynchronous (poll query() / sync())."""
        ops, in0, in1, in2, out = map(_i32, (ops, in0, in1, in2, out))
        n = len(ops)
        assert len(in0) == len(in1) == len(in2) == len(out) == n
        p = lambda a: a.ctypes.data_as(_i32p)
        _check(lib().iyk_hip_gate_batch(self.h, arena.ptr, arena.slots, n, p(ops), p(in0), p(in1), p(in2), p(out)),
               "iyk_hip_gate_batch")

    def gate_host(self, op, in0=None, in1=None, in2=None):
        """cufhe::Nand(out, in0, in1, st) shape: host in, host out (synchronised here)."""
        n1 = current_params().n + 1
        out = np.zeros(n1, dtype=np.uint32)
        args = [None if a is None else np.ascontiguousarray(a, dtype=np.uint32) for a in (in0, in1, in2)]
        ptr = lambda a: a.ctypes.data_as(_u32p) if a is not None else None
        code = OPS[op] if isinstance(op, str) else int(op)
        _check(lib().iyk_hip_gate_host(self.h, code, ptr(args[0]), ptr(args[1]), ptr(args[2]),
                                       out.ctypes.data_as(_u32p)), "iyk_hip_gate_host")
        self.sync()
        return out

    def blind_rotate_batch(self, arena, ia, ib, sa, sb, off, d_tlwe1_ptr):
        ia, ib, sa, sb = map(_i32, (ia, ib, sa, sb))
        off = np.ascontiguousarray(off, dtype=np.uint32)
        p = lambda a: a.ctypes.data_as(_i32p)
        _check(lib().iyk_hip_blind_rotate_batch(self.h, arena.ptr, arena.slots, len(ia), p(ia), p(ib), p(sa), p(sb),
                                                off.ctypes.data_as(_u32p), _vp(int(d_tlwe1_ptr))),
               "iyk_hip_blind_rotate_batch")

    def bootstrap_trlwe_batch(self, arena, ia, ib, sa, sb, off, d_trlwe_ptr, trlwe_slots=None, trlwe_out=None):
        """GateBootstrappingTLWE2TRLWElvl01NTT shape: rotation only, TRLWE (2N words) per job, written to row
        trlwe_out[job] of the TRLWE buffer (row `job` when trlwe_out is None)."""
        ia, ib, sa, sb = map(_i32, (ia, ib, sa, sb))
        off = np.ascontiguousarray(off, dtype=np.uint32)
        p = lambda a: a.ctypes.data_as(_i32p)
        to = None if trlwe_out is None else _i32(trlwe_out)
        _check(lib().iyk_hip_bootstrap_trlwe_batch(self.h, arena.ptr, arena.slots, len(ia), p(ia), p(ib), p(sa), p(sb),
                                                   off.ctypes.data_as(_u32p), _vp(int(d_trlwe_ptr)),
                                                   len(ia) if trlwe_slots is None else int(trlwe_slots),
                                                   None if to is None else p(to)),
               "iyk_hip_bootstrap_trlwe_batch")

    def sample_extract_keyswitch_batch(self, d_trlwe_ptr, trlwe_index, out_slot, arena, trlwe_slots=None):
        """SampleExtractAndKeySwitch shape: TRLWE -> TLWE lvl0 into arena slots."""
        ti, os_ = _i32(trlwe_index), _i32(out_slot)
        p = lambda a: a.ctypes.data_as(_i32p)
        nt = (int(ti.max()) + 1 if len(ti) else 0) if trlwe_slots is None else int(trlwe_slots)
        _check(lib().iyk_hip_sample_extract_keyswitch_batch(self.h, _vp(int(d_trlwe_ptr)), nt, len(ti), p(ti), p(os_),
                                                            arena.ptr, arena.slots),
               "iyk_hip_sample_extract_keyswitch_batch")

    def last_batch_timing(self):
        """(blind_rotate_ms, keyswitch_ms) of the most recent batch, from HIP events on this stream."""
        br, ks = ctypes.c_float(), ctypes.c_float()
        _check(lib().iyk_hip_last_batch_timing(self.h, ctypes.byref(br), ctypes.byref(ks)),
               "iyk_hip_last_batch_timing")
        return br.value, ks.value

    def timing_log_begin(self):
        _check(lib().iyk_hip_timing_log_begin(self.h), "iyk_hip_timing_log_begin")

    def timing_log_end(self):
        """(batches, blind_rotate_ms_total, keyswitch_ms_total) since timing_log_begin(); synchronises."""
        nb, br, ks = ctypes.c_uint64(), ctypes.c_double(), ctypes.c_double()
        _check(lib().iyk_hip_timing_log_end(self.h, ctypes.byref(nb), ctypes.byref(br), ctypes.byref(ks)),
               "iyk_hip_timing_log_end")
        return nb.value, br.value, ks.value
