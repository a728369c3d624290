"""ctypes binding of oracle/libiyk_oracle.so — test infrastructure only (see oracle/tfhe_oracle.c)."""
import ctypes
import os

import numpy as np

from iyokan_amd.params import IykParams

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_u32p = ctypes.POINTER(ctypes.c_uint32)
_i32p = ctypes.POINTER(ctypes.c_int32)
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_ROOT, "oracle", "libiyk_oracle.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: run `make -C oracle`")
        L = ctypes.CDLL(path)
        L.orc_new.restype = ctypes.c_void_p
        L.orc_new.argtypes = [ctypes.POINTER(IykParams), _u32p, _u32p]
        L.orc_free.argtypes = [ctypes.c_void_p]
        L.orc_gate.argtypes = [ctypes.c_void_p, ctypes.c_int, _u32p, _u32p, _u32p, _u32p, ctypes.c_int]
        L.orc_gate_batch.argtypes = [ctypes.c_void_p, ctypes.c_uint32, _i32p, _i32p, _i32p, _i32p, _i32p, _u32p, ctypes.c_int]
        L.orc_gate_batch_mode.argtypes = [ctypes.c_void_p, ctypes.c_uint32, _i32p, _i32p, _i32p, _i32p, _i32p, _u32p,
                                          ctypes.c_int, ctypes.c_int]
        L.orc_has_fp.argtypes = [ctypes.c_void_p]
        L.orc_has_fft.argtypes = [ctypes.c_void_p]
        L.orc_fft_rounding_distance.restype = ctypes.c_double
        L.orc_fft_rounding_distance.argtypes = [ctypes.c_void_p]
        L.orc_blind_rotate.argtypes = [ctypes.c_void_p, _u32p, _u32p, ctypes.c_int]
        L.orc_sample_extract0.argtypes = [ctypes.c_void_p, _u32p, _u32p]
        L.orc_keyswitch.argtypes = [ctypes.c_void_p, _u32p, _u32p]
        L.orc_tlwe1_phase.restype = ctypes.c_uint32
        L.orc_tlwe1_phase.argtypes = [ctypes.POINTER(IykParams), _u32p, _u32p]
        L.orc_selfcheck_product.argtypes = [ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint64]
        L.orc_selfcheck_field.argtypes = [ctypes.c_uint64, ctypes.c_uint32]
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(_u32p) if a is not None else None


class Oracle:
    def __init__(self, keys):
        self.keys = keys
        self.p = keys.params
        self.ctx = lib().orc_new(ctypes.byref(self.p), _p(keys.bk), _p(keys.ksk))

    def close(self):
        if self.ctx:
            lib().orc_free(self.ctx)
            self.ctx = None

    def __del__(self):
        self.close()

    def gate(self, op, in0=None, in1=None, in2=None, schoolbook=False):
        out = np.zeros(self.p.n + 1, dtype=np.uint32)
        args = [None if a is None else np.ascontiguousarray(a, dtype=np.uint32) for a in (in0, in1, in2)]
        lib().orc_gate(self.ctx, int(op), _p(args[0]), _p(args[1]), _p(args[2]), _p(out), int(schoolbook))
        return out

    # 0 .. 3: exact restatements, word-equal to each other.  4: TFHEpp's ALGORITHM (unsplit key, inexact FP64 products, two gates per
    # call; binary gates only) — decrypt-equal, NOT word-equal; timing only (bench.py's cpu_baseline), never a parity reference
    MODES = {"goldilocks": 0, "schoolbook": 1, "fp": 2, "fft": 3, "tfhepp_algorithm_inexact": 4}

    def has_fft(self):
        """True when the split-key complex-FFT restatement (oracle/tfhe_oracle_fft.c, the GPU's arithmetic) covers this set."""
        return bool(lib().orc_has_fft(self.ctx))

    def fft_rounding_distance(self):
        """Largest |x - round(x)| any rounding of the FFT restatement has seen so far (it aborts above 1/4)."""
        return float(lib().orc_fft_rounding_distance(self.ctx))

    def has_fp(self):
        """True when the FP64-field restatement (oracle/tfhe_oracle_fp.c) is exact for this parameter set."""
        return bool(lib().orc_has_fp(self.ctx))

    def gate_batch(self, ops, in0, in1, in2, out, arena, nthreads=1, mode="goldilocks"):
        """Same addressing as iyk_hip_gate_batch; arena (slots, n+1) uint32 is updated in place.
        mode: which of the oracle's exact product implementations runs the blind rotation."""
        conv = lambda a: np.ascontiguousarray(a, dtype=np.int32)
        ops, in0, in1, in2, out = map(conv, (ops, in0, in1, in2, out))
        assert arena.dtype == np.uint32 and arena.flags["C_CONTIGUOUS"]
        ip = lambda a: a.ctypes.data_as(_i32p)
        lib().orc_gate_batch_mode(self.ctx, len(ops), ip(ops), ip(in0), ip(in1), ip(in2), ip(out), _p(arena), nthreads,
                                  self.MODES[mode])
        return arena

    def bootstrap_lvl1(self, lin, schoolbook=False, mode=None):
        """blind rotate + sample extract(0): lvl0 TLWE -> lvl1 TLWE (N+1 words)."""
        lin = np.ascontiguousarray(lin, dtype=np.uint32)
        acc = np.zeros((self.p.k + 1) * self.p.N, dtype=np.uint32)
        lib().orc_blind_rotate(self.ctx, _p(lin), _p(acc), int(schoolbook) if mode is None else self.MODES[mode])
        t1 = np.zeros(self.p.N + 1, dtype=np.uint32)
        lib().orc_sample_extract0(self.ctx, _p(acc), _p(t1))
        return t1

    def keyswitch(self, tlwe1):
        tlwe1 = np.ascontiguousarray(tlwe1, dtype=np.uint32)
        out = np.zeros(self.p.n + 1, dtype=np.uint32)
        lib().orc_keyswitch(self.ctx, _p(tlwe1), _p(out))
        return out

    def tlwe1_phase(self, t1):
        t1 = np.ascontiguousarray(t1, dtype=np.uint32)
        return lib().orc_tlwe1_phase(ctypes.byref(self.p), _p(t1), _p(self.keys.s1))


class OracleBackend:
    """Ciphertext arena in host memory, gates evaluated by the CPU oracle: the frontier executor's backend
    interface (gate_batch / write / read ...) so that whole netlists run ENCRYPTED without a GPU.
    Test infrastructure only — BASELINE config #1 ("test0-equivalent self-check on the CPU oracle")."""

    def __init__(self, num_slots, oracle, nthreads=None):
        self.oracle = oracle
        self.arena = np.zeros((num_slots, oracle.p.n + 1), dtype=np.uint32)
        self.nthreads = nthreads or (os.cpu_count() or 1)

    def gate_batch(self, ops, in0, in1, in2, out):
        if len(ops):
            self.oracle.gate_batch(ops, in0, in1, in2, out, self.arena, nthreads=self.nthreads)

    def write(self, slot, tlwe):
        self.arena[slot] = np.asarray(tlwe, dtype=np.uint32)

    def write_many(self, slots, rows):
        self.arena[np.asarray(slots, dtype=np.int64)] = np.asarray(rows, dtype=np.uint32).reshape(len(slots), -1)

    def read(self, slot):
        return self.arena[slot].copy()

    def read_many(self, slots):
        return self.arena[np.asarray(slots, dtype=np.int64)].copy()

    def sync(self):
        pass


def adversarial_rows(n, seed=9):
    """TLWE-shaped rows of n + 1 words that no encryption would produce but the deterministic pipeline must still map
    identically everywhere: all-ones, the sign bit, the mod-switch rounding threshold and its neighbours (a' = (a +
    2^20) >> 21 for N = 1024), alternating extremes, a single non-zero coefficient, and uniform words.  Parity between
    the oracle's restatements and the HIP kernels on these rows exercises digit extremes (-Bg/2, Bg/2 - 1), exponent
    wrap-around (abar = 0, 2N - 1) and skipped CMUX steps without relying on how likely fresh ciphertexts hit them."""
    import numpy as np

    rng = np.random.default_rng(seed)
    rows = [
        np.full(n + 1, 0xFFFFFFFF, dtype=np.uint32),
        np.full(n + 1, 0x80000000, dtype=np.uint32),
        np.full(n + 1, 0x000FFFFF, dtype=np.uint32),            # just below the rounding threshold: abar = 0 everywhere
        np.full(n + 1, 0x00100000, dtype=np.uint32),            # exactly on it: abar = 1
        np.full(n + 1, 0xFFF00000, dtype=np.uint32),            # rounds up to 2N: abar wraps to 0
        np.where(np.arange(n + 1) % 2 == 0, 0, 0xFFFFFFFF).astype(np.uint32),
        np.zeros(n + 1, dtype=np.uint32),
        rng.integers(0, 2**32, size=n + 1, dtype=np.uint64).astype(np.uint32),
        rng.integers(0, 2**32, size=n + 1, dtype=np.uint64).astype(np.uint32),
    ]
    rows[6][n // 2] = 0x7FE00000                                # one coefficient, abar = 1023
    rows[6][n] = 0x40000000
    return np.stack(rows)
