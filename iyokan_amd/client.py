"""Key generation / bit encryption / decryption (ctypes over libiyokan_client.so).

Mirrors what `iyokan-packet genkey|genevalkey|enc|dec` does for the reference
(/root/reference/src/iyokan-packet.cpp:144-178); used to make synthetic, non-trivial inputs.

Randomness: `seed=None` (the default) draws keys / masks / noise from a ChaCha20 stream keyed with fresh
getrandom(2) entropy for every call — two encryptions never share a mask.  An explicit integer `seed` selects
a seeded, NON-cryptographic generator: reproducible fixtures for tests and benchmarks only (the same seed
reproduces the same masks, so never encrypt two messages meant to stay secret with one seed).
"""
import ctypes
import os

import numpy as np

from .params import IykParams

_LIB = None
_u32p = ctypes.POINTER(ctypes.c_uint32)
_u8p = ctypes.POINTER(ctypes.c_uint8)


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libiyokan_client.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
        lib = ctypes.CDLL(path)
        lib.iyk_client_keygen.argtypes = [ctypes.POINTER(IykParams), ctypes.c_uint64, ctypes.c_int, _u32p, _u32p, _u32p, _u32p]
        lib.iyk_client_encrypt_bits.argtypes = [ctypes.POINTER(IykParams), _u32p, ctypes.c_uint64, ctypes.c_int, _u8p,
                                                ctypes.c_uint64, _u32p]
        lib.iyk_client_decrypt_bits.argtypes = [ctypes.POINTER(IykParams), _u32p, _u32p, ctypes.c_uint64, _u8p]
        lib.iyk_client_phases.argtypes = [ctypes.POINTER(IykParams), _u32p, _u32p, ctypes.c_uint64, _u32p]
        lib.iyk_client_trivial.argtypes = [ctypes.POINTER(IykParams), ctypes.c_int, _u32p]
        lib.iyk_client_encrypt_trlwe.argtypes = [ctypes.POINTER(IykParams), _u32p, ctypes.c_uint64, ctypes.c_int, _u32p,
                                                 ctypes.c_uint64, _u32p]
        lib.iyk_client_trlwe_phases.argtypes = [ctypes.POINTER(IykParams), _u32p, _u32p, ctypes.c_uint64, _u32p]
        _LIB = lib
    return _LIB


def _p32(a):
    return a.ctypes.data_as(_u32p)


class KeySet:
    """SecretKey (s0, s1) + EvalKey material the GPU path needs (bk<lvl01> torus, iksk<lvl10>)."""

    def __init__(self, params, s0, s1, bk, ksk):
        self.params, self.s0, self.s1, self.bk, self.ksk = params, s0, s1, bk, ksk


def keygen(params: IykParams, seed=None) -> KeySet:
    """SecretKey + EvalKey material.  seed=None: OS entropy (CSPRNG); an int: reproducible fixture keys."""
    s0 = np.zeros(params.n, dtype=np.uint32)
    s1 = np.zeros(params.N, dtype=np.uint32)
    bk = np.zeros(params.bk_words, dtype=np.uint32)
    ksk = np.zeros(params.ksk_words, dtype=np.uint32)
    rc = _lib().iyk_client_keygen(ctypes.byref(params), 0 if seed is None else int(seed), int(seed is not None),
                                  _p32(s0), _p32(s1), _p32(bk), _p32(ksk))
    if rc != 0:
        raise RuntimeError(f"iyk_client_keygen failed: {rc}")
    return KeySet(params, s0, s1, bk, ksk)


def encrypt_bits(keys: KeySet, bits, seed=None) -> np.ndarray:
    """bootsSymEncrypt of a bit vector.  seed=None: fresh OS-keyed stream per call; an int: reproducible fixture."""
    bits = np.ascontiguousarray(np.asarray(bits, dtype=np.uint8).ravel())
    out = np.zeros((bits.size, keys.params.n + 1), dtype=np.uint32)
    _lib().iyk_client_encrypt_bits(ctypes.byref(keys.params), _p32(keys.s0), 0 if seed is None else int(seed),
                                   int(seed is not None), bits.ctypes.data_as(_u8p), bits.size, _p32(out))
    return out


def decrypt_bits(keys: KeySet, ct) -> np.ndarray:
    ct = np.ascontiguousarray(ct, dtype=np.uint32).reshape(-1, keys.params.n + 1)
    bits = np.zeros(ct.shape[0], dtype=np.uint8)
    _lib().iyk_client_decrypt_bits(ctypes.byref(keys.params), _p32(keys.s0), _p32(ct), ct.shape[0],
                                   bits.ctypes.data_as(_u8p))
    return bits


def encrypt_trlwe(keys: KeySet, msg, seed=None) -> np.ndarray:
    """trlweSymEncrypt<Lvl1> of message polynomials (count, N) torus words -> (count, 2N): a(X) then b(X)."""
    msg = np.ascontiguousarray(msg, dtype=np.uint32).reshape(-1, keys.params.N)
    out = np.zeros((msg.shape[0], 2 * keys.params.N), dtype=np.uint32)
    _lib().iyk_client_encrypt_trlwe(ctypes.byref(keys.params), _p32(keys.s1), 0 if seed is None else int(seed),
                                    int(seed is not None), _p32(msg), msg.shape[0], _p32(out))
    return out


def trlwe_phases(keys: KeySet, ct) -> np.ndarray:
    """b - a * s1 of TRLWE lvl1 rows (count, 2N) -> (count, N); trlweSymDecrypt is `(int32) phase > 0` per coefficient."""
    ct = np.ascontiguousarray(ct, dtype=np.uint32).reshape(-1, 2 * keys.params.N)
    out = np.zeros((ct.shape[0], keys.params.N), dtype=np.uint32)
    _lib().iyk_client_trlwe_phases(ctypes.byref(keys.params), _p32(keys.s1), _p32(ct), ct.shape[0], _p32(out))
    return out


def encrypt_ram_trlwe(keys: KeySet, bits, seed=None) -> np.ndarray:
    """encryptRAM (/root/reference/src/packet.hpp:104-118): one TRLWE per bit, +-mu in coefficient 0."""
    bits = np.asarray(bits, dtype=np.uint8).ravel()
    msg = np.zeros((bits.size, keys.params.N), dtype=np.uint32)
    mu = int(keys.params.mu)
    msg[:, 0] = np.where(bits == 1, np.uint32(mu), np.uint32((1 << 32) - mu))
    return encrypt_trlwe(keys, msg, seed)


def decrypt_ram_trlwe(keys: KeySet, ct) -> np.ndarray:
    """decryptRAM (:153-164): coefficient 0 of every TRLWE."""
    return (trlwe_phases(keys, ct)[:, 0].view(np.int32) > 0).astype(np.uint8)


def encrypt_rom_trlwe(keys: KeySet, bits, seed=None) -> np.ndarray:
    """encryptROM (:78-97): N bits per TRLWE, +-mu per coefficient, 0 beyond the last bit."""
    bits = np.asarray(bits, dtype=np.uint8).ravel()
    N = keys.params.N
    count = -(-bits.size // N)
    mu = int(keys.params.mu)
    msg = np.zeros(count * N, dtype=np.uint32)
    msg[: bits.size] = np.where(bits == 1, np.uint32(mu), np.uint32((1 << 32) - mu))
    return encrypt_trlwe(keys, msg.reshape(count, N), seed)


def decrypt_rom_trlwe(keys: KeySet, ct) -> np.ndarray:
    """decryptROM (:172-183): every coefficient of every TRLWE (the padding of the last block decrypts to noise signs)."""
    return (trlwe_phases(keys, ct).view(np.int32) > 0).astype(np.uint8).ravel()


def phases(keys: KeySet, ct) -> np.ndarray:
    ct = np.ascontiguousarray(ct, dtype=np.uint32).reshape(-1, keys.params.n + 1)
    out = np.zeros(ct.shape[0], dtype=np.uint32)
    _lib().iyk_client_phases(ctypes.byref(keys.params), _p32(keys.s0), _p32(ct), ct.shape[0], _p32(out))
    return out


def trivial(params: IykParams, bit: int) -> np.ndarray:
    out = np.zeros(params.n + 1, dtype=np.uint32)
    _lib().iyk_client_trivial(ctypes.byref(params), int(bit), _p32(out))
    return out
