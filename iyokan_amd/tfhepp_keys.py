"""Import of TFHEpp's own key archives: `TFHEpp::SecretKey` (iyokan-packet genkey) and `TFHEpp::EvalKey`
(iyokan-packet genevalkey) — the files a stock Iyokan deployment already has.

Reference call sites: the frontend starts from `readFromArchive<TFHEpp::EvalKey>(pr_.ekFile)`
(/root/reference/src/iyokan_cufhe.cpp:546-549), written by `doGenEvalKeyTFHEpp` (/root/reference/src/iyokan-packet.cpp:150-160:
iksk<lvl10param>, bk<lvl01param>, bkfft<lvl01param>, bkfft<lvl02param>, privksk4cb<lvl21param>) through cereal's
PortableBinaryOutputArchive (/root/reference/src/packet.hpp:325-344); `cufhe::Initialize(ek)` then needs exactly
`ek.getbk<lvl01param>()` (torus-domain BK) and `ek.getiksk<lvl10param>()` (/root/reference/src/iyokan_cufhe.cpp:530-536,734),
which is what iyk_hip_init takes.

PIN STATUS: UNVERIFIED AGAINST REAL TFHEpp.  TFHEpp is an empty submodule of the reference tree and its pinned version is
unknown, so the member order of `EvalKey::serialize` / `lweParams` cannot be read here.  What IS fixed by cereal and by
the reference's call sites, and all this reader relies on:
  * the archive starts with one byte, 1 = little-endian payload (cereal PortableBinary);
  * every key object hangs off a smart pointer, serialised as a u32 id: 0 = null, `0x80000000 | k` (k = 1, 2, ... in order
    of first appearance) = "object follows"  (cereal/types/memory.hpp, shared_ptr wrapper);
  * `BootstrappingKey<lvl01param>` = std::array<TRGSW<lvl1param>, n> of nested std::arrays of uint32 = n (k+1)l (k+1) N raw
    words, no size tags (cereal writes arithmetic std::arrays as one binary blob) — the layout of iyk_hip_init's bk_torus;
  * `KeySwitchingKey<lvl10param>` = std::array<std::array<std::array<TLWE<lvl0param>, 2^basebit - 1>, t>, N> =
    N t (2^basebit - 1) (n + 1) raw words — the layout of iyk_hip_init's ksk.
So the reader does not walk the struct: it SEARCHES the archive for an "object follows" id word that is followed by exactly
a blob of the wanted size and then by another plausible cereal item (a pointer id word, or a size tag of an unordered_map):
`find_blob`.  Both blobs have sizes no other member shares (bkfft<lvl01> is twice bk<lvl01>: doubles).  A false positive
needs two 32-bit coincidences at a fixed distance: < 10^-9 per archive.  When the secret key is at hand, `verify` decrypts
sample rows of both keys and checks message and noise — that check is cryptographic and does not depend on any recollection.
ASSUMED (stated, checked where possible): binary secret keys stored one uint32 per bit (`SecretKey` = key.lvl0 (n words),
key.lvl1 (N words), key.lvl2, params, in this order, right after the endianness byte); the parameter sets of
include/iyokan_hip_params.h.  tools/tfhepp_crosscheck.cpp writes both archives with real cereal + TFHEpp where a checkout
exists, so that whoever has one validates this reader and the oracle's conventions in one go.
"""
import struct

import numpy as np

PTR_NEW = 0x80000000
MAX_IDS = 64


class KeyImportError(ValueError):
    pass


def _u32(data, off):
    return struct.unpack_from("<I", data, off)[0]


def _plausible_next(data, off):
    """What may follow a key blob: end of archive, a pointer id (0 or 0x80000000 | small), or a u64 size tag of a map."""
    if off == len(data):
        return True
    if off + 4 > len(data):
        return False
    w = _u32(data, off)
    if w == 0 or (w & PTR_NEW and (w & 0x7FFFFFFF) <= MAX_IDS):
        return True
    if off + 8 <= len(data):
        return struct.unpack_from("<Q", data, off)[0] <= 4096     # unordered_map<string, ptr> with a few entries
    return False


def find_blob(data, nbytes, what):
    """Offsets o such that data[o-4:o] is an 'object follows' pointer id and data[o:o+nbytes] is followed by a plausible
    cereal item.  Exactly one is expected."""
    mv = memoryview(data)
    arr = np.frombuffer(mv, dtype=np.uint8)
    # candidate id words: bytes (k, 0, 0, 0x80), k = 1 .. MAX_IDS, at any byte offset (cereal does not align)
    hits = np.nonzero((arr[3:] == 0x80) & (arr[2:-1] == 0) & (arr[1:-2] == 0) & (arr[:-3] >= 1) & (arr[:-3] <= MAX_IDS))[0]
    found = [int(h) + 4 for h in hits if int(h) + 4 + nbytes <= len(data) and _plausible_next(data, int(h) + 4 + nbytes)]
    if len(found) != 1:
        raise KeyImportError(f"TFHEpp archive: expected exactly one {what} ({nbytes} bytes behind a cereal pointer id), "
                             f"found {len(found)}; is this an EvalKey of another parameter set?")
    return found[0]


def read_eval_key(data, params):
    """(bk_torus, ksk) of a `TFHEpp::EvalKey` archive for `params` (iyokan_amd.params.IykParams): uint32 arrays in the
    layouts iyk_hip_init takes."""
    if len(data) < 1 or data[0] not in (0, 1):
        raise KeyImportError("TFHEpp archive: bad endianness flag")
    if data[0] != 1:
        raise KeyImportError("TFHEpp archive: big-endian archives are not supported")
    bk_bytes, ksk_bytes = 4 * params.bk_words, 4 * params.ksk_words
    ob = find_blob(data, bk_bytes, "bk<lvl01param>")
    ok = find_blob(data, ksk_bytes, "iksk<lvl10param>")
    if not (ob + bk_bytes <= ok - 4 or ok + ksk_bytes <= ob - 4):
        raise KeyImportError("TFHEpp archive: the two key blobs overlap")
    bk = np.frombuffer(data, dtype="<u4", count=params.bk_words, offset=ob).astype(np.uint32)
    ksk = np.frombuffer(data, dtype="<u4", count=params.ksk_words, offset=ok).astype(np.uint32)
    return bk, ksk


def read_secret_key(data, params):
    """(s0, s1) of a `TFHEpp::SecretKey` archive: n + N words right behind the endianness byte, every one 0 or 1."""
    need = 1 + 4 * (params.n + params.k * params.N)
    if len(data) < need or data[0] != 1:
        raise KeyImportError("TFHEpp archive: not a little-endian SecretKey of this parameter set")
    s0 = np.frombuffer(data, dtype="<u4", count=params.n, offset=1).astype(np.uint32)
    s1 = np.frombuffer(data, dtype="<u4", count=params.k * params.N, offset=1 + 4 * params.n).astype(np.uint32)
    if s0.max(initial=0) > 1 or s1.max(initial=0) > 1:
        raise KeyImportError("TFHEpp archive: secret key words are not binary (another parameter set, or a key format "
                             "this reader does not know)")
    return s0, s1


def verify(params, s0, s1, bk, ksk, rows=64):
    """Cryptographic check of imported keys against the secret key: sample KSK rows must decrypt to s1[i] v 2^(32-(j+1) basebit)
    and sample TRGSW rows to s0[i] 2^(32-(j+1) Bgbit) on polynomial c, both within 6 sigma of their noise parameter."""
    p = params
    n1 = p.n + 1
    nb = (1 << p.basebit) - 1
    rng = np.random.default_rng(1)
    s0u = s0.astype(np.uint32)
    kr = ksk.reshape(p.N, p.t, nb, n1)
    for _ in range(rows):
        i, j, v = int(rng.integers(p.N)), int(rng.integers(p.t)), int(rng.integers(nb))
        row = kr[i, j, v]
        ph = (int(row[-1]) - int((row[:-1] * s0u).sum(dtype=np.uint32))) & 0xFFFFFFFF
        msg = (int(s1[i]) * (v + 1) << (32 - (j + 1) * p.basebit)) & 0xFFFFFFFF
        d = (ph - msg) & 0xFFFFFFFF
        d = d - (1 << 32) if d >= 1 << 31 else d
        if abs(d) / 2.0 ** 32 > 6 * p.alpha0 + 2.0 ** -31:
            raise KeyImportError(f"iksk row ({i}, {j}, {v + 1}) does not decrypt under this secret key")
    br = bk.reshape(p.n, (p.k + 1) * p.l, p.k + 1, p.N)
    s1i = s1.astype(np.int64)
    for _ in range(rows):
        i, r = int(rng.integers(p.n)), int(rng.integers((p.k + 1) * p.l))
        a, b = br[i, r, 0].astype(np.int64), br[i, r, 1].astype(np.int64)
        as0 = (a[0] * s1i[0] - (a[1:] * s1i[:0:-1]).sum()) & 0xFFFFFFFF           # (a * s1)[0] mod X^N + 1
        c, j = divmod(r, p.l)
        m = int(s0[i]) << (32 - (j + 1) * p.Bgbit)
        want = m if c == 1 else (-m * int(s1i[0]))
        d = (int(b[0]) - int(as0) - want) & 0xFFFFFFFF
        d = d - (1 << 32) if d >= 1 << 31 else d
        if abs(d) / 2.0 ** 32 > 6 * p.alpha1 + 2.0 ** -31:
            raise KeyImportError(f"bk row ({i}, {r}) does not decrypt under this secret key")
    return True


# ---- writers that follow the SAME cereal rules (tests, and the shape tools/tfhepp_crosscheck.cpp produces with real cereal) ----
def _ptr(k):
    return struct.pack("<I", PTR_NEW | k)


NULL = struct.pack("<I", 0)


def write_eval_key_like(params, bk, ksk, params_blob=b"", extra_before=(), extra_between=(), extra_after=()):
    """An archive with the structure cereal gives an EvalKey-like struct: endianness byte, an opaque `lweParams` blob, then
    pointer members in some order — `extra_*` are raw byte strings standing for the other members (null pointers, bkfft
    blobs behind their ids, maps) —, bk<lvl01> and iksk<lvl10> behind "object follows" pointer ids.  The member order of
    real TFHEpp is unknown; the reader must find the two blobs wherever they sit."""
    out = [b"\x01", params_blob] + list(extra_before)
    out += [_ptr(10), np.ascontiguousarray(bk, dtype="<u4").tobytes()] + list(extra_between)
    out += [_ptr(11), np.ascontiguousarray(ksk, dtype="<u4").tobytes()] + list(extra_after)
    return b"".join(out)


def write_secret_key_like(params, s0, s1, tail=b""):
    """endianness byte, key.lvl0 (n words), key.lvl1 (N words), then whatever follows (key.lvl2, params)."""
    return b"\x01" + np.ascontiguousarray(s0, dtype="<u4").tobytes() + np.ascontiguousarray(s1, dtype="<u4").tobytes() + tail
