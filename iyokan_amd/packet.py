"""Plain request / result packets in their TOML form.

The reference exchanges `PlainPacket`s (/root/reference/src/packet.hpp:193-223): named bit vectors
`bits`, RAM images `ram`, ROM images `rom`, and `numCycles`; `iyokan-packet toml2packet / packet2toml`
(/root/reference/src/iyokan-packet.cpp:28-142) convert them from / to TOML:

    cycles = 8                      # optional, -1 / absent = unspecified
    [[bits]]  name = "addr"  size = 64  bytes = [ 0x1d, ... ]     # bit i = bytes[i / 8] >> (i % 8) & 1
    [[ram]]   name = "ram"   size = 4096 bytes = [ ... ]
    [[rom]]   name = "rom"   size = 4096 bytes = [ ... ]

The binary form the reference's tools exchange (`readFromArchive` / `writeToArchive`,
/root/reference/src/packet.hpp:287-344: cereal::PortableBinary{Input,Output}Archive) is restated in
`from_archive` / `to_archive` from cereal's published encoding rules (cereal is not available here and the
reference tree holds no binary fixture: the tests round-trip self-written archives, a hand-assembled byte string, and
archives written by the C++ twin iyokan_amd/host/packet.hpp):
    1 byte  1 = little-endian payload | struct = its serialize() arguments in order | bool / Bit = 1 byte |
    size tag = u64 | std::string = size tag + bytes | std::vector<T> = size tag + elements |
    std::array<u32, K> = K raw words (TLWE lvl0: n+1; TRLWE lvl1: 2N) | unordered_map = size tag + (key, value)... |
    std::optional<int> = 1 byte `nullopt` flag (1 = empty) + the int32 when present.
"""
import struct

import numpy as np
import tomli

from .netlist import bits_from_bytes, bytes_from_bits


class PlainPacket:
    def __init__(self, bits=None, ram=None, rom=None, cycles=None):
        self.bits = dict(bits or {})   # name -> list of 0/1
        self.ram = dict(ram or {})
        self.rom = dict(rom or {})
        self.cycles = cycles           # None = unspecified

    # ---- TOML ------------------------------------------------------------------------
    @classmethod
    def from_toml_dict(cls, d):
        """doToml2Packet (/root/reference/src/iyokan-packet.cpp:191-233): `size` bits, filled from `bytes`
        lsb first; missing bytes read as 0, surplus bytes are ignored, only the low 8 bits of a byte count."""
        pkt = cls(cycles=d.get("cycles"))
        if pkt.cycles is not None and pkt.cycles < 0:
            pkt.cycles = None
        for kind in ("ram", "rom", "bits"):
            table = getattr(pkt, kind)
            for entry in d.get(kind, []):
                name, size, data = entry["name"], entry["size"], entry["bytes"]
                if not isinstance(name, str) or not isinstance(size, int) or size < 0:
                    raise ValueError(f"Invalid packet: bad {kind} entry {name!r}")
                table[name] = bits_from_bytes([int(b) & 0xFF for b in data], size)
        return pkt

    @classmethod
    def load(cls, path):
        with open(path, "rb") as f:
            return cls.from_toml_dict(tomli.load(f))

    def to_toml(self):
        out = []
        if self.cycles is not None:
            out.append(f"cycles = {self.cycles}")
        for kind in ("rom", "ram", "bits"):
            for name in sorted(getattr(self, kind)):
                bits = getattr(self, kind)[name]
                data = ", ".join(str(b) for b in bytes_from_bits(bits))
                out.append(f'[[{kind}]]\nname = "{name}"\nsize = {len(bits)}\nbytes = [{data}]')
        return "\n".join(out) + "\n"

    # ---- cereal PortableBinary ------------------------------------------------------------
    def to_archive(self) -> bytes:
        """writeToArchive(os, PlainPacket): serialize() = ar(ram, rom, bits, numCycles)."""
        out = [b"\x01"]
        for table in (self.ram, self.rom, self.bits):
            out.append(struct.pack("<Q", len(table)))
            for name, bits in table.items():
                key = name.encode()
                out.append(struct.pack("<Q", len(key)) + key + struct.pack("<Q", len(bits)) + bytes(int(b) & 1 for b in bits))
        out.append(b"\x01" if self.cycles is None else b"\x00" + struct.pack("<i", self.cycles))
        return b"".join(out)

    @classmethod
    def from_archive(cls, data: bytes):
        r = _ArchiveReader(data)
        pkt = cls()
        for table in (pkt.ram, pkt.rom, pkt.bits):
            for _ in range(r.size(1 << 20)):
                name = r.string()
                raw = r.take(r.size(1 << 32))
                if any(b > 1 for b in raw):
                    raise ValueError("Invalid archive: Bit value out of range")
                if name in table:
                    raise ValueError("Invalid archive: duplicate key")
                table[name] = list(raw)
        pkt.cycles = r.opt_int()
        r.expect_end()
        return pkt

    @staticmethod
    def convert(sources, rules):
        """doConvertPlain (/root/reference/src/iyokan-packet.cpp:268-296): every rule
        "(ram|rom|bits).NEW = PKT.OLD" copies entry OLD of the same kind from the packet named PKT."""
        import re

        pat = re.compile(r"(ram|rom|bits)\.([a-zA-Z0-9]+)\s*=\s*([a-zA-Z0-9]+)\.([a-zA-Z0-9]+)")
        out = PlainPacket()
        for rule in rules:
            m = pat.fullmatch(rule)
            if not m:
                raise ValueError(f"Invalid assignment: {rule}")
            kind, new, pkt, old = m.groups()
            getattr(out, kind).setdefault(new, list(getattr(sources[pkt], kind)[old]))
        return out

    # ---- comparison as the reference's test driver does it (test.rb assert_equal_packet) ----
    def same_content(self, other, check_cycles=True):
        """toml2packet(got) == toml2packet(expected) of the reference's test driver (test.rb:34-68): the same
        named entries with the same (size, zero-padded bytes), and the same `cycles` (absent = -1)."""
        if check_cycles and self.cycles != other.cycles:
            return False
        for kind in ("bits", "ram", "rom"):
            a, b = getattr(self, kind), getattr(other, kind)
            if set(a) != set(b):
                return False
            for name in a:
                if len(a[name]) != len(b[name]) or bytes_from_bits(a[name]) != bytes_from_bits(b[name]):
                    return False
        return True

    def diff(self, other):
        """Human-readable list of differences (for test failure messages)."""
        out = []
        if self.cycles != other.cycles:
            out.append(f"cycles: {self.cycles} != {other.cycles}")
        for kind in ("bits", "ram", "rom"):
            a, b = getattr(self, kind), getattr(other, kind)
            for name in sorted(set(a) | set(b)):
                if name not in a or name not in b:
                    out.append(f"{kind}.{name}: only on one side")
                elif len(a[name]) != len(b[name]) or a[name] != b[name]:
                    out.append(f"{kind}.{name}: {bytes_from_bits(a[name])[:8]}.. != {bytes_from_bits(b[name])[:8]}..")
        return out


class _ArchiveReader:
    def __init__(self, data):
        self.d, self.i = memoryview(bytes(data)), 0
        flag = self.take(1)[0]
        if flag > 1:
            raise ValueError("Invalid archive: bad endianness flag")
        self.end = "<" if flag == 1 else ">"

    def take(self, n):
        if self.i + n > len(self.d):
            raise ValueError("Invalid archive: truncated")
        v = self.d[self.i:self.i + n]
        self.i += n
        return v

    def size(self, limit):
        n = struct.unpack(self.end + "Q", self.take(8))[0]
        if n > limit:
            raise ValueError("Invalid archive: implausible size tag")
        return n

    def string(self):
        return bytes(self.take(self.size(1 << 20))).decode()

    def opt_int(self):
        flag = self.take(1)[0]
        if flag > 1:
            raise ValueError("Invalid archive: bad optional flag")
        return None if flag else struct.unpack(self.end + "i", self.take(4))[0]

    def words(self, n):
        return np.frombuffer(self.take(4 * n), dtype=self.end + "u4").astype(np.uint32)

    def expect_end(self):
        if self.i != len(self.d):
            raise ValueError("Invalid archive: trailing bytes")


class TFHEPacket:
    """Encrypted request / result packet (/root/reference/src/packet.hpp:208-223): `bits`, `ramInTLWE`, `romInTLWE`
    hold TLWE lvl0 rows (count, n+1), `ram` / `rom` TRLWE lvl1 rows (count, 2N) for the CMUX memories."""

    FIELDS = ("ram", "ramInTLWE", "rom", "romInTLWE", "bits")   # serialize() order

    def __init__(self, params, cycles=None):
        self.params, self.cycles = params, cycles
        for f in self.FIELDS:
            setattr(self, f, {})

    def _width(self, field):
        return 2 * self.params.N if field in ("ram", "rom") else self.params.n + 1

    @classmethod
    def encrypt(cls, keys, plain, seed=None, trlwe=True):
        """PlainPacket::encrypt (/root/reference/src/packet.hpp:225-262): every RAM / ROM image in BOTH forms — TLWE lvl0 rows
        (`ramInTLWE`, `romInTLWE`: what the MUX memories and this backend read) and TRLWE lvl1 (`ram`: one per bit, `rom`: N bits
        per ciphertext — what upstream's CMUX memories read), so that a request made here can drive either kind of run;
        trlwe=False leaves the TRLWE maps empty (a TRLWE lvl1 is 2N words = 8 KB per RAM bit, and its phases cost an O(N^2) product
        each: callers that only feed THIS backend — which reads the TLWE forms — should pass trlwe=False, as the C++ encryptPacket's
        default does)."""
        from . import client

        t = cls(keys.params, plain.cycles)
        k = 0
        sd = lambda: None if seed is None else seed + k
        for name, bits in plain.ram.items():
            k += 1
            t.ramInTLWE[name] = client.encrypt_bits(keys, bits, seed=sd())
            if trlwe:
                k += 1
                t.ram[name] = client.encrypt_ram_trlwe(keys, bits, seed=sd())
        for name, bits in plain.rom.items():
            k += 1
            t.romInTLWE[name] = client.encrypt_bits(keys, bits, seed=sd())
            if trlwe:
                k += 1
                t.rom[name] = client.encrypt_rom_trlwe(keys, bits, seed=sd())
        for name, bits in plain.bits.items():
            k += 1
            t.bits[name] = client.encrypt_bits(keys, bits, seed=sd())
        return t

    def decrypt(self, keys):
        """TFHEPacket::decrypt (:264-290): the TRLWE maps first, the TLWE maps fill in names the TRLWE maps do not have
        (unordered_map::emplace keeps the first); a ROM's TRLWE form decrypts to a multiple of N bits."""
        from . import client

        p = PlainPacket(cycles=self.cycles)
        for name, rows in self.ram.items():
            p.ram[name] = [int(b) for b in client.decrypt_ram_trlwe(keys, rows)]
        for name, rows in self.ramInTLWE.items():
            p.ram.setdefault(name, [int(b) for b in client.decrypt_bits(keys, rows)])
        for name, rows in self.rom.items():
            p.rom[name] = [int(b) for b in client.decrypt_rom_trlwe(keys, rows)]
        for name, rows in self.romInTLWE.items():
            p.rom.setdefault(name, [int(b) for b in client.decrypt_bits(keys, rows)])
        for name, rows in self.bits.items():
            p.bits[name] = [int(b) for b in client.decrypt_bits(keys, rows)]
        return p

    def to_archive(self) -> bytes:
        out = [b"\x01"]
        for f in self.FIELDS:
            table = getattr(self, f)
            out.append(struct.pack("<Q", len(table)))
            for name, rows in table.items():
                rows = np.ascontiguousarray(rows, dtype="<u4").reshape(-1, self._width(f))
                key = name.encode()
                out.append(struct.pack("<Q", len(key)) + key + struct.pack("<Q", rows.shape[0]) + rows.tobytes())
        out.append(b"\x01" if self.cycles is None else b"\x00" + struct.pack("<i", self.cycles))
        return b"".join(out)

    @classmethod
    def from_archive(cls, params, data: bytes):
        r = _ArchiveReader(data)
        t = cls(params)
        for f in cls.FIELDS:
            w = t._width(f)
            table = getattr(t, f)
            for _ in range(r.size(1 << 20)):
                name = r.string()
                count = r.size((1 << 34) // w)
                if name in table:
                    raise ValueError("Invalid archive: duplicate key")
                table[name] = r.words(count * w).reshape(count, w)
        t.cycles = r.opt_int()
        r.expect_end()
        return t
