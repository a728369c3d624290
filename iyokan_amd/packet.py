"""Plain request / result packets in their TOML form.

The reference exchanges `PlainPacket`s (/root/reference/src/packet.hpp:193-223): named bit vectors
`bits`, RAM images `ram`, ROM images `rom`, and `numCycles`; `iyokan-packet toml2packet / packet2toml`
(/root/reference/src/iyokan-packet.cpp:28-142) convert them from / to TOML:

    cycles = 8                      # optional, -1 / absent = unspecified
    [[bits]]  name = "addr"  size = 64  bytes = [ 0x1d, ... ]     # bit i = bytes[i / 8] >> (i % 8) & 1
    [[ram]]   name = "ram"   size = 4096 bytes = [ ... ]
    [[rom]]   name = "rom"   size = 4096 bytes = [ ... ]

Only the TOML form is handled here: the reference's test fixtures are TOML, and the binary form
(cereal PortableBinary) has no fixture in the reference tree to pin a reader against.
"""
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
