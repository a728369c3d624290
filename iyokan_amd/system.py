"""Blueprint -> one flat netlist (the caller side of the hot path needed by BASELINE config #4).

Covers the subset of Iyokan's TOML blueprint that the CAHP-with-MUX-memories systems use
(/root/reference/src/iyokan.hpp:1691-1895; example /root/reference/test/config-toml/cahp-ruby-mux.toml):

  [[file]]    type = "yosys-json" | "iyokanl1-json", path, name
  [[builtin]] type = "mux-rom" (in_addr_width, out_rdata_width)       -> makeROMWithMUX  (:2517-2593)
              type = "mux-ram" (in_addr_width, in_wdata_width = out_rdata_width) -> precompiled 8/16/16 netlist (:2595-2628)
  [connect]   "dst/port[a:b]" = "src/port[a:b]"   internal edge (dst input <- src output)
              "dst/port"      = "@name[a:b]"      system input  @name drives dst's input port
              "@name[a:b]"    = "src/port[a:b]"   system output @name reads src's output port

All sub-netlists are merged into ONE `Netlist`, connected inputs become alias wires of their drivers,
so the plaintext simulator, the levelizer and the frontier executor run the whole system as a single
DAG per clock.  ROM cells are INPUT-like nodes (`nl.rom[index]`), RAM cells DFFs (`nl.ram[index]`).
"""
import os
import re

import tomli

from . import netlist as N

_PORT = re.compile(r"^(?:(?P<node>[^/@\[]+)/|@)(?P<port>[^\[\]/]+)(?:\[(?P<a>\d+)(?::(?P<b>\d+))?\])?$")


def _parse_ports(s):
    m = _PORT.match(s.strip())
    if not m:
        raise ValueError(f"Invalid port string: {s!r}")
    node = m.group("node")  # None for @ports
    port = m.group("port")
    if m.group("a") is None:
        bits = [0]
    elif m.group("b") is None:
        bits = [int(m.group("a"))]
    else:
        a, b = int(m.group("a")), int(m.group("b"))
        if b < a:
            raise ValueError(f"Invalid port range: {s!r}")
        bits = list(range(a, b + 1))
    return node, port, bits


def make_rom_with_mux(in_addr_width, out_rdata_width):
    """MUX-tree ROM: per output bit, 2^aw ROM cells reduced by aw levels of MUX(A=even, B=odd, S=addr[i])."""
    nl = N.Netlist()
    nl.rom = {}
    addr = []
    for i in range(in_addr_width):
        nid = nl.add("INPUT")
        nl.inputs[("addr", i)] = nid
        addr.append(nid)
    for bit in range(out_rdata_width):
        work = []
        for i in range(1 << in_addr_width):
            nid = nl.add("INPUT")
            nl.rom[bit + i * out_rdata_width] = nid
            work.append(nid)
        for i in range(in_addr_width):
            work = [nl.add("MUX", [work[j], work[j + 1], addr[i]]) for j in range(0, len(work), 2)]
        out = nl.add("OUTPUT", [work[0]])
        nl.outputs[("rdata", bit)] = out
    return nl


class System:
    """Merged netlist + where the blueprint's named things ended up."""

    def __init__(self, nl, at_inputs, at_outputs, rom, ram):
        self.nl = nl
        self.at_inputs = at_inputs      # (name, bit) -> node id (INPUT)
        self.at_outputs = at_outputs    # (name, bit) -> node id
        self.rom = rom                  # builtin name -> {index: node id}
        self.ram = ram                  # builtin name -> {index: node id}

    def at_width(self, table, name):
        return 1 + max(b for (p, b) in table if p == name)


def load_blueprint(path, mux_ram_json=None):
    base = os.path.dirname(os.path.abspath(path))
    with open(path, "rb") as f:
        bp = tomli.load(f)
    parts = {}
    for fdesc in bp.get("file", []):
        p = fdesc["path"] if os.path.isabs(fdesc["path"]) else os.path.join(base, fdesc["path"])
        if fdesc["type"] == "yosys-json":
            parts[fdesc["name"]] = N.load_yosys_json(p)
        elif fdesc["type"] == "iyokanl1-json":
            parts[fdesc["name"]] = N.load_iyokanl1_json(p)
        else:
            raise ValueError(f"Invalid file type: {fdesc['type']}")
    for b in bp.get("builtin", []):
        if b["type"] == "mux-rom":
            parts[b["name"]] = make_rom_with_mux(b["in_addr_width"], b["out_rdata_width"])
        elif b["type"] == "mux-ram":
            key = (b["in_addr_width"], b["in_wdata_width"], b["out_rdata_width"])
            if key != (8, 16, 16):
                raise ValueError(f"mux-ram {key}: only the precompiled 8/16/16 netlist is available")
            parts[b["name"]] = N.load_iyokanl1_json(mux_ram_json or os.path.join(base, "mux-ram-8-16-16.min.json"), ram_width=16)
        else:
            raise ValueError(f"unsupported builtin type {b['type']} (CMUX memories are out of scope)")

    # ---- merge: renumber every part into one node space ---------------------------------------
    nl = N.Netlist()
    offset, rom, ram = {}, {}, {}
    for name, part in parts.items():
        offset[name] = nl.num_nodes
        off = offset[name]
        for k, ins in zip(part.kinds, part.ins):
            nl.kinds.append(k)
            nl.ins.append([i + off for i in ins])
        for nid, v in part.dff_init.items():
            nl.dff_init[nid + off] = v
        if getattr(part, "rom", None):
            rom[name] = {idx: nid + off for idx, nid in part.rom.items()}
        if part.ram:
            ram[name] = {idx: nid + off for idx, nid in part.ram.items()}

    def in_node(node, port, bit):
        try:
            return parts[node].inputs[(port, bit)] + offset[node]
        except KeyError:
            raise ValueError(f"no input port {node}/{port}[{bit}]")

    def out_node(node, port, bit):
        try:
            return parts[node].outputs[(port, bit)] + offset[node]
        except KeyError:
            raise ValueError(f"no output port {node}/{port}[{bit}]")

    at_inputs, at_outputs = {}, {}
    for dst, src in bp.get("connect", {}).items():
        dnode, dport, dbits = _parse_ports(dst)
        snode, sport, sbits = _parse_ports(src)
        if len(dbits) != len(sbits):
            raise ValueError(f"Invalid connect: {dst} = {src}")
        for db, sb in zip(dbits, sbits):
            if dnode is None:                      # "@out" = "node/port"
                if snode is None:
                    raise ValueError(f"Invalid connect: {dst} = {src}")
                at_outputs.setdefault((dport, db), out_node(snode, sport, sb))
            elif snode is None:                    # "node/port" = "@in"
                at_inputs.setdefault((sport, sb), in_node(dnode, dport, db))
            else:                                  # internal edge: dst input becomes an alias of src output
                d = in_node(dnode, dport, db)
                nl.kinds[d] = "OUTPUT"
                nl.ins[d] = [out_node(snode, sport, sb)]
    # system-level port tables; unconnected sub-net inputs stay plain INPUT nodes (value 0 unless set)
    nl.inputs = dict(at_inputs)
    nl.outputs = dict(at_outputs)
    for name, cells in ram.items():
        nl.ram.update({(name, idx): nid for idx, nid in cells.items()})
    nl.validate()
    return System(nl, at_inputs, at_outputs, rom, ram)
