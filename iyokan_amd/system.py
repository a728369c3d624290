"""Blueprint -> one flat netlist (the caller side of the hot path needed by BASELINE config #4).

Covers the subset of Iyokan's TOML blueprint that the CAHP-with-MUX-memories systems use
(/root/reference/src/iyokan.hpp:1691-1895; example /root/reference/test/config-toml/cahp-ruby-mux.toml):

  [[file]]    type = "yosys-json" | "iyokanl1-json", path, name
  [[builtin]] type = "mux-rom" (in_addr_width, out_rdata_width)       -> makeROMWithMUX  (:2517-2593)
              type = "mux-ram" (in_addr_width, in_wdata_width = out_rdata_width) -> makeRAMWithMUX (:2595-2762):
                       the precompiled 8/16/16 netlist when its JSON is at hand, else the generated DMUX/MUX form
              type = "rom" / "ram": the reference keeps these in CMUX memories evaluated by TFHEpp on the CPU
                       (out of scope); they have the same ports and the same observable
                       behaviour as the MUX forms (the reference's fixtures are shared between the two), so
                       here they are LOWERED to the MUX forms and run on the gate path
  [connect]   "dst/port[a:b]" = "src/port[a:b]"   internal edge (dst input <- src output)
              "dst/port"      = "@name[a:b]"      system input  @name drives dst's input port
              "@name[a:b]"    = "src/port[a:b]"   system output @name reads src's output port
              TOGND = ["@name[a:b]", ...]         declares (and thereby widens) @ports that drive nothing

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


def make_ram_with_mux(in_addr_width, data_width):
    """MUX RAM, the structure of make1bitRAMWithMUX (/root/reference/src/iyokan.hpp:2648-2762): wren is
    demultiplexed over the address bits (msb first: out0 = ANDNOT(in, a), out1 = AND(in, a)) into one
    write-select per word; cell (addr, bit) = DFF fed by MUX(A = itself, B = wdata[bit], S = select[addr]);
    the read side reduces the 2^aw cells of a bit with aw levels of MUX(even, odd, addr[i]), lsb first.
    The select tree is shared by all data bits (the reference rebuilds it per bit; same function)."""
    nl = N.Netlist()
    addr = []
    for i in range(in_addr_width):
        nid = nl.add("INPUT")
        nl.inputs[("addr", i)] = nid
        addr.append(nid)
    wren = nl.add("INPUT")
    nl.inputs[("wren", 0)] = wren
    sel = [wren]
    for a in reversed(addr):
        nxt = []
        for src in sel:
            nxt.append(nl.add("ANDNOT", [src, a]))
            nxt.append(nl.add("AND", [src, a]))
        sel = nxt
    for bit in range(data_width):
        wdata = nl.add("INPUT")
        nl.inputs[("wdata", bit)] = wdata
        work = []
        for a in range(1 << in_addr_width):
            mux = nl.add("MUX", [0, wdata, sel[a]])     # A patched below: the cell itself
            ram = nl.add("DFF", [mux])
            nl.ins[mux][0] = ram
            nl.ram[a * data_width + bit] = ram
            work.append(ram)
        for i in range(in_addr_width):
            work = [nl.add("MUX", [work[j], work[j + 1], addr[i]]) for j in range(0, len(work), 2)]
        out = nl.add("OUTPUT", [work[0]])
        nl.outputs[("rdata", bit)] = out
    nl.validate()
    return nl


class System:
    """Merged netlist + where the blueprint's named things ended up."""

    def __init__(self, nl, at_inputs, at_outputs, rom, ram, at_widths=None, ram_shapes=None, rom_shapes=None):
        self.nl = nl
        self.at_inputs = at_inputs      # (name, bit) -> node id (INPUT)
        self.at_outputs = at_outputs    # (name, bit) -> node id
        self.rom = rom                  # builtin name -> {index: node id}
        self.ram = ram                  # builtin name -> {index: node id}
        self.at_widths = at_widths or {}    # @port name -> declared width (incl. TOGND bits)
        self.ram_shapes = ram_shapes or {}  # builtin name -> (addr width, data width)
        self.rom_shapes = rom_shapes or {}

    def at_width(self, table, name):
        if name in self.at_widths:
            return self.at_widths[name]
        return 1 + max(b for (p, b) in table if p == name)


def load_blueprint(path, mux_ram_dir=None):
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
    ram_shapes, rom_shapes = {}, {}
    for b in bp.get("builtin", []):
        if b["type"] in ("mux-rom", "rom"):
            parts[b["name"]] = make_rom_with_mux(b["in_addr_width"], b["out_rdata_width"])
            rom_shapes[b["name"]] = (b["in_addr_width"], b["out_rdata_width"])
        elif b["type"] in ("mux-ram", "ram"):
            if b["in_wdata_width"] != b["out_rdata_width"]:
                raise ValueError("Invalid RAM size; RAM with different write/read data widths is not implemented")
            # the reference embeds minimised netlists for some shapes (USE_PRECOMPILED_BINARY, iyokan.hpp:2609-2625):
            # use the same file when it is at hand (next to the blueprint, one level up, or in mux_ram_dir)
            fname = "mux-ram-{}-{}-{}.min.json".format(b["in_addr_width"], b["in_wdata_width"], b["out_rdata_width"])
            cands = [os.path.join(d, fname) for d in ([mux_ram_dir] if mux_ram_dir else []) + [base, os.path.dirname(base)]]
            pre = next((c for c in cands if os.path.exists(c)), None)
            if pre:
                parts[b["name"]] = N.load_iyokanl1_json(pre, ram_width=b["in_wdata_width"])
            else:
                parts[b["name"]] = make_ram_with_mux(b["in_addr_width"], b["in_wdata_width"])
            ram_shapes[b["name"]] = (b["in_addr_width"], b["in_wdata_width"])
        else:
            raise ValueError(f"Invalid builtin type: {b['type']}")

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

    at_inputs, at_outputs, at_widths = {}, {}, {}

    def widen(name, bit):
        at_widths[name] = max(at_widths.get(name, 0), bit + 1)

    for dst, src in bp.get("connect", {}).items():
        if dst == "TOGND":
            for port_str in src:
                if not port_str.startswith("@"):
                    raise ValueError(f"Invalid port name for TOGND: {port_str}")
                _, gport, gbits = _parse_ports(port_str)
                for gb in gbits:
                    widen(gport, gb)
            continue
        if not dst or not src or (dst[0] == "@" and src[0] == "@"):
            raise ValueError(f"Invalid connect: {dst} = {src}")
        dnode, dport, dbits = _parse_ports(dst)
        snode, sport, sbits = _parse_ports(src)
        if len(dbits) != len(sbits):
            raise ValueError(f"Invalid connect: {dst} = {src}")
        for db, sb in zip(dbits, sbits):
            if dnode is None:                      # "@out" = "node/port"
                if snode is None:
                    raise ValueError(f"Invalid connect: {dst} = {src}")
                at_outputs.setdefault((dport, db), out_node(snode, sport, sb))
                widen(dport, db)
            elif snode is None:                    # "node/port" = "@in"
                at_inputs.setdefault((sport, sb), in_node(dnode, dport, db))
                widen(sport, sb)
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
    return System(nl, at_inputs, at_outputs, rom, ram, at_widths, ram_shapes, rom_shapes)
