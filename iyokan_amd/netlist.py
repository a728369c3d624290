"""Gate-level netlists: loaders, plaintext evaluation, levelisation.

The callers on either side of the hot path, kept to what BASELINE
configs #3/#4 need:
  * Iyokan-L1 JSON reader  — same semantics as IyokanL1JSONReader (/root/reference/src/iyokan.hpp:2354-2482):
    ports {type,id,portName,portBit,bits}, cells {type,id,input{A,B,S|D},ramAddress,ramBit}.
  * Yosys JSON reader      — same semantics as YosysJSONReader (/root/reference/src/iyokan.hpp:2130-2352):
    one module, `clock` port skipped, `$_AND_`.. `$_MUX_` (A,B,S) `$_NOT_` `$_DFF_P_`, constant
    bits "0"/"1" become CONSTZERO/CONSTONE drivers.
  * plaintext evaluator    — the reference's functional oracle is its plain backend
    (/root/reference/src/iyokan_plain.hpp:105-116); expected bits for encrypted runs come from here.
  * levelise()             — longest-path levels of the per-clock DAG (DFF / INPUT / RAM cells are
    level 0), the unit the frontier executor shards over GPUs.
"""
import json
from collections import defaultdict

import numpy as np

from .params import OPS

BINARY = ("AND", "NAND", "ANDNOT", "OR", "NOR", "ORNOT", "XOR", "XNOR")
_YOSYS = {"$_AND_": "AND", "$_NAND_": "NAND", "$_ANDNOT_": "ANDNOT", "$_OR_": "OR", "$_NOR_": "NOR",
          "$_ORNOT_": "ORNOT", "$_XOR_": "XOR", "$_XNOR_": "XNOR", "$_NOT_": "NOT", "$_MUX_": "MUX",
          "$_DFF_P_": "DFF"}


class Netlist:
    """Flat netlist.  Node kinds: the 12 gate kinds, INPUT (externally driven wire), OUTPUT (wire
    with one driver), DFF (also RAM cells).  `ins[i]` lists driver node ids in (A, B, S) / (D) order."""

    def __init__(self):
        self.kinds, self.ins = [], []
        self.inputs, self.outputs = {}, {}   # (port, bit) -> node id
        self.ram = {}                        # addr * width + bit -> node id (kind DFF)
        self.dff_init = {}                   # node id -> initial bit (default 0)

    def add(self, kind, ins=()):
        self.kinds.append(kind)
        self.ins.append(list(ins))
        return len(self.kinds) - 1

    @property
    def num_nodes(self):
        return len(self.kinds)

    def port_width(self, table, name):
        return 1 + max(b for (p, b) in table if p == name)

    def validate(self):
        need = {"MUX": 3, "NOT": 1, "OUTPUT": 1, "DFF": 1, "INPUT": 0, "CONSTONE": 0, "CONSTZERO": 0}
        for i, k in enumerate(self.kinds):
            want = need.get(k, 2)
            if len(self.ins[i]) != want:
                raise ValueError(f"node {i} ({k}) has {len(self.ins[i])} inputs, needs {want}")

    def roots(self):
        """root[i] = the non-alias node whose value node i carries (OUTPUT-kind nodes are pure aliases of
        their single driver; chains arise when blueprints wire one net's output into another's input)."""
        root = list(range(self.num_nodes))

        def find(i):
            path = []
            while self.kinds[i] == "OUTPUT":
                path.append(i)
                i = self.ins[i][0]
            for j in path:
                root[j] = i
            return i

        for i in range(self.num_nodes):
            find(i)
        return root

    def counts(self):
        c = defaultdict(int)
        for k in self.kinds:
            c[k] += 1
        return dict(c)

    def rotations(self):
        """Blind rotations per clock: binary gates 1, MUX 2."""
        c = self.counts()
        return sum(c.get(k, 0) for k in BINARY) + 2 * c.get("MUX", 0)

    # ---- levelisation -----------------------------------------------------------------
    def levelise(self):
        """levels[k] = node ids whose longest path from a source has k gates on it.  Sources (INPUT,
        DFF, CONST) are level 0 and not listed; OUTPUT wires are aliases of their driver and not listed."""
        self.validate()
        n = self.num_nodes
        level = [-1] * n
        order = self._topo()
        for i in order:
            k = self.kinds[i]
            if k in ("INPUT", "DFF"):
                level[i] = 0
            elif k in ("CONSTONE", "CONSTZERO"):
                level[i] = 1
            elif k == "OUTPUT":
                level[i] = level[self.ins[i][0]]
            else:
                level[i] = 1 + max(level[j] for j in self.ins[i])
        depth = max(level) if n else 0
        levels = [[] for _ in range(depth + 1)]
        for i in order:
            if self.kinds[i] not in ("INPUT", "DFF", "OUTPUT"):
                levels[level[i]].append(i)
        return [lv for lv in levels[1:]]

    def _topo(self):
        n = self.num_nodes
        indeg = [0] * n
        deps = [[] for _ in range(n)]
        for i in range(n):
            if self.kinds[i] == "DFF":
                continue  # a DFF's input belongs to the NEXT clock: it cuts the combinational graph
            for j in self.ins[i]:
                deps[j].append(i)
                indeg[i] += 1
        stack = [i for i in range(n) if indeg[i] == 0]
        order = []
        while stack:
            i = stack.pop()
            order.append(i)
            for d in deps[i]:
                indeg[d] -= 1
                if indeg[d] == 0:
                    stack.append(d)
        if len(order) != n:
            raise ValueError("combinational loop in netlist")
        return order


def load_iyokanl1_json(path, ram_width=None):
    d = json.load(open(path))
    nl = Netlist()
    ids = {}
    for p in d["ports"]:
        kind = "INPUT" if p["type"] == "input" else "OUTPUT"
        ids[p["id"]] = nl.add(kind)
        table = nl.inputs if kind == "INPUT" else nl.outputs
        table[(p["portName"], p["portBit"])] = ids[p["id"]]
    for c in d["cells"]:
        t = c["type"]
        kind = "DFF" if t in ("DFFP", "RAM") else t
        if kind not in OPS and kind != "DFF":
            raise ValueError(f"Invalid JSON of network. Invalid type: {t}")
        ids[c["id"]] = nl.add(kind)
        if t == "RAM":
            if ram_width is None:
                ram_width = 1 + max(x["ramBit"] for x in d["cells"] if x["type"] == "RAM")
            nl.ram[c["ramAddress"] * ram_width + c["ramBit"]] = ids[c["id"]]
    for p in d["ports"]:
        if p["type"] == "output":
            nl.ins[ids[p["id"]]] = [ids[b] for b in p["bits"]]
    for c in d["cells"]:
        i, inp = ids[c["id"]], c["input"]
        if c["type"] in BINARY:
            nl.ins[i] = [ids[inp["A"]], ids[inp["B"]]]
        elif c["type"] in ("DFFP", "RAM"):
            nl.ins[i] = [ids[inp["D"]]]
        elif c["type"] == "NOT":
            nl.ins[i] = [ids[inp["A"]]]
        elif c["type"] == "MUX":
            nl.ins[i] = [ids[inp["A"]], ids[inp["B"]], ids[inp["S"]]]
    nl.validate()
    return nl


def load_yosys_json(path):
    d = json.load(open(path))
    mods = d["modules"]
    if len(mods) != 1:
        raise ValueError(".modules should be an object of size 1")
    mod = next(iter(mods.values()))
    nl = Netlist()
    driver = {}   # yosys bit number -> node id
    consts = {}

    def const(v):
        if v not in consts:
            consts[v] = nl.add("CONSTONE" if v == "1" else "CONSTZERO")
        return consts[v]

    out_ports = []
    for name, port in mod["ports"].items():
        bits = port["bits"]
        if name == "clock" or (name == "reset" and len(bits) == 0):
            continue
        for i, b in enumerate(bits):
            if port["direction"] == "input":
                if isinstance(b, str):
                    raise ValueError("input port bit tied to a constant")
                nid = nl.add("INPUT")
                nl.inputs[(name, i)] = nid
                driver[b] = nid
            elif port["direction"] == "output":
                nid = nl.add("OUTPUT")
                nl.outputs[(name, i)] = nid
                out_ports.append((nid, b))
            else:
                raise ValueError(f"Invalid direction token: {port['direction']}")
    cells = []
    for cname, cell in mod["cells"].items():
        kind = _YOSYS.get(cell["type"])
        if kind is None:
            raise ValueError(f"unsupported cell type {cell['type']} ({cname})")
        nid = nl.add(kind)
        conn = cell["connections"]
        outbit = conn["Q"][0] if kind == "DFF" else conn["Y"][0]
        driver[outbit] = nid
        cells.append((nid, kind, conn))

    def src(b):
        return const(b) if isinstance(b, str) else driver[b]

    for nid, kind, conn in cells:
        if kind == "DFF":
            nl.ins[nid] = [src(conn["D"][0])]
        elif kind == "NOT":
            nl.ins[nid] = [src(conn["A"][0])]
        elif kind == "MUX":
            nl.ins[nid] = [src(conn["A"][0]), src(conn["B"][0]), src(conn["S"][0])]
        else:
            nl.ins[nid] = [src(conn["A"][0]), src(conn["B"][0])]
    for nid, b in out_ports:
        nl.ins[nid] = [src(b)]
    nl.validate()
    return nl


_PLAIN = {
    "AND": lambda a, b: a & b, "NAND": lambda a, b: 1 ^ (a & b), "ANDNOT": lambda a, b: a & (1 ^ b),
    "OR": lambda a, b: a | b, "NOR": lambda a, b: 1 ^ (a | b), "ORNOT": lambda a, b: a | (1 ^ b),
    "XOR": lambda a, b: a ^ b, "XNOR": lambda a, b: 1 ^ a ^ b,
}


class PlainSimulator:
    """Cycle-accurate plaintext run of one netlist with the reference's clocking protocol
    (/root/reference/src/iyokan_plain.cpp:452-548): optional reset cycle, then per cycle
    tick -> set inputs -> evaluate."""

    def __init__(self, nl):
        self.nl = nl
        self.val = np.zeros(nl.num_nodes, dtype=np.uint8)
        self.levels = nl.levelise()
        self.root = nl.roots()
        self.rins = [[self.root[j] for j in ins] for ins in nl.ins]
        self.dffs = [i for i, k in enumerate(nl.kinds) if k == "DFF"]
        self._prog = None
        for i, v in nl.dff_init.items():
            self.val[i] = v

    def set_input(self, port, bit, v):
        self.val[self.nl.inputs[(port, bit)]] = v

    def set_port(self, port, value):
        for (p, b), nid in self.nl.inputs.items():
            if p == port:
                self.val[nid] = (value >> b) & 1

    def get_output(self, port, bit):
        return int(self.val[self.root[self.nl.outputs[(port, bit)]]])

    def node_value(self, nid):
        return int(self.val[self.root[nid]])

    def get_port(self, port):
        return sum(self.get_output(p, b) << b for (p, b) in self.nl.outputs if p == port)

    def _compile(self):
        """Per level, per gate kind: (node ids, driver ids) as index arrays, so a clock of a 30 k-gate
        system is a few hundred numpy gathers instead of a Python loop over gates."""
        prog = []
        for lv in self.levels:
            by_kind = defaultdict(list)
            for i in lv:
                by_kind[self.nl.kinds[i]].append(i)
            steps = []
            for k, ids in by_kind.items():
                idx = np.asarray(ids, dtype=np.int64)
                ins = [np.asarray([self.rins[i][j] for i in ids], dtype=np.int64) for j in range(len(self.rins[ids[0]]))]
                steps.append((k, idx, ins))
            prog.append(steps)
        self._prog = prog
        self._dff_idx = np.asarray(self.dffs, dtype=np.int64)
        self._dff_src = np.asarray([self.rins[i][0] for i in self.dffs], dtype=np.int64)

    def evaluate(self):
        if self._prog is None:
            self._compile()
        v = self.val
        for steps in self._prog:
            for k, idx, ins in steps:   # gates of one level are independent: any order within it
                if k in _PLAIN:
                    v[idx] = _PLAIN[k](v[ins[0]], v[ins[1]])
                elif k == "MUX":
                    v[idx] = np.where(v[ins[2]] != 0, v[ins[1]], v[ins[0]])
                elif k == "NOT":
                    v[idx] = 1 ^ v[ins[0]]
                elif k == "CONSTONE":
                    v[idx] = 1
                elif k == "CONSTZERO":
                    v[idx] = 0

    def tick(self):
        if self._prog is None:
            self._compile()
        self.val[self._dff_idx] = self.val[self._dff_src]   # two-phase: the gather samples before the scatter commits

    def ram_image(self, nbits):
        return [int(self.val[self.nl.ram[i]]) for i in range(nbits)]


def bits_from_bytes(data, size):
    """TOML packet bit streams are little-endian within each byte (/root/reference/src/packet.hpp)."""
    return [(data[i // 8] >> (i % 8)) & 1 if i // 8 < len(data) else 0 for i in range(size)]


def bytes_from_bits(bits):
    out = [0] * ((len(bits) + 7) // 8)
    for i, b in enumerate(bits):
        out[i // 8] |= int(b) << (i % 8)
    return out
