"""TFHE parameter sets — ctypes mirror of include/iyokan_hip_params.h (the single source of truth).

The reference selects its set at compile time (`IYOKAN_80BIT_SECURITY`,
/root/reference/CMakeLists.txt:3,29-31); here it is a run-time value handed to iyk_hip_init.
"""
import ctypes


class IykParams(ctypes.Structure):
    _fields_ = [
        ("n", ctypes.c_uint32),
        ("N", ctypes.c_uint32),
        ("k", ctypes.c_uint32),
        ("l", ctypes.c_uint32),
        ("Bgbit", ctypes.c_uint32),
        ("t", ctypes.c_uint32),
        ("basebit", ctypes.c_uint32),
        ("mu", ctypes.c_uint32),
        ("alpha0", ctypes.c_double),
        ("alpha1", ctypes.c_double),
    ]

    # derived sizes (words are uint32) -- same formulas as the header's inline helpers
    @property
    def tlwe0_words(self):
        return self.n + 1

    @property
    def tlwe1_words(self):
        return self.k * self.N + 1

    @property
    def trgsw_rows(self):
        return (self.k + 1) * self.l

    @property
    def bk_words(self):
        return self.n * self.trgsw_rows * (self.k + 1) * self.N

    @property
    def ksk_words(self):
        return self.k * self.N * self.t * ((1 << self.basebit) - 1) * (self.n + 1)

    def gate_algorithmic_bytes(self, rotations=1, inputs=2):
        """B_gate = R*n*(k+1)l*(k+1)*N*8 + N*k*t*(n+1)*4 + (inputs+1)*(n+1)*4."""
        bk = rotations * self.n * self.trgsw_rows * (self.k + 1) * self.N * 8
        ks = self.N * self.k * self.t * (self.n + 1) * 4
        io = (inputs + 1) * (self.n + 1) * 4
        return bk + ks + io

    def as_dict(self):
        return {f: getattr(self, f) for f, _ in self._fields_}


def params_128bit():
    return IykParams(636, 1024, 1, 3, 6, 7, 2, 1 << 29, 0.000092511997467675, 0.0000000342338787018369)


def params_80bit():
    return IykParams(500, 1024, 1, 2, 10, 8, 2, 1 << 29, 2.44e-5, 3.73e-9)


def params_by_name(name):
    name = str(name).lower()
    if name in ("128", "128bit", "128-bit"):
        return params_128bit()
    if name in ("80", "80bit", "80-bit"):
        return params_80bit()
    raise ValueError(f"unknown parameter set {name!r}")


# gate op codes: must match iyk_gate_op in include/iyokan_hip.h
OPS = {
    "AND": 0, "NAND": 1, "ANDNOT": 2, "OR": 3, "NOR": 4, "ORNOT": 5, "XOR": 6, "XNOR": 7,
    "MUX": 8, "NOT": 9, "CONSTONE": 10, "CONSTZERO": 11, "COPY": 12,
}
OP_NAMES = {v: k for k, v in OPS.items()}
# blind rotations per gate kind (MUX = 2, binary gates = 1, the rest 0)
OP_ROTATIONS = {name: (2 if name == "MUX" else 1 if code < 8 else 0) for name, code in OPS.items()}
# plaintext semantics (/root/reference/src/iyokan_plain.hpp:105-116); MUX(a, b, s) = s ? b : a
PLAIN = {
    "AND": lambda a, b: a & b, "NAND": lambda a, b: 1 - (a & b), "ANDNOT": lambda a, b: a & (1 - b),
    "OR": lambda a, b: a | b, "NOR": lambda a, b: 1 - (a | b), "ORNOT": lambda a, b: a | (1 - b),
    "XOR": lambda a, b: a ^ b, "XNOR": lambda a, b: 1 - (a ^ b),
    "MUX": lambda a, b, s: b if s else a, "NOT": lambda a: 1 - a,
    "CONSTONE": lambda: 1, "CONSTZERO": lambda: 0, "COPY": lambda a: a,
}
