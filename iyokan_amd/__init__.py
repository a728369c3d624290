"""iyokan_amd — MI355X-native TFHE gate-bootstrapping backend for Iyokan (hot path only).

See DESIGN.md.  The compute path is libiyokan_hip.so (hand-written HIP for gfx950) behind the
C ABI in include/iyokan_hip.h; this package is the thin Python host side used by tests and
bench.py.  There is no CPU fallback: importing `iyokan_amd.hip` without the built library, or
calling it without a GPU, raises.
"""
from .params import IykParams, OPS, OP_NAMES, params_128bit, params_80bit, params_by_name  # noqa: F401
