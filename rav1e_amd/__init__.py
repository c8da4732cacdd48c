"""rav1e_amd -- MI355X (gfx950) block-kernel backend for rav1e's RDO inner loop.

The product is librav1e_hip.so (hand-written HIP, C ABI in include/rav1e_amd.h).
This package is the thin host-side mirror used by tests and bench.py: enums with
the reference's integer values, the v_frame Plane layout, and ctypes bindings
that hand raw device pointers / stream handles to the C ABI.  torch is only the
allocator and stream provider.  There is NO CPU fallback: importing
`rav1e_amd.api` without a built library raises.
"""
from .types import BlockSize, TxSize, TxType, FilterMode, DistKind, TX_DIMS  # noqa: F401

__version__ = "0.1.0"
