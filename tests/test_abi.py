"""The C-ABI library loads (no GPU needed) and exports every symbol that
include/rav1e_amd.h declares; struct layouts match the header."""
import ctypes as C
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "rav1e_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b((?:r1|rav1e)_\w+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from rav1e_amd import _lib
    L = _lib.load()
    names = header_functions()
    assert len(names) >= 15
    for n in names:
        assert hasattr(L, n), "missing export: " + n
        assert n in _lib.SYMBOLS, "python binding table lacks " + n
    assert L.r1_abi_version() == 6


def test_struct_layouts():
    from rav1e_amd import _lib, api
    assert C.sizeof(_lib.R1Plane) == 40
    assert api.DIST_CAND.itemsize == 8
    assert api.MC_CAND.itemsize == 8
    assert api.RDO_CAND.itemsize == 16


def test_no_oracle_in_product():
    """The product tree must not reference the oracle."""
    bad = []
    for d, _, files in os.walk(os.path.join(ROOT, "rav1e_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                txt = open(os.path.join(d, f)).read()
                if re.search(r"(#include|import|from)\s+[\"<]?.*oracle", txt):
                    bad.append(f)
    assert not bad, bad


def test_enum_values_match_reference_order():
    from rav1e_amd.types import BlockSize, TxSize, TxType
    assert BlockSize.BLOCK_64X64 == 12 and BlockSize.BLOCK_64X16 == 21
    assert TxSize.TX_64X64 == 4 and TxSize.TX_64X16 == 18 and TxSize.TX_4X16 == 13
    assert TxType.IDTX == 9 and TxType.WHT_WHT == 16
    assert TxSize.TX_16X64.dims == (16, 64)
