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
    assert L.r1_abi_version() == 7


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


def test_numpy_candidate_layouts_equal_the_header_compiled_as_c(tmp_path):
    """include/rav1e_amd.h compiled by gcc AS C (the boundary is a C ABI): sizeof and the offset of every field of the
    candidate / unit structs the Python host mirrors as numpy dtypes -- a renamed or re-ordered field on either side
    fails here, without a GPU"""
    import subprocess
    from rav1e_amd import api
    pairs = {"R1DistCand": api.DIST_CAND, "R1McCand": api.MC_CAND, "R1RdoCand": api.RDO_CAND,
             "R1IntraEdgeCand": api.INTRA_EDGE_CAND, "R1IntraCand": api.INTRA_CAND, "R1CflAcCand": api.CFL_AC_CAND,
             "R1CflAlphaCand": api.CFL_ALPHA_CAND, "R1CdefBlockCand": api.CDEF_BLOCK_CAND,
             "R1MeBlockCand": api.ME_BLOCK_CAND, "R1MeResult": api.ME_RESULT, "R1SgrSolveUnit": api.SGR_SOLVE_UNIT,
             "R1TrialUnit": api.TRIAL_UNIT}
    from rav1e_amd import _lib
    structs = {n: getattr(_lib, n) for n in dir(_lib)          # the ctypes mirrors of the parameter structs
               if n.startswith("R1") and isinstance(getattr(_lib, n), type) and issubclass(getattr(_lib, n), C.Structure)}
    assert len(structs) >= 7
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "rav1e_amd.h"', 'int main(void) {']
    for cname, dt in pairs.items():
        lines.append('  printf("%s size %%zu\\n", sizeof(%s));' % (cname, cname))
        for f in dt.names:
            lines.append('  printf("%s %s %%zu\\n", offsetof(%s, %s));' % (cname, f, cname, f))
    for cname, st in structs.items():
        lines.append('  printf("%s size %%zu\\n", sizeof(%s));' % (cname, cname))
        for f, *_ in st._fields_:       # (a trailing underscore on the Python side: `lambda` is a keyword there)
            lines.append('  printf("%s %s %%zu\\n", offsetof(%s, %s));' % (cname, f, cname, f.rstrip("_")))
    lines += ['  return 0;', '}']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c11", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split("\n")
    got = {tuple(l.split()[:2]): int(l.split()[2]) for l in out if l}
    for cname, dt in pairs.items():
        assert got[(cname, "size")] == dt.itemsize, (cname, got[(cname, "size")], dt.itemsize)
        for f in dt.names:
            assert got[(cname, f)] == dt.fields[f][1], (cname, f, got[(cname, f)], dt.fields[f][1])
    for cname, st in structs.items():
        assert got[(cname, "size")] == C.sizeof(st), (cname, got[(cname, "size")], C.sizeof(st))
        for f, *_ in st._fields_:
            assert got[(cname, f)] == getattr(st, f).offset, (cname, f, got[(cname, f)], getattr(st, f).offset)
