"""oracle/cdef.c::r1o_cdef_strength_search against tests/golden/cdef_search_ref.npz: the CDEF
strength search of rdo_loop_decision (src/rdo.rs:2104-2560, CDEF leg) whose arithmetic was
produced by EXECUTING the reference's cdef.rs / rdo.rs / dist.rs text (gen_cdef_search_ref.py)."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
REF = dict(np.load(os.path.join(GOLDEN, "cdef_search_ref.npz")))
# + the CDEF-only cases of loop_decision_ref.npz (ldc*) and the CDEF leg's first pass of its both-filters case (ldb*):
# the same arrays, made by EXECUTING rdo_loop_decision itself (gen_loop_decision_ref.py)
LOOP = np.load(os.path.join(GOLDEN, "loop_decision_ref.npz"))
REF.update({k: LOOP[k] for k in LOOP.files if k.startswith(("ldc", "ldb"))})
CASES = sorted(k[:-5] for k in REF if k.endswith("_meta"))


@pytest.mark.parametrize("name", CASES)
def test_strength_search_equals_the_executed_reference(oracle, name):
    got_err, got_best, want_err, want_best = O.cdef_search_oracle(oracle, REF, name)
    assert np.array_equal(got_best, want_best), (name, got_best, want_best)
    bad = np.argwhere(got_err != want_err)
    assert len(bad) == 0, (name, bad[:4], got_err[tuple(bad[0])], want_err[tuple(bad[0])])


def test_rejects_bad_parameters(oracle):
    prm = O.CdefSearchParams()
    prm.n_idx, prm.planes, prm.area_sb_w, prm.area_sb_h = 9, 3, 1, 1
    assert oracle.r1o_cdef_strength_search(None, None, None, 0, 0, 0, None, 0, C.byref(prm), None, None) != 0
