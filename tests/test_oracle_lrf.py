"""Pin oracle/lrf.c (self-guided loop restoration as lrf_filter_frame applies it) against the
frames of the independent model in tests/lrf_util.py (tests/golden/gen_lrf_golden.py)."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib as O

G = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "lrf_golden.npz")))
CASES = sorted(k[:-5] for k in G if k.endswith("_meta"))
# the same frames filtered by the reference's own lrf_filter_frame text (tests/golden/gen_lrf_ref.py)
# + sgrproj_solve results of the reference's text
REF = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "lrf_ref.npz")))
IN_FRAME = 3          # R1O_SGR_EDGE_LEFT | R1O_SGR_EDGE_ABOVE
REF_CASES = sorted(k[:-5] for k in REF if k.endswith("_meta"))


def run_oracle(oracle, cdef, debl, ydec, fh, us, sh, units, bd):
    h, w = cdef.shape
    pc, pd = O.plane_from_image(cdef, bd, 16, 16), O.plane_from_image(debl, bd, 16, 16)
    po = O.plane_from_image(cdef, bd, 16, 16)         # out starts as a copy of the CDEF output
    cc, cd, co = pc.cstruct(), pd.cstruct(), po.cstruct()
    u = np.ascontiguousarray(units)
    assert oracle.r1o_lrf_filter_plane(C.byref(cc), C.byref(cd), C.byref(co), ydec, w, h, fh, us,
                                       u.shape[1], u.shape[0], sh, u.ctypes.data, bd) == 0
    return po.view().copy()


@pytest.mark.parametrize("name", CASES)
def test_sgrproj_frames_match_the_independent_model(oracle, name):
    w, h, ydec, fh, us, sh, bd = [int(v) for v in G[name + "_meta"]]
    got = run_oracle(oracle, G[name + "_cdef"], G[name + "_debl"], ydec, fh, us, sh, G[name + "_units"], bd)
    bad = np.argwhere(got != G[name + "_out"])
    assert len(bad) == 0, (name, bad[:5])


def test_units_without_a_filter_and_flat_input_are_untouched(oracle):
    img = np.full((80, 96), 77)
    units = np.zeros((2, 2), O.LRF_UNIT)
    units["filter"] = [[3, 0], [0, 3]]
    units["set"] = 3
    units["xqd"] = (-20, 40)
    out = run_oracle(oracle, img, img, 0, 80, 64, 64, units, 8)
    assert np.array_equal(out, img)                    # a flat frame is a fixed point of the filter
    rng = np.random.default_rng(0)
    yy, xx = np.mgrid[0:80, 0:96]
    img = 60 + xx + yy // 2 + rng.integers(-3, 4, (80, 96))   # texture the filter acts on
    out = run_oracle(oracle, img, img, 0, 80, 64, 64, units, 8)
    assert np.array_equal(out[:56, 64:], img[:56, 64:])   # unit (0, 1): RESTORE_NONE, stripe 0
    assert (out[:56, :64] != img[:56, :64]).any()


@pytest.mark.parametrize("bd", [8, 10])
def test_sgrproj_solve_matches_the_independent_model(oracle, bd):
    """xqd of sgrproj_solve for every parameter set on small units (interior, frame corner,
    odd height) against tests/lrf_util.py::solve_unit (exact moments, emulated fma)."""
    import lrf_util as L
    rng = np.random.default_rng(31 + bd)
    h, w = 72, 88
    yy, xx = np.mgrid[0:h, 0:w]
    src = np.clip((np.sin(xx / 6.0) + np.cos(yy / 5.0) + 2) / 4 * ((1 << bd) - 1), 0, (1 << bd) - 1).astype(np.int64)
    cdef = np.clip(src + rng.integers(-9, 10, (h, w)) * (1 << (bd - 8)), 0, (1 << bd) - 1)
    pc, ps = O.plane_from_image(cdef, bd, 16, 16), O.plane_from_image(src, bd, 16, 16)
    cc, cs = pc.cstruct(), ps.cstruct()
    for (x0, y0, uw, uh) in ((0, 0, 24, 20), (32, 16, 28, 17), (64, 40, 24, 32)):
        for set_ in range(16):
            got = np.zeros(2, np.int8)
            oracle.r1o_sgrproj_solve(C.byref(cc), C.byref(cs), x0, y0, uw, uh, set_, IN_FRAME, bd, got.ctypes.data)
            want = L.solve_unit(cdef, src, x0, y0, uw, uh, set_, bd)
            assert tuple(int(v) for v in got) == want, (bd, x0, y0, set_, got, want)


@pytest.mark.parametrize("bd", [8, 10])
def test_edge_flags_equal_cropping_the_picture(oracle, bd):
    """R1O_SGR_EDGE_*: a unit solved with a flag CLEAR must not depend on anything beyond that side -- it equals the
    unit solved in a picture cropped there.  The independent model (tests/lrf_util.py: a unit inside a whole picture,
    i.e. every neighbour it has is visible) run on the four crops gives the four flag combinations."""
    import lrf_util as L
    rng = np.random.default_rng(71 + bd)
    h, w = 80, 96
    yy, xx = np.mgrid[0:h, 0:w]
    src = np.clip((np.sin(xx / 5.0) + np.cos(yy / 4.0) + 2) / 4 * ((1 << bd) - 1), 0, (1 << bd) - 1).astype(np.int64)
    cdef = np.clip(src + rng.integers(-7, 8, (h, w)) * (1 << (bd - 8)), 0, (1 << bd) - 1)
    x0, y0, uw, uh = 40, 24, 32, 24
    cdef[:, x0 - 4:x0] = (1 << bd) - 1          # what lies just outside the unit is nothing like its inside:
    cdef[y0 - 2:y0, :] = 0                      # seeing it or not must show in the weights
    pc, ps = O.plane_from_image(cdef, bd, 16, 16), O.plane_from_image(src, bd, 16, 16)
    cc, cs = pc.cstruct(), ps.cstruct()
    seen = set()
    for set_ in (0, 3, 7, 9, 10, 12, 14, 15):
        for edges in range(4):
            got = np.zeros(2, np.int8)
            oracle.r1o_sgrproj_solve(C.byref(cc), C.byref(cs), x0, y0, uw, uh, set_, edges, bd, got.ctypes.data)
            cx, cy = (x0 - 4 if edges & 1 else x0), (y0 - 2 if edges & 2 else y0)     # what is left of the crop
            want = L.solve_unit(cdef[cy:, cx:], src[cy:, cx:], x0 - cx, y0 - cy, uw, uh, set_, bd)
            assert tuple(int(v) for v in got) == want, (bd, set_, edges, got, want)
            seen.add((set_, edges, want))
    # the flags matter: for some set the four neighbourhoods give different weights
    assert any(len({w_ for (s_, e_, w_) in seen if s_ == set_}) > 1 for set_ in (0, 3, 7, 9, 10, 12, 14, 15))


@pytest.mark.parametrize("name", REF_CASES)
def test_sgrproj_frames_equal_the_executed_reference(oracle, name):
    """oracle/lrf.c against RestorationState::lrf_filter_frame of the reference's own text"""
    w, h, ydec, fh, us, sh, bd = [int(v) for v in REF[name + "_meta"]]
    got = run_oracle(oracle, REF[name + "_cdef"], REF[name + "_debl"], ydec, fh, us, sh, REF[name + "_units"], bd)
    bad = np.argwhere(got != REF[name + "_out"])
    assert len(bad) == 0, (name, bad[:5])


@pytest.mark.parametrize("bd", [8, 10])
def test_sgrproj_solve_equals_the_executed_reference(oracle, bd):
    """xqd of oracle/lrf.c::r1o_sgrproj_solve against sgrproj_solve of the reference's text
    (setup_integral_image + sgrproj_solve as src/rdo.rs:2651-2684 calls them)"""
    cdef, src = REF["solve%d_cdef" % bd].astype(np.int64), REF["solve%d_src" % bd].astype(np.int64)
    pc, ps = O.plane_from_image(cdef, bd, 16, 16), O.plane_from_image(src, bd, 16, 16)
    cc, cs = pc.cstruct(), ps.cstruct()
    for (x0, y0, uw, uh, set_, q0, q1) in REF["solve%d_cases" % bd].tolist():
        got = np.zeros(2, np.int8)
        oracle.r1o_sgrproj_solve(C.byref(cc), C.byref(cs), x0, y0, uw, uh, set_, IN_FRAME, bd, got.ctypes.data)
        assert (int(got[0]), int(got[1])) == (q0, q1), (bd, x0, y0, uw, uh, set_, got, (q0, q1))


# ---- the restoration leg of rdo_loop_decision (everything but the rate) -------------------------
# Which pixels left of / above a unit the filter sees (oracle/lrf.c, setup_integral_image): the vectors made by calling
# the reference's functions on slices of whole-frame planes (lrf_ref.npz's solve cases, lrf_search_ref.npz) are the
# "everything around the unit exists" case -- LEFT | ABOVE, the left flag void at x = 0.  The vectors made by
# executing rdo_loop_decision (loop_decision_ref.npz) carry the flags of each unit as that function set things up.
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SEARCH = dict(np.load(os.path.join(GOLDEN, "lrf_search_ref.npz")))
# + the restoration-only cases of loop_decision_ref.npz: the same rows, but made by EXECUTING rdo_loop_decision itself
# (gen_loop_decision_ref.py) instead of a hand-stated loop around its callees
LOOP = np.load(os.path.join(GOLDEN, "loop_decision_ref.npz"))
SEARCH.update({k: LOOP[k] for k in LOOP.files if k.startswith("ldl")})


def search_cases():
    return sorted(k[:-5] for k in SEARCH if k.endswith("_meta"))


@pytest.mark.parametrize("case", search_cases())
def test_lrf_search_units_equal_the_executed_reference(oracle, case):
    """oracle/lrf.c::r1o_lrf_search_unit against setup_integral_image + sgrproj_solve +
    sgrproj_stripe_filter + rdo_loop_plane_error of the reference's own text, unit by unit and set by
    set (and the "no filter option"), luma and both chroma planes (lrf_search_ref.npz)"""
    W, H, xdec, ydec, bd, lru_sb = [int(v) for v in SEARCH[case + "_meta"]]
    assert len(SEARCH[case + "_rows"]) == len(SEARCH[case + "_err"]) > 20
    pin = [O.plane_from_image(SEARCH[case + "_in%d" % p].astype(np.int64), bd, 16, 16) for p in range(3)]
    psrc = [O.plane_from_image(SEARCH[case + "_src%d" % p].astype(np.int64), bd, 16, 16) for p in range(3)]
    scales = np.ascontiguousarray(SEARCH[case + "_scales"])
    dscale = SEARCH[case + "_dscale"]
    rows, errs = SEARCH[case + "_rows"], SEARCH[case + "_err"]
    edges = SEARCH[case + "_edges"] if case + "_edges" in SEARCH else np.full(len(rows), IN_FRAME)
    for (pli, x, y, w, h, set_, q0, q1), want, edge in zip(rows, errs, edges):
        xd, yd = (0, 0) if pli == 0 else (xdec, ydec)
        ci, cs = pin[pli].cstruct(), psrc[pli].cstruct()
        xqd, err = np.zeros(2, np.int8), np.zeros(1, np.uint64)
        rc = oracle.r1o_lrf_search_unit(C.byref(ci), C.byref(cs), int(x), int(y), int(w), int(h), int(set_), int(edge),
                                        int(pli != 0), xd, yd, scales.ctypes.data, scales.shape[1],
                                        int(dscale[pli]), bd, xqd.ctypes.data, err.ctypes.data)
        assert rc == 0
        assert (int(xqd[0]), int(xqd[1])) == (int(q0), int(q1)), (case, pli, x, y, set_)
        assert int(err[0]) == int(want), (case, pli, x, y, w, h, set_, int(err[0]), int(want))


def test_lrf_search_rejects_partial_blocks(oracle):
    a = np.full((64, 64), 100, np.int64)
    p = O.plane_from_image(a, 8, 16, 16)
    c = p.cstruct()
    xqd, err = np.zeros(2, np.int8), np.zeros(1, np.uint64)
    args = lambda w, h, s: (C.byref(c), C.byref(c), 0, 0, w, h, s, 0, 0, 0, 0, None, 0, 1 << 14, 8, xqd.ctypes.data,
                            err.ctypes.data)
    assert oracle.r1o_lrf_search_unit(*args(60, 64, 3)) == -1      # a width that cuts an 8x8 block
    assert oracle.r1o_lrf_search_unit(*args(64, 64, 16)) == -1     # no such parameter set
    assert oracle.r1o_lrf_search_unit(*args(64, 64, 255)) == 0 and int(err[0]) == 0   # identical planes
