"""Pin oracle/deblock.c: the filter pass against frames produced by an independent model in
the AV1 specification's formulation, the level-search tallies against brute force
(tests/golden/gen_deblock_golden.py, tests/deblock_util.py)."""
import os

import numpy as np
import pytest

import deblock_util as D
import oracle_lib as O

HERE = os.path.dirname(__file__)
# deblock_ref.npz: frames filtered / tallies summed by the reference's own source text
# (gen_deblock_ref.py executes src/deblock.rs); deblock_golden.npz: the independent
# specification-formulation model + brute force.
GS = {n: dict(np.load(os.path.join(HERE, "golden", n + ".npz"))) for n in ("deblock_ref", "deblock_golden")}
G = GS["deblock_golden"]
CASES = [(n, k[:-5]) for n in GS for k in sorted(GS[n]) if k.endswith("_meta")]


def run_filter(oracle, img, pli, xd, yd, blocks, state, cw, ch, bd):
    p = O.plane_from_image(img, bd, 16, 16)
    pc = p.cstruct()
    import ctypes as C
    assert oracle.r1o_deblock_plane(state.ctypes.data, C.byref(pc), pli, xd, yd, blocks.ctypes.data,
                                    blocks.shape[1], blocks.shape[1], blocks.shape[0], cw, ch, bd) == 0
    return p.view().copy()


def run_sse(oracle, rec, src, pli, xd, yd, blocks, cw, ch, bd):
    import ctypes as C
    pr, ps = O.plane_from_image(rec, bd, 16, 16), O.plane_from_image(src, bd, 16, 16)
    cr, cs = pr.cstruct(), ps.cstruct()
    tv, th = np.zeros(65, np.int64), np.zeros(65, np.int64)
    assert oracle.r1o_deblock_sse_plane(C.byref(cr), C.byref(cs), pli, xd, yd, blocks.ctypes.data,
                                        blocks.shape[1], blocks.shape[1], blocks.shape[0], cw, ch, bd,
                                        tv.ctypes.data, th.ctypes.data) == 0
    return tv, th


@pytest.mark.parametrize("fixture,name", CASES)
def test_filter_matches_the_specification_model(oracle, fixture, name):
    G = GS[fixture]
    w, h, cw, ch, bd, xdec, ydec = [int(v) for v in G[name + "_meta"]]
    blocks = np.ascontiguousarray(G[name + "_blocks"])
    state = np.ascontiguousarray(G[name + "_state"])
    for pli in range(3):
        xd, yd = (0, 0) if pli == 0 else (xdec, ydec)
        got = run_filter(oracle, G["%s_p%d_rec" % (name, pli)], pli, xd, yd, blocks, state, cw, ch, bd)
        want = G["%s_p%d_out" % (name, pli)]
        bad = np.argwhere(got != want)
        assert len(bad) == 0, (name, pli, bad[:4])


@pytest.mark.parametrize("fixture,name", CASES)
def test_level_search_tallies_are_the_per_level_sse(oracle, fixture, name):
    """after sse_optimize's prefix sum, tally[L] = SSE of filtering every edge at level L"""
    G = GS[fixture]
    w, h, cw, ch, bd, xdec, ydec = [int(v) for v in G[name + "_meta"]]
    blocks = np.ascontiguousarray(G[name + "_blocks"])
    for pli in range(3):
        xd, yd = (0, 0) if pli == 0 else (xdec, ydec)
        tv, th = run_sse(oracle, G["%s_p%d_rec" % (name, pli)], G["%s_p%d_src" % (name, pli)], pli, xd, yd,
                         blocks, cw, ch, bd)
        assert np.array_equal(np.cumsum(tv)[:64], G["%s_p%d_tv" % (name, pli)]), (name, pli, "v")
        assert np.array_equal(np.cumsum(th)[:64], G["%s_p%d_th" % (name, pli)]), (name, pli, "h")
        lv = np.zeros(2, np.uint8)
        oracle.r1o_deblock_pick_levels(tv.ctypes.data, th.ctypes.data, pli, lv.ctypes.data)
        gv, gh = G["%s_p%d_tv" % (name, pli)], G["%s_p%d_th" % (name, pli)]
        if pli == 0:
            assert (lv[0], lv[1]) == (int(np.argmin(gv)), int(np.argmin(gh)))   # first minimum
        else:
            assert lv[0] == int(np.argmin(gv + gh))
        if name + "_levels" in G:      # what the reference's sse_optimize picked
            want = G[name + "_levels"]
            assert (lv[0], lv[1]) == (want[0], want[1]) if pli == 0 else lv[0] == want[pli + 1]


def test_level_zero_and_skipped_inter_interiors_are_untouched(oracle):
    rng = np.random.default_rng(9)
    blocks = D.random_blocks(rng, 16, 16, 1, 1, p_skip=1.0, p_intra=0.0)
    img = rng.integers(0, 256, (64, 64))
    out = run_filter(oracle, img, 0, 0, 0, blocks, D.make_state([0, 0, 9, 9]), 64, 64, 8)
    assert np.array_equal(out, img)
    # all-skip inter frame: only block edges are filtered, transform edges inside blocks are not
    out = run_filter(oracle, img, 0, 0, 0, blocks, D.make_state([30, 30, 0, 0]), 64, 64, 8)
    n4w = 1 << (blocks["n4_log2"] & 7)
    n4h = 1 << ((blocks["n4_log2"] >> 3) & 7)
    changed = np.argwhere(out != img)
    for y, x in changed[:400]:
        b_w, b_h = int(n4w[y // 4, x // 4]) * 4, int(n4h[y // 4, x // 4]) * 4
        dx, dy = x % b_w, y % b_h
        assert min(dx, b_w - 1 - dx) < 7 or min(dy, b_h - 1 - dy) < 7
