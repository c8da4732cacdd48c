"""Pin oracle/mc.c.

mc_ref.npz    put_8tap / prep_8tap / mc_avg outputs computed by the reference's own
              source text (src/mc.rs:110-479 transpiled and executed by
              tests/golden/gen_mc_ref.py) -- the reference-derived pin.
mc_golden.npz the reference's tap table + an independent NumPy model
              (tests/golden/gen_mc_golden.py) -- kept as a second opinion.
"""
import os

import numpy as np
import pytest

import oracle_lib as O

GOLD = {n: np.load(os.path.join(os.path.dirname(__file__), "golden", n + ".npz"))
        for n in ("mc_ref", "mc_golden")}
G = GOLD["mc_ref"]


@pytest.fixture(params=["mc_ref", "mc_golden"])
def gold(request):
    return GOLD[request.param]


def run_case(oracle, k, G):
    bd, w, h, cf, rf, mx, my, _ = map(int, k.split("_"))
    win = np.ascontiguousarray(G["win_" + k])
    hbd = int(bd > 8)
    ws = win.shape[1]
    src = win.ctypes.data + (3 * ws + 3) * win.itemsize
    put = np.zeros((h, w), win.dtype)
    prep = np.zeros((h, w), np.int16)
    oracle.r1o_put_8tap(O.ptr(put), w, src, ws, w, h, cf, rf, mx, my, bd, hbd)
    oracle.r1o_prep_8tap(O.ptr(prep), src, ws, w, h, cf, rf, mx, my, bd, hbd)
    return put, prep


def test_put_prep_match_golden(oracle, gold):
    G = gold
    for k in G["cases"]:
        put, prep = run_case(oracle, k, G)
        assert np.array_equal(put, G["put_" + k]), k
        assert np.array_equal(prep, G["prep_" + k]), k


def test_avg_matches_golden(oracle, gold):
    G = gold
    for bd in (8, 10, 12):
        t1, t2 = G["avg_t1_%d" % bd], G["avg_t2_%d" % bd]
        want = G["avg_out_%d" % bd]
        out = np.zeros_like(want)
        oracle.r1o_mc_avg(O.ptr(out), 16, O.ptr(np.ascontiguousarray(t1)),
                          O.ptr(np.ascontiguousarray(t2)), 16, 16, bd, int(bd > 8))
        assert np.array_equal(out, want)


def test_filter_table_properties(gold):
    T = gold["filters"].astype(int)
    assert np.array_equal(T, GOLD["mc_ref"]["filters"].astype(int))
    assert (T.sum(axis=2) == 128).all()           # unity DC gain
    for s in range(6):                            # phase f and 16-f are mirror images
        for f in range(1, 16):
            assert list(T[s][16 - f]) == list(T[s][f][::-1]), (s, f)
    assert (np.abs(T[:, 1:, :]) < 128).all()      # fits int8 for every non-zero phase


def test_flat_plane_is_preserved(oracle):
    for bd in (8, 10, 12):
        dt = np.uint8 if bd == 8 else np.uint16
        for val in (0, 1, (1 << bd) - 1, 77):
            win = np.full((23, 23), val, dt)
            src = win.ctypes.data + (3 * 23 + 3) * win.itemsize
            for cf, rf in ((0, 0), (0, 5), (9, 0), (3, 12)):
                for m in range(4):
                    put = np.zeros((16, 16), dt)
                    oracle.r1o_put_8tap(O.ptr(put), 16, src, 23, 16, 16, cf, rf, m, m, bd,
                                        int(bd > 8))
                    assert (put == val).all(), (bd, val, cf, rf, m)
