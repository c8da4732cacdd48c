"""Lookahead cost maps of the oracle (src/api/lookahead.rs:30-268): definition
checks on small frames (the reference holds no vectors for these; they are
compositions of get_intra_edges / DC_PRED / get_satd, each pinned elsewhere)."""
import ctypes as C

import numpy as np

import oracle_lib as O


def satd8(d):
    h = np.array([[1]])
    for _ in range(3):
        h = np.block([[h, h], [h, -h]])
    return (np.abs(h @ d @ h.T).sum() + 4) >> 3


def test_intra_costs_definition(oracle):
    rng = np.random.default_rng(0)
    p = O.HostPlane(40, 24, 8, rng=rng)
    img = p.view().astype(np.int64)
    got = np.zeros(3 * 5, np.uint32)
    ps = p.cstruct()
    oracle.r1o_estimate_intra_costs(C.byref(ps), 8, O.ptr(got))
    for by in range(3):
        for bx in range(5):
            blk = img[by * 8:by * 8 + 8, bx * 8:bx * 8 + 8]
            if bx == 0 and by == 0:
                dc = 128
            elif by == 0:
                dc = (img[0:8, bx * 8 - 1].sum() + 4) // 8
            elif bx == 0:
                dc = (img[by * 8 - 1, 0:8].sum() + 4) // 8
            else:
                dc = (img[by * 8:by * 8 + 8, bx * 8 - 1].sum() + img[by * 8 - 1, bx * 8:bx * 8 + 8].sum() + 8) // 16
            assert got[by * 5 + bx] == satd8(blk - dc), (bx, by)


def test_inter_costs_and_block_difference(oracle):
    rng = np.random.default_rng(1)
    a = O.HostPlane(32, 16, 10, rng=rng)
    b = O.HostPlane(32, 16, 10, rng=rng)
    pa, pb = a.cstruct(), b.cstruct()
    mvs = np.array([[[-9, 17], [8, -8], [0, 0], [-1, 1]], [[64, -64], [7, 7], [-7, -7], [100, 3]]],
                   np.int16)
    got = np.zeros(8, np.uint32)
    oracle.r1o_estimate_inter_costs(C.byref(pa), C.byref(pb), O.ptr(mvs), O.ptr(got))
    A = a.data.astype(np.int64)
    B = b.data.astype(np.int64)
    for by in range(2):
        for bx in range(4):
            rx = int((bx * 64 + int(mvs[by, bx, 1])) / 8)        # truncation toward zero
            ry = int((by * 64 + int(mvs[by, bx, 0])) / 8)
            o = A[a.yorigin + by * 8:a.yorigin + by * 8 + 8, a.xorigin + bx * 8:a.xorigin + bx * 8 + 8]
            r = B[b.yorigin + ry:b.yorigin + ry + 8, b.xorigin + rx:b.xorigin + rx + 8]
            assert got[by * 4 + bx] == satd8(o - r), (bx, by)
    tot = oracle.r1o_importance_block_difference(C.byref(pa), C.byref(pb))
    want = 0
    for by in range(2):
        for bx in range(4):
            so = a.view()[by * 8:by * 8 + 8, bx * 8:bx * 8 + 8].astype(np.int64).sum()
            sr = b.view()[by * 8:by * 8 + 8, bx * 8:bx * 8 + 8].astype(np.int64).sum()
            want += abs((so + 32) // 64 - (sr + 32) // 64)
    assert tot == want


def test_activity_scales_definition(oracle):
    """variance_8x8 = sum(s^2) - round(sum(s)^2 / 64); scale = ssim_boost(var, var): flat blocks
    get the largest scale, and the float formula of src/activity.rs:194-274 is met within 5 %."""
    import ctypes as C
    rng = np.random.default_rng(12)
    for bd in (8, 10, 12):
        hp = O.HostPlane(100, 52, bd, rng=rng)          # not multiples of 8: the padding is read
        hp.view()[:8, :8] = 77                          # one flat block
        pc = hp.cstruct()
        wb, hb = 13, 7
        var, sc = np.zeros((hb, wb), np.uint32), np.zeros((hb, wb), np.uint32)
        oracle.r1o_activity_scales(C.byref(pc), O.ptr(var), O.ptr(sc))
        a = hp.data[hp.yorigin:hp.yorigin + hb * 8, hp.xorigin:hp.xorigin + wb * 8].astype(np.int64)
        blk = a.reshape(hb, 8, wb, 8)
        s1, s2 = blk.sum((1, 3)), (blk * blk).sum((1, 3))
        assert np.array_equal(var, np.minimum(s2 - ((s1 * s1 + 32) >> 6), 0xFFFFFFFF))
        assert var[0, 0] == 0 and sc[0, 0] == sc.max()
        v = (var >> (2 * (bd - 8))).astype(np.float64)
        # ssim_boost = (C1 / C3) * (svar + dvar + C2) / sqrt(C1^2 + svar * dvar), Q14
        want = (1 << 14) * (3355.0 / 12338.0) * (2 * v + 16128) / np.sqrt(3355.0 ** 2 + v * v)
        rel = np.abs(sc.astype(np.float64) - want) / want
        assert rel.max() < 0.05, rel.max()
