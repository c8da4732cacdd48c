"""Lookahead cost maps of the oracle (src/api/lookahead.rs:30-268): definition checks on small
frames.  The pin is tests/test_oracle_lookahead_ref.py (vectors from executing the reference's
text); these stay as a second, independent statement."""
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
            # the block is the origin of the TileRect the reference hands to predict_intra
            # (lookahead.rs:84-89): PredictionVariant::NONE, pred_dc_128, everywhere
            assert got[by * 5 + bx] == satd8(blk - 128), (bx, by)


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


# ---- update_block_importances (src/api/internal.rs:911-1068): the f32 propagation
def importance_model(intra, future, inter, mvs, w, h, length, ref_imp):
    """Independent restatement with numpy float32 scalars (one rounding per operation),
    written from the reference text: floor-to-block of the reference position, the four
    overlap areas, sequential += in source raster order."""
    f32 = np.float32
    out = ref_imp.astype(np.float32).copy()
    for y in range(h):
        for x in range(w):
            i = y * w + x
            rx, ry = x * 64 + int(mvs[i, 1]), y * 64 + int(mvs[i, 0])
            ic, nc = f32(intra[i]), f32(inter[i])
            frac = f32(0.0) if ic <= nc else f32(1.0) - nc / ic
            amount = (ic + f32(future[i])) * frac / f32(length)
            tlx, tly = (rx // 64) * 64, (ry // 64) * 64          # Python floor division
            for (bx, by, ax, ay) in ((tlx, tly, tlx + 64 - rx, tly + 64 - ry),
                                     (tlx + 64, tly, rx - tlx, tly + 64 - ry),
                                     (tlx, tly + 64, tlx + 64 - rx, ry - tly),
                                     (tlx + 64, tly + 64, rx - tlx, ry - tly)):
                dx, dy = bx // 64, by // 64
                if 0 <= dx < w and 0 <= dy < h:
                    out[dy * w + dx] = out[dy * w + dx] + amount * (f32(ax * ay) / f32(4096))
    return out


def n_moved(a, b):
    return int((a > b).sum())


def importance_case(rng, w, h, mv_range):
    n = w * h
    intra = rng.integers(0, 5000, n).astype(np.uint32)
    inter = np.where(rng.random(n) < 0.3, intra + rng.integers(0, 50, n),
                     (intra * rng.random(n)).astype(np.int64)).astype(np.uint32)
    k = min(3, n)
    intra[:k], inter[:k] = [0, 0, 7][:k], [0, 5, 7][:k]        # 0 / 0 never divides: intra <= inter
    future = (rng.random(n) * 3000).astype(np.float32)
    mvs = rng.integers(-mv_range, mv_range + 1, (n, 2)).astype(np.int16)
    mvs[::7] = (mvs[::7] // 64) * 64                           # block-aligned: zero-area neighbours
    ref_imp = (rng.random(n) * 100).astype(np.float32)
    return intra, future, inter, mvs, ref_imp


def test_update_block_importances_against_the_float32_model(oracle):
    rng = np.random.default_rng(41)
    for (w, h, mvr, length) in ((24, 16, 300, 1), (17, 9, 2000, 3), (5, 4, 40, 7), (1, 1, 64, 2)):
        intra, future, inter, mvs, ref_imp = importance_case(rng, w, h, mvr)
        want = importance_model(intra, future, inter, mvs, w, h, length, ref_imp)
        got = ref_imp.copy()
        oracle.r1o_update_block_importances(O.ptr(intra), O.ptr(future), O.ptr(inter), O.ptr(mvs), w, h,
                                            length, O.ptr(got))
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (w, h)
        assert (got >= ref_imp).all() and (n_moved(got, ref_imp) or w * h == 1)


def test_update_block_importances_conserves_what_stays_in_frame(oracle):
    """zero motion: every block keeps its own amount (one overlap of area 1, three of area 0)"""
    rng = np.random.default_rng(42)
    w, h = 12, 7
    intra, future, inter, mvs, ref_imp = importance_case(rng, w, h, 0)
    mvs[:] = 0
    got = np.zeros(w * h, np.float32)
    oracle.r1o_update_block_importances(O.ptr(intra), O.ptr(future), O.ptr(inter), O.ptr(mvs), w, h, 2,
                                        O.ptr(got))
    ic, nc = intra.astype(np.float32), inter.astype(np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        frac = np.where(ic <= nc, np.float32(0), np.float32(1) - nc / ic).astype(np.float32)
    assert np.array_equal(got, ((ic + future) * frac / np.float32(2)).astype(np.float32))
