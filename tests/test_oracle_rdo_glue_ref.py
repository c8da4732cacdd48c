"""The N4 glue of the oracle (estimate_rate, the TxDistEstRate evaluation of encode_tx_block,
compute_tx_distortion, rdo_cfl_alpha, predict_inter_compound) against rdo_glue_ref.npz -- vectors
produced by executing the reference's own text (tests/golden/gen_rdo_glue_ref.py)."""
import ctypes as C
import os

import numpy as np

import oracle_lib as O
import rdo_glue_cases as RC


def test_estimate_rate_reproduces_the_executed_reference(oracle):
    G = np.load(RC.GOLD)
    oracle.r1o_estimate_rate.restype = C.c_uint64
    oracle.r1o_estimate_rate.argtypes = [C.c_int, C.c_int, C.c_uint64]
    for qi, ts, d, want in G["rate"]:
        assert oracle.r1o_estimate_rate(int(qi), int(ts), int(d)) == int(want), (qi, ts, d)
    # the reference's own test (src/rdo.rs:2749-2752, `estimate_rate_test`): the first table entry
    import re
    tab = open(os.path.join(os.path.dirname(__file__), "..", "oracle", "rate_table.inc")).read()
    first = int(re.search(r"\{\s*\{\s*\{\s*(\d+)", tab).group(1))
    assert oracle.r1o_estimate_rate(0, 0, 0) == first == 99999


def test_encode_tx_block_tx_domain(oracle):
    G = np.load(RC.GOLD)

    def full_cand(bd, ts, tt, qidx, src, pred):
        w, h = RC.TX_W[ts], RC.TX_H[ts]
        c = np.zeros(1, O.RDO_CAND)
        c["ox"], c["oy"], c["rx"], c["ry"], c["tx_type"] = 8, 8, 8, 8, tt
        pa, pb = src.cstruct(), pred.cstruct()
        sad, satd = np.zeros(1, np.uint32), np.zeros(1, np.uint32)
        eob, dist, rate = np.zeros(1, np.uint16), np.zeros(1, np.uint64), np.zeros(1, np.uint64)
        assert oracle.r1o_rdo_full_cand_batch(C.byref(pa), C.byref(pb), w, h, ts, O.ptr(c), 1, qidx, 0, 0, 0,
                                              O.ptr(sad), O.ptr(satd), O.ptr(eob), O.ptr(dist), O.ptr(rate),
                                              None) == 0
        return int(dist[0]), int(rate[0])
    assert RC.check_tx_blocks(G, full_cand) == 156


def test_encode_tx_block_pixel_domain_and_compute_distortion(oracle):
    """rav1e's default tune: dequantize -> inverse_transform_add into the reconstruction
    (src/encoder.rs:1588-1614), then compute_distortion (src/rdo.rs:254-347) -- the oracle's
    r1o_rdo_pixel_cand_batch against the executed reference text (gen_rdo_pixel_ref.py)."""
    G = np.load(RC.GOLD_PIXEL)

    def pixel_cand(bd, ts, tt, qidx, src, pred, kind, scales, stride):
        w, h = RC.TX_W[ts], RC.TX_H[ts]
        hbd = int(bd > 8)
        c = np.zeros(1, O.RDO_CAND)
        c["ox"], c["oy"], c["rx"], c["ry"], c["tx_type"] = 8, 8, 8, 8, tt
        pa, pb = src.cstruct(), pred.cstruct()
        eob, dist = np.zeros(1, np.uint16), np.zeros(1, np.uint64)
        qc = np.zeros(min(w, 32) * min(h, 32), np.int32 if hbd else np.int16)
        rec = np.zeros((h, w), np.uint16 if hbd else np.uint8)
        sc = None if scales is None else O.ptr(np.ascontiguousarray(scales))
        assert oracle.r1o_rdo_pixel_cand_batch(C.byref(pa), C.byref(pb), w, h, ts, O.ptr(c), 1, qidx, 0, 0, 0, kind,
                                               sc, stride, 0, 0, None, None, O.ptr(eob), O.ptr(dist), O.ptr(qc),
                                               O.ptr(rec), None) == 0
        return int(eob[0]), int(dist[0]), qc, rec
    assert RC.check_pixel_blocks(G, pixel_cand) == 482 * 4


def test_compute_tx_distortion(oracle):
    G = np.load(RC.GOLD)

    def make_sse(bd, srcs, recs):
        def sse(p, x, y, w, h):
            c = np.zeros(1, O.DIST_CAND)
            c["ox"], c["oy"], c["rx"], c["ry"] = x, y, x, y
            pa, pb = srcs[p].cstruct(), recs[p].cstruct()
            out = np.zeros(1, np.uint64)
            assert oracle.r1o_dist_scaled_batch(2, C.byref(pa), C.byref(pb), w, h, O.ptr(c), 1, None, 0,
                                                1 if p else 0, 1 if p else 0, O.ptr(out)) == 0
            return int(out[0])
        return sse
    assert RC.check_compute_tx_distortion(G, make_sse) == 2 * 11 * 8


def test_rdo_cfl_alpha(oracle):
    G = np.load(RC.GOLD)

    def alpha_search(bd, xdec, ydec, srcs, recs, uv_ts, pli, cx, cy, lx, ly, w_pad, h_pad, vw, vh, variant):
        hbd = int(bd > 8)
        dt = np.uint16 if hbd else np.uint8
        tw, th = RC.TX_W[uv_ts], RC.TX_H[uv_ts]
        rec, src, luma = recs[pli], srcs[pli], recs[0]
        edge = np.zeros(257, dt)
        lens = (C.c_int * 2)()
        tile = rec.block_ptr(0, 0)
        oracle.r1o_get_intra_edges(O.ptr(edge), lens, tile, rec.stride, cx, cy, rec.width, rec.height, uv_ts, bd,
                                   13, 1, 0, 0, 0, hbd)
        ac = np.zeros(tw * th, np.int16)
        oracle.r1o_pred_cfl_ac(O.ptr(ac), luma.block_ptr(lx, ly), luma.stride, tw, th, w_pad, h_pad, xdec, ydec, hbd)
        s = src.view()[cy:cy + vh, cx:cx + vw].astype(np.int64)
        costs = {}
        for a in range(-16, 17):
            out = np.zeros((th, tw), dt)
            assert oracle.r1o_dispatch_predict_intra(13 if a else 0, variant, O.ptr(out), tw, uv_ts, bd, O.ptr(ac), a,
                                                     0, O.ptr(edge), lens[0], lens[1], tw, th, hbd) == 0
            d = s - out[:vh, :vw].astype(np.int64)
            costs[a] = int((d * d).sum())
        best, best_a, count = costs[0], 0, 2
        for a in range(1, 17):
            if costs[a] < best:
                best, best_a, count = costs[a], a, count + 2
            if costs[-a] < best:
                best, best_a, count = costs[-a], -a, count + 2
            if count < a:
                break
        return best_a
    assert RC.check_cfl_alpha(G, alpha_search) == 4 * 8 * 2


def test_predict_inter_compound(oracle):
    G = np.load(RC.GOLD)

    def compound(bd, filt, refs, w, h, p0, p1):
        tmps = []
        for hp, (x, y, cf, rf) in zip(refs, (p0, p1)):
            c = np.zeros(1, O.MC_CAND)
            c["rx"], c["ry"], c["col_frac"], c["row_frac"], c["mode_x"], c["mode_y"] = x, y, cf, rf, filt, filt
            t = np.zeros(w * h, np.int16)
            pc = hp.cstruct()
            assert oracle.r1o_mc_prep_batch(C.byref(pc), w, h, O.ptr(c), 1, O.ptr(t)) == 0
            tmps.append(t)
        out = np.zeros(w * h, np.uint8 if bd == 8 else np.uint16)
        assert oracle.r1o_mc_avg_batch(O.ptr(tmps[0]), O.ptr(tmps[1]), w, h, 1, bd, 1 if bd == 8 else 2, O.ptr(out)) == 0
        return out
    assert RC.check_compound(G, compound) == 3 * 2 * 21
