"""Shared driver of tests/test_oracle_rdo_glue_ref.py and the GPU test of the same vectors
(tests/golden/rdo_glue_ref.npz, produced by executing the reference's text:
tests/golden/gen_rdo_glue_ref.py).  A backend supplies the kernels: the CPU oracle or the C ABI."""
import os

import numpy as np

import oracle_lib as O
from rav1e_amd import rdo_glue as RG
from rav1e_amd.types import BlockSize, TxSize

GOLD = os.path.join(os.path.dirname(__file__), "golden", "rdo_glue_ref.npz")
GOLD_PIXEL = os.path.join(os.path.dirname(__file__), "golden", "rdo_pixel_ref.npz")
GOLD_TXSEARCH = os.path.join(os.path.dirname(__file__), "golden", "rdo_txsearch_ref.npz")
TX_W = [4, 8, 16, 32, 64, 4, 8, 8, 16, 16, 32, 32, 64, 4, 16, 8, 32, 16, 64]
TX_H = [4, 8, 16, 32, 64, 8, 4, 16, 8, 32, 16, 64, 32, 16, 4, 32, 8, 64, 16]


def check_tx_blocks(G, full_cand):
    """encode_tx_block (TxDistEstRate): full_cand(bd, ts, tt, qidx, src_plane, pred_plane) ->
    (tx_dist, est_rate) of the zero-motion candidate at (8, 8)"""
    n = 0
    for k in G["tb_keys"]:
        k = str(k)
        bd, ts, tt, qidx = [int(v) for v in k.split("_")]
        w, h = TX_W[ts], TX_H[ts]
        src = O.HostPlane(w + 16, h + 16, bd, 16, 16)
        pred = O.HostPlane(w + 16, h + 16, bd, 16, 16)
        src.view()[8:8 + h, 8:8 + w] = G["tb_src_" + k]
        pred.view()[8:8 + h, 8:8 + w] = G["tb_pred_" + k]
        got = full_cand(bd, ts, tt, qidx, src, pred)
        want = G["tb_out_" + k]
        assert (int(got[0]), int(got[1])) == (int(want[0]), int(want[1])), (k, got, want)
        n += 1
    return n


def check_pixel_blocks(G, pixel_cand):
    """encode_tx_block's pixel-domain leg + compute_distortion (gen_rdo_pixel_ref.py):
    pixel_cand(bd, ts, tt, qidx, src_plane, pred_plane, dist_kind, scales, scale_stride) ->
    (eob, dist, qcoeffs[coded area], rec[h, w]) of the zero-motion candidate at (8, 8);
    dist_kind 2 = sse_wxh (Tune::Psnr), 3 = cdef_dist_wxh (Tune::Psychovisual); scales = the
    per-importance-block DistortionScale grid (None: temporal RDO off)."""
    n = 0
    for k in G["px_keys"]:
        k = str(k)
        bd, ts, tt, qidx = [int(v) for v in k.split("_")]
        w, h = TX_W[ts], TX_H[ts]
        src = O.HostPlane(w + 16, h + 16, bd, 16, 16)
        pred = O.HostPlane(w + 16, h + 16, bd, 16, 16)
        src.view()[8:8 + h, 8:8 + w] = G["px_src_" + k]
        pred.view()[8:8 + h, 8:8 + w] = G["px_pred_" + k]
        sblk = G["px_scales_" + k]
        gw, gh = (w + 16 + 7) // 8, (h + 16 + 7) // 8
        grid = np.full((gh, gw), 1 << 14, np.uint32)
        grid[1:1 + sblk.shape[0], 1:1 + sblk.shape[1]] = sblk
        want = [int(v) for v in G["px_dist_" + k]]       # [sse, cdef, sse scaled, cdef scaled]
        for j, (kind, sc) in enumerate(((2, None), (3, None), (2, grid), (3, grid))):
            eob, dist, qc, rec = pixel_cand(bd, ts, tt, qidx, src, pred, kind, sc, gw)
            assert int(eob) == int(G["px_eob_" + k][0]), (k, "eob", eob)
            assert np.array_equal(np.asarray(qc).astype(np.int64).ravel(), G["px_qc_" + k].astype(np.int64)), (k, "qcoeffs")
            assert np.array_equal(np.asarray(rec).astype(np.int64).reshape(h, w), G["px_rec_" + k].astype(np.int64)), (k, "rec")
            assert int(dist) == want[j], (k, "dist", j, int(dist), want[j])
            n += 1
    return n


def check_compute_tx_distortion(G, make_sse):
    """make_sse(bd, planes_src, planes_rec) -> sse_wxh(plane, x, y, w, h)"""
    n = 0
    for bd in G["td_keys"]:
        bd = int(bd)
        srcs = [O.plane_from_image(G["td_src_%d_%d" % (bd, p)], bd, 16 >> (1 if p else 0), 16 >> (1 if p else 0))
                for p in range(3)]
        recs = [O.plane_from_image(G["td_rec_%d_%d" % (bd, p)], bd, 16 >> (1 if p else 0), 16 >> (1 if p else 0))
                for p in range(3)]
        sse = make_sse(bd, srcs, recs)
        scales = [int(v) for v in G["td_scales_%d" % bd]]
        fw, fh = srcs[0].width, srcs[0].height
        for row in G["td_rows_%d" % bd]:
            bw, bh, bx, by, skip, luma_only, is_chroma, txd, want = [int(v) for v in row]
            bs = BlockSize["BLOCK_%dX%d" % (bw, bh)]
            got = RG.compute_tx_distortion(sse, fw, fh, bs, bool(is_chroma), bx, by, txd, bool(skip),
                                           bool(luma_only), scales, 1, 1)
            assert got == want, (bd, tuple(int(v) for v in row), got)
            n += 1
    return n


def check_cfl_alpha(G, alpha_search):
    """alpha_search(bd, xdec, ydec, planes_src, planes_rec, uv_tx_size, pli, cx, cy, luma_x, luma_y,
    w_pad, h_pad, vis_w, vis_h, variant) -> alpha"""
    n = 0
    for k0 in G["cfl_keys"]:
        k0 = str(k0)
        bd = int(k0.split("_")[0])
        xdec, ydec = [int(v) for v in G["cfl_dec_" + k0]]
        srcs, recs = [], []
        for p in range(3):
            xd, yd = (xdec, ydec) if p else (0, 0)
            srcs.append(O.plane_from_image(G["cfl_src_%s_%d" % (k0, p)], bd, 16 >> xd, 16 >> yd))
            recs.append(O.plane_from_image(G["cfl_rec_%s_%d" % (k0, p)], bd, 16 >> xd, 16 >> yd))
        fw, fh = srcs[0].width, srcs[0].height
        for row in G["cfl_rows_" + k0]:
            bw, bh, lts, bx, by, au, av = [int(v) for v in row]
            bs = BlockSize["BLOCK_%dX%d" % (bw, bh)]
            uv_ts = RG.largest_chroma_tx_size(bs, xdec, ydec)
            tw, th = TxSize(uv_ts).dims
            cx, cy = (bx << 2) >> xdec, (by << 2) >> ydec
            vw, vh = RG.clip_visible_bsize((fw + xdec) >> xdec, (fh + ydec) >> ydec, tw, th, cx, cy)
            w_pad, h_pad, lx, ly = RG.luma_ac_pads(bs, lts, bx, by, (fw + 3) // 4, (fh + 3) // 4, xdec, ydec)
            variant = 0 if (cx == 0 and cy == 0) else (1 if cy == 0 else (2 if cx == 0 else 3))
            for pli, want in ((1, au), (2, av)):
                got = alpha_search(bd, xdec, ydec, srcs, recs, int(uv_ts), pli, cx, cy, lx, ly, w_pad, h_pad,
                                   vw, vh, variant)
                assert int(got) == want, (k0, tuple(int(v) for v in row), pli, got)
                n += 1
    return n


def check_compound(G, compound):
    """compound(bd, filter_mode, ref_planes, w, h, (x0, y0, col_frac0, row_frac0), (x1, ...)) -> (h, w)"""
    n = 0
    for k in G["pic_keys"]:
        k = str(k)
        bd, filt = int(k.split("_")[0]), {"REGULAR": 0, "SHARP": 2}[k.split("_")[1]]
        refs = []
        for i in range(2):
            pad = G["pic_ref%d_%d" % (i, bd)]
            hp = O.HostPlane(pad.shape[1] - 48, pad.shape[0] - 48, bd, 24, 24)
            hp.data[hp.yorigin - 24:hp.yorigin + hp.height + 24, hp.xorigin - 24:hp.xorigin + hp.width + 24] = pad
            refs.append(hp)
        preds = G["pic_pred_" + k]
        off = 0
        for row in G["pic_rows_" + k]:
            w, h, x, y, r0, c0, r1, c1, _ = [int(v) for v in row]
            ps = []
            for (mr, mc) in ((r0, c0), (r1, c1)):
                rf, cf, px, py = RG.get_mv_params(mr, mc, x, y)
                ps.append((px, py, cf, rf))
            got = np.asarray(compound(bd, filt, refs, w, h, ps[0], ps[1])).reshape(h, w)
            want = preds[off:off + w * h].reshape(h, w)
            off += w * h
            assert np.array_equal(got.astype(np.int64), want.astype(np.int64)), (k, tuple(int(v) for v in row))
            n += 1
    return n


def padded_plane(a, bd, pad):
    """HostPlane whose visible area + `pad` pixels of real padding on every side is the array `a`"""
    H, W = a.shape
    hp = O.HostPlane(W - 2 * pad, H - 2 * pad, bd, pad, pad)
    hp.data[hp.yorigin - pad:hp.yorigin - pad + H, hp.xorigin - pad:hp.xorigin - pad + W] = a
    return hp


def check_txsearch(G, txsearch, dist_scaled, bds=(8, 10, 12), stride=1):
    """rdo_tx_type_decision's per-type evaluations (gen_rdo_txsearch_ref.py part 2).
    txsearch(bd, ts, mask, qidx, is_intra, src_plane, pred_plane, ox, oy, kind, grid) ->
        (eob[nt], dist[nt], qcoeffs[nt, coded area], rec[nt, h, w])       -- ONE launch, all types
    dist_scaled(kind, bd, src_plane, rec_plane, x, y, vw, vh, grid) -> distortion over the visible part
        (blocks cut by the frame edge: compute_distortion clips, the fused kernel measures the whole
        transform block, so the host takes the reconstruction and r1_dist_scaled_batch there).
    grid: the scale grid (covers the padding too; the reference saw its top-left frame part)."""
    fw, fh, pad = [int(v) for v in G["tsr_frame"]]
    n = 0
    planes = {}
    for k in [str(k) for k in G["tsr_keys"]][::stride]:
        bd, ts, inter, qidx, bx, by = [int(v) for v in k.split("_")]
        if bd not in bds:
            continue
        if bd not in planes:
            planes[bd] = (padded_plane(G["tsr_src_%d" % bd], bd, pad), padded_plane(G["tsr_pred_%d" % bd], bd, pad),
                          np.ascontiguousarray(G["tsr_scales_%d" % bd]))
        src, pred, grid = planes[bd]
        w, h = TX_W[ts], TX_H[ts]
        types = [int(t) for t in G["tsr_types_" + k]]
        mask = sum(1 << t for t in types)
        ox, oy = bx * 4, by * 4
        vw, vh = [int(v) for v in G["tsr_vis_" + k]]
        want = G["tsr_dist_" + k]
        for j, (kind, sc) in enumerate(((2, None), (3, None), (2, grid), (3, grid))):
            eob, dist, qc, rec = txsearch(bd, ts, mask, qidx, 0 if inter else 1, src, pred, ox, oy, kind, sc)
            assert [int(v) for v in eob] == [int(v) for v in G["tsr_eob_" + k]], (k, "eob", list(eob))
            assert np.array_equal(np.asarray(qc).astype(np.int64).reshape(len(types), -1),
                                  G["tsr_qc_" + k].astype(np.int64)), (k, "qcoeffs")
            rec = np.asarray(rec).astype(np.int64).reshape(len(types), h, w)
            assert np.array_equal(rec, G["tsr_rec_" + k].astype(np.int64)), (k, "rec")
            if (vw, vh) == (w, h):
                assert [int(v) for v in dist] == [int(v) for v in want[:, j]], (k, "dist", j, list(dist), list(want[:, j]))
            else:
                for s, t in enumerate(types):
                    rp = O.HostPlane(src.width, src.height, bd, src.xpad, src.ypad)
                    rp.data[...] = pred.data
                    rp.data[rp.yorigin + oy:rp.yorigin + oy + h, rp.xorigin + ox:rp.xorigin + ox + w] = rec[s]
                    got = dist_scaled(kind, bd, src, rp, ox, oy, vw, vh, sc)
                    assert int(got) == int(want[s, j]), (k, "clipped dist", t, j, int(got), int(want[s, j]))
            n += len(types)
    return n


def check_txsplit(G, txsearch_pred, dist_scaled, bds=(8, 10, 12)):
    """The next transform depth of an inter block (gen_rdo_txsearch_ref.py part 4: write_tx_tree executed whole with
    tx_size < bsize, then compute_distortion on the block) as a COMPOSITION of the type search's dense-prediction form:
    txsearch_pred(bd, ts, mask, qidx, src_plane, preds[n_tx, th, tw], positions[(x, y)], kind, grid) ->
        (eob[n_tx, nt], dist[n_tx, nt], qcoeffs[n_tx, nt, coded], rec[n_tx, nt, th, tw])      -- ONE launch
    The transform blocks are rdo_glue.tx_split_blocks', their predictions the same rectangles of the block's prediction.
    Checked per type: eob and coefficients of every transform block in call order, the assembled reconstruction, the
    block's four distortions over its visible part -- and, where every transform block is a whole number of 8x8 tiles
    inside the frame, that the block's distortion IS the sum of the launch's own per-block distortions (cdef_dist, and
    SSE without a scale grid)."""
    fw, fh, pad = [int(v) for v in G["tsr_frame"]]
    n, n_sum = 0, 0
    planes = {}
    for k in [str(k) for k in G["txs_keys"]]:
        f = k.split("_")
        bd, ts, qidx, bx, by = int(f[0]), int(f[2]), int(f[3]), int(f[4]), int(f[5])
        if bd not in bds:
            continue
        if bd not in planes:
            planes[bd] = (padded_plane(G["tsr_src_%d" % bd], bd, pad), padded_plane(G["tsr_pred_%d" % bd], bd, pad),
                          np.ascontiguousarray(G["tsr_scales_%d" % bd]))
        src, pred, grid = planes[bd]
        bw, bh, tw, th, n_tx = [int(v) for v in G["txs_geom_" + k]]
        types = [int(t) for t in G["txs_types_" + k]]
        mask = sum(1 << t for t in types)
        blocks = RG.tx_split_blocks(bx, by, bw, bh, tw, th, (fw + 3) // 4, (fh + 3) // 4)
        assert len(blocks) == n_tx, (k, blocks)
        pv = pred.data[pred.yorigin:, pred.xorigin:]
        preds = np.stack([pv[y:y + th, x:x + tw] for (_, x, y) in blocks])
        ox, oy = bx * 4, by * 4
        vw, vh = min(bw, fw - ox), min(bh, fh - oy)
        want = G["txs_dist_" + k]
        for j, (kind, sc) in enumerate(((2, None), (3, None), (2, grid), (3, grid))):
            eob, dist, qc, rec = txsearch_pred(bd, ts, mask, qidx, src, preds, [(x, y) for (_, x, y) in blocks], kind, sc)
            eob = np.asarray(eob).astype(np.int64).reshape(n_tx, len(types))
            assert np.array_equal(eob.T, G["txs_eob_" + k].astype(np.int64)), (k, "eob")
            qc = np.asarray(qc).astype(np.int64).reshape(n_tx, len(types), -1)
            assert np.array_equal(qc.transpose(1, 0, 2), G["txs_qc_" + k].astype(np.int64)), (k, "qcoeffs")
            rec = np.asarray(rec).astype(np.int64).reshape(n_tx, len(types), th, tw)
            dist = np.asarray(dist).astype(np.uint64).reshape(n_tx, len(types))
            for s, t in enumerate(types):
                blk = pv[oy:oy + bh, ox:ox + bw].astype(np.int64).copy()
                for b, (_, x, y) in enumerate(blocks):
                    blk[y - oy:y - oy + th, x - ox:x - ox + tw] = rec[b, s]
                assert np.array_equal(blk, G["txs_rec_" + k][s].astype(np.int64)), (k, "rec", t)
                rp = O.HostPlane(src.width, src.height, bd, src.xpad, src.ypad)
                rp.data[...] = pred.data
                rp.data[rp.yorigin + oy:rp.yorigin + oy + bh, rp.xorigin + ox:rp.xorigin + ox + bw] = blk
                got = dist_scaled(kind, bd, src, rp, ox, oy, vw, vh, sc)
                assert int(got) == int(want[s, j]), (k, "block dist", t, j, int(got), int(want[s, j]))
                # (not for the weighted SSE with a scale grid: get_weighted_sse rounds ONCE per call, `(sum + 32) >> 6`,
                # dist.rs:281 -- the sum of the quadrants' rounded values is not the block's; cdef_dist_wxh rounds per 8x8 tile)
                if j != 2 and (vw, vh) == (bw, bh) and tw % 8 == 0 and th % 8 == 0 and len(blocks) == (bw // tw) * (bh // th):
                    assert int(dist[:, s].sum()) == int(want[s, j]), (k, "sum of the transform blocks' distortions", t, j)
                    n_sum += 1
            n += len(types)
    assert n_sum > 100
    return n


def check_compute_distortion(G, make_dist):
    """compute_distortion with chroma (gen_rdo_txsearch_ref.py part 3).
    make_dist(bd, planes_src, planes_rec, grid_or_None, xdec, ydec) -> dist_wxh(kind, plane, x, y, w, h)"""
    n = 0
    for k0 in [str(k) for k in G["cd_keys"]]:
        bd = int(k0.split("_")[0])
        xdec, ydec = [int(v) for v in G["cd_dec_" + k0]]
        srcs, recs = [], []
        for p in range(3):
            xd, yd = (xdec, ydec) if p else (0, 0)
            srcs.append(O.plane_from_image(G["cd_src_%s_%d" % (k0, p)], bd, 16 >> xd, 16 >> yd))
            recs.append(O.plane_from_image(G["cd_rec_%s_%d" % (k0, p)], bd, 16 >> xd, 16 >> yd))
        grid = np.ascontiguousarray(G["cd_scales_" + k0])
        ds3 = [int(v) for v in G["cd_dist_scale_" + k0]]
        fw, fh = [int(v) for v in G["cd_frame_" + k0]]
        fns = {0: make_dist(bd, srcs, recs, None, xdec, ydec), 1: make_dist(bd, srcs, recs, grid, xdec, ydec)}
        for row in G["cd_rows_" + k0]:
            bw, bh, bx, by, tune_i, use_scales, luma_only, want = [int(v) for v in row]
            bs = BlockSize["BLOCK_%dX%d" % (bw, bh)]
            got = RG.compute_distortion(fns[use_scales], fw, fh, bs, True, bx, by, bool(luma_only), ds3,
                                        bool(tune_i), xdec, ydec)
            assert got == want, (k0, tuple(int(v) for v in row), got)
            n += 1
    return n
