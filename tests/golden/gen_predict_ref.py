#!/usr/bin/env python3
"""tests/golden/predict_ref.npz: intra prediction, edge building and CFL-AC vectors
computed by the REFERENCE'S OWN SOURCE TEXT, transpiled by tools/rustlite:

  dispatch_predict_intra + every kernel it reaches        src/predict.rs:705-1505
    (pred_dc*, pred_v/h, pred_paeth, pred_smooth*, pred_cfl*, pred_directional,
     filter_edge, upsample_edge, select_ief_strength/upsample, dr_intra_derivative,
     sm_weight_arrays, IntraEdgeFilterParameters::use_smooth_filter)
  pred_cfl_ac::<T, XDEC, YDEC>                            src/predict.rs:1020-1063
  get_intra_edges, IntraEdge::new, has_top_right,
  has_bottom_left, supersample_chroma_bsize               src/partition.rs:400-898

Sections of the file
  (top level)  same record layout as predict_golden.npz (dispatch_predict_intra cases)
  e_*          get_intra_edges cases: tile pixels, geometry, the two availability
               answers the reference computed, resulting edge buffer + lengths
  a_*          pred_cfl_ac cases

Run in the build container:  python tests/golden/gen_predict_ref.py
"""
import numpy as np

import reflib as L
from reflib import R

TX_W = [4, 8, 16, 32, 64, 4, 8, 8, 16, 16, 32, 32, 64, 4, 16, 8, 32, 16, 64]
TX_H = [4, 8, 16, 32, 64, 8, 4, 16, 8, 32, 16, 64, 32, 16, 4, 32, 8, 64, 16]
MODE_ANGLE = {1: 90, 2: 180, 3: 45, 4: 135, 5: 113, 6: 157, 7: 203, 8: 67}
MODES = ["DC_PRED", "V_PRED", "H_PRED", "D45_PRED", "D135_PRED", "D113_PRED", "D157_PRED", "D203_PRED",
         "D67_PRED", "SMOOTH_PRED", "SMOOTH_V_PRED", "SMOOTH_H_PRED", "PAETH_PRED", "UV_CFL_PRED"]


def main():
    c = L.crate("predict.rs", "partition.rs", "context/block_unit.rs")
    dispatch = c.get("dispatch_predict_intra")
    get_edges = c.get("get_intra_edges")
    cfl_ac = c.get("pred_cfl_ac")
    has_tr, has_bl = c.get("has_top_right"), c.get("has_bottom_left")
    supersample = c.get("supersample_chroma_bsize")
    PM = [L.enum(c, "PredictionMode", n) for n in MODES]
    assert [m.disc for m in PM] == list(range(14))
    PV = [L.enum(c, "PredictionVariant", n) for n in ("NONE", "LEFT", "TOP", "BOTH")]
    c.autoload("TxSize")
    TxSize = [L.enum(c, "TxSize", v[0]) for v in c.enums["TxSize"].variants]
    BS = {v[0]: L.enum(c, "BlockSize", v[0]) for v in c.enums["BlockSize"].variants}
    IntraEdge, IEF = L.struct(c, "IntraEdge"), L.struct(c, "IntraEdgeFilterParameters")
    INTRA_FRAME = L.enum(c, "RefType", "INTRA_FRAME")
    NONE_FRAME = L.enum(c, "RefType", "NONE_FRAME")
    TBO, BO = L.struct(c, "TileBlockOffset"), L.struct(c, "BlockOffset")
    rng = np.random.default_rng(20260927)

    def ief_params(ief):
        if ief == 0:
            return R.NONE
        refs = R.Some(R.array(INTRA_FRAME, NONE_FRAME))
        mode = R.Some(PM[9]) if ief == 2 else R.Some(PM[0])     # a SMOOTH neighbour / a DC neighbour
        return R.Some(IEF(plane=0, above_ref_frame_types=refs, left_ref_frame_types=refs,
                          above_mode=mode, left_mode=R.Some(PM[0])))

    def run_dispatch(mode, variant, angle, ief, bd, ts, e, left_len, above_len, aw, ah, ac):
        w, h = TX_W[ts], TX_H[ts]
        g = L.pixel_type(bd)
        dt = L.np_dtype(bd)
        # the block sits at (8, 8) of a plane whose visible size ends avail_w / avail_h
        # pixels after the block origin (frame-edge blocks reach into the padding)
        p = L.plane_from_array(np.zeros((8 + ah, 8 + aw), dt), bd, xpad=80, ypad=80)
        reg = p._region(8, 8, w, h)
        ev = [int(v) for v in e]
        edge = IntraEdge(R.RSlice(ev, 128 - left_len, left_len), R.RSlice(ev, 128, 1),
                         R.RSlice(ev, 129, above_len))
        acs = R.RSlice([int(v) for v in ac]) if ac is not None else R.RSlice([])
        dispatch(g, PM[mode], PV[variant], reg, TxSize[ts], bd, acs, angle, ief_params(ief), edge, None)
        cfg = p.cfg
        out = np.zeros((h, w), np.uint16)
        for y in range(h):
            base = (cfg.yorigin + 8 + y) * cfg.stride + cfg.xorigin + 8
            out[y] = p.data[base:base + w]
        return out

    recs = {k: [] for k in ("ts", "mode", "variant", "angle", "ief", "bd", "left_len", "above_len",
                            "avail_w", "avail_h", "off")}
    edges, outs, acs = [], [], []
    off = 0
    for ts in range(19):
        w, h = TX_W[ts], TX_H[ts]
        for bd in ((8, 10, 12) if max(w, h) <= 8 else (8, 10) if max(w, h) <= 16 else
                   ((8,) if ts % 2 else (10,))):
            def new_edge(smooth_edge):
                if smooth_edge:
                    e = np.cumsum(rng.integers(-6, 7, 257)) * (1 << (bd - 8)) + (1 << (bd - 1))
                    return np.clip(e, 0, (1 << bd) - 1).astype(np.uint16)
                return rng.integers(0, 1 << bd, 257).astype(np.uint16)

            def emit(mode, variant, angle, ief, left_len, above_len, aw, ah, out, e, ac=None):
                for k, v in zip(recs, (ts, mode, variant, angle, ief, bd, left_len, above_len, aw, ah, off)):
                    recs[k].append(v)
                edges.append(e)
                outs.append(out.astype(np.uint16).ravel())
                acs.append(ac if ac is not None else np.zeros(0, np.int16))

            for mode, base in MODE_ANGLE.items():
                deltas = (-3, -2, -1, 0, 1, 2, 3) if w * h <= 256 else (-3, 0, 2)
                for d in deltas:
                    pa = base + 3 * d
                    if pa in (90, 180) and mode not in (1, 2):
                        continue
                    for ief in (0, 1, 2):
                        e = new_edge(ief != 0)
                        above_len = w + (h if pa < 90 else 0)
                        left_len = h + (w if pa > 180 else 0)
                        aw = w if rng.random() < 0.7 else int(rng.integers(1, w + 1))
                        ah = h if rng.random() < 0.7 else int(rng.integers(1, h + 1))
                        out = run_dispatch(mode, 3, pa, ief, bd, ts, e, left_len, above_len, aw, ah, None)
                        emit(mode, 3, pa, ief, left_len, above_len, aw, ah, out, e)
                        off += w * h
            for mode in (0, 9, 10, 11, 12, 13):
                for variant in ((0, 1, 2, 3) if mode in (0, 13) else (3,)):
                    e = new_edge(False)
                    alpha = int(rng.integers(-16, 17)) if mode == 13 else 0
                    if mode == 13 and alpha == 0:
                        alpha = 5
                    ac = None
                    if mode == 13:
                        ac = rng.integers(-(1 << (bd + 2)), 1 << (bd + 2), w * h).astype(np.int16)
                        ac -= np.int16(ac.astype(np.int64).sum() // (w * h))
                    out = run_dispatch(mode, variant, alpha, 0, bd, ts, e, h, w, w, h, ac)
                    emit(mode, variant, alpha, 0, h, w, w, h, out, e, ac)
                    off += w * h
        print("tx size", ts, len(edges), flush=True)
    d = {k: np.asarray(v, np.int32) for k, v in recs.items()}
    d["edges"] = np.stack(edges)
    d["out"] = np.concatenate(outs)
    d["ac_off"] = np.cumsum([0] + [len(a) for a in acs]).astype(np.int64)
    d["ac"] = np.concatenate(acs)

    # ---------------- get_intra_edges
    # A tile region inside a plane (the plane's visible size may cut the tile: rect_w/h),
    # partitions of several sizes, every tx block of the partition.
    e_rec = {k: [] for k in ("case", "x", "y", "ts", "bd", "mode", "enable_ief", "angle_delta", "has_tr",
                             "has_bl", "rect_w", "rect_h", "xdec", "left_len", "above_len")}
    e_tiles, e_edges = [], []
    geoms = [  # (plane_w, plane_h, tile_x, tile_y, tile_w, tile_h, xdec, ydec)
        (96, 64, 0, 0, 96, 64, 0, 0), (96, 64, 32, 0, 64, 64, 0, 0), (90, 60, 0, 0, 128, 64, 0, 0),
        (48, 32, 0, 0, 48, 32, 1, 1), (45, 30, 0, 0, 64, 32, 1, 1)]
    parts = [("BLOCK_8X8", 1), ("BLOCK_16X16", 2), ("BLOCK_16X16", 1), ("BLOCK_32X32", 3), ("BLOCK_32X16", 2),
             ("BLOCK_8X16", 1), ("BLOCK_64X64", 3), ("BLOCK_4X4", 0), ("BLOCK_16X8", 6), ("BLOCK_8X32", 15)]
    case = 0
    for bd in (8, 10, 12):
        g = L.pixel_type(bd)
        dt = L.np_dtype(bd)
        for (pw, ph, tx0, ty0, tw, th, xdec, ydec) in geoms:
            img = rng.integers(0, 1 << bd, (ph + 80, pw + 80)).astype(dt)      # real pixels in the padding too
            plane = L.plane_from_padded(img, bd, 40, 40, xdec, ydec)
            region = plane._region(tx0, ty0, tw, th)
            rect_w, rect_h = min(tw, pw - tx0), min(th, ph - ty0)
            e_tiles.append(img[40 + ty0: 40 + ty0 + th, 40 + tx0: 40 + tx0 + tw].astype(np.uint16))
            for (bname, ts) in parts:
                bs = BS[bname]
                bw, bh = (int(v) for v in bname[6:].split("X"))
                w, h = TX_W[ts], TX_H[ts]
                # block sizes are luma sizes; a chroma plane sees them decimated
                pbw, pbh = max(bw >> xdec, 4), max(bh >> ydec, 4)
                if w > pbw or h > pbh:
                    continue
                for _ in range(5):
                    # partition origin in luma 4x4 units, inside the tile
                    box = int(rng.integers(0, max(1, (tw << xdec) // bw))) * (bw // 4)
                    boy = int(rng.integers(0, max(1, (th << ydec) // bh))) * (bh // 4)
                    if xdec and bw == 4:
                        box |= 1       # chroma of a 4x4 luma block is coded with the odd block
                        boy |= 1
                    bx, by = int(rng.integers(0, pbw // w)), int(rng.integers(0, pbh // h))
                    x = ((box >> xdec) << 2 >> 0) // 1
                    x = (box * 4 >> xdec) // 4 * 4 if not (xdec and bw == 4) else ((box - 1) * 4 >> xdec)
                    y = (boy * 4 >> ydec) // 4 * 4 if not (ydec and bh == 4) else ((boy - 1) * 4 >> ydec)
                    x += bx * w
                    y += by * h
                    if x >= rect_w or y >= rect_h or x + w > tw or y + h > th:
                        continue
                    mode = int(rng.integers(-1, 14))
                    enable_ief = int(rng.integers(0, 2))
                    ad = int(rng.integers(-3, 4)) if 1 <= mode <= 8 else 0
                    pbo = TBO(BO(x=box, y=boy))
                    po = R.PlaneOffset(x, y)
                    buf = R.Aligned(R.RSlice([0xFFFF] * 257))
                    opt_mode = R.Some(PM[mode]) if mode >= 0 else R.NONE
                    iparam = c.G["_E"]("IntraParam", "AngleDelta", 0, (ad,)) if ad else \
                        L.enum(c, "IntraParam", "None")
                    edge = get_edges(g, buf, region, pbo, bx, by, bs, po, TxSize[ts], bd, opt_mode,
                                     bool(enable_ief), iparam)
                    left, tl, above = edge._0, edge._1, edge._2
                    # the two availability answers, asked again with the reference's own expressions
                    bx4, by4 = bx * (w >> 2), by * (h >> 2)
                    have_top = by4 != 0 or (boy > 1 if ydec else boy > 0)
                    have_left = bx4 != 0 or (box > 1 if xdec else box > 0)
                    sps = supersample({}, bs, xdec, ydec)
                    tr = bool(y != 0 and has_tr({}, sps, pbo, have_top, x + w < rect_w, TxSize[ts], by4, bx4,
                                                 xdec, ydec))
                    bl = bool(x != 0 and has_bl({}, sps, pbo, y + h < rect_h, have_left, TxSize[ts], by4, bx4,
                                                 xdec, ydec))
                    full = np.full(257, 0xFFFF, np.uint16)
                    full[128 - left.len():128] = left.tolist()
                    full[128] = tl.tolist()[0]
                    full[129:129 + above.len()] = above.tolist()
                    for k, v in zip(e_rec, (case, x, y, ts, bd, mode, enable_ief, ad, int(tr), int(bl), rect_w,
                                            rect_h, xdec, left.len(), above.len())):
                        e_rec[k].append(v)
                    e_edges.append(full)
            case += 1
    for k, v in e_rec.items():
        d["e_" + k] = np.asarray(v, np.int32)
    d["e_edges"] = np.stack(e_edges)
    for i, t in enumerate(e_tiles):
        d["e_tile_%d" % i] = t
    print("edge cases", len(e_edges), flush=True)

    # ---------------- pred_cfl_ac
    a_rec = {k: [] for k in ("bd", "bw", "bh", "w_pad", "h_pad", "xdec", "ydec", "off")}
    a_luma, a_out = [], []
    aoff = 0
    for bd in (8, 10, 12):
        g = L.pixel_type(bd)
        dt = L.np_dtype(bd)
        for (xdec, ydec) in ((0, 0), (1, 0), (1, 1)):
            for pbs in ("BLOCK_4X4", "BLOCK_8X8", "BLOCK_16X16", "BLOCK_32X32", "BLOCK_8X16", "BLOCK_16X8",
                        "BLOCK_4X16", "BLOCK_32X8", "BLOCK_16X32"):
                bw, bh = (int(v) for v in pbs[6:].split("X"))
                for rep in range(2):
                    w_pad = int(rng.integers(0, bw // 4)) if rep and bw > 4 else 0
                    h_pad = int(rng.integers(0, bh // 4)) if rep and bh > 4 else 0
                    lw, lh = max(bw << xdec, 8), max(bh << ydec, 8)
                    luma = rng.integers(0, 1 << bd, (lh, lw)).astype(dt)
                    reg = L.plane_from_array(luma, bd).as_region()
                    ac = R.RSlice([0] * (bw * bh))
                    cfl_ac({**g, "XDEC": xdec, "YDEC": ydec}, ac, reg, BS[pbs], w_pad, h_pad, None)
                    for k, v in zip(a_rec, (bd, bw, bh, w_pad, h_pad, xdec, ydec, aoff)):
                        a_rec[k].append(v)
                    a_luma.append(luma.astype(np.uint16).ravel())
                    a_out.append(np.array(ac.tolist(), np.int16))
                    aoff += bw * bh
    for k, v in a_rec.items():
        d["a_" + k] = np.asarray(v, np.int32)
    d["a_luma_off"] = np.cumsum([0] + [len(a) for a in a_luma]).astype(np.int64)
    d["a_luma"] = np.concatenate(a_luma)
    d["a_out"] = np.concatenate(a_out)
    L.save("predict_ref.npz", d)
    print(len(edges), "dispatch cases,", len(e_edges), "edge cases,", len(a_out), "cfl_ac cases")


if __name__ == "__main__":
    main()
