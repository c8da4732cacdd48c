#!/usr/bin/env python3
"""tests/golden/rdo_txsearch_ref.npz: the transform-TYPE search of one prediction
(rdo_tx_type_decision, src/rdo.rs:1701-1817) and compute_distortion's chroma leg, computed by the
REFERENCE'S OWN SOURCE TEXT through tools/rustlite.

Part 1 -- which types the loop visits ("ts_mask"): for every TxSize x is_inter x use_reduced_set
  get_tx_set                       src/context/transform_unit.rs:123-148   (executed)
  av1_tx_used                      src/context/transform_unit.rs:37-44     (the static, read)
  RAV1E_TX_TYPES                   src/transform/mod.rs:28-44              (the static, read)
  through a three-line probe that restates the loop's filter (rdo.rs:1731-1736):
      for &tx_type in tx_types { if av1_tx_used[tx_set as usize][tx_type as usize] == 0 { continue; } .. }

Part 2 -- what each visited type evaluates to ("tsr_*"): several TxTypes on ONE (source, prediction)
  pair -- the loop body of rdo_tx_type_decision for an inter block / an intra block whose transform
  block is the whole block (write_tx_tree / write_tx_blocks -> one encode_tx_block call):
  encode_tx_block, RDOType::PixelDistRealRate   src/encoder.rs:1404-1661   (whole function)
  compute_distortion(.., luma_only = true)      src/rdo.rs:254-347          (whole function; that
      is the call of rdo.rs:1790: `compute_distortion(fi, ts, bsize, is_chroma_block, tile_bo, true)`)
  at bit depths 8 / 10 / 12, for is_inter = false and true (quantizer offsets + tx set), at block
  positions that move the DistortionScale grid offset, and for blocks CUT BY THE FRAME EDGE (frame
  102 x 78: visible widths that are not multiples of 4) -- compute_distortion clips to the visible
  part (clip_visible_bsize) while encode_tx_block transforms the whole block out of the padding.
  motion_compensate (rdo.rs:1738-1742) is not run: the prediction is what `rec` holds on entry and
  does not depend on the type (the GPU side makes it once per candidate; r1_mc_put_batch is pinned by
  mc_ref.npz).

Part 3 -- compute_distortion with chroma ("cd_*"): is_chroma_block = true, luma_only = false on
  4:2:0 / 4:2:2 / 4:4:4 planes (src/rdo.rs:305-345: per-plane sse_wxh on the decimated planes, the
  4 + visible rule of sub-8x8 blocks, fi.dist_scale[p]), both tunes, with the per-importance-block
  scale grid, blocks cut by the frame edge.

Part 4 -- the NEXT TRANSFORM DEPTH of an inter block ("txs_*", round 6): rdo_tx_size_type (src/rdo.rs:745-815) tries
  tx_size one step down; for an inter block that is rdo_tx_type_decision again with the smaller tx_size on the SAME
  prediction (no re-prediction), i.e. per TxType
  write_tx_tree(.., bsize, tx_size < bsize, tx_type, skip = false, luma_only = true, ..)   src/encoder.rs:2409-2483
      (EXECUTED WHOLE: the bw x bh loop over transform blocks, the mi-grid clip, ts.qc.update, get_qidx ->
      encode_tx_block per transform block)
  compute_distortion(.., bsize, .., luma_only = true)                                        src/rdo.rs:254-347
  on square and rectangular blocks (16x16 -> 4 x 8x8, 32x32 -> 4 x 16x16, 8x8 -> 4 x 4x4, 16x8 -> 2 x 8x8, 8x16,
  64x64 -> 4 x 32x32 ...), bit depths 8 / 10 / 12, interior and frame-edge positions (the mi-grid clip drops transform
  blocks that start outside the frame).  Per (case, type): eob and qcoeffs of every transform block in call order,
  the reconstructed block, the four distortions of part 2.

Stand-ins: as gen_rdo_pixel_ref.py (get_func = the impl_1d_tx! networks of gen_fwd_tx_golden.py; a
ContextWriter that records what write_coeffs_lv_map is handed; v_frame's ChromaSampling).
Hand-stated (plain data): FrameInvariants / Sequence / TileStateMut / CodedFrameData field values.

Run in the build container:  python tests/golden/gen_rdo_txsearch_ref.py
"""
import numpy as np

import gen_fwd_tx_golden as FT
import reflib as L
from reflib import R
from gen_rdo_glue_ref import FILES, TX_W, TX_H, Obj, BitCounter, make_struct
from gen_rdo_pixel_ref import CoeffRecorder

PROBE = """
pub fn r1_probe_tx_used(tx_size: TxSize, is_inter: bool, use_reduced_set: bool, tx_type: TxType) -> bool {
  let tx_set = get_tx_set(tx_size, is_inter, use_reduced_set);
  av1_tx_used[tx_set as usize][tx_type as usize] != 0
}
pub fn r1_probe_rav1e_len() -> usize { RAV1E_TX_TYPES.len() }
pub fn r1_probe_rav1e_type(i: usize) -> TxType { RAV1E_TX_TYPES[i] }
"""
PAD = 40          # plane padding of the part-2 frames: a 32x32 block at the last visible pixel stays inside


def padded_to_plane(a, bd, pad):
    return L.plane_from_padded(a, bd, pad, pad)


def plane_block(p, x, y, w, h, dtype):
    cfg = p.cfg
    out = np.zeros((h, w), dtype)
    for r in range(h):
        base = (cfg.yorigin + y + r) * cfg.stride + cfg.xorigin + x
        out[r] = p.data[base:base + w]
    return out


def main():
    c = L.crate(*(FILES + ["context/transform_unit.rs"]))
    L.load_v_frame_types(c)
    c.load_text("<probe: the type filter of rdo_tx_type_decision's loop>", PROBE)
    ns, _ = FT.load_reference_1d()

    def get_func(_g, t):
        idx = t.disc if hasattr(t, "disc") else int(t)
        name, n = FT.TXFM[idx], FT.TXFM_LEN[idx]

        def run(coeffs):
            buf = FT.Buf(n)
            for i in range(n):
                buf[i] = FT.V(np.array([coeffs[i]], np.int32))
            ns[name](buf)
            for i in range(n):
                coeffs[i] = int(buf[i].v[0])
        return run
    c.define_py("get_func", get_func)

    TxSize = [L.enum(c, "TxSize", v[0]) for v in c.enums["TxSize"].variants]
    TxType = [L.enum(c, "TxType", v[0]) for v in c.enums["TxType"].variants]
    BlockSize = {v[0]: L.enum(c, "BlockSize", v[0]) for v in c.enums["BlockSize"].variants}
    DS = L.struct(c, "DistortionScale")
    TileStateMut = L.struct(c, "TileStateMut")
    PSBO, SBO = L.struct(c, "PlaneSuperBlockOffset"), L.struct(c, "SuperBlockOffset")
    TBO, BO = L.struct(c, "TileBlockOffset"), L.struct(c, "BlockOffset")
    PlaneOffset = R.PlaneOffset
    qc_default = c.get("default", owner="QuantizationContext")
    qc_update = c.get("update", owner="QuantizationContext")
    etb = c.get("encode_tx_block")
    cdist = c.get("compute_distortion")
    used = c.get("r1_probe_tx_used")
    NEWMV = L.enum(c, "PredictionMode", "NEWMV")
    RDO_PIX = L.enum(c, "RDOType", "PixelDistRealRate")
    IP_NONE = L.enum(c, "IntraParam", "None")
    rng = np.random.default_rng(20260929)
    out = {}

    # ---------------- part 1: the loop's type filter
    rav1e_types = [c.get("r1_probe_rav1e_type")({}, i) for i in range(int(c.get("r1_probe_rav1e_len")({})))]
    rav1e_ids = [int(t.disc) for t in rav1e_types]
    out["rav1e_tx_types"] = np.array(rav1e_ids, np.int32)
    masks = np.zeros((19, 2, 2, 2), np.uint32)       # [tx_size][is_inter][reduced][0: RAV1E_TX_TYPES only, 1: all 16]
    for ts in range(19):
        for inter in (0, 1):
            for red in (0, 1):
                for t in range(16):
                    if used({}, TxSize[ts], bool(inter), bool(red), TxType[t]):
                        masks[ts, inter, red, 1] |= 1 << t
                        if t in rav1e_ids:
                            masks[ts, inter, red, 0] |= 1 << t
    out["ts_mask"] = masks
    print("type filter:", [hex(int(masks[ts, 0, 0, 0])) for ts in range(19)], flush=True)

    # ---------------- part 2: every visited type on one (source, prediction)
    fw, fh = 102, 78
    imp_w, imp_h = (fw + 7) // 8, (fh + 7) // 8

    def tile_state(planes_in, planes_rec, qc, w=fw, h=fh):
        inp = Obj(planes=R.RSlice(planes_in))
        return make_struct(
            TileStateMut, sbo=PSBO(SBO(x=0, y=0)), sb_size_log2=6, sb_width=(w + 63) // 64,
            sb_height=(h + 63) // 64, mi_width=(w + 3) // 4, mi_height=(h + 3) // 4, width=w, height=h,
            input=inp, input_tile=Obj(planes=R.RSlice([p.as_region() for p in planes_in])),
            rec=Obj(planes=R.RSlice([p.as_region() for p in planes_rec])), qc=qc)

    def frame_invariants(bd, qidx, tune, scales, w=fw, h=fh, cs="Cs420", dist_scale=(1 << 14,) * 3, iw=imp_w, ih=imp_h):
        cfd = R.NONE
        if scales is not None:
            cfd = R.Some(Obj(distortion_scales=R.RSlice([DS(int(v)) for v in scales.ravel()]), w_in_imp_b=iw,
                             h_in_imp_b=ih))
        return Obj(sequence=Obj(bit_depth=bd, enable_intra_edge_filter=True,
                                chroma_sampling=L.enum(c, "ChromaSampling", cs)),
                   width=w, height=h, w_in_b=(w + 3) // 4, h_in_b=(h + 3) // 4,
                   use_tx_domain_distortion=False, base_q_idx=qidx,
                   dc_delta_q=R.RSlice([0, 0, 0]), ac_delta_q=R.RSlice([0, 0, 0]),
                   dist_scale=R.RSlice([DS(int(v)) for v in dist_scale]),
                   config=Obj(temporal_rdo=(lambda: scales is not None), tune=L.enum(c, "Tune", tune)),
                   coded_frame_data=cfd, cpu_feature_level=None, use_reduced_tx_set=False)

    # block origins in 4-pixel units: interior ones that move the 8x8 grid phase, and ones whose
    # block crosses the right / bottom / both frame edges (frame 102 x 78)
    def positions(w, h):
        inner = [(2, 3), (5, 4), (8, 1)]
        bx_r = (fw - w // 2 - 1) // 4        # right edge cuts the block roughly in half (visible_w % 4 == 2)
        by_b = (fh - h // 2 - 1) // 4
        edge = [(bx_r, 2), (3, by_b), (bx_r, by_b)]
        return inner, edge

    keys = []
    SIZES = (0, 1, 2, 3, 5, 6, 7, 8, 9, 10, 13, 14, 15, 16)
    ci = 0
    for bd in (8, 10, 12):
        g = dict(L.pixel_type(bd), W="BitCounter")
        g1 = L.pixel_type(bd)
        dt = L.np_dtype(bd)
        mx = (1 << bd) - 1
        H, W = fh + 2 * PAD, fw + 2 * PAD
        yy, xx = np.mgrid[0:H, 0:W]
        src = mx * (0.5 + 0.3 * np.sin(xx / 5.0) * np.cos(yy / 3.5)) + rng.integers(-(8 << (bd - 8)), (8 << (bd - 8)) + 1, (H, W))
        src = np.clip(src, 0, mx)
        src[PAD + 12:PAD + 20, PAD + 8:PAD + 40] = rng.choice([0, mx], (8, 32))     # a saturated patch
        # prediction error that grows left to right: small residuals (eob 0 / few coefficients) to large ones
        amp = (2 + (xx * 28) // W) << (bd - 8)
        pred = np.clip(src + rng.integers(-1000, 1001, (H, W)) * amp // 1000 +
                       (rng.random((H, W)) < 0.004) * rng.integers(-mx // 2, mx // 2, (H, W)), 0, mx)
        src, pred = src.astype(dt), pred.astype(dt)
        # the frame's own scale grid (what the reference indexes) inside a larger one that also covers
        # the invisible part of edge blocks (which the fused kernel's whole-block distortion touches)
        gh, gw = (H - PAD + 7) // 8, (W - PAD + 7) // 8
        big = rng.integers(1 << 12, 1 << 16, (gh, gw)).astype(np.uint32)
        scales = np.ascontiguousarray(big[:imp_h, :imp_w])
        out["tsr_src_%d" % bd], out["tsr_pred_%d" % bd], out["tsr_scales_%d" % bd] = src, pred, big
        for ts in SIZES:
            w, h = TX_W[ts], TX_H[ts]
            bsize = BlockSize["BLOCK_%dX%d" % (w, h)]
            inner, edge = positions(w, h)
            for inter in (0, 1):
                types = [t for t in rav1e_ids if (int(masks[ts, inter, 0, 0]) >> t) & 1]
                # two interior and two edge positions per (size, inter, bd), rotating through the lists
                for (bx, by), clipped in ((inner[ci % 3], False), (inner[(ci + 1) % 3], False), (edge[ci % 3], True),
                                          (edge[(ci + 2) % 3], True)):
                    qidx = (40, 100, 170, 230)[ci % 4] if w * h <= 256 else (80, 160)[ci % 2]
                    k = "%d_%d_%d_%d_%d_%d" % (bd, ts, inter, qidx, bx, by)
                    keys.append(k)
                    ox, oy = bx * 4, by * 4
                    eobs, qcs, recs, dists = [], [], [], []
                    for tt in types:
                        p_in, p_rec = padded_to_plane(src, bd, PAD), padded_to_plane(pred, bd, PAD)
                        qc = qc_default({})
                        qc_update({}, qc, qidx, TxSize[ts], not inter, bd, 0, 0)
                        tsm = tile_state([p_in], [p_rec], qc)
                        fi = frame_invariants(bd, qidx, "Psnr", None)
                        wr, cw = BitCounter(), CoeffRecorder()
                        bo = TBO(BO(x=bx, y=by))
                        # an intra block would predict first (encoder.rs:1452-1475); the prediction is
                        # already in `rec`, so the inter arm is taken for both and is_intra reaches the
                        # quantizer through qc_update above
                        has_coeff, d0 = etb(g, fi, tsm, cw, wr, 0, bo, 0, 0, bo, NEWMV, TxSize[ts], TxType[tt], bsize,
                                            PlaneOffset(x=ox, y=oy), False, qidx, R.RSlice([]), IP_NONE, RDO_PIX,
                                            False)
                        assert has_coeff is True and d0._0 == 0 and len(cw.calls) == 1 and not wr.bits
                        qcf, eob, cw_, ch_ = cw.calls[0]
                        eobs.append(eob)
                        qcs.append(np.array(qcf, np.int32))
                        recs.append(plane_block(p_rec, ox, oy, w, h, dt))
                        dd = []
                        for (tune, sc) in (("Psnr", None), ("Psychovisual", None), ("Psnr", scales),
                                           ("Psychovisual", scales)):
                            fi = frame_invariants(bd, qidx, tune, sc)
                            dd.append(cdist(g1, fi, tsm, bsize, False, bo, True)._0)
                        dists.append(dd)
                    vw = w if ox + w <= fw else fw - ox
                    vh = h if oy + h <= fh else fh - oy
                    out["tsr_types_" + k] = np.array(types, np.int32)
                    out["tsr_eob_" + k] = np.array(eobs, np.int32)
                    out["tsr_qc_" + k] = np.stack(qcs)
                    out["tsr_rec_" + k] = np.stack(recs)
                    out["tsr_dist_" + k] = np.array(dists, np.uint64)     # [type][sse, cdef, sse scaled, cdef scaled]
                    out["tsr_vis_" + k] = np.array([vw, vh], np.int32)
                    print(len(keys), k, "types", types, "vis", (vw, vh), "eob", eobs, flush=True)
                    ci += 1
    out["tsr_keys"] = np.array(keys)
    out["tsr_frame"] = np.array([fw, fh, PAD], np.int32)
    print("type search:", len(keys), "cases", flush=True)

    # ---------------- part 4: the next transform depth of an inter block (write_tx_tree executed whole)
    wtt = c.get("write_tx_tree")

    class Blocks:            # cw.bc.blocks[tile_bo].segmentation_idx (get_qidx, encoder.rs:1383-1394)
        def __getitem__(self, _k):
            return Obj(segmentation_idx=0)

    class TreeRecorder(CoeffRecorder):
        def __init__(self):
            CoeffRecorder.__init__(self)
            self.bc = Obj(blocks=Blocks())
    seg = Obj(features=R.RSlice([R.RSlice([False] * 8) for _ in range(8)]), data=R.RSlice([R.RSlice([0] * 8) for _ in range(8)]))
    # (block, transform size one depth down): BlockSize name, TxSize index
    SPLITS = (("BLOCK_16X16", 1), ("BLOCK_32X32", 2), ("BLOCK_8X8", 0), ("BLOCK_16X8", 1), ("BLOCK_8X16", 1), ("BLOCK_64X64", 3),
              ("BLOCK_32X16", 2), ("BLOCK_16X32", 2), ("BLOCK_8X4", 0), ("BLOCK_4X8", 0))
    keys4 = []
    ci = 0
    for bd in (8, 10, 12):
        g = dict(L.pixel_type(bd), W="BitCounter")
        g1 = L.pixel_type(bd)
        dt = L.np_dtype(bd)
        src, pred = out["tsr_src_%d" % bd], out["tsr_pred_%d" % bd]
        scales = np.ascontiguousarray(out["tsr_scales_%d" % bd][:imp_h, :imp_w])
        for (bsn, ts) in SPLITS:
            bw_, bh_ = [int(v) for v in bsn[6:].split("X")]
            tw_, th_ = TX_W[ts], TX_H[ts]
            types = [t for t in rav1e_ids if (int(masks[ts, 1, 0, 0]) >> t) & 1]
            inner, edge = positions(bw_, bh_)
            for (bx, by) in (inner[ci % 3], edge[ci % 3], edge[(ci + 1) % 3]):
                qidx = (40, 100, 170, 230)[ci % 4]
                k = "%d_%s_%d_%d_%d_%d" % (bd, bsn[6:], ts, qidx, bx, by)
                keys4.append(k)
                ox, oy = bx * 4, by * 4
                eobs, qcs, recs, dists = [], [], [], []
                for tt in types:
                    p_in, p_rec = padded_to_plane(src, bd, PAD), padded_to_plane(pred, bd, PAD)
                    qc = qc_default({})
                    tsm = tile_state([p_in, p_in, p_in], [p_rec, p_rec, p_rec], qc)     # planes[1].cfg is read for (xdec, ydec)
                    tsm.segmentation = seg
                    fi = frame_invariants(bd, qidx, "Psnr", None)
                    wr, cw = BitCounter(), TreeRecorder()
                    bo = TBO(BO(x=bx, y=by))
                    has_coeff, d0 = wtt(g, fi, tsm, cw, wr, NEWMV, 0, bo, BlockSize[bsn], TxSize[ts], TxType[tt], False, True,
                                        RDO_PIX, False)
                    assert d0._0 == 0 and not wr.bits
                    eobs.append([cl[1] for cl in cw.calls])
                    qcs.append(np.array([cl[0] for cl in cw.calls], np.int32))
                    recs.append(plane_block(p_rec, ox, oy, bw_, bh_, dt))
                    dd = []
                    for (tune, sc) in (("Psnr", None), ("Psychovisual", None), ("Psnr", scales), ("Psychovisual", scales)):
                        fi = frame_invariants(bd, qidx, tune, sc)
                        dd.append(cdist(g1, fi, tsm, BlockSize[bsn], False, bo, True)._0)
                    dists.append(dd)
                n_tx = len(eobs[0])
                assert all(len(e) == n_tx for e in eobs)
                out["txs_types_" + k] = np.array(types, np.int32)
                out["txs_eob_" + k] = np.array(eobs, np.int32)            # [type][transform block in call order]
                out["txs_qc_" + k] = np.stack(qcs)                         # [type][transform block][coded area]
                out["txs_rec_" + k] = np.stack(recs)                       # [type][bh][bw]
                out["txs_dist_" + k] = np.array(dists, np.uint64)
                out["txs_geom_" + k] = np.array([bw_, bh_, tw_, th_, n_tx], np.int32)
                print("split", len(keys4), k, "types", types, "tx blocks", n_tx, "eob", eobs[0], flush=True)
                ci += 1
    out["txs_keys"] = np.array(keys4)
    print("tx split:", len(keys4), "cases", flush=True)

    # ---------------- part 3: compute_distortion with chroma
    keys = []
    for (bd, cs, xdec, ydec) in ((8, "Cs420", 1, 1), (10, "Cs420", 1, 1), (12, "Cs420", 1, 1), (10, "Cs422", 1, 0),
                                 (8, "Cs444", 0, 0)):
        g1 = L.pixel_type(bd)
        dt = L.np_dtype(bd)
        mx = (1 << bd) - 1
        cw_, ch_ = 100, 76
        iw, ih = (cw_ + 7) // 8, (ch_ + 7) // 8
        planes_in, planes_rec = [], []
        k0 = "%d_%s" % (bd, cs)
        for pli in range(3):
            xd, yd = (xdec, ydec) if pli else (0, 0)
            pw, ph = (cw_ + xd) >> xd, (ch_ + yd) >> yd
            a = rng.integers(0, mx + 1, (ph, pw))
            b = np.clip(a + rng.integers(-9 << (bd - 8), (9 << (bd - 8)) + 1, (ph, pw)), 0, mx)
            a, b = a.astype(dt), b.astype(dt)
            planes_in.append(L.plane_from_array(a, bd, 16 >> xd, 16 >> yd, xdec=xd, ydec=yd))
            planes_rec.append(L.plane_from_array(b, bd, 16 >> xd, 16 >> yd, xdec=xd, ydec=yd))
            out["cd_src_%s_%d" % (k0, pli)], out["cd_rec_%s_%d" % (k0, pli)] = a, b
        tsm = tile_state(planes_in, planes_rec, None, cw_, ch_)
        ds3 = (1 << 14, 23000, 9000)
        scales = rng.integers(1 << 12, 1 << 16, (ih, iw)).astype(np.uint32)
        out["cd_scales_" + k0] = scales
        out["cd_dist_scale_" + k0] = np.array(ds3, np.uint32)
        out["cd_dec_" + k0] = np.array([xdec, ydec], np.int32)
        rows = []
        for (bs, bx, by) in (("BLOCK_8X8", 2, 2), ("BLOCK_16X16", 4, 8), ("BLOCK_32X32", 16, 8), ("BLOCK_32X32", 20, 14),
                             ("BLOCK_64X64", 16, 16), ("BLOCK_4X4", 1, 1), ("BLOCK_4X4", 3, 3), ("BLOCK_16X8", 22, 18),
                             ("BLOCK_8X16", 24, 16), ("BLOCK_4X8", 5, 3), ("BLOCK_8X4", 3, 5), ("BLOCK_64X64", 0, 0),
                             ("BLOCK_16X16", 23, 17)):
            for tune_i, tune in enumerate(("Psnr", "Psychovisual")):
                for use_scales in (0, 1):
                    for luma_only in (0, 1):
                        fi = frame_invariants(bd, 100, tune, scales if use_scales else None, cw_, ch_, cs, ds3, iw, ih)
                        d = cdist(g1, fi, tsm, BlockSize[bs], True, TBO(BO(x=bx, y=by)), bool(luma_only))
                        bsw, bsh = [int(v) for v in bs[6:].split("X")]
                        rows.append((bsw, bsh, bx, by, tune_i, use_scales, luma_only, d._0))
        out["cd_rows_" + k0] = np.array(rows, np.uint64)
        out["cd_frame_" + k0] = np.array([cw_, ch_], np.int32)
        keys.append(k0)
        print("compute_distortion (chroma)", k0, len(rows), "rows", flush=True)
    out["cd_keys"] = np.array(keys)
    L.save("rdo_txsearch_ref.npz", out)


if __name__ == "__main__":
    main()
