#!/usr/bin/env python3
"""tests/golden/rdo_glue_ref.npz: the glue of the full RDO candidate computed by the
REFERENCE'S OWN SOURCE TEXT, transpiled by tools/rustlite and executed here:

  estimate_rate                          src/rdo.rs:127-139           (whole function)
  encode_tx_block, RDOType::TxDistEstRate src/encoder.rs:1404-1661     (whole function: diff ->
      forward_transform -> QuantizationContext::quantize -> dequantize -> transform-domain
      distortion -> estimate_rate -> RawDistortion * bias * dist_scale; inter mode, so the
      prediction is what `rec` holds on entry; no entropy coder on this path)
  compute_tx_distortion                  src/rdo.rs:349-434           (whole function, skip and
      non-skip, luma only and with chroma, blocks cut by the frame edge)
  rdo_cfl_alpha                          src/rdo.rs:1593-1688         (whole function: luma_ac,
      get_intra_edges, 33 x UV_CFL_PRED + sse_wxh with the sequential selection and early exit)
  PredictionMode::predict_inter_compound src/predict.rs:339-382       (whole function: get_mv_params,
      prep_8tap x 2, mc_avg through InterCompoundBuffers)
  and what they call (forward.rs, quantize/mod.rs, dist.rs, predict.rs, partition.rs, mc.rs,
  tiling/*.rs, context/*.rs).

What stands behind names the transpiler cannot expand:
  * get_func (impl_1d_tx! macro body, forward_shared.rs:201-218): the 1-D networks of the same
    macro body as translated by gen_fwd_tx_golden.py (round 1); the 2-D driver forward_transform
    itself (forward.rs:71-161) is executed as written.
  * v_frame 0.3.9's ChromaSampling (reflib.V_FRAME_TEXT).
Hand-stated (plain data): FrameInvariants / Sequence / TileStateMut field values, a Writer that
records add_bits_frac, reference frames as planes.

Run in the build container:  python tests/golden/gen_rdo_glue_ref.py
"""
import numpy as np

import gen_fwd_tx_golden as FT
import reflib as L
from reflib import R

FILES = ["rdo.rs", "encoder.rs", "quantize/mod.rs", "quantize/tables.rs", "scan_order.rs", "transform/mod.rs",
         "transform/forward.rs", "transform/forward_shared.rs", "transform/inverse.rs", "dist.rs",
         "tiling/tile_state.rs", "tiling/tile.rs", "tiling/plane_region.rs", "context/mod.rs",
         "context/block_unit.rs", "predict.rs", "partition.rs", "util/mod.rs", "util/align.rs",
         "util/uninit.rs", "activity.rs", "mc.rs", "rdo_tables.rs", "frame/mod.rs"]
TX_W = [4, 8, 16, 32, 64, 4, 8, 8, 16, 16, 32, 32, 64, 4, 16, 8, 32, 16, 64]
TX_H = [4, 8, 16, 32, 64, 8, 4, 16, 8, 32, 16, 64, 32, 16, 4, 32, 8, 64, 16]


class Obj:
    def __init__(self, **kw):
        self.__dict__.update(kw)


class BitCounter:
    """a Writer that only records what encode_tx_block adds (w.add_bits_frac)"""

    def __init__(self):
        self.bits = []

    def add_bits_frac(self, v):
        self.bits.append(int(v))


def make_struct(cls, **kw):
    vals = {f: None for f in cls._fields}
    vals.update(kw)
    return cls(**vals)


def main():
    c = L.crate(*FILES)
    L.load_v_frame_types(c)
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

    G = c.G
    TxSize = [L.enum(c, "TxSize", v[0]) for v in c.enums["TxSize"].variants]
    TxType = [L.enum(c, "TxType", v[0]) for v in c.enums["TxType"].variants]
    BlockSize = {v[0]: L.enum(c, "BlockSize", v[0]) for v in c.enums["BlockSize"].variants}
    DS = L.struct(c, "DistortionScale")
    ScaledDistortion = L.struct(c, "ScaledDistortion")
    TileStateMut = L.struct(c, "TileStateMut")
    PSBO, SBO = L.struct(c, "PlaneSuperBlockOffset"), L.struct(c, "SuperBlockOffset")
    TBO, BO = L.struct(c, "TileBlockOffset"), L.struct(c, "BlockOffset")
    PlaneOffset = R.PlaneOffset
    qc_default = c.get("default", owner="QuantizationContext")
    qc_update = c.get("update", owner="QuantizationContext")
    rng = np.random.default_rng(20260927)
    out = {}

    # ---------------- estimate_rate
    er = c.get("estimate_rate")
    rows = []
    for _ in range(600):
        qi, ts = int(rng.integers(0, 256)), int(rng.integers(0, 19))
        d = int(rng.integers(0, 1 << int(rng.integers(1, 26))))
        rows.append((qi, ts, d, er({}, qi, TxSize[ts], d)))
    for qi in (0, 31, 32, 255):
        for ts in (0, 4, 18):
            for d in (0, 1, 4999, 5000, 5001, 49 * 5000 - 1, 49 * 5000, 50 * 5000, 10 ** 7):
                rows.append((qi, ts, d, er({}, qi, TxSize[ts], d)))
    out["rate"] = np.array(rows, np.uint64)

    def tile_state(bd, planes_in, planes_rec, fw, fh, qc=None):
        """TileStateMut over a whole small frame (one tile at the origin)"""
        inp = Obj(planes=R.RSlice(planes_in))
        return make_struct(
            TileStateMut, sbo=PSBO(SBO(x=0, y=0)), sb_size_log2=6, sb_width=(fw + 63) // 64,
            sb_height=(fh + 63) // 64, mi_width=(fw + 3) // 4, mi_height=(fh + 3) // 4, width=fw, height=fh,
            input=inp, input_tile=Obj(planes=R.RSlice([p.as_region() for p in planes_in])),
            rec=Obj(planes=R.RSlice([p.as_region() for p in planes_rec])), qc=qc)

    def frame_invariants(bd, fw, fh, qidx, tx_domain=True, cs="Cs420", dist_scale=(1 << 14,) * 3):
        return Obj(sequence=Obj(bit_depth=bd, enable_intra_edge_filter=True,
                                chroma_sampling=L.enum(c, "ChromaSampling", cs)),
                   width=fw, height=fh, w_in_b=(fw + 3) // 4, h_in_b=(fh + 3) // 4,
                   use_tx_domain_distortion=tx_domain, base_q_idx=qidx,
                   dc_delta_q=R.RSlice([0, 0, 0]), ac_delta_q=R.RSlice([0, 0, 0]),
                   dist_scale=R.RSlice([DS(int(v)) for v in dist_scale]),
                   config=Obj(temporal_rdo=lambda: False, tune=L.enum(c, "Tune", "Psnr")),
                   coded_frame_data=R.NONE, cpu_feature_level=None, use_reduced_tx_set=False)

    # ---------------- encode_tx_block, RDOType::TxDistEstRate
    etb = c.get("encode_tx_block")
    NEWMV = L.enum(c, "PredictionMode", "NEWMV")
    RDO_EST = L.enum(c, "RDOType", "TxDistEstRate")
    IP_NONE = G["_E"]("IntraParam", "None", 0, ()) if "IntraParam" in c.enums or c.autoload("IntraParam") else None
    IP_NONE = L.enum(c, "IntraParam", "None")
    keys = []
    cases = []
    for bd in (8, 10):
        for ts in (0, 1, 2, 3, 4, 6, 7, 10, 13, 18):
            types = [0, 1, 6, 9, 10, 15] if max(TX_W[ts], TX_H[ts]) <= 16 else ([0, 9] if max(TX_W[ts], TX_H[ts]) == 32 else [0])
            for tt in types:
                for qidx in ((35, 160) if TX_W[ts] * TX_H[ts] <= 256 else (100,)):
                    cases.append((bd, ts, tt, qidx))
    fw, fh = 128, 128
    for (bd, ts, tt, qidx) in cases:
        g = dict(L.pixel_type(bd), W="BitCounter")     # encode_tx_block<T: Pixel, W: Writer>
        dt = L.np_dtype(bd)
        mx = (1 << bd) - 1
        w, h = TX_W[ts], TX_H[ts]
        bx, by = 16, 8                       # block offset in 4x4 units: (64, 32) px
        src = rng.integers(0, mx + 1, (fh, fw))
        pred = np.clip(src + rng.integers(-(6 << (bd - 8)), (6 << (bd - 8)) + 1, (fh, fw)) +
                       (rng.random((fh, fw)) < 0.05) * rng.integers(-mx // 3, mx // 3, (fh, fw)), 0, mx)
        src, pred = src.astype(dt), pred.astype(dt)
        p_in, p_rec = L.plane_from_array(src, bd, 16, 16), L.plane_from_array(pred, bd, 16, 16)
        qc = qc_default({})
        qc_update({}, qc, qidx, TxSize[ts], False, bd, 0, 0)
        tsm = tile_state(bd, [p_in], [p_rec], fw, fh, qc)
        fi = frame_invariants(bd, fw, fh, qidx)
        wr = BitCounter()
        bo = TBO(BO(x=bx, y=by))
        bsize = BlockSize["BLOCK_64X64"]      # the partition; only its subsampled_size is read (p = 0)
        has_coeff, dist = etb(g, fi, tsm, None, wr, 0, bo, 0, 0, bo, NEWMV, TxSize[ts], TxType[tt], bsize,
                              PlaneOffset(x=bx * 4, y=by * 4), False, qidx, R.RSlice([]), IP_NONE, RDO_EST, False)
        k = "%d_%d_%d_%d" % (bd, ts, tt, qidx)
        keys.append(k)
        out["tb_src_" + k] = src[by * 4:by * 4 + h, bx * 4:bx * 4 + w]
        out["tb_pred_" + k] = pred[by * 4:by * 4 + h, bx * 4:bx * 4 + w]
        assert len(wr.bits) == 1 and has_coeff is True
        out["tb_out_" + k] = np.array([dist._0, wr.bits[0]], np.uint64)
        # the reconstruction is untouched on this path (no inverse transform)
        assert np.array_equal(L.plane_to_array(p_rec, dt), pred)
    out["tb_keys"] = np.array(keys)
    print("encode_tx_block:", len(keys), "cases", flush=True)

    # ---------------- compute_tx_distortion
    ctd = c.get("compute_tx_distortion")
    keys = []
    for bd in (8, 10):
        g = L.pixel_type(bd)
        dt = L.np_dtype(bd)
        mx = (1 << bd) - 1
        fw, fh = 100, 76                       # not multiples of 8: blocks get cut by the frame edge
        planes_in, planes_rec, imgs = [], [], []
        for pli in range(3):
            dec = 1 if pli else 0
            pw, ph = (fw + dec) >> dec, (fh + dec) >> dec
            a = rng.integers(0, mx + 1, (ph, pw))
            b = np.clip(a + rng.integers(-9 << (bd - 8), (9 << (bd - 8)) + 1, (ph, pw)), 0, mx)
            a, b = a.astype(dt), b.astype(dt)
            imgs.append((a, b))
            planes_in.append(L.plane_from_array(a, bd, 16 >> dec, 16 >> dec, xdec=dec, ydec=dec))
            planes_rec.append(L.plane_from_array(b, bd, 16 >> dec, 16 >> dec, xdec=dec, ydec=dec))
            out["td_src_%d_%d" % (bd, pli)], out["td_rec_%d_%d" % (bd, pli)] = a, b
        tsm = tile_state(bd, planes_in, planes_rec, fw, fh)
        scales = (1 << 14, 23000, 9000)
        fi = frame_invariants(bd, fw, fh, 100, dist_scale=scales)
        out["td_scales_%d" % bd] = np.array(scales, np.uint32)
        rows = []
        for (bs, bx, by) in (("BLOCK_8X8", 2, 2), ("BLOCK_16X16", 4, 8), ("BLOCK_32X32", 16, 8), ("BLOCK_32X32", 20, 14),
                             ("BLOCK_64X64", 16, 16), ("BLOCK_4X4", 1, 1), ("BLOCK_4X4", 3, 3), ("BLOCK_16X8", 22, 18),
                             ("BLOCK_8X16", 24, 16), ("BLOCK_4X8", 5, 3), ("BLOCK_64X64", 0, 0)):
            for skip in (True, False):
                for luma_only in (True, False):
                    for is_chroma in (True, False):
                        txd = int(rng.integers(0, 1 << 20))
                        d = ctd(g, fi, tsm, BlockSize[bs], is_chroma, TBO(BO(x=bx, y=by)), ScaledDistortion(txd),
                                skip, luma_only)
                        bsw, bsh = [int(v) for v in bs[6:].split("X")]
                        rows.append((bsw, bsh, bx, by, int(skip), int(luma_only), int(is_chroma), txd, d._0))
        out["td_rows_%d" % bd] = np.array(rows, np.uint64)
        keys.append(str(bd))
    out["td_keys"] = np.array(keys)
    print("compute_tx_distortion done", flush=True)

    # ---------------- rdo_cfl_alpha
    cfl = c.get("rdo_cfl_alpha")
    alpha_of = c.get("alpha", owner="CFLParams")
    keys = []
    for (bd, cs, xdec, ydec) in ((8, "Cs420", 1, 1), (10, "Cs420", 1, 1), (8, "Cs444", 0, 0), (10, "Cs422", 1, 0)):
        g = L.pixel_type(bd)
        dt = L.np_dtype(bd)
        mx = (1 << bd) - 1
        fw, fh = 96, 72
        yy, xx = np.mgrid[0:fh, 0:fw]
        luma = np.clip(mx * (0.5 + 0.35 * np.sin(xx / 6.0) * np.cos(yy / 5.0)) + rng.integers(-6, 7, (fh, fw)), 0, mx)
        planes_in, planes_rec = [], []
        imgs = []
        for pli in range(3):
            xd, yd = (xdec, ydec) if pli else (0, 0)
            pw, ph = (fw + xd) >> xd, (fh + yd) >> yd
            if pli == 0:
                a = luma
            else:   # chroma correlated with (subsampled) luma, so that a non-zero alpha wins
                sub = luma[::1 << yd, ::1 << xd][:ph, :pw]
                a = np.clip((0.5 * mx + (0.45 if pli == 1 else -0.3) * (sub - 0.5 * mx)) + rng.integers(-4, 5, (ph, pw)), 0, mx)
            rec = np.clip(a + rng.integers(-3, 4, a.shape), 0, mx)
            a, rec = a.astype(dt), rec.astype(dt)
            imgs.append((a, rec))
            planes_in.append(L.plane_from_array(a, bd, 16 >> xd, 16 >> yd, xdec=xd, ydec=yd))
            planes_rec.append(L.plane_from_array(rec, bd, 16 >> xd, 16 >> yd, xdec=xd, ydec=yd))
        k0 = "%d_%s" % (bd, cs)
        for pli in range(3):
            out["cfl_src_%s_%d" % (k0, pli)], out["cfl_rec_%s_%d" % (k0, pli)] = imgs[pli]
        fi = frame_invariants(bd, fw, fh, 100, cs=cs)
        rows = []
        for (bs, lts, bx, by) in (("BLOCK_16X16", 2, 4, 4), ("BLOCK_32X32", 3, 8, 8), ("BLOCK_8X8", 1, 6, 2),
                                   ("BLOCK_16X16", 2, 0, 0), ("BLOCK_32X32", 3, 16, 8), ("BLOCK_16X16", 2, 20, 14),
                                   ("BLOCK_8X8", 1, 22, 16), ("BLOCK_32X16", 10, 8, 4)):
            # the prediction loop writes into rec: fresh planes per block
            fresh = [L.plane_from_array(imgs[p][1], bd, 16 >> (xdec if p else 0), 16 >> (ydec if p else 0),
                                        xdec=xdec if p else 0, ydec=ydec if p else 0) for p in range(3)]
            tsm = tile_state(bd, planes_in, fresh, fw, fh)
            r = cfl(g, tsm, TBO(BO(x=bx, y=by)), BlockSize[bs], TxSize[lts], fi)
            if r.is_none():
                au, av = 0, 0
            else:
                p = r.unwrap()
                au, av = alpha_of({}, p, 0), alpha_of({}, p, 1)
            bsw, bsh = [int(v) for v in bs[6:].split("X")]
            rows.append((bsw, bsh, lts, bx, by, au, av))
        out["cfl_rows_" + k0] = np.array(rows, np.int32)
        out["cfl_dec_" + k0] = np.array([xdec, ydec], np.int32)
        keys.append(k0)
        print("rdo_cfl_alpha", k0, rows, flush=True)
    out["cfl_keys"] = np.array(keys)

    # ---------------- predict_inter_compound
    pic = c.get("predict_inter_compound", owner="PredictionMode")
    ICB = L.struct(c, "InterCompoundBuffers")
    MV = L.struct(c, "MotionVector")
    TileRect = L.struct(c, "TileRect")
    LAST, ALTREF = L.enum(c, "RefType", "LAST_FRAME"), L.enum(c, "RefType", "ALTREF_FRAME")
    keys = []
    for bd in (8, 10, 12):
        g = L.pixel_type(bd)
        dt = L.np_dtype(bd)
        mx = (1 << bd) - 1
        fw, fh = 96, 80
        refs = [rng.integers(0, mx + 1, (fh, fw)).astype(dt) for _ in range(2)]
        pads = [np.pad(r, 24, mode="edge") for r in refs]
        planes = [L.plane_from_padded(p, bd, 24, 24) for p in pads]
        out["pic_ref0_%d" % bd], out["pic_ref1_%d" % bd] = pads
        frames = [R.Some(Obj(frame=Obj(planes=R.RSlice([planes[0]])))), R.Some(Obj(frame=Obj(planes=R.RSlice([planes[1]]))))]
        frames += [R.NONE] * 6
        ref_frames = [0, 0, 0, 0, 0, 0, 1]         # LAST_FRAME -> slot 0, ALTREF_FRAME -> slot 1
        for filt in ("REGULAR", "SHARP"):
            fi = Obj(rec_buffer=Obj(frames=R.RSlice(frames)), ref_frames=R.RSlice(ref_frames),
                     default_filter=L.enum(c, "FilterMode", filt), sequence=Obj(bit_depth=bd), cpu_feature_level=None)
            rows, preds = [], []
            for (w, h) in ((8, 8), (16, 16), (32, 16), (4, 4), (64, 64), (16, 4), (8, 32)):
                for rep in range(3):
                    x, y = int(rng.integers(0, fw - w + 1)) & ~3, int(rng.integers(0, fh - h + 1)) & ~3
                    mvs = [(int(rng.integers(-60, 61)), int(rng.integers(-60, 61))) for _ in range(2)]
                    if rep == 0:
                        mvs[0] = (mvs[0][0] & ~7, mvs[0][1] & ~7)      # one full-pel reference
                    dst_img = np.zeros((fh, fw), dt)
                    dst = L.plane_from_array(dst_img, bd, 24, 24)
                    area = G["_E"]("Area", "Rect", 0, (x, y, w, h))
                    area = c.G["_E"]("Area", "StartingAt", 1, (x, y))
                    reg = dst.as_region().subregion(area)
                    buf = ICB(data=R.RSlice([0] * (2 * 128 * 128)))
                    pic(g, NEWMV, fi, TileRect(x=0, y=0, width=fw, height=fh), 0, PlaneOffset(x=x, y=y), reg, w, h,
                        R.RSlice([LAST, ALTREF]), R.RSlice([MV(row=mvs[0][0], col=mvs[0][1]), MV(row=mvs[1][0], col=mvs[1][1])]),
                        buf)
                    got = L.plane_to_array(dst, dt)[y:y + h, x:x + w]
                    rows.append((w, h, x, y, mvs[0][0], mvs[0][1], mvs[1][0], mvs[1][1], len(preds)))
                    preds.append(got.ravel())
            k = "%d_%s" % (bd, filt)
            out["pic_rows_" + k] = np.array(rows, np.int32)
            out["pic_pred_" + k] = np.concatenate(preds).astype(dt)
            keys.append(k)
    out["pic_keys"] = np.array(keys)
    L.save("rdo_glue_ref.npz", out)


if __name__ == "__main__":
    main()
