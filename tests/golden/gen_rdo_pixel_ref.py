#!/usr/bin/env python3
"""tests/golden/rdo_pixel_ref.npz: the PIXEL-DOMAIN leg of the RDO candidate (rav1e's default
configuration; BASELINE config 4) computed by the REFERENCE'S OWN SOURCE TEXT through tools/rustlite:

  encode_tx_block, RDOType::PixelDistRealRate, fi.use_tx_domain_distortion = false
                                          src/encoder.rs:1404-1661  (whole function: diff ->
      forward_transform -> QuantizationContext::quantize -> [cw.write_coeffs_lv_map: recorded, see
      below] -> dequantize -> inverse_transform_add INTO ts.rec (:1588-1614) -> ScaledDistortion::zero())
  compute_distortion                       src/rdo.rs:254-347        (whole function, luma_only:
      Tune::Psnr -> sse_wxh, Tune::Psychovisual -> cdef_dist_wxh, each with the distortion_scale
      closure; with and without temporal RDO (per-importance-block DistortionScale grid))
  and what they call: forward.rs, quantize/mod.rs, transform/inverse.rs:1633-1705 (the 2-D driver as
  written), dist.rs, rdo.rs distortion_scale / sse_wxh / cdef_dist_wxh.

What stands behind names the transpiler cannot expand:
  * get_func (impl_1d_tx! macro body, forward_shared.rs:201-218): as in gen_rdo_glue_ref.py.
  * `cw` (ContextWriter): the entropy coder is out of scope; a stand-in records the
    (qcoeffs, eob) that write_coeffs_lv_map is handed and returns true.  The recorded values ARE
    outputs of the executed text (QuantizationContext::quantize) and are stored as `px_qc_*`.
  * v_frame 0.3.9's ChromaSampling (reflib.V_FRAME_TEXT).
Hand-stated (plain data): FrameInvariants / Sequence / TileStateMut / CodedFrameData field values.

Run in the build container:  python tests/golden/gen_rdo_pixel_ref.py
"""
import numpy as np

import gen_fwd_tx_golden as FT
import reflib as L
from reflib import R
from gen_rdo_glue_ref import FILES, TX_W, TX_H, Obj, BitCounter, make_struct


class CoeffRecorder:
    """stand-in for ContextWriter on this path: write_coeffs_lv_map only"""

    def __init__(self):
        self.calls = []

    def write_coeffs_lv_map(self, w, p, tx_bo, qcoeffs, eob, mode, tx_size, tx_type, plane_bsize, xdec, ydec,
                            reduced, clip_w, clip_h):
        self.calls.append(([int(v) for v in qcoeffs], int(eob), int(clip_w), int(clip_h)))
        return True


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
    NEWMV = L.enum(c, "PredictionMode", "NEWMV")
    RDO_PIX = L.enum(c, "RDOType", "PixelDistRealRate")
    IP_NONE = L.enum(c, "IntraParam", "None")
    rng = np.random.default_rng(20260928)
    out = {}

    fw, fh = 128, 128
    imp_w, imp_h = fw // 8, fh // 8

    def tile_state(planes_in, planes_rec, qc):
        inp = Obj(planes=R.RSlice(planes_in))
        return make_struct(
            TileStateMut, sbo=PSBO(SBO(x=0, y=0)), sb_size_log2=6, sb_width=(fw + 63) // 64,
            sb_height=(fh + 63) // 64, mi_width=(fw + 3) // 4, mi_height=(fh + 3) // 4, width=fw, height=fh,
            input=inp, input_tile=Obj(planes=R.RSlice([p.as_region() for p in planes_in])),
            rec=Obj(planes=R.RSlice([p.as_region() for p in planes_rec])), qc=qc)

    def frame_invariants(bd, qidx, tune, scales):
        cfd = R.NONE
        if scales is not None:
            cfd = R.Some(Obj(distortion_scales=R.RSlice([DS(int(v)) for v in scales.ravel()]), w_in_imp_b=imp_w,
                             h_in_imp_b=imp_h))
        return Obj(sequence=Obj(bit_depth=bd, enable_intra_edge_filter=True,
                                chroma_sampling=L.enum(c, "ChromaSampling", "Cs420")),
                   width=fw, height=fh, w_in_b=(fw + 3) // 4, h_in_b=(fh + 3) // 4,
                   use_tx_domain_distortion=False, base_q_idx=qidx,
                   dc_delta_q=R.RSlice([0, 0, 0]), ac_delta_q=R.RSlice([0, 0, 0]),
                   dist_scale=R.RSlice([DS(1 << 14)] * 3),
                   config=Obj(temporal_rdo=(lambda: scales is not None), tune=L.enum(c, "Tune", tune)),
                   coded_frame_data=cfd, cpu_feature_level=None, use_reduced_tx_set=False)

    cases = []
    for bd in (8, 10):
        for ts in (0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 13, 14, 15, 16, 17, 18):
            m = max(TX_W[ts], TX_H[ts])
            types = [0, 1, 3, 6, 9, 10, 11, 15] if m <= 16 else ([0, 9] if m == 32 else [0])
            if TX_W[ts] * TX_H[ts] >= 1024:
                qs = (100,)
            elif TX_W[ts] * TX_H[ts] > 256:
                qs = (60, 180)
            else:
                qs = (20, 90, 200)
            for tt in types:
                for qidx in qs:
                    cases.append((bd, ts, tt, qidx))
    keys = []
    for ci, (bd, ts, tt, qidx) in enumerate(cases):
        g = dict(L.pixel_type(bd), W="BitCounter")
        g1 = L.pixel_type(bd)
        dt = L.np_dtype(bd)
        mx = (1 << bd) - 1
        w, h = TX_W[ts], TX_H[ts]
        bx, by = 8, 4                          # block at (32, 16) px
        # smooth-ish source with texture; prediction = source + noise + a few outliers, with a
        # saturated patch so that the reconstruction clamp is reached
        yy, xx = np.mgrid[0:fh, 0:fw]
        src = mx * (0.5 + 0.3 * np.sin(xx / (3.0 + ci % 7)) * np.cos(yy / (2.0 + ci % 5))) + \
            rng.integers(-(10 << (bd - 8)), (10 << (bd - 8)) + 1, (fh, fw))
        src = np.clip(src, 0, mx)
        if ci % 3 == 0:
            src[by * 4:by * 4 + h // 2, bx * 4:bx * 4 + w] = rng.choice([0, mx], (h // 2, w))
        amp = (3, 9, 30)[ci % 3] << (bd - 8)
        pred = np.clip(src + rng.integers(-amp, amp + 1, (fh, fw)) +
                       (rng.random((fh, fw)) < 0.04) * rng.integers(-mx // 2, mx // 2, (fh, fw)), 0, mx)
        src, pred = src.astype(dt), pred.astype(dt)
        bsize = BlockSize["BLOCK_%dX%d" % (w, h)]
        k = "%d_%d_%d_%d" % (bd, ts, tt, qidx)
        keys.append(k)
        out["px_src_" + k] = src[by * 4:by * 4 + h, bx * 4:bx * 4 + w]
        out["px_pred_" + k] = pred[by * 4:by * 4 + h, bx * 4:bx * 4 + w]
        scales = rng.integers(1 << 12, 1 << 16, (imp_h, imp_w)).astype(np.uint32)
        out["px_scales_" + k] = scales[by // 2:by // 2 + (h + 7) // 8, bx // 2:bx // 2 + (w + 7) // 8]
        dists = []
        rec_blk = None
        for (tune, sc) in (("Psnr", None), ("Psychovisual", None), ("Psnr", scales), ("Psychovisual", scales)):
            if tune == "Psychovisual" and (w < 8 or h < 8):
                # compute_distortion would still call cdef_dist_wxh (8x8 kernels clipped); keep it
                pass
            p_in, p_rec = L.plane_from_array(src, bd, 16, 16), L.plane_from_array(pred, bd, 16, 16)
            qc = qc_default({})
            qc_update({}, qc, qidx, TxSize[ts], False, bd, 0, 0)
            tsm = tile_state([p_in], [p_rec], qc)
            fi = frame_invariants(bd, qidx, tune, sc)
            wr, cw = BitCounter(), CoeffRecorder()
            bo = TBO(BO(x=bx, y=by))
            has_coeff, d0 = etb(g, fi, tsm, cw, wr, 0, bo, 0, 0, bo, NEWMV, TxSize[ts], TxType[tt], bsize,
                                PlaneOffset(x=bx * 4, y=by * 4), False, qidx, R.RSlice([]), IP_NONE, RDO_PIX, False)
            assert has_coeff is True and d0._0 == 0 and len(cw.calls) == 1 and not wr.bits
            rec = L.plane_to_array(p_rec, dt)
            blk = rec[by * 4:by * 4 + h, bx * 4:bx * 4 + w].copy()
            outside = rec.copy()
            outside[by * 4:by * 4 + h, bx * 4:bx * 4 + w] = pred[by * 4:by * 4 + h, bx * 4:bx * 4 + w]
            assert np.array_equal(outside, pred)          # only the block was reconstructed
            if rec_blk is None:
                rec_blk = blk
                qcf, eob, cw_, ch_ = cw.calls[0]
                out["px_qc_" + k] = np.array(qcf, np.int32)
                out["px_eob_" + k] = np.array([eob], np.int32)
                assert (cw_, ch_) == (w, h)
            else:
                assert np.array_equal(blk, rec_blk)
            d = cdist(g1, fi, tsm, bsize, False, bo, True)
            dists.append(d._0)
        out["px_rec_" + k] = rec_blk
        out["px_dist_" + k] = np.array(dists, np.uint64)   # [sse, cdef, sse scaled, cdef scaled]
        if ci % 40 == 0:
            print(ci, "/", len(cases), k, "eob", eob, dists, flush=True)
    out["px_keys"] = np.array(keys)
    print("pixel-domain encode_tx_block + compute_distortion:", len(keys), "cases")
    L.save("rdo_pixel_ref.npz", out)


if __name__ == "__main__":
    main()
