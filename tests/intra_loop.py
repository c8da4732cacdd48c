"""Closed-loop intra-only reconstruction of a 4:2:0 frame -- BASELINE.json
configs[0] ("tests/small_input.y4m speed-10 intra-only", the plumbing case) as
a composition of the hot-path entry points.  Test infrastructure.

What the loop follows in the reference, block by block in coding order:
  * speed 10 partitioning: every block is 32x32 luma + 16x16 chroma
    (PartitionRange 32..32, src/api/config/speedsettings.rs:184-190), one
    transform block per plane (bsize.tx_size(), rdo_tx_decision off), DCT_DCT;
  * intra_frame_rdo_mode_decision's pre-screen (src/rdo.rs:1434-1506):
    get_intra_edges(IntraParam::None) -> the 13 RAV1E_INTRA_MODES predicted
    from that edge set -> get_satd against the source;
  * encode_tx_block (src/encoder.rs:1434-1661): get_intra_edges for the chosen
    mode -> predict_intra -> diff -> forward_transform -> quantize ->
    dequantize -> inverse_transform_add into the reconstruction, which is what
    the next block's edges read;
  * the chroma planes use the luma mode as uv mode (always a member of the
    reference's uv mode set, src/rdo.rs:1520-1560) with 4:2:0 geometry.
What it leaves out (host-side serial state, out of scope in SURVEY.md 8): the
entropy coder, so the mode decision is "minimum SATD" instead of the
reference's rate-distortion cost over the three best SATD modes.  The loop is
therefore NOT an encoder; it is the data path an encoder drives, closed over
its own reconstruction, so that an error in any stage propagates into every
later block and shows up in the final planes.

Two backends with the same interface: OracleBackend (CPU restatement, ctypes)
and DeviceBackend (the product's C ABI through rav1e_amd.api).  run_frame()
returns the per-block records and the reconstructed planes.
"""
import ctypes as C

import numpy as np

import oracle_lib as O

TX_32X32, TX_16X16 = 3, 2
BASE_ANGLE = [0, 90, 180, 45, 135, 113, 157, 203, 67, 0, 0, 0, 0, 0]
SMOOTH_MODES = (9, 10, 11)
LUMA_PAD, CHROMA_PAD = 88, 44    # src/frame/mod.rs:22-23, >> chroma decimation


def variant_of(x, y):
    """PredictionVariant::new(x, y) (src/predict.rs:212-218): 0 NONE 1 LEFT 2 TOP 3 BOTH"""
    return 0 if (x == 0 and y == 0) else 1 if y == 0 else 2 if x == 0 else 3


def remap_mode(mode, variant):
    """PredictionMode::predict_intra's PAETH fallbacks (src/predict.rs:116-140)"""
    if mode == 12:
        return {0: 0, 2: 1, 1: 2}.get(variant, 12)
    return mode


def block_geometry(bx, by, fw, fh, bs):
    """(has_top_right, has_bottom_left) for square blocks of half the superblock
    size coded in z-order inside raster-ordered 64x64 superblocks (the outcome of
    has_top_right / has_bottom_left, src/partition.rs:400-560, for this regular
    partition): the top-right block exists and was coded earlier unless the block
    is the bottom-right quadrant of its superblock; the bottom-left block was
    coded earlier only for the top-left quadrant (it belongs to the superblock on
    the left)."""
    qx, qy = (bx // bs) % 2, (by // bs) % 2
    has_tr = by > 0 and bx + bs < fw and not (qx == 1 and qy == 1)
    has_bl = bx > 0 and by + bs < fh and qx == 0 and qy == 0
    return bool(has_tr), bool(has_bl)


class OracleBackend:
    def __init__(self, lib, bit_depth=8):
        self.L, self.bd = lib, bit_depth
        self.hbd = int(bit_depth > 8)
        self.dt = np.uint16 if self.hbd else np.uint8
        self.ct = np.int32 if self.hbd else np.int16

    def make_plane(self, img, pad):
        return O.plane_from_image(img, self.bd, pad, pad)

    def blank_plane(self, w, h, pad):
        return O.HostPlane(w, h, self.bd, pad, pad)

    def edges(self, rec, x, y, ts, mode, flags):
        e = np.zeros(257, self.dt)
        li = (C.c_int * 2)()
        self.L.r1o_get_intra_edges(O.ptr(e), li, rec.block_ptr(0, 0), rec.stride, x, y, rec.width,
                                   rec.height, ts, self.bd, mode, flags & 1, 0, (flags >> 1) & 1,
                                   (flags >> 2) & 1, self.hbd)
        return e, (li[0], li[1])

    def predict(self, ts, w, h, mode, variant, angle, ief, edge, lens, aw, ah):
        out = np.zeros((h, w), self.dt)
        assert self.L.r1o_dispatch_predict_intra(mode, variant, O.ptr(out), w, ts, self.bd, None,
                                                 angle, ief, O.ptr(edge), lens[0], lens[1], aw, ah,
                                                 self.hbd) == 0
        return out

    def satd13(self, src, x, y, ts, w, h, cands, edge, lens):
        out = np.zeros(len(cands), np.uint32)
        for k, (mode, variant, angle, ief) in enumerate(cands):
            p = self.predict(ts, w, h, mode, variant, angle, ief, edge, lens, w, h)
            out[k] = self.L.r1o_get_satd(src.block_ptr(x, y), src.stride, O.ptr(p), w, w, h, self.hbd)
        return out

    def select(self, satds):
        return int(np.argmin(satds))

    def code_block(self, src, x, y, ts, w, h, pred, qindex, kind, xdec, ydec):
        c = np.zeros(1, O.RDO_CAND)
        c["ox"], c["oy"] = x, y
        ps = src.cstruct()
        sad, satd = np.zeros(1, np.uint32), np.zeros(1, np.uint32)
        eob, dist = np.zeros(1, np.uint16), np.zeros(1, np.uint64)
        q = np.zeros((1, min(w, 32) * min(h, 32)), self.ct)
        rec = np.zeros((1, h, w), self.dt)
        pr = np.ascontiguousarray(pred[None])
        assert self.L.r1o_rdo_pixel_cand_batch(
            C.byref(ps), None, w, h, ts, O.ptr(c), 1, qindex, 1, 0, 0, kind, None, 0, xdec, ydec,
            O.ptr(sad), O.ptr(satd), O.ptr(eob), O.ptr(dist), O.ptr(q), O.ptr(rec), O.ptr(pr)) == 0
        return dict(sad=int(sad[0]), satd=int(satd[0]), eob=int(eob[0]), dist=int(dist[0]),
                    qcoeffs=q[0].astype(np.int64), rec=rec[0])

    def store(self, plane, x, y, block):
        plane.view()[y:y + block.shape[0], x:x + block.shape[1]] = block

    def pixels(self, plane):
        return plane.view().copy()


class DeviceBackend:
    """Same interface over the product's C ABI; planes are rav1e_amd.api.Plane in HBM."""

    def __init__(self, ctx, bit_depth=8):
        import torch
        from rav1e_amd import api
        self.ctx, self.bd, self.api, self.torch = ctx, bit_depth, api, torch
        self.hbd = int(bit_depth > 8)
        self.dt = np.uint16 if self.hbd else np.uint8

    def make_plane(self, img, pad):
        hp = O.plane_from_image(img, self.bd, pad, pad)
        return self.api.Plane.from_numpy(hp.data, hp.width, hp.height, self.bd, pad, pad)

    def blank_plane(self, w, h, pad):
        return self.api.Plane(w, h, self.bd, pad, pad)

    def edges(self, rec, x, y, ts, mode, flags):
        ec = np.zeros(1, self.api.INTRA_EDGE_CAND)
        ec["x"], ec["y"], ec["mode"], ec["flags"] = x, y, mode, flags
        return self.ctx.intra_edges_batch(rec, (0, 0, rec.width, rec.height), ts, ec)

    def _cands(self, rows, aw, ah):
        ic = np.zeros(len(rows), self.api.INTRA_CAND)
        for k, (mode, variant, angle, ief) in enumerate(rows):
            ic[k] = (mode, variant, angle, ief, aw, ah, 0)
        return ic

    def predict(self, ts, w, h, mode, variant, angle, ief, edge, lens, aw, ah):
        return self.ctx.predict_intra_batch(ts, self._cands([(mode, variant, angle, ief)], aw, ah),
                                            edge, lens, self.bd)[0]

    def satd13(self, src, x, y, ts, w, h, cands, edge, lens):
        pos = self.torch.tensor([[x, y]], dtype=self.torch.int16, device="cuda")
        return self.ctx.intra_satd_batch(src, ts, self._cands(cands, w, h), len(cands), pos, edge,
                                         lens).cpu().numpy().view(np.uint32)

    def select(self, satds):
        keys = self.torch.from_numpy(np.ascontiguousarray(satds).view(np.int32)).cuda()
        return int(self.ctx.prescreen_select_batch(keys, len(satds), 0, 1).cpu().numpy()[0, 0])

    def code_block(self, src, x, y, ts, w, h, pred, qindex, kind, xdec, ydec):
        c = np.zeros(1, self.api.RDO_CAND)
        c["ox"], c["oy"] = x, y
        o = self.ctx.rdo_pixel_cand_batch(src, None, w, h, c, qindex, kind, xdec=xdec, ydec=ydec,
                                          is_intra=1, want_qcoeffs=True, want_rec=True,
                                          pred=pred.reshape(1, h, w).contiguous())
        return dict(sad=int(o["sad"].cpu().numpy().view(np.uint32)[0]),
                    satd=int(o["satd"].cpu().numpy().view(np.uint32)[0]),
                    eob=int(o["eob"].cpu().numpy().view(np.uint16)[0]),
                    dist=int(o["dist"].cpu().numpy().view(np.uint64)[0]),
                    qcoeffs=o["qcoeffs"][0].cpu().numpy().astype(np.int64), rec=o["rec"][0])

    def store(self, plane, x, y, block):
        # device-to-device copy of the reconstruction into the plane the next block predicts from
        h, w = block.shape
        plane.data[plane.yorigin + y:plane.yorigin + y + h,
                   plane.xorigin + x:plane.xorigin + x + w] = block

    def pixels(self, plane):
        a = plane.data[plane.yorigin:plane.yorigin + plane.height,
                       plane.xorigin:plane.xorigin + plane.width].cpu().numpy()
        return a.view(self.dt).copy()


def _to_host(block, dt):
    return block if isinstance(block, np.ndarray) else block.cpu().numpy().view(dt)


def run_frame(be, y, u, v, qindex=100):
    """-> (records, (rec_y, rec_u, rec_v)) for one 4:2:0 frame of 32x32 blocks"""
    records, planes, _, _ = _run_frame(be, y, u, v, qindex)
    return records, planes


def _run_frame(be, y, u, v, qindex):
    fh, fw = y.shape
    assert fw % 32 == 0 and fh % 32 == 0
    src = [be.make_plane(y, LUMA_PAD), be.make_plane(u, CHROMA_PAD), be.make_plane(v, CHROMA_PAD)]
    rec = [be.blank_plane(fw, fh, LUMA_PAD), be.blank_plane(fw // 2, fh // 2, CHROMA_PAD),
           be.blank_plane(fw // 2, fh // 2, CHROMA_PAD)]
    modes = {}                       # (bx, by) -> luma mode, for the edge-filter neighbour test
    records = []
    for by in range(0, fh, 32):
        for bx in range(0, fw, 32):
            has_tr, has_bl = block_geometry(bx, by, fw, fh, 32)
            flags = 1 | (int(has_tr) << 1) | (int(has_bl) << 2)
            var = variant_of(bx, by)
            smooth = (modes.get((bx, by - 32)) in SMOOTH_MODES) or (modes.get((bx - 32, by)) in SMOOTH_MODES)
            ief = 2 if smooth else 1
            # ---- pre-screen: 13 modes from one edge set
            e0, l0 = be.edges(rec[0], bx, by, TX_32X32, -1, flags)
            cands = []
            for m in range(13):
                pm = remap_mode(m, var)
                cands.append((pm, var, BASE_ANGLE[pm], ief if 1 <= pm <= 8 else 0))
            satds = be.satd13(src[0], bx, by, TX_32X32, 32, 32, cands, e0, l0)
            best = be.select(satds)                # first minimum = head of the stable sort of rdo.rs:1504
            modes[(bx, by)] = best
            r = dict(bx=bx, by=by, satds=np.asarray(satds).copy(), mode=best, planes=[])
            # ---- encode_tx_block per plane
            for pli in range(3):
                dec = 0 if pli == 0 else 1
                x, yy, bs = bx >> dec, by >> dec, 32 >> dec
                ts = TX_32X32 if pli == 0 else TX_16X16
                pm = remap_mode(best, var)
                e, ln = be.edges(rec[pli], x, yy, ts, best, flags)
                pred = be.predict(ts, bs, bs, pm, var, BASE_ANGLE[pm], ief if 1 <= pm <= 8 else 0, e,
                                  ln, bs, bs)
                out = be.code_block(src[pli], x, yy, ts, bs, bs, pred, qindex, 3 if pli == 0 else 2,
                                    dec, dec)
                be.store(rec[pli], x, yy, out["rec"])
                out["rec"] = _to_host(out["rec"], be.dt).copy()
                out["pred"] = _to_host(pred, be.dt).copy()
                r["planes"].append(out)
            records.append(r)
    return records, tuple(be.pixels(p) for p in rec), rec, src


def psnr(a, b, peak=255.0):
    mse = np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)
    return 99.0 if mse == 0 else 10.0 * np.log10(peak * peak / mse)


# ------------------------------------------------------------------ post filters
def deblock_blocks(records, fw, fh):
    """R1DeblockBlock entries (per 4x4) of the regular 32x32 partition: tx 32x32 luma,
    16x16 chroma, intra, skip = every plane's eob is 0 (include/rav1e_amd.h)."""
    import deblock_util as D
    b = np.zeros((fh // 4, fw // 4), D.DEBLOCK_BLOCK)
    for r in records:
        skip = int(all(p["eob"] == 0 for p in r["planes"]))
        v = b[r["by"] // 4:r["by"] // 4 + 8, r["bx"] // 4:r["bx"] // 4 + 8]
        v["tx_log2"], v["uvtx_log2"], v["n4_log2"] = 3 | (3 << 3), 2 | (2 << 3), 3 | (3 << 3)
        v["flags"] = skip | 2
    return b


def _oracle_post(be, rec, src, blocks, fw, fh, cdef):
    import deblock_util as D
    L = be.L
    levels = [0, 0, 0, 0]
    for pli in range(3):
        dec = int(pli > 0)
        tv, th = np.zeros(65, np.int64), np.zeros(65, np.int64)
        pc, sc = rec[pli].cstruct(), src[pli].cstruct()
        assert L.r1o_deblock_sse_plane(C.byref(pc), C.byref(sc), pli, dec, dec, blocks.ctypes.data,
                                       blocks.shape[1], blocks.shape[1], blocks.shape[0], fw, fh, be.bd,
                                       tv.ctypes.data, th.ctypes.data) == 0
        lv = np.zeros(2, np.uint8)
        L.r1o_deblock_pick_levels(tv.ctypes.data, th.ctypes.data, pli, lv.ctypes.data)
        if pli == 0:
            levels[0], levels[1] = int(lv[0]), int(lv[1])
        else:
            levels[1 + pli] = int(lv[0])
    state = D.make_state(levels)
    for pli in range(3):
        dec = int(pli > 0)
        pc = rec[pli].cstruct()
        assert L.r1o_deblock_plane(state.ctypes.data, C.byref(pc), pli, dec, dec, blocks.ctypes.data,
                                   blocks.shape[1], blocks.shape[1], blocks.shape[0], fw, fh, be.bd) == 0
    deblocked = [p.view().copy() for p in rec]
    skip = np.ascontiguousarray(blocks["flags"] & 1).astype(np.uint8)
    ci = np.zeros(((fh + 63) // 64, (fw + 63) // 64), np.uint8)
    ystr, uvstr = np.array(cdef["y"], np.uint8), np.array(cdef["uv"], np.uint8)
    out = []
    for pli in range(3):
        dec = int(pli > 0)
        dst = be.blank_plane(fw >> dec, fh >> dec, rec[pli].xpad)
        a, b, c = rec[0].cstruct(), rec[pli].cstruct(), dst.cstruct()
        L.r1o_cdef_filter_tile_plane(C.byref(a), C.byref(b), C.byref(c), pli, dec, dec, fw, fh,
                                     O.ptr(skip), skip.shape[1], skip.shape[1], skip.shape[0], O.ptr(ci),
                                     ci.shape[1], O.ptr(ystr), O.ptr(uvstr), cdef["damping"], be.bd)
        out.append(dst.view().copy())
    return levels, deblocked, out


def _device_post(be, rec, src, blocks, fw, fh, cdef):
    import deblock_util as D
    torch, ctx = be.torch, be.ctx
    dblocks = torch.from_numpy(blocks.view(np.uint8).reshape(blocks.shape + (8,)).copy()).cuda()
    levels = [0, 0, 0, 0]
    tallies = ctx.deblock_sse_frame(rec, src, 1, 1, dblocks, fw, fh)      # six tallies, one launch
    for pli in range(3):
        dec = int(pli > 0)
        t = ctx.deblock_sse_plane(rec[pli], src[pli], pli, dec, dec, dblocks, fw, fh)
        assert torch.equal(t, tallies[pli])                                 # = the per-plane entry point
        lv = ctx.deblock_pick_levels(t, pli)
        if pli == 0:
            levels[0], levels[1] = int(lv[0]), int(lv[1])
        else:
            levels[1 + pli] = int(lv[0])
    state = D.make_state(levels)
    ctx.deblock_frame(state, rec, 1, 1, dblocks, fw, fh)                    # three planes, two launches
    deblocked = [be.pixels(p) for p in rec]
    skip = torch.from_numpy(np.ascontiguousarray(blocks["flags"] & 1).astype(np.uint8)).cuda()
    ci = torch.zeros(((fh + 63) // 64, (fw + 63) // 64), dtype=torch.uint8, device="cuda")
    out = []
    for pli in range(3):
        dec = int(pli > 0)
        dst = be.blank_plane(fw >> dec, fh >> dec, rec[pli].xpad)
        ctx.cdef_filter_frame_plane(rec[0], rec[pli], dst, pli, dec, dec, fw, fh, skip, ci, cdef["y"],
                                    cdef["uv"], cdef["damping"], be.bd)
        out.append(be.pixels(dst))
    return levels, deblocked, out


CDEF_FIXED = {"y": [2 * 4 + 1] + [0] * 7, "uv": [1 * 4 + 1] + [0] * 7}


def run_frame_with_filters(be, y, u, v, qindex=100):
    """run_frame, then the post-filter chain on the reconstruction (src/encoder.rs:3263-3322):
    deblock level search (sse_optimize) -> deblock in place -> CDEF.  The CDEF strengths are fixed
    (their search is rdo_loop_decision, host side); damping = 3 + (qindex >> 6) as
    FrameInvariants::cdef_damping.  -> dict"""
    fh, fw = y.shape
    records, planes, rec, src = _run_frame(be, y, u, v, qindex)
    blocks = deblock_blocks(records, fw, fh)
    cdef = dict(CDEF_FIXED, damping=3 + (qindex >> 6))
    post = _oracle_post if isinstance(be, OracleBackend) else _device_post
    levels, deblocked, cdeffed = post(be, rec, src, blocks, fw, fh, cdef)
    return dict(records=records, rec=planes, levels=levels, deblocked=deblocked, cdef=cdeffed)
