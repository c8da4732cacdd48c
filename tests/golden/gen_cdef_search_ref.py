#!/usr/bin/env python3
"""tests/golden/cdef_search_ref.npz: the CDEF strength search of rdo_loop_decision computed by
the REFERENCE'S OWN SOURCE TEXT, transpiled by tools/rustlite and executed here:

  cdef_analyze_superblock_range, cdef_analyze_superblock, cdef_filter_superblock and
  everything below them                                        src/cdef.rs:30-560
  rdo_loop_plane_error, distortion_scale, sse_wxh,
  compute_rd_cost, the Distortion types                        src/rdo.rs:142-224,443-459,558-723,2027-2093
  cdef_dist_kernel, get_weighted_sse                           src/dist.rs:234-372

Hand-stated: the control flow AROUND those calls -- the loop of rdo_loop_decision's CDEF leg
with RestorationFilter::None (src/rdo.rs:2366-2530: per superblock of the analysis area, skip
test rdo.rs:2196-2211, cdef_index 0 .. (1 << cdef_bits) - 1, filter into the working copy,
error of the three planes, rate 0, "first smallest cost wins"), the scratch copy of the area
(rdo.rs:2277-2284: planes without padding), and the encoder-state containers (FrameInvariants
fields, TileBlocks, the Tile / Frame wrappers), which are plain data here.
(Round 5: gen_loop_decision_ref.py executes rdo_loop_decision itself; its CDEF-only cases go through the same tests and
confirm this hand-stated loop.)

Keys per case <c>: <c>_meta = [W, H, xdec, ydec, bd, damping, n_idx, area_sb_w, area_sb_h, planes],
<c>_rec{0,1,2} / <c>_src{0,1,2} (whole-frame planes), <c>_skip (per 4x4), <c>_ystr / <c>_uvstr,
<c>_scales (per 8x8 luma block, Q14), <c>_dscale (fi.dist_scale), <c>_err ([sby][sbx][8], u64),
<c>_best ([sby][sbx], -1 = skipped superblock).

Run in the build container:  python tests/golden/gen_cdef_search_ref.py
"""
import time

import numpy as np

import reflib as L
from reflib import R


class Obj:
    def __init__(self, **kw):
        self.__dict__.update(kw)


class TileBlocks:
    """cols()/rows() in 4x4 units, [TileBlockOffset].skip -- a window of the frame's grid"""

    def __init__(self, skip, x0, y0, cols, rows):
        self.skip, self.x0, self.y0, self.c, self.r = skip, x0, y0, cols, rows

    def cols(self):
        return self.c

    def rows(self):
        return self.r

    def __getitem__(self, bo):
        b = bo._0
        if not (0 <= b.x < self.c and 0 <= b.y < self.r):
            raise R.Panic("TileBlocks index out of the subset")
        return Obj(skip=bool(self.skip[self.y0 + b.y, self.x0 + b.x]))


CASES = [
    # W, H, xdec, ydec, bd, n_idx, area (sb_w, sb_h), planes, p_skip
    (136, 72, 1, 1, 8, 8, (1, 1), 3, 0.35),
    (200, 136, 1, 1, 10, 8, (2, 2), 3, 0.3),
    (128, 64, 0, 0, 8, 4, (1, 1), 3, 0.5),
    (96, 80, 1, 0, 12, 2, (2, 1), 3, 0.2),
    (72, 136, 1, 1, 8, 8, (1, 2), 1, 0.4),
]


def main():
    c = L.crate("cdef.rs", "rdo.rs", "dist.rs", "activity.rs", "context/superblock_unit.rs",
                "context/block_unit.rs", "tiling/plane_region.rs")
    c.define_enum("ChromaSampling", ["Cs420", "Cs422", "Cs444", "Cs400"])
    analyze_range = c.get("cdef_analyze_superblock_range")
    filter_sb = c.get("cdef_filter_superblock")
    plane_error = c.get("rdo_loop_plane_error")
    rd_cost = c.get("compute_rd_cost")
    TSBO, SBO = c.G["S_TileSuperBlockOffset"], c.G["S_SuperBlockOffset"]
    PBO, BO = L.struct(c, "PlaneBlockOffset"), L.struct(c, "BlockOffset")
    DS = c.G["S_DistortionScale"]
    rng = np.random.default_rng(20260926)
    out = {}
    for ci, (W, H, xdec, ydec, bd, n_idx, (asw, ash), planes, p_skip) in enumerate(CASES):
        t0 = time.time()
        g = L.pixel_type(bd)
        dt = L.np_dtype(bd)
        cs = L.enum(c, "ChromaSampling", "Cs400" if planes == 1 else "Cs420")
        yy, xx = np.mgrid[0:H, 0:W]
        base = ((np.sin(xx / 5.0 + ci) + np.cos((yy + xx * (ci + 1)) / 7.0)) * 40 + 128)
        Y = np.clip(base + rng.integers(-4, 5, (H, W)), 0, 255).astype(np.int64) << (bd - 8)
        cw, ch = W >> xdec, H >> ydec
        U = (np.clip(128 + 50 * np.sin(xx[:ch, :cw] / 3.0) + rng.integers(-3, 4, (ch, cw)), 0, 255)).astype(np.int64) << (bd - 8)
        V = np.clip(Y[::1 << ydec, ::1 << xdec][:ch, :cw] // 2 + (40 << (bd - 8)), 0, (1 << bd) - 1)
        src = [Y, U, V]
        # the reconstruction: the source + coding noise with ringing (what CDEF is there to remove)
        rec = [np.clip(s + rng.integers(-10 << (bd - 8), (10 << (bd - 8)) + 1, s.shape) *
                       (rng.random(s.shape) < 0.3), 0, (1 << bd) - 1) for s in src]
        mi_cols, mi_rows = 2 * ((W + 7) // 8), 2 * ((H + 7) // 8)
        skip = (rng.random((mi_rows, mi_cols)) < p_skip).astype(np.uint8)
        skip[:16, :16] = 1 if ci % 2 else skip[:16, :16]      # a completely skipped superblock
        ystr = rng.integers(0, 64, 8).astype(np.uint8)
        uvstr = rng.integers(0, 64, 8).astype(np.uint8)
        ystr[0], uvstr[0] = 0, 3
        ystr[1], uvstr[1] = 63, 60
        ystr[2:5] = [2 * 4 + 1, 5 * 4 + 2, 13 * 4 + 3]       # rav1e's own presets among them
        uvstr[2:5] = [1 * 4 + 0, 3 * 4 + 1, 7 * 4 + 3]
        damping = int(rng.integers(3, 7))
        gw, gh = (W + 7) // 8, (H + 7) // 8
        grid = rng.integers(1 << 12, 1 << 16, (gh, gw)).astype(np.uint32)
        dscale = rng.integers(1 << 13, 1 << 15, 3).astype(np.uint32)
        n_sbx, n_sby = (mi_cols + 15) // 16, (mi_rows + 15) // 16
        err = np.zeros((n_sby, n_sbx, 8), np.uint64)
        best = np.full((n_sby, n_sbx), -1, np.int8)
        scales = R.RSlice([DS(int(v)) for v in grid.ravel()])
        for ay0 in range(0, n_sby, ash):
            for ax0 in range(0, n_sbx, asw):
                # ---- geometry of rdo_loop_decision (rdo.rs:2149-2166, 2184-2190, 2277-2296)
                sb_w, sb_h = min(asw, n_sbx - ax0), min(ash, n_sby - ay0)
                crop_w, crop_h = W - ax0 * 64, H - ay0 * 64
                pixel_w, pixel_h = min(crop_w, sb_w * 64), min(crop_h, sb_h * 64)
                aw, ah = (pixel_w + 7) >> 3 << 3, (pixel_h + 7) >> 3 << 3
                tb = TileBlocks(skip, ax0 * 16, ay0 * 16, min(sb_w * 16, mi_cols - ax0 * 16),
                                min(sb_h * 16, mi_rows - ay0 * 16))

                def cut(a, pl, pad_to):
                    xd, yd = (0, 0) if pl == 0 else (xdec, ydec)
                    x0, y0 = (ax0 * 64) >> xd, (ay0 * 64) >> yd
                    w_, h_ = aw >> xd, ah >> yd
                    sub = a[y0:y0 + h_, x0:x0 + w_]
                    # a frame whose size is not a multiple of 8 is allocated rounded up; the rows /
                    # columns beyond the visible picture replicate the edge (Frame padding)
                    sub = np.pad(sub, ((0, h_ - sub.shape[0]), (0, w_ - sub.shape[1])), mode="edge")
                    return L.plane_from_array(sub.astype(dt), bd, xpad=0, ypad=0, xdec=xd, ydec=yd)
                rec_subset = Obj(planes=R.RSlice([cut(rec[pl], pl, 8) for pl in range(3)]))
                src_planes = [cut(src[pl], pl, 8) for pl in range(3)]
                src_subset = Obj(planes=R.RSlice([p.as_region() for p in src_planes]))
                cdef_ref = Obj(planes=R.RSlice([cut(rec[pl], pl, 8) for pl in range(3)]))   # rec_subset.clone()
                fi = Obj(sequence=Obj(bit_depth=bd, chroma_sampling=cs, use_128x128_superblock=False),
                         cdef_damping=damping, cdef_y_strengths=R.RSlice([int(v) for v in ystr]),
                         cdef_uv_strengths=R.RSlice([int(v) for v in uvstr]), cpu_feature_level=None,
                         config=Obj(temporal_rdo=lambda: True),
                         coded_frame_data=R.Some(Obj(distortion_scales=scales, w_in_imp_b=gw)),
                         dist_scale=R.RSlice([DS(int(v)) for v in dscale]), **{"lambda": 123.0})
                setattr(fi, "lambda_v", 123.0)
                ts = Obj()
                ts.to_frame_block_offset = lambda tbo: PBO(BO(x=tbo._0.x, y=tbo._0.y))   # tile at the frame origin
                base_sbo = TSBO(SBO(x=ax0, y=ay0))
                dirs = analyze_range(g, fi, rec_subset, tb, sb_w, sb_h)
                tile_out = Obj(planes=R.RSlice([p.as_region() for p in cdef_ref.planes]))
                for sby in range(sb_h):
                    for sbx in range(sb_w):
                        blk = skip[ay0 * 16 + 16 * sby: ay0 * 16 + min(16 * sby + 16, tb.rows()),
                                   ax0 * 16 + 16 * sbx: ax0 * 16 + min(16 * sbx + 16, tb.cols())]
                        if blk.all():          # cdef_skip (rdo.rs:2196-2211)
                            continue
                        loop_sbo = TSBO(SBO(x=sbx, y=sby))
                        best_cost, best_new = -1.0, -1
                        for idx in range(n_idx):
                            filter_sb(g, fi, rec_subset, tile_out, tb, loop_sbo, idx, dirs[sby * sb_w + sbx])
                            e = 0
                            for pli in range(planes):
                                e += plane_error(g, base_sbo, loop_sbo, 1, 1, fi, ts, tb, cdef_ref, src_subset, pli)._0
                            err[ay0 + sby, ax0 + sbx, idx] = e
                            cost = float(e)                      # compute_rd_cost(fi, 0, err): lambda * 0 + err
                            if best_cost < 0.0 or cost < best_cost:
                                best_cost, best_new = cost, idx
                        best[ay0 + sby, ax0 + sbx] = best_new
        # compute_rd_cost executed once on the reference's text for the record (rate 0 -> the error itself)
        SD = L.struct(c, "ScaledDistortion")
        assert rd_cost(g, Obj(lambda_=7.5), 0, SD(12345)) == 12345.0
        k = "s%d" % ci
        out[k + "_meta"] = np.array([W, H, xdec, ydec, bd, damping, n_idx, asw, ash, planes], np.int32)
        for pl in range(3):
            out[k + "_rec%d" % pl] = rec[pl].astype(np.uint16)
            out[k + "_src%d" % pl] = src[pl].astype(np.uint16)
        out[k + "_skip"], out[k + "_ystr"], out[k + "_uvstr"] = skip, ystr, uvstr
        out[k + "_scales"], out[k + "_dscale"] = grid, dscale
        out[k + "_err"], out[k + "_best"] = err, best
        print(k, W, H, "best:", best.ravel().tolist(), "%.0f s" % (time.time() - t0), flush=True)
    L.save("cdef_search_ref.npz", out)


if __name__ == "__main__":
    main()
