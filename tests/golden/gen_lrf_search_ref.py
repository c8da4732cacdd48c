#!/usr/bin/env python3
"""tests/golden/lrf_search_ref.npz: the restoration-filter leg of rdo_loop_decision computed by the
REFERENCE'S OWN SOURCE TEXT, transpiled by tools/rustlite and executed here:

  setup_integral_image, sgrproj_solve, sgrproj_stripe_filter and the box sums / filters below
  them                                                            src/lrf.rs:150-400,530-1096
  rdo_loop_plane_error, distortion_scale, sse_wxh, the Distortion types
                                                                  src/rdo.rs:142-224,558-723,2027-2093
  cdef_dist_kernel, get_weighted_sse, apply_ssim_boost           src/dist.rs:234-372, src/activity.rs

Hand-stated: the control flow AROUND those calls -- the per-plane, per-restoration-unit loop of
src/rdo.rs:2575-2763 (the "no filter option" error, the unit's visible size, the integral image of
the unit alone, and per parameter set: solve -> filter into the working copy -> error), the working
copy itself (lrf_ref = a clone of the CDEF output) and the encoder-state containers.  The RATE of a
choice (cw.fc.count_lrf_switchable: the entropy coder's CDFs) and the comparison of costs are NOT
here: the device returns (xqd, error) per (unit, set) and the host adds its rate.

Round 5: gen_loop_decision_ref.py executes rdo_loop_decision itself and showed what this hand-stated loop has
DIFFERENT from it: the function filters every unit on a scratch copy of the AREA it is deciding, so the pixels left
of / above a unit exist only inside that area; here setup_integral_image is handed slices of WHOLE-FRAME planes, so
every unit with x > 0 / y > 0 sees its real neighbours.  These vectors therefore stand for "everything around the
unit exists" (R1SgrSolveUnit.edges = LEFT | ABOVE, what a unit in the middle of a multi-unit area sees); the
function's own behaviour is in loop_decision_ref.npz.

Keys per case <c>: <c>_meta = [W, H, xdec, ydec, bd, lru_sb], <c>_in{0,1,2} (the CDEF output) /
<c>_src{0,1,2}, <c>_scales (per 8x8 luma block, Q14), <c>_dscale (fi.dist_scale),
<c>_rows = [pli, x, y, w, h, set (255 = no filter), xqd0, xqd1], <c>_err (u64, one per row).

Run in the build container:  python tests/golden/gen_lrf_search_ref.py
"""
import os
import time

import numpy as np

import reflib as L
from reflib import R
from gen_lrf_ref import Obj, PixelVec
from gen_cdef_search_ref import TileBlocks

CASES = [
    # W, H, xdec, ydec, bd, lru_sb (superblocks per restoration unit side), sets tried
    (136, 72, 1, 1, 8, 1, (0, 5, 9, 10, 13, 14, 15)),
    (192, 128, 1, 1, 10, 2, (1, 11, 15)),
    (104, 64, 0, 0, 8, 1, (3, 12, 14)),
    (96, 80, 1, 0, 12, 1, (7, 10, 15)),
]


def main():
    c = L.crate("lrf.rs", "rdo.rs", "dist.rs", "activity.rs", "context/superblock_unit.rs",
                "context/block_unit.rs", "tiling/plane_region.rs")
    c.define_enum("ChromaSampling", ["Cs420", "Cs422", "Cs444", "Cs400"])
    setup_ii, solve, stripe = c.get("setup_integral_image"), c.get("sgrproj_solve"), c.get("sgrproj_stripe_filter")
    plane_error = c.get("rdo_loop_plane_error")
    IIB = c.get("zeroed", owner="IntegralImageBuffer")
    SOLVE_STRIDE, SOLVE_SIZE = c.const_value("SOLVE_IMAGE_STRIDE"), c.const_value("SOLVE_IMAGE_SIZE")
    TSBO, SBO = c.G["S_TileSuperBlockOffset"], c.G["S_SuperBlockOffset"]
    PBO, BO = L.struct(c, "PlaneBlockOffset"), L.struct(c, "BlockOffset")
    DS = c.G["S_DistortionScale"]
    Rect = lambda x, y, w, h: R.REnum("Area", "Rect", 0, (x, y, w, h))
    out = {}
    only = os.environ.get("R1_LRF_SEARCH_CASES")      # "0,2": a subset (the mutation check of the tests)
    for ci, (W, H, xdec, ydec, bd, lru_sb, sets) in enumerate(CASES):
        if only and str(ci) not in only.split(","):
            continue
        t0 = time.time()
        rng = np.random.default_rng([20260928, ci])      # per case: a subset run sees the same inputs
        g = dict(L.pixel_type(bd))
        g["U"] = g["T"]
        dt = L.np_dtype(bd)
        cs = L.enum(c, "ChromaSampling", "Cs420")
        yy, xx = np.mgrid[0:H, 0:W]
        base = ((np.sin(xx / 6.0 + ci) + np.cos((yy + xx * (ci + 1)) / 9.0)) * 45 + 128)
        Y = np.clip(base + rng.integers(-3, 4, (H, W)), 0, 255).astype(np.int64) << (bd - 8)
        cw, ch = W >> xdec, H >> ydec
        U = (np.clip(128 + 50 * np.sin(xx[:ch, :cw] / 4.0) + rng.integers(-3, 4, (ch, cw)), 0, 255)).astype(np.int64) << (bd - 8)
        V = np.clip(Y[::1 << ydec, ::1 << xdec][:ch, :cw] // 2 + (40 << (bd - 8)), 0, (1 << bd) - 1)
        src = [Y, U, V]
        # the CDEF output: the source + coding noise
        lin = [np.clip(s + rng.integers(-9 << (bd - 8), (9 << (bd - 8)) + 1, s.shape), 0, (1 << bd) - 1) for s in src]
        gw, gh = (W + 7) // 8, (H + 7) // 8
        grid = rng.integers(1 << 12, 1 << 16, (gh, gw)).astype(np.uint32)
        dscale = rng.integers(1 << 13, 1 << 15, 3).astype(np.uint32)
        scales = R.RSlice([DS(int(v)) for v in grid.ravel()])
        mi_cols, mi_rows = 2 * gw, 2 * gh
        tb = TileBlocks(np.zeros((mi_rows, mi_cols), np.uint8), 0, 0, mi_cols, mi_rows)

        def mk(a, pl):
            xd, yd = (0, 0) if pl == 0 else (xdec, ydec)
            p = L.plane_from_padded(np.pad(a, 16, mode="edge").astype(dt), bd, 16, 16, xdec=xd, ydec=yd)
            p.data = PixelVec(p.data)
            return p
        lrf_input = Obj(planes=R.RSlice([mk(lin[pl], pl) for pl in range(3)]))
        lrf_ref = Obj(planes=R.RSlice([mk(lin[pl], pl) for pl in range(3)]))       # cdef_ref.clone()
        src_planes = [mk(src[pl], pl) for pl in range(3)]
        src_subset = Obj(planes=R.RSlice([p.as_region() for p in src_planes]))
        fi = Obj(sequence=Obj(bit_depth=bd, chroma_sampling=cs, use_128x128_superblock=False),
                 cpu_feature_level=None, config=Obj(temporal_rdo=lambda: True),
                 coded_frame_data=R.Some(Obj(distortion_scales=scales, w_in_imp_b=gw)),
                 dist_scale=R.RSlice([DS(int(v)) for v in dscale]))
        ts = Obj()
        ts.to_frame_block_offset = lambda tbo: PBO(BO(x=tbo._0.x, y=tbo._0.y))      # tile at the frame origin
        base_sbo = TSBO(SBO(x=0, y=0))
        buf = IIB({}, SOLVE_SIZE)
        rows, errs = [], []
        n_sbx, n_sby = (W + 63) // 64, (H + 63) // 64
        for pli in range(3):
            xd, yd = (0, 0) if pli == 0 else (xdec, ydec)
            unit_x, unit_y = (64 * lru_sb) >> xd, (64 * lru_sb) >> yd          # the unit in pixels of this plane
            for lru_y in range((n_sby + lru_sb - 1) // lru_sb):
                for lru_x in range((n_sbx + lru_sb - 1) // lru_sb):
                    loop_sbo = TSBO(SBO(x=lru_x * lru_sb, y=lru_y * lru_sb))
                    px0, py0 = (lru_x * lru_sb * 64) >> xd, (lru_y * lru_sb * 64) >> yd
                    # rdo.rs:2645-2654 (unit_size is one number in the reference: square units; 4:2:2 planes
                    # halve one side only here, which its loop never produces -- our case list keeps ydec = 0
                    # to 64-row units of a 32-wide... the min() below is the reference's)
                    vis_w = min(unit_x, (W >> xd) - px0)
                    vis_h = min(unit_y, (H >> yd) - py0)
                    e = plane_error(g, base_sbo, loop_sbo, lru_sb, lru_sb, fi, ts, tb, lrf_input, src_subset, pli)._0
                    rows.append((pli, px0, py0, vis_w, vis_h, 255, 0, 0))
                    errs.append(e)
                    lp = lrf_input.planes[pli]
                    sl = R.PlaneSlice(lp, px0, py0)
                    setup_ii(g, buf, SOLVE_STRIDE, vis_w, vis_h, vis_w, vis_h, sl, sl)
                    for set_ in sets:
                        x = solve(g, set_, fi, buf, src_planes[pli]._region(px0, py0, (W >> xd) - px0, (H >> yd) - py0),
                                  sl, vis_w, vis_h)
                        xqd = R.RSlice([int(x[0]), int(x[1])])
                        stripe(g, set_, xqd, fi, buf, SOLVE_STRIDE, sl,
                               lrf_ref.planes[pli].region_mut(Rect(px0, py0, vis_w, vis_h)))
                        e = plane_error(g, base_sbo, loop_sbo, lru_sb, lru_sb, fi, ts, tb, lrf_ref, src_subset, pli)._0
                        rows.append((pli, px0, py0, vis_w, vis_h, set_, int(x[0]), int(x[1])))
                        errs.append(e)
        k = "s%d" % ci
        out[k + "_meta"] = np.array([W, H, xdec, ydec, bd, lru_sb], np.int32)
        for pl in range(3):
            out[k + "_in%d" % pl] = lin[pl].astype(np.uint16)
            out[k + "_src%d" % pl] = src[pl].astype(np.uint16)
        out[k + "_scales"], out[k + "_dscale"] = grid, dscale
        out[k + "_rows"] = np.array(rows, np.int32)
        out[k + "_err"] = np.array(errs, np.uint64)
        print(k, W, H, bd, len(rows), "rows, %.0f s" % (time.time() - t0), flush=True)
    L.save("lrf_search_ref.npz", out)


if __name__ == "__main__":
    main()
