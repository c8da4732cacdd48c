#!/usr/bin/env python3
"""tests/golden/dist_ref.npz: distortion vectors computed by the REFERENCE'S OWN
SOURCE TEXT, transpiled by tools/rustlite and executed here:

  get_sad, get_satd, get_weighted_sse, cdef_dist_kernel   src/dist.rs:31-372
  apply_ssim_boost, ssim_boost_rsqrt                      src/activity.rs:109-186
  DistortionScale::{new, mul_u64}, RawDistortion * scale  src/rdo.rs:558-700
  cdef_dist_wxh, sse_wxh, distortion_scale                src/rdo.rs:142-224,443-459

Hand-stated (no reference text to run): the v_frame Plane layout and the
macro-generated PlaneRegion accessors (tools/rustlite/runtime.py), and the two
encoder-state objects distortion_scale() reads (fi.config.temporal_rdo(),
fi.coded_frame_data.distortion_scales / w_in_imp_b), which are plain data here.

Run in the build container:  python tests/golden/gen_dist_ref.py
"""
import numpy as np

import reflib as L
from reflib import R

BLOCK_SIZES = [(4, 4), (4, 8), (8, 4), (8, 8), (8, 16), (16, 8), (16, 16), (16, 32), (32, 16),
               (32, 32), (32, 64), (64, 32), (64, 64), (64, 128), (128, 64), (128, 128),
               (4, 16), (16, 4), (8, 32), (32, 8), (16, 64), (64, 16)]


class Obj:
    def __init__(self, **kw):
        self.__dict__.update(kw)


def main():
    c = L.crate("dist.rs", "activity.rs", "rdo.rs")
    sad, satd = c.get("get_sad"), c.get("get_satd")
    wsse, cdk = c.get("get_weighted_sse"), c.get("cdef_dist_kernel")
    boost = c.get("apply_ssim_boost")
    ds_new = c.get("new", owner="DistortionScale")
    ds_mul = c.get("mul_u64", owner="DistortionScale")
    cdef_wxh, sse_wxh = c.get("cdef_dist_wxh"), c.get("sse_wxh")
    dscale_fn = c.get("distortion_scale")
    DS = c.G["S_DistortionScale"]
    rng = np.random.default_rng(20260924)
    out = {}

    def pair(bd, w, h, kind):
        dt = L.np_dtype(bd)
        a = rng.integers(0, 1 << bd, (h, w)).astype(dt)
        if kind == 0:
            b = rng.integers(0, 1 << bd, (h, w)).astype(dt)
        elif kind == 1:    # a noisy copy: the regime an encoder sees
            b = np.clip(a.astype(np.int64) + rng.integers(-9, 10, (h, w)), 0, (1 << bd) - 1).astype(dt)
        else:              # extremes
            a = rng.choice([0, (1 << bd) - 1], (h, w)).astype(dt)
            b = ((1 << bd) - 1 - a).astype(dt)
        return a, b

    # ---- SAD / SATD: the 22 block sizes + sizes whose edge tiles fall back to SAD
    keys, r_sad, r_satd = [], [], []
    for bd in (8, 10, 12):
        g = L.pixel_type(bd)
        for (w, h) in BLOCK_SIZES + [(12, 12), (20, 8), (8, 20), (36, 36), (4, 12), (12, 4)]:
            for kind in range(3 if w * h <= 1024 else 1):
                a, b = pair(bd, w, h, kind)
                ra, rb = L.plane_from_array(a, bd).as_region(), L.plane_from_array(b, bd).as_region()
                k = "%d_%d_%d_%d" % (bd, w, h, kind)
                out["d_org_" + k], out["d_ref_" + k] = a, b
                keys.append(k)
                r_sad.append(sad(g, ra, rb, w, h, bd, None))
                r_satd.append(satd(g, ra, rb, w, h, bd, None))
    out["d_keys"] = np.array(keys)
    out["d_sad"] = np.array(r_sad, np.uint32)
    out["d_satd"] = np.array(r_satd, np.uint32)

    # ---- get_weighted_sse: scale = 1.0, random in [0.5, 1.5), maximum (src/asm/shared/dist/sse.rs:89-193)
    keys, res = [], []
    for bd in (8, 10, 12):
        g = L.pixel_type(bd)
        for (w, h) in BLOCK_SIZES:
            for kind in range(3):
                a, b = pair(bd, w, h, kind)
                stride = 1
                while stride < w // 4:
                    stride *= 2
                stride += int(rng.integers(0, 2)) * 4   # strides wider than the block occur too
                if kind == 0:
                    sc = np.full((h // 4, stride), 1 << 14, np.uint32)
                elif kind == 1:
                    sc = rng.integers(1 << 13, 3 << 13, (h // 4, stride)).astype(np.uint32)
                else:
                    sc = np.full((h // 4, stride), (1 << 28) - 1, np.uint32)
                ra, rb = L.plane_from_array(a, bd).as_region(), L.plane_from_array(b, bd).as_region()
                k = "%d_%d_%d_%d" % (bd, w, h, kind)
                out["w_org_" + k], out["w_ref_" + k], out["w_scale_" + k] = a, b, sc
                keys.append(k)
                res.append(wsse(g, ra, rb, R.RSlice([int(v) for v in sc.ravel()]), stride, w, h, bd, None))
    out["w_keys"] = np.array(keys)
    out["w_out"] = np.array(res, np.uint64)

    # ---- cdef_dist_kernel: all w, h in 1..8 (src/asm/shared/dist/cdef_dist.rs:83-163)
    keys, res = [], []
    for bd in (8, 10, 12):
        g = L.pixel_type(bd)
        for w in range(1, 9):
            for h in range(1, 9):
                for kind in range(3):
                    a, b = pair(bd, w, h, kind)
                    ra, rb = L.plane_from_array(a, bd).as_region(), L.plane_from_array(b, bd).as_region()
                    k = "%d_%d_%d_%d" % (bd, w, h, kind)
                    out["k_org_" + k], out["k_ref_" + k] = a, b
                    keys.append(k)
                    res.append(cdk(g, ra, rb, w, h, bd, None))
    out["k_keys"] = np.array(keys)
    out["k_out"] = np.array(res, np.uint32)

    # ---- apply_ssim_boost
    rows, res = [], []
    for bd in (8, 10, 12):
        sh = 2 * (bd - 8)
        for _ in range(400):
            e1, e2 = rng.integers(0, 22), rng.integers(0, 22)
            svar = int(rng.integers(0, 1 << e1)) << sh
            dvar = int(rng.integers(0, 1 << e2)) << sh
            inp = int(rng.integers(0, 1 << int(rng.integers(1, 24))))
            rows.append((inp, min(svar, 2**32 - 1), min(dvar, 2**32 - 1), bd))
            res.append(boost({}, *rows[-1]))
    out["b_in"] = np.array(rows, np.uint64)
    out["b_out"] = np.array(res, np.uint32)

    # ---- DistortionScale::new / mul_u64
    rows, res = [], []
    for _ in range(300):
        num = int(rng.integers(0, 1 << int(rng.integers(1, 40))))
        den = int(rng.integers(1, 1 << int(rng.integers(1, 30))))
        rows.append((num, den))
        res.append(ds_new({}, num, den)._0)
    out["s_new_in"] = np.array(rows, np.uint64)
    out["s_new_out"] = np.array(res, np.uint32)
    rows, res = [], []
    for _ in range(300):
        s = int(rng.integers(1, 1 << 28))
        d = int(rng.integers(0, 1 << int(rng.integers(1, 34))))
        rows.append((s, d))
        res.append(ds_mul({}, DS(s), d))
    out["s_mul_in"] = np.array(rows, np.uint64)
    out["s_mul_out"] = np.array(res, np.uint64)

    # ---- cdef_dist_wxh / sse_wxh on planes with the frame's 8x8 DistortionScale grid
    W, H = 160, 96
    for bd in (8, 10, 12):
        g = L.pixel_type(bd)
        dt = L.np_dtype(bd)
        org = rng.integers(0, 1 << bd, (H, W)).astype(dt)
        ref = np.clip(org.astype(np.int64) + rng.integers(-12, 13, (H, W)), 0, (1 << bd) - 1).astype(dt)
        gw, gh = (W + 7) // 8, (H + 7) // 8
        grid = rng.integers(1 << 10, 1 << 17, (gh, gw)).astype(np.uint32)
        out["f_org_%d" % bd], out["f_ref_%d" % bd], out["f_scales_%d" % bd] = org, ref, grid
        for (xdec, ydec) in ((0, 0), (1, 1)):
            # a chroma plane is addressed in its own pixels; its importance lookup is in luma units
            pw, ph = (W >> xdec), (H >> ydec)
            po = L.plane_from_array(org[:ph, :pw], bd, xdec=xdec, ydec=ydec)
            pr = L.plane_from_array(ref[:ph, :pw], bd, xdec=xdec, ydec=ydec)
            ro, rr = po.as_region(), pr.as_region()
            scales = R.RSlice([DS(int(v)) for v in grid.ravel()])
            for use_grid in (1, 0):
                fi = Obj(config=Obj(temporal_rdo=lambda use_grid=use_grid: bool(use_grid)),
                         coded_frame_data=R.Some(Obj(distortion_scales=scales, w_in_imp_b=gw)))
                for (w, h) in [(4, 4), (8, 8), (16, 16), (32, 32), (64, 64), (8, 16), (32, 8), (4, 16),
                               (12, 20), (24, 8), (40, 64), (8, 12)]:
                    if w > pw or h > ph:
                        continue
                    n = 6
                    cands = np.zeros((n, 4), np.int16)
                    cands[:, 0] = rng.integers(0, pw - w + 1, n) & ~3
                    cands[:, 1] = rng.integers(0, ph - h + 1, n) & ~3
                    cands[:, 2] = np.clip(cands[:, 0] + rng.integers(-2, 3, n), 0, pw - w)
                    cands[:, 3] = np.clip(cands[:, 1] + rng.integers(-2, 3, n), 0, ph - h)
                    for kind in ((2, 3) if xdec == 0 else (2,)):
                        if kind == 2 and (w % 4 or h % 4):
                            continue
                        res = []
                        for (ox, oy, rx, ry) in cands.tolist():
                            a_org = c.G["_E"]("Area", "StartingAt", 1, (ox, oy))
                            a_ref = c.G["_E"]("Area", "StartingAt", 1, (rx, ry))
                            s1, s2 = ro.subregion(a_org), rr.subregion(a_ref)

                            def bias(area, bsize, s1=s1, fi=fi):
                                return dscale_fn(g, fi, s1.subregion(area).frame_block_offset(), bsize)
                            if kind == 3:
                                d = cdef_wxh(g, s1, s2, w, h, bd, bias, None)
                            else:
                                d = sse_wxh(g, s1, s2, w, h, bias, bd, None)
                            res.append(d._0)
                        k = "%d_%d_%d_%d_%d_%d" % (bd, kind, w, h, xdec, use_grid)
                        out["f_cands_" + k] = cands
                        out["f_out_" + k] = np.array(res, np.uint64)
    out["f_keys"] = np.array(sorted(k[len("f_cands_"):] for k in out if k.startswith("f_cands_")))
    L.save("dist_ref.npz", out)


if __name__ == "__main__":
    main()
