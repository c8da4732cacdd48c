#!/usr/bin/env python3
"""tests/golden/lrf_ref.npz: self-guided loop restoration computed by the REFERENCE'S OWN
SOURCE TEXT (src/lrf.rs: RestorationState::lrf_filter_frame 1482-1585 with
RestorationPlane::restoration_unit_by_stripe 1295-1320, setup_integral_image 530-628,
sgrproj_stripe_filter 630-845 and the box sums / filters 150-400 below them;
sgrproj_solve 847-1096 called as the restoration search calls it, src/rdo.rs:2651-2684),
transpiled by tools/rustlite and executed here.  Same key layout as lrf_golden.npz (the
independent spec model), so every test that reads that file also runs on this one;
plus <s>_solve_* = units, sets and the (xqd0, xqd1) sgrproj_solve returned.

Hand-stated: the Frame wrapper (a list of planes that can be cloned), the FrameInvariants
fields read, and -- for the single-plane cases -- that the plane under test is presented to
lrf_filter_frame as plane 0 of a Cs400 frame carrying that plane's own decimation.

Run in the build container:  python tests/golden/gen_lrf_ref.py
"""
import numpy as np

import reflib as L
from reflib import R
from gen_lrf_golden import CASES, LRF_UNIT


class Obj:
    def __init__(self, **kw):
        self.__dict__.update(kw)


class PixelVec(list):
    """Plane::data as the reference indexes it: data[i] and data[range]"""
    def __getitem__(self, i):
        if isinstance(i, R.RRange):
            return R.RSlice(self, i.lo or 0, (len(self) if i.hi is None else i.hi) - (i.lo or 0))
        return list.__getitem__(self, i)


class Frame:
    def __init__(self, planes):
        for p in planes:
            p.data = PixelVec(p.data)
        self.planes = R.RSlice(planes)

    def clone(self):
        return Frame([R.Plane(PixelVec(p.data), p.cfg) for p in self.planes.tolist()])


def main():
    c = L.crate("lrf.rs", "tiling/plane_region.rs")
    c.define_enum("ChromaSampling", ["Cs420", "Cs422", "Cs444", "Cs400"])
    cs400 = L.enum(c, "ChromaSampling", "Cs400")
    filter_frame = c.get("lrf_filter_frame", owner="RestorationState")
    rp_new = c.get("new", owner="RestorationPlane")
    RState = L.struct(c, "RestorationState")
    RUnit = L.struct(c, "RestorationUnit")
    setup_ii, solve = c.get("setup_integral_image"), c.get("sgrproj_solve")
    IIB = c.get("zeroed", owner="IntegralImageBuffer")
    SOLVE_STRIDE, SOLVE_SIZE = c.const_value("SOLVE_IMAGE_STRIDE"), c.const_value("SOLVE_IMAGE_SIZE")
    NONE_F = R.REnum("RestorationFilter", "None", 0)
    out = {}
    for ci, (name, w, h, ydec, fh, us, sh, bd) in enumerate(CASES):
        rng = np.random.default_rng(500 + ci)        # the same inputs as lrf_golden.npz
        yy, xx = np.mgrid[0:h, 0:w]
        base = (np.sin(xx / 9.0) + np.cos(yy / 7.0) + 2) / 4 * ((1 << bd) - 1)
        debl = np.clip(base + rng.integers(-12, 13, (h, w)) * (1 << (bd - 8)), 0, (1 << bd) - 1).astype(np.int64)
        cdef = np.clip(debl + rng.integers(-3, 4, (h, w)) * (1 << (bd - 8)), 0, (1 << bd) - 1).astype(np.int64)
        cols = max((w + us // 2) // us, 1)
        rows = max((h + us // 2) // us, 1)
        units = np.zeros((rows, cols), LRF_UNIT)
        units["filter"] = rng.choice([0, 3, 3, 3], (rows, cols))
        units["set"] = rng.integers(0, 16, (rows, cols))
        units["xqd"][..., 0] = rng.integers(-96, 32, (rows, cols))
        units["xqd"][..., 1] = rng.integers(-32, 96, (rows, cols))
        units["xqd"][..., 0] = np.where(units["set"] >= 10, np.where(units["set"] >= 14, units["xqd"][..., 0], 0),
                                        units["xqd"][..., 0])
        g = dict(L.pixel_type(bd))
        g["U"] = g["T"]          # sgrproj_stripe_filter<T, U>: input and output pixels of the same type
        dt = L.np_dtype(bd)
        xdec = 1 if name.startswith("chroma") else 0
        # fi.width / fi.height: the luma frame whose decimation gives this plane (crop_w = (fw + xdec) >> xdec)
        fw = (w << xdec) - (xdec if (w << xdec) > 1 and name.endswith("odd") else 0)
        assert (fw + (1 << xdec >> 1)) >> xdec == w and (fh + (1 << ydec >> 1)) >> ydec == h, (name, fw, fh)
        fi = Obj(sequence=Obj(bit_depth=bd, chroma_sampling=cs400, enable_cdef=True), width=fw, height=fh,
                 cpu_feature_level=None)
        mk = lambda a: L.plane_from_padded(np.pad(a, 16, mode="edge").astype(dt), bd, 16, 16, xdec, ydec)
        frame_out, pre_cdef = Frame([mk(cdef)]), Frame([mk(debl)])
        rp = rp_new({}, 3, us, 0, 0, 1, 1, 1 if sh == 32 else 0, cols, rows)
        for ry in range(rows):
            for rx in range(cols):
                u = units[ry, rx]
                f = NONE_F if int(u["filter"]) == 0 else R.REnum(
                    "RestorationFilter", "Sgrproj", 2, (int(u["set"]), R.array(int(u["xqd"][0]), int(u["xqd"][1]))))
                rp.units[ry][rx] = RUnit(filter=f)
        state = RState(planes=R.RSlice([rp, rp, rp]))
        filter_frame(g, state, frame_out, pre_cdef, fi)
        res = L.plane_to_array(frame_out.planes[0], dt)
        out[name + "_meta"] = np.array([w, h, ydec, fh, us, sh, bd])
        out[name + "_units"] = units
        out[name + "_cdef"], out[name + "_debl"], out[name + "_out"] = cdef.astype(dt), debl.astype(dt), res
        print(name, "changed", int((res != cdef).sum()), "of", cdef.size, flush=True)
    # ---- sgrproj_solve as rdo_loop_decision calls it (rdo.rs:2651-2684)
    for bd in (8, 10):
        rng = np.random.default_rng(31 + bd)
        h, w = 72, 88
        yy, xx = np.mgrid[0:h, 0:w]
        src = np.clip((np.sin(xx / 6.0) + np.cos(yy / 5.0) + 2) / 4 * ((1 << bd) - 1), 0, (1 << bd) - 1).astype(np.int64)
        cdef = np.clip(src + rng.integers(-9, 10, (h, w)) * (1 << (bd - 8)), 0, (1 << bd) - 1)
        g = dict(L.pixel_type(bd))
        g["U"] = g["T"]          # sgrproj_stripe_filter<T, U>: input and output pixels of the same type
        dt = L.np_dtype(bd)
        fi = Obj(sequence=Obj(bit_depth=bd), cpu_feature_level=None)
        pc = L.plane_from_padded(np.pad(cdef, 16, mode="edge").astype(dt), bd, 16, 16)
        ps = L.plane_from_padded(np.pad(src, 16, mode="edge").astype(dt), bd, 16, 16)
        pc.data, ps.data = PixelVec(pc.data), PixelVec(ps.data)
        buf = IIB({}, SOLVE_SIZE)
        rects, res = [(0, 0, 24, 20), (32, 16, 28, 17), (64, 40, 24, 32)], []
        for (x0, y0, uw, uh) in rects:
            sl = R.PlaneSlice(pc, x0, y0)
            setup_ii(g, buf, SOLVE_STRIDE, uw, uh, uw, uh, sl, sl)
            for set_ in range(16):
                r = solve(g, set_, fi, buf, ps._region(x0, y0, w - x0, h - y0), sl, uw, uh)
                res.append((x0, y0, uw, uh, set_, int(r[0]), int(r[1])))
        out["solve%d_cdef" % bd], out["solve%d_src" % bd] = cdef.astype(dt), src.astype(dt)
        out["solve%d_cases" % bd] = np.array(res, np.int32)
        print("solve", bd, len(res), flush=True)
    L.save("lrf_ref.npz", out)


if __name__ == "__main__":
    main()
