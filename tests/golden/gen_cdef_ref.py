#!/usr/bin/env python3
"""tests/golden/cdef_ref.npz: whole-frame CDEF vectors computed by the REFERENCE'S
OWN SOURCE TEXT (src/cdef.rs:30-625: cdef_find_dir, first_max_element, constrain,
pad_into_tmp16, cdef_filter_block, adjust_strength, cdef_analyze_superblock,
cdef_filter_superblock, cdef_filter_tile; src/context/superblock_unit.rs offsets),
transpiled by tools/rustlite and executed here.  Same key layout as
cdef_golden.npz, so every test that reads that file also runs on this one.

Hand-stated: the v_frame Plane / PlaneRegion accessors (tools/rustlite/runtime.py)
and the encoder-state containers the drivers read -- FrameInvariants fields,
TileBlocks (skip flags, cdef index per superblock) -- which are plain data here.

Run in the build container:  python tests/golden/gen_cdef_ref.py
"""
import numpy as np

import reflib as L
from reflib import R


class Obj:
    def __init__(self, **kw):
        self.__dict__.update(kw)


class TileBlocks:
    """what cdef.rs reads of TileBlocks: cols()/rows() in 4x4 units, [bo].skip, get_cdef(sbo)"""

    def __init__(self, skip, cdef_index):
        self.skip, self.ci = skip, cdef_index

    def cols(self):
        return self.skip.shape[1]

    def rows(self):
        return self.skip.shape[0]

    def __getitem__(self, bo):
        b = bo._0
        return Obj(skip=bool(self.skip[b.y, b.x]))

    def get_cdef(self, sbo):
        return int(self.ci[sbo._0.y, sbo._0.x])


def main():
    c = L.crate("cdef.rs", "context/superblock_unit.rs", "context/block_unit.rs", "tiling/plane_region.rs")
    c.define_enum("ChromaSampling", ["Cs420", "Cs422", "Cs444", "Cs400"])
    filter_tile = c.get("cdef_filter_tile")
    analyze = c.get("cdef_analyze_superblock")
    TSBO, SBO = c.G["S_TileSuperBlockOffset"], c.G["S_SuperBlockOffset"]
    cs420 = L.enum(c, "ChromaSampling", "Cs420")
    rng = np.random.default_rng(20260925)
    out = {}
    case = 0
    for (W, H, xdec, ydec, bd) in ((72, 40, 1, 1, 8), (136, 72, 1, 1, 10), (64, 64, 0, 0, 8),
                                   (80, 24, 1, 0, 10), (24, 88, 1, 1, 12), (200, 136, 1, 1, 8)):
        for rep in range(2):
            g = L.pixel_type(bd)
            dt = L.np_dtype(bd)
            yy, xx = np.mgrid[0:H, 0:W]
            base = ((np.sin(xx / 5.0 + rep) + np.cos((yy + xx * (rep + 1)) / 7.0)) * 40 + 128)
            Y = np.clip(base + rng.integers(-20, 21, (H, W)), 0, 255).astype(np.int64) << (bd - 8)
            Y = np.clip(Y + rng.integers(0, 1 << (bd - 8), (H, W)), 0, (1 << bd) - 1)
            cw, ch = W >> xdec, H >> ydec
            U = rng.integers(0, 1 << bd, (ch, cw))
            V = np.clip((Y[::1 << ydec, ::1 << xdec][:ch, :cw] // 2 + rng.integers(-30, 31, (ch, cw))),
                        0, (1 << bd) - 1)
            skip = (rng.random((H // 4, W // 4)) < 0.35).astype(np.uint8)
            if rep:
                skip[:2] = 1
            nsb_y, nsb_x = -(-H // 64), -(-W // 64)
            cdef_index = rng.integers(0, 8, (nsb_y, nsb_x)).astype(np.uint8)
            ystr = rng.integers(0, 64, 8).astype(np.uint8)
            uvstr = rng.integers(0, 64, 8).astype(np.uint8)
            ystr[0], uvstr[0] = 0, 3          # zero primary; sec == 3 -> 4
            ystr[1], uvstr[1] = 63, 60
            damping = int(rng.integers(3, 7))

            planes_in = [L.plane_from_array(a.astype(dt), bd, xpad=8, ypad=8, xdec=xd, ydec=yd)
                         for a, (xd, yd) in zip((Y, U, V), ((0, 0), (xdec, ydec), (xdec, ydec)))]
            planes_out = [L.plane_from_array(np.zeros(a.shape, dt), bd, xpad=8, ypad=8, xdec=xd, ydec=yd)
                          for a, (xd, yd) in zip((Y, U, V), ((0, 0), (xdec, ydec), (xdec, ydec)))]
            fi = Obj(sequence=Obj(bit_depth=bd, chroma_sampling=cs420), cdef_damping=damping,
                     cdef_y_strengths=R.RSlice([int(v) for v in ystr]),
                     cdef_uv_strengths=R.RSlice([int(v) for v in uvstr]), cpu_feature_level=None)
            frame = Obj(planes=R.RSlice(planes_in))
            tile_out = Obj(planes=R.RSlice([p.as_region() for p in planes_out]))
            tb = TileBlocks(skip, cdef_index)
            filter_tile(g, fi, frame, tb, tile_out)
            # directions / variances as cdef_analyze_superblock reports them
            dirs = np.zeros((H // 8, W // 8), np.int32)
            vars_ = np.zeros((H // 8, W // 8), np.int32)
            for sby in range(nsb_y):
                for sbx in range(nsb_x):
                    d = analyze(g, fi, frame, tb, TSBO(SBO(x=sbx, y=sby)))
                    for by in range(8):
                        for bx in range(8):
                            fy, fx = sby * 8 + by, sbx * 8 + bx
                            if fy < H // 8 and fx < W // 8:
                                dirs[fy, fx] = d.dir[bx][by]
                                vars_[fy, fx] = d.var[bx][by]
            k = "c%d" % case
            out[k + "_meta"] = np.array([W, H, xdec, ydec, bd, damping], np.int32)
            for p, a in enumerate((Y, U, V)):
                out[k + "_in%d" % p] = a.astype(np.uint16)
                out[k + "_out%d" % p] = L.plane_to_array(planes_out[p], np.uint16)
            out[k + "_skip"], out[k + "_ci"] = skip, cdef_index
            out[k + "_ystr"], out[k + "_uvstr"] = ystr, uvstr
            out[k + "_dir"], out[k + "_var"] = dirs, vars_
            case += 1
            print(k, W, H, xdec, ydec, bd, flush=True)
    L.save("cdef_ref.npz", out)


if __name__ == "__main__":
    main()
