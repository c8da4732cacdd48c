#!/usr/bin/env python3
"""tests/golden/lrf_geometry_ref.npz: the restoration-unit GEOMETRY that the restoration leg of rdo_loop_decision
walks, computed by the REFERENCE'S OWN SOURCE TEXT through tools/rustlite:

  RestorationState::new                      src/lrf.rs:1321-1480   (whole function: unit sizes from the quantizer
      and the chroma-stretch test, the tiling restriction, the 4:2:2 / 4:4:4 tie, the last-unit rule of cols / rows,
      the sb shifts and stripe heights handed to RestorationPlane::new)
  vis_width / vis_height of a unit           src/rdo.rs:2645-2654   (through a two-line probe that restates
      `unit_size.min((crop >> dec) - loop_sbo.plane_offset(..))`)

This is the part of gen_lrf_search_ref.py / the frame tools' unit lists that used to be hand-stated (which unit gets
which visible size, how many units a plane has); rav1e_amd.rdo_glue.restoration_plane_configs /
restoration_search_units restate it on the product side and tests/test_oracle_txsearch_ref.py compares them with
these rows.

Keys: geo_in  = [width, height, xdec, ydec, base_q_idx, enable_large_lru, enable_restoration, sb128, tile cols, tile rows,
                 tile_width_sb, tile_height_sb] per case;
      geo_cfg = [case][plane][unit_size, sb_h_shift, sb_v_shift, stripe_height, cols, rows];
      geo_vis = [case][plane][vis_w of the last unit column that starts inside the plane, vis_h of the last row] (the probe)

Run in the build container:  python tests/golden/gen_lrf_geometry_ref.py
"""
import numpy as np

import reflib as L
from reflib import R
from gen_rdo_glue_ref import Obj

PROBE = """
pub fn r1_probe_lru_vis(unit_size: usize, crop: usize, dec: usize, plane_offset: usize) -> usize {
  unit_size.min((crop >> dec) - plane_offset)
}
"""


def main():
    c = L.crate("lrf.rs", "context/superblock_unit.rs", "context/block_unit.rs", "tiling/plane_region.rs", "util/mod.rs")
    c.define_enum("ChromaSampling", ["Cs420", "Cs422", "Cs444", "Cs400"])
    c.load_text("<probe: the visible size of a restoration unit, rdo.rs:2645-2654>", PROBE)
    new = c.get("new", owner="RestorationState")
    vis = c.get("r1_probe_lru_vis")
    rng = np.random.default_rng(20260930)
    cases = []
    sizes = [(3840, 2160), (1920, 1080), (1280, 720), (352, 288), (176, 144), (642, 364), (1000, 600), (4096, 2304), (260, 130),
             (64, 64), (96, 32), (2, 2)]
    for (w, h) in sizes:
        for (xd, yd) in ((1, 1), (1, 0), (0, 0)):
            for q in (40, 161, 170, 201, 255):
                cases.append((w, h, xd, yd, q, 1, 1, 0, 1, 1, 0, 0))
    for (w, h) in ((3840, 2160), (1920, 1080), (642, 364)):
        for large in (0, 1):
            for rest in (0, 1):
                for sb128 in (0, 1):
                    cases.append((w, h, 1, 1, 120, large, rest, sb128, 1, 1, 0, 0))
    # tiling restricts the unit to a power-of-two number of superblocks that divides the tile
    for (w, h) in ((3840, 2160), (1920, 1080)):
        sbw, sbh = (w + 63) // 64, (h + 63) // 64
        for (tc, tr) in ((2, 1), (4, 2), (2, 4)):
            tw, th = -(-sbw // tc), -(-sbh // tr)
            for q in (100, 180, 230):
                for (xd, yd) in ((1, 1), (0, 0)):
                    cases.append((w, h, xd, yd, q, 1, 1, 0, tc, tr, tw, th))
    for _ in range(40):
        w, h = int(rng.integers(16, 4200)), int(rng.integers(16, 2400))
        cases.append((w, h, 1, 1, int(rng.integers(0, 256)), 1, 1, 0, 1, 1, 0, 0))
    cfgs, viss = [], []
    for (w, h, xd, yd, q, large, rest, sb128, tc, tr, tw, th) in cases:
        sbl = 7 if sb128 else 6
        fi = Obj(sequence=Obj(use_128x128_superblock=bool(sb128), enable_large_lru=bool(large), enable_restoration=bool(rest),
                              tiling=Obj(cols=tc, rows=tr, tile_width_sb=tw, tile_height_sb=th)),
                 width=w, height=h, base_q_idx=q, sb_width=(w + (1 << sbl) - 1) >> sbl, sb_height=(h + (1 << sbl) - 1) >> sbl)
        inp = Obj(planes=R.RSlice([Obj(cfg=Obj(xdec=0, ydec=0)), Obj(cfg=Obj(xdec=xd, ydec=yd)), Obj(cfg=Obj(xdec=xd, ydec=yd))]))
        rs = new({"T": "u8"}, fi, inp)
        row, vrow = [], []
        for pli, p in enumerate(rs.planes):
            cf = p.cfg
            row.append((int(cf.unit_size), int(cf.sb_h_shift), int(cf.sb_v_shift), int(cf.stripe_height), int(cf.cols), int(cf.rows)))
            dx, dy = (xd, yd) if pli else (0, 0)
            us = int(cf.unit_size)
            lx = min(int(cf.cols) - 1, max((w >> dx) - 1, 0) // us)
            ly = min(int(cf.rows) - 1, max((h >> dy) - 1, 0) // us)
            vrow.append((int(vis({}, us, w, dx, lx * us)), int(vis({}, us, h, dy, ly * us))))
        cfgs.append(row)
        viss.append(vrow)
    out = {"geo_in": np.array(cases, np.int32), "geo_cfg": np.array(cfgs, np.int32), "geo_vis": np.array(viss, np.int32)}
    print(len(cases), "cases; e.g.", cases[0], cfgs[0], viss[0])
    L.save("lrf_geometry_ref.npz", out)


if __name__ == "__main__":
    main()
