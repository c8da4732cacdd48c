#!/usr/bin/env python3
"""Golden frames for the self-guided loop restoration filter from the independent model in
tests/lrf_util.py (the reference, src/lrf.rs, holds no vectors).

    python tests/golden/gen_lrf_golden.py     # writes tests/golden/lrf_golden.npz
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import lrf_util as L   # noqa: E402

LRF_UNIT = np.dtype([("filter", "u1"), ("set", "u1"), ("xqd", "i1", (2,))])
CASES = [
    # name, plane w, h (already decimated), ydec, luma frame height, unit_size, stripe_height, bd
    ("luma_8", 104, 150, 0, 150, 64, 64, 8),
    ("luma_10", 96, 136, 0, 136, 64, 64, 10),
    ("chroma420_8", 72, 76, 1, 152, 32, 32, 8),
    ("chroma422_12", 56, 140, 0, 140, 64, 64, 12),
    ("luma_8_bigunit", 150, 80, 0, 80, 128, 64, 8),
    ("luma_8_odd", 70, 143, 0, 143, 64, 64, 8),               # odd height: the last stripe is odd
    ("chroma420_10_odd", 53, 71, 1, 141, 32, 32, 10),
]


def main():
    out = {}
    for ci, (name, w, h, ydec, fh, us, sh, bd) in enumerate(CASES):
        rng = np.random.default_rng(500 + ci)
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
        want = L.lrf_plane(cdef, debl, ydec, w, h, fh, us, units, sh, bd)
        dt = np.uint8 if bd == 8 else np.uint16
        out[name + "_meta"] = np.array([w, h, ydec, fh, us, sh, bd])
        out[name + "_units"] = units
        out[name + "_cdef"] = cdef.astype(dt)
        out[name + "_debl"] = debl.astype(dt)
        out[name + "_out"] = want.astype(dt)
        print(name, "changed", int((want != cdef).sum()), "of", cdef.size)
    np.savez_compressed(os.path.join(HERE, "lrf_golden.npz"), **out)


if __name__ == "__main__":
    main()
