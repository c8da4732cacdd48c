#!/usr/bin/env python3
"""tests/golden/activity_ref.npz: the spatial DistortionScale grid computed by the
REFERENCE'S OWN SOURCE TEXT (src/activity.rs:21-186: ActivityMask::from_plane,
fill_scales, variance_8x8, ssim_boost, apply_ssim_boost, ssim_boost_rsqrt),
transpiled by tools/rustlite and executed here.

Run in the build container:  python tests/golden/gen_activity_ref.py
"""
import numpy as np

import reflib as L
from reflib import R


def main():
    c = L.crate("activity.rs", "rdo.rs")
    from_plane = c.get("from_plane", owner="ActivityMask")
    fill = c.get("fill_scales", owner="ActivityMask")
    DS = L.struct(c, "DistortionScale")
    rng = np.random.default_rng(20260928)
    out = {}
    n = 0
    for bd in (8, 10, 12):
        for (w, h) in ((64, 48), (100, 52), (37, 29), (16, 8)):
            g = L.pixel_type(bd)
            yy, xx = np.mgrid[0:h, 0:w]
            img = (np.sin(xx / 6.0) * 50 + np.cos(yy / 4.0) * 30 + 128) * (1 << (bd - 8))
            img = np.clip(img + rng.integers(-(1 << (bd - 4)), 1 << (bd - 4), (h, w)) *
                          (rng.random((h, w)) < 0.5), 0, (1 << bd) - 1).astype(L.np_dtype(bd))
            if n % 4 == 3:
                img[:] = rng.integers(0, 1 << bd, (h, w))      # high-variance blocks
            padded = np.pad(img, 16, mode="edge")                # Frame::pad replicates the edges
            plane = L.plane_from_padded(padded, bd, 16, 16)
            mask = from_plane(g, plane)
            var = np.array(mask.variances.tolist(), np.uint32)
            scales = R.RSlice([DS(0) for _ in range(len(var))])
            fill({}, mask, bd, scales)
            k = "%d_%d_%d" % (bd, w, h)
            out["img_" + k] = img
            out["var_" + k] = var.reshape((h + 7) // 8, (w + 7) // 8)
            out["scale_" + k] = np.array([s._0 for s in scales.tolist()], np.uint32).reshape(var.shape[0] and
                                                                                             ((h + 7) // 8, (w + 7) // 8))
            n += 1
    out["keys"] = np.array(sorted(k[4:] for k in out if k.startswith("img_")))
    L.save("activity_ref.npz", out)


if __name__ == "__main__":
    main()
