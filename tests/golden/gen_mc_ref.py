#!/usr/bin/env python3
"""tests/golden/mc_ref.npz: put_8tap / prep_8tap / mc_avg vectors computed by the
REFERENCE'S OWN SOURCE TEXT (src/mc.rs:110-479: SUBPEL_FILTERS, get_filter,
run_filter, put_8tap, prep_8tap, mc_avg), transpiled by tools/rustlite and
executed here.  Same key layout as mc_golden.npz, so every test that reads that
file also runs on this one.

Run in the build container:  python tests/golden/gen_mc_ref.py
"""
import numpy as np

import reflib as L
from reflib import R


def main():
    c = L.crate("mc.rs")
    put, prep, avg = c.get("put_8tap"), c.get("prep_8tap"), c.get("mc_avg")
    FM = [L.enum(c, "FilterMode", n) for n in ("REGULAR", "SMOOTH", "SHARP", "BILINEAR")]
    taps = np.array(c.const_value("SUBPEL_FILTERS").tolist(), np.int16)
    rng = np.random.default_rng(20260923)
    out = {"filters": taps}
    cases = []
    sizes = [(2, 2), (2, 4), (4, 2), (4, 4), (4, 8), (8, 4), (8, 8), (16, 8), (8, 16), (16, 16), (32, 16),
             (16, 32), (32, 32), (64, 64), (128, 8), (8, 64), (64, 16)]
    pairs = [(mx, my) for mx in range(3) for my in range(3)] + [(3, 3)]  # the 9 8-tap pairs + bilinear
    for bd in (8, 10, 12):
        g = L.pixel_type(bd)
        dt = L.np_dtype(bd)
        for (w, h) in sizes:
            nrep = 4 + len(pairs) if w * h <= 1024 else 6
            for rep in range(nrep):
                if rep < 4:
                    cf, rf = [(0, 0), (0, 4), (4, 0), (4, 4)][rep]  # the reference's own bench/test cases
                    mx, my = pairs[int(rng.integers(len(pairs)))]
                else:
                    cf, rf = int(rng.integers(0, 16)), int(rng.integers(0, 16))
                    mx, my = pairs[(rep - 4) % len(pairs)]
                if rep % 3 == 1:    # extreme-valued windows exercise the clamps and the i16 intermediate
                    win = rng.choice([0, (1 << bd) - 1], size=(h + 7, w + 7)).astype(dt)
                else:
                    win = rng.integers(0, 1 << bd, size=(h + 7, w + 7)).astype(dt)
                sp = L.plane_from_array(win, bd)
                src = R.PlaneSlice(sp, 3, 3)
                dp = L.plane_from_array(np.zeros((h, w), dt), bd)
                put(g, dp.as_region(), src, w, h, cf, rf, FM[mx], FM[my], bd, None)
                tmp = R.RSlice([0] * (w * h))
                prep(g, tmp, src, w, h, cf, rf, FM[mx], FM[my], bd, None)
                k = "%d_%d_%d_%d_%d_%d_%d_%d" % (bd, w, h, cf, rf, mx, my, rep)
                out["win_" + k] = win
                out["put_" + k] = L.plane_to_array(dp, dt)
                out["prep_" + k] = np.array(tmp.tolist(), np.int16).reshape(h, w)
                cases.append(k)
        # mc_avg (src/mc.rs:454-479) on prep outputs of two random predictions (in-range inputs)
        t1 = out["prep_" + cases[-1]].astype(np.int64)
        lim = (8191 if bd == 8 else 32767 - 8192)
        t1 = rng.integers(-8192 if bd > 8 else 0, lim, size=(16, 16))
        t2 = rng.integers(-8192 if bd > 8 else 0, lim, size=(16, 16))
        dp = L.plane_from_array(np.zeros((16, 16), dt), bd)
        avg(g, dp.as_region(), R.RSlice([int(v) for v in t1.ravel()]), R.RSlice([int(v) for v in t2.ravel()]),
            16, 16, bd, None)
        out["avg_t1_%d" % bd] = t1.astype(np.int16)
        out["avg_t2_%d" % bd] = t2.astype(np.int16)
        out["avg_out_%d" % bd] = L.plane_to_array(dp, dt)
    out["cases"] = np.array(cases)
    L.save("mc_ref.npz", out)
    print(len(cases), "cases")


if __name__ == "__main__":
    main()
