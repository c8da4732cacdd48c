#!/usr/bin/env python3
"""A/B of the two put_8tap formulations on one box: r1_mc_put_batch (dot4 / dot2 on the
VALU) against r1_mc_batch_mfma (horizontal pass on the matrix cores), 8-bit, block sizes
8 / 16 / 32 / 64, the bench's own 4K candidate lists.  One JSON line per (size, variant).

  python tools/bench_mc_mfma.py [--steps 30] [--only mfma|dot4]   (the --only forms are what the
  rocprofv3 --pmc passes of tools/gpu_mfma.sh run, so that the counters belong to one variant)
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--only", default="")
    ap.add_argument("--sizes", default="64,32,16,8")
    a = ap.parse_args()
    import torch
    from rav1e_amd import workload as W
    from rav1e_amd.api import Context, Plane, MC_CAND
    ctx = Context(0)
    fw, fh = 3840, 2160
    ref = Plane.from_numpy(W.random_plane_array(fw, fh, 8, 2), fw, fh, 8, 88, 88)
    lad = W.speed6_ladder(fw, fh, 16)
    for s in [int(x) for x in a.sizes.split(",")]:
        rc = lad[s]
        c = np.zeros(len(rc), MC_CAND)
        for f in ("rx", "ry", "col_frac", "row_frac", "mode_x", "mode_y"):
            c[f] = rc[f]
        dc = torch.from_numpy(c.view(np.uint8).reshape(-1).copy()).cuda()
        n = len(c)
        out = torch.empty((n, s, s), dtype=torch.uint8, device="cuda")
        variants = {"dot4": lambda: ctx.put_8tap_batch(ref, s, s, dc, n=n, out=out),
                    "mfma": lambda: ctx.mc_batch_mfma(ref, s, s, dc, n=n, out=out)}
        res = {}
        for name, fn in variants.items():
            if a.only and a.only != name:
                continue
            for _ in range(200):     # clocks
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.steps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / a.steps
            res[name] = ms
            print(json.dumps({"kernel": "put_8tap %dx%d 8-bit" % (s, s), "variant": name, "candidates": n,
                              "ms": round(ms, 4), "Mpixels_per_s": round(n * s * s / ms / 1e3, 1)}))
        if len(res) == 2:
            print(json.dumps({"kernel": "put_8tap %dx%d 8-bit" % (s, s),
                              "mfma_over_dot4_time": round(res["mfma"] / res["dot4"], 4)}))
    ctx.close()


if __name__ == "__main__":
    main()
