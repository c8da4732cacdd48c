#!/usr/bin/env python3
"""The transform-type search fan-out against independent single-type candidates, same box, same inputs:

    python tools/bench_txsearch.py [--bit-depth 8|10] [--k 4] [--reps 20] [--kind 3] [--sizes 8,16,32]

For every size: n = K predictions per block of a 3840x2160 plane (speed-6 ladder candidates, 1/16-pel
fractions), mask = RAV1E_TX_TYPES cut by the inter tx set (7 types at 8x8 / 16x16, 2 at 32x32).
  fanout      one r1_rdo_txsearch_batch launch (mc + diff once, the chain per type)
  independent nt r1_rdo_pixel_cand_batch launches, candidates' tx_type = t, SAD / SATD off
              (kind 0: r1_rdo_full_cand_batch)
One JSON line per size: ms of both, the ratio, (candidate, type) evaluations per second, and a bit-exact
comparison of every slot of the fan-out with the independent launches.  HIP events on the launch stream."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--bit-depth", type=int, default=10)
    ap.add_argument("--k", type=int, default=4)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--kind", type=int, default=3, help="3 cdef_dist, 2 weighted SSE, 0 transform-domain distortion")
    ap.add_argument("--sizes", default="8,16,32")
    ap.add_argument("--qindex", type=int, default=100)
    ap.add_argument("--sustain-ms", type=float, default=150.0)
    ap.add_argument("--fanout-only", action="store_true", help="rocprofv3 / PMC passes: only the fan-out launches")
    args = ap.parse_args()
    import time
    import torch
    from rav1e_amd import rdo_glue as RG, workload as W
    from rav1e_amd.api import Context, Plane
    from rav1e_amd.types import TxSize
    ctx = Context(0)
    fw, fh, bd = args.width, args.height, args.bit_depth
    a = W.random_plane_array(fw, fh, bd, 1)
    # a reference that resembles the source: residuals of a few grey levels, eobs of every length
    b = np.clip(a.astype(np.int64) + np.random.default_rng(5).integers(-6 << (bd - 8), (6 << (bd - 8)) + 1, a.shape), 0,
                (1 << bd) - 1).astype(a.dtype)
    org, ref = Plane.from_numpy(a, fw, fh, bd, 88, 88), Plane.from_numpy(b, fw, fh, bd, 88, 88)
    scales = torch.from_numpy(np.random.default_rng(9).integers(1 << 12, 1 << 16, ((fh + 7) // 8, (fw + 7) // 8)).astype(np.int32)).cuda()
    sizes = [int(s) for s in args.sizes.split(",")]
    cands = W.speed6_ladder(fw, fh, args.k, mv_range=4, sizes=sizes)

    def timed(f):
        f()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        while (time.perf_counter() - t0) * 1e3 < args.sustain_ms:
            f()
        torch.cuda.synchronize()
        ev = []
        for _ in range(args.reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            f()
            e1.record()
            ev.append((e0, e1))
        torch.cuda.synchronize()
        ms = sorted(x.elapsed_time(y) for x, y in ev)
        return ms[len(ms) // 2]

    for s in sizes:
        c = cands[s]
        n = len(c)
        ts = int(TxSize.by_dims(s, s))
        mask = ctx.tx_type_mask(ts, True)
        types = RG.tx_type_slots(mask)
        nt = len(types)
        dev = torch.from_numpy(c.view(np.uint8).reshape(-1).copy()).cuda()
        sc = scales if args.kind else None
        fo = {"eob": torch.empty((n, nt), dtype=torch.int16, device="cuda"), "dist": torch.empty((n, nt), dtype=torch.int64, device="cuda")}
        fan = lambda: ctx.rdo_txsearch_batch(org, ref, s, s, dev, mask, args.qindex, args.kind, scales=sc, n=n, outs=fo)
        ms_fan = timed(fan)
        if args.fanout_only:
            print(json.dumps({"size": s, "bd": bd, "kind": args.kind, "n": n, "types": types, "fanout_ms": round(ms_fan, 4)}), flush=True)
            continue
        devs, outs = [], []
        for t in types:
            ct = c.copy()
            ct["tx_type"] = t
            devs.append(torch.from_numpy(ct.view(np.uint8).reshape(-1).copy()).cuda())
            if args.kind:
                outs.append({"eob": torch.empty(n, dtype=torch.int16, device="cuda"), "dist": torch.empty(n, dtype=torch.int64, device="cuda")})
            else:
                outs.append({"eob": torch.empty(n, dtype=torch.int16, device="cuda"), "tx_dist": torch.empty(n, dtype=torch.int64, device="cuda")})

        def ind():
            for j in range(nt):
                if args.kind:
                    ctx.rdo_pixel_cand_batch(org, ref, s, s, devs[j], args.qindex, args.kind, scales=sc, n=n, outs=outs[j],
                                             want_sad=False, want_satd=False)
                else:
                    ctx.rdo_full_cand_batch(org, ref, s, s, devs[j], args.qindex, n=n, outs=outs[j], want_sad=False,
                                            want_satd=False, want_rate=False)
        ms_ind = timed(ind)
        ms_fan2 = timed(fan)      # once more after the other side: both saw the same clocks
        ok = all(torch.equal(fo["eob"][:, j], outs[j]["eob"]) and
                 torch.equal(fo["dist"][:, j], outs[j]["dist" if args.kind else "tx_dist"]) for j in range(nt))
        best = min(ms_fan, ms_fan2)
        print(json.dumps({"size": s, "bd": bd, "kind": args.kind, "n": n, "types": types, "fanout_ms": round(best, 4),
                          "fanout_ms_runs": [round(ms_fan, 4), round(ms_fan2, 4)], "independent_ms": round(ms_ind, 4),
                          "ratio": round(best / ms_ind, 4), "evals_per_s": round(n * nt / (best * 1e-3)),
                          "Mpixels_per_s": round(n * nt * s * s / (best * 1e-3) / 1e6), "slots_equal_independent": bool(ok)}),
              flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
