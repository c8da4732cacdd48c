#!/usr/bin/env python3
"""r1_lrf_search_batch on a 3840x2160 4:2:0 frame: every 64x64 luma / 32x32 chroma unit x (no filter + 8 sets).
    python tools/bench_lrf_search.py [--bit-depth 8|10] [--reps 20]
One JSON line: ms of the luma launch, of one chroma launch, the stage (luma + 2 chroma), and a checksum of the
results (A/B runs of two libraries must print the same one)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bit-depth", type=int, default=8)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--sustain-ms", type=float, default=150.0)
    args = ap.parse_args()
    import torch
    from rav1e_amd import api, workload as W
    from rav1e_amd.api import Context, Plane
    fw, fh, bd = 3840, 2160, args.bit_depth
    ctx = Context(0)
    sets = [255, 1, 3, 5, 7, 9, 11, 13, 15]
    mk = lambda w, h, seed, pad: Plane.from_numpy(W.random_plane_array(w, h, bd, seed, pad, pad), w, h, bd, pad, pad)
    planes = [(mk(fw, fh, 2, 88), mk(fw, fh, 1, 88)), (mk(fw // 2, fh // 2, 30, 44), mk(fw // 2, fh // 2, 31, 44))]
    scales = torch.from_numpy(np.random.default_rng(9).integers(1 << 12, 1 << 16, ((fh + 7) // 8, (fw + 7) // 8)).astype(np.int32)).cuda()

    def unit_list(pw, ph, us):
        u = [(x, y, min(us, pw - x), min(us, ph - y), s_, 0, (0, 0)) for y in range(0, ph, us) for x in range(0, pw, us) for s_ in sets]
        return torch.from_numpy(np.array(u, api.SGR_SOLVE_UNIT).view(np.uint8).reshape(-1).copy()).cuda()
    ul, uc = unit_list(fw, fh, 64), unit_list(fw // 2, fh // 2, 32)
    luma = lambda: ctx.lrf_search_batch(planes[0][0], planes[0][1], ul, scales=scales, max_w=64, max_h=64)
    chroma = lambda: ctx.lrf_search_batch(planes[1][0], planes[1][1], uc, is_chroma=True, xdec=1, ydec=1, scales=scales, max_w=32, max_h=32)

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
        ms = sorted(a.elapsed_time(b) for a, b in ev)
        return ms[len(ms) // 2]
    ml, mc = timed(luma), timed(chroma)
    xl, el = luma()
    xc, ec = chroma()
    cs = int((el.sum() + ec.sum() * 3 + xl.to(torch.int64).sum() * 7 + xc.to(torch.int64).sum() * 11).item()) & 0xFFFFFFFF
    print(json.dumps({"bd": bd, "luma_ms": round(ml, 4), "chroma_plane_ms": round(mc, 4), "stage_ms": round(ml + 2 * mc, 4),
                      "checksum": cs}))
    ctx.close()


if __name__ == "__main__":
    main()
