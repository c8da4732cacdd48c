#!/usr/bin/env python3
"""Whole-frame hierarchical motion estimation
on a 4K luma frame: one JSON line per job configuration (tiles x reference frames), plus
the CPU oracle (one thread, one tile x one reference) for scale.

    python tools/bench_me.py [--width 3840 --height 2160 --bit-depth 8 --reps 5 --cpu]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def texture(w, h, bd, seed):
    rng = np.random.default_rng(seed)
    f = rng.standard_normal((h + 64, w + 64)).astype(np.float32)
    for _ in range(3):
        f = (np.roll(f, 1, 0) + 2 * f + np.roll(f, -1, 0)) / 4
        f = (np.roll(f, 1, 1) + 2 * f + np.roll(f, -1, 1)) / 4
    return ((f - f.min()) / (f.max() - f.min()) * ((1 << bd) - 1)).astype(np.int64)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--bit-depth", type=int, default=8)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--sustain-ms", type=float, default=100.0,
                    help="untimed runs of a row for this long before it is timed (sustained clocks); 0 = off")
    ap.add_argument("--cpu", action="store_true", help="also time the CPU oracle (tens of seconds)")
    ap.add_argument("--only", type=int, default=-1, help="run only tile-ME configuration i (for profiling)")
    ap.add_argument("--tile-only", action="store_true", help="stop after the tile-ME configurations")
    args = ap.parse_args()
    import torch
    import oracle_lib as O
    from rav1e_amd import workload as W
    from rav1e_amd.api import Context, Plane, me_lambdas
    w, h, bd = args.width, args.height, args.bit_depth
    f = texture(w, h, bd, 1)
    rng = np.random.default_rng(2)
    org = f[32:32 + h, 32:32 + w]
    shifts = [(5, -9), (-3, 2), (12, 7), (0, -1)]
    refs = [np.clip(f[32 + dy:32 + dy + h, 32 + dx:32 + dx + w] + rng.integers(-2, 3, (h, w)), 0,
                    (1 << bd) - 1) for dx, dy in shifts]
    po = O.me_pyramid(org, bd)
    prs = [O.me_pyramid(r, bd) for r in refs]
    dev = lambda pyr: [Plane.from_numpy(p.data, p.width, p.height, bd, p.xpad, p.ypad) for p in pyr]
    do, drs = dev(po), [dev(p) for p in prs]
    lam = me_lambdas(30.0)
    ctx = Context(0)
    rows, cols = h // 4, w // 4

    def launch_label(n_jobs):
        # r1_estimate_tile_motion_batch, launch_mode 0: one persistent launch, pinned from 8 jobs on
        force = os.environ.get("R1_ME_PERSISTENT")
        if force in (None, "1", "3"):
            pinned = force == "1" or (force is None and n_jobs >= 8)
            return "one persistent launch (k_me_persist), jobs %s" % ("pinned to an XCD each" if pinned else "not pinned")
        return "per-call launches (R1_ME_NO_GRAPH)" if os.environ.get("R1_ME_NO_GRAPH") else "hipGraph replay of the diagonal launches"

    def tiles_of(nx, ny):
        tw = -(-(w // nx) // 64) * 64
        th = -(-(h // ny) // 64) * 64
        return [(x, y, min(tw, w - x), min(th, h - y)) for y in range(0, h, th) for x in range(0, w, tw)]

    for ci, (nx, ny, nref) in enumerate(((1, 1, 1), (1, 1, 4), (4, 2, 1), (2, 2, 4), (4, 4, 4), (4, 2, 3))):
        if args.only >= 0 and ci != args.only:
            continue
        tl = tiles_of(nx, ny)
        stats = [torch.zeros((rows, cols, 2), dtype=torch.int32, device="cuda") for _ in range(nref)]
        jobs = [dict(org=do, ref=drs[r], stats=stats[r], tile=t) for r in range(nref) for t in tl]

        def run():
            for s in stats:
                s.zero_()
            ctx.estimate_tile_motion(jobs, cols, rows, bd, lam)
        run()
        torch.cuda.synchronize()
        W.sustain_clocks(run, args.sustain_ms)
        t0 = time.perf_counter()
        for _ in range(args.reps):
            run()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / args.reps * 1e3
        row = {"kernel": "estimate_tile_motion", "frame": "%dx%d" % (w, h), "bit_depth": bd,
               "tiles": len(tl), "refs": nref, "jobs": len(jobs), "ms": round(ms, 3),
               "launch": launch_label(len(jobs)),
               "Mpixels_s": round(w * h * nref / ms / 1e3, 1),
               "frames_refs_per_s": round(nref / ms * 1e3, 1)}
        # the dependent chain of the largest tile: a block needs its left and its upper neighbour of the same pass
        # (get_subset_predictors, src/me.rs:420-452), so pass 3 (16x16 blocks) of a tile of c x r blocks is c + r - 1
        # block searches long whatever the number of waves; passes 1 and 2 run skewed beside it
        tw_, th_ = max(t[2] for t in tl), max(t[3] for t in tl)
        steps = -(-tw_ // 16) + -(-th_ // 16) - 1
        row["chain_steps_pass3"] = steps
        row["us_per_chain_step"] = round(ms * 1e3 / steps, 2)
        if args.cpu and len(jobs) > 1:
            # every job of the concurrent launch against the oracle run tile by tile
            L = O.lib()
            L.r1o_set_threads(1)
            same = True
            for r in range(nref):
                st = np.zeros((rows, cols), O.ME_STATS)
                for t in tl:
                    O.me_oracle(L, po, prs[r], cols, rows, t, bd, lam, st)
                got = stats[r].cpu().numpy().reshape(rows, -1).view(O.ME_STATS).reshape(rows, cols)
                same = same and bool(np.array_equal(got, st))
            row["all_jobs_equal_oracle"] = same
        print(json.dumps(row), flush=True)
    # the glue that feeds the search from a resident frame: pad the full-resolution plane, make
    # the half- and quarter-resolution planes (each call pads its output) -- 3 launches
    def glue():
        ctx.plane_pad(do[0], w, h)
        hres = ctx.plane_downsample(do[0], w, h, 1)
        return hres, ctx.plane_downsample(hres, w, h, 2)
    hres, qres = glue()
    torch.cuda.synchronize()
    view = lambda t: t.cpu().numpy() if bd == 8 else t.cpu().numpy().view(np.uint16)
    same = bool(np.array_equal(view(hres.data), po[1].data) and np.array_equal(view(qres.data), po[2].data))
    t0 = time.perf_counter()
    for _ in range(20):
        glue()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 20 * 1e3
    moved = (do[0].data.numel() + 2 * hres.data.numel() + 2 * qres.data.numel()) * do[0].data.element_size()
    print(json.dumps({"kernel": "r1_plane_pad + 2 x r1_plane_downsample (allocation included)",
                      "frame": "%dx%d" % (w, h), "bit_depth": bd, "ms": round(ms, 4),
                      "GB_s": round(moved / ms / 1e6, 1), "equals_host_pyramid": same}), flush=True)
    if args.only >= 0 or args.tile_only:
        ctx.close()
        return
    # ---- RDO-time estimate_motion (full-pel + SATD + sub-pel diamond) on every block of a size ----
    from rav1e_amd.api import ME_BLOCK_CAND, ME_RESULT
    st0 = torch.zeros((rows, cols, 2), dtype=torch.int32, device="cuda")
    ctx.estimate_tile_motion([dict(org=do, ref=drs[0], stats=st0, tile=(0, 0, w, h))], cols, rows, bd, lam)
    job = dict(org=do, ref=drs[0], stats=st0, tile=(0, 0, w, h))
    blk_c = {}
    for s in (64, 32, 16, 8):
        nx, ny = w // s, h // s
        c = np.zeros(nx * ny, ME_BLOCK_CAND)
        c["bx"] = np.tile(np.arange(nx) * (s // 4), ny)
        c["by"] = np.repeat(np.arange(ny) * (s // 4), nx)
        c["w"] = c["h"] = s
        c["corner"] = 7
        c["pmv"] = rng.integers(-16, 17, (len(c), 2, 2))
        blk_c[s] = c
        dc = torch.from_numpy(c.view(np.uint8).reshape(-1).copy()).cuda()
        f = lambda: ctx.estimate_motion_batch(job, dc, cols, rows, bd, lam, max_w=s, max_h=s, n=len(c))
        f()
        torch.cuda.synchronize()
        W.sustain_clocks(f, args.sustain_ms)
        t0 = time.perf_counter()
        for _ in range(args.reps):
            f()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / args.reps * 1e3
        print(json.dumps({"kernel": "estimate_motion (full-pel + SATD + sub-pel) %dx%d" % (s, s),
                          "blocks": len(c), "bit_depth": bd, "ms": round(ms, 3),
                          "Mpixels_s": round(len(c) * s * s / ms / 1e3, 1),
                          "blocks_per_s": round(len(c) / ms * 1e3)}), flush=True)
    if args.cpu:
        L = O.lib()
        L.r1o_set_threads(os.cpu_count() or 1)
        hst = st0.cpu().numpy().reshape(rows, -1).view(O.ME_STATS).reshape(rows, cols)
        for s in (16,):
            t0 = time.perf_counter()
            want = O.me_block_oracle(L, po, prs[0], cols, rows, (0, 0, w, h), bd, lam, hst, None, blk_c[s])
            dt = time.perf_counter() - t0
            got = ctx.estimate_motion_batch(job, blk_c[s], cols, rows, bd, lam, max_w=s, max_h=s)
            same = bool(np.array_equal(got.cpu().numpy().view(ME_RESULT), want))
            print(json.dumps({"kernel": "estimate_motion %dx%d (CPU oracle, %d threads)" % (s, s, os.cpu_count()),
                              "ms": round(dt * 1e3, 1), "blocks_per_s": round(len(want) / dt),
                              "gpu_equals_oracle_at_4k": same}), flush=True)
        L.r1o_set_threads(1)
        st = np.zeros((rows, cols), O.ME_STATS)
        t0 = time.perf_counter()
        O.me_oracle(L, po, prs[0], cols, rows, (0, 0, w, h), bd, lam, st)
        dt = time.perf_counter() - t0
        got = torch.zeros((rows, cols, 2), dtype=torch.int32, device="cuda")
        ctx.estimate_tile_motion([dict(org=do, ref=drs[0], stats=got, tile=(0, 0, w, h))], cols, rows, bd, lam)
        same = bool(np.array_equal(got.cpu().numpy().reshape(rows, -1).view(O.ME_STATS).reshape(rows, cols), st))
        print(json.dumps({"kernel": "estimate_tile_motion (CPU oracle, 1 thread)", "ms": round(dt * 1e3, 1),
                          "Mpixels_s": round(w * h / dt / 1e6, 1), "gpu_equals_oracle_at_4k": same}), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
