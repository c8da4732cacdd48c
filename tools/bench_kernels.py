#!/usr/bin/env python3
"""Per-stage microbenchmarks for every hot-path row of SURVEY.md 8(a): one JSON
line per kernel with Mpixels/s and the achieved fraction of the HBM roofline
computed from the row's ALGORITHMIC bytes.  Complements
bench.py (the fused headline candidate).  Run on the GPU box:

    python tools/bench_kernels.py [--bit-depth 8|10] [--reps 20] > gpurun_out/kernels.jsonl

Inputs are synthetic (uniform random planes, seeds as in bench.py) and resident
in HBM; timing uses HIP events on the launch stream.
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
HBM_PEAK = 8000.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--bit-depth", type=int, default=8)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--sustain-ms", type=float, default=100.0,
                    help="untimed launches for this long before a row is timed (sustained clocks); 0 = off")
    args = ap.parse_args()
    import torch
    from rav1e_amd import api, workload as W
    from rav1e_amd.api import Context, Plane
    from rav1e_amd.types import TxSize

    ctx = Context(0)
    fw, fh, bd = args.width, args.height, args.bit_depth
    bpp = 1 if bd == 8 else 2
    cb = 2 if bpp == 1 else 4
    ct = torch.int16 if bpp == 1 else torch.int32
    org = Plane.from_numpy(W.random_plane_array(fw, fh, bd, 1), fw, fh, bd, 88, 88)
    ref = Plane.from_numpy(W.random_plane_array(fw, fh, bd, 2), fw, fh, bd, 88, 88)
    rng = np.random.default_rng(3)
    # The rows that read PLANES (a1-a5, a8, a9) rotate through NP plane pairs whose footprint exceeds
    # the 256 MB Infinity Cache, so that a launch finds its planes in HBM, not in L3 (round-2 review:
    # "scale past L3 before reading them").  Beside the fraction by ALGORITHMIC bytes (every candidate
    # counted with its own block and window, the contract's figure: > 1 is possible because the K
    # candidates of a block overlap in L2) the line carries `hbm_frac_unique`: the bytes a launch can
    # at most pull from HBM (plane footprints + descriptors + outputs) over the same time.
    plane_bytes = org.data.numel() * org.data.element_size()
    NP = max(2, int(np.ceil(300e6 / (2 * plane_bytes))))
    pairs = [(org, ref)]
    for i in range(1, NP):
        a = W.random_plane_array(fw, fh, bd, 100 + 2 * i)
        pairs.append((Plane.from_numpy(a, fw, fh, bd, 88, 88),
                      Plane.from_numpy(np.roll(a, 7919 * i), fw, fh, bd, 88, 88)))
    vis_bytes = fw * fh * bpp

    def timeit(fn, reps=args.reps):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        W.sustain_clocks(fn, args.sustain_ms)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    def timeit_fresh(fn, restore, reps=args.reps):
        """an IN-PLACE filter on data-dependent paths: every rep starts from the same pixels (restore() runs
        untimed between the reps; each rep has its own event pair).  Round 3 timed deblock_filter_frame with
        plain timeit(): the plane was filtered over and over, went flat, every edge took the widest filter and
        the row read 0.195 ms where a fresh frame takes ~0.10 (the frame pipeline's figure)."""
        restore()
        fn()
        W.sustain_clocks(lambda: (restore(), fn()), args.sustain_ms)
        ms = []
        for _ in range(reps):
            restore()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            ms.append(e0.elapsed_time(e1))
        return sum(ms) / len(ms)

    def timeit_rot(fn, reps=args.reps):
        """fn(org_i, ref_i): a different plane pair every launch (see NP above)"""
        for i in range(3):
            fn(*pairs[i % NP])
        torch.cuda.synchronize()
        rot = [0]

        def one():
            fn(*pairs[rot[0] % NP])
            rot[0] += 1
        W.sustain_clocks(one, args.sustain_ms)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(reps):
            fn(*pairs[(3 + i) % NP])
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    def report(name, ms, pixels, abytes, extra=None, unique=None):
        gbs = abytes / (ms * 1e-3) / 1e9
        d = {"kernel": name, "ms": round(ms, 4), "Mpixels_s": round(pixels / (ms * 1e-3) / 1e6, 1),
             "algorithmic_GB_s": round(gbs, 1), "hbm_frac": round(gbs / HBM_PEAK, 4),
             "bit_depth": bd}
        if unique is not None:
            d["unique_GB_s"] = round(unique / (ms * 1e-3) / 1e9, 1)
            d["hbm_frac_unique"] = round(unique / (ms * 1e-3) / 1e9 / HBM_PEAK, 4)
            d["planes_rotated"] = NP
        if extra:
            d.update(extra)
        print(json.dumps(d), flush=True)

    def grid_cands(dtype, s, k, mv=32, fields=("ox", "oy", "rx", "ry")):
        nx, ny = fw // s, fh // s
        n = nx * ny * k
        c = np.zeros(n, dtype)
        bx = np.repeat(np.tile(np.arange(nx, dtype=np.int32) * s, ny), k)
        by = np.repeat(np.repeat(np.arange(ny, dtype=np.int32) * s, nx), k)
        if "ox" in fields:
            c["ox"], c["oy"] = bx, by
        c["rx"] = bx + rng.integers(-mv, mv + 1, n)
        c["ry"] = by + rng.integers(-mv, mv + 1, n)
        return c

    # ---- a1/a2: SAD / SATD, K=32 / K=8 offsets per block ----
    for s in (64, 32, 16, 8):
        for kind, k, nm in ((0, 32, "sad"), (1, 8, "satd")):
            c = grid_cands(api.DIST_CAND, s, k)
            dc = torch.from_numpy(c.view(np.uint8).reshape(-1).copy()).cuda()
            out = torch.empty(len(c), dtype=torch.int32, device="cuda")
            ms = timeit_rot(lambda o, r: ctx.dist_batch(kind, o, r, s, s, dc, n=len(c), out=out))
            report("get_%s %dx%d" % (nm, s, s), ms, len(c) * s * s, len(c) * (2 * s * s * bpp + 4),
                   {"candidates": len(c)}, unique=2 * vis_bytes + len(c) * 12)
    # ---- a3-a5: weighted SSE / cdef_dist with a scale grid ----
    scales = torch.from_numpy(rng.integers(1 << 12, 1 << 16, ((fh + 7) // 8, (fw + 7) // 8))
                              .astype(np.int32)).cuda()
    for s in (64, 16, 8):
        c = grid_cands(api.DIST_CAND, s, 4, mv=2)
        dc = torch.from_numpy(c.view(np.uint8).reshape(-1).copy()).cuda()
        out = torch.empty(len(c), dtype=torch.int64, device="cuda")
        for kind, nm in ((2, "weighted_sse"), (3, "cdef_dist")):
            ms = timeit_rot(lambda o, r: ctx.dist_scaled_batch(kind, o, r, s, s, dc, scales, n=len(c), out=out))
            report("%s %dx%d" % (nm, s, s), ms, len(c) * s * s,
                   len(c) * (2 * s * s * bpp + (s // 4) ** 2 * 4 + 8), {"candidates": len(c)},
                   unique=2 * vis_bytes + scales.numel() * 4 + len(c) * 16)
    # ---- a6 / a13 / a12: transforms and quantizer on every tx block of a frame ----
    for s, tsz in ((64, TxSize.TX_64X64), (32, TxSize.TX_32X32), (16, TxSize.TX_16X16),
                   (8, TxSize.TX_8X8), (4, TxSize.TX_4X4)):
        n = (fw // s) * (fh // s) * 4
        res = torch.randint(-255, 256, (n, s, s), dtype=torch.int16, device="cuda")
        co = torch.empty((n, s * s), dtype=ct, device="cuda")
        ms = timeit(lambda: ctx.forward_transform_batch(res, int(tsz), 0, bd, out=co))
        report("forward_transform %dx%d DCT_DCT" % (s, s), ms, n * s * s, n * s * s * (2 + cb),
               {"blocks": n})
        area = min(s, 32) ** 2
        ms = timeit(lambda: ctx.quantize_batch(co, int(tsz), 0, 100, bd, 0))
        report("quantize+dequantize %dx%d" % (s, s), ms, n * s * s, n * (area * cb + 2 * area * cb + 2),
               {"blocks": n})
        q = ctx.quantize_batch(co, int(tsz), 0, 100, bd, 0)
        pred = torch.randint(0, 1 << bd, (n, s, s), dtype=torch.int32, device="cuda").to(
            torch.uint8 if bpp == 1 else torch.int16)
        rec = torch.empty_like(pred)
        ms = timeit(lambda: ctx.inverse_transform_add_batch(q["rcoeffs"], pred, int(tsz), 0, bd, out=rec))
        report("inverse_transform_add %dx%d" % (s, s), ms, n * s * s,
               n * (area * cb + 2 * s * s * bpp), {"blocks": n})
    # ---- a8 / a9: put_8tap / prep_8tap (all 16x16 fractions, REGULAR) ----
    for s in (64, 16, 8):
        c = grid_cands(api.MC_CAND, s, 8, fields=())
        c["col_frac"] = rng.integers(0, 16, len(c))
        c["row_frac"] = rng.integers(0, 16, len(c))
        dc = torch.from_numpy(c.view(np.uint8).reshape(-1).copy()).cuda()
        out = torch.empty((len(c), s, s), dtype=torch.uint8 if bpp == 1 else torch.int16, device="cuda")
        ms = timeit_rot(lambda o, r: ctx.put_8tap_batch(r, s, s, dc, n=len(c), out=out))
        report("put_8tap %dx%d" % (s, s), ms, len(c) * s * s,
               len(c) * (((s + 7) ** 2 + s * s) * bpp), {"candidates": len(c)},
               unique=vis_bytes + len(c) * (8 + s * s * bpp))
        out16 = torch.empty((len(c), s, s), dtype=torch.int16, device="cuda")
        ms = timeit_rot(lambda o, r: ctx.prep_8tap_batch(r, s, s, dc, n=len(c), out=out16))
        report("prep_8tap %dx%d" % (s, s), ms, len(c) * s * s,
               len(c) * ((s + 7) ** 2 * bpp + 2 * s * s), {"candidates": len(c)},
               unique=vis_bytes + len(c) * (8 + 2 * s * s))
    # ---- a10 / a11: intra edges + prediction on every tx block ----
    for s, tsz in ((32, TxSize.TX_32X32), (16, TxSize.TX_16X16), (8, TxSize.TX_8X8), (4, TxSize.TX_4X4)):
        nx, ny = fw // s, fh // s
        n = nx * ny
        ec = np.zeros(n, api.INTRA_EDGE_CAND)
        ec["x"] = np.tile(np.arange(nx) * s, ny)
        ec["y"] = np.repeat(np.arange(ny) * s, nx)
        ec["mode"] = rng.integers(0, 13, n)
        ec["flags"] = 7
        dec = torch.from_numpy(ec.view(np.uint8).reshape(-1).copy()).cuda()
        ms = timeit(lambda: ctx.intra_edges_batch(ref, (0, 0, fw, fh), int(tsz), dec, n=n))
        report("get_intra_edges %dx%d" % (s, s), ms, n * s * s, n * (2 * (2 * s) + 1) * bpp * 2,
               {"blocks": n})
        edges, lens = ctx.intra_edges_batch(ref, (0, 0, fw, fh), int(tsz), dec, n=n)
        ic = np.zeros(n, api.INTRA_CAND)
        pm = ec["mode"].astype(np.int32)
        var = np.where((ec["x"] == 0) & (ec["y"] == 0), 0, np.where(ec["y"] == 0, 1,
                                                                     np.where(ec["x"] == 0, 2, 3)))
        pm = np.where((pm == 12) & (var == 0), 0, np.where((pm == 12) & (var == 2), 1,
                      np.where((pm == 12) & (var == 1), 2, pm)))
        ic["mode"], ic["variant"] = pm, var
        ic["angle"] = np.array([0, 90, 180, 45, 135, 113, 157, 203, 67] + [0] * 5)[pm]
        ic["ief"] = 1
        ic["avail_w"] = ic["avail_h"] = s
        dic = torch.from_numpy(ic.view(np.uint8).reshape(-1).copy()).cuda()
        ms = timeit(lambda: ctx.predict_intra_batch(int(tsz), dic, edges, lens, bd, n=n))
        report("predict_intra %dx%d (mixed modes)" % (s, s), ms, n * s * s,
               n * ((2 * (2 * s) + 1) * bpp + s * s * bpp), {"blocks": n})
    # ---- N1: intra mode pre-screen, 13 modes per block in one launch ----
    for s, tsz in ((32, TxSize.TX_32X32), (16, TxSize.TX_16X16), (8, TxSize.TX_8X8)):
        nx, ny = fw // s, fh // s
        nb = nx * ny
        ec = np.zeros(nb, api.INTRA_EDGE_CAND)
        ec["x"] = np.tile(np.arange(nx) * s, ny)
        ec["y"] = np.repeat(np.arange(ny) * s, nx)
        ec["mode"] = -1
        ec["flags"] = 7
        edges, lens = ctx.intra_edges_batch(ref, (0, 0, fw, fh), int(tsz), ec)
        var = np.where((ec["x"] == 0) & (ec["y"] == 0), 0, np.where(ec["y"] == 0, 1,
                                                                     np.where(ec["x"] == 0, 2, 3)))
        pm = np.tile(np.arange(13), nb)
        v13 = np.repeat(var, 13)
        pm = np.where((pm == 12) & (v13 == 0), 0, np.where((pm == 12) & (v13 == 2), 1,
                      np.where((pm == 12) & (v13 == 1), 2, pm)))
        ic = np.zeros(nb * 13, api.INTRA_CAND)
        ic["mode"], ic["variant"] = pm, v13
        ic["angle"] = np.array([0, 90, 180, 45, 135, 113, 157, 203, 67, 0, 0, 0, 0])[pm]
        ic["ief"] = np.where((pm >= 1) & (pm <= 8), 1, 0)
        ic["avail_w"] = ic["avail_h"] = s
        dic = torch.from_numpy(ic.view(np.uint8).reshape(-1).copy()).cuda()
        pos = torch.from_numpy(np.stack([ec["x"], ec["y"]], 1).astype(np.int16)).cuda()
        ms = timeit(lambda: ctx.intra_satd_batch(org, int(tsz), dic, 13, pos, edges, lens, n=nb * 13))
        report("intra pre-screen %dx%d (13 modes: predict + SATD, fused)" % (s, s), ms, nb * 13 * s * s,
               nb * ((2 * (2 * s) + 1) * bpp + s * s * bpp + 13 * 4), {"blocks": nb})
    # ---- a14: whole-frame CDEF, luma ----
    dst = Plane(fw, fh, bd)
    skip = torch.zeros((fh // 4, fw // 4), dtype=torch.uint8, device="cuda")
    ci = torch.zeros(((fh + 63) // 64, (fw + 63) // 64), dtype=torch.uint8, device="cuda")
    ystr = [36] * 8
    ms = timeit(lambda: ctx.cdef_filter_frame_plane(ref, ref, dst, 0, 0, 0, fw, fh, skip, ci, ystr,
                                                    ystr, 5, bd))
    report("cdef_filter_tile luma (find_dir + filter)", ms, fw * fh, 2 * fw * fh * bpp)
    # ---- a14: the CDEF strength search of rdo_loop_decision, 4:2:0 frame, rav1e's 8 presets ----
    cw2, ch2 = fw // 2, fh // 2
    ch_rec = [Plane.from_numpy(W.random_plane_array(cw2, ch2, bd, 30 + k, 44, 44), cw2, ch2, bd, 44, 44)
              for k in range(2)]
    ch_src = [Plane.from_numpy(W.random_plane_array(cw2, ch2, bd, 40 + k, 44, 44), cw2, ch2, bd, 44, 44)
              for k in range(2)]
    presets = [0 * 4 + 0, 1 * 4 + 0, 2 * 4 + 1, 3 * 4 + 1, 5 * 4 + 2, 7 * 4 + 3, 10 * 4 + 3, 13 * 4 + 3]   # encoder.rs:897-916
    skip_s = torch.zeros((2 * ((fh + 7) // 8), 2 * ((fw + 7) // 8)), dtype=torch.uint8, device="cuda")
    scl = torch.full(((fh + 7) // 8, (fw + 7) // 8), 1 << 14, dtype=torch.int32, device="cuda")
    for n_idx in (8, 1):
        ms = timeit(lambda: ctx.cdef_strength_search([ref, ch_rec[0], ch_rec[1]], [org, ch_src[0], ch_src[1]], skip_s,
                                                     presets, presets, 5, bd, n_idx, 1, 1, fw, fh, scales=scl))
        report("cdef strength search 4:2:0, %d cdef_index candidates (filter + distortion per candidate, "
               "no plane written)" % n_idx, ms, fw * fh * 3 // 2 * n_idx, 2 * (fw * fh * 3 // 2) * bpp,
               {"pixels_filtered_and_measured": fw * fh * 3 // 2 * n_idx})
    # ---- N3: deblock filter + level search, 4:2:0 frame (luma + two chroma planes) ----
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import deblock_util as D
    blocks = D.random_blocks(rng, fw // 4, (fh + 3) // 4, 1, 1)
    dblocks = torch.from_numpy(blocks.view(np.uint8).reshape(blocks.shape + (8,)).copy()).cuda()
    state = D.make_state([24, 20, 16, 16])
    planes3 = [(org, ref, 0, 0, 0), ]
    cw_, ch_ = fw // 2, fh // 2
    for pli in (1, 2):
        a = Plane.from_numpy(W.random_plane_array(cw_, ch_, bd, 10 + pli, 44, 44), cw_, ch_, bd, 44, 44)
        b = Plane.from_numpy(W.random_plane_array(cw_, ch_, bd, 20 + pli, 44, 44), cw_, ch_, bd, 44, 44)
        planes3.append((a, b, pli, 1, 1))

    rec3, src3 = [a for (a, b, pli, xd, yd) in planes3], [b for (a, b, pli, xd, yd) in planes3]

    def run_filter():          # the frame entry point (one call, what tools/frame_pipeline.py times)
        ctx.deblock_frame(state, rec3, 1, 1, dblocks, fw, fh)

    def run_filter_planes():   # the per-plane entry point, three calls (what this row timed up to round 3)
        for (a, b, pli, xd, yd) in planes3:
            ctx.deblock_plane(state, a, pli, xd, yd, dblocks, fw, fh)

    tall = torch.zeros((3, 2, 65), dtype=torch.int64, device="cuda")

    def run_sse():
        ctx.deblock_sse_frame(rec3, src3, 1, 1, dblocks, fw, fh, tallies=tall)

    def run_sse_planes():
        for (a, b, pli, xd, yd) in planes3:
            ctx.deblock_sse_plane(a, b, pli, xd, yd, dblocks, fw, fh, tallies=tall[pli])
    npx = fw * fh * 3 // 2
    fresh = [a.data.clone() for (a, b, pli, xd, yd) in planes3]

    def restore():
        for (a, b, pli, xd, yd), f0 in zip(planes3, fresh):
            a.data.copy_(f0)
    ms = timeit_fresh(run_filter, restore)
    report("deblock_filter_frame 4:2:0 (3 planes, in place; fresh pixels every rep)", ms, npx,
           2 * npx * bpp + blocks.size * 8)
    ms = timeit_fresh(run_filter_planes, restore)
    report("deblock_filter_frame 4:2:0 through the per-plane entry point (3 calls)", ms, npx,
           2 * npx * bpp + blocks.size * 8)
    restore()
    ms = timeit(run_sse)
    report("deblock sse_optimize tallies 4:2:0 (3 planes, frame entry point)", ms, npx, 2 * npx * bpp + blocks.size * 8)
    ms = timeit(run_sse_planes)
    report("deblock sse_optimize tallies 4:2:0 through the per-plane entry point (3 calls)", ms, npx,
           2 * npx * bpp + blocks.size * 8)
    # ---- N3 (last stage): self-guided loop restoration, luma plane, every unit filtered ----
    us = 64
    ucols, urows = max((fw + us // 2) // us, 1), max((fh + us // 2) // us, 1)
    units = np.zeros((urows, ucols, 4), np.uint8)
    units[..., 0] = 3
    units[..., 1] = rng.integers(0, 16, (urows, ucols))
    units[..., 2] = rng.integers(-96, 32, (urows, ucols)).astype(np.int8).view(np.uint8)
    units[..., 3] = rng.integers(-32, 96, (urows, ucols)).astype(np.int8).view(np.uint8)
    dunits = torch.from_numpy(units).cuda()
    lrf_out = Plane(fw, fh, bd)
    lrf_out.data.copy_(ref.data)
    ms = timeit(lambda: ctx.lrf_sgrproj_plane(ref, org, lrf_out, 0, fw, fh, fh, us, dunits, 64))
    report("lrf sgrproj luma (all units, mixed sets)", ms, fw * fh, 3 * fw * fh * bpp)
    # ---- N3: the restoration leg of rdo_loop_decision but the rate (rdo.rs:2575-2763): per 64x64 luma /
    # 32x32 chroma unit of a 4:2:0 frame the no-filter error + (solve, filter, error) for the speed-6 list of
    # parameter sets (SGRPROJ_REDUCED_SETS, lrf.rs:86; speed >= 5, speedsettings.rs:143-144) ----
    sets = [255, 1, 3, 5, 7, 9, 11, 13, 15]
    scl = torch.from_numpy(np.random.default_rng(9).integers(1 << 12, 1 << 16, ((fh + 7) // 8, (fw + 7) // 8))
                           .astype(np.int32)).cuda()
    cw, ch = fw // 2, fh // 2
    cin = [Plane.from_numpy(W.random_plane_array(cw, ch, bd, 40 + i, 44, 44), cw, ch, bd, 44, 44) for i in range(4)]

    def unit_list(pw, ph, us_):
        u = [(x, y, min(us_, pw - x), min(us_, ph - y), s_, 0, (0, 0))
             for y in range(0, ph, us_) for x in range(0, pw, us_) for s_ in sets]
        return torch.from_numpy(np.array(u, api.SGR_SOLVE_UNIT).view(np.uint8).reshape(-1).copy()).cuda()
    ul, uc = unit_list(fw, fh, 64), unit_list(cw, ch, 32)

    def lrf_search():
        ctx.lrf_search_batch(ref, org, ul, scales=scl, max_w=64, max_h=64)
        ctx.lrf_search_batch(cin[0], cin[1], uc, is_chroma=True, xdec=1, ydec=1, scales=scl, max_w=32, max_h=32)
        ctx.lrf_search_batch(cin[2], cin[3], uc, is_chroma=True, xdec=1, ydec=1, scales=scl, max_w=32, max_h=32)
    ms = timeit(lrf_search)
    npairs = (ul.numel() + 2 * uc.numel()) // api.SGR_SOLVE_UNIT.itemsize
    report("lrf search 4:2:0 (no filter + 8 parameter sets per unit: solve, filter, error)", ms,
           (fw * fh + 2 * cw * ch) * len(sets), (fw * fh + 2 * cw * ch) * 2 * bpp + npairs * 10, {"pairs": npairs})
    ctx.close()


if __name__ == "__main__":
    main()
