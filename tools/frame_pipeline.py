#!/usr/bin/env python3
"""One coded 4K frame's device-resident work, end to end, with every piece of this backend:

  lookahead cost maps -> hierarchical ME (tiles x references) -> block importances ->
  RDO-time sub-pel ME ->
  intra pre-screen -> RDO candidates (pixel-domain chain) -> deblock level search ->
  deblock -> CDEF -> loop restoration

Nothing returns to the host between the stages except the scalars a real encoder's control
flow needs; the stage inputs are synthetic (the encoder's decisions are not modelled), the
sizes are those of a 4K speed-6 frame.  Prints one JSON line with the per-stage device times.

    python tools/frame_pipeline.py [--bit-depth 8] [--reps 5]
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bit-depth", type=int, default=8)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--sustain-ms", type=float, default=100.0,
                    help="untimed passes of a stage for this long before it is timed (sustained clocks); 0 = off")
    args = ap.parse_args()
    import torch
    import deblock_util as D
    import oracle_lib as O
    from rav1e_amd import api, tiles, workload as W
    from rav1e_amd.api import Context, Plane, me_lambdas
    from rav1e_amd.types import TxSize
    fw, fh, bd = 3840, 2160, args.bit_depth
    rng = np.random.default_rng(0)
    ctx = Context(0)
    # band-limited source, references = shifted + noisy copies
    f = rng.standard_normal((fh + 64, fw + 64)).astype(np.float32)
    for _ in range(3):
        f = (np.roll(f, 1, 0) + 2 * f + np.roll(f, -1, 0)) / 4
        f = (np.roll(f, 1, 1) + 2 * f + np.roll(f, -1, 1)) / 4
    f = ((f - f.min()) / (f.max() - f.min()) * ((1 << bd) - 1)).astype(np.int64)
    org_img = f[32:32 + fh, 32:32 + fw]
    refs_img = [np.clip(f[32 + dy:32 + dy + fh, 32 + dx:32 + dx + fw] + rng.integers(-2, 3, (fh, fw)), 0,
                        (1 << bd) - 1) for dx, dy in ((5, -9), (-3, 2), (12, 7))]
    dev = lambda pyr: [Plane.from_numpy(p.data, p.width, p.height, bd, p.xpad, p.ypad) for p in pyr]
    org = dev(O.me_pyramid(org_img, bd))
    refs = [dev(O.me_pyramid(r, bd)) for r in refs_img]
    rows, cols = fh // 4, fw // 4
    lam = me_lambdas(30.0)
    stages = {}

    def timed(name, fn):
        fn()
        torch.cuda.synchronize()
        W.sustain_clocks(fn, args.sustain_ms)
        t0 = time.perf_counter()
        for _ in range(args.reps):
            fn()
        torch.cuda.synchronize()
        stages[name] = round((time.perf_counter() - t0) / args.reps * 1e3, 3)
        if os.environ.get("R1_VERBOSE"):
            print(name, stages[name], file=sys.stderr, flush=True)

    # 1 lookahead cost maps
    timed("lookahead_intra_costs", lambda: ctx.estimate_intra_costs(org[0]))
    # 2 hierarchical ME: 8 tiles x 3 references
    rects = W.tile_rects(8, fw, fh)
    stats = [torch.zeros((rows, cols, 2), dtype=torch.int32, device="cuda") for _ in refs]
    jobs = [dict(org=org, ref=refs[r], stats=stats[r], tile=(x0, y0, x1 - x0, y1 - y0))
            for r in range(len(refs)) for (x0, y0, x1, y1) in rects]
    timed("estimate_tile_motion_8tiles_x_3refs", lambda: ctx.estimate_tile_motion(jobs, cols, rows, bd, lam))
    assert ctx.me_status(wait=True)[0], "a persistent tile-ME launch flagged a timed-out wait"
    # 2b block importances: update_block_importances over the three references -- SATD map at
    # the ME's vectors (every second MEStats entry), then the f32 propagation
    hb, wb = fh // 8, fw // 8
    intra_costs = ctx.estimate_intra_costs(org[0]).reshape(-1)
    future = torch.zeros(hb * wb, dtype=torch.float32, device="cuda")
    ref_imp = [torch.zeros(hb * wb, dtype=torch.float32, device="cuda") for _ in refs]

    def importances():
        for r in range(len(refs)):
            mv = stats[r].view(torch.int16).reshape(rows, cols, 4)[0:2 * hb:2, 0:2 * wb:2, 0:2].contiguous()
            inter = ctx.estimate_inter_costs(org[0], refs[r][0], mv)
            ctx.update_block_importances(intra_costs, future, inter.reshape(-1), mv, wb, hb, len(refs),
                                         ref_imp[r])
    timed("update_block_importances_3refs", importances)
    # 3 RDO-time sub-pel ME on every 16x16 block, first reference
    c = np.zeros((fw // 16) * (fh // 16), api.ME_BLOCK_CAND)
    c["bx"] = np.tile(np.arange(fw // 16) * 4, fh // 16)
    c["by"] = np.repeat(np.arange(fh // 16) * 4, fw // 16)
    c["w"] = c["h"] = 16
    c["corner"] = 7
    dc = torch.from_numpy(c.view(np.uint8).reshape(-1).copy()).cuda()
    job0 = dict(org=org, ref=refs[0], stats=stats[0], tile=(0, 0, fw, fh))
    timed("estimate_motion_subpel_16x16_all_blocks",
          lambda: ctx.estimate_motion_batch(job0, dc, cols, rows, bd, lam, max_w=16, max_h=16, n=len(c)))
    # 4 intra pre-screen, 13 modes on every 16x16 block
    s = 16
    nb = (fw // s) * (fh // s)
    ec = np.zeros(nb, api.INTRA_EDGE_CAND)
    ec["x"] = np.tile(np.arange(fw // s) * s, fh // s)
    ec["y"] = np.repeat(np.arange(fh // s) * s, fw // s)
    ec["mode"], ec["flags"] = -1, 7
    var = np.where((ec["x"] == 0) & (ec["y"] == 0), 0, np.where(ec["y"] == 0, 1, np.where(ec["x"] == 0, 2, 3)))
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
    dec = torch.from_numpy(ec.view(np.uint8).reshape(-1).copy()).cuda()

    def prescreen():
        edges, lens = ctx.intra_edges_batch(refs[0][0], (0, 0, fw, fh), int(TxSize.TX_16X16), dec, n=nb)
        ctx.intra_satd_batch(org[0], int(TxSize.TX_16X16), dic, 13, pos, edges, lens, n=nb * 13)
    timed("intra_prescreen_16x16_13modes", prescreen)
    # 5 RDO candidates, pixel-domain chain, speed-6 ladder, K = 16 per block
    cands = tiles.shard_candidates(fw, fh, 16, 0, 1)
    dcands = {k: torch.from_numpy(v.view(np.uint8).reshape(-1).copy()).cuda() for k, v in cands.items()}
    scales = torch.from_numpy(rng.integers(1 << 12, 1 << 16, ((fh + 7) // 8, (fw + 7) // 8)).astype(np.int32)).cuda()
    outs = {k: {} for k in cands}

    def rdo():
        for k in W.LADDER:
            ctx.rdo_pixel_cand_batch(org[0], refs[0][0], k, k, dcands[k], 100, 3, scales=scales,
                                     n=len(cands[k]), outs=outs[k])
    timed("rdo_pixel_candidates_ladder_K16", rdo)
    # 6-8 post filters on a 4:2:0 frame
    blocks = D.random_blocks(rng, cols, rows, 1, 1)
    dblocks = torch.from_numpy(blocks.view(np.uint8).reshape(blocks.shape + (8,)).copy()).cuda()
    state = D.make_state([24, 20, 16, 16])
    cw, ch = fw // 2, fh // 2
    chroma = [Plane.from_numpy(W.random_plane_array(cw, ch, bd, 30 + i, 44, 44), cw, ch, bd, 44, 44) for i in range(4)]
    planes3 = [(refs[0][0], org[0], 0, 0, 0), (chroma[0], chroma[1], 1, 1, 1), (chroma[2], chroma[3], 2, 1, 1)]
    tall = torch.zeros((3, 2, 65), dtype=torch.int64, device="cuda")
    rec3, src3 = [a for (a, b, p, xd, yd) in planes3], [b for (a, b, p, xd, yd) in planes3]
    timed("deblock_level_search_420", lambda: ctx.deblock_sse_frame(rec3, src3, 1, 1, dblocks, fw, fh, tallies=tall))
    timed("deblock_filter_420", lambda: ctx.deblock_frame(state, rec3, 1, 1, dblocks, fw, fh))
    dst = Plane(fw, fh, bd)
    skip = torch.zeros((fh // 4, fw // 4), dtype=torch.uint8, device="cuda")
    ci = torch.zeros(((fh + 63) // 64, (fw + 63) // 64), dtype=torch.uint8, device="cuda")
    timed("cdef_luma", lambda: ctx.cdef_filter_frame_plane(refs[0][0], refs[0][0], dst, 0, 0, 0, fw, fh, skip, ci,
                                                           [36] * 8, [36] * 8, 5, bd))
    # 8b the CDEF strength search of rdo_loop_decision (8 presets, 4:2:0) -- not part of the sum: rav1e's
    # default picks the strength from the quantizer (cdef_bits 0); reported beside the stages
    skip_s = torch.zeros((2 * ((fh + 7) // 8), 2 * ((fw + 7) // 8)), dtype=torch.uint8, device="cuda")
    presets = [0, 4, 9, 13, 22, 31, 43, 55]    # encoder.rs:897-916
    cdef_search = lambda: ctx.cdef_strength_search(rec3, src3, skip_s, presets, presets, 5, bd, 8, 1, 1, fw, fh,
                                                   scales=scales)
    for _ in range(2):
        cdef_search()
    torch.cuda.synchronize()
    W.sustain_clocks(cdef_search, args.sustain_ms)
    t0 = time.perf_counter()
    for _ in range(args.reps):
        cdef_search()
    torch.cuda.synchronize()
    cdef_search_ms = round((time.perf_counter() - t0) / args.reps * 1e3, 3)
    # the restoration search of rdo_loop_decision (rdo.rs:2575-2763) but the rate: per 64x64 luma / 32x32
    # chroma unit the no-filter error + (solve, filter, error) for the 8 parameter sets of speed >= 5
    sets = [255, 1, 3, 5, 7, 9, 11, 13, 15]

    def unit_list(pw, ph, us_):
        u = [(x, y, min(us_, pw - x), min(us_, ph - y), s_, (0, 0, 0))
             for y in range(0, ph, us_) for x in range(0, pw, us_) for s_ in sets]
        return torch.from_numpy(np.array(u, api.SGR_SOLVE_UNIT).view(np.uint8).reshape(-1).copy()).cuda()
    ul, uc = unit_list(fw, fh, 64), unit_list(fw // 2, fh // 2, 32)

    def lrf_search():
        ctx.lrf_search_batch(rec3[0], src3[0], ul, scales=scales, max_w=64, max_h=64)
        for pl in (1, 2):
            ctx.lrf_search_batch(rec3[pl], src3[pl], uc, is_chroma=True, xdec=1, ydec=1, scales=scales, max_w=32,
                                 max_h=32)
    lrf_search()
    torch.cuda.synchronize()
    W.sustain_clocks(lrf_search, args.sustain_ms)
    t0 = time.perf_counter()
    for _ in range(args.reps):
        lrf_search()
    torch.cuda.synchronize()
    lrf_search_ms = round((time.perf_counter() - t0) / args.reps * 1e3, 3)
    # 9 loop restoration (self-guided), luma, every 64x64 unit
    us = 64
    units = np.zeros((max((fh + 32) // us, 1), max((fw + 32) // us, 1), 4), np.uint8)
    units[..., 0] = 3
    units[..., 1] = rng.integers(0, 16, units.shape[:2])
    units[..., 2] = rng.integers(-96, 32, units.shape[:2]).astype(np.int8).view(np.uint8)
    units[..., 3] = rng.integers(-32, 96, units.shape[:2]).astype(np.int8).view(np.uint8)
    dunits = torch.from_numpy(units).cuda()
    lrf_out = Plane(fw, fh, bd)
    timed("lrf_sgrproj_luma", lambda: ctx.lrf_sgrproj_plane(dst, refs[0][0], lrf_out, 0, fw, fh, fh, us, dunits, 64))
    total = round(sum(stages.values()), 3)
    # The wavefront-bound ME of frame N+1 leaves most CUs idle: run it on a second stream next to
    # the throughput-bound stages of frame N (they touch different buffers).
    s_me, s_rdo = torch.cuda.Stream(), torch.cuda.Stream()

    def overlapped():
        with torch.cuda.stream(s_me):
            ctx.estimate_tile_motion(jobs, cols, rows, bd, lam)
            importances()
        with torch.cuda.stream(s_rdo):
            ctx.estimate_motion_batch(job0, dc, cols, rows, bd, lam, max_w=16, max_h=16, n=len(c))
            prescreen()
            rdo()
            ctx.deblock_sse_frame(rec3, src3, 1, 1, dblocks, fw, fh, tallies=tall)
            ctx.deblock_frame(state, rec3, 1, 1, dblocks, fw, fh)
            ctx.cdef_filter_frame_plane(refs[0][0], refs[0][0], dst, 0, 0, 0, fw, fh, skip, ci, [36] * 8,
                                        [36] * 8, 5, bd)
            ctx.lrf_sgrproj_plane(dst, refs[0][0], lrf_out, 0, fw, fh, fh, us, dunits, 64)
    torch.cuda.synchronize()
    overlapped()
    torch.cuda.synchronize()
    W.sustain_clocks(overlapped, args.sustain_ms)
    t0 = time.perf_counter()
    for _ in range(args.reps):
        overlapped()
    torch.cuda.synchronize()
    ov = round((time.perf_counter() - t0) / args.reps * 1e3, 3)
    print(json.dumps({"frame": "%dx%d %d-bit" % (fw, fh, bd), "stage_ms": stages, "sum_ms": total,
                      "frames_per_s_if_serial": round(1e3 / total, 1),
                      "cdef_strength_search_8_presets_420_ms (optional stage, not in the sum)": cdef_search_ms,
                      "lrf_search_8_sets_420_ms (optional stage, not in the sum)": lrf_search_ms,
                      "two_stream_ms (ME of the next frame beside the other stages)": ov,
                      "frames_per_s_two_streams": round(1e3 / ov, 1)}))
    ctx.close()


if __name__ == "__main__":
    main()
