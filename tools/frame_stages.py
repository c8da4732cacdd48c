#!/usr/bin/env python3
"""The device-resident work of ONE coded 3840x2160 4:2:0 frame (BASELINE.json configs[3]: "4K 10-bit 4:2:0
speed-4 full RDO + CDEF, all kernels"), stage by stage, with every piece of this backend:

  lookahead cost maps -> hierarchical ME (8 tiles x 3 references) -> block importances ->
  RDO-time sub-pel ME -> intra pre-screen (13 modes) ->
  RDO candidates, pixel-domain chain: luma ladder + both chroma planes ->
  transform-type search (7 RAV1E_TX_TYPES on one prediction per 16x16 / 8x8 block) ->
  deblock level search -> deblock -> CDEF strength search -> CDEF -> restoration search ->
  rdo_loop_decision's iteration with both filters on (working copy, restoration leg on it, second pass of both legs) ->
  loop restoration

build() returns the stages as callables on resident inputs plus, for bench.py, a strided parity sample per stage
against the CPU oracle (tests/oracle_lib.py; test infrastructure, outside every timed region).  Stage inputs are
synthetic (the encoder's decisions are not modelled); sizes are those of a 4K frame with K candidates per block.
Used by tools/frame_pipeline.py (wall-clock per stage + the two-stream plan) and bench.py::config_lines
(`config4_frame_4k_10bit`, HIP events per stage, inside the driver's run)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

PRESETS = [0, 4, 9, 13, 22, 31, 43, 55]      # fi.cdef_y_strengths / cdef_uv_strengths, src/encoder.rs:897-916
LRF_SETS = [255, 1, 3, 5, 7, 9, 11, 13, 15]  # no filter + the 8 parameter sets of speed >= 5 (src/rdo.rs:2575-2763)


def build(ctx, bd, fw=3840, fh=2160, k=16, qindex=100, seed=0):
    import torch
    import deblock_util as D
    import oracle_lib as O
    from rav1e_amd import api, rdo_glue as RG, tiles, workload as W
    from rav1e_amd.api import Plane, me_lambdas
    from rav1e_amd.types import TxSize
    rng = np.random.default_rng(seed)
    dt = np.uint8 if bd == 8 else np.uint16
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
    h_org = O.me_pyramid(org_img, bd)
    h_refs = [O.me_pyramid(r, bd) for r in refs_img]
    org, refs = dev(h_org), [dev(r) for r in h_refs]
    rows, cols = fh // 4, fw // 4
    lam = me_lambdas(30.0)
    stages, optional, checks = [], [], {}
    TS = {64: 4, 32: 3, 16: 2, 8: 1, 4: 0}

    # 0 (untimed) the deblocking stage filters the reconstruction IN PLACE and its run time depends on what it finds
    # (an already filtered plane takes fewer taps): every pass starts from the pristine reconstruction.  Consumers drop
    # stages whose name ends in "_untimed" from their sums.
    restore = []
    stages.append(("restore_reconstruction_untimed", lambda: [p.data.copy_(b) for p, b in restore]))
    # 1 lookahead cost maps
    stages.append(("lookahead_intra_costs", lambda: ctx.estimate_intra_costs(org[0])))
    # 2 hierarchical ME: 8 tiles x 3 references
    rects = W.tile_rects(8, fw, fh)
    stats = [torch.zeros((rows, cols, 2), dtype=torch.int32, device="cuda") for _ in refs]
    jobs = [dict(org=org, ref=refs[r], stats=stats[r], tile=(x0, y0, x1 - x0, y1 - y0))
            for r in range(len(refs)) for (x0, y0, x1, y1) in rects]

    def tile_me():
        ctx.estimate_tile_motion(jobs, cols, rows, bd, lam)
    stages.append(("estimate_tile_motion_8tiles_x_3refs", tile_me))

    def chk_tile_me():
        # one (tile, reference) job of the concurrent launch against the oracle: every MEStats entry of the tile
        assert ctx.me_status(wait=True)[0], "a persistent tile-ME launch flagged a timed-out wait"
        for s in stats:
            s.zero_()
        tile_me()
        L = O.lib()
        L.r1o_set_threads(os.cpu_count() or 1)
        x0, y0, x1, y1 = rects[3]
        st = np.zeros((rows, cols), O.ME_STATS)
        O.me_oracle(L, h_org, h_refs[1], cols, rows, (x0, y0, x1 - x0, y1 - y0), bd, lam, st)
        got = stats[1].cpu().numpy().reshape(rows, -1).view(O.ME_STATS).reshape(rows, cols)
        sub = (slice(y0 // 4, y1 // 4), slice(x0 // 4, x1 // 4))
        return int(st[sub].size), bool(np.array_equal(got[sub], st[sub]))
    checks["estimate_tile_motion_8tiles_x_3refs"] = chk_tile_me
    # 2b block importances over the three references
    hb, wb = fh // 8, fw // 8
    intra_costs = ctx.estimate_intra_costs(org[0]).reshape(-1)
    future = torch.zeros(hb * wb, dtype=torch.float32, device="cuda")
    ref_imp = [torch.zeros(hb * wb, dtype=torch.float32, device="cuda") for _ in refs]

    def importances():
        for r in range(len(refs)):
            mv = stats[r].view(torch.int16).reshape(rows, cols, 4)[0:2 * hb:2, 0:2 * wb:2, 0:2].contiguous()
            inter = ctx.estimate_inter_costs(org[0], refs[r][0], mv)
            ctx.update_block_importances(intra_costs, future, inter.reshape(-1), mv, wb, hb, len(refs), ref_imp[r])
    stages.append(("update_block_importances_3refs", importances))
    # 3 RDO-time sub-pel ME on every 16x16 block, first reference
    c = np.zeros((fw // 16) * (fh // 16), api.ME_BLOCK_CAND)
    c["bx"] = np.tile(np.arange(fw // 16) * 4, fh // 16)
    c["by"] = np.repeat(np.arange(fh // 16) * 4, fw // 16)
    c["w"] = c["h"] = 16
    c["corner"] = 7
    dc = torch.from_numpy(c.view(np.uint8).reshape(-1).copy()).cuda()
    job0 = dict(org=org, ref=refs[0], stats=stats[0], tile=(0, 0, fw, fh))
    stages.append(("estimate_motion_subpel_16x16_all_blocks",
                   lambda: ctx.estimate_motion_batch(job0, dc, cols, rows, bd, lam, max_w=16, max_h=16, n=len(c))))

    def chk_subpel():
        L = O.lib()
        L.r1o_set_threads(os.cpu_count() or 1)
        hst = stats[0].cpu().numpy().reshape(rows, -1).view(O.ME_STATS).reshape(rows, cols)
        idx = np.arange(0, len(c), max(1, len(c) // 96))[:96]
        sub = np.ascontiguousarray(c[idx])
        want = O.me_block_oracle(L, h_org, h_refs[0], cols, rows, (0, 0, fw, fh), bd, lam, hst, None, sub)
        got = ctx.estimate_motion_batch(job0, sub, cols, rows, bd, lam, max_w=16, max_h=16)
        return len(sub), bool(np.array_equal(got.cpu().numpy().view(api.ME_RESULT), want))
    checks["estimate_motion_subpel_16x16_all_blocks"] = chk_subpel
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
        return ctx.intra_satd_batch(org[0], int(TxSize.TX_16X16), dic, 13, pos, edges, lens, n=nb * 13)
    stages.append(("intra_prescreen_16x16_13modes", prescreen))
    # 5 RDO candidates, pixel-domain chain: luma ladder, K per block
    cands = tiles.shard_candidates(fw, fh, k, 0, 1)
    dcands = {kk: torch.from_numpy(v.view(np.uint8).reshape(-1).copy()).cuda() for kk, v in cands.items()}
    h_scales = rng.integers(1 << 12, 1 << 16, ((fh + 7) // 8, (fw + 7) // 8)).astype(np.uint32)
    scales = torch.from_numpy(h_scales.view(np.int32)).cuda()
    outs = {kk: {} for kk in cands}

    for kk in W.LADDER:      # one launch per block size: each is a stage of its own (bench.py keys its counters by launch)
        stages.append(("rdo_pixel_luma_%dx%d_K%d" % (kk, kk, k),
                       # (round 6: no SAD / SATD of the candidate here, as on the chroma planes since round 5 --
                       # rdo_tx_size_type -> encode_tx_block -> compute_distortion never takes them, src/rdo.rs:1073,
                       # src/encoder.rs:1404-1661; they belong to the motion search.  bench.py --chain pixel keeps both
                       # figures: `value` with them, `rdo_only` without)
                       lambda kk=kk: ctx.rdo_pixel_cand_batch(org[0], refs[0][0], kk, kk, dcands[kk], qindex, 3, scales=scales,
                                                              n=len(cands[kk]), outs=outs[kk], want_sad=False,
                                                              want_satd=False)))

    def sample(n, m=16):
        return np.arange(0, n, max(1, n // m))[:m]

    def chk_rdo(kk):
        L = O.lib()
        L.r1o_set_threads(os.cpu_count() or 1)
        pa, pb = h_org[0].cstruct(), h_refs[0][0].cstruct()
        n_chk, ok = 0, True
        if True:
            idx = sample(len(cands[kk]))
            sub = np.ascontiguousarray(cands[kk][idx])
            eob, dist = np.zeros(len(sub), np.uint16), np.zeros(len(sub), np.uint64)
            assert L.r1o_rdo_pixel_cand_batch(C.byref(pa), C.byref(pb), kk, kk, TS[kk], O.ptr(sub), len(sub), qindex, 0, 0, 0,
                                              3, O.ptr(h_scales), h_scales.shape[1], 0, 0, None, None,
                                              O.ptr(eob), O.ptr(dist), None, None, None) == 0
            ix = torch.from_numpy(idx.astype(np.int64)).cuda()
            n_chk += len(idx)
            ok = ok and np.array_equal(outs[kk]["dist"].index_select(0, ix).cpu().numpy().view(np.uint64), dist) and \
                np.array_equal(outs[kk]["eob"].index_select(0, ix).cpu().numpy().view(np.uint16), eob)
        return n_chk, bool(ok)
    for kk in W.LADDER:
        checks["rdo_pixel_luma_%dx%d_K%d" % (kk, kk, k)] = lambda kk=kk: chk_rdo(kk)
    # 5b the same chain on both chroma planes of the 4:2:0 frame: ladder 32/16/8/4, weighted SSE with the luma
    # scale grid at chroma decimation (compute_distortion's chroma leg, src/rdo.rs:305-345)
    cw, ch = fw // 2, fh // 2
    h_chroma = []
    for i in range(4):
        a = O.HostPlane(cw, ch, bd, 44, 44)
        a.data = W.random_plane_array(cw, ch, bd, 30 + i, 44, 44)
        h_chroma.append(a)
    # a reconstruction that resembles its source plane (residuals of a few grey levels)
    for i in (0, 2):
        nz = np.random.default_rng(40 + i).integers(-5 << (bd - 8), (5 << (bd - 8)) + 1, h_chroma[i].data.shape)
        h_chroma[i].data = np.clip(h_chroma[i + 1].data.astype(np.int64) + nz, 0, (1 << bd) - 1).astype(dt)
    chroma = [Plane.from_numpy(a.data, cw, ch, bd, 44, 44) for a in h_chroma]
    ccands = W.speed6_ladder(cw, ch, k, seed=7, mv_range=16, sizes=(32, 16, 8, 4))
    dccands = {kk: torch.from_numpy(v.view(np.uint8).reshape(-1).copy()).cuda() for kk, v in ccands.items()}
    couts = {(p, kk): {} for p in (0, 1) for kk in ccands}

    def rdo_chroma():
        for p in (0, 1):
            for kk in ccands:
                ctx.rdo_pixel_cand_batch(chroma[2 * p + 1], chroma[2 * p], kk, kk, dccands[kk], qindex, 2, scales=scales,
                                         xdec=1, ydec=1, n=len(ccands[kk]), outs=couts[(p, kk)], want_sad=False,
                                         want_satd=False)
    stages.append(("rdo_pixel_candidates_chroma_2planes_K%d" % k, rdo_chroma))

    def chk_rdo_chroma():
        L = O.lib()
        L.r1o_set_threads(os.cpu_count() or 1)
        n_chk, ok = 0, True
        for p in (0, 1):
            pa, pb = h_chroma[2 * p + 1].cstruct(), h_chroma[2 * p].cstruct()
            for kk in ccands:
                idx = sample(len(ccands[kk]), 8)
                sub = np.ascontiguousarray(ccands[kk][idx])
                eob, dist = np.zeros(len(sub), np.uint16), np.zeros(len(sub), np.uint64)
                assert L.r1o_rdo_pixel_cand_batch(C.byref(pa), C.byref(pb), kk, kk, TS[kk], O.ptr(sub), len(sub), qindex, 0, 0,
                                                  0, 2, O.ptr(h_scales), h_scales.shape[1], 1, 1, None, None, O.ptr(eob),
                                                  O.ptr(dist), None, None, None) == 0
                ix = torch.from_numpy(idx.astype(np.int64)).cuda()
                n_chk += len(idx)
                o = couts[(p, kk)]
                ok = ok and np.array_equal(o["dist"].index_select(0, ix).cpu().numpy().view(np.uint64), dist) and \
                    np.array_equal(o["eob"].index_select(0, ix).cpu().numpy().view(np.uint16), eob)
        return n_chk, bool(ok)
    checks["rdo_pixel_candidates_chroma_2planes_K%d" % k] = chk_rdo_chroma
    # 5c transform-type search: ONE prediction per 16x16 and per 8x8 luma block, the 7 RAV1E_TX_TYPES
    tsearch = {}
    for kk in (16, 8):
        c1 = np.ascontiguousarray(cands[kk][::k])
        mask = ctx.tx_type_mask(TS[kk], True)
        nt = bin(mask).count("1")
        tsearch[kk] = dict(c=c1, mask=mask, nt=nt, dev=torch.from_numpy(c1.view(np.uint8).reshape(-1).copy()).cuda(),
                           outs={"eob": torch.empty((len(c1), nt), dtype=torch.int16, device="cuda"),
                                 "dist": torch.empty((len(c1), nt), dtype=torch.int64, device="cuda")})

    def txsearch():
        for kk, t in tsearch.items():
            ctx.rdo_txsearch_batch(org[0], refs[0][0], kk, kk, t["dev"], t["mask"], qindex, 3, scales=scales, n=len(t["c"]),
                                   outs=t["outs"])
    stages.append(("tx_type_search_16x16_8x8_7types", txsearch))

    def chk_txsearch():
        L = O.lib()
        L.r1o_set_threads(os.cpu_count() or 1)
        pa, pb = h_org[0].cstruct(), h_refs[0][0].cstruct()
        n_chk, ok = 0, True
        for kk, t in tsearch.items():
            idx = sample(len(t["c"]))
            sub = np.ascontiguousarray(t["c"][idx])
            eob, dist = np.zeros((len(sub), t["nt"]), np.uint16), np.zeros((len(sub), t["nt"]), np.uint64)
            assert L.r1o_rdo_txsearch_batch(C.byref(pa), C.byref(pb), None, kk, kk, TS[kk], O.ptr(sub), len(sub), t["mask"],
                                            qindex, 0, 0, 0, 3, O.ptr(h_scales), h_scales.shape[1], 0, 0, None, None,
                                            O.ptr(eob), O.ptr(dist), None, None, None) == 0
            ix = torch.from_numpy(idx.astype(np.int64)).cuda()
            n_chk += len(idx) * t["nt"]
            ok = ok and np.array_equal(t["outs"]["dist"].index_select(0, ix).cpu().numpy().view(np.uint64), dist) and \
                np.array_equal(t["outs"]["eob"].index_select(0, ix).cpu().numpy().view(np.uint16), eob)
        return n_chk, bool(ok)
    checks["tx_type_search_16x16_8x8_7types"] = chk_txsearch
    # 6-8 post filters on the 4:2:0 frame
    blocks = D.random_blocks(rng, cols, rows, 1, 1)
    dblocks = torch.from_numpy(blocks.view(np.uint8).reshape(blocks.shape + (8,)).copy()).cuda()
    state = D.make_state([24, 20, 16, 16])
    planes3 = [(refs[0][0], org[0], 0, 0, 0), (chroma[0], chroma[1], 1, 1, 1), (chroma[2], chroma[3], 2, 1, 1)]
    h_rec3, h_src3 = [h_refs[0][0], h_chroma[0], h_chroma[2]], [h_org[0], h_chroma[1], h_chroma[3]]
    tall = torch.zeros((3, 2, 65), dtype=torch.int64, device="cuda")
    rec3, src3 = [a for (a, b, p, xd, yd) in planes3], [b for (a, b, p, xd, yd) in planes3]
    restore.extend((p, p.data.clone()) for p in rec3)
    stages.append(("deblock_level_search_420", lambda: ctx.deblock_sse_frame(rec3, src3, 1, 1, dblocks, fw, fh, tallies=tall)))

    def chk_deblock_search():
        # the level-search tallies of both chroma planes (whole planes) against the oracle; must run before the
        # deblocking stage touches the reconstruction
        L = O.lib()
        L.r1o_set_threads(os.cpu_count() or 1)
        got = ctx.deblock_sse_frame(rec3, src3, 1, 1, dblocks, fw, fh).cpu().numpy()
        ok, n_chk = True, 0
        for pli in (1, 2):
            want = np.zeros((2, 65), np.int64)
            pc, sc = h_rec3[pli].cstruct(), h_src3[pli].cstruct()
            assert L.r1o_deblock_sse_plane(C.byref(pc), C.byref(sc), pli, 1, 1, blocks.ctypes.data, blocks.shape[1],
                                           blocks.shape[1], blocks.shape[0], fw, fh, bd, want[0].ctypes.data,
                                           want[1].ctypes.data) == 0
            ok = ok and np.array_equal(got[pli], want)
            n_chk += want.size
        return n_chk, bool(ok)
    checks["deblock_level_search_420"] = chk_deblock_search
    stages.append(("deblock_filter_420", lambda: ctx.deblock_frame(state, rec3, 1, 1, dblocks, fw, fh)))

    def chk_deblock():
        # one application of the filter on FRESH copies of the three planes (the timed stage filters in place);
        # the whole first chroma plane against the oracle
        L = O.lib()
        L.r1o_set_threads(os.cpu_count() or 1)
        hp = O.HostPlane(cw, ch, bd, 44, 44)
        hp.data = h_chroma[0].data.copy()
        fresh = [Plane.from_numpy(h_refs[0][0].data, fw, fh, bd, 88, 88), Plane.from_numpy(hp.data, cw, ch, bd, 44, 44),
                 Plane.from_numpy(h_chroma[2].data, cw, ch, bd, 44, 44)]
        ctx.deblock_frame(state, fresh, 1, 1, dblocks, fw, fh)
        pc = hp.cstruct()
        assert L.r1o_deblock_plane(state.ctypes.data, C.byref(pc), 1, 1, 1, blocks.ctypes.data, blocks.shape[1],
                                   blocks.shape[1], blocks.shape[0], fw, fh, bd) == 0
        got = fresh[1].data.cpu().numpy().view(dt)
        return int(hp.data.size), bool(np.array_equal(got, hp.data))
    checks["deblock_filter_420"] = chk_deblock
    in_place = {"deblock_filter_420"}     # stages that change their input planes: verify() does not run them
    skip_s = torch.zeros((2 * ((fh + 7) // 8), 2 * ((fw + 7) // 8)), dtype=torch.uint8, device="cuda")
    cdef_search = lambda: ctx.cdef_strength_search(rec3, src3, skip_s, PRESETS, PRESETS, 5, bd, 8, 1, 1, fw, fh, scales=scales)
    stages.append(("cdef_strength_search_8_presets_420", cdef_search))
    dst = Plane(fw, fh, bd)
    skip = torch.zeros((fh // 4, fw // 4), dtype=torch.uint8, device="cuda")
    ci = torch.zeros(((fh + 63) // 64, (fw + 63) // 64), dtype=torch.uint8, device="cuda")
    stages.append(("cdef_luma", lambda: ctx.cdef_filter_frame_plane(refs[0][0], refs[0][0], dst, 0, 0, 0, fw, fh, skip, ci,
                                                                    [36] * 8, [36] * 8, 5, bd)))

    # the restoration units of the frame are laid out by the host driver below as RestorationState::new does for this
    # quantizer (rdo_glue, pinned by lrf_geometry_ref.npz: 64-pixel luma / 32-pixel chroma units at qindex <= 160 on
    # 4:2:0), every set per unit, edge flags from the unit's place in its area

    lrf_res = {}
    # 8b rdo_loop_decision with BOTH filters on (speed 4: cdef and lrf, src/api/config/speedsettings.rs:78-79,168-171):
    # the reference alternates the two legs until no choice changes (src/rdo.rs:2366-2374).  The two stages above are
    # the first pass of each leg.  Here the iteration's device calls as the host driver (rav1e_amd/loop_decision.py)
    # makes them on THIS frame: the CDEF working copy, the restoration leg on it, and the SECOND pass -- every CDEF
    # trial with the unit's restoration choice applied to the trial's output before the error is taken, the working
    # copy again, the restoration leg again for the areas whose cdef_index moved.  The host decisions in between (costs,
    # picks; stated rates as in tests/golden/gen_loop_decision_ref.py) are made once at build time and the calls
    # replayed: what is timed is the device work of passes 1 and 2.
    from rav1e_amd import loop_decision as LD
    work3 = [Plane(fw, fh, bd, 88, 88), Plane(cw, ch, bd, 44, 44), Plane(cw, ch, bd, 44, 44)]
    for wp, rp_ in zip(work3, rec3):
        wp.data.copy_(rp_.data)

    class Recorder(LD.DeviceBackend):
        log = []

        def trial(self, units, sb_sel):
            Recorder.log.append(("trial", [u.copy() for u in units], sb_sel.copy()))
            return super().trial(units, sb_sel)

        def apply(self, index_sb):
            Recorder.log.append(("apply", index_sb.copy()))
            return super().apply(index_sb)

        def lrf_search(self, pli, rows):
            Recorder.log.append(("lrf", pli, rows.copy()))
            return super().lrf_search(pli, rows)
    h_skip_s = np.zeros((2 * ((fh + 7) // 8), 2 * ((fw + 7) // 8)), np.uint8)
    lam2 = 90.0 * (1 << (2 * (bd - 8)))
    rec_be = Recorder(ctx, rec3, work3, src3, skip_s, PRESETS, PRESETS, 5, bd, 8, 1, 1, fw, fh, (1, 1), scales, [1 << 14] * 3)
    ld = LD.LoopDecision(rec_be, fw, fh, 1, 1, qindex, h_skip_s, lam2, lambda pli, f_: 24 if f_ is None else 96 + 8 * f_[0],
                         8, LD.SGR_SETS["Reduced"])
    assert ld.area == (1, 1), "the replay below assumes one superblock per area (qindex <= 160)"
    ld.run(max_passes=2)
    torch.cuda.synchronize()
    calls = Recorder.log
    trials = [c_ for c_ in calls if c_[0] == "trial"]
    apply_at = [i_ for i_, c_ in enumerate(calls) if c_[0] == "apply"]
    applies = [calls[i_] for i_ in apply_at]
    lrfs = [[c_ for c_ in calls[i_:] if c_[0] == "lrf"][:3] for i_ in apply_at]
    loop_info = {"passes_run": ld.passes, "superblocks": int(ld.n_sbx * ld.n_sby),
                 "pass2_trial_units": [int(len(u)) for u in trials[1][1]] if len(trials) > 1 else None,
                 "pass1_choices_sgrproj": int(sum(len(u) for u in trials[1][1])) if len(trials) > 1 else 0,
                 "pass2_restoration_units": [int(len(c_[2]) // 9) for c_ in lrfs[1]] if len(lrfs) > 1 else None}
    dev_rows = lambda r_: torch.from_numpy(np.ascontiguousarray(r_).view(np.uint8).reshape(-1).copy()).cuda()
    idx_dev = [torch.from_numpy(a_[1]).cuda() for a_ in applies]
    ckw = dict(y_strengths=PRESETS, uv_strengths=PRESETS, damping=5, bit_depth=bd, n_idx=8, xdec=1, ydec=1, crop_w=fw, crop_h=fh)
    stages.append(("cdef_apply_area_420", lambda: ctx.cdef_apply_area(rec3, work3, skip_s, idx_dev[0], **ckw)))

    def lrf_replay(k):
        rows_dev = [(c_[1], dev_rows(c_[2]), int(c_[2]["w"].max()), int(c_[2]["h"].max())) for c_ in lrfs[k]]

        def run():
            for (pl, rd, mw, mh) in rows_dev:
                lrf_res[(k, pl)] = ctx.lrf_search_batch(work3[pl], src3[pl], rd, is_chroma=pl > 0, xdec=int(pl > 0), ydec=int(pl > 0),
                                                        scales=scales, max_w=mw, max_h=mh)
        return run
    lrf_pass1 = lrf_replay(0)
    stages.append(("lrf_search_8_sets_420", lrf_pass1))      # the restoration leg, on the CDEF working copy (rdo.rs:2575-2582)

    def host_work():
        hs = []
        for wp in work3:
            a_ = wp.data.cpu().numpy()
            hp = O.HostPlane(wp.width, wp.height, bd, wp.xpad, wp.ypad)
            hp.data = a_ if a_.dtype == np.uint8 else a_.view(np.uint16)
            hs.append(hp)
        return hs

    def chk_lrf_search():
        L = O.lib()
        n_chk, ok = 0, True
        ctx.cdef_apply_area(rec3, work3, skip_s, idx_dev[0], **ckw)
        lrf_pass1()
        torch.cuda.synchronize()
        h_work = host_work()
        for (_, pl, hu) in lrfs[0]:
            xd = int(pl > 0)
            xqd, err = lrf_res[(0, pl)][0].cpu().numpy(), lrf_res[(0, pl)][1].cpu().numpy().view(np.uint64)
            pc, sc = h_work[pl].cstruct(), h_src3[pl].cstruct()
            for i in sample(len(hu), 27):      # strided over (unit, set): every set appears
                wx, we = np.zeros(2, np.int8), np.zeros(1, np.uint64)
                assert L.r1o_lrf_search_unit(C.byref(pc), C.byref(sc), int(hu["x"][i]), int(hu["y"][i]), int(hu["w"][i]),
                                             int(hu["h"][i]), int(hu["set"][i]), int(hu["edges"][i]), int(pl != 0), xd, xd, h_scales.ctypes.data,
                                             h_scales.shape[1], 1 << 14, bd, wx.ctypes.data, we.ctypes.data) == 0
                ok = ok and np.array_equal(xqd[i], wx) and int(err[i]) == int(we[0])
                n_chk += 1
        return n_chk, bool(ok)
    checks["lrf_search_8_sets_420"] = chk_lrf_search
    if len(trials) > 1 and len(applies) > 1:
        t2_units = np.concatenate(trials[1][1])
        t2 = (dev_rows(t2_units), [len(u) for u in trials[1][1]])
        t2_sel = torch.from_numpy(trials[1][2]).cuda()
        t2_scratch = ctx.cdef_lrf_trial_scratch(rec3, skip_s, 8, 1, 1)
        t2_outs = {}
        trial2 = lambda: ctx.cdef_lrf_trial_batch(rec3, work3, src3, skip_s, t2, scales=scales, sb_sel=t2_sel, scratch=t2_scratch,
                                                  outs=t2_outs, **ckw)
        stages.append(("cdef_lrf_trial_pass2_420", trial2))
        stages.append(("cdef_apply_area_pass2_420", lambda: ctx.cdef_apply_area(rec3, work3, skip_s, idx_dev[1], **ckw)))
        if len(lrfs) > 1 and lrfs[1]:
            stages.append(("lrf_search_pass2_420", lrf_replay(1)))

        def chk_trial2():
            # a strided sample of superblocks through oracle/loop_decision.c (r1o_cdef_lrf_trial with sb_sel = the sample):
            # the per-plane errors of all eight indices, restored planes included
            import loop_decision_util as U
            L = U.sigs(O.lib())
            ctx.cdef_apply_area(rec3, work3, skip_s, idx_dev[0], **ckw)
            trial2()
            torch.cuda.synchronize()
            got = t2_outs["err_planes"].cpu().numpy().view(np.uint64)
            n_sby, n_sbx = got.shape[:2]
            sel = np.zeros((n_sby, n_sbx), np.uint8)
            pick = sample(n_sby * n_sbx, 20)
            sel.reshape(-1)[pick] = trials[1][2].reshape(-1)[pick]
            h_work = host_work()
            prm = O.CdefSearchParams()
            prm.y_strengths[:] = PRESETS
            prm.uv_strengths[:] = PRESETS
            prm.damping, prm.bit_depth, prm.n_idx, prm.planes = 5, bd, 8, 3
            prm.xdec, prm.ydec, prm.crop_w, prm.crop_h, prm.area_sb_w, prm.area_sb_h = 1, 1, fw, fh, 1, 1
            prm.dist_scale[:] = [1 << 14] * 3
            p3 = lambda hs: (O.Plane * 3)(*[h_.cstruct() for h_ in hs])
            err = np.zeros((n_sby, n_sbx, 8), np.uint64)
            errp = np.zeros((n_sby, n_sbx, 8, 3), np.uint64)
            best = np.zeros((n_sby, n_sbx), np.int8)
            n_units = (C.c_int32 * 3)(*t2[1])
            hu = np.ascontiguousarray(t2_units)
            assert L.r1o_cdef_lrf_trial(p3(h_rec3), p3(h_work), p3(h_src3), h_skip_s.ctypes.data, h_skip_s.shape[1],
                                        h_skip_s.shape[1], h_skip_s.shape[0], h_scales.ctypes.data, h_scales.shape[1],
                                        C.byref(prm), hu.ctypes.data, n_units, sel.ctypes.data, err.ctypes.data,
                                        errp.ctypes.data, best.ctypes.data) == 0
            m = sel.astype(bool)
            return int(m.sum()) * 24, bool(np.array_equal(got[m], errp[m]) and m.sum() > 0)
        checks["cdef_lrf_trial_pass2_420"] = chk_trial2

    us = 64
    units = np.zeros((max((fh + 32) // us, 1), max((fw + 32) // us, 1), 4), np.uint8)
    units[..., 0] = 3
    units[..., 1] = rng.integers(0, 16, units.shape[:2])
    units[..., 2] = rng.integers(-96, 32, units.shape[:2]).astype(np.int8).view(np.uint8)
    units[..., 3] = rng.integers(-32, 96, units.shape[:2]).astype(np.int8).view(np.uint8)
    dunits = torch.from_numpy(units).cuda()
    lrf_out = Plane(fw, fh, bd)
    stages.append(("lrf_sgrproj_luma", lambda: ctx.lrf_sgrproj_plane(dst, refs[0][0], lrf_out, 0, fw, fh, fh, us, dunits, 64)))

    # The wavefront-bound ME of frame N+1 leaves most CUs idle: run it on a second stream next to the
    # throughput-bound stages of frame N (they touch different buffers).
    s_me, s_rdo = torch.cuda.Stream(), torch.cuda.Stream()
    by_name = dict(stages)
    main_order = [n for n, _ in stages if n not in ("lookahead_intra_costs", "estimate_tile_motion_8tiles_x_3refs",
                                                    "update_block_importances_3refs", "restore_reconstruction_untimed")]

    def overlapped():
        with torch.cuda.stream(s_me):
            tile_me()
            importances()
        with torch.cuda.stream(s_rdo):
            by_name["restore_reconstruction_untimed"]()
            for n in main_order:
                by_name[n]()

    # Four streams: what a frame PIPELINE keeps in flight together -- the motion search and pre-screens of frame n + 1,
    # the candidate chains of frame n (luma; chroma + type search), the post-filter decisions of frame n - 1.  The
    # groups touch disjoint buffers and have no order among themselves inside a pass (a real pipeline orders them
    # ACROSS frames); the latency-bound groups fill the issue slots the VALU-bound chains leave.
    s4 = [torch.cuda.Stream() for _ in range(4)]
    post = [n for n in main_order if n.split("_")[0] in ("deblock", "cdef", "lrf")]
    luma = [n for n in main_order if n.startswith("rdo_pixel_luma")]
    chroma_tx = [n for n in main_order if n.startswith("rdo_pixel_candidates_chroma") or n.startswith("tx_type_search")]
    front = [n for n in main_order if n not in post and n not in luma and n not in chroma_tx]
    plan4 = {"me+importances+subpel+prescreen": front, "pixel chain luma": luma, "pixel chain chroma + type search": chroma_tx,
             "post-filter decisions + filters": post}

    def pipelined4():
        with torch.cuda.stream(s4[0]):
            tile_me()
            importances()
            for n in front:
                by_name[n]()
        with torch.cuda.stream(s4[1]):
            for n in luma:
                by_name[n]()
        with torch.cuda.stream(s4[2]):
            for n in chroma_tx:
                by_name[n]()
        with torch.cuda.stream(s4[3]):
            by_name["restore_reconstruction_untimed"]()
            for n in post:
                by_name[n]()
    px = {"rdo_pixel_candidates_chroma_2planes_K%d" % k: 2 * sum(len(v) * kk * kk for kk, v in ccands.items()),
          "tx_type_search_16x16_8x8_7types": sum(len(t["c"]) * t["nt"] * kk * kk for kk, t in tsearch.items())}
    bpp = 1 if bd == 8 else 2
    fpx, pyr = fw * fh * bpp, fw * fh * bpp * 21 // 16                  # a plane; its ME pyramid (1 + 1/4 + 1/16)
    chain = lambda s_, n_: (bpp * ((s_ + 7) * (s_ + 7) + 2 * s_ * s_) + 18) * n_   # window + source block + descriptor + results
    # ALGORITHMIC bytes per stage (what the stage must read and write once; SURVEY 8(d) per-unit figures where it has them)
    alg = {"lookahead_intra_costs": fpx + 4 * hb * wb,
           "estimate_tile_motion_8tiles_x_3refs": 3 * (2 * pyr + 8 * rows * cols),
           "update_block_importances_3refs": 3 * (2 * fpx + 24 * hb * wb),
           "estimate_motion_subpel_16x16_all_blocks": len(c) * (bpp * (256 + 23 * 23) + 24),
           "intra_prescreen_16x16_13modes": nb * (bpp * (256 + 2 * 65) + 13 * 8),
           "rdo_pixel_candidates_chroma_2planes_K%d" % k: 2 * sum(chain(kk, len(v)) for kk, v in ccands.items()),
           "tx_type_search_16x16_8x8_7types": sum((bpp * ((kk + 7) * (kk + 7) + kk * kk) + 16 + 10 * t["nt"]) * len(t["c"])
                                                  for kk, t in tsearch.items()),
           "deblock_level_search_420": 2 * fpx * 3 // 2, "deblock_filter_420": 2 * fpx * 3 // 2,
           "cdef_strength_search_8_presets_420": 2 * fpx * 3 // 2, "cdef_luma": 2 * fpx,
           "lrf_search_8_sets_420": 2 * fpx * 3 // 2, "lrf_sgrproj_luma": 3 * fpx,
           # the working copy: read rec, write work; the restoration leg: the working copy + the source once;
           # a pass-2 trial: rec + source once, plus -- for the planes under a restoration choice -- every index' trial
           # output written and read back (8 x), which is this implementation's traffic, not compulsory: not counted
           "cdef_apply_area_420": 2 * fpx * 3 // 2, "cdef_apply_area_pass2_420": 2 * fpx * 3 // 2,
           "lrf_search_pass2_420": 2 * fpx * 3 // 2,
           "cdef_lrf_trial_pass2_420": 2 * fpx * 3 // 2}
    for kk in W.LADDER:
        nm = "rdo_pixel_luma_%dx%d_K%d" % (kk, kk, k)
        px[nm] = len(cands[kk]) * kk * kk
        alg[nm] = chain(kk, len(cands[kk]))

    def verify():
        """every stage once, in order, each followed by its parity sample against the CPU oracle (stages that filter
        in place are checked on fresh copies and not run) -> {stage: (n_checked, ok)}"""
        res = {}
        for name, fn in stages:
            if name not in in_place:
                fn()
                torch.cuda.synchronize()
            if name in checks:
                res[name] = checks[name]()
        return res
    return dict(stages=stages, checks=checks, verify=verify, overlapped=overlapped, pipelined4=pipelined4, plan4=plan4,
                candidate_pixels=px, frame=(fw, fh, bd),
                loop_decision=loop_info,
                algorithmic_bytes=alg, luma_launch_n={kk: len(cands[kk]) for kk in W.LADDER},
                working_set_bytes=sum(int(p.data.numel() * p.data.element_size()) for p in [org[0], refs[0][0]] + chroma),
                keep=(org, refs, chroma, stats, outs, couts, tsearch, dblocks, dunits, lrf_out, dst, work3))


def time_stages(stages, reps=20, warm=5, sustain_ms=300.0):
    """The timing scheme of bench.py::config_lines: `warm` untimed passes over all stages, passes for `sustain_ms`
    more (sustained clocks), then `reps` passes with a HIP event pair around every stage on the launch stream
    -> ({stage: mean ms}, wall ms per pass)"""
    import time
    import torch
    for _ in range(warm):
        for _, f in stages:
            f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < sustain_ms:
        for _, f in stages:
            f()
        torch.cuda.synchronize()
    ev = []
    t0 = time.perf_counter()
    for _ in range(reps):
        for tag, f in stages:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            f()
            e1.record()
            ev.append((tag, e0, e1))
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / reps * 1e3
    per = {}
    for tag, a, b in ev:
        per.setdefault(tag, []).append(a.elapsed_time(b))
    return {t: sum(v) / len(v) for t, v in per.items()}, wall
