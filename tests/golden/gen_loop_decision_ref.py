#!/usr/bin/env python3
"""tests/golden/loop_decision_ref.npz: rdo_loop_decision ITSELF (src/rdo.rs:2104-2763), transpiled by
tools/rustlite and executed whole -- the loop that gen_cdef_search_ref.py and gen_lrf_search_ref.py state by hand
around the executed callees: which superblocks / restoration units an area has, the skip tests, the scratch copies,
the no-filter option, the visible size of a unit, the order solve -> filter -> error per parameter set, "first
smallest cost wins", the final filter pass, and the iteration between the two legs.

What is executed (the reference's own text):
  rdo_loop_decision and everything it calls in rdo.rs / cdef.rs / lrf.rs / dist.rs / activity.rs;
  RestorationState::new (lrf.rs:1321-1480) for the unit geometry of each frame;
  Tile / TileMut (tiling/tile.rs), TileRestorationState* (tiling/tile_restoration_state.rs), TileBlocks*
  (tiling/tile_blocks.rs), FrameBlocks (context/block_unit.rs): read from the tree, their macro_rules! expanded
  textually (refmacro.expand) and the raw-pointer row access of the tiled views rewritten to indices
  (refmacro.unpointer: same elements);
  AsTile for Frame (frame/mod.rs).
What is NOT the reference's text:
  v_frame's Frame (`struct Frame { planes }`), Plane, PlaneRegion (tools/rustlite/runtime.py stand-ins, as in every
  generator); TileStateMut / FrameInvariants / ContextWriter are plain objects carrying the fields the function reads;
  Area::to_rect hands back the stand-in Rect; ptr::null / ptr::null_mut;
  cw.fc.count_lrf_switchable (the entropy coder's adaptive CDFs) is replaced by a STATED rate:
      rate(None) = RATE_NONE, rate(Sgrproj { set, .. }) = RATE_SGR + RATE_PER_SET * set      (1/8 bit units)
  so that the cost comparison and the final choice are exercised too.
Calls are recorded by wrapping the transpiled rdo_loop_plane_error / sgrproj_solve / cdef_filter_superblock /
compute_rd_cost: arguments and results as the function itself produced them.

Cases, in the formats the existing tests read:
  ldl<k>  restoration only   -> <c>_meta, _in{0,1,2}, _src{0,1,2}, _scales, _dscale, _rows, _err   (= lrf_search_ref.npz)
                                + _edges (per row: 1 = the unit's slice does not start in column 0 of the area's scratch
                                copy, 2 = not in row 0 -- the pixels left of / above it exist for setup_integral_image), _cost (f64 per row), _choice = [pli, x, y, set (255 none), xqd0, xqd1] per unit,
                                _rate = [RATE_NONE, RATE_SGR, RATE_PER_SET], _lambda, _geo = RestorationPlaneConfig rows
  ldc<k>  CDEF only          -> <c>_meta, _rec{0,1,2}, _src{0,1,2}, _skip, _ystr, _uvstr, _scales, _dscale, _err, _best
                                (= cdef_search_ref.npz) + _areas = [sbx0, sby0, sb_w, sb_h] per call
  every case: <c>_q = [base_q_idx, 1 = all sixteen parameter sets / 0 = the reduced eight]
  ldb<k>  both filters on    -> the CDEF leg's first pass in the ldc format (_err, _best = the pick of that pass), the
                                whole event trace (<c>_trace, <c>_trace_err) and the final choices (_best_final,
                                _choice): the interleaving of the two legs, recorded for the host-side integration
  trace rows: [kind, a, b, c, d, e, f]  kind 0 = plane error (pli, loop_sbx, loop_sby, sb_w, sb_h, frame: 0 = the first
  frame the calls of an area are made on -- the CDEF working copy, which is also the restoration input -- 1 = the
  restoration working copy; value in _trace_err), 1 = solve (set, px, py, vis_w, vis_h, xqd0 * 256 + (xqd1 & 255)),
  2 = cdef_filter_superblock (loop_sbx, loop_sby, index), 3 = area start (sbx0, sby0)

Run in the build container:  python tests/golden/gen_loop_decision_ref.py      (about 3 minutes)
"""
import os
import time

import numpy as np

import reflib as L
import refmacro
from reflib import R
from gen_lrf_ref import Obj, PixelVec

RATE_NONE, RATE_SGR, RATE_PER_SET = 24, 96, 8

# kind, W, H, xdec, ydec, bd, base_q_idx, sgr complexity, cdef_bits, p_skip, lambda
CASES = [
    ("ldl0", 136, 72, 1, 1, 8, 100, "Full", 0, 0.0, 90.0),
    ("ldl1", 192, 128, 1, 1, 10, 180, "Reduced", 0, 0.0, 400.0),
    ("ldl2", 104, 64, 0, 0, 8, 100, "Reduced", 0, 0.0, 60.0),
    ("ldl3", 96, 80, 1, 0, 12, 100, "Reduced", 0, 0.0, 2000.0),
    ("ldc0", 136, 72, 1, 1, 8, 100, "Full", 3, 0.35, 90.0),
    ("ldc1", 200, 136, 1, 1, 10, 180, "Full", 3, 0.3, 400.0),
    ("ldc2", 96, 80, 1, 0, 12, 100, "Full", 1, 0.2, 2000.0),
    ("ldb0", 136, 72, 1, 1, 8, 100, "Reduced", 2, 0.25, 90.0),
    # round 6: the later passes on the device (r1_cdef_lrf_trial_batch) -- 10-bit 4:2:0 (BASELINE configs[3]'s format),
    # and an area of SEVERAL superblocks (128-pixel luma units under qindex 180: a trial reads its neighbours' current
    # CDEF output left of / above itself)
    ("ldb1", 136, 72, 1, 1, 10, 100, "Reduced", 2, 0.25, 400.0),
    ("ldb2", 192, 128, 1, 1, 8, 180, "Reduced", 1, 0.2, 90.0),
]


def crate():
    src = L.REF_SRC + "/"
    c = L.crate("lrf.rs", "rdo.rs", "dist.rs", "activity.rs", "cdef.rs", "deblock.rs", "partition.rs", "predict.rs",
                "transform/mod.rs", "context/superblock_unit.rs", "context/block_unit.rs", "tiling/plane_region.rs",
                "util/mod.rs")
    c.define_enum("ChromaSampling", ["Cs420", "Cs422", "Cs444", "Cs400"])
    c.load_text("<v_frame 0.3.9: Frame>", "pub struct Frame<T: Pixel> { pub planes: [Plane<T>; 3] }")
    c.load_text("tiling/tile.rs", refmacro.expand(open(src + "tiling/tile.rs").read()))
    for f in ("tiling/tile_restoration_state.rs", "tiling/tile_blocks.rs"):
        c.load_text(f, refmacro.unpointer(refmacro.expand(open(src + f).read())))
    c.load("frame/mod.rs")
    for n in ("null", "null_mut"):
        c.define_py(n, lambda _g: R.RPtr(None, 0))
        c.fns[n][0].modpath = ("ptr",)
    return c


def override(c, owner, name, fn):
    info = c.methods[owner][name][0]
    c.pyfn(info)
    c.G[info.pyname] = fn


def index_by_offset_too(c, owner, name):
    """TileBlocks* implement Index twice -- by row (usize) and by TileBlockOffset (tile_blocks.rs: `&self[bo.0.y][bo.0.x]`);
    the transpiler's method table keeps one `index` per type, so the row form also takes the offset form's argument"""
    info = c.methods[owner][name][0]
    row = c.pyfn(info)

    def index(_g, this, i):
        if isinstance(i, int):
            return row(_g, this, i)
        return row(_g, this, i._0.y)[i._0.x]
    c.G[info.pyname] = index


def wrap(c, name, rec):
    f = c.get(name)

    def w(*a, **k):
        r = f(*a, **k)
        rec(a, r)
        return r
    c.G[f.__name__] = w


def images(rng, ci, W, H, xdec, ydec, bd):
    yy, xx = np.mgrid[0:H, 0:W]
    base = ((np.sin(xx / 6.0 + ci) + np.cos((yy + xx * (ci % 3 + 1)) / 9.0)) * 45 + 128)
    Y = np.clip(base + rng.integers(-3, 4, (H, W)), 0, 255).astype(np.int64) << (bd - 8)
    cw, ch = W >> xdec, H >> ydec
    U = (np.clip(128 + 50 * np.sin(xx[:ch, :cw] / 4.0) + rng.integers(-3, 4, (ch, cw)), 0, 255)).astype(np.int64) << (bd - 8)
    V = np.clip(Y[::1 << ydec, ::1 << xdec][:ch, :cw] // 2 + (40 << (bd - 8)), 0, (1 << bd) - 1)
    src = [Y, U, V]
    # the reconstruction: the source + coding noise, some of it ringing (what the two filters are there to remove)
    rec = []
    for s in src:
        r = s + rng.integers(-5 << (bd - 8), (5 << (bd - 8)) + 1, s.shape) * (rng.random(s.shape) < 0.7)
        # the right half: blurred instead of noisy, so that the solved weights are not the same everywhere
        h = s.shape[1] // 2
        b = (s + np.roll(s, 1, axis=1) + np.roll(s, -1, axis=1) + np.roll(s, 1, axis=0)) // 4
        r[:, h:] = b[:, h:] + rng.integers(-1 << (bd - 8), (1 << (bd - 8)) + 1, (s.shape[0], s.shape[1] - h))
        rec.append(np.clip(r, 0, (1 << bd) - 1))
    return src, rec


def main():
    c = crate()
    override(c, "Area", "to_rect", lambda _g, a, xdec, ydec, pw, ph: R.Rect(*R.area_to_rect(a, xdec, ydec, pw, ph)))
    for owner, name in (("TileBlocks", "index"), ("TileBlocksMut", "index"), ("TileBlocksMut", "index_mut")):
        index_by_offset_too(c, owner, name)
    loop_decision = c.get("rdo_loop_decision")
    Frame, TileRect = c.G["S_Frame"], L.struct(c, "TileRect")

    def frame_clone(fr):
        # v_frame's Frame derives Clone: new planes, new pixel storage (the runtime's struct copy is the Copy-type kind
        # and would leave the clone's planes pointing at the original's pixels: the CDEF working copy IS such a clone)
        return Frame(planes=R.RSlice([R.Plane(R.PlaneData(list(p.data)), p.cfg) for p in fr.planes.tolist()]))
    Frame.clone = frame_clone
    TileMut_new, Tile_new = c.get("new", owner="TileMut"), c.get("new", owner="Tile")
    RS_new = c.get("new", owner="RestorationState")
    TRSM_new = c.get("new", owner="TileRestorationStateMut")
    FB_new = c.get("new", owner="FrameBlocks")
    as_tbm = c.get("as_tile_blocks_mut", owner="FrameBlocks")
    IIB = c.get("zeroed", owner="IntegralImageBuffer")
    PSBO, SBO, TSBO = L.struct(c, "PlaneSuperBlockOffset"), L.struct(c, "SuperBlockOffset"), c.G["S_TileSuperBlockOffset"]
    PBO, BO = L.struct(c, "PlaneBlockOffset"), L.struct(c, "BlockOffset")
    DS = c.G["S_DistortionScale"]
    trace, trace_err, costs = [], [], []
    state = {}

    def rec_err(a, r):
        # rdo_loop_plane_error(g, base_sbo, offset_sbo, sb_w, sb_h, fi, ts, blocks, test, src, pli)
        trace.append((0, a[10], a[2]._0.x, a[2]._0.y, a[3], a[4], state["classify"](a[8])))
        trace_err.append(int(r._0))

    def rec_solve(a, r):
        # sgrproj_solve(g, set, fi, integral, input (src region), cdeffed (PlaneSlice), cdef_w, cdef_h)
        sl = a[5]
        trace.append((1, a[1], sl.x, sl.y, a[6], a[7], int(r[0]) * 256 + (int(r[1]) & 255)))
        trace_err.append(0)

    def rec_cdef(a, r):
        # cdef_filter_superblock(g, fi, input, output, blocks, tile_sbo, cdef_index, cdef_dirs)
        trace.append((2, a[5]._0.x, a[5]._0.y, a[6], 0, 0, 0))
        trace_err.append(0)

    wrap(c, "rdo_loop_plane_error", rec_err)
    wrap(c, "sgrproj_solve", rec_solve)
    wrap(c, "cdef_filter_superblock", rec_cdef)
    wrap(c, "compute_rd_cost", lambda a, r: costs.append((len(trace), float(r))))
    out = {}
    only = os.environ.get("R1_LOOP_DECISION_CASES")      # "ldl0,ldc2": a subset (the mutation check of the tests)
    for ci, (name, W, H, xdec, ydec, bd, q, sgr, cdef_bits, p_skip, lam) in enumerate(CASES):
        if only and name not in only.split(","):
            continue
        t0 = time.time()
        kind = name[2]
        rng = np.random.default_rng([20261004, ci])
        g = dict(L.pixel_type(bd))
        g["U"] = g["T"]
        dt = L.np_dtype(bd)
        cs = L.enum(c, "ChromaSampling", {(1, 1): "Cs420", (1, 0): "Cs422", (0, 0): "Cs444"}[(xdec, ydec)])
        src, rec = images(rng, ci, W, H, xdec, ydec, bd)

        def mk(a, pl):
            xd, yd = (0, 0) if pl == 0 else (xdec, ydec)
            p = L.plane_from_padded(np.pad(a, 16, mode="edge").astype(dt), bd, 16, 16, xdec=xd, ydec=yd)
            p.data = PixelVec(p.data)
            return p
        rec_frame = Frame(planes=R.RSlice([mk(rec[p], p) for p in range(3)]))
        in_frame = Frame(planes=R.RSlice([mk(src[p], p) for p in range(3)]))
        rect = TileRect(x=0, y=0, width=W, height=H)
        sbw, sbh = (W + 63) // 64, (H + 63) // 64
        gw, gh = (W + 7) // 8, (H + 7) // 8
        grid = rng.integers(1 << 12, 1 << 16, (gh, gw)).astype(np.uint32)
        dscale = rng.integers(1 << 13, 1 << 15, 3).astype(np.uint32)
        ystr = rng.integers(0, 64, 8).astype(np.uint8)
        uvstr = rng.integers(0, 64, 8).astype(np.uint8)
        ystr[0], uvstr[0] = 0, 0
        ystr[1:4] = [2 * 4 + 1, 5 * 4 + 2, 13 * 4 + 3]
        uvstr[1:4] = [1 * 4 + 0, 3 * 4 + 1, 7 * 4 + 3]
        damping = int(rng.integers(3, 7))
        fi = Obj(sequence=Obj(bit_depth=bd, chroma_sampling=cs, use_128x128_superblock=False, enable_cdef=kind in "cb",
                              enable_restoration=kind in "lb", enable_large_lru=True,
                              tiling=Obj(cols=1, rows=1, tile_width_sb=sbw, tile_height_sb=sbh)),
                 cpu_feature_level=None, width=W, height=H, base_q_idx=q, sb_width=sbw, sb_height=sbh,
                 coded_frame_data=R.Some(Obj(distortion_scales=R.RSlice([DS(int(v)) for v in grid.ravel()]), w_in_imp_b=gw)),
                 dist_scale=R.RSlice([DS(int(v)) for v in dscale]),
                 config=Obj(temporal_rdo=lambda: True,
                            speed_settings=Obj(lru_on_skip=True, sgr_complexity=L.enum(c, "SGRComplexityLevel", sgr))),
                 cdef_bits=cdef_bits, cdef_damping=damping, cdef_y_strengths=R.RSlice([int(v) for v in ystr]),
                 cdef_uv_strengths=R.RSlice([int(v) for v in uvstr]), lambda_=lam)
        rs = RS_new(g, fi, in_frame)
        geo = [(int(p.cfg.unit_size), int(p.cfg.sb_h_shift), int(p.cfg.sb_v_shift), int(p.cfg.stripe_height), int(p.cfg.cols),
                int(p.cfg.rows)) for p in rs.planes]
        mi_cols, mi_rows = 2 * gw, 2 * gh
        fb = FB_new({}, mi_cols, mi_rows)
        skip = (rng.random((mi_rows, mi_cols)) < p_skip).astype(np.uint8)
        if p_skip > 0:
            skip[:16, :16] = 1                                   # a completely skipped superblock
        for y in range(mi_rows):
            for x in range(mi_cols):
                fb.blocks[y * mi_cols + x].skip = bool(skip[y, x])
        ts = Obj(sbo=PSBO(SBO(x=0, y=0)), sb_size_log2=6, sb_width=sbw, sb_height=sbh, width=W, height=H,
                 rec=TileMut_new(g, rec_frame, rect), input_tile=Tile_new(g, in_frame, rect),
                 restoration=TRSM_new({}, rs, PSBO(SBO(x=0, y=0)), sbw, sbh),
                 integral_buffer=IIB({}, c.const_value("SOLVE_IMAGE_SIZE")), deblock=None)
        ts.to_frame_block_offset = lambda tbo: PBO(BO(x=tbo._0.x, y=tbo._0.y))      # the tile at the frame origin

        def rate(w, rs_, filt, pli):
            return RATE_NONE if filt.var == "None" else RATE_SGR + RATE_PER_SET * int(filt.p[0])
        cw = Obj(bc=Obj(blocks=as_tbm({}, fb)), fc=Obj(count_lrf_switchable=rate))
        del trace[:], trace_err[:], costs[:]
        # the area of one call: the largest restoration unit of the three planes, in superblocks (rdo.rs:2119-2141)
        asw = max(1 << p[1] for p in geo)
        ash = max(1 << p[2] for p in geo)
        areas = []
        for ay in range(0, sbh, ash):
            for ax in range(0, sbw, asw):
                trace.append((3, ax, ay, 0, 0, 0, 0))
                trace_err.append(0)
                # which frame a plane error is taken on: the first frame seen in the restoration leg's no-filter
                # option is its input, cdef_filter_superblock's output is the CDEF working copy, the rest is the
                # restoration working copy -- told apart by the order rdo_loop_decision makes the calls in
                seen = {}

                def classify(fr, seen=seen):
                    k = id(fr)
                    if k not in seen:
                        seen[k] = len(seen)
                    return seen[k]
                state["classify"] = classify
                loop_decision(g, TSBO(SBO(x=ax, y=ay)), fi, ts, cw, None, False)
                areas.append((ax, ay, min(asw, sbw - ax), min(ash, sbh - ay)))
        tr = np.array(trace, np.int64)
        te = np.array(trace_err, np.uint64)
        cost_at = dict(costs)                       # trace length when compute_rd_cost returned -> the cost
        choice = []
        for pli, p in enumerate(rs.planes):
            xd, yd = (0, 0) if pli == 0 else (xdec, ydec)
            for uy in range(int(p.cfg.rows)):
                for ux in range(int(p.cfg.cols)):
                    f = p.units.units[uy * int(p.units.cols) + ux].filter
                    if f.var == "None":
                        choice.append((pli, ux * int(p.cfg.unit_size), uy * int(p.cfg.unit_size), 255, 0, 0))
                    else:
                        choice.append((pli, ux * int(p.cfg.unit_size), uy * int(p.cfg.unit_size), int(f.p[0]), int(f.p[1][0]),
                                       int(f.p[1][1])))
        best = np.full((sbh, sbw), -1, np.int8)
        for sy in range(sbh):
            for sx in range(sbw):
                best[sy, sx] = int(fb.blocks[(sy * 16) * mi_cols + sx * 16].cdef_index) if kind in "cb" else -1
        out[name + "_geo"] = np.array(geo, np.int32)
        out[name + "_q"] = np.array([q, 1 if sgr == "Full" else 0], np.int32)      # base_q_idx, all 16 sets / the reduced 8
        out[name + "_areas"] = np.array(areas, np.int32)
        out[name + "_rate"] = np.array([RATE_NONE, RATE_SGR, RATE_PER_SET], np.int32)
        out[name + "_lambda"] = np.array([lam], np.float64)
        out[name + "_scales"], out[name + "_dscale"] = grid, dscale
        for pl in range(3):
            out[name + "_src%d" % pl] = src[pl].astype(np.uint16)
        if kind == "l":
            # one pass of the restoration leg per area: rows in the order the function made the calls
            rows, errs, rcost, edges = [], [], [], []
            ax = ay = 0
            pending = None
            for i, (k, a, b, c_, d, e, f) in enumerate(tr.tolist()):
                if k == 3:
                    ax, ay = a, b
                elif k == 0:
                    xd, yd = (0, 0) if a == 0 else (xdec, ydec)
                    px, py = ((ax + b) * 64) >> xd, ((ay + c_) * 64) >> yd
                    if f == 0:                                           # the no-filter option: size known at the solve
                        pending = (len(rows), a)
                        rows.append([a, px, py, 0, 0, 255, 0, 0])
                        edges.append(0)
                    else:
                        assert rows[-1][5] != 255 and rows[-1][0] == a and (rows[-1][1], rows[-1][2]) == (px, py)
                    errs.append(int(te[i]))
                    rcost.append(cost_at[i + 1])         # compute_rd_cost of this option: the next call made
                elif k == 1:
                    pli = pending[1]
                    xd, yd = (0, 0) if pli == 0 else (xdec, ydec)
                    px, py = ((ax * 64) >> xd) + b, ((ay * 64) >> yd) + c_
                    if rows[pending[0]][3] == 0:
                        rows[pending[0]][3:5] = [d, e]
                        assert (rows[pending[0]][1], rows[pending[0]][2]) == (px, py)
                    x0, x1 = f >> 8, ((f & 255) ^ 128) - 128
                    rows.append([pli, px, py, d, e, a, x0, x1])
                    # b, c_ = where the unit's slice starts IN THE AREA'S scratch copy: what setup_integral_image looks at
                    # (lrf.rs: `cdeffed.x == 0`, `clamp(y, 0, ..)`) to decide whether pixels left of / above the unit exist
                    edges.append((1 if b > 0 else 0) | (2 if c_ > 0 else 0))
                    edges[pending[0]] = edges[-1]
            assert len(rows) == len(errs)
            out[name + "_meta"] = np.array([W, H, xdec, ydec, bd, asw], np.int32)
            for pl in range(3):
                out[name + "_in%d" % pl] = rec[pl].astype(np.uint16)
            out[name + "_rows"] = np.array(rows, np.int32)
            out[name + "_edges"] = np.array(edges, np.uint8)
            out[name + "_err"] = np.array(errs, np.uint64)
            out[name + "_cost"] = np.array(rcost, np.float64)
            out[name + "_choice"] = np.array(choice, np.int32)
        else:
            # the CDEF leg's FIRST pass over an area (every restoration choice still None): per (superblock, index) the sum
            # of the plane errors that follow the trial's cdef_filter_superblock.  With both filters on, the pass ends at
            # the area's first sgrproj_solve; later passes (restoration applied inside the trials) stay in the trace only
            n_idx = 1 << cdef_bits
            err = np.zeros((sbh, sbw, 8), np.uint64)
            ax = ay = 0
            cur = None
            first_pass = True
            trials, done = {}, set()              # per superblock: trial indices seen so far / its final pass seen
            for i, (k, a, b, c_, d, e, f) in enumerate(tr.tolist()):
                if k == 3:
                    ax, ay = a, b
                    cur, first_pass = None, True
                elif k == 1:
                    first_pass = False
                elif k == 2 and first_pass:
                    sb = (ay + b, ax + a)
                    if sb in done:
                        cur = None                # a later pass over the area
                    elif c_ < trials.get(sb, 0):  # an index again: the final pass with the chosen index, not a trial
                        done.add(sb)
                        cur = None
                    else:
                        trials[sb] = trials.get(sb, 0) + 1
                        cur = sb + (c_,)
                elif k == 0 and cur is not None and first_pass:
                    err[cur[0], cur[1], cur[2]] += te[i]
            tried = set(trials)
            best1 = np.full((sbh, sbw), -1, np.int8)
            for (sy, sx) in tried:
                # "first smallest cost wins" over the indices the loop visited; the rate is the same for every index
                best1[sy, sx] = int(np.argmin(err[sy, sx, :trials[(sy, sx)]]))
            if kind == "c":
                for sy in range(sbh):
                    for sx in range(sbw):
                        if (sy, sx) not in tried:
                            best[sy, sx] = -1                    # a skipped superblock: never searched (rdo.rs:2196-2211)
                assert np.array_equal(best, best1), (best, best1)    # what set_cdef left in the blocks = the argmin
            out[name + "_meta"] = np.array([W, H, xdec, ydec, bd, damping, n_idx, asw, ash, 3], np.int32)
            for pl in range(3):
                out[name + "_rec%d" % pl] = rec[pl].astype(np.uint16)
            out[name + "_skip"], out[name + "_ystr"], out[name + "_uvstr"] = skip, ystr, uvstr
            out[name + "_err"], out[name + "_best"] = err, best1
            if kind == "b":
                out[name + "_trace"], out[name + "_trace_err"] = tr.astype(np.int32), te
                out[name + "_best_final"], out[name + "_choice"] = best, np.array(choice, np.int32)
        print(name, W, H, bd, "geo", geo, "areas", len(areas), "events", len(tr), "best", best.ravel().tolist(),
              "choices", [tuple(r[3:]) for r in choice][:6], "%.0f s" % (time.time() - t0), flush=True)
    L.save("loop_decision_ref.npz", out)


if __name__ == "__main__":
    main()
