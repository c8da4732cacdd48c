#!/usr/bin/env python3
"""tests/golden/me_ref.npz: hierarchical motion estimation computed by the REFERENCE'S
OWN SOURCE TEXT (src/me.rs:153-1523: estimate_tile_motion, estimate_sb_motion,
refine_subsampled_sb_motion, save_me_stats, get_mv_range, get_subset_predictors,
estimate_motion, refine_subsampled_motion_estimate, full_pixel_me, get_best_predictor,
fullpel_diamond_search, hexagon_search, uneven_multi_hex_search, subpel_diamond_search,
get_fullpel_mv_rd, get_subpel_mv_rd, compute_mv_rd, full_search, get_mv_rate; with them
get_sad / get_satd of src/dist.rs, predict_inter_single / get_mv_params of src/predict.rs
and put_8tap of src/mc.rs), transpiled by tools/rustlite and executed here.

Hand-stated (encoder state, plain data here): the fields of FrameInvariants / TileStateMut /
ReferenceFrame / InterConfig the search reads, the per-tile window over FrameMEStats
(TileMEStats, src/tiling/tile_motion_stats.rs), and the half / quarter resolution planes
(Plane::downsampled of the absent v_frame crate: 2x2 box filter, stored in the file as inputs).

Keys per tile case <c>: <c>_meta = [w, h, bd, tile x, y, w, h, allow_hp, allow_full_search,
me_range_scale, n_refs, has_prev], <c>_lambda (by ssdec, the values me.rs:175-177 computes
from me_lambda), <c>_org{0,1,2} / <c>_ref<r>_{0,1,2} (visible areas: full, half, quarter),
<c>_prev<r> / <c>_stats<r> (rows x cols x [row, col, normalized_sad]) and, for the block
cases, <c>_blk (bx, by, w, h, corner, pmv) / <c>_blkout (row, col, sad, cost) /
<c>_blkcfg = [use_satd, filter mode].

Run in the build container:  python tests/golden/gen_me_ref.py
"""
import sys
import time

import numpy as np

import reflib as L
from reflib import R


class Obj:
    def __init__(self, **kw):
        self.__dict__.update(kw)


def box_down2(img):
    h, w = img.shape
    e = np.pad(img.astype(np.int64), ((0, h & 1), (0, w & 1)), mode="edge")
    return (e[0::2, 0::2] + e[0::2, 1::2] + e[1::2, 0::2] + e[1::2, 1::2] + 2) >> 2


def pyramid(img, bd):
    """[full, half, quarter] planes with replicated padding 88 / 44 / 22 (Plane::pad)"""
    h1 = box_down2(img)
    h2 = box_down2(h1)
    return [img, h1, h2], [L.plane_from_padded(np.pad(a, p, mode="edge"), bd, p, p)
                           for a, p in ((img, 88), (h1, 44), (h2, 22))]


def smooth(rng, w, h, bd):
    f = rng.standard_normal((h + 64, w + 64))
    for _ in range(3):
        f = (np.roll(f, 1, 0) + 2 * f + np.roll(f, -1, 0)) / 4
        f = (np.roll(f, 1, 1) + 2 * f + np.roll(f, -1, 1)) / 4
    return ((f - f.min()) / (f.max() - f.min()) * ((1 << bd) - 1)).astype(np.int64)


class Lock:
    """RwLock<[FrameMEStats; REF_FRAMES]>::read().expect(..)"""
    def __init__(self, v):
        self.v = v

    def read(self):
        return self

    def expect(self, _msg):
        return self.v


class TileMEStats:
    """src/tiling/tile_motion_stats.rs: a window of `cols` x `rows` 4x4 units at (x, y)
    of a frame's MEStats rows"""
    def __init__(self, rows_, x, y, cols, rows):
        self.data, self.x_, self.y_, self.cols_, self.rows_ = rows_, x, y, cols, rows

    def x(self):
        return self.x_

    def y(self):
        return self.y_

    def cols(self):
        return self.cols_

    def rows(self):
        return self.rows_

    def as_const(self):
        return self

    def __getitem__(self, r):
        if not 0 <= r < self.rows_:
            raise R.Panic("TileMEStats row %d of %d" % (r, self.rows_))
        return R.RSlice(self.data[self.y_ + r], self.x_, self.cols_)


class World:
    """one frame + its references as the reference's search sees them"""

    def __init__(self, c, bd, org, refs, prevs, tile, allow_hp, full, scale, me_lambda, filt="REGULAR",
                 use_satd=True):
        self.c, self.bd = c, bd
        h, w = org.shape
        G = c.G
        self.MEStats, self.MV = L.struct(c, "MEStats"), L.struct(c, "MotionVector")
        FrameME = L.struct(c, "FrameMEStats")
        cols, rows = (w + 3) // 4, (h + 3) // 4
        self.cols, self.rows = cols, rows
        self.org_imgs, org_planes = pyramid(org, bd)
        self.ref_imgs, frames = [], []
        REFS = ("LAST_FRAME", "LAST2_FRAME", "LAST3_FRAME", "GOLDEN_FRAME", "BWDREF_FRAME", "ALTREF2_FRAME",
                "ALTREF_FRAME")
        self.ref_types = [L.enum(c, "RefType", n) for n in REFS[:len(refs)]]

        def frame_stats(a):
            flat = [self.MEStats(mv=self.MV(row=int(a[y, x, 0]), col=int(a[y, x, 1])),
                                 normalized_sad=int(a[y, x, 2])) for y in range(rows) for x in range(cols)]
            return FrameME(stats=R.RSlice(flat), cols=cols, rows=rows)
        zero = np.zeros((rows, cols, 3), np.int64)
        # reference slot k holds reference k; slot 0 (LAST_FRAME's) carries the previous frame's
        # FrameMEStats of every reference (me.rs:706-708; all-zero MEStats when that frame was
        # not motion-searched -- the reference always has the array)
        prev_arr = R.RSlice([frame_stats(prevs[k] if k < len(prevs) else zero) for k in range(8)])
        for k, ref in enumerate(refs):
            imgs, planes = pyramid(ref, bd)
            self.ref_imgs.append(imgs)
            frames.append(R.Some(Obj(frame=Obj(planes=R.RSlice([planes[0]])), input_hres=planes[1],
                                     input_qres=planes[2], frame_me_stats=Lock(prev_arr))))
        while len(frames) < 8:
            frames.append(R.NONE)
        ref_frames = list(range(len(refs))) + [0] * (7 - len(refs))
        motion = Obj(use_satd_subpel=use_satd, me_allow_full_search=bool(full))
        self.fi = Obj(rec_buffer=Obj(frames=R.RSlice(frames)), ref_frames=R.RSlice(ref_frames), w_in_b=cols,
                      h_in_b=rows, me_lambda=float(me_lambda), sequence=Obj(bit_depth=bd), cpu_feature_level=None,
                      config=Obj(speed_settings=Obj(motion=motion)), me_range_scale=scale,
                      allow_high_precision_mv=bool(allow_hp), default_filter=L.enum(c, "FilterMode", filt))
        tx, ty, tw, th = tile
        self.tile = tile
        # the statistics this frame's search writes: one list of MEStats rows per reference
        self.stat_rows = [[[self.MEStats(mv=self.MV(row=0, col=0), normalized_sad=0) for _ in range(cols)]
                           for _ in range(rows)] for _ in refs]
        PBO, BO = L.struct(c, "PlaneBlockOffset"), L.struct(c, "BlockOffset")
        mi_w, mi_h = min((tw + 3) // 4, cols - tx // 4), min((th + 3) // 4, rows - ty // 4)
        me_stats = [TileMEStats(sr, tx // 4, ty // 4, mi_w, mi_h) for sr in self.stat_rows]
        while len(me_stats) < 7:
            me_stats.append(me_stats[-1])
        ts = Obj(sb_width=(tw + 63) // 64, sb_height=(th + 63) // 64, width=tw, height=th, mi_width=mi_w,
                 mi_height=mi_h, me_stats=R.RSlice(me_stats),
                 input_tile=Obj(planes=R.RSlice([org_planes[0]._region(tx, ty, tw, th)])),
                 input_hres=org_planes[1], input_qres=org_planes[2])
        ts.to_frame_block_offset = lambda tbo: PBO(BO(x=tx // 4 + tbo._0.x, y=ty // 4 + tbo._0.y))
        self.ts = ts
        refs_list = R.RSlice(list(self.ref_types))
        self.inter_cfg = Obj(allowed_ref_frames=lambda: refs_list)

    def stats_array(self, k):
        a = np.zeros((self.rows, self.cols, 3), np.int64)
        for y in range(self.rows):
            for x in range(self.cols):
                s = self.stat_rows[k][y][x]
                a[y, x] = (s.mv.row, s.mv.col, s.normalized_sad)
        return a


def lambdas(me_lambda):
    # me.rs:175-177 (the generator states the same expression for the file's input record)
    return [int(me_lambda * 256.0 / (1 << (2 * s)) * (0.5 if s == 0 else 0.125)) for s in range(3)]


TILE_CASES = [
    # name, w, h, bd, kind, tile, n_refs, prev, allow_hp, full search, range scale, me_lambda
    ("t0", 128, 64, 8, "smooth", None, 1, False, 1, 0, 1, 0.31),
    ("t1", 136, 72, 8, "noise", None, 1, True, 1, 1, 1, 0.31),       # cropped superblocks, full-search stage
    ("t2", 136, 104, 10, "smooth", None, 2, True, 0, 0, 1, 1.24),    # 10-bit, two references
    ("t3", 192, 128, 8, "flat", (64, 0, 128, 128), 1, True, 1, 0, 1, 0.31),   # tile inside the frame, ties
    ("t4", 128, 128, 8, "shift", (0, 64, 128, 64), 1, True, 1, 0, 2, 0.08),   # true motion, lower tile
    ("t5", 256, 192, 8, "shift", None, 3, True, 1, 1, 1, 0.5),      # three references, 4 x 3 superblocks
    # delta-shaped SAD landscapes: noise (odd cases: texture + noise), the reference an exact copy
    # displaced by (dy, dx) quarter-resolution pixels, the first superblock column undisplaced (its
    # SAD 0 makes the neighbours' early-exit threshold small, me.rs:768-769): in the extensive first
    # pass only the search patterns -- cross arms, the uneven multi-hexagon at its scales, hexagon
    # and square refinements -- can find the displacement
    ("p0", 320, 192, 8, ("delta", 4, -2), None, 1, False, 1, 0, 1, 0.6),
    ("p1", 320, 192, 8, ("delta", 2, 3), None, 1, False, 1, 0, 1, 0.6),
    ("p2", 320, 192, 8, ("delta", -2, -3), None, 1, False, 1, 0, 1, 0.6),
    ("p3", 320, 192, 8, ("delta", 8, 4), None, 1, False, 1, 0, 1, 0.6),
    ("p4", 320, 192, 10, ("delta", -12, 9), None, 1, False, 1, 0, 1, 2.4),
    ("p5", 320, 192, 8, ("delta", 0, -9), None, 1, False, 1, 1, 1, 0.6),
    ("p6", 320, 192, 8, ("delta", 5, 0), None, 1, False, 0, 0, 1, 0.6),
    ("p7", 320, 192, 8, ("delta", 1, 1), None, 1, False, 1, 0, 1, 0.6),
]


def images(rng, kind, w, h, bd, n_refs):
    if isinstance(kind, tuple):
        _, dy, dx = kind
        big = rng.integers(0, 1 << bd, (h + 256, w + 256))
        if (dy + dx) % 2:
            f = rng.standard_normal(big.shape)
            for _ in range(4):
                f = (np.roll(f, 1, 0) + 2 * f + np.roll(f, -1, 0)) / 4
                f = (np.roll(f, 1, 1) + 2 * f + np.roll(f, -1, 1)) / 4
            big = ((f - f.min()) / (f.max() - f.min()) * ((1 << bd) - 1)).astype(np.int64)
            big = np.clip(big + rng.integers(-3, 4, big.shape), 0, (1 << bd) - 1)
        org = big[128:128 + h, 128:128 + w]
        ref = big[128 + 4 * dy:128 + 4 * dy + h, 128 + 4 * dx:128 + 4 * dx + w].copy()
        ref[:, :64] = org[:, :64]
        return org, [ref] * n_refs
    if kind == "noise":
        return rng.integers(0, 1 << bd, (h, w)), [rng.integers(0, 1 << bd, (h, w)) for _ in range(n_refs)]
    if kind == "flat":
        org = np.full((h, w), 1 << (bd - 1), np.int64)
        org[::16, ::16] += 3
        ref = org.copy()
        ref[8::32, 8::32] -= 2
        return org, [ref] * n_refs
    big = smooth(rng, w, h, bd)
    org = big[32:32 + h, 32:32 + w]
    refs = []
    for k in range(n_refs):
        dx, dy = [(5, -3), (-9, 6), (12, 2)][k % 3] if kind == "shift" else [(2, 1), (-3, 2), (1, -4)][k % 3]
        r = big[32 - dy:32 - dy + h, 32 - dx:32 - dx + w]
        refs.append(np.clip(r + rng.integers(-2, 3, r.shape), 0, (1 << bd) - 1))
    return org, refs


def main():
    only = sys.argv[1:] or None
    c = L.crate("me.rs", "mc.rs", "dist.rs", "predict.rs", "context/block_unit.rs", "context/superblock_unit.rs")
    est_tile, est = c.get("estimate_tile_motion"), c.get("estimate_motion")
    SM = "MVSamplingMode"
    out = {}
    for ci, (name, w, h, bd, kind, tile, n_refs, use_prev, hp, full, scale, mel) in enumerate(TILE_CASES):
        if only and name not in only:
            continue
        t0 = time.time()
        rng = np.random.default_rng(7100 + ci)
        org, refs = images(rng, kind, w, h, bd, n_refs)
        cols, rows = (w + 3) // 4, (h + 3) // 4
        tile = tile or (0, 0, w, h)
        prevs = []
        for _ in range(n_refs):
            p = np.zeros((rows, cols, 3), np.int64)
            if use_prev:
                p[..., 0] = rng.integers(-96, 97, (rows, cols))
                p[..., 1] = rng.integers(-96, 97, (rows, cols))
                p[..., 2] = rng.integers(0, 1 << 20, (rows, cols))
            prevs.append(p)
        wd = World(c, bd, org, refs, prevs, tile, hp, full, scale, mel)
        g = L.pixel_type(bd)
        est_tile(g, wd.fi, wd.ts, wd.inter_cfg)
        out[name + "_meta"] = np.array([w, h, bd, *tile, hp, full, scale, n_refs, int(use_prev)])
        out[name + "_lambda"] = np.array(lambdas(mel), np.uint32)
        dt = L.np_dtype(bd)
        for s in range(3):
            out["%s_org%d" % (name, s)] = wd.org_imgs[s].astype(dt)
        for k in range(n_refs):
            for s in range(3):
                out["%s_ref%d_%d" % (name, k, s)] = wd.ref_imgs[k][s].astype(dt)
            out["%s_stats%d" % (name, k)] = wd.stats_array(k).astype(np.int64)
            out["%s_prev%d" % (name, k)] = prevs[k]
        print(name, "tile ME done in %.0f s" % (time.time() - t0), "nonzero mvs:",
              int((out[name + "_stats0"][..., :2] != 0).any(-1).sum()), flush=True)
        # ---- RDO-time estimate_motion(.., Some(pmv), corner, false, 0, None) on blocks of the tile,
        # from the statistics the three passes left (src/rdo.rs:1183-1196)
        if name in ("t0", "t2", "t4", "t5"):
            t0 = time.time()
            tx, ty, tw, th = tile
            blks, res = [], []
            TBO, BO = L.struct(c, "TileBlockOffset"), L.struct(c, "BlockOffset")
            sizes = [(16, 16), (8, 8), (32, 32), (64, 64), (16, 8), (8, 16), (32, 16), (64, 32)]
            nblk = 14 if bd == 8 else 8
            for bi in range(nblk):
                bw, bh = sizes[bi % len(sizes)]
                bx = int(rng.integers(0, max(1, (tw - bw) // bw + 1))) * (bw // 4)
                by = int(rng.integers(0, max(1, (th - bh) // bh + 1))) * (bh // 4)
                if bx * 4 + bw > tw or by * 4 + bh > th:
                    continue
                cm = int(rng.integers(0, 5))
                if cm == 0:
                    corner, code = L.enum(c, SM, "INIT"), 0
                else:
                    right, bottom = bool((cm - 1) & 1), bool((cm - 1) & 2)
                    corner = R.REnum(SM, "CORNER", 1, (right, bottom))
                    code = 1 | (right << 1) | (bottom << 2)
                pmv = [(int(rng.integers(-40, 41)), int(rng.integers(-40, 41))) for _ in range(2)]
                pm = R.Some(R.array(*[wd.MV(row=r_, col=c_) for r_, c_ in pmv]))
                rt = wd.ref_types[bi % n_refs]
                if n_refs > 1 and bi % n_refs != 0:
                    continue   # the batch entry point searches one reference per call: reference 0 here
                r = est(g, wd.fi, wd.ts, bw, bh, TBO(BO(x=bx, y=by)), rt, pm, corner, False, 0, R.NONE)
                r = r.p[0]
                blks.append((bx, by, bw, bh, code, pmv[0][0], pmv[0][1], pmv[1][0], pmv[1][1]))
                res.append((r.mv.row, r.mv.col, r.rd.sad, r.rd.cost))
            out[name + "_blk"] = np.array(blks, np.int64)
            out[name + "_blkout"] = np.array(res, np.int64)
            out[name + "_blkcfg"] = np.array([1, 0])
            print(name, len(blks), "block searches in %.0f s" % (time.time() - t0), flush=True)
    if only:
        print("partial run: nothing written")
        return
    L.save("me_ref.npz", out)


if __name__ == "__main__":
    main()
