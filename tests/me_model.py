"""A second, independent restatement of the reference's hierarchical motion
estimation -- estimate_tile_motion and everything it reaches with pmv = None
(src/me.rs:153-1523) -- used to cross-check oracle/me.c.  Test infrastructure.

The reference holds no vectors and no tests for me.rs, so the C oracle (a
line-by-line restatement written for speed: explicit state, flattened loops)
was "parity unpinned".  This model was written from the reference text again,
without looking at oracle/me.c, in a different style: plain Python integers,
(row, col) tuples, NumPy only for the pixel sums, np.float32 for the one f32
expression.  Two transcriptions that agree bit for bit on every MEStats entry
do not prove the reference's behaviour, but they rule out the slips a single
transcription can hide (a pattern entry, a clamp, an integer-division sign).
Only the full-pel path exists here: with pmv = None estimate_motion never runs
the sub-pel search (me.rs:596-620).
"""
import numpy as np

MI = 4
U32_MAX, U64_MAX = (1 << 32) - 1, (1 << 64) - 1
MV_LOW, MV_UPP = -(1 << 14), 1 << 14                      # src/context/mod.rs:131-132


def tdiv(a, b):
    """Rust's `/` on signed integers: truncation toward zero"""
    q = abs(a) // abs(b)
    return q if (a >= 0) == (b >= 0) else -q


def ilog(v):
    """v_frame ILog::ilog: bits needed for v (0 -> 0)"""
    return int(v).bit_length()


DIAMOND = [(1 << 3, 0), (0, 1 << 3), (-1 << 3, 0), (0, -1 << 3)]          # (row, col); me.rs:932-935
HEXAGON = [(r << 3, c << 3) for c, r in zip([0, 2, 2, 0, -2, -2], [-2, -1, 1, 2, 1, -1])]
SQUARE = [(r << 3, c << 3) for c, r in zip([-1, 0, 1, -1, 1, -1, 0, 1], [1, 1, 1, 0, 0, -1, -1, -1])]
UMH = [(r << 3, c << 3) for c, r in zip([-2, -1, 0, 1, 2, 3, 4, 3, 2, 1, 0, -1, -2, 3, -4, -3],
                                         [4, 4, 4, 4, 4, 2, 0, -2, -4, -4, -4, -4, -4, -2, 0, 2])]


class Model:
    def __init__(self, org3, ref3, w_in_b, h_in_b, bit_depth, lambdas, allow_hp=1,
                 allow_full_search=0, me_range_scale=1):
        """org3 / ref3: [full, half, quarter] planes with .data / .xorigin / .yorigin (padding
        edge-replicated, as Plane::pad leaves it); w_in_b, h_in_b: frame size in 4x4 units"""
        self.org = [np.asarray(p.data, np.int64) for p in org3]
        self.ref = [np.asarray(p.data, np.int64) for p in ref3]
        self.oo = [(p.yorigin, p.xorigin) for p in org3]
        self.ro = [(p.yorigin, p.xorigin) for p in ref3]
        self.w_in_b, self.h_in_b, self.bd = w_in_b, h_in_b, bit_depth
        self.lambdas, self.allow_hp = lambdas, allow_hp
        self.allow_full_search, self.me_range_scale = allow_full_search, me_range_scale

    # ---- cost of one candidate ------------------------------------------------------------
    def mv_rate(self, a, b):                                       # me.rs:1509-1523
        def rate(d):
            if not self.allow_hp:
                d >>= 1
            return 2 * ilog(abs(d))
        return rate(a[0] - b[0]) + rate(a[1] - b[1])

    def sad_at(self, ss, ox, oy, rx, ry, w, h):
        oy0, ox0 = self.oo[ss]
        ry0, rx0 = self.ro[ss]
        a = self.org[ss][oy0 + oy:oy0 + oy + h, ox0 + ox:ox0 + ox + w]
        b = self.ref[ss][ry0 + ry:ry0 + ry + h, rx0 + rx:rx0 + rx + w]
        assert a.shape == (h, w) and b.shape == (h, w), "search left the padded plane"
        return int(np.abs(a - b).sum())

    def mv_rd(self, blk, mv, pmv=None):                            # compute_mv_rd, me.rs:1445-1463
        ss, px, py, w, h, lam = blk["ss"], blk["px"], blk["py"], blk["w"], blk["h"], blk["lam"]
        pmv = pmv if pmv is not None else blk.get("pmv", ((0, 0), (0, 0)))
        sad = self.sad_at(ss, px, py, px + tdiv(mv[1], 8), py + tdiv(mv[0], 8), w, h)
        rate = min(self.mv_rate(mv, pmv[0]), self.mv_rate(mv, pmv[1]) + 1)
        return (256 * sad + rate * lam, sad)

    def fullpel_rd(self, blk, mv):                                 # get_fullpel_mv_rd, me.rs:1386-1410
        x0, x1, y0, y1 = blk["rng"]
        if mv[1] < x0 or mv[1] > x1 or mv[0] < y0 or mv[0] > y1:
            return (U64_MAX, U32_MAX)
        return self.mv_rd(blk, mv)

    # ---- searches: state = [mv, (cost, sad)] -----------------------------------------------
    def best_predictor(self, blk, preds):                          # me.rs:870-890
        best = [(0, 0), (U64_MAX, U32_MAX)]
        for mv in preds:
            rd = self.fullpel_rd(blk, mv)
            if rd[0] < best[1][0]:
                best = [mv, rd]
        return best

    def diamond(self, blk, cur):                                   # me.rs:952-995
        log2 = 1
        while True:
            cand = [(0, 0), (U64_MAX, U32_MAX)]
            for (dr, dc) in DIAMOND:
                mv = (cur[0][0] + (dr << log2), cur[0][1] + (dc << log2))
                rd = self.fullpel_rd(blk, mv)
                if rd[0] < cand[1][0]:
                    cand = [mv, rd]
            if cur[1][0] <= cand[1][0]:
                if log2 == 0:
                    break
                log2 -= 1
            else:
                cur = cand
        return cur

    def hexagon(self, blk, cur):                                   # me.rs:1056-1136
        idx, cand = 0, [(0, 0), (U64_MAX, U32_MAX)]
        for i in range(6):
            mv = (cur[0][0] + HEXAGON[i][0], cur[0][1] + HEXAGON[i][1])
            rd = self.fullpel_rd(blk, mv)
            if rd[0] < cand[1][0]:
                idx, cand = i, [mv, rd]
        while cand[1][0] < cur[1][0]:
            cur, centre = cand, idx
            cand = [(0, 0), (U64_MAX, U32_MAX)]
            for off in (5, 6, 7):
                i = (centre + off) % 6
                mv = (cur[0][0] + HEXAGON[i][0], cur[0][1] + HEXAGON[i][1])
                rd = self.fullpel_rd(blk, mv)
                if rd[0] < cand[1][0]:
                    idx, cand = i, [mv, rd]
        cand = [(0, 0), (U64_MAX, U32_MAX)]
        for (dr, dc) in SQUARE:
            mv = (cur[0][0] + dr, cur[0][1] + dc)
            rd = self.fullpel_rd(blk, mv)
            if rd[0] < cand[1][0]:
                cand = [mv, rd]
        return cand if cand[1][0] < cur[1][0] else cur

    def umh(self, blk, cur, me_range=24):                          # me.rs:1172-1302
        def consider(mv):
            nonlocal cur
            rd = self.fullpel_rd(blk, mv)
            if rd[0] < cur[1][0]:
                cur = [mv, rd]
        centre = cur[0]
        # "HORIZONTAL_LINE" is declared with col 0 and row +-1: the long arm steps the ROW
        for i in range(1, me_range + 1, 2):
            for (dr, dc) in ((-1 << 3, 0), (1 << 3, 0)):
                consider((centre[0] + dr * i, centre[1] + dc * i))
        for i in range(1, (me_range >> 1) + 1, 2):
            for (dr, dc) in ((0, -1 << 3), (0, 1 << 3)):
                consider((centre[0] + dr * i, centre[1] + dc * i))
        centre = cur[0]
        for row in range(-2, 3):                 # the 5x5 offsets are NOT scaled to full pel
            for col in range(-2, 3):
                if row or col:
                    consider((centre[0] + row, centre[1] + col))
        centre = cur[0]
        for i in range(1, (me_range >> 2) + 1):
            for (dr, dc) in UMH:
                consider((centre[0] + dr * i, centre[1] + dc * i))
        return self.hexagon(blk, cur)

    def full_search(self, blk, x_lo, x_hi, y_lo, y_hi, step):      # me.rs:1466-1507
        best = [(0, 0), (U64_MAX, U32_MAX)]
        for y in range(y_lo, y_hi + 1, step):
            for x in range(x_lo, x_hi + 1, step):
                mv = (8 * (y - blk["py"]), 8 * (x - blk["px"]))
                rd = self.mv_rd(blk, mv, ((0, 0), (0, 0)))         # both callers pass zero pmv
                if rd[0] < best[1][0]:
                    best = [mv, rd]
        return best

    # ---- predictors -----------------------------------------------------------------------
    def subsets(self, stats, prev, tile, bx, by, pix_w, pix_h, rng, corner, ssdec):
        """get_subset_predictors (me.rs:386-533); stats: the frame's MEStats array (structured,
        fields row / col / normalized_sad), tile = (x, y, w, h) in pixels; bx, by tile-relative"""
        tx, ty = tile[0] // MI, tile[1] // MI
        tcols, trows = tile[2] // MI, tile[3] // MI
        x0, x1, y0, y1 = rng
        min_sad = U32_MAX

        def cand(s):
            nonlocal min_sad
            min_sad = min(min_sad, int(s["normalized_sad"]))
            r, c = tdiv(int(s["row"]), 8) * 8, tdiv(int(s["col"]), 8) * 8
            return (min(max(r, y0), y1), min(max(c, x0), x1))

        def at(y, x):
            return stats[ty + y, tx + x]
        w = ((pix_w << ssdec) + MI - 1) >> 2
        h = ((pix_h << ssdec) + MI - 1) >> 2
        chw, chh = min(w >> 1, tcols - 1 - bx), min(h >> 1, trows - 1 - by)
        b, c = [], []
        if bx > 0:
            b.append(cand(at(by + chh, bx - 1)))
        if by > 0:
            b.append(cand(at(by - 1, bx + chw)))
        if corner is not None and corner[0] and bx + w < tcols:
            b.append(cand(at(by + chh, bx + w)))
        if corner is not None and corner[1] and by + h < trows:
            b.append(cand(at(by + h, bx + chw)))
        if corner is not None:
            median = cand(at(by + chh, bx + chw))
        elif len(b) != 3:
            median = None
        else:
            median = (sorted(m[0] for m in b)[1], sorted(m[1] for m in b)[1])
        b.append((0, 0))
        if prev is not None:
            rows, cols = prev.shape
            fx, fy = tx + bx, ty + by
            phw, phh = min(w >> 1, cols - 1 - fx), min(h >> 1, rows - 1 - fy)
            if fx > 0:
                c.append(cand(prev[fy + phh, fx - 1]))
            if fy > 0:
                c.append(cand(prev[fy - 1, fx + phw]))
            if fx + w < cols:
                c.append(cand(prev[fy + phh, fx + w]))
            if fy + h < rows:
                c.append(cand(prev[fy + h, fx + phw]))
            c.append(cand(prev[fy + phh, fx + phw]))
        min_sad = ((min_sad * (pix_w * pix_h)) >> 14) & U32_MAX
        dec = lambda m: (m[0] >> ssdec, m[1] >> ssdec)
        return dict(min_sad=min_sad, median=None if median is None else dec(median),
                    b=[dec(m) for m in b], c=[dec(m) for m in c])

    def mv_range(self, fbx, fby, blk_w, blk_h):                    # get_mv_range, me.rs:339-362
        bw, bh = 128 + blk_w * 8, 128 + blk_h * 8
        x0 = -fbx * 32 - bw
        x1 = (self.w_in_b - fbx - blk_w // MI) * 32 + bw
        y0 = -fby * 32 - bh
        y1 = (self.h_in_b - fby - blk_h // MI) * 32 + bh
        return (max(x0, MV_LOW + 1), min(x1, MV_UPP - 1), max(y0, MV_LOW + 1), min(y1, MV_UPP - 1))

    def block(self, tile, bx, by, w, h, ssdec):
        fbx, fby = tile[0] // MI + bx, tile[1] // MI + by
        x0, x1, y0, y1 = self.mv_range(fbx, fby, w << ssdec, h << ssdec)
        return dict(ss=ssdec, px=(fbx * MI) >> ssdec, py=(fby * MI) >> ssdec, w=w, h=h,
                    lam=self.lambdas[ssdec], rng=(x0 >> ssdec, x1 >> ssdec, y0 >> ssdec, y1 >> ssdec))

    # ---- estimate_motion with pmv = None, refine, the tile loop ------------------------------
    def full_pixel_me(self, blk, sub, extensive, ssdec):           # me.rs:682-858
        best = [(0, 0), (U64_MAX, U32_MAX)]

        def try_cands(preds):
            nonlocal best
            r = self.diamond(blk, self.best_predictor(blk, preds))
            if r[1][0] < best[1][0]:
                best = r
        if not extensive:
            try_cands(([sub["median"]] if sub["median"] is not None else []) + sub["b"] + sub["c"])
            return best
        f = np.float32(sub["min_sad"]) * np.float32(1.2)
        thresh = (min(int(f), U32_MAX) + (((blk["w"] * blk["h"]) << (self.bd - 8)) & U32_MAX)) & U32_MAX
        if sub["median"] is not None:
            try_cands([sub["median"]])
            if best[1][1] < thresh:
                return best
        try_cands(sub["b"])
        if best[1][1] < thresh:
            return best
        try_cands(sub["c"])
        if best[1][1] < thresh:
            return best
        best = self.umh(blk, best)
        if not self.allow_full_search or best[1][1] < thresh:
            return best
        rx, ry = (192 * self.me_range_scale) >> ssdec, (64 * self.me_range_scale) >> ssdec
        x0, x1, y0, y1 = blk["rng"]
        r = self.full_search(blk, blk["px"] + max(-rx, tdiv(x0, 8)), blk["px"] + min(rx, tdiv(x1, 8)),
                             blk["py"] + max(-ry, tdiv(y0, 8)), blk["py"] + min(ry, tdiv(y1, 8)), 4 >> ssdec)
        return r if r[1][0] < best[1][0] else best

    def save(self, stats, tile, bx, by, size_in_b, mv, sad, w, h):
        nsad = ((sad << 14) // (w * h)) & U32_MAX
        tx, ty = tile[0] // MI, tile[1] // MI
        xe, ye = min(bx + size_in_b, tile[2] // MI), min(by + size_in_b, tile[3] // MI)
        v = stats[ty + by:ty + ye, tx + bx:tx + xe]
        v["row"], v["col"], v["normalized_sad"] = mv[0], mv[1], nsad

    def estimate_tile_motion(self, stats, tile, prev=None):        # me.rs:153-218
        tw, th = tile[2], tile[3]
        prev_ss = None
        for log2b in (4, 3, 2):
            init = log2b == 4
            ssdec = {0: 2, 1: 1}.get(4 - log2b, 0)
            new_sub = prev_ss is not None and prev_ss != ssdec
            prev_ss = ssdec
            for sby in range((th + 63) // 64):
                for sbx in range((tw + 63) // 64):
                    sb_w, sb_h = min(64, tw - sbx * 64), min(64, th - sby * 64)
                    for (refine, lg) in ((True, log2b + 1), (False, log2b)):
                        if refine and not new_sub:
                            continue
                        size = MI << lg
                        for y in range(0, sb_h, size):
                            for x in range(0, sb_w, size):
                                bx, by = sbx * 16 + x // MI, sby * 16 + y // MI
                                w = min(size, sb_w - x + (1 << ssdec) - 1) >> ssdec
                                h = min(size, sb_h - y + (1 << ssdec) - 1) >> ssdec
                                blk = self.block(tile, bx, by, w, h, ssdec)
                                if refine:                          # me.rs:622-680
                                    s = stats[tile[1] // MI + by, tile[0] // MI + bx]
                                    mr, mc = int(s["row"]) >> ssdec, int(s["col"]) >> ssdec
                                    x0, x1, y0, y1 = blk["rng"]
                                    r = self.full_search(
                                        blk, blk["px"] + max(tdiv(mc, 8) - 1, tdiv(x0, 8)),
                                        blk["px"] + min(tdiv(mc, 8) + 2, tdiv(x1, 8)),
                                        blk["py"] + max(tdiv(mr, 8) - 1, tdiv(y0, 8)),
                                        blk["py"] + min(tdiv(mr, 8) + 2, tdiv(y1, 8)), 1)
                                else:
                                    corner = None if init else (bool(x & size), bool(y & size))
                                    # the range handed down is the DECIMATED one (estimate_motion shifts it
                                    # before full_pixel_me, me.rs:557-559), although the predictors it
                                    # clamps are still in full-resolution units at that point
                                    sub = self.subsets(stats, prev, tile, bx, by, w, h, blk["rng"], corner, ssdec)
                                    r = self.full_pixel_me(blk, sub, init, ssdec)
                                mv = (r[0][0] << ssdec, r[0][1] << ssdec)
                                self.save(stats, tile, bx, by, 1 << lg, mv, r[1][1], w, h)
        return stats


# ---- the RDO-time call: estimate_motion(.., Some(pmv), CORNER {..}, false, 0, None) ------------
# (src/rdo.rs:1183-1196 -> me.rs:536-620: full-pel search with the predicted MVs in the rate,
# SATD re-cost of the winner, sub-pel diamond through put_8tap)
def _hadamard(n):
    h = np.array([[1]], np.int64)
    while h.shape[0] < n:
        h = np.block([[h, h], [h, -h]])
    return h


def satd(a, b):
    """get_satd (src/dist.rs:156-221) for blocks whose sides are multiples of the tile size"""
    h, w = a.shape
    n = min(w, h, 8)
    H = _hadamard(n)
    total = 0
    for y in range(0, h, n):
        for x in range(0, w, n):
            d = a[y:y + n, x:x + n] - b[y:y + n, x:x + n]
            total += int(np.abs(H @ d @ H.T).sum())
    ln = n.bit_length() - 1
    return (total + ((1 << ln) >> 1)) >> ln


def _put_8tap(taps, win, w, h, cf, rf, mode, bd):
    """put_8tap (src/mc.rs:250-352) on a (h + 7, w + 7) window whose [3, 3] is the block origin"""
    rs = lambda v, b: (v + ((1 << b) >> 1)) >> b
    ib = 2 if bd == 12 else 4
    maxv = (1 << bd) - 1
    pick = lambda frac, length: taps[mode if (mode == 3 or length > 4) else min(mode, 1) + 4][frac]
    xf, yf = pick(cf, w), pick(rf, h)
    blk = win[3:3 + h, 3:3 + w]
    if cf == 0 and rf == 0:
        return blk.copy()
    if cf == 0:
        return np.clip(rs(sum(int(yf[k]) * win[k:k + h, 3:3 + w] for k in range(8)), 7), 0, maxv)
    if rf == 0:
        s = sum(int(xf[k]) * win[3:3 + h, k:k + w] for k in range(8))
        return np.clip(rs(rs(s, 7 - ib), ib), 0, maxv)
    mid = rs(sum(int(xf[k]) * win[:, k:k + w] for k in range(8)), 7 - ib)
    return np.clip(rs(sum(int(yf[k]) * mid[k:k + h, :] for k in range(8)), 7 + ib), 0, maxv)


def estimate_motion_block(m, taps, stats, prev, tile, bx, by, w, h, corner, pmv, use_satd, filter_mode=0):
    """m: Model; taps: SUBPEL_FILTERS (6, 16, 8); corner: (right, bottom); pmv: ((r, c), (r, c))
    -> (row, col, sad, cost)"""
    blk = m.block(tile, bx, by, w, h, 0)
    blk["pmv"] = pmv
    sub = m.subsets(stats, prev, tile, bx, by, w, h, blk["rng"], corner, 0)
    best = m.full_pixel_me(blk, sub, False, 0)
    px, py = blk["px"], blk["py"]
    oy0, ox0 = m.oo[0]
    ry0, rx0 = m.ro[0]
    org = m.org[0][oy0 + py:oy0 + py + h, ox0 + px:ox0 + px + w]
    x0, x1, y0, y1 = blk["rng"]

    def rd_of(pred, mv):
        d = satd(org, pred) if use_satd else int(np.abs(org - pred).sum())
        rate = min(m.mv_rate(mv, pmv[0]), m.mv_rate(mv, pmv[1]) + 1)
        return (256 * d + rate * blk["lam"], d)

    def in_range(mv):
        return x0 <= mv[1] <= x1 and y0 <= mv[0] <= y1
    if use_satd:                                   # me.rs:596-613
        mv = best[0]
        if in_range(mv):
            ry, rx = py + tdiv(mv[0], 8), px + tdiv(mv[1], 8)
            best = [mv, rd_of(m.ref[0][ry0 + ry:ry0 + ry + h, rx0 + rx:rx0 + rx + w], mv)]
        else:
            best = [mv, (U64_MAX, U32_MAX)]
    mc_w, mc_h = 1 << (w - 1).bit_length(), (h + 1) & ~1

    def subpel_rd(mv):                             # get_subpel_mv_rd, me.rs:1412-1442
        if not in_range(mv):
            return (U64_MAX, U32_MAX)
        ro, co = mv[0] >> 3, mv[1] >> 3            # get_mv_params, predict.rs:284-297 (luma)
        rf, cf = (mv[0] << 1) & 0xF, (mv[1] << 1) & 0xF
        wy, wx = ry0 + py + ro - 3, rx0 + px + co - 3
        win = m.ref[0][wy:wy + mc_h + 7, wx:wx + mc_w + 7]
        assert win.shape == (mc_h + 7, mc_w + 7), "sub-pel window left the padded plane"
        pred = _put_8tap(taps, win, mc_w, mc_h, cf, rf, filter_mode, m.bd)
        return rd_of(pred[:h, :w], mv)
    log2, end = 2, 0 if m.allow_hp else 1          # subpel_diamond_search, me.rs:1310-1383
    while True:
        cand = [(0, 0), (U64_MAX, U32_MAX)]
        for (dc, dr) in zip([0, 1, 0, -1], [1, 0, -1, 0]):
            mv = (best[0][0] + (dr << log2), best[0][1] + (dc << log2))
            rd = subpel_rd(mv)
            if rd[0] < cand[1][0]:
                cand = [mv, rd]
        if best[1][0] <= cand[1][0]:
            if log2 == end:
                break
            log2 -= 1
        else:
            best = cand
    return (best[0][0], best[0][1], best[1][1], best[1][0])
