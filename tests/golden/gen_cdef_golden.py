#!/usr/bin/env python3
"""Generate tests/golden/cdef_golden.npz from an INDEPENDENT NumPy model of
CDEF written in the AV1 specification's formulation (section 7.15: per-pixel
CdefAvailable tests against the frame bounds, whole-plane vectorised taps with
an availability mask) -- not the reference's padded-u16-tile formulation with
the CDEF_VERY_LARGE sentinel (src/cdef.rs:161-298) that oracle/cdef.c restates.
The reference holds no CDEF vectors ("parity unpinned"); its one
known answer (first_max_element tie-break, cdef.rs:304-309) is checked in
tests/test_oracle_cdef.py.

Cases: small 4:2:0 / 4:4:4 frames (dims multiples of 8, not of 64), 8- and
10-bit, random skip flags, per-superblock strengths incl. zero primary /
secondary and the sec == 3 -> 4 rule; outputs for all three planes plus the
per-8x8 direction / variance maps.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
DIRS = [[(-1, 1), (-2, 2)], [(0, 1), (-1, 2)], [(0, 1), (0, 2)], [(0, 1), (1, 2)],
        [(1, 1), (2, 2)], [(1, 0), (2, 1)], [(1, 0), (2, 0)], [(1, 0), (2, -1)]]   # spec Cdef_Directions


def find_dir(blk, cs):
    """spec 7.15.2 on one 8x8 block (int array)"""
    x = (blk.astype(np.int64) >> cs) - 128
    part = np.zeros((8, 15), np.int64)
    for i in range(8):
        for j in range(8):
            v = x[i, j]
            part[0, i + j] += v
            part[1, i + j // 2] += v
            part[2, i] += v
            part[3, 3 + i - j // 2] += v
            part[4, 7 + i - j] += v
            part[5, 3 - i // 2 + j] += v
            part[6, j] += v
            part[7, i // 2 + j] += v
    div = [0, 840, 420, 280, 210, 168, 140, 120, 105]
    cost = np.zeros(8, np.int64)
    cost[2] = (part[2, :8] ** 2).sum() * 105
    cost[6] = (part[6, :8] ** 2).sum() * 105
    for d in (0, 4):
        cost[d] = part[d, 7] ** 2 * 105 + sum((part[d, i] ** 2 + part[d, 14 - i] ** 2) * div[i + 1]
                                               for i in range(7))
    for d in (1, 3, 5, 7):
        cost[d] = (part[d, 3:8] ** 2).sum() * 105 + sum(
            (part[d, j] ** 2 + part[d, 10 - j] ** 2) * div[2 * j + 2] for j in range(3))
    best = int(np.argmax(cost))          # first maximum
    return best, int((cost[best] - cost[(best + 4) & 7]) >> 10)


def msb(v):
    return int(v).bit_length() - 1


def constrain(diff, thr, damping):
    if thr == 0:
        return np.zeros_like(diff)
    shift = max(0, damping - msb(thr))
    ad = np.abs(diff)
    mag = np.clip(thr - (ad >> shift), 0, ad)
    return np.sign(diff) * mag


def filter_plane(src, W, H, xdec, ydec, blocks, bd):
    """blocks: list of (bx8, by8, pri, sec, dir, damping) in 8x8-luma units.
    Whole-plane formulation: a tap is available iff inside the frame."""
    out = src.copy().astype(np.int64)
    cs = bd - 8
    xs, ys = 8 >> xdec, 8 >> ydec
    for (bx, by, pri, sec, d, damp) in blocks:
        y0, x0 = by * ys, bx * xs
        yy, xx = np.mgrid[y0:y0 + ys, x0:x0 + xs]
        x = src[yy, xx].astype(np.int64)
        s = np.zeros_like(x)
        mx, mn = x.copy(), x.copy()
        ptap = [[4, 2], [3, 3]][(pri >> cs) & 1]
        stap = [2, 1]
        for k in range(2):
            for (dd, thr, tap) in ((d, pri, ptap[k]), ((d + 2) & 7, sec, stap[k]),
                                   ((d + 6) & 7, sec, stap[k])):
                oy, ox = DIRS[dd][k]
                for sgn in (1, -1):
                    ty, tx = yy + sgn * oy, xx + sgn * ox
                    # all-or-nothing halo: a tap is usable iff it lies inside the frame
                    av = (ty >= 0) & (ty < H) & (tx >= 0) & (tx < W)
                    p = src[np.clip(ty, 0, H - 1), np.clip(tx, 0, W - 1)].astype(np.int64)
                    c = constrain(p - x, thr, damp)
                    s += np.where(av, tap * c, 0)
                    mx = np.where(av, np.maximum(mx, p), mx)
                    mn = np.where(av, np.minimum(mn, p), mn)
        v = x + ((8 + s - (s < 0)) >> 4)
        out[yy, xx] = np.clip(v, mn, mx)
    return out


def adjust_strength(strength, var):
    i = min(msb(var >> 6), 12) if (var >> 6) else 0
    return (strength * (4 + i) + 8) >> 4 if var else 0


def run_frame(planes, bd, xdec, ydec, skip_mi, cdef_index, ystr, uvstr, damping):
    Y = planes[0]
    H, W = Y.shape
    cs = bd - 8
    nby, nbx = H // 8, W // 8
    dirs = np.zeros((nby, nbx), np.int32)
    vars_ = np.zeros((nby, nbx), np.int32)
    outs = []
    per_plane = [[], [], []]
    for by in range(nby):
        for bx in range(nbx):
            sk = skip_mi[2 * by:2 * by + 2, 2 * bx:2 * bx + 2].all()
            if sk:
                continue
            d, v = find_dir(Y[8 * by:8 * by + 8, 8 * bx:8 * bx + 8], cs)
            dirs[by, bx], vars_[by, bx] = d, v
            ci = cdef_index[by // 8, bx // 8]
            py, sy = int(ystr[ci]) // 4, int(ystr[ci]) % 4
            pu, su = int(uvstr[ci]) // 4, int(uvstr[ci]) % 4
            sy += sy == 3
            su += su == 3
            per_plane[0].append((bx, by, adjust_strength(py << cs, v), sy << cs, d if py else 0,
                                 damping + cs))
            duv = ([7, 0, 2, 4, 5, 6, 6, 6][d] if xdec != ydec else d) if pu else 0
            for p in (1, 2):
                per_plane[p].append((bx, by, pu << cs, su << cs, duv, damping + cs - 1))
    for p in range(3):
        xd, yd = (0, 0) if p == 0 else (xdec, ydec)
        ph, pw = planes[p].shape
        outs.append(filter_plane(planes[p], pw, ph, xd, yd, per_plane[p], bd))
    return outs, dirs, vars_


def main():
    rng = np.random.default_rng(99)
    out = {}
    case = 0
    for (W, H, xdec, ydec, bd) in ((72, 40, 1, 1, 8), (136, 72, 1, 1, 10), (64, 64, 0, 0, 8),
                                   (80, 24, 1, 0, 10), (24, 88, 1, 1, 12)):
        for rep in range(2):
            # piecewise-smooth content with edges so that directions / variances vary
            yy, xx = np.mgrid[0:H, 0:W]
            base = ((np.sin(xx / 5.0 + rep) + np.cos((yy + xx * (rep + 1)) / 7.0)) * 40 + 128)
            Y = np.clip(base + rng.integers(-20, 21, (H, W)), 0, 255).astype(np.int64) << (bd - 8)
            Y = np.clip(Y + rng.integers(0, 1 << (bd - 8), (H, W)), 0, (1 << bd) - 1)
            cw, ch = W >> xdec, H >> ydec
            U = rng.integers(0, 1 << bd, (ch, cw))
            V = np.clip((Y[::1 << ydec, ::1 << xdec][:ch, :cw] // 2 + rng.integers(-30, 31, (ch, cw))),
                        0, (1 << bd) - 1)
            skip = (rng.random((H // 4, W // 4)) < 0.35).astype(np.uint8)
            skip[: 2] = 1 if rep else skip[: 2]
            nsb_y, nsb_x = -(-H // 64), -(-W // 64)
            cdef_index = rng.integers(0, 8, (nsb_y, nsb_x)).astype(np.uint8)
            ystr = rng.integers(0, 64, 8).astype(np.uint8)
            uvstr = rng.integers(0, 64, 8).astype(np.uint8)
            ystr[0], uvstr[0] = 0, 3          # zero primary; sec == 3 -> 4
            ystr[1], uvstr[1] = 63, 60
            damping = int(rng.integers(3, 7))
            outs, dirs, vars_ = run_frame([Y, U, V], bd, xdec, ydec, skip, cdef_index, ystr, uvstr,
                                          damping)
            k = "c%d" % case
            out[k + "_meta"] = np.array([W, H, xdec, ydec, bd, damping], np.int32)
            for p, (a, o) in enumerate(zip((Y, U, V), outs)):
                out[k + "_in%d" % p] = a.astype(np.uint16)
                out[k + "_out%d" % p] = o.astype(np.uint16)
            out[k + "_skip"], out[k + "_ci"] = skip, cdef_index
            out[k + "_ystr"], out[k + "_uvstr"] = ystr, uvstr
            out[k + "_dir"], out[k + "_var"] = dirs, vars_
            case += 1
    path = os.path.join(HERE, "cdef_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote %s: %d frames, %.1f KiB" % (path, case, os.path.getsize(path) / 1024))


if __name__ == "__main__":
    sys.exit(main())
