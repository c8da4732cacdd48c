"""Independent model of the AV1 self-guided restoration filter as applied by
RestorationState::lrf_filter_frame (src/lrf.rs:1482-1585): the padded stripe is built
explicitly (rows outside the 64-row stripe come from the deblocked frame, at most two of
them, then replicate; columns outside the unit: up to 4 / 3 real pixels, replicate at the
frame edge), box sums are taken directly (no integral images), every output row is computed
from scratch (no rolling buffers).  AV1 specification 7.17.3 ("box filter process") and
7.17.2 in the encoder's stripe arrangement."""
import numpy as np

SGR_S = [(140, 3236), (112, 2158), (93, 1618), (80, 1438), (70, 1295), (58, 1177), (47, 1079),
         (37, 996), (30, 925), (25, 863), (0, 2589), (0, 1618), (0, 1177), (0, 925), (56, 0), (22, 0)]


def padded_stripe(cdef, debl, x0, y0, w, h, crop_w, crop_h):
    """S[j + 4][i + 4] for i in [-4, w + 3), j in [-4, h2 + 2)"""
    h2 = h + (h & 1)
    lu = 0 if x0 == 0 else 4
    ru = min(3, (crop_w - x0) - w)
    S = np.zeros((h2 + 6, w + 7), np.int64)
    for j in range(-4, h2 + 2):
        cy = min(max(y0 + j, 0), crop_h - 1)
        ly = min(max(cy, y0 - 2), y0 + h2 + 1)
        src = cdef if y0 <= ly < y0 + h2 else debl
        for i in range(-4, w + 3):
            xi = min(max(i, -lu), w + ru - 1)
            S[j + 4, i + 4] = src[ly, x0 + xi]
    return S


def ab(S, cx, cy, r, s, bd):
    """(a, b) of the (2r+1)^2 box centred on stripe pixel (cx, cy)"""
    win = S[cy + 4 - r:cy + 5 + r, cx + 4 - r:cx + 5 + r]
    n = (2 * r + 1) ** 2
    total, ssq = int(win.sum()), int((win * win).sum())
    sh = bd - 8
    sc_ssq = (ssq + ((1 << (2 * sh)) >> 1)) >> (2 * sh)
    sc_sum = (total + ((1 << sh) >> 1)) >> sh
    p = max(0, sc_ssq * n - sc_sum * sc_sum)
    z = (p * s + (1 << 19)) >> 20
    a = 256 if z >= 255 else (1 if z == 0 else ((z << 8) + z // 2) // (z + 1))
    one_by_n = 455 if r == 1 else 164
    b = ((256 - a) * total * one_by_n + (1 << 11)) >> 12
    return a, b


def sgr_unit(cdef, debl, out, x0, y0, w, h, crop_w, crop_h, set_, xqd, bd):
    S = padded_stripe(cdef, debl, x0, y0, w, h, crop_w, crop_h)
    s2, s1 = SGR_S[set_]
    w0, w1 = int(xqd[0]), int(xqd[1])
    w2 = 128 - w0 - w1
    for y in range(h):
        for x in range(w):
            p = int(cdef[y0 + y, x0 + x])
            # pass with radius 1: 3x3 weights (corners 3, others 4), shift 9
            if s1 > 0:
                A = B = 0
                for dy in (-1, 0, 1):
                    for dx in (-1, 0, 1):
                        wt = 3 if dx and dy else 4
                        a, b = ab(S, x + dx, y + dy, 1, s1, bd)
                        A += wt * a
                        B += wt * b
                f1 = (A * p + B + (1 << 8)) >> 9
            else:
                f1 = p << 4
            # pass with radius 2: (a, b) exist on the odd rows -1, 1, 3, ..
            if s2 > 0:
                if y % 2 == 0:
                    A = B = 0
                    for dy in (-1, 1):
                        for dx, wt in ((-1, 5), (0, 6), (1, 5)):
                            a, b = ab(S, x + dx, y + dy, 2, s2, bd)
                            A += wt * a
                            B += wt * b
                    f2 = (A * p + B + (1 << 8)) >> 9
                else:
                    A = B = 0
                    for dx, wt in ((-1, 5), (0, 6), (1, 5)):
                        a, b = ab(S, x + dx, y, 2, s2, bd)
                        A += wt * a
                        B += wt * b
                    f2 = (A * p + B + (1 << 7)) >> 8
            else:
                # the reference shares the even row's value with the odd row (lrf.rs:741-750);
                # with r0 = 0 the bitstream carries w0 = 0 and the value is unused
                f2 = int(cdef[y0 + (y & ~1), x0 + x]) << 4
            v = w0 * f2 + w1 * (p << 4) + w2 * f1
            out[y0 + y, x0 + x] = min(max((v + (1 << 10)) >> 11, 0), (1 << bd) - 1)


def lrf_plane(cdef, debl, ydec, crop_w, crop_h, frame_h, unit_size, units, stripe_height, bd):
    """units: structured array (rows, cols) with filter / set / xqd -> filtered copy of cdef"""
    out = cdef.copy()
    rows, cols = units.shape
    for si in range((frame_h + 7) // 64 + 1):
        if si == 0:
            y0, sz = 0, (64 - 8) >> ydec
        else:
            y0 = (si * 64 - 8) >> ydec
            sz = min(64 >> ydec, crop_h - y0)
        if sz <= 0:
            continue
        for rux in range(cols):
            x0 = rux * unit_size
            size = crop_w - x0 if rux == cols - 1 else unit_size
            u = units[min(si * stripe_height // unit_size, rows - 1), rux]
            if u["filter"] == 3:
                sgr_unit(cdef, debl, out, x0, y0, size, sz, crop_w, crop_h, int(u["set"]), u["xqd"], bd)
    return out


def solve_unit(cdef, src, x0, y0, w, h, set_, bd):
    """sgrproj_solve (src/lrf.rs:847-1096) on the unit hard-clipped at its right / bottom edge as
    rdo_loop_decision sets it up (src/rdo.rs:2651-2676): independent restatement -- per-pixel box
    filters from the padded unit, exact integer moments, the 2x2 solve in IEEE doubles with the
    reference's fused multiply-adds emulated through exact rationals."""
    from fractions import Fraction as Fr
    S = padded_stripe(cdef, cdef, x0, y0, w, h, x0 + w, y0 + h)
    s2, s1 = SGR_S[set_]
    h00 = h11 = h01 = c0 = c1 = 0
    for y in range(h):
        for x in range(w):
            p = int(cdef[y0 + y, x0 + x])
            if s1 > 0:
                A = B = 0
                for dy in (-1, 0, 1):
                    for dx in (-1, 0, 1):
                        wt = 3 if dx and dy else 4
                        a, b = ab(S, x + dx, y + dy, 1, s1, bd)
                        A += wt * a
                        B += wt * b
                f1 = (A * p + B + (1 << 8)) >> 9
            else:
                f1 = p << 4
            if s2 > 0:
                A = B = 0
                if y % 2 == 0:
                    for dy in (-1, 1):
                        for dx, wt in ((-1, 5), (0, 6), (1, 5)):
                            a, b = ab(S, x + dx, y + dy, 2, s2, bd)
                            A += wt * a
                            B += wt * b
                    f2 = (A * p + B + (1 << 8)) >> 9
                else:
                    for dx, wt in ((-1, 5), (0, 6), (1, 5)):
                        a, b = ab(S, x + dx, y, 2, s2, bd)
                        A += wt * a
                        B += wt * b
                    f2 = (A * p + B + (1 << 7)) >> 8
            else:
                f2 = int(cdef[y0 + (y & ~1), x0 + x]) << 4
            u = p << 4
            sv = (int(src[y0 + y, x0 + x]) << 4) - u
            f1 -= u
            f2 -= u
            h00 += f2 * f2
            h11 += f1 * f1
            h01 += f1 * f2
            c0 += f2 * sv
            c1 += f1 * sv
    n = float(w) * float(h)
    H00, H01, H11 = float(h00) / n, float(h01) / n, float(h11) / n
    C0, C1 = float(c0) * (128.0 / n), float(c1) * (128.0 / n)

    def fma(a, b, c):
        return float(Fr(a) * Fr(b) + Fr(c))

    def rnd(v):          # f64::round: half away from zero
        import math
        return int(math.floor(abs(v) + 0.5)) * (1 if v >= 0 else -1)
    xq0 = xq1 = 0
    if s2 == 0:
        if H11 != 0.0:
            xq1 = rnd(C1 / H11)
    elif s1 == 0:
        if H00 != 0.0:
            xq0 = rnd(C0 / H00)
    else:
        det = fma(H00, H11, -(H01 * H01))
        if det != 0.0:
            xq0 = rnd(fma(H11, C0, -(H01 * C1)) / det)
            xq1 = rnd(fma(H00, C1, -(H01 * C0)) / det)
    xqd0 = min(max(xq0, -96), 31)
    xqd1 = min(max(128 - xqd0 - xq1, -32), 95)
    return xqd0, xqd1
