#!/usr/bin/env python3
"""Generate tests/golden/predict_golden.npz: intra-prediction vectors from an
INDEPENDENT model written in the AV1 specification's own formulation
(section 7.11.2: AboveRow[-1..], LeftCol[-1..] with negative indices, the
intra edge filter / upsample processes 7.11.2.9-12 acting in place), not in the
reference's offset-array formulation (src/predict.rs:1301-1505), which is what
oracle/predict.c restates.  The reference's own known answers (4x4, 10 modes +
27 angles, src/predict.rs:1523-1618) are checked in tests/test_oracle_predict.py;
these vectors extend the pin to every size, angle delta, edge-filter and
upsample path, for which the reference holds no vectors ("parity unpinned").

Tables (Sm_Weights, Dr_Intra_Derivative) are spec data read from
oracle/intra_tables.inc.  Edge buffers use the reference's IntraEdgeBuffer
layout (left right-aligned ending at index 128 bottom->top, top-left at 128,
above from 129) so the same arrays feed the oracle and the GPU.
"""
import os
import re
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
TX_W = [4, 8, 16, 32, 64, 4, 8, 8, 16, 16, 32, 32, 64, 4, 16, 8, 32, 16, 64]
TX_H = [4, 8, 16, 32, 64, 8, 4, 16, 8, 32, 16, 64, 32, 16, 4, 32, 8, 64, 16]
MODE_ANGLE = {1: 90, 2: 180, 3: 45, 4: 135, 5: 113, 6: 157, 7: 203, 8: 67}


def tables():
    t = open(os.path.join(ROOT, "oracle", "intra_tables.inc")).read()
    sm = [int(x) for x in re.search(r"kR1SmWeights\[128\] = \{(.*?)\}", t).group(1).split(",")]
    dr = [int(x) for x in re.search(r"kR1DrIntraDerivative\[91\] = \{(.*?)\}", t).group(1).split(",")]
    return sm, dr


SM, DR = tables()


class Neg:
    """array indexable from -2"""

    def __init__(self, n):
        self.a = [0] * (n + 2)

    def __getitem__(self, i):
        return self.a[i + 2]

    def __setitem__(self, i, v):
        self.a[i + 2] = v


def strength_sel(w, h, ftype, delta):      # spec 7.11.2.9
    d, bwh = abs(delta), w + h
    s = 0
    if ftype == 0:
        if bwh <= 8:
            s = 1 if d >= 56 else 0
        elif bwh <= 16:
            s = 1 if d >= 40 else 0
        elif bwh <= 24:
            s = 1 if d >= 8 else 0
            if d >= 16: s = 2
            if d >= 32: s = 3
        elif bwh <= 32:
            s = 1
            if d >= 4: s = 2
            if d >= 32: s = 3
        else:
            s = 3
    else:
        if bwh <= 8:
            if d >= 40: s = 1
            if d >= 64: s = 2
        elif bwh <= 16:
            if d >= 20: s = 1
            if d >= 48: s = 2
        elif bwh <= 24:
            if d >= 4: s = 3
        else:
            s = 3
    return s


def use_upsample(w, h, ftype, delta):      # spec 7.11.2.10
    d, bwh = abs(delta), w + h
    if d <= 0 or d >= 40:
        return 0
    return int(bwh <= 8) if ftype else int(bwh <= 16)


def edge_filter(buf, sz, strength):        # spec 7.11.2.12 (buf indexed from -1)
    K = [[0, 4, 8, 4, 0], [0, 5, 6, 5, 0], [2, 4, 4, 4, 2]]
    if strength == 0:
        return
    edge = [buf[i - 1] for i in range(sz)]
    for i in range(1, sz):
        s = 0
        for j in range(5):
            k = min(max(i - 2 + j, 0), sz - 1)
            s += K[strength - 1][j] * edge[k]
        buf[i - 1] = (s + 8) >> 4


def edge_upsample(buf, num_px, bd):        # spec 7.11.2.11
    dup = [0] * (num_px + 3)
    dup[0] = buf[-1]
    for i in range(-1, num_px):
        dup[i + 2] = buf[i]
    dup[num_px + 2] = buf[num_px - 1]
    buf[-2] = dup[0]
    for i in range(num_px):
        s = -dup[i] + 9 * dup[i + 1] + 9 * dup[i + 2] - dup[i + 3]
        s = min(max((s + 8) >> 4, 0), (1 << bd) - 1)
        buf[2 * i - 1] = s
        buf[2 * i] = dup[i + 2]


def directional(edge, left_len, above_len, w, h, p_angle, bd, ief, avail_w, avail_h):
    n = 2 * (w + h) + 4
    A, L = Neg(n), Neg(n)
    for i in range(above_len):
        A[i] = int(edge[129 + i])
    lb = min(left_len, w + h)
    for i in range(lb):
        L[i] = int(edge[127 - i])
    A[-1] = L[-1] = int(edge[128])
    up_a = up_l = 0
    if ief:
        ft = 1 if ief == 2 else 0
        if p_angle != 90 and p_angle != 180:
            edge_filter(A, min(w, avail_w) + (h if p_angle < 90 else 0) + 1,
                        strength_sel(w, h, ft, p_angle - 90))
            edge_filter(L, min(h, avail_h) + (w if p_angle > 180 else 0) + 1,
                        strength_sel(w, h, ft, p_angle - 180))
        up_a = use_upsample(w, h, ft, p_angle - 90)
        if up_a:
            edge_upsample(A, w + (h if p_angle < 90 else 0), bd)
        up_l = use_upsample(w, h, ft, p_angle - 180)
        if up_l:
            edge_upsample(L, h + (w if p_angle > 180 else 0), bd)
    dx = DR[p_angle] if p_angle < 90 else (DR[180 - p_angle] if 90 < p_angle < 180 else 0)
    dy = DR[p_angle - 90] if 90 < p_angle < 180 else (DR[270 - p_angle] if p_angle > 180 else 0)
    out = np.zeros((h, w), np.int64)
    for i in range(h):
        for j in range(w):
            if p_angle < 90:
                idx = (i + 1) * dx
                base = (idx >> (6 - up_a)) + (j << up_a)
                sh = ((idx << up_a) >> 1) & 31
                mb = (w + h - 1) << up_a
                v = (A[base] * (32 - sh) + A[base + 1] * sh + 16) >> 5 if base < mb else A[mb]
            elif 90 < p_angle < 180:
                idx = (j << 6) - (i + 1) * dx
                base = idx >> (6 - up_a)
                if base >= -(1 << up_a):
                    sh = ((idx << up_a) >> 1) & 31
                    v = (A[base] * (32 - sh) + A[base + 1] * sh + 16) >> 5
                else:
                    idx = (i << 6) - (j + 1) * dy
                    base = idx >> (6 - up_l)
                    sh = ((idx << up_l) >> 1) & 31
                    v = (L[base] * (32 - sh) + L[base + 1] * sh + 16) >> 5
            elif p_angle > 180:
                idx = (j + 1) * dy
                base = (idx >> (6 - up_l)) + (i << up_l)
                sh = ((idx << up_l) >> 1) & 31
                v = (L[base] * (32 - sh) + L[base + 1] * sh + 16) >> 5
            elif p_angle == 90:
                v = A[j]
            else:
                v = L[i]
            out[i, j] = min(max(v, 0), (1 << bd) - 1)
    return out


def non_directional(mode, variant, edge, w, h, bd, alpha, ac):
    above = edge[129:129 + w].astype(np.int64)
    left = edge[127 - np.arange(h)].astype(np.int64)       # top -> bottom
    tl = int(edge[128])
    if mode in (0, 13):
        if variant == 0:
            dc = 128 << (bd - 8)
        elif variant == 1:
            dc = (left.sum() + (h >> 1)) // h
        elif variant == 2:
            dc = (above.sum() + (w >> 1)) // w
        else:
            dc = (left.sum() + above.sum() + ((w + h) >> 1)) // (w + h)
        out = np.full((h, w), dc, np.int64)
        if mode == 13 and alpha != 0:
            q6 = alpha * ac.astype(np.int64).reshape(h, w)
            q0 = np.sign(q6) * ((np.abs(q6) + 32) >> 6)
            out = np.clip(dc + q0, 0, (1 << bd) - 1)
        return out
    if mode == 12:                                            # Paeth, spec 7.11.2.2
        base = above[None, :] + left[:, None] - tl
        pl, pt, ptl = np.abs(base - left[:, None]), np.abs(base - above[None, :]), np.abs(base - tl)
        return np.where((pl <= pt) & (pl <= ptl), left[:, None] + 0 * base,
                        np.where(pt <= ptl, above[None, :] + 0 * base, tl))
    ww = np.array(SM[w:2 * w], np.int64)
    wh = np.array(SM[h:2 * h], np.int64)
    if mode == 9:
        p = (wh[:, None] * above[None, :] + (256 - wh[:, None]) * left[h - 1] +
             ww[None, :] * left[:, None] + (256 - ww[None, :]) * above[w - 1])
        return (p + 256) >> 9
    if mode == 10:
        return (wh[:, None] * above[None, :] + (256 - wh[:, None]) * left[h - 1] + 128) >> 8
    if mode == 11:
        return (ww[None, :] * left[:, None] + (256 - ww[None, :]) * above[w - 1] + 128) >> 8
    raise ValueError(mode)


def main():
    rng = np.random.default_rng(2026)
    recs = {k: [] for k in ("ts", "mode", "variant", "angle", "ief", "bd", "left_len", "above_len",
                            "avail_w", "avail_h", "off")}
    edges, outs, acs = [], [], []
    off = 0
    for ts in range(19):
        w, h = TX_W[ts], TX_H[ts]
        for bd in (8, 10) if max(w, h) <= 16 else ((8,) if ts % 2 else (10,)):
            def new_edge(smooth_edge):
                if smooth_edge:
                    e = np.cumsum(rng.integers(-6, 7, 257)) + (1 << (bd - 1))
                    return np.clip(e, 0, (1 << bd) - 1).astype(np.uint16)
                return rng.integers(0, 1 << bd, 257).astype(np.uint16)

            def emit(mode, variant, angle, ief, left_len, above_len, aw, ah, out, e, ac=None):
                for k, v in zip(recs, (ts, mode, variant, angle, ief, bd, left_len, above_len, aw,
                                       ah, off)):
                    recs[k].append(v)
                edges.append(e)
                outs.append(out.astype(np.uint16).ravel())
                acs.append(ac if ac is not None else np.zeros(0, np.int16))

            # directional: base modes x angle deltas x edge-filter settings
            for mode, base in MODE_ANGLE.items():
                deltas = (-3, -2, -1, 0, 1, 2, 3) if w * h <= 256 else (-3, 0, 2)
                for d in deltas:
                    p = base + 3 * d
                    if p in (90, 180) and mode not in (1, 2):
                        continue
                    for ief in (0, 1, 2):
                        e = new_edge(ief != 0)
                        above_len = w + (h if p < 90 else 0)
                        left_len = h + (w if p > 180 else 0)
                        aw = w if rng.random() < 0.7 else int(rng.integers(1, w + 1))
                        ah = h if rng.random() < 0.7 else int(rng.integers(1, h + 1))
                        out = directional(e, left_len, above_len, w, h, p, bd, ief, aw, ah)
                        emit(mode, 3, p, ief, left_len, above_len, aw, ah, out, e)
                        off += w * h
            # non-directional
            for mode in (0, 9, 10, 11, 12, 13):
                for variant in ((0, 1, 2, 3) if mode in (0, 13) else (3,)):
                    e = new_edge(False)
                    alpha = int(rng.integers(-16, 17)) if mode == 13 else 0
                    if mode == 13 and alpha == 0:
                        alpha = 5
                    ac = None
                    if mode == 13:
                        ac = rng.integers(-(1 << (bd + 2)), 1 << (bd + 2), w * h).astype(np.int16)
                        ac -= np.int16(ac.astype(np.int64).sum() // (w * h))
                    out = non_directional(mode, variant, e, w, h, bd, alpha, ac)
                    emit(mode, variant, alpha, 0, h, w, w, h, out, e, ac)
                    off += w * h
    d = {k: np.asarray(v, np.int32) for k, v in recs.items()}
    d["edges"] = np.stack(edges)
    d["out"] = np.concatenate(outs)
    d["ac_off"] = np.cumsum([0] + [len(a) for a in acs]).astype(np.int64)
    d["ac"] = np.concatenate(acs)
    path = os.path.join(HERE, "predict_golden.npz")
    np.savez_compressed(path, **d)
    print("wrote %s: %d cases, %.1f KiB" % (path, len(edges), os.path.getsize(path) / 1024))


if __name__ == "__main__":
    sys.exit(main())
