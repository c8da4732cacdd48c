#!/usr/bin/env python3
"""Generate tests/golden/mc_golden.npz.

The AV1 sub-pel filter taps are parsed (regex) from the reference's own table
`SUBPEL_FILTERS` in /root/reference/src/mc.rs:110-219 -- the file is read
where it lies, never copied.  put_8tap / prep_8tap / mc_avg outputs are then
computed by the small vectorised NumPy model below, which is written from the
arithmetic of src/mc.rs:250-479 independently of oracle/mc.c (2-D convolution
over whole planes rather than per-pixel tap loops), so the C oracle is checked
against (a) the reference's tap data and (b) a second implementation.

Run in the build container:  python tests/golden/gen_mc_golden.py
"""
import os
import re

import numpy as np

REF = "/root/reference/src/mc.rs"
HERE = os.path.dirname(os.path.abspath(__file__))


def parse_filters():
    src = open(REF).read()
    m = re.search(r"const SUBPEL_FILTERS[^=]*=\s*\[(.*?)\n\];", src, re.S)
    nums = [int(x) for x in re.findall(r"-?\d+", m.group(1))]
    t = np.array(nums, dtype=np.int64).reshape(6, 16, 8)
    assert (t.sum(axis=2) == 128).all()
    return t


def rs(v, b):
    return (v + ((1 << b) >> 1)) >> b


def model(T, win, w, h, cf, rf, mx, my, bd):
    """win: (h+7, w+7) int64 window whose [3,3] is the block origin.
    returns put (h,w), prep (h,w) as int64"""
    ib = 2 if bd == 12 else 4
    maxv = (1 << bd) - 1
    bias = 0 if bd == 8 else 8192

    def filt(mode, frac, length):
        idx = mode if (mode == 3 or length > 4) else min(mode, 1) + 4
        return T[idx][frac]
    xf, yf = filt(mx, cf, w), filt(my, rf, h)
    blk = win[3:3 + h, 3:3 + w]
    if cf == 0 and rf == 0:
        return blk.copy(), (blk << ib) - bias
    if cf == 0:
        s = sum(yf[k] * win[k:k + h, 3:3 + w] for k in range(8))
        return np.clip(rs(s, 7), 0, maxv), rs(s, 7 - ib) - bias
    if rf == 0:
        s = sum(xf[k] * win[3:3 + h, k:k + w] for k in range(8))
        return np.clip(rs(rs(s, 7 - ib), ib), 0, maxv), rs(s, 7 - ib) - bias
    mid = rs(sum(xf[k] * win[:, k:k + w] for k in range(8)), 7 - ib)
    mid = mid.astype(np.int16).astype(np.int64)  # i16 intermediate
    s = sum(yf[k] * mid[k:k + h, :] for k in range(8))
    return np.clip(rs(s, 7 + ib), 0, maxv), rs(s, 7) - bias


def main():
    T = parse_filters()
    rng = np.random.default_rng(7)
    out = {"filters": T.astype(np.int16)}
    cases = []
    sizes = [(2, 2), (4, 4), (4, 8), (8, 4), (8, 8), (16, 8), (8, 16), (16, 16), (32, 16),
             (64, 64), (128, 8)]
    for bd in (8, 10, 12):
        for (w, h) in sizes:
            for rep in range(6):
                cf, rf = [(0, 0), (0, 4), (4, 0), (4, 4)][rep] if rep < 4 else \
                    (int(rng.integers(1, 16)), int(rng.integers(1, 16)))
                mx, my = int(rng.integers(0, 4)), int(rng.integers(0, 4))
                # extreme-valued windows every other case to exercise clamps
                if rep % 2:
                    win = rng.choice([0, (1 << bd) - 1], size=(h + 7, w + 7)).astype(np.int64)
                else:
                    win = rng.integers(0, 1 << bd, size=(h + 7, w + 7)).astype(np.int64)
                put, prep = model(T, win, w, h, cf, rf, mx, my, bd)
                k = "%d_%d_%d_%d_%d_%d_%d_%d" % (bd, w, h, cf, rf, mx, my, rep)
                dt = np.uint8 if bd == 8 else np.uint16
                out["win_" + k] = win.astype(dt)
                out["put_" + k] = put.astype(dt)
                out["prep_" + k] = prep.astype(np.int16)
                cases.append(k)
    # mc_avg (src/mc.rs:454-479)
    for bd in (8, 10, 12):
        ib = 2 if bd == 12 else 4
        t1 = rng.integers(-8192, 8192, size=(16, 16)).astype(np.int64)
        t2 = rng.integers(-8192, 8192, size=(16, 16)).astype(np.int64)
        bias = 0 if bd == 8 else 16384
        avg = np.clip(rs(t1 + t2 + bias, ib + 1), 0, (1 << bd) - 1)
        out["avg_t1_%d" % bd] = t1.astype(np.int16)
        out["avg_t2_%d" % bd] = t2.astype(np.int16)
        out["avg_out_%d" % bd] = avg.astype(np.uint8 if bd == 8 else np.uint16)
    out["cases"] = np.array(cases)
    path = os.path.join(HERE, "mc_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(cases), "cases")


if __name__ == "__main__":
    main()
