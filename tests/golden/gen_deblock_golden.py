#!/usr/bin/env python3
"""Golden vectors for the deblocking filter and its level search.

The reference (src/deblock.rs) holds no vectors.  The filter is a normative AV1 decoder
process, so the expected frames here come from tests/deblock_util.py::spec_deblock_plane --
an independent model in the SPECIFICATION's formulation (see that file's header) -- and the
expected level-search tallies from brute force (every edge filtered at every level 0..63).
Inputs are synthetic: per-block DC steps + low-amplitude noise, so flat / non-flat, high-edge-
variance and masked-out lines all occur; block structures are random partitions.

    python tests/golden/gen_deblock_golden.py      # writes tests/golden/deblock_golden.npz
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import deblock_util as D   # noqa: E402


def blocky_image(rng, w, h, bd, blocks, xdec, ydec, noise):
    """piecewise-constant per block + noise: what a coarse quantizer leaves behind"""
    img = np.zeros((h, w), np.int64)
    base = rng.integers(0, 1 << bd, (blocks.shape[0] // 2 + 1, blocks.shape[1] // 2 + 1))
    yy, xx = np.mgrid[0:h, 0:w]
    img = base[(yy << ydec) // 8, (xx << xdec) // 8] // 2 + (1 << (bd - 2))
    step = rng.integers(-6 << (bd - 8), (6 << (bd - 8)) + 1, base.shape)
    img = img * 0 + (1 << (bd - 1)) + step[(yy << ydec) // 8, (xx << xdec) // 8] * 2
    img = img + rng.integers(-noise, noise + 1, (h, w))
    return np.clip(img, 0, (1 << bd) - 1)


CASES = [
    # (name, luma w, h, crop_w, crop_h, bd, (xdec, ydec), levels, deltas, block_deltas, noise)
    ("420_8", 96, 64, 96, 64, 8, (1, 1), [20, 14, 12, 9], False, False, 1),
    ("420_8_crop", 96, 64, 90, 58, 8, (1, 1), [34, 40, 22, 30], False, False, 2),
    ("420_10", 64, 64, 64, 64, 10, (1, 1), [12, 12, 8, 8], False, False, 2),
    ("420_12_deltas", 64, 48, 64, 48, 12, (1, 1), [25, 18, 16, 16], True, False, 6),
    ("444_8_blockdeltas", 64, 48, 64, 48, 8, (0, 0), [9, 11, 10, 13], True, True, 1),
    ("422_10", 64, 32, 64, 32, 10, (1, 0), [40, 30, 20, 35], False, False, 3),
    ("420_8_strong", 64, 64, 64, 64, 8, (1, 1), [63, 63, 63, 63], False, False, 3),
    ("420_8_weak", 64, 64, 64, 64, 8, (1, 1), [1, 2, 1, 3], False, False, 1),
    ("420_8_vonly", 64, 32, 64, 32, 8, (1, 1), [17, 0, 0, 5], False, False, 1),
]


def main():
    out = {}
    for ci, (name, w, h, cw, ch, bd, (xdec, ydec), levels, deltas, bdel, noise) in enumerate(CASES):
        rng = np.random.default_rng(1000 + ci)
        blocks = D.random_blocks(rng, w // 4, h // 4, xdec, ydec, deltas=bdel)
        state = D.make_state(levels, rng, deltas, bdel)
        out[name + "_blocks"] = blocks
        out[name + "_state"] = state
        out[name + "_meta"] = np.array([w, h, cw, ch, bd, xdec, ydec])
        for pli in range(3):
            pw, ph = (w, h) if pli == 0 else (w >> xdec, h >> ydec)
            xd, yd = (0, 0) if pli == 0 else (xdec, ydec)
            rec = blocky_image(rng, pw, ph, bd, blocks, xd, yd, noise)
            src = np.clip(rec + rng.integers(-3 << (bd - 8), (3 << (bd - 8)) + 1, rec.shape), 0, (1 << bd) - 1)
            want = D.spec_deblock_plane(rec.copy(), pli, xd, yd, blocks, state, cw, ch, bd)
            tv, th = D.brute_force_tallies(rec, src, pli, xd, yd, blocks, cw, ch, bd)
            dt = np.uint8 if bd == 8 else np.uint16
            out["%s_p%d_rec" % (name, pli)] = rec.astype(dt)
            out["%s_p%d_src" % (name, pli)] = src.astype(dt)
            out["%s_p%d_out" % (name, pli)] = want.astype(dt)
            out["%s_p%d_tv" % (name, pli)] = tv
            out["%s_p%d_th" % (name, pli)] = th
            print(name, pli, "changed px:", int((want != rec).sum()), "of", rec.size)
    np.savez_compressed(os.path.join(HERE, "deblock_golden.npz"), **out)


if __name__ == "__main__":
    main()
