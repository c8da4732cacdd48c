#!/usr/bin/env python3
"""tests/golden/deblock_ref.npz: deblocking-filter frames and level-search tallies
computed by the REFERENCE'S OWN SOURCE TEXT (src/deblock.rs:23-1668: deblock_plane,
filter_v_edge / filter_h_edge, deblock_size, deblock_level, deblock_adjusted_level, the
size-4/6/8/14 filters and their sse_* twins, sse_plane, sse_optimize), transpiled by
tools/rustlite and executed here.  Same key layout as deblock_golden.npz (+ <case>_levels
= what sse_optimize picked), so the tests that read that file also run on this one.

Hand-stated: the per-4x4 `Block` records and `TileBlocks` container the filter reads
(plain data here, built from the same 8-byte wire format the C ABI uses), and the
random AV1 partition trees that fill them.

The reference indexes ref_deltas with ref_frames[0].to_index(), which panics for intra
blocks: frame-level delta cases therefore contain inter blocks only (the reference never
enables them otherwise, src/encoder.rs:539).

Run in the build container:  python tests/golden/gen_deblock_ref.py
"""
import numpy as np

import reflib as L
from reflib import R

DEBLOCK_BLOCK = np.dtype([("tx_log2", "u1"), ("uvtx_log2", "u1"), ("n4_log2", "u1"), ("flags", "u1"),
                          ("deltas", "i1", (4,))])
DEBLOCK_STATE = np.dtype([("levels", "u1", (4,)), ("sharpness", "u1"), ("deltas_enabled", "u1"),
                          ("block_deltas_enabled", "u1"), ("block_delta_shift", "u1"),
                          ("block_delta_multi", "u1"), ("ref_deltas", "i1", (8,)),
                          ("mode_deltas", "i1", (2,)), ("reserved", "u1", (5,))])


def lg(v):
    return int(v).bit_length() - 1


def partition_blocks(rng, mi_cols, mi_rows, xdec, ydec, p_skip, p_intra, deltas, valid=lambda w, h: True):
    """random AV1 partition trees (the ten partition types) over 64x64 superblocks; every
    leaf gets a transform size that divides it"""
    out = np.zeros((mi_rows, mi_cols), DEBLOCK_BLOCK)

    def leaf(x, y, w, h):
        if x >= mi_cols or y >= mi_rows:
            return
        txw = 1 << rng.integers(max(0, lg(w) - 2), min(lg(w), 4) + 1)
        txh = 1 << rng.integers(max(0, lg(h) - 2), min(lg(h), 4) + 1)
        while txw > 4 * txh:
            txw //= 2
        while txh > 4 * txw:
            txh //= 2
        uw = min(max((w * 4) >> xdec, 4), 32) // 4
        uh = min(max((h * 4) >> ydec, 4), 32) // 4
        intra = rng.random() < p_intra
        flags = (1 if rng.random() < p_skip else 0) | (2 if intra else 0) | \
                (0 if intra or rng.random() < 0.5 else 4) | ((0 if intra else int(rng.integers(0, 7))) << 3)
        b = out[y:y + h, x:x + w]
        b["tx_log2"], b["uvtx_log2"] = lg(txw) | (lg(txh) << 3), lg(uw) | (lg(uh) << 3)
        b["n4_log2"], b["flags"] = lg(w) | (lg(h) << 3), flags
        if deltas:
            b["deltas"] = rng.integers(-3, 4, 4)

    def node(x, y, n):
        if x >= mi_cols or y >= mi_rows:
            return
        h2, q = n // 2, n // 4
        types = ["NONE"]
        if n >= 2:
            types += ["SPLIT"] * (6 if n >= 8 else 3) + ["HORZ", "VERT"]
        if n >= 4:
            types += ["HORZ_A", "HORZ_B", "VERT_A", "VERT_B", "HORZ_4", "VERT_4"]
        if n == 16 and rng.random() < 0.9:
            types = ["SPLIT"]
        LEAVES = {"NONE": [(n, n)], "SPLIT": [], "HORZ": [(n, h2)], "VERT": [(h2, n)],
                  "HORZ_A": [(h2, h2), (n, h2)], "HORZ_B": [(h2, h2), (n, h2)], "VERT_A": [(h2, h2), (h2, n)],
                  "VERT_B": [(h2, h2), (h2, n)], "HORZ_4": [(n, q)], "VERT_4": [(q, n)]}
        # block shapes the chroma subsampling does not allow are not offered (the reference panics on them)
        types = [t for t in types if all(valid(w_, h_) for (w_, h_) in LEAVES[t])]
        t = types[int(rng.integers(0, len(types)))]
        if t == "NONE":
            leaf(x, y, n, n)
        elif t == "SPLIT":
            for (dx, dy) in ((0, 0), (h2, 0), (0, h2), (h2, h2)):
                node(x + dx, y + dy, h2)
        elif t == "HORZ":
            leaf(x, y, n, h2), leaf(x, y + h2, n, h2)
        elif t == "VERT":
            leaf(x, y, h2, n), leaf(x + h2, y, h2, n)
        elif t == "HORZ_A":
            leaf(x, y, h2, h2), leaf(x + h2, y, h2, h2), leaf(x, y + h2, n, h2)
        elif t == "HORZ_B":
            leaf(x, y, n, h2), leaf(x, y + h2, h2, h2), leaf(x + h2, y + h2, h2, h2)
        elif t == "VERT_A":
            leaf(x, y, h2, h2), leaf(x, y + h2, h2, h2), leaf(x + h2, y, h2, n)
        elif t == "VERT_B":
            leaf(x, y, h2, n), leaf(x + h2, y, h2, h2), leaf(x + h2, y + h2, h2, h2)
        elif t == "HORZ_4":
            for k in range(4):
                leaf(x, y + k * q, n, q)
        else:
            for k in range(4):
                leaf(x + k * q, y, q, n)

    for sy in range(0, mi_rows, 16):
        for sx in range(0, mi_cols, 16):
            node(sx, sy, 16)
    return out


def blocky_image(rng, w, h, bd, blocks, xdec, ydec, noise):
    base = rng.integers(0, 1 << bd, (blocks.shape[0] // 2 + 1, blocks.shape[1] // 2 + 1))
    yy, xx = np.mgrid[0:h, 0:w]
    step = rng.integers(-6 << (bd - 8), (6 << (bd - 8)) + 1, base.shape)
    img = (1 << (bd - 1)) + step[(yy << ydec) // 8, (xx << xdec) // 8] * 2
    return np.clip(img + rng.integers(-noise, noise + 1, (h, w)), 0, (1 << bd) - 1)


class Obj:
    def __init__(self, **kw):
        self.__dict__.update(kw)


class TileBlocks:
    def __init__(self, grid):
        self.grid = grid      # list of rows of Block objects

    def cols(self):
        return len(self.grid[0])

    def rows(self):
        return len(self.grid)

    def __getitem__(self, i):
        if isinstance(i, int):
            return self.grid[i]
        return self.grid[i._0.y][i._0.x]


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
    ("420_10_big", 192, 128, 188, 122, 10, (1, 1), [28, 22, 18, 14], False, False, 4),
]


def main():
    c = L.crate("deblock.rs", "partition.rs", "predict.rs", "transform/mod.rs", "context/block_unit.rs")
    deblock_plane, sse_plane = c.get("deblock_plane"), c.get("sse_plane")
    sse_optimize = c.get("sse_optimize")
    DState = L.struct(c, "DeblockState")
    txs = {}
    for v in c.enums["TxSize"].variants:
        e = L.enum(c, "TxSize", v[0])
        wh = [int(t) for t in v[0][3:].split("X")]
        txs[(wh[0], wh[1])] = e
    bss = {}
    for v in c.enums["BlockSize"].variants:
        if v[0].startswith("BLOCK_") and v[0] != "BLOCK_INVALID":
            wh = [int(t) for t in v[0][6:].split("X")]
            bss[(wh[0], wh[1])] = L.enum(c, "BlockSize", v[0])
    REFS = [L.enum(c, "RefType", n) for n in ("INTRA_FRAME", "LAST_FRAME", "LAST2_FRAME", "LAST3_FRAME",
                                               "GOLDEN_FRAME", "BWDREF_FRAME", "ALTREF2_FRAME", "ALTREF_FRAME")]
    NONE_FRAME = L.enum(c, "RefType", "NONE_FRAME")
    DC, GLOBALMV, NEWMV = (L.enum(c, "PredictionMode", n) for n in ("DC_PRED", "GLOBALMV", "NEWMV"))
    out = {}
    for ci, (name, w, h, cw, ch, bd, (xdec, ydec), levels, deltas, bdel, noise) in enumerate(CASES):
        rng = np.random.default_rng(2000 + ci)
        g = L.pixel_type(bd)
        dt = L.np_dtype(bd)
        sub = c.get("subsampled_size", owner="BlockSize")

        def valid(w4, h4, xdec=xdec, ydec=ydec):
            return sub({}, bss[(4 * w4, 4 * h4)], xdec, ydec).var == "Ok"
        wire = partition_blocks(rng, w // 4, h // 4, xdec, ydec, 0.5, 0.0 if deltas else 0.2, bdel, valid)
        st = np.zeros(1, DEBLOCK_STATE)
        st["levels"] = levels
        if deltas:
            st["deltas_enabled"] = 1
            st["ref_deltas"] = [1, 0, 0, 0, -1, 0, -1, -1]
            st["mode_deltas"] = [0, 2]
        if bdel:
            st["block_deltas_enabled"], st["block_delta_shift"] = 1, 1
            st["block_delta_multi"] = int(rng.integers(0, 2))
        s = st[0]
        state = DState(levels=R.array(*[int(v) for v in s["levels"]]), sharpness=0,
                       deltas_enabled=bool(s["deltas_enabled"]), delta_updates_enabled=False,
                       ref_deltas=R.array(*[int(v) for v in s["ref_deltas"]]),
                       mode_deltas=R.array(*[int(v) for v in s["mode_deltas"]]),
                       block_deltas_enabled=bool(s["block_deltas_enabled"]),
                       block_delta_shift=int(s["block_delta_shift"]),
                       block_delta_multi=bool(s["block_delta_multi"]))
        grid = []
        for row in wire:
            r = []
            for b in row:
                f = int(b["flags"])
                intra = bool(f & 2)
                ref = REFS[0] if intra else REFS[1 + ((f >> 3) & 7)]      # wire carries to_index()
                n4w, n4h = 1 << (int(b["n4_log2"]) & 7), 1 << (int(b["n4_log2"]) >> 3)
                txw, txh = 4 << (int(b["tx_log2"]) & 7), 4 << (int(b["tx_log2"]) >> 3)
                r.append(Obj(mode=DC if intra else (NEWMV if f & 4 else GLOBALMV), skip=bool(f & 1),
                             ref_frames=R.array(ref, NONE_FRAME), bsize=bss[(4 * n4w, 4 * n4h)],
                             n4_w=n4w, n4_h=n4h, txsize=txs[(txw, txh)],
                             deblock_deltas=R.array(*[int(v) for v in b["deltas"]])))
            grid.append(r)
        tb = TileBlocks(grid)
        out[name + "_blocks"], out[name + "_state"] = wire, st
        out[name + "_meta"] = np.array([w, h, cw, ch, bd, xdec, ydec])
        recs, srcs = [], []
        for pli in range(3):
            pw, ph = (w, h) if pli == 0 else (w >> xdec, h >> ydec)
            xd, yd = (0, 0) if pli == 0 else (xdec, ydec)
            rec = blocky_image(rng, pw, ph, bd, wire, xd, yd, noise)
            src = np.clip(rec + rng.integers(-3 << (bd - 8), (3 << (bd - 8)) + 1, rec.shape), 0, (1 << bd) - 1)
            prec = L.plane_from_padded(np.pad(rec, 16, mode="edge"), bd, 16, 16, xd, yd)
            psrc = L.plane_from_padded(np.pad(src, 16, mode="edge"), bd, 16, 16, xd, yd)
            recs.append(prec)
            srcs.append(psrc)
            # level search on the unfiltered reconstruction
            tv, th = R.repeat(0, 65), R.repeat(0, 65)
            sse_plane(g, prec.as_region(), psrc.as_region(), tv, th, pli, tb, cw, ch, bd)
            out["%s_p%d_tv" % (name, pli)] = np.cumsum(np.array(tv.tolist(), np.int64))[:64]
            out["%s_p%d_th" % (name, pli)] = np.cumsum(np.array(th.tolist(), np.int64))[:64]
            out["%s_p%d_rec" % (name, pli)] = rec.astype(dt)
            out["%s_p%d_src" % (name, pli)] = src.astype(dt)
        lv = sse_optimize(g, Obj(planes=R.RSlice([p.as_region() for p in recs])),
                          Obj(planes=R.RSlice([p.as_region() for p in srcs])), tb, cw, ch, bd, False)
        out[name + "_levels"] = np.array(lv.tolist(), np.uint8)
        for pli in range(3):
            deblock_plane(g, state, recs[pli].as_region(), pli, tb, cw, ch, bd)
            res = L.plane_to_array(recs[pli], dt)
            out["%s_p%d_out" % (name, pli)] = res
            print(name, pli, "changed px:", int((res != out["%s_p%d_rec" % (name, pli)]).sum()), flush=True)
    L.save("deblock_ref.npz", out)


if __name__ == "__main__":
    main()
