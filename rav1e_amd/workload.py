"""Synthetic RDO-candidate workloads and tile sharding (8e).

`speed6_ladder` builds, for a frame of the given size, the candidate lists the
bench and the scale tests use: for every block of the speed-6 top-down size
ladder 64x64 / 32x32 / 16x16 / 8x8 (src/api/config/speedsettings.rs:119-191
partition range, src/encoder.rs:2936-2978) that lies inside the frame, K
inter candidates whose motion vectors are uniform in +-`mv_range` full pels
with uniform 1/16-pel fractions -- the shape of the candidate sets that
src/me.rs:1445-1454 and src/rdo.rs:1281-1357 evaluate one call at a time.

`tile_rects` mirrors the reference's uniform tiling (TilingInfo::from_target_tiles,
src/tiling/tiler.rs:56-150, and the col/row split search of
src/encoder.rs:248-277): tiles are whole-superblock rectangles; rank r owns
tile r.
"""
import numpy as np

RDO_CAND = np.dtype([("ox", "<i2"), ("oy", "<i2"), ("rx", "<i2"), ("ry", "<i2"),
                     ("col_frac", "u1"), ("row_frac", "u1"), ("mode_x", "u1"), ("mode_y", "u1"),
                     ("tx_type", "u1"), ("reserved", "u1", (3,))])

LADDER = (64, 32, 16, 8)
SB = 64


def tile_split(n_tiles, frame_w, frame_h):
    """(cols, rows) of the uniform tile grid for a power-of-two tile count:
    alternate col/row splits, always halving the longer tile side
    (src/encoder.rs:248-277).  8 tiles on 3840x2160 -> 4 cols x 2 rows."""
    assert n_tiles >= 1 and (n_tiles & (n_tiles - 1)) == 0, "power-of-two tile counts"
    cols = rows = 1
    while cols * rows < n_tiles:
        if frame_w / cols >= frame_h / rows:
            cols *= 2
        else:
            rows *= 2
    return cols, rows


def tile_rects(n_tiles, frame_w, frame_h):
    """[(x0, y0, x1, y1)] in pixels, superblock-aligned, covering the frame."""
    cols, rows = tile_split(n_tiles, frame_w, frame_h)
    sb_w, sb_h = -(-frame_w // SB), -(-frame_h // SB)
    tw, th = -(-sb_w // cols), -(-sb_h // rows)   # tile size in SBs (uniform spacing)
    # like TilingInfo::from_target_tiles (tiler.rs:86-104): once the tile size is rounded up to
    # whole superblocks the grid is recomputed from it -- a small frame yields fewer tiles
    cols, rows = -(-sb_w // tw), -(-sb_h // th)
    if cols * rows != n_tiles:
        raise ValueError("a %dx%d frame splits into %d x %d superblock-aligned tiles, not %d"
                         % (frame_w, frame_h, cols, rows, n_tiles))
    rects = []
    for r in range(rows):
        for c in range(cols):
            x0, y0 = c * tw * SB, r * th * SB
            x1, y1 = min((c + 1) * tw * SB, frame_w), min((r + 1) * th * SB, frame_h)
            rects.append((x0, y0, x1, y1))
    return rects


# the seven transform types rav1e's RDO searches (RAV1E_TX_TYPES, src/transform/mod.rs:28-44):
# DCT_DCT, ADST_DCT, DCT_ADST, ADST_ADST, IDTX, V_DCT, H_DCT -- by TxType discriminant
RDO_TX_TYPES = (0, 1, 2, 3, 9, 10, 11)


def rdo_tx_types_for(size):
    """the subset of RDO_TX_TYPES that valid_av1_transform (src/transform/mod.rs:405-417) admits
    for a square transform of this size: all seven up to 16x16, DCT_DCT / IDTX at 32, DCT_DCT at 64"""
    return RDO_TX_TYPES if size <= 16 else ((0, 9) if size == 32 else (0,))


def speed6_ladder(frame_w, frame_h, k, seed=3, mv_range=32, rect=None, tx_type=0, sizes=LADDER,
                  mix_filters=False, mix_tx_types=False):
    """-> {size: structured array of RDO_CAND}; blocks restricted to `rect`
    (a tile) when given.  Deterministic per (seed, size): a tile's list is the
    subset of the whole-frame list, so sharded and unsharded runs evaluate the
    same candidates.  sizes: the block ladder (chroma planes of 4:2:0 use 32/16/8/4);
    mix_filters: one of the 9 8-tap filter pairs per candidate (REGULAR / SMOOTH / SHARP in
    each direction) instead of REGULAR/REGULAR; mix_tx_types: one of rdo_tx_types_for(size) per
    candidate instead of `tx_type`."""
    out = {}
    for lvl, s in enumerate(sizes):
        rng = np.random.default_rng([seed, lvl])
        nx, ny = frame_w // s, frame_h // s
        n = nx * ny * k
        c = np.zeros(n, RDO_CAND)
        bx = np.repeat(np.tile(np.arange(nx, dtype=np.int32) * s, ny), k)
        by = np.repeat(np.repeat(np.arange(ny, dtype=np.int32) * s, nx), k)
        c["ox"], c["oy"] = bx, by
        c["rx"] = bx + rng.integers(-mv_range, mv_range + 1, n)
        c["ry"] = by + rng.integers(-mv_range, mv_range + 1, n)
        c["col_frac"] = rng.integers(0, 16, n)
        c["row_frac"] = rng.integers(0, 16, n)
        c["tx_type"] = tx_type
        # drawn after everything else, so the default lists do not depend on these options
        if mix_filters:
            c["mode_x"] = rng.integers(0, 3, n)
            c["mode_y"] = rng.integers(0, 3, n)
        if mix_tx_types:
            c["tx_type"] = np.array(rdo_tx_types_for(s), np.uint8)[rng.integers(0, len(rdo_tx_types_for(s)), n)]
        if rect is not None:
            x0, y0, x1, y1 = rect
            keep = (bx >= x0) & (bx < x1) & (by >= y0) & (by < y1)
            c = c[keep]
        out[s] = c
    return out


def algorithmic_bytes_per_cand(w, h, bpp, n_dist_out=2):
    """compulsory traffic of one fused candidate:
    reference window + source block + coefficients + distortion words."""
    cb = 2 if bpp == 1 else 4
    return bpp * ((w + 7) * (h + 7) + w * h) + cb * w * h + 4 * n_dist_out


def plane_layout(width, height, bit_depth, xpad=88, ypad=88):
    """v_frame 0.3.9 PlaneConfig::new: 64-byte aligned xorigin / stride."""
    bpp = 1 if bit_depth == 8 else 2
    al = 64 // bpp
    xorigin = (xpad + al - 1) // al * al
    stride = (xorigin + width + xpad + al - 1) // al * al
    return {"bpp": bpp, "xorigin": xorigin, "yorigin": ypad, "stride": stride,
            "alloc_height": ypad + height + ypad}


def random_plane_array(width, height, bit_depth, seed, xpad=88, ypad=88):
    """Whole padded allocation filled uniformly over the bit depth's range
    (the reference's benches fill planes with random::<u8>(), benches/dist.rs:73-81)."""
    lay = plane_layout(width, height, bit_depth, xpad, ypad)
    rng = np.random.default_rng(seed)
    dt = np.uint8 if lay["bpp"] == 1 else np.uint16
    return rng.integers(0, 1 << bit_depth, size=(lay["alloc_height"], lay["stride"]), dtype=dt)


def sustain_clocks(one_pass, ms=150.0):
    """Untimed passes of `one_pass` for `ms` milliseconds before a benchmark row is timed: the GPU is
    then at the clocks a continuously running encoder holds, not at what the host-side preparation of
    the row (candidate lists, oracle legs) let them fall to -- short rows read 10-20 % slow otherwise
    (bench.py::sustain has the measurement)."""
    import time
    import torch
    if ms <= 0:
        return
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < ms:
        for _ in range(4):
            one_pass()
        torch.cuda.synchronize()
