"""Shared by tests/golden/gen_deblock_golden.py and the deblock tests: random block
structures in the wire format of r1o_deblock_block / R1DeblockBlock, and an independent
model of the AV1 loop filter in the SPECIFICATION's formulation (AV1 spec 7.14: boolean
limit / blimit / thresh masks, signed narrow filter, generic tap-window wide filter, all
vertical edges of a plane before all horizontal ones) -- not a transcription of
src/deblock.rs, which works with per-line level thresholds, unsigned arithmetic, explicit
coefficient tables and an interleaved edge order."""
import numpy as np

DEBLOCK_BLOCK = np.dtype([("tx_log2", "u1"), ("uvtx_log2", "u1"), ("n4_log2", "u1"), ("flags", "u1"),
                          ("deltas", "i1", (4,))])
assert DEBLOCK_BLOCK.itemsize == 8
DEBLOCK_STATE = np.dtype([("levels", "u1", (4,)), ("sharpness", "u1"), ("deltas_enabled", "u1"),
                          ("block_deltas_enabled", "u1"), ("block_delta_shift", "u1"),
                          ("block_delta_multi", "u1"), ("ref_deltas", "i1", (8,)),
                          ("mode_deltas", "i1", (2,)), ("reserved", "u1", (5,))])
assert DEBLOCK_STATE.itemsize == 24


def _log2(v):
    return int(v).bit_length() - 1


def random_blocks(rng, mi_cols, mi_rows, xdec, ydec, p_skip=0.5, p_intra=0.2, deltas=False):
    """Random AV1 partition trees over the frame's 64x64 superblocks, each leaf with a random
    transform size dividing it.  -> DEBLOCK_BLOCK array (mi_rows, mi_cols)."""
    out = np.zeros((mi_rows, mi_cols), DEBLOCK_BLOCK)

    def leaf(x, y, w, h):       # in 4x4 units
        if x >= mi_cols or y >= mi_rows:
            return
        txw = 1 << rng.integers(max(0, _log2(w) - 2), min(_log2(w), 4) + 1)
        txh = 1 << rng.integers(max(0, _log2(h) - 2), min(_log2(h), 4) + 1)
        while txw > 4 * txh:
            txw //= 2
        while txh > 4 * txw:
            txh //= 2
        uw = min(max((w * 4) >> xdec, 4), 32) // 4      # largest_chroma_tx_size, chroma 4x4 units
        uh = min(max((h * 4) >> ydec, 4), 32) // 4
        intra = rng.random() < p_intra
        flags = (1 if rng.random() < p_skip else 0) | (2 if intra else 0) | \
                (0 if intra or rng.random() < 0.5 else 4) | ((0 if intra else int(rng.integers(1, 8))) << 3)
        b = out[y:y + h, x:x + w]
        b["tx_log2"] = _log2(txw) | (_log2(txh) << 3)
        b["uvtx_log2"] = _log2(uw) | (_log2(uh) << 3)
        b["n4_log2"] = _log2(w) | (_log2(h) << 3)
        b["flags"] = flags
        if deltas:
            b["deltas"] = rng.integers(-3, 4, 4)

    def node(x, y, n):
        """one square of the AV1 partition tree (n in 4x4 units): the ten partition types of
        the specification -- leaves are never split again, so the blocks sharing a chroma
        4x4 always agree on their chroma transform size (what the in-place two-pass filter,
        like any decoder, relies on)."""
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
        t = types[int(rng.integers(0, len(types)))]
        if t == "NONE":
            leaf(x, y, n, n)
        elif t == "SPLIT":
            for (dx, dy) in ((0, 0), (h2, 0), (0, h2), (h2, h2)):
                node(x + dx, y + dy, h2)
        elif t == "HORZ":
            leaf(x, y, n, h2); leaf(x, y + h2, n, h2)
        elif t == "VERT":
            leaf(x, y, h2, n); leaf(x + h2, y, h2, n)
        elif t == "HORZ_A":
            leaf(x, y, h2, h2); leaf(x + h2, y, h2, h2); leaf(x, y + h2, n, h2)
        elif t == "HORZ_B":
            leaf(x, y, n, h2); leaf(x, y + h2, h2, h2); leaf(x + h2, y + h2, h2, h2)
        elif t == "VERT_A":
            leaf(x, y, h2, h2); leaf(x, y + h2, h2, h2); leaf(x + h2, y, h2, n)
        elif t == "VERT_B":
            leaf(x, y, h2, n); leaf(x + h2, y, h2, h2); leaf(x + h2, y + h2, h2, h2)
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


def make_state(levels, rng=None, deltas=False, block_deltas=False):
    s = np.zeros(1, DEBLOCK_STATE)
    s["levels"] = levels
    if deltas:
        s["deltas_enabled"] = 1
        s["ref_deltas"] = [1, 0, 0, 0, -1, 0, -1, -1]
        s["mode_deltas"] = [0, 2]
    if block_deltas:
        s["block_deltas_enabled"] = 1
        s["block_delta_shift"] = 1
        s["block_delta_multi"] = int(rng.integers(0, 2)) if rng is not None else 1
    return s


# ---------------------------------------------------------------- specification model
def _field(b, name):
    return int(b[name])


def _tx_mi(b, pli, vertical):
    v = _field(b, "tx_log2" if pli == 0 else "uvtx_log2")
    return 1 << ((v & 7) if vertical else ((v >> 3) & 7))


def _level(state, b, pli, vertical):
    s = state[0]
    idx = (0 if vertical else 1) if pli == 0 else pli + 1
    lvl = int(s["levels"][idx])
    if s["block_deltas_enabled"]:
        d = int(b["deltas"][idx if s["block_delta_multi"] else 0]) << int(s["block_delta_shift"])
        lvl = min(max(lvl + d, 0), 63)
    if s["deltas_enabled"]:
        intra = (int(b["flags"]) >> 1) & 1
        sh = lvl >> 5
        lvl = lvl + (int(s["ref_deltas"][(int(b["flags"]) >> 3) & 7]) << sh)
        if not intra:
            lvl += int(s["mode_deltas"][(int(b["flags"]) >> 2) & 1]) << sh
        lvl = min(max(lvl, 0), 63)
    return lvl


def _narrow(F, hev, bd):
    """7.14.6.3; F: dict index -> sample (negative = p side)"""
    half = 0x80 << (bd - 8)
    lo, hi = -(1 << (bd - 1)), (1 << (bd - 1)) - 1
    c4 = lambda v: min(max(v, lo), hi)
    ps1, ps0, qs0, qs1 = F[-2] - half, F[-1] - half, F[0] - half, F[1] - half
    f = c4(ps1 - qs1) if hev else 0
    f = c4(f + 3 * (qs0 - ps0))
    f1 = c4(f + 4) >> 3
    f2 = c4(f + 3) >> 3
    out = dict(F)
    out[0] = c4(qs0 - f1) + half
    out[-1] = c4(ps0 + f2) + half
    if not hev:
        f = (f1 + 1) >> 1
        out[1] = c4(qs1 - f) + half
        out[-2] = c4(ps1 + f) + half
    return out


def _wide(F, log2size, pli):
    """7.14.6.4: every output is a (2n+1)-tap window with the ends clamped"""
    n = 6 if log2size == 4 else (3 if pli == 0 else 2)
    n2 = 0 if (log2size == 3 and pli == 0) else 1
    out = dict(F)
    for i in range(-n, n):
        t = 0
        for j in range(-n, n + 1):
            p = min(max(i + j, -(n + 1)), n)
            t += F[p] * (2 if abs(j) <= n2 else 1)
        out[i] = (t + (1 << (log2size - 1))) >> log2size
    return out


def _filter_line(F, size, lvl, bd, pli):
    """sample filtering process 7.14.6 with sharpness 0; F has taps -size/2 .. size/2-1
    (size 14 stands for the specification's filter length 16)."""
    sh = bd - 8
    limit = max(1, lvl)
    blimit = 2 * (lvl + 2) + limit
    thresh = lvl >> 4
    hev = abs(F[-2] - F[-1]) > (thresh << sh) or abs(F[1] - F[0]) > (thresh << sh)
    nlim = {4: 2, 6: 3, 8: 4, 14: 4}[size]
    mask = abs(F[-1] - F[0]) * 2 + abs(F[-2] - F[1]) // 2 <= (blimit << sh)
    for i in range(1, nlim):
        mask = mask and abs(F[-i - 1] - F[-i]) <= (limit << sh) and abs(F[i] - F[i - 1]) <= (limit << sh)
    if not mask:
        return F
    one = 1 << sh
    flat = size != 4 and all(abs(F[-i - 1] - F[-1]) <= one and abs(F[i] - F[0]) <= one
                             for i in range(1, 3 if size == 6 else 4))
    if not flat:
        return _narrow(F, hev, bd)
    flat2 = size == 14 and all(abs(F[-i - 1] - F[-1]) <= one and abs(F[i] - F[0]) <= one for i in range(4, 7))
    return _wide(F, 4 if flat2 else 3, pli)


def edges_of(blocks, pli, xdec, ydec, crop_w, crop_h, vertical, size_vertical=None):
    """(bx, by, size, block, prev) of every edge the filter visits, plane extent as deblock_plane"""
    if size_vertical is None:
        size_vertical = vertical
    mi_rows, mi_cols = blocks.shape
    cols = ((min(mi_cols, (crop_w + 3) >> 2) + ((1 << xdec) >> 1)) >> xdec) << xdec
    rows = ((min(mi_rows, (crop_h + 3) >> 2) + ((1 << ydec) >> 1)) >> ydec) << ydec
    sx, sy = 1 << xdec, 1 << ydec
    for by in range(0 if vertical else sy, rows, sy):
        for bx in range(sx if vertical else 0, cols, sx):
            b = blocks[by, bx]
            if vertical:
                if ((bx >> xdec) & (_tx_mi(b, pli, True) - 1)) != 0:
                    continue
                prev = blocks[by | ydec, (bx | xdec) - sx]
                block_edge = (bx & ((1 << (int(b["n4_log2"]) & 7)) - 1)) == 0
            else:
                if ((by >> ydec) & (_tx_mi(b, pli, False) - 1)) != 0:
                    continue
                prev = blocks[(by | ydec) - sy, bx | xdec]
                block_edge = (by & ((1 << ((int(b["n4_log2"]) >> 3) & 7)) - 1)) == 0
            skip, pskip = int(b["flags"]) & 1, int(prev["flags"]) & 1
            intra, pintra = (int(b["flags"]) >> 1) & 1, (int(prev["flags"]) >> 1) & 1
            if not (block_edge or not skip or not pskip or intra or pintra):
                continue
            n = min(_tx_mi(b, pli, size_vertical), _tx_mi(prev, pli, size_vertical)) * 4
            size = min(14 if pli == 0 else 6, n)
            yield bx, by, size, b, prev


def spec_deblock_plane(img, pli, xdec, ydec, blocks, state, crop_w, crop_h, bd):
    """img: 2-D int array (visible plane), filtered in place: pass 0 all vertical edges,
    pass 1 all horizontal edges (AV1 spec 7.14.2)."""
    s = state[0]
    if (pli == 0 and s["levels"][0] == 0 and s["levels"][1] == 0) or (pli > 0 and s["levels"][pli + 1] == 0):
        return img
    for vertical in (True, False):
        for bx, by, size, b, prev in edges_of(blocks, pli, xdec, ydec, crop_w, crop_h, vertical):
            lvl = _level(state, b, pli, vertical)
            if lvl == 0:
                lvl = _level(state, prev, pli, vertical)
            if lvl == 0:
                continue
            px, py, h = (bx >> xdec) * 4, (by >> ydec) * 4, size // 2
            for i in range(4):
                if vertical:
                    F = {k: int(img[py + i, px + k]) for k in range(-h, h)}
                else:
                    F = {k: int(img[py + k, px + i]) for k in range(-h, h)}
                G = _filter_line(F, size, lvl, bd, pli)
                for k in range(-h, h):
                    if vertical:
                        img[py + i, px + k] = G[k]
                    else:
                        img[py + k, px + i] = G[k]
    return img


def brute_force_tallies(rec, src, pli, xdec, ydec, blocks, crop_w, crop_h, bd):
    """What sse_plane's tallies mean: after the prefix sum, entry L is the SSE (over the
    pixels an edge of that size may change) of filtering every edge, each on the UNfiltered
    reconstruction, at level L.  -> (v, h) int64[64] AFTER the prefix sum."""
    # The horizontal-edge sizes come from transform WIDTHS (below), so near the top / bottom
    # of the frame a line may reach up to 7 rows outside it: the reference then reads the
    # planes' padding rows.  Here: edge-replicated, like oracle_lib.plane_from_image.
    PAD = 8
    rec = np.pad(rec, ((PAD, PAD), (0, 0)), mode="edge")
    src = np.pad(src, ((PAD, PAD), (0, 0)), mode="edge")
    out = []
    for vertical in (True, False):
        t = np.zeros(64, np.int64)
        # sse_h_edge hands `true` to deblock_size (src/deblock.rs:1258): sizes from tx WIDTHS
        for bx, by, size, b, prev in edges_of(blocks, pli, xdec, ydec, crop_w, crop_h, vertical, True):
            px, py, h = (bx >> xdec) * 4, (by >> ydec) * 4 + PAD, size // 2
            lo, hi = (-h, h) if size == 4 else (-h + 1, h - 1)
            for i in range(4):
                if vertical:
                    F = {k: int(rec[py + i, px + k]) for k in range(-h, h)}
                    S = {k: int(src[py + i, px + k]) for k in range(-h, h)}
                else:
                    F = {k: int(rec[py + k, px + i]) for k in range(-h, h)}
                    S = {k: int(src[py + k, px + i]) for k in range(-h, h)}
                for lvl in range(64):
                    G = F if lvl == 0 else _filter_line(F, size, lvl, bd, pli)
                    t[lvl] += sum((G[k] - S[k]) ** 2 for k in range(lo, hi))
        out.append(t)
    return out
