"""Host-side glue of the RDO path: the small pieces of encoder logic that sit between the
reference's callers and the batch entry points -- descriptor arithmetic, no pixels.  Each function
restates a reference function (file:line); tests/golden/rdo_glue_ref.npz pins them together with
the kernels by executing the reference's own text (tests/golden/gen_rdo_glue_ref.py).

  get_mv_params              src/predict.rs:284-297   (MV -> integer offset + 1/16-pel fractions)
  clip_visible_bsize         src/rdo.rs:228-251
  luma_ac_pads               src/predict.rs:644-688   (luma_ac's w_pad / h_pad and its luma offset)
  largest_chroma_tx_size     src/partition.rs:385-393
  chroma_dist_dims           src/rdo.rs:404-413       (chroma_w / chroma_h of compute_tx_distortion)
  scale_distortion           src/rdo.rs:613-615,674-695   (RawDistortion * DistortionScale -> mul_u64)
  compute_tx_distortion      src/rdo.rs:349-434       (composition; the SSE comes from a callable:
                                                       Context.dist_scaled_batch on the GPU)
  compute_distortion         src/rdo.rs:254-347       (composition: luma by tune, chroma sse_wxh per plane)
  rav1e_tx_types / tx_type_slots   src/transform/mod.rs:28-44, src/rdo.rs:1731-1736
                                                      (which slot of r1_rdo_txsearch_batch holds which TxType)
"""
from .types import BlockSize, TxSize

MI_SIZE_LOG2 = 2
DIST_SCALE_SHIFT = 14


def get_mv_params(mv_row, mv_col, po_x, po_y, xdec=0, ydec=0):
    """-> (row_frac, col_frac, x, y): put_8tap / prep_8tap read the block whose top-left pixel is
    (x, y) (the reference slices at (x - 3, y - 3) and steps 3 in: the filter's own margin)"""
    row_offset = mv_row >> (3 + ydec)
    col_offset = mv_col >> (3 + xdec)
    row_frac = (mv_row << (1 - ydec)) & 0xf
    col_frac = (mv_col << (1 - xdec)) & 0xf
    return row_frac, col_frac, po_x + col_offset, po_y + row_offset


def clip_visible_bsize(frame_w, frame_h, blk_w, blk_h, x, y):
    vw = blk_w if x + blk_w <= frame_w else (0 if x >= frame_w else frame_w - x)
    vh = blk_h if y + blk_h <= frame_h else (0 if y >= frame_h else frame_h - y)
    return vw, vh


def largest_chroma_tx_size(bsize, xdec, ydec):
    """TxSize of the chroma transform block of a luma partition: the subsampled block size, its
    largest rectangular transform, 64-point sides coded as 32 (av1_get_coded_tx_size)"""
    w, h = BlockSize(bsize).dims
    cw, ch = max(4, w >> xdec), max(4, h >> ydec)     # subsampled_size: sub-8x8 blocks share a 4x4
    return TxSize.by_dims(min(cw, 32), min(ch, 32))


def luma_ac_pads(bsize, luma_tx_size, bo_x, bo_y, w_in_b, h_in_b, xdec, ydec):
    """luma_ac: (w_pad, h_pad, luma_x, luma_y) -- pads in 4-pixel units of the chroma block, and the
    luma position pred_cfl_ac reads (sub-8x8 partitions read from the 8x8 that contains them)"""
    bw, bh = BlockSize(bsize).dims
    if (xdec and bw == 4) or (ydec and bh == 4):          # is_sub8x8 -> sub8x8_offset
        bo_x += -1 if (xdec and bw == 4) else 0
        bo_y += -1 if (ydec and bh == 4) else 0
    clipped_bw = min((w_in_b - bo_x) << MI_SIZE_LOG2, bw)
    clipped_bh = min((h_in_b - bo_y) << MI_SIZE_LOG2, bh)
    tw, th = TxSize(luma_tx_size).dims
    max_luma_w = ((clipped_bw + tw - 1) // tw) * tw if bw > 8 else bw
    max_luma_h = ((clipped_bh + th - 1) // th) * th if bh > 8 else bh
    return (bw - max_luma_w) >> (2 + xdec), (bh - max_luma_h) >> (2 + ydec), bo_x << MI_SIZE_LOG2, bo_y << MI_SIZE_LOG2


def chroma_dist_dims(bsize, visible_w, visible_h, xdec, ydec):
    bw, bh = BlockSize(bsize).dims
    cw = (visible_w + xdec) >> xdec if (bw >= 8 or xdec == 0) else (4 + visible_w + xdec) >> xdec
    ch = (visible_h + ydec) >> ydec if (bh >= 8 or ydec == 0) else (4 + visible_h + ydec) >> ydec
    return cw, ch


def scale_distortion(dist, scale_q14):
    """Distortion * DistortionScale (mul_u64): (scale * dist + 2^13) >> 14"""
    return (int(scale_q14) * int(dist) + (1 << DIST_SCALE_SHIFT >> 1)) >> DIST_SCALE_SHIFT


def compute_tx_distortion(sse_wxh, frame_w, frame_h, bsize, is_chroma_block, bo_x, bo_y, tx_dist, skip,
                          luma_only, dist_scale, xdec=1, ydec=1, monochrome=False):
    """compute_tx_distortion for one block.  sse_wxh(plane_index, x, y, w, h) -> the RawDistortion
    of sse_wxh on that plane at plane position (x, y) (r1_dist_scaled_batch, kind WSSE; the per-
    block bias of temporal RDO is inside it).  tx_dist: the ScaledDistortion the transform blocks
    accumulated.  dist_scale: fi.dist_scale (three Q14 values)."""
    bw, bh = BlockSize(bsize).dims
    if not skip:
        vw, vh = bw, bh
    else:
        vw, vh = clip_visible_bsize(frame_w, frame_h, bw, bh, bo_x << MI_SIZE_LOG2, bo_y << MI_SIZE_LOG2)
    if vw == 0 or vh == 0:
        return 0
    x, y = bo_x << MI_SIZE_LOG2, bo_y << MI_SIZE_LOG2
    dist = scale_distortion(sse_wxh(0, x, y, vw, vh), dist_scale[0]) if skip else int(tx_dist)
    if is_chroma_block and not luma_only and skip and not monochrome:
        cw, ch = chroma_dist_dims(bsize, vw, vh, xdec, ydec)
        # Area::BlockStartingAt on a decimated plane: (bo >> dec) << 2 (tiling/plane_region.rs:100-108)
        cx, cy = (bo_x >> xdec) << MI_SIZE_LOG2, (bo_y >> ydec) << MI_SIZE_LOG2
        for p in (1, 2):
            dist += scale_distortion(sse_wxh(p, cx, cy, cw, ch), dist_scale[p])
    return dist


def compute_distortion(dist_wxh, frame_w, frame_h, bsize, is_chroma_block, bo_x, bo_y, luma_only, dist_scale,
                       tune_psychovisual, xdec=1, ydec=1, monochrome=False):
    """compute_distortion for one block.  dist_wxh(kind, plane_index, x, y, w, h) -> the RawDistortion of
    sse_wxh (kind 2) / cdef_dist_wxh (kind 3) on that plane at plane position (x, y) over the VISIBLE w x h
    (r1_dist_scaled_batch; the per-importance-block bias of temporal RDO is inside it).  Luma: cdef_dist_wxh
    under Tune::Psychovisual, sse_wxh under Tune::Psnr, times fi.dist_scale[0]; chroma (is_chroma_block and
    not luma_only): sse_wxh per plane on the decimated planes, times fi.dist_scale[p]."""
    bw, bh = BlockSize(bsize).dims
    x, y = bo_x << MI_SIZE_LOG2, bo_y << MI_SIZE_LOG2
    vw, vh = clip_visible_bsize(frame_w, frame_h, bw, bh, x, y)
    if vw == 0 or vh == 0:
        return 0
    dist = scale_distortion(dist_wxh(3 if tune_psychovisual else 2, 0, x, y, vw, vh), dist_scale[0])
    if is_chroma_block and not luma_only and not monochrome:
        cw, ch = chroma_dist_dims(bsize, vw, vh, xdec, ydec)
        cx, cy = (bo_x >> xdec) << MI_SIZE_LOG2, (bo_y >> ydec) << MI_SIZE_LOG2
        for p in (1, 2):
            dist += scale_distortion(dist_wxh(2, p, cx, cy, cw, ch), dist_scale[p])
    return dist


RAV1E_TX_TYPES = (0, 1, 2, 3, 9, 10, 11)   # DCT_DCT, ADST_DCT, DCT_ADST, ADST_ADST, IDTX, V_DCT, H_DCT


def tx_type_slots(tx_type_mask):
    """slot j of r1_rdo_txsearch_batch's outputs -> TxType: the set bits of the mask in ascending order,
    which is also the order of RAV1E_TX_TYPES (the loop order of rdo_tx_type_decision)"""
    return [t for t in range(16) if (int(tx_type_mask) >> t) & 1]


RESTORATION_TILESIZE_MAX_LOG2 = 8


def tx_split_blocks(bx, by, bw, bh, tw, th, mi_width, mi_height):
    """The transform blocks write_tx_tree (src/encoder.rs:2440-2476) visits for a block at 4x4-unit position (bx, by) of
    bw x bh pixels coded with tw x th transforms: raster order, blocks that START outside the tile's 4x4 grid skipped.
    -> [(index in the bw/tw x bh/th grid, pixel x, pixel y)].  For an inter block these are the candidates of the next
    transform depth (rdo_tx_size_type, src/rdo.rs:745-815): one r1_rdo_txsearch_batch launch in its dense-prediction
    form, the block's prediction cut into the same rectangles."""
    out = []
    for j in range(bh // th):
        for i in range(bw // tw):
            tx, ty = bx + i * (tw // 4), by + j * (th // 4)
            if tx >= mi_width or ty >= mi_height:
                continue
            out.append((j * (bw // tw) + i, tx * 4, ty * 4))
    return out


def restoration_plane_configs(width, height, xdec, ydec, base_q_idx, enable_large_lru=True, enable_restoration=True,
                              use_128x128_superblock=False, tiling=(1, 1, 0, 0)):
    """RestorationState::new (src/lrf.rs:1321-1480): the restoration-unit geometry of the three planes of a frame --
    per plane dict(unit_size, sb_h_shift, sb_v_shift, stripe_height, cols, rows).  tiling = (cols, rows,
    tile_width_sb, tile_height_sb).  cols / rows follow the specification's last-unit rule (a remainder of less than
    half a unit stretches the last unit instead of starting a new one)."""
    stripe_uv_decimate = 1 if (xdec > 0 and ydec > 0) else 0
    y_sb_log2 = 7 if use_128x128_superblock else 6
    uv_sb_h_log2, uv_sb_v_log2 = y_sb_log2 - xdec, y_sb_log2 - ydec
    if enable_large_lru and enable_restoration:
        assert width > 1 and height > 1
        lrf_base_shift = 0 if base_q_idx > 200 else (1 if base_q_idx > 160 else 2)
        if stripe_uv_decimate:
            if lrf_base_shift == 2:
                lrf_chroma_shift = 1
            else:
                us = 1 << (RESTORATION_TILESIZE_MAX_LOG2 - lrf_base_shift)
                unshifted = ((width >> xdec) - 1) % us <= us // 2 or ((height >> ydec) - 1) % us <= us // 2
                shifted = ((width >> xdec) - 1) % (us >> 1) <= us // 4 or ((height >> ydec) - 1) % (us >> 1) <= us // 4
                lrf_chroma_shift = 1 if (unshifted and not shifted) else 0
        else:
            lrf_chroma_shift = 0
        lrf_y_shift, lrf_uv_shift = lrf_base_shift, lrf_base_shift + lrf_chroma_shift
    else:
        lrf_y_shift = 1 if use_128x128_superblock else 2
        lrf_uv_shift = lrf_y_shift + stripe_uv_decimate
    y_unit = 1 << (RESTORATION_TILESIZE_MAX_LOG2 - lrf_y_shift)
    uv_unit = 1 << (RESTORATION_TILESIZE_MAX_LOG2 - lrf_uv_shift)
    t_cols, t_rows, tw_sb, th_sb = tiling
    if t_cols > 1 or t_rows > 1:
        tz = lambda v: (v & -v).bit_length() - 1 if v else 64     # usize::trailing_zeros (0 -> the type's width)
        hz, vz = tz(tw_sb), tz(th_sb)
        y_unit = min(y_unit, 1 << (y_sb_log2 + min(hz, vz)))
        uv_unit = min(uv_unit, min(1 << (uv_sb_h_log2 + hz), 1 << (uv_sb_v_log2 + vz)))
    if ydec == 0 and y_unit != uv_unit:
        y_unit = min(uv_unit, y_unit)
        uv_unit = y_unit
    y_log2, uv_log2 = y_unit.bit_length() - 1, uv_unit.bit_length() - 1
    y_cols = max((width + (y_unit >> 1)) // y_unit, 1)
    y_rows = max((height + (y_unit >> 1)) // y_unit, 1)
    uv_cols = max((((width + (1 << xdec >> 1)) >> xdec) + (uv_unit >> 1)) // uv_unit, 1)
    uv_rows = max((((height + (1 << ydec >> 1)) >> ydec) + (uv_unit >> 1)) // uv_unit, 1)
    y = dict(unit_size=y_unit, sb_h_shift=y_log2 - y_sb_log2, sb_v_shift=y_log2 - y_sb_log2, stripe_height=64, cols=y_cols,
             rows=y_rows)
    uv = dict(unit_size=uv_unit, sb_h_shift=uv_log2 - uv_sb_h_log2, sb_v_shift=uv_log2 - uv_sb_v_log2,
              stripe_height=64 >> stripe_uv_decimate, cols=uv_cols, rows=uv_rows)
    return [y, dict(uv), dict(uv)]


def restoration_search_units(cfg, crop_w, crop_h, dec_x=0, dec_y=0):
    """The (x, y, vis_w, vis_h) rectangles the restoration leg of rdo_loop_decision solves and filters, for a WHOLE
    plane: one per restoration unit (x, y) < (cfg.cols, cfg.rows) -- has_restoration_unit(.., stretch = false) -- at
    loop_sbo.plane_offset = unit index x unit_size, clipped to the visible plane: vis = min(unit_size, (crop >> dec) -
    offset) (src/rdo.rs:2645-2654; the stretched remainder of a last unit is filtered at frame time, not searched).
    This is the `units` list of r1_lrf_search_batch (one entry per parameter set each)."""
    us, out = cfg["unit_size"], []
    pw, ph = crop_w >> dec_x, crop_h >> dec_y
    for uy in range(cfg["rows"]):
        for ux in range(cfg["cols"]):
            x, y = ux * us, uy * us
            if x < pw and y < ph:
                out.append((x, y, min(us, pw - x), min(us, ph - y)))
    return out


def restoration_area_sb(cfgs):
    """The area one rdo_loop_decision call decides, in superblocks: the largest restoration unit of the planes
    (src/rdo.rs:2119-2141).  cfgs: restoration_plane_configs(..)."""
    return (max(1 << c["sb_h_shift"] for c in cfgs), max(1 << c["sb_v_shift"] for c in cfgs))


SGR_EDGE_LEFT, SGR_EDGE_ABOVE = 1, 2


def restoration_unit_edges(x, y, dec_x, dec_y, area_sb, sb_log2=6):
    """R1SgrSolveUnit.edges of the unit at plane pixel (x, y): rdo_loop_decision filters on a scratch copy of the
    area (src/rdo.rs:2277-2296), so setup_integral_image (src/lrf.rs: `cdeffed.x == 0`, `clamp(y, 0, ..)`) finds
    pixels left of / above the unit exactly when the unit does not start the area's first unit column / row."""
    aw, ah = (area_sb[0] << sb_log2) >> dec_x, (area_sb[1] << sb_log2) >> dec_y
    return (SGR_EDGE_LEFT if x % aw else 0) | (SGR_EDGE_ABOVE if y % ah else 0)


def compute_rd_cost(lam, rate, distortion):
    """compute_rd_cost (src/rdo.rs:718-723): fi.lambda.mul_add(rate / 8, distortion) -- ONE rounding (f64 fused
    multiply-add), reproduced here through exact rationals"""
    from fractions import Fraction
    return float(Fraction(float(lam)) * Fraction(int(rate), 8) + Fraction(int(distortion)))


def pick_restoration_filter(options, lam, best_cost=-1.0):
    """The choice of one restoration unit (src/rdo.rs:2617-2745).  options: [(rate, err)] in the order the leg tries
    them -- the no-filter option first, then one per parameter set, err = what r1_lrf_search_batch returned, rate =
    cw.fc.count_lrf_switchable for that option.  Returns (index into options or None when nothing beat best_cost,
    the cost kept).  First smallest cost wins; the no-filter option also wins when no cost was kept before."""
    pick = None
    for i, (rate, err) in enumerate(options):
        cost = compute_rd_cost(lam, rate, err)
        if (i == 0 and best_cost < 0.0) or cost < best_cost:
            best_cost, pick = cost, i
    return pick, best_cost

