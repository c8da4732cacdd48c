/*
 * oracle/cdef.c -- TEST INFRASTRUCTURE (see r1_oracle.h).
 * Scalar restatement of the reference's CDEF
 *   first_max_element, cdef_find_dir     src/cdef.rs:64-143
 *   constrain                            src/cdef.rs:146-159
 *   pad_into_tmp16, cdef_filter_block    src/cdef.rs:161-298
 *   adjust_strength                      src/cdef.rs:313-321
 *   cdef_analyze_superblock              src/cdef.rs:340-373
 *   cdef_filter_superblock / _tile       src/cdef.rs:405-625
 * msb() is v_frame 0.3.9 math::msb (31 - leading_zeros).
 */
#include <stdlib.h>
#include <string.h>

#include "r1_oracle.h"

#define VERY_LARGE 0x8000
enum { HAVE_LEFT = 1, HAVE_RIGHT = 2, HAVE_TOP = 4, HAVE_BOTTOM = 8, HAVE_ALL = 15 };

static inline int32_t getp(const void *p, int hbd, ptrdiff_t i) {
  return hbd ? ((const uint16_t *)p)[i] : ((const uint8_t *)p)[i];
}
static inline int msb(int32_t x) { return 31 - __builtin_clz((uint32_t)x); }

/* img: top-left of the 8x8 luma block; returns dir, writes var */
int r1o_cdef_find_dir(const void *img, ptrdiff_t stride, uint32_t *var, int coeff_shift,
                      int hbd) {
  static const int32_t DIV[9] = {0, 840, 420, 280, 210, 168, 140, 120, 105};
  int32_t cost[8] = {0}, partial[8][15];
  memset(partial, 0, sizeof(partial));
  for (int i = 0; i < 8; i++)
    for (int j = 0; j < 8; j++) {
      const int32_t x = (getp(img, hbd, i * stride + j) >> coeff_shift) - 128;
      partial[0][i + j] += x;
      partial[1][i + j / 2] += x;
      partial[2][i] += x;
      partial[3][3 + i - j / 2] += x;
      partial[4][7 + i - j] += x;
      partial[5][3 - i / 2 + j] += x;
      partial[6][j] += x;
      partial[7][i / 2 + j] += x;
    }
  for (int i = 0; i < 8; i++) {
    cost[2] += partial[2][i] * partial[2][i];
    cost[6] += partial[6][i] * partial[6][i];
  }
  cost[2] *= DIV[8];
  cost[6] *= DIV[8];
  for (int i = 0; i < 7; i++) {
    cost[0] += (partial[0][i] * partial[0][i] + partial[0][14 - i] * partial[0][14 - i]) * DIV[i + 1];
    cost[4] += (partial[4][i] * partial[4][i] + partial[4][14 - i] * partial[4][14 - i]) * DIV[i + 1];
  }
  cost[0] += partial[0][7] * partial[0][7] * DIV[8];
  cost[4] += partial[4][7] * partial[4][7] * DIV[8];
  for (int i = 1; i < 8; i += 2) {
    for (int j = 0; j < 5; j++) cost[i] += partial[i][3 + j] * partial[i][3 + j];
    cost[i] *= DIV[8];
    for (int j = 0; j < 3; j++)
      cost[i] += (partial[i][j] * partial[i][j] + partial[i][10 - j] * partial[i][10 - j]) *
                 DIV[2 * j + 2];
  }
  int best = 0;                       /* first maximum wins ties (cdef.rs:64-73) */
  for (int i = 1; i < 8; i++)
    if (cost[i] > cost[best]) best = i;
  *var = (uint32_t)((cost[best] - cost[(best + 4) & 7]) >> 10);
  return best;
}

static inline int32_t constrain(int32_t diff, int32_t threshold, int32_t damping) {
  if (!threshold) return 0;
  int shift = damping - msb(threshold);
  if (shift < 0) shift = 0;
  const int32_t ad = abs(diff);
  int32_t mag = threshold - (ad >> shift);
  mag = mag < 0 ? 0 : (mag > ad ? ad : mag);
  return diff < 0 ? -mag : mag;
}

/* input: top-left pixel of the block in the (deblocked) source plane; pixels
 * two beyond each side flagged in `edges` must be readable.  dst: top-left of
 * the destination block. */
void r1o_cdef_filter_block(void *dst, ptrdiff_t dstride, const void *input, ptrdiff_t istride,
                           int pri_strength, int sec_strength, int dir, int damping,
                           int bit_depth, int xdec, int ydec, int edges, int hbd) {
  const int xsize = 8 >> xdec, ysize = 8 >> ydec;
  /* always go through the padded u16 copy: with HAVE_ALL every halo pixel is
   * real, which is exactly the fast path's direct read */
  uint16_t tmp[12 * 12];
  for (int i = 0; i < 144; i++) tmp[i] = VERY_LARGE;
  const int x0 = (edges & HAVE_LEFT) ? -2 : 0, x1 = xsize + ((edges & HAVE_RIGHT) ? 2 : 0);
  const int y0 = (edges & HAVE_TOP) ? -2 : 0, y1 = ysize + ((edges & HAVE_BOTTOM) ? 2 : 0);
  for (int y = y0; y < y1; y++)
    for (int x = x0; x < x1; x++)
      tmp[(y + 2) * 12 + (x + 2)] = (uint16_t)getp(input, hbd, y * istride + x);
  const int coeff_shift = bit_depth - 8;
  static const int PRI[2][2] = {{4, 2}, {3, 3}}, SEC[2][2] = {{2, 1}, {2, 1}};
  const int *pri_taps = PRI[(pri_strength >> coeff_shift) & 1];
  const int *sec_taps = SEC[(pri_strength >> coeff_shift) & 1];
  /* (dy, dx) for k = 0, 1 -- the reference's cdef_directions table */
  static const int8_t D[8][2][2] = {{{-1, 1}, {-2, 2}}, {{0, 1}, {-1, 2}}, {{0, 1}, {0, 2}},
                                    {{0, 1}, {1, 2}},   {{1, 1}, {2, 2}},  {{1, 0}, {2, 1}},
                                    {{1, 0}, {2, 0}},   {{1, 0}, {2, -1}}};
  for (int i = 0; i < ysize; i++)
    for (int j = 0; j < xsize; j++) {
      const uint16_t *c = tmp + (i + 2) * 12 + (j + 2);
      const int32_t x = *c;
      int32_t sum = 0, mx = x, mn = x;
      for (int k = 0; k < 2; k++) {
        const int o0 = D[dir][k][0] * 12 + D[dir][k][1];
        const int o1 = D[(dir + 2) & 7][k][0] * 12 + D[(dir + 2) & 7][k][1];
        const int o2 = D[(dir + 6) & 7][k][0] * 12 + D[(dir + 6) & 7][k][1];
        const int32_t p[2] = {c[o0], c[-o0]};
        for (int t = 0; t < 2; t++) {
          sum += pri_taps[k] * constrain(p[t] - x, pri_strength, damping);
          if (p[t] != VERY_LARGE && p[t] > mx) mx = p[t];
          if (p[t] < mn) mn = p[t];
        }
        const int32_t s[4] = {c[o1], c[-o1], c[o2], c[-o2]};
        for (int t = 0; t < 4; t++) {
          if (s[t] != VERY_LARGE && s[t] > mx) mx = s[t];
          if (s[t] < mn) mn = s[t];
          sum += sec_taps[k] * constrain(s[t] - x, sec_strength, damping);
        }
      }
      int32_t v = x + ((8 + sum - (sum < 0)) >> 4);
      v = v < mn ? mn : (v > mx ? mx : v);
      if (hbd) ((uint16_t *)dst)[i * dstride + j] = (uint16_t)v;
      else ((uint8_t *)dst)[i * dstride + j] = (uint8_t)v;
    }
}

int r1o_cdef_adjust_strength(int strength, int var) {
  const int i = (var >> 6) != 0 ? (msb(var >> 6) < 12 ? msb(var >> 6) : 12) : 0;
  return var != 0 ? (strength * (4 + i) + 8) >> 4 : 0;
}

/* cdef_filter_tile for ONE plane p, on the whole frame -- the only way the
 * reference calls it (src/encoder.rs:3301-3321; cdef_analyze_superblock even
 * addresses the input frame with the tile-relative superblock offset,
 * cdef.rs:355-357, so a non-zero tile origin is not meaningful).
 *   luma / in / out: whole-frame planes (r1o_plane); `in` is plane p of the
 *     deblocked frame, `out` the destination (same geometry).
 *   tile_w, tile_h: output.planes[0].rect() size in LUMA pixels.
 *   skip_mi: TileBlocks skip flags, one per 4x4 luma block of the tile
 *     (row stride mi_stride; mi_cols x mi_rows valid).
 *   cdef_index_sb: per 64x64 superblock of the tile (row stride sb_stride).
 *   y_strengths / uv_strengths: fi.cdef_y_strengths / cdef_uv_strengths. */
void r1o_cdef_filter_tile_plane(const r1o_plane *luma, const r1o_plane *in, const r1o_plane *out,
                                int p, int xdec, int ydec, int tile_w, int tile_h,
                                const uint8_t *skip_mi, int mi_stride, int mi_cols,
                                int mi_rows, const uint8_t *cdef_index_sb, int sb_stride,
                                const uint8_t *y_strengths, const uint8_t *uv_strengths,
                                int damping, int bit_depth) {
  const int hbd = in->bytes_per_px == 2, bpp = in->bytes_per_px;
  const int tile_x = 0, tile_y = 0;
  const int coeff_shift = bit_depth - 8;
  const int fbw = (tile_w + 63) / 64, fbh = (tile_h + 63) / 64;
  for (int fby = 0; fby < fbh; fby++)
    for (int fbx = 0; fbx < fbw; fbx++) {
      const int ci = cdef_index_sb[fby * sb_stride + fbx];
      const int ys = y_strengths[ci], uvs = uv_strengths[ci];
      const int pri_y = ys / 4, pri_uv = uvs / 4;
      int sec_y = ys % 4, sec_uv = uvs % 4;
      if (sec_y == 3) sec_y++;
      if (sec_uv == 3) sec_uv++;
      const int in_xoff = tile_x + fbx * 64, in_yoff = tile_y + fby * 64;
      const int xavail = luma->width - in_xoff, yavail = luma->height - in_yoff;
      const int have_top = (fby + tile_y > 0) ? HAVE_TOP : 0;
      const int have_left = (fbx + tile_x > 0) ? HAVE_LEFT : 0;
      int edges = have_top | HAVE_BOTTOM;
      for (int by = 0; by < 8; by++) {
        if (by + 1 >= (yavail >> 3)) edges &= ~HAVE_BOTTOM;
        edges &= ~HAVE_LEFT;
        edges |= have_left;
        edges |= HAVE_RIGHT;
        for (int bx = 0; bx < 8; bx++) {
          if (bx + 1 >= (xavail >> 3)) edges &= ~HAVE_RIGHT;
          const int mx = fbx * 16 + 2 * bx, my = fby * 16 + 2 * by;
          if (mx < mi_cols && my < mi_rows) {
            /* Block::skip is a bool: any non-zero byte is true */
            const int skip = skip_mi[my * mi_stride + mx] && skip_mi[my * mi_stride + mx + 1] &&
                             skip_mi[(my + 1) * mi_stride + mx] &&
                             skip_mi[(my + 1) * mi_stride + mx + 1];
            const int xs = 8 >> xdec, ysz = 8 >> ydec;
            const int px = (in_xoff >> xdec) + bx * xs, py = (in_yoff >> ydec) + by * ysz;
            const uint8_t *src = (const uint8_t *)in->data +
                                 ((size_t)(in->yorigin + py) * in->stride + in->xorigin + px) * bpp;
            uint8_t *dst = (uint8_t *)out->data +
                           ((size_t)(out->yorigin + py) * out->stride + out->xorigin + px) * bpp;
            if (!skip) {
              uint32_t var = 0;
              const uint8_t *lsrc =
                  (const uint8_t *)luma->data +
                  ((size_t)(luma->yorigin + in_yoff + 8 * by) * luma->stride + luma->xorigin +
                   in_xoff + 8 * bx) * luma->bytes_per_px;
              const int dir = r1o_cdef_find_dir(lsrc, luma->stride, &var, coeff_shift,
                                                luma->bytes_per_px == 2);
              int lpri, lsec, ldamp = damping + coeff_shift, ldir;
              if (p == 0) {
                lpri = r1o_cdef_adjust_strength(pri_y << coeff_shift, (int)var);
                lsec = sec_y << coeff_shift;
                ldir = pri_y != 0 ? dir : 0;
              } else {
                static const uint8_t UVDIR[8] = {7, 0, 2, 4, 5, 6, 6, 6};
                lpri = pri_uv << coeff_shift;
                lsec = sec_uv << coeff_shift;
                ldamp -= 1;
                ldir = pri_uv != 0 ? (xdec != ydec ? UVDIR[dir] : dir) : 0;
              }
              r1o_cdef_filter_block(dst, out->stride, src, in->stride, lpri, lsec, ldir, ldamp,
                                    bit_depth, xdec, ydec, edges, hbd);
            } else {
              for (int i = 0; i < ysz; i++)
                memcpy(dst + (size_t)i * out->stride * bpp, src + (size_t)i * in->stride * bpp,
                       (size_t)xs * bpp);
            }
          }
          edges |= HAVE_LEFT;
        }
        edges |= HAVE_TOP;
      }
    }
}

/* The CDEF leg of rdo_loop_decision (src/rdo.rs:2104-2560) when no restoration filter is
 * in play (RestorationFilter::None / no LRU, rdo.rs:2432-2451, 2504-2520): the strength
 * search.  The reference cuts the deblocked reconstruction into analysis areas of
 * area_sb_w x area_sb_h superblocks (the extent of the largest restoration unit; one
 * superblock when restoration is off), works on a scratch copy of each (rdo.rs:2277-2284) --
 * so CDEF sees the AREA's borders as picture edges -- and for every superblock that is not
 * completely skipped tries cdef_index 0 .. n_idx-1:
 *   cdef_filter_superblock with that index (cdef.rs:405-560),
 *   err = sum over planes of [sum over the 8x8 blocks of the superblock inside the block
 *         grid of: cdef_dist_kernel * bias (luma) / sse_wxh with bias (chroma)] * dist_scale[pli]
 *         (rdo_loop_plane_error, rdo.rs:2027-2093),
 *   cost = compute_rd_cost(rate = 0, err) = err as f64 (rdo.rs:718-723); the first index with
 *   the smallest cost wins (rdo.rs:2523-2527).
 * rec / src: whole-frame planes (rec deblocked), areas at multiples of the area size; scales:
 * fi.coded_frame_data.distortion_scales, one per 8x8 luma block of the frame (NULL = default).
 * err: [n_sby][n_sbx][8] (zero for skipped superblocks), best: [n_sby][n_sbx], -1 = skipped. */
int r1o_cdef_strength_search(const r1o_plane *rec, const r1o_plane *src, const uint8_t *skip_mi,
                             int mi_stride, int mi_cols, int mi_rows, const uint32_t *scales,
                             int scale_stride, const r1o_cdef_search_params *p, uint64_t *err,
                             int8_t *best) {
  if (p->n_idx < 1 || p->n_idx > 8 || (p->planes != 1 && p->planes != 3)) return -1;
  if (p->area_sb_w < 1 || p->area_sb_h < 1) return -1;
  const int bd = p->bit_depth, coeff_shift = bd - 8, hbd = rec[0].bytes_per_px == 2;
  const int n_sbx = (mi_cols + 15) / 16, n_sby = (mi_rows + 15) / 16;
  memset(err, 0, sizeof(uint64_t) * 8 * (size_t)n_sbx * n_sby);
  for (int fby = 0; fby < n_sby; fby++)
    for (int fbx = 0; fbx < n_sbx; fbx++) {
      /* the area this superblock belongs to, and the area's own geometry */
      const int ax0 = fbx / p->area_sb_w * p->area_sb_w, ay0 = fby / p->area_sb_h * p->area_sb_h;
      const int sbx = fbx - ax0, sby = fby - ay0;
      const int sb_w = p->area_sb_w < n_sbx - ax0 ? p->area_sb_w : n_sbx - ax0;
      const int sb_h = p->area_sb_h < n_sby - ay0 ? p->area_sb_h : n_sby - ay0;
      const int crop_w = p->crop_w - ax0 * 64, crop_h = p->crop_h - ay0 * 64;
      const int pixel_w = crop_w < sb_w * 64 ? crop_w : sb_w * 64;
      const int pixel_h = crop_h < sb_h * 64 ? crop_h : sb_h * 64;
      const int area_w = (pixel_w + 7) >> 3 << 3, area_h = (pixel_h + 7) >> 3 << 3;
      /* tileblocks_subset: sb_w * 16 x sb_h * 16 block units, clipped to the block grid */
      const int blk_cols = sb_w * 16 < mi_cols - ax0 * 16 ? sb_w * 16 : mi_cols - ax0 * 16;
      const int blk_rows = sb_h * 16 < mi_rows - ay0 * 16 ? sb_h * 16 : mi_rows - ay0 * 16;
      int8_t *bo = best + fby * n_sbx + fbx;
      uint64_t *eo = err + 8 * ((size_t)fby * n_sbx + fbx);
      /* cdef_skip (rdo.rs:2196-2211) */
      int all_skip = 1;
      for (int y = 16 * sby; y < 16 * sby + 16 && y < blk_rows; y++)
        for (int x = 16 * sbx; x < 16 * sbx + 16 && x < blk_cols; x++)
          all_skip &= skip_mi[(size_t)(ay0 * 16 + y) * mi_stride + ax0 * 16 + x] != 0;
      if (all_skip) { *bo = -1; continue; }
      /* cdef_filter_superblock's edge logic on the area frame (cdef.rs:424-459) */
      const int in_xoff = sbx * 64, in_yoff = sby * 64;
      const int xavail = area_w - in_xoff, yavail = area_h - in_yoff;
      const int have_top = sby > 0 ? HAVE_TOP : 0, have_left = sbx > 0 ? HAVE_LEFT : 0;
      for (int idx = 0; idx < p->n_idx; idx++) {
        const int ys = p->y_strengths[idx], uvs = p->uv_strengths[idx];
        const int pri_y = ys / 4, pri_uv = uvs / 4;
        int sec_y = ys % 4, sec_uv = uvs % 4;
        if (sec_y == 3) sec_y++;
        if (sec_uv == 3) sec_uv++;
        uint64_t plane_sum[3] = {0, 0, 0};
        int edges = have_top | HAVE_BOTTOM;
        for (int by = 0; by < 8; by++) {
          if (by + 1 >= (yavail >> 3)) edges &= ~HAVE_BOTTOM;
          edges &= ~HAVE_LEFT;
          edges |= have_left;
          edges |= HAVE_RIGHT;
          for (int bx = 0; bx < 8; bx++) {
            if (bx + 1 >= (xavail >> 3)) edges &= ~HAVE_RIGHT;
            const int mx = sbx * 16 + 2 * bx, my = sby * 16 + 2 * by;   /* area block units */
            if (mx < blk_cols && my < blk_rows) {
              const uint8_t *sk = skip_mi + (size_t)(ay0 * 16 + my) * mi_stride + ax0 * 16 + mx;
              const int skip = sk[0] && sk[1] && sk[mi_stride] && sk[mi_stride + 1];
              /* frame position of the block in luma pixels */
              const int flx = ax0 * 64 + in_xoff + 8 * bx, fly = ay0 * 64 + in_yoff + 8 * by;
              uint32_t var = 0;
              int dir = 0;
              if (!skip) {
                const r1o_plane *l = &rec[0];
                dir = r1o_cdef_find_dir((const uint8_t *)l->data +
                                            ((size_t)(l->yorigin + fly) * l->stride + l->xorigin + flx) *
                                                l->bytes_per_px,
                                        l->stride, &var, coeff_shift, hbd);
              }
              const uint32_t bias = scales ? scales[(size_t)(fly >> 3) * scale_stride + (flx >> 3)] : (1u << 14);
              for (int pl = 0; pl < p->planes; pl++) {
                const int xdec = pl ? p->xdec : 0, ydec = pl ? p->ydec : 0;
                const int xs = 8 >> xdec, ysz = 8 >> ydec;
                const r1o_plane *rp = &rec[pl], *sp = &src[pl];
                const int px = flx >> xdec, py = fly >> ydec;
                const uint8_t *rpx = (const uint8_t *)rp->data +
                                     ((size_t)(rp->yorigin + py) * rp->stride + rp->xorigin + px) * rp->bytes_per_px;
                const uint8_t *spx = (const uint8_t *)sp->data +
                                     ((size_t)(sp->yorigin + py) * sp->stride + sp->xorigin + px) * sp->bytes_per_px;
                uint16_t blk16[64];
                uint8_t *blk = (uint8_t *)blk16;
                if (!skip) {
                  int lpri, lsec, ldamp = p->damping + coeff_shift, ldir;
                  if (pl == 0) {
                    lpri = r1o_cdef_adjust_strength(pri_y << coeff_shift, (int)var);
                    lsec = sec_y << coeff_shift;
                    ldir = pri_y != 0 ? dir : 0;
                  } else {
                    static const uint8_t UVDIR[8] = {7, 0, 2, 4, 5, 6, 6, 6};
                    lpri = pri_uv << coeff_shift;
                    lsec = sec_uv << coeff_shift;
                    ldamp -= 1;
                    ldir = pri_uv != 0 ? (xdec != ydec ? UVDIR[dir] : dir) : 0;
                  }
                  r1o_cdef_filter_block(blk, 8, rpx, rp->stride, lpri, lsec, ldir, ldamp, bd, xdec, ydec,
                                        edges, hbd);
                } else {   /* the working copy keeps the deblocked pixels */
                  for (int i = 0; i < ysz; i++)
                    memcpy(blk + (size_t)i * 8 * rp->bytes_per_px, rpx + (size_t)i * rp->stride * rp->bytes_per_px,
                           (size_t)xs * rp->bytes_per_px);
                }
                if (pl == 0) {
                  const uint64_t raw = r1o_cdef_dist_kernel(spx, sp->stride, blk, 8, 8, 8, bd, hbd);
                  plane_sum[0] += ((uint64_t)bias * raw + 8192) >> 14;     /* RawDistortion * bias */
                } else {
                  uint32_t cell[4] = {bias, bias, bias, bias};              /* sse_wxh: |_, _| bias */
                  plane_sum[pl] += r1o_get_weighted_sse(spx, sp->stride, blk, 8, cell, 2, xs, ysz, hbd);
                }
              }
            }
            edges |= HAVE_LEFT;
          }
          edges |= HAVE_TOP;
        }
        uint64_t e = 0;
        for (int pl = 0; pl < p->planes; pl++)
          e += ((uint64_t)p->dist_scale[pl] * plane_sum[pl] + 8192) >> 14;   /* Distortion * dist_scale */
        eo[idx] = e;
      }
      int b = 0;   /* f64 costs of u64 errors: exact below 2^53 */
      for (int idx = 1; idx < p->n_idx; idx++)
        if ((double)eo[idx] < (double)eo[b]) b = idx;
      *bo = (int8_t)b;
    }
  return 0;
}
