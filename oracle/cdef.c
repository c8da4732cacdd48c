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
            const int skip = skip_mi[my * mi_stride + mx] & skip_mi[my * mi_stride + mx + 1] &
                             skip_mi[(my + 1) * mi_stride + mx] &
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
