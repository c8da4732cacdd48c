/*
 * oracle/predict.c -- TEST INFRASTRUCTURE (see r1_oracle.h).
 * Scalar restatement of the reference's intra prediction
 *   PredictionMode::predict_intra      src/predict.rs:205-249 (PAETH / CFL remaps)
 *   rust::dispatch_predict_intra       src/predict.rs:705-784
 *   pred_dc/_128/_left/_top, pred_v/h  src/predict.rs:786-857
 *   pred_paeth                         src/predict.rs:859-888
 *   pred_smooth/_h/_v                  src/predict.rs:890-1018
 *   pred_cfl_ac, pred_cfl_inner        src/predict.rs:1020-1099
 *   select_ief_strength/_upsample      src/predict.rs:1133-1201
 *   filter_edge, upsample_edge         src/predict.rs:1203-1266
 *   pred_directional                   src/predict.rs:1301-1505
 * and of the edge builder
 *   get_intra_edges                    src/partition.rs:639-898
 * The availability decisions has_top_right / has_bottom_left depend on the
 * partition tree (encoder state) and stay with the caller: they come in as
 * booleans.
 *
 * Edge layout (IntraEdgeBuffer, src/partition.rs:600-636): a buffer of
 * 4*64+1 pixels; `left` is right-aligned and ends at index 128 (ordered
 * bottom to top), top_left sits at index 128, `above` starts at 129.
 */
#include <stdlib.h>
#include <string.h>

#include "r1_oracle.h"

#define R1_TABLE_QUAL static const
#include "intra_tables.inc"

#define MAXTX 64
enum { DC_PRED = 0, V_PRED, H_PRED, D45_PRED, D135_PRED, D113_PRED, D157_PRED,
       D203_PRED, D67_PRED, SMOOTH_PRED, SMOOTH_V_PRED, SMOOTH_H_PRED, PAETH_PRED,
       UV_CFL_PRED };
enum { VAR_NONE = 0, VAR_LEFT, VAR_TOP, VAR_BOTH };

static inline int32_t getp(const void *p, int hbd, ptrdiff_t i) {
  return hbd ? ((const uint16_t *)p)[i] : ((const uint8_t *)p)[i];
}
static inline void setp(void *p, int hbd, ptrdiff_t i, int32_t v) {
  if (hbd) ((uint16_t *)p)[i] = (uint16_t)v;
  else ((uint8_t *)p)[i] = (uint8_t)v;
}
static inline int32_t clampi(int32_t v, int32_t lo, int32_t hi) {
  return v < lo ? lo : (v > hi ? hi : v);
}

int r1o_intra_mode_to_angle(int mode) {
  static const int16_t a[9] = {0, 90, 180, 45, 135, 113, 157, 203, 67};
  return mode >= 0 && mode < 9 ? a[mode] : 0;
}

int r1o_select_ief_strength(int width, int height, int smooth, int angle_delta) {
  const int wh = width + height, d = abs(angle_delta);
  if (smooth) {
    if (wh <= 8) return d >= 64 ? 2 : (d >= 40 ? 1 : 0);
    if (wh <= 16) return d >= 48 ? 2 : (d >= 20 ? 1 : 0);
    if (wh <= 24) return d >= 4 ? 3 : 0;
    return 3;
  }
  if (wh <= 8) return d >= 56 ? 1 : 0;
  if (wh <= 16) return d >= 40 ? 1 : 0;
  if (wh <= 24) return d >= 32 ? 3 : (d >= 16 ? 2 : (d >= 8 ? 1 : 0));
  if (wh <= 32) return d >= 32 ? 3 : (d >= 4 ? 2 : 1);
  return 3;
}

int r1o_select_ief_upsample(int width, int height, int smooth, int angle_delta) {
  const int wh = width + height, d = abs(angle_delta);
  if (d == 0 || d >= 40) return 0;
  return smooth ? wh <= 8 : wh <= 16;
}

/* edge: i32 working copy, length `len`; filters entries 1..size-1 */
static void filter_edge(int size, int strength, int32_t *edge, int len) {
  static const int K[3][5] = {{0, 4, 8, 4, 0}, {0, 5, 6, 5, 0}, {2, 4, 4, 4, 2}};
  if (strength == 0) return;
  int32_t f[4 * MAXTX + 1];
  memcpy(f, edge, sizeof(int32_t) * len);
  for (int i = 1; i < size; i++) {
    int32_t s = 0;
    for (int j = 0; j < 5; j++) {
      int k = i + j - 2;
      if (k < 0) k = 0;                 /* saturating_sub */
      if (k > size - 1) k = size - 1;
      s += K[strength - 1][j] * edge[k];
    }
    f[i] = (s + 8) >> 4;
  }
  memcpy(edge, f, sizeof(int32_t) * len);
}

static void upsample_edge(int size, int32_t *edge, int bit_depth) {
  int32_t dup[MAXTX + 8];
  dup[0] = edge[0];
  for (int i = 0; i <= size; i++) dup[1 + i] = edge[i];
  dup[size + 2] = edge[size];
  edge[0] = dup[0];
  for (int i = 0; i < size; i++) {
    int32_t s = -dup[i] + 9 * dup[i + 1] + 9 * dup[i + 2] - dup[i + 3];
    s = (s + 8) / 16;                    /* Rust `/`: truncation toward zero */
    s = clampi(s, 0, (1 << bit_depth) - 1);
    edge[2 * i + 1] = s;
    edge[2 * i + 2] = dup[i + 2];
  }
}

/* left: `left_len` pixels ordered bottom->top (left[left_len-1] is beside row 0);
 * above: `above_len` pixels.  avail_w / avail_h = remaining plane extent at the
 * block (min'ed with the block size by the callee), src/predict.rs:1346-1353.
 * ief: 0 = None, 1 = Some(non-smooth), 2 = Some(smooth). */
static void pred_directional(void *out, ptrdiff_t stride, const void *above, int above_len,
                             const void *left, int left_len, int32_t top_left, int p_angle,
                             int width, int height, int bit_depth, int ief, int avail_w,
                             int avail_h, int hbd) {
  const int32_t sample_max = (1 << bit_depth) - 1;
  const int enable = ief != 0;
  int upsample_above = 0, upsample_left = 0;
  /* working edges as i32; index 0 = top-left when the edge filter is enabled */
  int32_t af[4 * MAXTX + 1], lf[4 * MAXTX + 1];
  const int flen = (width + height) * 2 + 1;
  int32_t above_raw[2 * MAXTX], left_raw[2 * MAXTX];
  for (int i = 0; i < above_len; i++) above_raw[i] = getp(above, hbd, i);
  for (int i = 0; i < left_len; i++) left_raw[i] = getp(left, hbd, i);
  const int32_t *above_edge = above_raw, *left_edge = left_raw;
  int left_edge_len = left_len;
  if (enable) {
    memset(af, 0, sizeof(af));
    memset(lf, 0, sizeof(lf));
    const int al = above_len < flen - 1 ? above_len : flen - 1;
    const int ll = left_len < flen - 1 ? left_len : flen - 1;
    for (int i = 0; i < al; i++) af[1 + i] = above_raw[i];
    for (int i = 1; i <= ll; i++) lf[i] = left_raw[left_len - i];
    const int smooth = ief == 2;
    if (p_angle != 90 && p_angle != 180) {
      af[0] = top_left;
      lf[0] = top_left;
      const int npa = (width < avail_w ? width : avail_w) + (p_angle < 90 ? height : 0) + 1;
      const int npl = (height < avail_h ? height : avail_h) + (p_angle > 180 ? width : 0) + 1;
      filter_edge(npa, r1o_select_ief_strength(width, height, smooth, p_angle - 90), af, flen);
      filter_edge(npl, r1o_select_ief_strength(width, height, smooth, p_angle - 180), lf, flen);
    }
    const int na = width + (p_angle < 90 ? height : 0);
    const int nl = height + (p_angle > 180 ? width : 0);
    upsample_above = r1o_select_ief_upsample(width, height, smooth, p_angle - 90);
    if (upsample_above) upsample_edge(na, af, bit_depth);
    upsample_left = r1o_select_ief_upsample(width, height, smooth, p_angle - 180);
    if (upsample_left) upsample_edge(nl, lf, bit_depth);
    /* left_filtered.reverse() */
    for (int i = 0; i < flen / 2; i++) {
      int32_t t = lf[i];
      lf[i] = lf[flen - 1 - i];
      lf[flen - 1 - i] = t;
    }
    above_edge = af;
    left_edge = lf;
    left_edge_len = flen;
  }
  int dx = 0, dy = 0;
  if (p_angle < 90) dx = kR1DrIntraDerivative[p_angle];
  else if (p_angle > 90 && p_angle < 180) dx = kR1DrIntraDerivative[180 - p_angle];
  if (p_angle > 90 && p_angle < 180) dy = kR1DrIntraDerivative[p_angle - 90];
  else if (p_angle > 180) dy = kR1DrIntraDerivative[270 - p_angle];
  const int offset_above = enable << upsample_above;
  const int offset_left = enable << upsample_left;
  const int l = left_edge_len - 1;
  for (int i = 0; i < height; i++)
    for (int j = 0; j < width; j++) {
      int32_t v;
      if (p_angle < 90) {
        const int idx = (i + 1) * dx;
        const int base = (idx >> (6 - upsample_above)) + (j << upsample_above);
        const int shift = ((idx << upsample_above) >> 1) & 31;
        const int max_base_x = (height + width - 1) << upsample_above;
        if (base < max_base_x) {
          const int32_t a = above_edge[base + offset_above], b = above_edge[base + 1 + offset_above];
          v = (a * (32 - shift) + b * shift + 16) >> 5;
        } else {
          v = above_edge[max_base_x + offset_above];
        }
      } else if (p_angle > 90 && p_angle < 180) {
        int idx = (j << 6) - (i + 1) * dx;
        int base = idx >> (6 - upsample_above);
        if (base >= -(1 << upsample_above)) {
          const int shift = ((idx << upsample_above) >> 1) & 31;
          const int32_t a = (!enable && base < 0) ? top_left : above_edge[base + offset_above];
          const int32_t b = above_edge[base + 1 + offset_above];
          v = (a * (32 - shift) + b * shift + 16) >> 5;
        } else {
          idx = (i << 6) - (j + 1) * dy;
          base = idx >> (6 - upsample_left);
          const int shift = ((idx << upsample_left) >> 1) & 31;
          int32_t a, b;
          if (!enable && base < 0) a = top_left;
          else if (base + offset_left == -2) a = left_edge[0];
          else a = left_edge[l - (base + offset_left)];
          if (base + offset_left == -2) b = left_edge[1];
          else b = left_edge[l - (base + offset_left + 1)];
          v = (a * (32 - shift) + b * shift + 16) >> 5;
        }
      } else { /* p_angle > 180 */
        const int idx = (j + 1) * dy;
        const int base = (idx >> (6 - upsample_left)) + (i << upsample_left);
        const int shift = ((idx << upsample_left) >> 1) & 31;
        int ia = l - (base + offset_left), ib = l - (base + offset_left + 1);
        if (ia < 0) ia = 0;              /* saturating_sub */
        if (ib < 0) ib = 0;
        v = (left_edge[ia] * (32 - shift) + left_edge[ib] * shift + 16) >> 5;
      }
      setp(out, hbd, i * stride + j, clampi(v, 0, sample_max));
    }
}

static void fill(void *out, ptrdiff_t stride, int w, int h, int32_t v, int hbd) {
  for (int r = 0; r < h; r++)
    for (int c = 0; c < w; c++) setp(out, hbd, r * stride + c, v);
}

static int32_t scaled_luma_q0(int alpha_q3, int ac_q3) {
  const int32_t q6 = alpha_q3 * ac_q3;
  const int32_t a = (abs(q6) + 32) >> 6;
  return q6 < 0 ? -a : a;
}

/* edge: pointer to the IntraEdgeBuffer base (257 pixels); left_len / above_len
 * = init_left / init_above of IntraEdge::new.  `mode`, `variant`, `angle` are
 * the arguments of dispatch_predict_intra (after predict_intra's remaps; for
 * UV_CFL_PRED `angle` carries alpha).  ac: w*h i16 for CFL else NULL.
 * Returns -1 for modes the reference leaves unimplemented. */
int r1o_dispatch_predict_intra(int mode, int variant, void *dst, ptrdiff_t stride,
                               int tx_size, int bit_depth, const int16_t *ac, int angle,
                               int ief, const void *edge, int left_len, int above_len,
                               int avail_w, int avail_h, int hbd) {
  const int width = r1o_tx_width(tx_size), height = r1o_tx_height(tx_size);
  const int bpp = hbd ? 2 : 1;
  const uint8_t *e = (const uint8_t *)edge;
  const void *above = e + (size_t)(2 * MAXTX + 1) * bpp;
  const int32_t top_left = getp(edge, hbd, 2 * MAXTX);
  /* left_slice = last `height` of left; left_and_left_below = last w+h (saturating) */
  const int ls_len = left_len < height ? left_len : height;
  const void *left_slice = e + (size_t)(2 * MAXTX - ls_len) * bpp;
  const int lb_len = left_len < width + height ? left_len : width + height;
  const void *left_below = e + (size_t)(2 * MAXTX - lb_len) * bpp;
  int do_dc = -1;
  switch (mode) {
    case DC_PRED: do_dc = variant; break;
    case UV_CFL_PRED: do_dc = variant; break;
    case V_PRED:
      if (angle == 90) {
        for (int r = 0; r < height; r++)
          for (int c = 0; c < width; c++) setp(dst, hbd, r * stride + c, getp(above, hbd, c));
        return 0;
      }
      /* fallthrough */
    case H_PRED:
      if (mode == H_PRED && angle == 180) {
        for (int r = 0; r < height; r++)
          fill((uint8_t *)dst + (size_t)r * stride * bpp, stride, width, 1,
               getp(left_slice, hbd, ls_len - 1 - r), hbd);
        return 0;
      }
      /* fallthrough */
    case D45_PRED: case D135_PRED: case D113_PRED: case D157_PRED: case D203_PRED:
    case D67_PRED:
      pred_directional(dst, stride, above, above_len, left_below, lb_len, top_left, angle, width,
                       height, bit_depth, ief, avail_w, avail_h, hbd);
      return 0;
    case SMOOTH_PRED: case SMOOTH_V_PRED: case SMOOTH_H_PRED: {
      const int32_t below_pred = getp(left_slice, hbd, 0);
      const int32_t right_pred = getp(above, hbd, width - 1);
      const uint8_t *ww = kR1SmWeights + width, *wh = kR1SmWeights + height;
      for (int r = 0; r < height; r++)
        for (int c = 0; c < width; c++) {
          const uint32_t a = (uint32_t)getp(above, hbd, c);
          const uint32_t l = (uint32_t)getp(left_slice, hbd, height - 1 - r);
          uint32_t p;
          if (mode == SMOOTH_PRED)
            p = (wh[r] * a + (256 - wh[r]) * (uint32_t)below_pred + ww[c] * l +
                 (256 - ww[c]) * (uint32_t)right_pred + 256) >> 9;
          else if (mode == SMOOTH_H_PRED)
            p = (ww[c] * l + (256 - ww[c]) * (uint32_t)right_pred + 128) >> 8;
          else
            p = (wh[r] * a + (256 - wh[r]) * (uint32_t)below_pred + 128) >> 8;
          setp(dst, hbd, r * stride + c, (int32_t)p);
        }
      return 0;
    }
    case PAETH_PRED:
      for (int r = 0; r < height; r++)
        for (int c = 0; c < width; c++) {
          const int32_t rl = getp(left_slice, hbd, height - 1 - r), rt = getp(above, hbd, c);
          const int32_t base = rt + rl - top_left;
          const int32_t pl = abs(base - rl), pt = abs(base - rt), ptl = abs(base - top_left);
          setp(dst, hbd, r * stride + c,
               (pl <= pt && pl <= ptl) ? rl : (pt <= ptl ? rt : top_left));
        }
      return 0;
    default:
      return -1;
  }
  /* DC family (also the base of CFL) */
  uint32_t avg;
  if (do_dc == VAR_NONE) {
    avg = 128u << (bit_depth - 8);
  } else if (do_dc == VAR_LEFT) {
    uint32_t s = 0;                     /* pred_dc_left sums left[..] = the whole slice */
    for (int i = 0; i < ls_len; i++) s += (uint32_t)getp(left_slice, hbd, i);
    avg = (s + (uint32_t)(height >> 1)) / (uint32_t)height;
  } else if (do_dc == VAR_TOP) {
    uint32_t s = 0;
    for (int i = 0; i < width; i++) s += (uint32_t)getp(above, hbd, i);
    avg = (s + (uint32_t)(width >> 1)) / (uint32_t)width;
  } else {
    uint32_t s = 0;
    for (int i = 0; i < height; i++) s += (uint32_t)getp(left_slice, hbd, i);
    for (int i = 0; i < width; i++) s += (uint32_t)getp(above, hbd, i);
    const uint32_t len = (uint32_t)(width + height);
    avg = (s + (len >> 1)) / len;
  }
  fill(dst, stride, width, height, (int32_t)avg, hbd);
  if (mode == UV_CFL_PRED && angle != 0) {
    const int32_t smax = (1 << bit_depth) - 1;
    const int32_t a0 = getp(dst, hbd, 0);
    for (int r = 0; r < height; r++)
      for (int c = 0; c < width; c++)
        setp(dst, hbd, r * stride + c,
             clampi(a0 + scaled_luma_q0((int16_t)angle, ac[r * width + c]), 0, smax));
  }
  return 0;
}

/* PredictionMode::predict_intra (predict.rs:205-249): x, y = block position
 * relative to the tile; intra_param: angle_delta (directional) or alpha (CFL). */
int r1o_predict_intra(int mode, int x, int y, void *dst, ptrdiff_t stride, int tx_size,
                      int bit_depth, const int16_t *ac, int angle_delta, int alpha, int ief,
                      const void *edge, int left_len, int above_len, int avail_w, int avail_h,
                      int hbd) {
  const int variant = (x == 0 && y == 0) ? VAR_NONE : (y == 0 ? VAR_LEFT : (x == 0 ? VAR_TOP : VAR_BOTH));
  if (mode == PAETH_PRED)
    mode = variant == VAR_NONE ? DC_PRED : variant == VAR_TOP ? V_PRED
           : variant == VAR_LEFT ? H_PRED : PAETH_PRED;
  else if (mode == UV_CFL_PRED && alpha == 0)
    mode = DC_PRED;
  const int angle = mode == UV_CFL_PRED ? alpha : r1o_intra_mode_to_angle(mode) + angle_delta * 3;
  return r1o_dispatch_predict_intra(mode, variant, dst, stride, tx_size, bit_depth, ac, angle, ief,
                                    edge, left_len, above_len, avail_w, avail_h, hbd);
}

/* pred_cfl_ac (predict.rs:1020-1063): luma -> subsampled, mean-removed Q3 AC.
 * luma: top-left of the luma block; (bw, bh) = chroma plane block size. */
void r1o_pred_cfl_ac(int16_t *ac, const void *luma, ptrdiff_t stride, int bw, int bh,
                     int w_pad, int h_pad, int xdec, int ydec, int hbd) {
  const int max_luma_w = (bw - w_pad * 4) << xdec, max_luma_h = (bh - h_pad * 4) << ydec;
  const int max_luma_x = (max_luma_w > 8 ? max_luma_w : 8) - (1 << xdec);
  const int max_luma_y = (max_luma_h > 8 ? max_luma_h : 8) - (1 << ydec);
  int32_t sum = 0;
  for (int sy = 0; sy < bh; sy++)
    for (int sx = 0; sx < bw; sx++) {
      int ly = sy << ydec, lx = sx << xdec;
      const int y = ly < max_luma_y ? ly : max_luma_y, x = lx < max_luma_x ? lx : max_luma_x;
      int16_t s = (int16_t)getp(luma, hbd, y * stride + x);
      if (xdec) s += (int16_t)getp(luma, hbd, y * stride + x + 1);
      if (ydec) s += (int16_t)(getp(luma, hbd, (y + 1) * stride + x) +
                               getp(luma, hbd, (y + 1) * stride + x + 1));
      s = (int16_t)(s << (3 - xdec - ydec));
      ac[sy * bw + sx] = s;
      sum += s;
    }
  int shift = 0;
  while ((1 << shift) < bw * bh) shift++;
  const int16_t avg = (int16_t)((sum + (1 << (shift - 1))) >> shift);
  for (int i = 0; i < bw * bh; i++) ac[i] = (int16_t)(ac[i] - avg);
}

/* get_intra_edges (partition.rs:639-898).  `tile`: top-left of the tile region
 * in the reconstructed plane (stride in elements); (x, y) = block position in
 * the tile; rect_w/rect_h = visible tile extent (min(rect size, plane size -
 * rect origin)); has_tr / has_bl: the caller's has_top_right / has_bottom_left
 * answers; mode < 0 = None (everything needed).  edge: 257-pixel buffer.
 * Writes init_left / init_above to lens[0..1]. */
void r1o_get_intra_edges(void *edge, int lens[2], const void *tile, ptrdiff_t stride, int x,
                         int y, int rect_w, int rect_h, int tx_size, int bit_depth, int mode,
                         int enable_ief, int angle_delta, int has_tr, int has_bl, int hbd) {
  const int txw = r1o_tx_width(tx_size), txh = r1o_tx_height(tx_size);
  const int32_t base = 128 << (bit_depth - 8);
  int init_left = 0, init_above = 0;
  int needs_left = 1, needs_topleft = 1, needs_top = 1, needs_topright = 1, needs_bottomleft = 1,
      needs_tl_filter = 0;
#define L(i) (2 * MAXTX - 1 - (i))            /* left[2*MAX - 1 - i] */
#define A(i) (2 * MAXTX + 1 + (i))
#define DST(yy, xx) getp(tile, hbd, (ptrdiff_t)(yy) * stride + (xx))
  if (mode >= 0) {
    if (mode == PAETH_PRED)
      mode = (x == 0 && y == 0) ? DC_PRED : (x == 0 ? V_PRED : (y == 0 ? H_PRED : PAETH_PRED));
    const int directional = mode >= V_PRED && mode <= D67_PRED;
    const int p_angle = r1o_intra_mode_to_angle(mode) + angle_delta * 3;
    const int dc_or_cfl = mode == DC_PRED || mode == UV_CFL_PRED;
    needs_left = (!dc_or_cfl || x != 0) || (p_angle > 90 && p_angle != 180);
    needs_topleft = mode == PAETH_PRED || (directional && p_angle != 90 && p_angle != 180);
    needs_top = (!dc_or_cfl || y != 0) || (p_angle != 90 && p_angle < 180);
    needs_topright = directional && p_angle < 90;
    needs_bottomleft = directional && p_angle > 180;
    needs_tl_filter = enable_ief && p_angle > 90 && p_angle < 180;
  }
  if (needs_left) {
    const int th = y + txh > rect_h ? rect_h - y : txh;
    if (x != 0) {
      for (int i = 0; i < th; i++) setp(edge, hbd, L(i), DST(y + i, x - 1));
      if (th < txh) {
        const int32_t v = DST(y + th - 1, x - 1);
        for (int i = th; i < txh; i++) setp(edge, hbd, L(i), v);
      }
    } else {
      const int32_t v = y != 0 ? DST(y - 1, 0) : base + 1;
      for (int i = 0; i < txh; i++) setp(edge, hbd, L(i), v);
    }
    init_left += txh;
  }
  if (needs_top) {
    const int tw = x + txw > rect_w ? rect_w - x : txw;
    if (y != 0) {
      for (int i = 0; i < tw; i++) setp(edge, hbd, A(i), DST(y - 1, x + i));
      if (tw < txw) {
        const int32_t v = DST(y - 1, x + tw - 1);
        for (int i = tw; i < txw; i++) setp(edge, hbd, A(i), v);
      }
    } else {
      const int32_t v = x != 0 ? DST(0, x - 1) : base - 1;
      for (int i = 0; i < txw; i++) setp(edge, hbd, A(i), v);
    }
    init_above += txw;
  }
  if (needs_topright) {
    int num_avail = 0;
    if (y != 0 && has_tr) {
      num_avail = rect_w - x - txw;
      if (num_avail > txw) num_avail = txw;
      if (num_avail < 0) num_avail = 0;
    }
    for (int i = 0; i < num_avail; i++) setp(edge, hbd, A(txw + i), DST(y - 1, x + txw + i));
    if (num_avail < txh) {
      const int32_t v = getp(edge, hbd, A(txw + num_avail - 1));
      for (int i = txw + num_avail; i < txw + txh; i++) setp(edge, hbd, A(i), v);
    }
    init_above += txh;
  }
  if (needs_bottomleft) {
    int num_avail = 0;
    if (x != 0 && has_bl) {
      num_avail = rect_h - y - txh;
      if (num_avail > txh) num_avail = txh;
      if (num_avail < 0) num_avail = 0;
    }
    for (int i = 0; i < num_avail; i++) setp(edge, hbd, L(txh + i), DST(y + txh + i, x - 1));
    if (num_avail < txw) {
      const int32_t v = getp(edge, hbd, 2 * MAXTX - txh - num_avail);
      for (int i = 2 * MAXTX - txh - txw; i < 2 * MAXTX - txh - num_avail; i++)
        setp(edge, hbd, i, v);
    }
    init_left += txw;
  }
  if (needs_topleft) {
    int32_t tl = (x == 0 && y == 0) ? base : (y == 0 ? DST(0, x - 1) : (x == 0 ? DST(y - 1, 0) : DST(y - 1, x - 1)));
    if (needs_tl_filter && txw + txh >= 24) {
      const uint32_t l = (uint32_t)getp(edge, hbd, 2 * MAXTX - 1), a = (uint32_t)getp(edge, hbd, A(0));
      tl = (int32_t)((l * 5 + (uint32_t)tl * 6 + a * 5 + 8) >> 4);
    }
    setp(edge, hbd, 2 * MAXTX, tl);
  } else {
    setp(edge, hbd, 2 * MAXTX, base);
  }
  lens[0] = init_left;
  lens[1] = init_above;
#undef L
#undef A
#undef DST
}
