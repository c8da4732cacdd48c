/*
 * lrf.c -- CPU oracle: loop restoration, self-guided (SGRPROJ) stripe filter
 * (SURVEY.md 8f "N3", last stage of the post-filter chain).
 * TEST INFRASTRUCTURE ONLY (see r1_oracle.h).
 *
 * Restates src/lrf.rs of the reference:
 *   SGRPROJ_PARAMS_S 55-72, sgrproj_box_ab_internal 176-201 (+ _r1 / _r2),
 *   sgrproj_box_f_r0 / _r1 / _r2 242-341, sgrproj_sum_finish 345-363,
 *   get_integral_square 368-380, VertPaddedIter 402-476, HorzPaddedIter 492-524,
 *   setup_integral_image 530-627, sgrproj_stripe_filter 630-830,
 *   RestorationPlane::restoration_unit_index_by_stripe 1297-1307,
 *   RestorationState::lrf_filter_frame 1482-1585 (the Sgrproj arm; the encoder
 *   never selects Wiener: src/rdo.rs:2508 `unreachable!() // coming soon`),
 *   sgrproj_solve 847-1096 with the integral image rdo_loop_decision builds
 *   for it (src/rdo.rs:2651-2676: the unit hard-clipped at its right / bottom
 *   edge, monolithic -- no stripes).
 *
 * Pinning: the reference has no vectors for this file; the filter is a
 * normative AV1 decoder process (spec 7.17), so tests/golden/gen_lrf_golden.py
 * holds an independent model (direct box sums on an explicitly padded stripe,
 * no integral images, no rolling row buffers) whose frames this file matched.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "r1_oracle.h"

static const uint32_t SGR_S[16][2] = {
  { 140, 3236 }, { 112, 2158 }, { 93, 1618 }, { 80, 1438 }, { 70, 1295 }, { 58, 1177 },
  { 47, 1079 },  { 37, 996 },   { 30, 925 },  { 25, 863 },  { 0, 2589 },  { 0, 1618 },
  { 0, 1177 },   { 0, 925 },    { 56, 0 },    { 22, 0 } };

#define IMG_MAX (256 + 128)          /* STRIPE_IMAGE_MAX */
#define IMG_STRIDE (IMG_MAX + 6 + 2) /* STRIPE_IMAGE_STRIDE */
#define IMG_HEIGHT (64 + 6 + 2)

static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
static int imin(int a, int b) { return a < b ? a : b; }

static uint32_t px(const r1o_plane *p, int x, int y) {
  const size_t i = (size_t)(p->yorigin + y) * p->stride + p->xorigin + x;
  return p->bytes_per_px == 1 ? ((const uint8_t *)p->data)[i] : ((const uint16_t *)p->data)[i];
}

/* setup_integral_image: (x0, y0) = the unit's origin in the plane; crop_w /
 * crop_h are ABSOLUTE here (the reference passes them relative to the slice).
 * The reference decides what lies left of / above the unit from the slice's position IN ITS PLANE
 * (lrf.rs: `cdeffed.x == 0`, `clamp(y, 0, crop - 1)`).  For the frame filter that plane is the
 * frame: left_real = x0 != 0, top_min = 0.  For the restoration search it is rdo_loop_decision's
 * scratch copy of the AREA being decided (rdo.rs:2277-2296: no padding, origin = the area's): a
 * unit in the area's first column / row sees nothing left of / above itself wherever the area
 * lies in the frame -- left_real / top_min come from the caller's edge flags there. */
static void setup_integral_image(uint32_t *ii, uint32_t *sq, int crop_w, int crop_h, int stripe_w,
                                 int stripe_h, const r1o_plane *cdeffed, const r1o_plane *deblocked,
                                 int x0, int y0, int left_real, int top_min) {
  const int left_w = 4, right_w = 3;
  const int left_uniques = left_real ? left_w : 0;
  const int right_uniques = imin(right_w, (crop_w - x0) - stripe_w);
  const int h2 = stripe_h + (stripe_h & 1);
  const int rows_above = 4, rows_below = 2;
  const int stripe_begin = y0, stripe_end = y0 + h2;
  const int nrows = rows_above + h2 + rows_below, ncols = left_w + stripe_w + right_w;
  for (int j = 0; j < nrows; j++) {
    const int y = y0 - rows_above + j;
    const int cy = clampi(y, top_min, crop_h - 1);
    const int ly = clampi(cy, stripe_begin - 2, stripe_end + 1);
    const r1o_plane *src = (ly >= stripe_begin && ly < stripe_end) ? cdeffed : deblocked;
    uint32_t sum = 0, sqs = 0;
    for (int i = 0; i < ncols; i++) {
      /* HorzPaddedIter over row[x0 - left_uniques ..][..row_uniques] from start_index */
      const int idx = clampi((left_real ? 0 : -left_w) + i, 0, left_uniques + stripe_w + right_uniques - 1);
      const uint32_t cur = px(src, x0 - left_uniques + idx, ly);
      sum += cur;
      sqs += cur * cur;
      ii[j * IMG_STRIDE + i] = sum + (j ? ii[(j - 1) * IMG_STRIDE + i] : 0);
      sq[j * IMG_STRIDE + i] = sqs + (j ? sq[(j - 1) * IMG_STRIDE + i] : 0);
    }
  }
}

static uint32_t integral_square(const uint32_t *ii, int x, int y, int size) {
  return ii[y * IMG_STRIDE + x] + ii[(y + size) * IMG_STRIDE + x + size] - ii[(y + size) * IMG_STRIDE + x] -
         ii[y * IMG_STRIDE + x + size];
}

static void sum_finish(uint32_t ssq, uint32_t sum, uint32_t n, uint32_t one_over_n, uint32_t s, int bd,
                       uint32_t *a_out, uint32_t *b_out) {
  const int bdm8 = bd - 8;
  const uint32_t scaled_ssq = (ssq + ((1u << (2 * bdm8)) >> 1)) >> (2 * bdm8);
  const uint32_t scaled_sum = (sum + ((1u << bdm8) >> 1)) >> bdm8;
  const uint32_t t = scaled_ssq * n, u = scaled_sum * scaled_sum;
  const uint32_t p = t > u ? t - u : 0; /* saturating_sub */
  const uint32_t z = (p * s + (1u << 19)) >> 20;
  const uint32_t a = z >= 255 ? 256 : (z == 0 ? 1 : ((z << 8) + z / 2) / (z + 1));
  const uint32_t b = ((1u << 8) - a) * sum * one_over_n;
  *a_out = a;
  *b_out = (b + (1u << 11)) >> 12;
}

/* sgrproj_box_ab_r{1,2}: one row of stripe_w + 2 (a, b) pairs at integral row y */
static void box_ab(int r, uint32_t *af, uint32_t *bf, const uint32_t *ii, const uint32_t *sq, int y,
                   int stripe_w, uint32_t s, int bd) {
  const int d = 2 * r + 1;
  for (int x = 0; x < stripe_w + 2; x++)
    sum_finish(integral_square(sq, x, y, d), integral_square(ii, x, y, d), (uint32_t)(d * d),
               r == 1 ? 455 : 164, s, bd, &af[x], &bf[x]);
}

static void stripe_filter(int set, const int8_t *xqd, int bd, const uint32_t *ii, const uint32_t *sq,
                          const r1o_plane *cdeffed, const r1o_plane *out, int x0, int y0, int stripe_w,
                          int stripe_h) {
  static uint32_t a_r2[2][IMG_MAX + 2], b_r2[2][IMG_MAX + 2], f_r2_0[IMG_MAX], f_r2_1[IMG_MAX];
  static uint32_t a_r1[3][IMG_MAX + 2], b_r1[3][IMG_MAX + 2], f_r1[IMG_MAX];
#pragma omp threadprivate(a_r2, b_r2, f_r2_0, f_r2_1, a_r1, b_r1, f_r1)
  const uint32_t s_r2 = SGR_S[set][0], s_r1 = SGR_S[set][1];
  /* r = 1 works on the integral images moved by (1, 1) */
  const uint32_t *ii1 = ii + IMG_STRIDE + 1, *sq1 = sq + IMG_STRIDE + 1;
  if (s_r2 > 0) box_ab(2, a_r2[0], b_r2[0], ii, sq, 0, stripe_w, s_r2, bd);
  if (s_r1 > 0) {
    box_ab(1, a_r1[0], b_r1[0], ii1, sq1, 0, stripe_w, s_r1, bd);
    box_ab(1, a_r1[1], b_r1[1], ii1, sq1, 1, stripe_w, s_r1, bd);
  }
  const int w0 = xqd[0], w1 = xqd[1], w2 = 128 - w0 - w1;
  for (int y = 0; y < stripe_h; y += 2) {
    const uint32_t *fr2[2];
    if (s_r2 > 0) {
      const int k0 = (y / 2) % 2, k1 = (y / 2 + 1) % 2;
      box_ab(2, a_r2[k1], b_r2[k1], ii, sq, y + 2, stripe_w, s_r2, bd);
      for (int x = 0; x < stripe_w; x++) {
        const uint32_t a = 5 * (a_r2[k0][x] + a_r2[k0][x + 2]) + 6 * a_r2[k0][x + 1];
        const uint32_t b = 5 * (b_r2[k0][x] + b_r2[k0][x + 2]) + 6 * b_r2[k0][x + 1];
        const uint32_t ao = 5 * (a_r2[k1][x] + a_r2[k1][x + 2]) + 6 * a_r2[k1][x + 1];
        const uint32_t bo = 5 * (b_r2[k1][x] + b_r2[k1][x + 2]) + 6 * b_r2[k1][x + 1];
        const uint32_t v = (a + ao) * px(cdeffed, x0 + x, y0 + y) + b + bo;
        f_r2_0[x] = (v + (1u << 8)) >> 9;
        /* the reference reads cdeffed.row(y + 1) even when that row is past the stripe */
        const uint32_t vo = ao * px(cdeffed, x0 + x, y0 + y + 1) + bo;
        f_r2_1[x] = (vo + (1u << 7)) >> 8;
      }
      fr2[0] = f_r2_0;
      fr2[1] = f_r2_1;
    } else {
      for (int x = 0; x < stripe_w; x++) f_r2_0[x] = px(cdeffed, x0 + x, y0 + y) << 4;
      fr2[0] = fr2[1] = f_r2_0;
    }
    for (int dy = 0; dy < imin(2, stripe_h - y); dy++) {
      const int yy = y + dy;
      if (s_r1 > 0) {
        box_ab(1, a_r1[(yy + 2) % 3], b_r1[(yy + 2) % 3], ii1, sq1, yy + 2, stripe_w, s_r1, bd);
        const uint32_t *a0 = a_r1[yy % 3], *a1 = a_r1[(yy + 1) % 3], *a2 = a_r1[(yy + 2) % 3];
        const uint32_t *b0 = b_r1[yy % 3], *b1 = b_r1[(yy + 1) % 3], *b2 = b_r1[(yy + 2) % 3];
        for (int x = 0; x < stripe_w; x++) {
          const uint32_t a = 3 * (a0[x] + a2[x] + a0[x + 2] + a2[x + 2]) +
                             4 * (a1[x] + a0[x + 1] + a1[x + 1] + a2[x + 1] + a1[x + 2]);
          const uint32_t b = 3 * (b0[x] + b2[x] + b0[x + 2] + b2[x + 2]) +
                             4 * (b1[x] + b0[x + 1] + b1[x + 1] + b2[x + 1] + b1[x + 2]);
          const uint32_t v = a * px(cdeffed, x0 + x, y0 + yy) + b;
          f_r1[x] = (v + (1u << 8)) >> 9;
        }
      } else {
        for (int x = 0; x < stripe_w; x++) f_r1[x] = px(cdeffed, x0 + x, y0 + yy) << 4;
      }
      for (int x = 0; x < stripe_w; x++) {
        const int32_t u = (int32_t)px(cdeffed, x0 + x, y0 + yy) << 4;
        const int32_t v = w0 * (int32_t)fr2[dy][x] + w1 * u + w2 * (int32_t)f_r1[x];
        const int32_t s = (v + (1 << 10)) >> 11;
        const int32_t o = clampi(s, 0, (1 << bd) - 1);
        const size_t i = (size_t)(out->yorigin + y0 + yy) * out->stride + out->xorigin + x0 + x;
        if (out->bytes_per_px == 1) ((uint8_t *)out->data)[i] = (uint8_t)o;
        else ((uint16_t *)out->data)[i] = (uint16_t)o;
      }
    }
  }
}

/* lrf_filter_frame for one plane: crop_w / crop_h in pixels of THIS plane,
 * frame_h the luma frame height (stripe count), units: unit_rows x unit_cols */
int r1o_lrf_filter_plane(const r1o_plane *cdeffed, const r1o_plane *deblocked, const r1o_plane *out,
                         int ydec, int crop_w, int crop_h, int frame_h, int unit_size, int unit_cols,
                         int unit_rows, int stripe_height, const r1o_lrf_unit *units, int bd) {
  if (unit_size > 256 || unit_size < 32) return -1;
  const int stripe_n = (frame_h + 7) / 64 + 1;
  uint32_t *ii = (uint32_t *)calloc((size_t)IMG_STRIDE * IMG_HEIGHT * 2, sizeof(uint32_t));
  uint32_t *sq = ii + IMG_STRIDE * IMG_HEIGHT;
  for (int si = 0; si < stripe_n; si++) {
    int y0, sz;
    if (si == 0) {
      y0 = 0;
      sz = (64 - 8) >> ydec;
    } else {
      y0 = (si * 64 - 8) >> ydec;
      sz = imin(64 >> ydec, crop_h - y0);
    }
    if (sz <= 0) continue; /* the reference's usize arithmetic never gets here on valid sizes */
    for (int rux = 0; rux < unit_cols; rux++) {
      const int x0 = rux * unit_size;
      const int size = rux == unit_cols - 1 ? crop_w - x0 : unit_size;
      const int ruy = imin(si * stripe_height / unit_size, unit_rows - 1);
      const r1o_lrf_unit *u = &units[ruy * unit_cols + imin(rux, unit_cols - 1)];
      if (u->filter != 3 || size <= 0) continue; /* RESTORE_SGRPROJ */
      if (size > IMG_MAX) { free(ii); return -1; }
      setup_integral_image(ii, sq, crop_w, crop_h, size, sz, cdeffed, deblocked, x0, y0, x0 != 0, 0);
      stripe_filter(u->set, u->xqd, bd, ii, sq, cdeffed, out, x0, y0, size, sz);
    }
  }
  free(ii);
  return 0;
}

/* sgrproj_solve for one restoration unit of the RDO (src/rdo.rs:2651-2676):
 * integral image of the unit with crop = the unit's own right / bottom edge,
 * cdeffed == deblocked (monolithic), then the least-squares projection
 * weights.  input: the source plane (same coordinates). */
void r1o_sgrproj_solve(const r1o_plane *cdeffed, const r1o_plane *input, int x0, int y0, int w, int h,
                       int set, int edges, int bd, int8_t *xqd_out) {
  const int rows = h + (h & 1) + 6;
  uint32_t *ii = (uint32_t *)calloc((size_t)IMG_STRIDE * rows * 2, sizeof(uint32_t));
  uint32_t *sq = ii + (size_t)IMG_STRIDE * rows;
  setup_integral_image(ii, sq, x0 + w, y0 + h, w, h, cdeffed, cdeffed, x0, y0, (edges & R1O_SGR_EDGE_LEFT) && x0 > 0,
                       (edges & R1O_SGR_EDGE_ABOVE) ? 0 : y0);
  static uint32_t a_r2[2][IMG_MAX + 2], b_r2[2][IMG_MAX + 2], f_r2_0[IMG_MAX], f_r2_1[IMG_MAX];
  static uint32_t a_r1[3][IMG_MAX + 2], b_r1[3][IMG_MAX + 2], f_r1[IMG_MAX];
#pragma omp threadprivate(a_r2, b_r2, f_r2_0, f_r2_1, a_r1, b_r1, f_r1)
  const uint32_t s_r2 = SGR_S[set][0], s_r1 = SGR_S[set][1];
  const uint32_t *ii1 = ii + IMG_STRIDE + 1, *sq1 = sq + IMG_STRIDE + 1;
  double hm[2][2] = { { 0, 0 }, { 0, 0 } }, c[2] = { 0, 0 };
  if (s_r2 > 0) box_ab(2, a_r2[0], b_r2[0], ii, sq, 0, w, s_r2, bd);
  if (s_r1 > 0) {
    box_ab(1, a_r1[0], b_r1[0], ii1, sq1, 0, w, s_r1, bd);
    box_ab(1, a_r1[1], b_r1[1], ii1, sq1, 1, w, s_r1, bd);
  }
  for (int y = 0; y < h; y += 2) {
    const uint32_t *fr2[2];
    if (s_r2 > 0) {
      const int k0 = (y / 2) % 2, k1 = (y / 2 + 1) % 2;
      box_ab(2, a_r2[k1], b_r2[k1], ii, sq, y + 2, w, s_r2, bd);
      for (int x = 0; x < w; x++) {
        const uint32_t a = 5 * (a_r2[k0][x] + a_r2[k0][x + 2]) + 6 * a_r2[k0][x + 1];
        const uint32_t b = 5 * (b_r2[k0][x] + b_r2[k0][x + 2]) + 6 * b_r2[k0][x + 1];
        const uint32_t ao = 5 * (a_r2[k1][x] + a_r2[k1][x + 2]) + 6 * a_r2[k1][x + 1];
        const uint32_t bo = 5 * (b_r2[k1][x] + b_r2[k1][x + 2]) + 6 * b_r2[k1][x + 1];
        f_r2_0[x] = ((a + ao) * px(cdeffed, x0 + x, y0 + y) + b + bo + (1u << 8)) >> 9;
        f_r2_1[x] = (ao * px(cdeffed, x0 + x, y0 + y + 1) + bo + (1u << 7)) >> 8;
      }
      fr2[0] = f_r2_0;
      fr2[1] = f_r2_1;
    } else {
      for (int x = 0; x < w; x++) f_r2_0[x] = px(cdeffed, x0 + x, y0 + y) << 4;
      fr2[0] = fr2[1] = f_r2_0;
    }
    for (int dy = 0; dy < imin(2, h - y); dy++) {
      const int yy = y + dy;
      if (s_r1 > 0) {
        box_ab(1, a_r1[(yy + 2) % 3], b_r1[(yy + 2) % 3], ii1, sq1, yy + 2, w, s_r1, bd);
        const uint32_t *a0 = a_r1[yy % 3], *a1 = a_r1[(yy + 1) % 3], *a2 = a_r1[(yy + 2) % 3];
        const uint32_t *b0 = b_r1[yy % 3], *b1 = b_r1[(yy + 1) % 3], *b2 = b_r1[(yy + 2) % 3];
        for (int x = 0; x < w; x++) {
          const uint32_t a = 3 * (a0[x] + a2[x] + a0[x + 2] + a2[x + 2]) +
                             4 * (a1[x] + a0[x + 1] + a1[x + 1] + a2[x + 1] + a1[x + 2]);
          const uint32_t b = 3 * (b0[x] + b2[x] + b0[x + 2] + b2[x + 2]) +
                             4 * (b1[x] + b0[x + 1] + b1[x + 1] + b2[x + 1] + b1[x + 2]);
          f_r1[x] = (a * px(cdeffed, x0 + x, y0 + yy) + b + (1u << 8)) >> 9;
        }
      } else {
        for (int x = 0; x < w; x++) f_r1[x] = px(cdeffed, x0 + x, y0 + yy) << 4;
      }
      /* process_line: i64 sums of the line, then into the f64 accumulators */
      int64_t l00 = 0, l11 = 0, l01 = 0, lc0 = 0, lc1 = 0;
      for (int x = 0; x < w; x++) {
        const int32_t u = (int32_t)px(cdeffed, x0 + x, y0 + yy) << 4;
        const int64_t sv = ((int32_t)px(input, x0 + x, y0 + yy) << 4) - u;
        const int64_t f2 = (int32_t)fr2[dy][x] - u, f1 = (int32_t)f_r1[x] - u;
        l00 += f2 * f2; l11 += f1 * f1; l01 += f1 * f2; lc0 += f2 * sv; lc1 += f1 * sv;
      }
      hm[0][0] += (double)l00; hm[1][1] += (double)l11; hm[0][1] += (double)l01;
      c[0] += (double)lc0; c[1] += (double)lc1;
    }
  }
  free(ii);
  const double n = (double)w * (double)h;
  hm[0][0] /= n; hm[0][1] /= n; hm[1][1] /= n;
  hm[1][0] = hm[0][1];
  c[0] *= 128.0 / n;
  c[1] *= 128.0 / n;
  double xq0 = 0, xq1 = 0;
  if (s_r2 == 0) {
    if (hm[1][1] != 0.) xq1 = round(c[1] / hm[1][1]);
  } else if (s_r1 == 0) {
    if (hm[0][0] != 0.) xq0 = round(c[0] / hm[0][0]);
  } else {
    const double det = fma(hm[0][0], hm[1][1], -hm[0][1] * hm[1][0]);
    if (det != 0.) {
      const double div1 = fma(hm[1][1], c[0], -hm[0][1] * c[1]);
      const double div2 = fma(hm[0][0], c[1], -hm[1][0] * c[0]);
      xq0 = round(div1 / det);
      xq1 = round(div2 / det);
    }
  }
  /* `as i32` saturates, NaN -> 0 */
  const int q0 = xq0 != xq0 ? 0 : (xq0 > 2147483647. ? 2147483647 : (xq0 < -2147483648. ? (-2147483647 - 1) : (int)xq0));
  const int q1 = xq1 != xq1 ? 0 : (xq1 > 2147483647. ? 2147483647 : (xq1 < -2147483648. ? (-2147483647 - 1) : (int)xq1));
  const int xqd0 = clampi(q0, -96, 31);
  const int64_t t = (int64_t)128 - xqd0 - q1; /* i32 arithmetic in the reference; wraps only for absurd q1 */
  const int xqd1 = (int)(t < -32 ? -32 : (t > 95 ? 95 : t));
  xqd_out[0] = (int8_t)xqd0;
  xqd_out[1] = (int8_t)xqd1;
}

/* rdo_loop_plane_error's sum over the blocks of a rectangle (src/rdo.rs:2027-2093), before
 * `* fi.dist_scale[pli]`: `test` = the pixels under test, pixel (x0, y0) of the plane at test[0],
 * row stride tstride pixels.  Per 8x8-luma block: cdef_dist_kernel * bias (luma) / sse_wxh with
 * |_, _| bias on (8 >> xdec) x (8 >> ydec) pixels (chroma). */
uint64_t r1o_loop_plane_error_rect(const r1o_plane *src, const void *test, int tstride, int x0, int y0,
                                   int w, int h, int is_chroma, int xdec, int ydec, const uint32_t *scales,
                                   int scale_stride, int bd) {
  const int bw = 8 >> xdec, bh = 8 >> ydec, hbd = src->bytes_per_px == 2;
  const size_t bpp = (size_t)src->bytes_per_px;
  uint64_t plane_sum = 0;
  for (int by = 0; by < h / bh; by++)
    for (int bx = 0; bx < w / bw; bx++) {
      const int px_ = x0 + bx * bw, py_ = y0 + by * bh;
      const uint8_t *spx = (const uint8_t *)src->data +
                           ((size_t)(src->yorigin + py_) * src->stride + src->xorigin + px_) * bpp;
      const uint8_t *tpx = (const uint8_t *)test + ((size_t)by * bh * tstride + (size_t)bx * bw) * bpp;
      const uint32_t bias = scales ? scales[(size_t)((py_ << ydec) >> 3) * scale_stride + ((px_ << xdec) >> 3)]
                                   : (1u << 14);
      if (!is_chroma) {
        const uint64_t raw = r1o_cdef_dist_kernel(spx, src->stride, tpx, tstride, 8, 8, bd, hbd);
        plane_sum += ((uint64_t)bias * raw + 8192) >> 14;               /* RawDistortion * bias */
      } else {
        uint32_t cell[4] = {bias, bias, bias, bias};                      /* sse_wxh: |_, _| bias */
        plane_sum += r1o_get_weighted_sse(spx, src->stride, tpx, tstride, cell, 2, bw, bh, hbd);
      }
    }
  return plane_sum;
}

/* A restoration unit's CURRENT choice applied to one rectangle of `plane`, as the later passes of
 * rdo_loop_decision's CDEF leg do to the superblock under trial (src/rdo.rs:2458-2489):
 * setup_integral_image with crop = stripe = the rectangle (hard-clipped right and below; left /
 * above per `edges`, read from the same plane -- the area's working copy), sgrproj_stripe_filter
 * with (set, xqd).  out: w * h pixels, dense. */
int r1o_sgr_filter_rect(const r1o_plane *plane, int x0, int y0, int w, int h, int set, const int8_t *xqd,
                        int edges, int bd, void *out_px) {
  if (w <= 0 || h <= 0 || w > IMG_MAX || set < 0 || set > 15) return -1;
  const int rows = h + (h & 1) + 6;
  uint32_t *ii = (uint32_t *)calloc((size_t)IMG_STRIDE * rows * 2, sizeof(uint32_t));
  uint32_t *sq = ii + (size_t)IMG_STRIDE * rows;
  setup_integral_image(ii, sq, x0 + w, y0 + h, w, h, plane, plane, x0, y0, (edges & R1O_SGR_EDGE_LEFT) && x0 > 0,
                       (edges & R1O_SGR_EDGE_ABOVE) ? 0 : y0);
  r1o_plane out = *plane;
  out.data = out_px;
  out.stride = w;
  out.xorigin = -x0;
  out.yorigin = -y0;
  stripe_filter(set, xqd, bd, ii, sq, plane, &out, x0, y0, w, h);
  free(ii);
  return 0;
}

/* The restoration leg of rdo_loop_decision for ONE plane, one (restoration unit, parameter set)
 * pair -- everything but the entropy coder's rate (src/rdo.rs:2575-2763):
 *   set < 16: sgrproj_solve on the unit (above), then the unit filtered with the weights it
 *             returned -- sgrproj_stripe_filter on the unit's OWN integral image
 *             (setup_integral_image with crop = the unit: "hard-clipping to the superblock
 *             boundary", rdo.rs:2651-2666, 2688-2702), the whole unit as one stripe;
 *   set = 255: the "no filter option" (rdo.rs:2617-2643): the unit of lrf_in as it is;
 *   err = rdo_loop_plane_error (rdo.rs:2027-2093) of that against the source: over the
 *         8x8-luma blocks of the unit  cdef_dist_kernel * bias (luma)  /  sse_wxh with |_, _| bias
 *         on (8 >> xdec) x (8 >> ydec) pixels (chroma), bias = the DistortionScale of the block's
 *         8x8 luma position, summed, * fi.dist_scale[pli].
 * edges: R1O_SGR_EDGE_LEFT / R1O_SGR_EDGE_ABOVE -- whether the 4 columns left of / the 2 rows above
 * the unit are pixels the filter sees: they are when the unit is not in the first unit column /
 * row of the AREA rdo_loop_decision is deciding (its scratch copy has no pixels outside the area;
 * found by executing rdo_loop_decision itself, tests/golden/gen_loop_decision_ref.py).
 * (x0, y0, w, h): the unit in pixels of THIS plane, w % (8 >> xdec) == 0 and h % (8 >> ydec) == 0
 * (the visible frame a multiple of 8 luma pixels: otherwise the reference's last blocks read the
 * working copy beyond what the filter wrote).  scales: one per 8x8 luma block of the frame (NULL:
 * the default scale).  Returns -1 on geometry it does not take. */
int r1o_lrf_search_unit(const r1o_plane *lrf_in, const r1o_plane *src, int x0, int y0, int w, int h,
                        int set, int edges, int is_chroma, int xdec, int ydec, const uint32_t *scales,
                        int scale_stride, uint32_t dist_scale, int bd, int8_t *xqd_out,
                        uint64_t *err_out) {
  const int bw = 8 >> xdec, bh = 8 >> ydec;
  if (w <= 0 || h <= 0 || w > IMG_MAX || w % bw || h % bh || (set != 255 && (set < 0 || set > 15))) return -1;
  if (!is_chroma && (xdec || ydec)) return -1;
  const size_t bpp = (size_t)lrf_in->bytes_per_px;
  uint8_t *tmp = (uint8_t *)malloc((size_t)w * h * bpp);
  xqd_out[0] = xqd_out[1] = 0;
  if (set == 255) {
    for (int y = 0; y < h; y++)
      memcpy(tmp + (size_t)y * w * bpp,
             (const uint8_t *)lrf_in->data +
                 ((size_t)(lrf_in->yorigin + y0 + y) * lrf_in->stride + lrf_in->xorigin + x0) * bpp,
             (size_t)w * bpp);
  } else {
    r1o_sgrproj_solve(lrf_in, src, x0, y0, w, h, set, edges, bd, xqd_out);
    const int rows = h + (h & 1) + 6;
    uint32_t *ii = (uint32_t *)calloc((size_t)IMG_STRIDE * rows * 2, sizeof(uint32_t));
    uint32_t *sq = ii + (size_t)IMG_STRIDE * rows;
    setup_integral_image(ii, sq, x0 + w, y0 + h, w, h, lrf_in, lrf_in, x0, y0, (edges & R1O_SGR_EDGE_LEFT) && x0 > 0,
                         (edges & R1O_SGR_EDGE_ABOVE) ? 0 : y0);
    /* a plane whose pixel (x0, y0) is tmp[0] */
    r1o_plane out = *lrf_in;
    out.data = tmp;
    out.stride = w;
    out.xorigin = -x0;
    out.yorigin = -y0;
    stripe_filter(set, xqd_out, bd, ii, sq, lrf_in, &out, x0, y0, w, h);
    free(ii);
  }
  const uint64_t plane_sum = r1o_loop_plane_error_rect(src, tmp, w, x0, y0, w, h, is_chroma, xdec, ydec, scales,
                                                       scale_stride, bd);
  free(tmp);
  *err_out = ((uint64_t)dist_scale * plane_sum + 8192) >> 14;            /* Distortion * dist_scale */
  return 0;
}

