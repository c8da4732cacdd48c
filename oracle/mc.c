/*
 * oracle/mc.c -- TEST INFRASTRUCTURE (see r1_oracle.h).
 * Scalar restatement of the reference's motion-compensation kernels:
 *   SUBPEL_FILTERS      src/mc.rs:110-219 (AV1 spec interpolation filters)
 *   get_filter          src/mc.rs:238-247
 *   put_8tap            src/mc.rs:250-353
 *   prep_8tap           src/mc.rs:360-451 (PREP_BIAS 355-357)
 *   mc_avg              src/mc.rs:454-479
 * round_shift is v_frame::math::round_shift: (v + (1 << b >> 1)) >> b.
 */
#include "r1_oracle.h"

/* AV1 sub-pel interpolation kernels, 1/16-pel phases.  Sets: 0 regular,
 * 1 smooth, 2 sharp, 3 bilinear, 4 regular (4-tap, for dim <= 4),
 * 5 smooth (4-tap). */
static const int16_t FILTERS[6][16][8] = {
    {{0, 0, 0, 128, 0, 0, 0, 0},
     {0, 2, -6, 126, 8, -2, 0, 0},
     {0, 2, -10, 122, 18, -4, 0, 0},
     {0, 2, -12, 116, 28, -8, 2, 0},
     {0, 2, -14, 110, 38, -10, 2, 0},
     {0, 2, -14, 102, 48, -12, 2, 0},
     {0, 2, -16, 94, 58, -12, 2, 0},
     {0, 2, -14, 84, 66, -12, 2, 0},
     {0, 2, -14, 76, 76, -14, 2, 0},
     {0, 2, -12, 66, 84, -14, 2, 0},
     {0, 2, -12, 58, 94, -16, 2, 0},
     {0, 2, -12, 48, 102, -14, 2, 0},
     {0, 2, -10, 38, 110, -14, 2, 0},
     {0, 2, -8, 28, 116, -12, 2, 0},
     {0, 0, -4, 18, 122, -10, 2, 0},
     {0, 0, -2, 8, 126, -6, 2, 0}},
    {{0, 0, 0, 128, 0, 0, 0, 0},
     {0, 2, 28, 62, 34, 2, 0, 0},
     {0, 0, 26, 62, 36, 4, 0, 0},
     {0, 0, 22, 62, 40, 4, 0, 0},
     {0, 0, 20, 60, 42, 6, 0, 0},
     {0, 0, 18, 58, 44, 8, 0, 0},
     {0, 0, 16, 56, 46, 10, 0, 0},
     {0, -2, 16, 54, 48, 12, 0, 0},
     {0, -2, 14, 52, 52, 14, -2, 0},
     {0, 0, 12, 48, 54, 16, -2, 0},
     {0, 0, 10, 46, 56, 16, 0, 0},
     {0, 0, 8, 44, 58, 18, 0, 0},
     {0, 0, 6, 42, 60, 20, 0, 0},
     {0, 0, 4, 40, 62, 22, 0, 0},
     {0, 0, 4, 36, 62, 26, 0, 0},
     {0, 0, 2, 34, 62, 28, 2, 0}},
    {{0, 0, 0, 128, 0, 0, 0, 0},
     {-2, 2, -6, 126, 8, -2, 2, 0},
     {-2, 6, -12, 124, 16, -6, 4, -2},
     {-2, 8, -18, 120, 26, -10, 6, -2},
     {-4, 10, -22, 116, 38, -14, 6, -2},
     {-4, 10, -22, 108, 48, -18, 8, -2},
     {-4, 10, -24, 100, 60, -20, 8, -2},
     {-4, 10, -24, 90, 70, -22, 10, -2},
     {-4, 12, -24, 80, 80, -24, 12, -4},
     {-2, 10, -22, 70, 90, -24, 10, -4},
     {-2, 8, -20, 60, 100, -24, 10, -4},
     {-2, 8, -18, 48, 108, -22, 10, -4},
     {-2, 6, -14, 38, 116, -22, 10, -4},
     {-2, 6, -10, 26, 120, -18, 8, -2},
     {-2, 4, -6, 16, 124, -12, 6, -2},
     {0, 2, -2, 8, 126, -6, 2, -2}},
    {{0, 0, 0, 128, 0, 0, 0, 0},
     {0, 0, 0, 120, 8, 0, 0, 0},
     {0, 0, 0, 112, 16, 0, 0, 0},
     {0, 0, 0, 104, 24, 0, 0, 0},
     {0, 0, 0, 96, 32, 0, 0, 0},
     {0, 0, 0, 88, 40, 0, 0, 0},
     {0, 0, 0, 80, 48, 0, 0, 0},
     {0, 0, 0, 72, 56, 0, 0, 0},
     {0, 0, 0, 64, 64, 0, 0, 0},
     {0, 0, 0, 56, 72, 0, 0, 0},
     {0, 0, 0, 48, 80, 0, 0, 0},
     {0, 0, 0, 40, 88, 0, 0, 0},
     {0, 0, 0, 32, 96, 0, 0, 0},
     {0, 0, 0, 24, 104, 0, 0, 0},
     {0, 0, 0, 16, 112, 0, 0, 0},
     {0, 0, 0, 8, 120, 0, 0, 0}},
    {{0, 0, 0, 128, 0, 0, 0, 0},
     {0, 0, -4, 126, 8, -2, 0, 0},
     {0, 0, -8, 122, 18, -4, 0, 0},
     {0, 0, -10, 116, 28, -6, 0, 0},
     {0, 0, -12, 110, 38, -8, 0, 0},
     {0, 0, -12, 102, 48, -10, 0, 0},
     {0, 0, -14, 94, 58, -10, 0, 0},
     {0, 0, -12, 84, 66, -10, 0, 0},
     {0, 0, -12, 76, 76, -12, 0, 0},
     {0, 0, -10, 66, 84, -12, 0, 0},
     {0, 0, -10, 58, 94, -14, 0, 0},
     {0, 0, -10, 48, 102, -12, 0, 0},
     {0, 0, -8, 38, 110, -12, 0, 0},
     {0, 0, -6, 28, 116, -10, 0, 0},
     {0, 0, -4, 18, 122, -8, 0, 0},
     {0, 0, -2, 8, 126, -4, 0, 0}},
    {{0, 0, 0, 128, 0, 0, 0, 0},
     {0, 0, 30, 62, 34, 2, 0, 0},
     {0, 0, 26, 62, 36, 4, 0, 0},
     {0, 0, 22, 62, 40, 4, 0, 0},
     {0, 0, 20, 60, 42, 6, 0, 0},
     {0, 0, 18, 58, 44, 8, 0, 0},
     {0, 0, 16, 56, 46, 10, 0, 0},
     {0, 0, 14, 54, 48, 12, 0, 0},
     {0, 0, 12, 52, 52, 12, 0, 0},
     {0, 0, 12, 48, 54, 14, 0, 0},
     {0, 0, 10, 46, 56, 16, 0, 0},
     {0, 0, 8, 44, 58, 18, 0, 0},
     {0, 0, 6, 42, 60, 20, 0, 0},
     {0, 0, 4, 40, 62, 22, 0, 0},
     {0, 0, 4, 36, 62, 26, 0, 0},
     {0, 0, 2, 34, 62, 30, 0, 0}}};

static inline int32_t px(const void *p, int hbd, ptrdiff_t i) {
  return hbd ? (int32_t)((const uint16_t *)p)[i]
             : (int32_t)((const uint8_t *)p)[i];
}
static inline void put_px(void *p, int hbd, ptrdiff_t i, int32_t v) {
  if (hbd)
    ((uint16_t *)p)[i] = (uint16_t)v;
  else
    ((uint8_t *)p)[i] = (uint8_t)v;
}
static inline int32_t round_shift(int32_t v, int b) {
  return (v + ((1 << b) >> 1)) >> b;
}
static inline int32_t clampi(int32_t v, int32_t lo, int32_t hi) {
  return v < lo ? lo : (v > hi ? hi : v);
}

/* mc.rs:238-247: 4-tap variants when the filtered dimension is <= 4 */
static const int16_t *get_filter(int mode, int frac, int length) {
  int idx = (mode == 3 /*BILINEAR*/ || length > 4) ? mode
                                                   : (mode < 1 ? mode : 1) + 4;
  return FILTERS[idx][frac];
}

const int16_t *r1o_get_filter(int mode, int frac, int length) { return get_filter(mode, frac, length); }

static inline int32_t run_filter_px(const void *src, int hbd, ptrdiff_t base,
                                    ptrdiff_t step, const int16_t *f) {
  int32_t s = 0;
  for (int i = 0; i < 8; i++) s += f[i] * px(src, hbd, base + i * step);
  return s;
}
static inline int32_t run_filter_i16(const int16_t *p, ptrdiff_t step,
                                     const int16_t *f) {
  int32_t s = 0;
  for (int i = 0; i < 8; i++) s += f[i] * (int32_t)p[i * step];
  return s;
}

/* The shared H-then-V structure of put/prep.  The reference processes the
 * 2-D case in 8-column groups through an i16 intermediate of (h+7) rows
 * (mc.rs:313-351); the grouping does not change any value, so the oracle
 * keeps one (h+7) x w intermediate. */
void r1o_put_8tap(void *dst, ptrdiff_t ds, const void *src, ptrdiff_t ss,
                  int w, int h, int col_frac, int row_frac, int mode_x,
                  int mode_y, int bit_depth, int hbd) {
  const int16_t *yf = get_filter(mode_y, row_frac, h);
  const int16_t *xf = get_filter(mode_x, col_frac, w);
  const int32_t maxv = (1 << bit_depth) - 1;
  const int ib = 4 - (bit_depth == 12 ? 2 : 0);
  if (col_frac == 0 && row_frac == 0) {
    for (int r = 0; r < h; r++)
      for (int c = 0; c < w; c++) put_px(dst, hbd, r * ds + c, px(src, hbd, r * ss + c));
  } else if (col_frac == 0) {
    for (int r = 0; r < h; r++)
      for (int c = 0; c < w; c++)
        put_px(dst, hbd, r * ds + c,
               clampi(round_shift(run_filter_px(src, hbd, (r - 3) * ss + c, ss, yf), 7),
                      0, maxv));
  } else if (row_frac == 0) {
    for (int r = 0; r < h; r++)
      for (int c = 0; c < w; c++)
        put_px(dst, hbd, r * ds + c,
               clampi(round_shift(round_shift(run_filter_px(src, hbd, r * ss + c - 3, 1, xf),
                                              7 - ib),
                                  ib),
                      0, maxv));
  } else {
    int16_t mid[(128 + 7) * 128];
    for (int r = 0; r < h + 7; r++)
      for (int c = 0; c < w; c++)
        mid[r * w + c] = (int16_t)round_shift(
            run_filter_px(src, hbd, (r - 3) * ss + c - 3, 1, xf), 7 - ib);
    for (int r = 0; r < h; r++)
      for (int c = 0; c < w; c++)
        put_px(dst, hbd, r * ds + c,
               clampi(round_shift(run_filter_i16(mid + r * w + c, w, yf), 7 + ib), 0, maxv));
  }
}

void r1o_prep_8tap(int16_t *tmp, const void *src, ptrdiff_t ss, int w, int h,
                   int col_frac, int row_frac, int mode_x, int mode_y,
                   int bit_depth, int hbd) {
  const int16_t *yf = get_filter(mode_y, row_frac, h);
  const int16_t *xf = get_filter(mode_x, col_frac, w);
  const int ib = 4 - (bit_depth == 12 ? 2 : 0);
  const int32_t bias = bit_depth == 8 ? 0 : 8192; /* PREP_BIAS */
  if (col_frac == 0 && row_frac == 0) {
    for (int r = 0; r < h; r++)
      for (int c = 0; c < w; c++)
        tmp[r * w + c] =
            (int16_t)((int16_t)(px(src, hbd, r * ss + c) << ib) - (int16_t)bias);
  } else if (col_frac == 0) {
    for (int r = 0; r < h; r++)
      for (int c = 0; c < w; c++)
        tmp[r * w + c] = (int16_t)(
            round_shift(run_filter_px(src, hbd, (r - 3) * ss + c, ss, yf), 7 - ib) - bias);
  } else if (row_frac == 0) {
    for (int r = 0; r < h; r++)
      for (int c = 0; c < w; c++)
        tmp[r * w + c] = (int16_t)(
            round_shift(run_filter_px(src, hbd, r * ss + c - 3, 1, xf), 7 - ib) - bias);
  } else {
    int16_t mid[(128 + 7) * 128];
    for (int r = 0; r < h + 7; r++)
      for (int c = 0; c < w; c++)
        mid[r * w + c] = (int16_t)round_shift(
            run_filter_px(src, hbd, (r - 3) * ss + c - 3, 1, xf), 7 - ib);
    for (int r = 0; r < h; r++)
      for (int c = 0; c < w; c++)
        tmp[r * w + c] =
            (int16_t)(round_shift(run_filter_i16(mid + r * w + c, w, yf), 7) - bias);
  }
}

void r1o_mc_avg(void *dst, ptrdiff_t ds, const int16_t *t1, const int16_t *t2,
                int w, int h, int bit_depth, int hbd) {
  const int32_t maxv = (1 << bit_depth) - 1;
  const int ib = 4 - (bit_depth == 12 ? 2 : 0);
  const int32_t bias = bit_depth == 8 ? 0 : 8192 * 2;
  for (int r = 0; r < h; r++)
    for (int c = 0; c < w; c++)
      put_px(dst, hbd, r * ds + c,
             clampi(round_shift((int32_t)t1[r * w + c] + (int32_t)t2[r * w + c] + bias, ib + 1),
                    0, maxv));
}
