/*
 * oracle/dist.c -- TEST INFRASTRUCTURE (see r1_oracle.h).
 * Scalar restatement of the reference's distortion kernels:
 *   get_sad            src/dist.rs:31-52
 *   get_satd           src/dist.rs:156-221 (hadamard4_1d 61, hadamard8_1d 84,
 *                      hadamard2d 122)
 *   get_weighted_sse   src/dist.rs:234-283
 *   cdef_dist_kernel   src/dist.rs:302-372
 *   apply_ssim_boost   src/activity.rs:159-186 (ssim_boost_rsqrt 109-145)
 *   cdef_dist_wxh      src/rdo.rs:142-173, DistortionScale::mul_u64 rdo.rs:613
 */
#include "r1_oracle.h"

static inline int32_t px(const void *p, int hbd, ptrdiff_t i) {
  return hbd ? (int32_t)((const uint16_t *)p)[i]
             : (int32_t)((const uint8_t *)p)[i];
}

uint32_t r1o_get_sad(const void *org, ptrdiff_t os, const void *ref,
                     ptrdiff_t rs, int w, int h, int hbd) {
  uint32_t sum = 0;
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      int32_t d = px(org, hbd, y * os + x) - px(ref, hbd, y * rs + x);
      sum += (uint32_t)(d < 0 ? -d : d);
    }
  return sum;
}

/* Unnormalised Hadamard butterflies in the order the reference fixes
 * (dist.rs:95-117): pairs (0,1)(2,3).. then (0,2)(1,3).. then (0,4)(1,5).. */
static void hadamard_1d(int32_t *d, int n, int count, int stride0,
                        int stride1) {
  for (int i = 0; i < count; i++) {
    int32_t *s = d + i * stride0;
    if (n == 4) {
      int32_t a0 = s[0] + s[stride1], a1 = s[0] - s[stride1];
      int32_t a2 = s[2 * stride1] + s[3 * stride1],
              a3 = s[2 * stride1] - s[3 * stride1];
      s[0] = a0 + a2;
      s[stride1] = a1 + a3;
      s[2 * stride1] = a0 - a2;
      s[3 * stride1] = a1 - a3;
    } else {
      int32_t a[8], b[8];
      for (int k = 0; k < 4; k++) {
        a[2 * k] = s[(2 * k) * stride1] + s[(2 * k + 1) * stride1];
        a[2 * k + 1] = s[(2 * k) * stride1] - s[(2 * k + 1) * stride1];
      }
      b[0] = a[0] + a[2]; b[2] = a[0] - a[2];
      b[1] = a[1] + a[3]; b[3] = a[1] - a[3];
      b[4] = a[4] + a[6]; b[6] = a[4] - a[6];
      b[5] = a[5] + a[7]; b[7] = a[5] - a[7];
      for (int k = 0; k < 4; k++) {
        s[k * stride1] = b[k] + b[k + 4];
        s[(k + 4) * stride1] = b[k] - b[k + 4];
      }
    }
  }
}

uint32_t r1o_get_satd(const void *org, ptrdiff_t os, const void *ref,
                      ptrdiff_t rs, int w, int h, int hbd) {
  int size = w < h ? w : h;
  if (size > 8) size = 8;
  /* the reference only ever selects 4 or 8 (dist.rs:166-167) */
  if (size != 4) size = 8;
  uint64_t sum = 0;
  for (int cy = 0; cy < h; cy += size) {
    int ch = h - cy < size ? h - cy : size;
    for (int cx = 0; cx < w; cx += size) {
      int cw = w - cx < size ? w - cx : size;
      const ptrdiff_t oo = cy * os + cx, ro = cy * rs + cx;
      if (cw != size || ch != size) {
        /* edge chunk: SAD fallback (dist.rs:186-191) */
        uint32_t s = 0;
        for (int y = 0; y < ch; y++)
          for (int x = 0; x < cw; x++) {
            int32_t d =
                px(org, hbd, oo + y * os + x) - px(ref, hbd, ro + y * rs + x);
            s += (uint32_t)(d < 0 ? -d : d);
          }
        sum += s;
        continue;
      }
      int32_t buf[64];
      for (int y = 0; y < size; y++)
        for (int x = 0; x < size; x++)
          buf[y * size + x] =
              px(org, hbd, oo + y * os + x) - px(ref, hbd, ro + y * rs + x);
      /* vertical then horizontal (hadamard2d, dist.rs:122-139) */
      hadamard_1d(buf, size, size, 1, size);
      hadamard_1d(buf, size, size, size, 1);
      for (int i = 0; i < size * size; i++)
        sum += (uint64_t)(buf[i] < 0 ? -(int64_t)buf[i] : buf[i]);
    }
  }
  int ln = size == 4 ? 2 : 3; /* msb(size) */
  return (uint32_t)((sum + ((1u << ln) >> 1)) >> ln);
}

uint64_t r1o_get_weighted_sse(const void *s1, ptrdiff_t st1, const void *s2,
                              ptrdiff_t st2, const uint32_t *scale,
                              size_t scale_stride, int w, int h, int hbd) {
  /* 4x4 cells (IMPORTANCE_BLOCK_SIZE >> 1), each cell's SSE weighted by its
   * own Q14 scale and rounded off by GET_WEIGHTED_SSE_SHIFT=8 before the sum */
  uint64_t sse = 0;
  for (int cy = 0; cy + 4 <= h; cy += 4)
    for (int cx = 0; cx + 4 <= w; cx += 4) {
      uint32_t sum = 0;
      for (int y = 0; y < 4; y++)
        for (int x = 0; x < 4; x++) {
          int32_t c = px(s1, hbd, (cy + y) * st1 + cx + x) -
                      px(s2, hbd, (cy + y) * st2 + cx + x);
          sum += (uint32_t)(c * c);
        }
      uint32_t sc = scale[(size_t)(cy >> 2) * scale_stride + (cx >> 2)];
      sse += ((uint64_t)sum * sc + 128) >> 8;
    }
  /* den = DistortionScale::new(1, 256).0 = ((1<<14) + 128) / 256 = 64 */
  const uint64_t den = ((1ull << 14) + 128) / 256;
  return (sse + (den >> 1)) / den;
}

static inline int ilog2_u64(uint64_t x) {
  int k = 0;
  while (x >>= 1) k++;
  return k;
}

uint32_t r1o_apply_ssim_boost(uint32_t input, uint32_t svar32, uint32_t dvar32,
                              int bit_depth) {
  const int coeff_shift = bit_depth - 8;
  const uint64_t svar = svar32 >> (2 * coeff_shift);
  const uint64_t dvar = dvar32 >> (2 * coeff_shift);
  const uint64_t C1 = 3355, C2 = 16128, C3 = 12338;
  const uint64_t RATIO = (((C1 << 15) / C3) + 1) >> 1;
  /* ssim_boost_rsqrt (activity.rs:109-145), INSHIFT 16, OUTSHIFT 14 */
  const uint64_t x = C1 * C1 + svar * dvar;
  const int k = ilog2_u64(x) >> 1;
  const int s = 2 * k - 14;
  const uint16_t t = (uint16_t)(s > 0 ? x >> s : x << -s);
  const int rshift = (uint8_t)(14 + ((s + 16) >> 1));
  const int32_t n = (int32_t)t - 32768;
  const int32_t inner = -13490 + ((n * 6711) >> 15);
  const int32_t rsqrt = 23557 + ((n * inner) >> 15);
  const uint64_t norm = (uint16_t)rsqrt;
  return (uint32_t)(((uint64_t)input *
                     (((RATIO * (svar + dvar + C2)) * norm) >> 14)) >>
                    rshift);
}

static const uint16_t AREA_DIVISORS[64] = {
    16384, 8192, 5461, 4096, 3277, 2731, 2341, 2048, 1820, 1638, 1489,
    1365,  1260, 1170, 1092, 1024, 964,  910,  862,  819,  780,  745,
    712,   683,  655,  630,  607,  585,  565,  546,  529,  512,  496,
    482,   468,  455,  443,  431,  420,  410,  400,  390,  381,  372,
    364,   356,  349,  341,  334,  328,  321,  315,  309,  303,  298,
    293,   287,  282,  278,  273,  269,  264,  260,  256};

uint32_t r1o_cdef_dist_kernel(const void *src, ptrdiff_t ss, const void *dst,
                              ptrdiff_t ds, int w, int h, int bit_depth,
                              int hbd) {
  uint32_t sum_s = 0, sum_d = 0, sum_s2 = 0, sum_d2 = 0, sum_sd = 0;
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      uint32_t s = (uint32_t)px(src, hbd, y * ss + x);
      uint32_t d = (uint32_t)px(dst, hbd, y * ds + x);
      sum_s += s;
      sum_d += d;
      sum_s2 += s * s;
      sum_d2 += d * d;
      sum_sd += s * d;
    }
  const uint32_t sse = sum_d2 + sum_s2 - 2 * sum_sd;
  const uint64_t S = sum_s, D = sum_d;
  const uint64_t div = AREA_DIVISORS[w * h - 1];
  const uint32_t ms = (uint32_t)((S * S * div + (1u << 14 >> 1)) >> 14);
  const uint32_t md = (uint32_t)((D * D * div + (1u << 14 >> 1)) >> 14);
  uint32_t svar = sum_s2 > ms ? sum_s2 - ms : 0; /* saturating_sub */
  uint32_t dvar = sum_d2 > md ? sum_d2 - md : 0;
  /* scale variances to 8x8 size: shift = AREA_DIVISOR_BITS - 6 = 8 */
  svar = (uint32_t)(((uint64_t)svar * div + (1u << 8 >> 1)) >> 8);
  dvar = (uint32_t)(((uint64_t)dvar * div + (1u << 8 >> 1)) >> 8);
  return r1o_apply_ssim_boost(sse, svar, dvar, bit_depth);
}

uint64_t r1o_cdef_dist_wxh(const void *s1, ptrdiff_t st1, const void *s2,
                           ptrdiff_t st2, int w, int h, int bit_depth, int hbd,
                           const uint32_t *bias, size_t bias_stride) {
  uint64_t sum = 0;
  const int bpp = hbd ? 2 : 1;
  for (int y = 0; y < h; y += 8)
    for (int x = 0; x < w; x += 8) {
      int kh = h - y < 8 ? h - y : 8, kw = w - x < 8 ? w - x : 8;
      const uint8_t *a = (const uint8_t *)s1 + (y * st1 + x) * bpp;
      const uint8_t *b = (const uint8_t *)s2 + (y * st2 + x) * bpp;
      uint64_t v = r1o_cdef_dist_kernel(a, st1, b, st2, kw, kh, bit_depth, hbd);
      uint64_t sc = bias ? bias[(size_t)(y >> 3) * bias_stride + (x >> 3)]
                         : (1u << 14);
      sum += (sc * v + (1u << 14 >> 1)) >> 14; /* DistortionScale::mul_u64 */
    }
  return sum;
}

void r1o_diff(int16_t *dst, const void *src1, ptrdiff_t st1, const void *src2,
              ptrdiff_t st2, int w, int h, int hbd) {
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++)
      dst[y * w + x] =
          (int16_t)(px(src1, hbd, y * st1 + x) - px(src2, hbd, y * st2 + x));
}
