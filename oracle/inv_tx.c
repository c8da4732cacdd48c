/*
 * oracle/inv_tx.c -- TEST INFRASTRUCTURE (see r1_oracle.h).
 * Inverse 2-D transform + reconstruction, restating
 *   inverse_transform_add      src/transform/inverse.rs:1633-1705
 *   INV_INTERMEDIATE_SHIFTS    src/transform/inverse.rs:1710-1711
 *   INV_TXFM_FNS               src/transform/inverse.rs:1593-1626
 *   half_btf / clamp_value     src/transform/mod.rs:297-315
 *   get_1d_tx_types            src/transform/mod.rs:364-402 (VTX/HTX tables)
 * The 1-D networks come from inv_tx_1d.inc (generated from the NumPy
 * restatement oracle/inv_tx_np.py by tools/gen_inv_tx1d.py).
 */
#include <string.h>

#include "r1_oracle.h"

typedef int32_t T;
#define ITX1D_FN static inline
/* wrapping i32 (Rust release mode) */
#define ITX_ADD(a, b) ((T)((uint32_t)(a) + (uint32_t)(b)))
#define ITX_SUB(a, b) ((T)((uint32_t)(a) - (uint32_t)(b)))
#define ITX_NEG(a) ((T)(0u - (uint32_t)(a)))
#define ITX_MUL(a, m) ((T)((uint32_t)(a) * (uint32_t)(m)))
#define ITX_BTF(w0, a, w1, b)                                              \
  ((T)((uint32_t)(w0) * (uint32_t)(a) + (uint32_t)(w1) * (uint32_t)(b) +  \
       2048u) >> 12)
#define ITX_BTF1(w0, a) ((T)((uint32_t)(w0) * (uint32_t)(a) + 2048u) >> 12)
#define ITX_CLAMP(x) clamp3((x), lo, hi)
#define ITX_RSHIFT(a, s) ((T)((uint32_t)(a) + (uint32_t)((1 << (s)) >> 1)) >> (s))
#define ITX_SAR(a, s) ((a) >> (s))
static inline T clamp3(T v, T lo, T hi) { return v < lo ? lo : (v > hi ? hi : v); }
#include "inv_tx_1d.inc"

typedef void (*itxfm_fn)(T *, T, T);
/* [1-D class: DCT, ADST, FLIPADST, IDTX, WHT][log2(n) - 2] */
static void flip_n(T *c, int n) {
  for (int i = 0; i < n / 2; i++) {
    T t = c[i];
    c[i] = c[n - 1 - i];
    c[n - 1 - i] = t;
  }
}
static void iflipadst4(T *c, T lo, T hi) { r1_iadst4(c, lo, hi); flip_n(c, 4); }
static void iflipadst8(T *c, T lo, T hi) { r1_iadst8(c, lo, hi); flip_n(c, 8); }
static void iflipadst16(T *c, T lo, T hi) { r1_iadst16(c, lo, hi); flip_n(c, 16); }
static const itxfm_fn ITXFM[5][5] = {
    {r1_idct4, r1_idct8, r1_idct16, r1_idct32, r1_idct64},
    {r1_iadst4, r1_iadst8, r1_iadst16, 0, 0},
    {iflipadst4, iflipadst8, iflipadst16, 0, 0},
    {r1_iidentity4, r1_iidentity8, r1_iidentity16, r1_iidentity32, 0},
    {r1_iwht4, 0, 0, 0, 0}};

static const uint8_t TX_W_LOG2[19] = {2, 3, 4, 5, 6, 2, 3, 3, 4, 4,
                                      5, 5, 6, 2, 4, 3, 5, 4, 6};
static const uint8_t TX_H_LOG2[19] = {2, 3, 4, 5, 6, 3, 2, 4, 3, 5,
                                      4, 6, 5, 4, 2, 5, 3, 6, 4};
static const uint8_t VTX[17] = {0, 1, 0, 1, 2, 0, 2, 1, 2, 3, 0, 3, 1, 3, 2, 3, 4};
static const uint8_t HTX[17] = {0, 0, 1, 1, 0, 2, 2, 2, 1, 3, 3, 0, 3, 1, 3, 2, 4};
static const uint8_t INV_SHIFT[19] = {0, 1, 2, 2, 2, 0, 0, 1, 1, 1,
                                      1, 1, 1, 1, 1, 2, 2, 2, 2};

/* cls 0..4, n in {4,8,16,32,64}; returns -1 when the reference has no kernel */
int r1o_inv_txfm_1d(int32_t *c, int cls, int n, int range_bits) {
  int l = 0;
  while ((4 << l) < n) l++;
  if (cls < 0 || cls > 4 || l > 4 || !ITXFM[cls][l]) return -1;
  const T hi = (T)((1ll << (range_bits - 1)) - 1), lo = (T)(-(1ll << (range_bits - 1)));
  ITXFM[cls][l](c, lo, hi);
  return 0;
}

/* dst: pixels (u8 when !hbd, u16 otherwise) holding the prediction on entry
 * and the reconstruction on return; stride in elements.  coeff32 selects the
 * coefficient type (T::Coeff = i16 for u8 pixels, i32 for u16). */
int r1o_inverse_transform_add(const void *coeffs, void *dst, ptrdiff_t stride,
                              int tx_size, int tx_type, int bd, int coeff32,
                              int hbd) {
  if (!r1o_valid_av1_transform(tx_size, tx_type)) return -1;
  const int wl = TX_W_LOG2[tx_size], hl = TX_H_LOG2[tx_size];
  const int w = 1 << wl, h = 1 << hl;
  const int wc = w < 32 ? w : 32, hc = h < 32 ? h : 32;
  const int rect1 = (wl > hl ? wl - hl : hl - wl) == 1;
  const int lossless = tx_type == 16;
  static T buf[64 * 64];
#pragma omp threadprivate(buf)
  memset(buf, 0, sizeof(T) * (size_t)w * h);
  /* rows */
  {
    const int range = bd + 8;
    const T hi = (T)((1ll << (range - 1)) - 1), lo = -hi - 1;
    const itxfm_fn fn = ITXFM[HTX[tx_type]][wl - 2];
    for (int r = 0; r < hc; r++) {
      T tmp[64] = {0};
      for (int c = 0; c < wc; c++) {
        T raw = coeff32 ? ((const int32_t *)coeffs)[c * hc + r]
                        : ((const int16_t *)coeffs)[c * hc + r];
        T val = rect1 ? ITX_RSHIFT(ITX_MUL(raw, 2896), 12)
                      : (lossless ? raw >> 2 : raw);
        tmp[c] = clamp3(val, lo, hi);
      }
      fn(tmp, lo, hi);
      memcpy(buf + r * w, tmp, sizeof(T) * w);
    }
  }
  /* columns */
  {
    const int range = bd + 6 > 16 ? bd + 6 : 16;
    const T hi = (T)((1ll << (range - 1)) - 1), lo = -hi - 1;
    const itxfm_fn fn = ITXFM[VTX[tx_type]][hl - 2];
    const int sh = INV_SHIFT[tx_size];
    const T pmax = (1 << bd) - 1;
    for (int c = 0; c < w; c++) {
      T tmp[64];
      for (int r = 0; r < h; r++)
        tmp[r] = clamp3(ITX_RSHIFT(buf[r * w + c], sh), lo, hi);
      fn(tmp, lo, hi);
      for (int r = 0; r < h; r++) {
        T v = hbd ? ((uint16_t *)dst)[r * stride + c]
                  : ((uint8_t *)dst)[r * stride + c];
        T res = lossless ? tmp[r] : ITX_RSHIFT(tmp[r], 4);
        v = clamp3(v + res, 0, pmax);
        if (hbd)
          ((uint16_t *)dst)[r * stride + c] = (uint16_t)v;
        else
          ((uint8_t *)dst)[r * stride + c] = (uint8_t)v;
      }
    }
  }
  return 0;
}

/* batch mirror of r1_inv_txfm_add_batch: n dense blocks; coeffs has
 * `coeff_stride` entries per block (only the first min(w,32)*min(h,32) are
 * read); pred/rec are dense w*h pixel blocks (rec may alias pred). */
int r1o_inv_txfm_add_batch(const void *coeffs, int coeff_stride, const void *pred,
                           void *rec, int n, int tx_size, int tx_type,
                           int bit_depth, int coeff_bytes, int bytes_per_px) {
  if (!r1o_valid_av1_transform(tx_size, tx_type)) return -1;
  const int w = 1 << TX_W_LOG2[tx_size], h = 1 << TX_H_LOG2[tx_size];
  const size_t pb = (size_t)w * h * bytes_per_px;
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n; i++) {
    uint8_t *d = (uint8_t *)rec + pb * i;
    if (rec != pred) memcpy(d, (const uint8_t *)pred + pb * i, pb);
    r1o_inverse_transform_add(
        (const uint8_t *)coeffs + (size_t)i * coeff_stride * coeff_bytes, d, w,
        tx_size, tx_type, bit_depth, coeff_bytes == 4, bytes_per_px == 2);
  }
  return 0;
}
