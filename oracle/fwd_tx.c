/*
 * oracle/fwd_tx.c -- TEST INFRASTRUCTURE (see r1_oracle.h).
 * Forward 2-D transform, restating
 *   forward_transform         src/transform/forward.rs:71-161
 *   Txfm2DFlipCfg::fwd        src/transform/forward_shared.rs:128-165
 *   FWD_TXFM_SHIFT_LS         src/transform/forward_shared.rs:22-64
 *   VTX_TAB / HTX_TAB         src/transform/mod.rs:364-402
 *   valid_av1_transform       src/transform/mod.rs:405-417
 *   av1_round_shift_array     src/transform/mod.rs:317-331
 * The 1-D networks come from fwd_tx_1d.inc (generated from the NumPy
 * restatement oracle/fwd_tx_np.py by tools/gen_tx1d.py).
 */
#include "r1_oracle.h"

typedef int32_t T;
#define TX1D_FN static inline
/* i32 ops with Rust release-mode (wrapping) semantics */
#define TX_ADD(a, b) ((T)((uint32_t)(a) + (uint32_t)(b)))
#define TX_SUB(a, b) ((T)((uint32_t)(a) - (uint32_t)(b)))
#define TX_MUL(a, m, s) \
  ((T)((uint32_t)(a) * (uint32_t)(m) + (uint32_t)((1 << (s)) >> 1)) >> (s))
#define TX_RSHIFT1(a) (TX_ADD((a), (T)((a) < 0)) >> 1)
#define TX_ADD_AVG(a, b) (TX_ADD(a, b) >> 1)
#define TX_SUB_AVG(a, b) (TX_SUB(a, b) >> 1)
#include "fwd_tx_1d.inc"

static void fidentity(T *c) { (void)c; }

typedef void (*txfm_fn)(T *);
/* TxfmType order: DCT4,8,16,32,64, ADST4,8,16, Identity4,8,16,32, WHT4 */
static const txfm_fn TXFM_FN[13] = {
    r1_fdct4, r1_fdct8, r1_fdct16, r1_fdct32, r1_fdct64, r1_fdst_vii_4,
    r1_fdst8, r1_fdst16, fidentity, fidentity, fidentity, fidentity, r1_fwht4};

static const uint8_t TX_W_LOG2[19] = {2, 3, 4, 5, 6, 2, 3, 3, 4, 4,
                                      5, 5, 6, 2, 4, 3, 5, 4, 6};
static const uint8_t TX_H_LOG2[19] = {2, 3, 4, 5, 6, 3, 2, 4, 3, 5,
                                      4, 6, 5, 4, 2, 5, 3, 6, 4};
/* 1-D types: 0 DCT 1 ADST 2 FLIPADST 3 IDTX 4 WHT */
static const uint8_t VTX[17] = {0, 1, 0, 1, 2, 0, 2, 1, 2, 3, 0, 3, 1, 3, 2, 3, 4};
static const uint8_t HTX[17] = {0, 0, 1, 1, 0, 2, 2, 2, 1, 3, 3, 0, 3, 1, 3, 2, 4};
static const int8_t TXFM_TYPE_LS[5][5] = {{0, 5, 5, 8, 12},
                                          {1, 6, 6, 9, -1},
                                          {2, 7, 7, 10, -1},
                                          {3, -1, -1, 11, -1},
                                          {4, -1, -1, -1, -1}};
/* shift class per TxSize: 0 = 4x4, 1 = [4,-1,0].., 2 = [4,-2,0].., 3 = 64-pt */
static const uint8_t SHIFT_CLASS[19] = {0, 1, 1, 2, 3, 1, 1, 1, 1, 2,
                                        2, 3, 3, 1, 1, 1, 1, 2, 2};
static const int8_t FWD_SHIFT[4][3][3] = {
    {{3, 0, 0}, {2, 0, 1}, {0, 0, 3}},
    {{4, -1, 0}, {2, 0, 1}, {0, 0, 3}},
    {{4, -2, 0}, {2, 0, 0}, {0, 0, 2}},
    {{4, -1, -2}, {2, 0, -1}, {0, 0, 1}}};

int r1o_tx_width(int tx_size) { return 1 << TX_W_LOG2[tx_size]; }
int r1o_tx_height(int tx_size) { return 1 << TX_H_LOG2[tx_size]; }

int r1o_valid_av1_transform(int tx_size, int tx_type) {
  if (tx_size < 0 || tx_size >= 19 || tx_type < 0 || tx_type >= 17) return 0;
  int wl = TX_W_LOG2[tx_size], hl = TX_H_LOG2[tx_size];
  int m = wl > hl ? wl : hl;
  if (tx_type == 16) return wl == 2 && hl == 2; /* WHT only exists at 4x4 */
  if (m == 6) return tx_type == 0;
  if (m == 5) return tx_type == 0 || tx_type == 9;
  return 1;
}

void r1o_fwd_txfm_1d(int32_t *coeffs, int txfm_type) {
  TXFM_FN[txfm_type](coeffs);
}

static inline void round_shift_array(T *a, int n, int bit) {
  if (bit == 0) return;
  if (bit > 0)
    for (int i = 0; i < n; i++) a[i] = (a[i] + ((1 << bit) >> 1)) >> bit;
  else
    for (int i = 0; i < n; i++) a[i] = (T)((uint32_t)a[i] << -bit);
}

int r1o_forward_transform(const int16_t *input, void *output, size_t stride,
                          int tx_size, int tx_type, int bd, int coeff32) {
  if (!r1o_valid_av1_transform(tx_size, tx_type)) return -1;
  const int w = 1 << TX_W_LOG2[tx_size], h = 1 << TX_H_LOG2[tx_size];
  const int tcol = TXFM_TYPE_LS[TX_H_LOG2[tx_size] - 2][VTX[tx_type]];
  const int trow = TXFM_TYPE_LS[TX_W_LOG2[tx_size] - 2][HTX[tx_type]];
  static const int8_t WHT_SHIFT[3] = {0, 0, 2};
  const int8_t *shift = tx_type == 16
                            ? WHT_SHIFT
                            : FWD_SHIFT[SHIFT_CLASS[tx_size]][(bd - 8) / 2];
  const int ud = tx_type == 4 || tx_type == 8 || tx_type == 14 || tx_type == 6;
  const int lr = tx_type == 5 || tx_type == 7 || tx_type == 15 || tx_type == 6;
  T tmp[64];
  T lbuf[64 * 64];
  /* columns */
  for (int c = 0; c < w; c++) {
    for (int r = 0; r < h; r++)
      tmp[r] = input[(size_t)(ud ? h - r - 1 : r) * stride + c];
    round_shift_array(tmp, h, -shift[0]);
    TXFM_FN[tcol](tmp);
    round_shift_array(tmp, h, -shift[1]);
    const int cc = lr ? w - c - 1 : c;
    for (int r = 0; r < h; r++) lbuf[r * w + cc] = tmp[r];
  }
  /* rows, stored transposed in <=32x32 chunks (forward.rs:135-159) */
  const int ostride = h < 32 ? h : 32;
  const int wc = w < 32 ? w : 32;
  for (int r = 0; r < h; r++) {
    T *row = lbuf + r * w;
    TXFM_FN[trow](row);
    round_shift_array(row, w, -shift[2]);
    const size_t base = (size_t)(r >= 32) * ostride * wc;
    for (int cg = 0; cg < w; cg += 32)
      for (int c = 0; c < wc; c++) {
        const size_t o = base + (size_t)h * cg + (size_t)c * ostride + (r & 31);
        if (coeff32)
          ((int32_t *)output)[o] = row[c + cg];
        else
          ((int16_t *)output)[o] = (int16_t)row[c + cg];
      }
  }
  return 0;
}
