// tx_common.hpp -- forward-transform configuration tables and the 1-D
// butterfly networks as device functions.
//
// Restates (reference file:line):
//   TxSize dims                 src/transform/mod.rs:101-167
//   VTX_TAB / HTX_TAB           src/transform/mod.rs:364-402
//   valid_av1_transform         src/transform/mod.rs:405-417
//   AV1_TXFM_TYPE_LS            src/transform/forward_shared.rs:85-109
//   FWD_TXFM_SHIFT_LS           src/transform/forward_shared.rs:22-64
//   get_flip_cfg                src/transform/forward_shared.rs:155-164
//   TxOperations for i32        src/transform/forward.rs:37-65
// The 1-D networks (fwd_tx_1d.inc) are straight-line SSA emitted by
// tools/gen_tx1d.py; they operate on a per-lane register array.
#pragma once
#include "common.hpp"

namespace r1tx {

typedef int32_t T;
#define TX1D_FN __device__ __forceinline__
// i32 with wrapping semantics (Rust release build); products of i32 are taken
// mod 2^32 exactly like `self * mul` in forward.rs:43.
#define TX_ADD(a, b) ((T)((uint32_t)(a) + (uint32_t)(b)))
#define TX_SUB(a, b) ((T)((uint32_t)(a) - (uint32_t)(b)))
#define TX_MUL(a, m, s) \
  ((T)((uint32_t)(a) * (uint32_t)(m) + (uint32_t)((1 << (s)) >> 1)) >> (s))
#define TX_RSHIFT1(a) (TX_ADD((a), (T)((a) < 0)) >> 1)
#define TX_ADD_AVG(a, b) (TX_ADD(a, b) >> 1)
#define TX_SUB_AVG(a, b) (TX_SUB(a, b) >> 1)
#include "fwd_tx_1d.inc"
#undef TX_MUL

// Second instantiation for residuals that come from pixels (fused kernel):
// every multiplier input is then < 2^18.4 in magnitude for all sizes, types
// and bit depths (tools/tx_range.py, tests/test_tx_range.py), so the
// full-rate 24-bit multiply returns the same low 32 product bits as the
// wrapping i32 multiply of forward.rs:43.
namespace m24 {
template <int M>
__device__ __forceinline__ T mul24(T a) {
  T r;
  asm("v_mul_i32_i24_e32 %0, %1, %2" : "=v"(r) : "n"(M), "v"(a));
  return r;
}
// (a * M + round) as ONE full-rate v_mad_i32_i24, spelled out: left to the
// compiler (__mul24 + add) about a third of the network's multiplies came out
// as quarter-rate v_mul_lo_u32 + v_add (the 24-bit range proof is lost in the
// instruction selector's depth-limited known-bits walk).  Multiplier in an
// SGPR, rounding constant in a VGPR (VOP3 has no literals on gfx9).
template <int M, int S>
__device__ __forceinline__ T mul_rs(T a) {
  T r;
  asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(M), "v"((1 << S) >> 1));
  return r >> S;
}
#define TX_MUL(a, m, s) (mul_rs<(m), (s)>(a))
#include "fwd_tx_1d.inc"
#undef TX_MUL
}  // namespace m24
#undef TX1D_FN

// 1-D transform classes: 0 DCT, 1 ADST, 2 FLIPADST, 3 IDTX, 4 WHT (VTX_TAB / HTX_TAB,
// src/transform/mod.rs:364-402).  Two bits per tx_type 0..15 in one immediate (WHT = 16 apart):
// a table in memory would be a dependent global load in the middle of a kernel.
constexpr uint32_t pack_tx_tab(const uint8_t (&t)[16]) {
  uint32_t v = 0;
  for (int i = 0; i < 16; i++) v |= (uint32_t)t[i] << (2 * i);
  return v;
}
__host__ __device__ inline int vtx_1d(int tx_type) {
  constexpr uint8_t t[16] = {0, 1, 0, 1, 2, 0, 2, 1, 2, 3, 0, 3, 1, 3, 2, 3};
  constexpr uint32_t k = pack_tx_tab(t);
  return tx_type == 16 ? 4 : (int)((k >> (2 * tx_type)) & 3u);
}
__host__ __device__ inline int htx_1d(int tx_type) {
  constexpr uint8_t t[16] = {0, 0, 1, 1, 0, 2, 2, 2, 1, 3, 3, 0, 3, 1, 3, 2};
  constexpr uint32_t k = pack_tx_tab(t);
  return tx_type == 16 ? 4 : (int)((k >> (2 * tx_type)) & 3u);
}
// get_flip_cfg (src/transform/forward_shared.rs:155-164) as bit masks over tx_type
__host__ __device__ inline bool ud_flip(int tx_type) {
  return ((1u << 4 | 1u << 8 | 1u << 14 | 1u << 6) >> tx_type) & 1u;
}
__host__ __device__ inline bool lr_flip(int tx_type) {
  return ((1u << 5 | 1u << 7 | 1u << 15 | 1u << 6) >> tx_type) & 1u;
}

static const uint8_t kTxWLog2[19] = {2, 3, 4, 5, 6, 2, 3, 3, 4, 4,
                                     5, 5, 6, 2, 4, 3, 5, 4, 6};
static const uint8_t kTxHLog2[19] = {2, 3, 4, 5, 6, 3, 2, 4, 3, 5,
                                     4, 6, 5, 4, 2, 5, 3, 6, 4};

inline bool valid_av1_transform(int tx_size, int tx_type) {
  if (tx_size < 0 || tx_size >= 19 || tx_type < 0 || tx_type >= 17) return false;
  const int wl = kTxWLog2[tx_size], hl = kTxHLog2[tx_size];
  const int m = wl > hl ? wl : hl;
  if (tx_type == 16) return wl == 2 && hl == 2;
  if (m == 6) return tx_type == 0;
  if (m == 5) return tx_type == 0 || tx_type == 9;
  return true;
}

struct Shift3 { int8_t s[3]; };
inline Shift3 fwd_shift(int tx_size, int tx_type, int bd) {
  static const uint8_t cls[19] = {0, 1, 1, 2, 3, 1, 1, 1, 1, 2,
                                  2, 3, 3, 1, 1, 1, 1, 2, 2};
  static const int8_t tab[4][3][3] = {{{3, 0, 0}, {2, 0, 1}, {0, 0, 3}},
                                      {{4, -1, 0}, {2, 0, 1}, {0, 0, 3}},
                                      {{4, -2, 0}, {2, 0, 0}, {0, 0, 2}},
                                      {{4, -1, -2}, {2, 0, -1}, {0, 0, 1}}};
  Shift3 r;
  if (tx_type == 16) {
    r.s[0] = 0; r.s[1] = 0; r.s[2] = 2;
  } else {
    for (int i = 0; i < 3; i++) r.s[i] = tab[cls[tx_size]][(bd - 8) / 2][i];
  }
  return r;
}

// Compile-time FWD_TXFM_SHIFT_LS (forward_shared.rs:22-64) for non-WHT types:
// the triple depends only on (tx size class, bit depth), so a kernel
// instantiated per size and bit depth gets immediate shift amounts (a shift
// by 0 disappears, the rounding constant folds).
constexpr int fwd_shift_class(int wl, int hl) {
  const int mx = wl > hl ? wl : hl, mn = wl > hl ? hl : wl;
  if (mx == 2) return 0;
  if (mx == 6 && mn >= 5) return 3;
  if ((mx == 5 && mn >= 4) || (mx == 6 && mn == 4)) return 2;
  return 1;
}
constexpr int fwd_shift_ct(int wl, int hl, int bd, int stage) {
  constexpr int8_t tab[4][3][3] = {{{3, 0, 0}, {2, 0, 1}, {0, 0, 3}},
                                   {{4, -1, 0}, {2, 0, 1}, {0, 0, 3}},
                                   {{4, -2, 0}, {2, 0, 0}, {0, 0, 2}},
                                   {{4, -1, -2}, {2, 0, -1}, {0, 0, 1}}};
  return tab[fwd_shift_class(wl, hl)][(bd - 8) / 2][stage];
}
template <int SHIFT>
__device__ __forceinline__ T shift_fwd_ct(T v) {
  if constexpr (SHIFT >= 0) return (T)((uint32_t)v << SHIFT);
  else return (v + ((1 << -SHIFT) >> 1)) >> -SHIFT;
}

// av1_round_shift_array with bit = -shift (transform/mod.rs:317-331)
__device__ __forceinline__ T shift_fwd(T v, int shift) {
  // shift > 0: left shift; shift < 0: rounding right shift by -shift
  if (shift >= 0) return (T)((uint32_t)v << shift);
  const int b = -shift;
  return (v + ((1 << b) >> 1)) >> b;
}

// Same dispatch on the 24-bit-multiply instantiation.
template <int N>
__device__ __forceinline__ void fwd_1d_m24(T *c, int k) {
  if (k == 3) return;
  if constexpr (N == 4) {
    if (k == 0) m24::r1_fdct4(c);
    else if (k == 4) m24::r1_fwht4(c);
    else m24::r1_fdst_vii_4(c);
  } else if constexpr (N == 8) {
    if (k == 0) m24::r1_fdct8(c); else m24::r1_fdst8(c);
  } else if constexpr (N == 16) {
    if (k == 0) m24::r1_fdct16(c); else m24::r1_fdst16(c);
  } else if constexpr (N == 32) {
    m24::r1_fdct32(c);
  } else {
    m24::r1_fdct64(c);
  }
}

// One 1-D forward transform of length N on a register array, class `k`
// (0 DCT, 1/2 ADST (flip handled by the caller), 3 identity, 4 WHT).
template <int N>
__device__ __forceinline__ void fwd_1d(T *c, int k) {
  if (k == 3) return;  // fidentity is a no-op (forward_shared.rs:1775)
  if constexpr (N == 4) {
    if (k == 0) r1_fdct4(c);
    else if (k == 4) r1_fwht4(c);
    else r1_fdst_vii_4(c);
  } else if constexpr (N == 8) {
    if (k == 0) r1_fdct8(c); else r1_fdst8(c);
  } else if constexpr (N == 16) {
    if (k == 0) r1_fdct16(c); else r1_fdst16(c);
  } else if constexpr (N == 32) {
    r1_fdct32(c);
  } else {
    r1_fdct64(c);
  }
}

}  // namespace r1tx
