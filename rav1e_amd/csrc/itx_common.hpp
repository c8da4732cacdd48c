// itx_common.hpp -- inverse-transform 1-D networks as device functions.
//
// Restates (reference file:line):
//   half_btf / clamp_value        src/transform/mod.rs:297-315
//   av1_i{dct,adst,identity,wht}* src/transform/inverse.rs:35-1588
//   INV_TXFM_FNS                  src/transform/inverse.rs:1593-1626
//   INV_INTERMEDIATE_SHIFTS       src/transform/inverse.rs:1710-1711
// The networks (inv_tx_1d.inc) are straight-line SSA emitted by
// tools/gen_inv_tx1d.py from the golden-pinned NumPy restatement.
//
// Multiplies: every operand of a half_btf is either a value clamped to
// `range` <= 20 bits (inverse.rs:1651, 1683) or a half_btf of such values
// (|.| < 2^21), and the cosines are 13-bit, so v_mad_i32_i24 returns exactly
// the low 32 bits the reference's wrapping i32 arithmetic produces.
#pragma once
#include "tx_common.hpp"

namespace r1itx {
typedef int32_t T;

#define ITX1D_FN __device__ __forceinline__
#define ITX_ADD(a, b) ((T)((uint32_t)(a) + (uint32_t)(b)))
#define ITX_SUB(a, b) ((T)((uint32_t)(a) - (uint32_t)(b)))
#define ITX_NEG(a) ((T)(0u - (uint32_t)(a)))
#define ITX_MUL(a, m) __mul24((a), (m))
#define ITX_BTF(w0, a, w1, b) ((T)((uint32_t)__mul24((w1), (b)) + ((uint32_t)__mul24((w0), (a)) + 2048u)) >> 12)
#define ITX_BTF1(w0, a) ((T)((uint32_t)__mul24((w0), (a)) + 2048u) >> 12)
#define ITX_CLAMP(x) clamp3((x), lo, hi)
#define ITX_RSHIFT(a, s) ((T)((uint32_t)(a) + (uint32_t)((1 << (s)) >> 1)) >> (s))
#define ITX_SAR(a, s) ((a) >> (s))
__device__ __forceinline__ T clamp3(T v, T lo, T hi) {
  return v < lo ? lo : (v > hi ? hi : v);   // v_med3_i32
}
#include "inv_tx_1d.inc"
#undef ITX1D_FN

static const uint8_t kInvShift[19] = {0, 1, 2, 2, 2, 0, 0, 1, 1, 1,
                                      1, 1, 1, 1, 1, 2, 2, 2, 2};

// One N-point inverse transform of class k (0 DCT, 1 ADST, 2 FLIPADST,
// 3 IDTX, 4 WHT) on a register array.  HALF: the upper half of the input is
// known to be zero (64-point sizes code 32 coefficients per line).
template <int N, bool HALF>
__device__ __forceinline__ void inv_1d(T *c, int k, T lo, T hi) {
  if constexpr (N == 4) {
    if (k == 0) r1_idct4(c, lo, hi);
    else if (k == 3) r1_iidentity4(c, lo, hi);
    else if (k == 4) r1_iwht4(c, lo, hi);
    else r1_iadst4(c, lo, hi);
  } else if constexpr (N == 8) {
    if (k == 0) r1_idct8(c, lo, hi);
    else if (k == 3) r1_iidentity8(c, lo, hi);
    else r1_iadst8(c, lo, hi);
  } else if constexpr (N == 16) {
    if (k == 0) r1_idct16(c, lo, hi);
    else if (k == 3) r1_iidentity16(c, lo, hi);
    else r1_iadst16(c, lo, hi);
  } else if constexpr (N == 32) {
    if (k == 0) r1_idct32(c, lo, hi);
    else r1_iidentity32(c, lo, hi);
  } else {
    if constexpr (HALF) r1_idct64_lo32(c, lo, hi);
    else r1_idct64(c, lo, hi);
  }
  if (N <= 16 && k == 2) {   // av1_iflipadst*: reverse the output
#pragma unroll
    for (int i = 0; i < N / 2; i++) {
      const T t = c[i];
      c[i] = c[N - 1 - i];
      c[N - 1 - i] = t;
    }
  }
}

}  // namespace r1itx
