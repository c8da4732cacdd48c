// inv_tx.hip -- batched inverse 2-D transform + reconstruction
// (reference: inverse_transform_add, src/transform/inverse.rs:1633-1705; x86
// entry src/asm/x86/transform/inverse.rs, shared wrapper
// src/asm/shared/transform/inverse.rs:30-36).
//
// Mapping (wave = 64, one wave per workgroup): a wave owns NC = 64 / max(W,H)
// blocks.
//  rows    lane = (block, row r < min(H,32)): the coefficients are stored
//          transposed (index c*min(H,32) + r), so for a fixed column the lanes
//          of a block read consecutive addresses -- coalesced.  2:1 rectangles
//          are scaled by 1/sqrt2, values clamped to bd+8 bits, the W-point
//          network runs in registers (64-point: inputs 32..63 are zero and
//          folded away at code-generation time), results go to an LDS tile
//          with an odd row stride.
//  columns lane = (block, column): intermediate round shift + clamp, H-point
//          network, >> 4, add to the prediction row by row (lanes of a block
//          touch consecutive pixels: coalesced loads and stores), clamp to
//          the pixel range.
#include <type_traits>

#include "itx_common.hpp"

namespace {
using r1itx::T;

template <int BPP, int WL, int HL>
__global__ __launch_bounds__(64) void k_inv_tx(
    const typename std::conditional<BPP == 1, int16_t, int32_t>::type *__restrict__ coeffs,
    int coeff_stride, const void *__restrict__ pred, void *__restrict__ rec, int n,
    int tx_type, int bd, int inv_shift) {
  typedef typename std::conditional<BPP == 1, uint8_t, uint16_t>::type PX;
  constexpr int W = 1 << WL, H = 1 << HL;
  constexpr int WC = W < 32 ? W : 32, HC = H < 32 ? H : 32;
  constexpr int P = W > H ? W : H, NC = 64 / P;
  constexpr int LSTRIDE = NC * W + 1;
  constexpr bool RECT1 = (WL > HL ? WL - HL : HL - WL) == 1;
  __shared__ T buf[HC * LSTRIDE];
  const int lane = threadIdx.x;
  const int krow = r1tx::htx_1d(tx_type), kcol = r1tx::vtx_1d(tx_type);
  const bool lossless = tx_type == 16;

  // ---- rows ----
  if (lane < NC * HC) {
    const int cl = lane / HC, r = lane % HC;
    const long long blk = (long long)blockIdx.x * NC + cl;
    if (blk < n) {
      const int range = bd + 8;
      const T hi = (T)((1 << (range - 1)) - 1), lo = -hi - 1;
      const auto *src = coeffs + blk * coeff_stride + r;
      T v[W];
#pragma unroll
      for (int c = 0; c < WC; c++) {
        const T raw = (T)src[c * HC];
        // raw is unclamped: full wrapping i32 multiply (inverse.rs:1668)
        const T val = RECT1 ? ((T)((uint32_t)raw * 2896u + 2048u) >> 12)
                            : (lossless ? raw >> 2 : raw);
        v[c] = r1itx::clamp3(val, lo, hi);
      }
#pragma unroll
      for (int c = WC; c < W; c++) v[c] = 0;
      r1itx::inv_1d<W, true>(v, krow, lo, hi);
#pragma unroll
      for (int c = 0; c < W; c++) buf[r * LSTRIDE + cl * W + c] = v[c];
    }
  }
  __syncthreads();
  // ---- columns ----
  if (lane < NC * W) {
    const int cl = lane / W, c = lane % W;
    const long long blk = (long long)blockIdx.x * NC + cl;
    if (blk < n) {
      const int range = bd + 6 > 16 ? bd + 6 : 16;
      const T hi = (T)((1 << (range - 1)) - 1), lo = -hi - 1;
      const T pmax = (T)((1 << bd) - 1);
      T v[H];
#pragma unroll
      for (int r = 0; r < HC; r++) {
        const T x = buf[r * LSTRIDE + cl * W + c];
        v[r] = r1itx::clamp3((x + ((1 << inv_shift) >> 1)) >> inv_shift, lo, hi);
      }
#pragma unroll
      for (int r = HC; r < H; r++) v[r] = 0;
      r1itx::inv_1d<H, true>(v, kcol, lo, hi);
      const PX *pp = (const PX *)pred + blk * (W * H) + c;
      PX *pr = (PX *)rec + blk * (W * H) + c;
#pragma unroll
      for (int r = 0; r < H; r++) {
        const T res = lossless ? v[r] : (v[r] + 8) >> 4;
        const T px = (T)pp[r * W] + res;
        pr[r * W] = (PX)(px < 0 ? 0 : (px > pmax ? pmax : px));
      }
    }
  }
}

template <int BPP, int WL, int HL>
int launch(const void *coeffs, int coeff_stride, const void *pred, void *rec, int n,
           int tx_type, int bd, int inv_shift, hipStream_t st) {
  typedef typename std::conditional<BPP == 1, int16_t, int32_t>::type CT;
  constexpr int W = 1 << WL, H = 1 << HL, P = W > H ? W : H, NC = 64 / P;
  const unsigned grid = (unsigned)((n + NC - 1) / NC);
  hipLaunchKernelGGL((k_inv_tx<BPP, WL, HL>), dim3(grid), dim3(64), 0, st,
                     (const CT *)coeffs, coeff_stride, pred, rec, n, tx_type, bd, inv_shift);
  R1_HIP_CHECK(hipGetLastError());
  return R1_OK;
}

}  // namespace

extern "C" int r1_inv_txfm_add_batch(r1_ctx *ctx, const void *coeffs, int coeff_stride,
                                     const void *pred, void *rec, int n, int tx_size,
                                     int tx_type, int bit_depth, int bytes_per_px,
                                     void *stream) {
  R1_REQUIRE(ctx);
  R1_REQUIRE(r1tx::valid_av1_transform(tx_size, tx_type));
  R1_REQUIRE(bit_depth == 8 || bit_depth == 10 || bit_depth == 12);
  R1_REQUIRE(bytes_per_px == 1 || bytes_per_px == 2);
  R1_REQUIRE((bytes_per_px == 1) == (bit_depth == 8));
  const int w = 1 << r1tx::kTxWLog2[tx_size], h = 1 << r1tx::kTxHLog2[tx_size];
  R1_REQUIRE(coeff_stride >= (w < 32 ? w : 32) * (h < 32 ? h : 32));
  if (n <= 0) return R1_OK;
  R1_REQUIRE(coeffs && pred && rec);
  hipStream_t st = (hipStream_t)stream;
  const int sh = r1itx::kInvShift[tx_size];
#define R1_ITX_CASE(ID, WL, HL)                                                         \
  case ID:                                                                              \
    return bytes_per_px == 1                                                            \
               ? launch<1, WL, HL>(coeffs, coeff_stride, pred, rec, n, tx_type,        \
                                   bit_depth, sh, st)                                   \
               : launch<2, WL, HL>(coeffs, coeff_stride, pred, rec, n, tx_type,        \
                                   bit_depth, sh, st);
  switch (tx_size) {
    R1_ITX_CASE(0, 2, 2) R1_ITX_CASE(1, 3, 3) R1_ITX_CASE(2, 4, 4)
    R1_ITX_CASE(3, 5, 5) R1_ITX_CASE(4, 6, 6) R1_ITX_CASE(5, 2, 3)
    R1_ITX_CASE(6, 3, 2) R1_ITX_CASE(7, 3, 4) R1_ITX_CASE(8, 4, 3)
    R1_ITX_CASE(9, 4, 5) R1_ITX_CASE(10, 5, 4) R1_ITX_CASE(11, 5, 6)
    R1_ITX_CASE(12, 6, 5) R1_ITX_CASE(13, 2, 4) R1_ITX_CASE(14, 4, 2)
    R1_ITX_CASE(15, 3, 5) R1_ITX_CASE(16, 5, 3) R1_ITX_CASE(17, 4, 6)
    R1_ITX_CASE(18, 6, 4)
  }
#undef R1_ITX_CASE
  return R1_EINVAL;
}
