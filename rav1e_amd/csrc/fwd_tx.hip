// fwd_tx.hip -- batched forward 2-D transform on dense int16 residual blocks
// (reference: forward_transform, src/transform/forward.rs:71-161; x86 entry
// src/asm/x86/transform/forward.rs:444-447).
//
// Mapping: one wave (workgroup of 64) owns NC = 64 / max(W,H) blocks.  Column
// pass: lane = (block, column), the H-point 1-D network runs entirely in that
// lane's registers.  The column results go through an LDS tile with an odd row
// stride (bank-conflict-free transpose), then lane = (block, row) runs the
// W-point row network in registers and stores straight into the reference's
// transposed, 32x32-chunked coefficient order (forward.rs:135-159): for a
// fixed column the H lanes of a block write H consecutive coefficients, so
// the global stores coalesce.
#include "tx_common.hpp"

namespace {
using r1tx::T;

template <int WL, int HL, typename CT>
__global__ __launch_bounds__(64) void k_fwd_tx(const int16_t *__restrict__ in,
                                               CT *__restrict__ out, int n,
                                               int tx_type, r1tx::Shift3 sh) {
  constexpr int W = 1 << WL, H = 1 << HL;
  constexpr int P = W > H ? W : H, NC = 64 / P;
  constexpr int LSTRIDE = NC * W + 1;
  // 64x64: the transpose goes through LDS in two halves of 32 rows (8.3 KB instead of
  // 16.6 KB per wave: 16 waves per CU instead of 9) -- lanes 0..31 pick up rows 0..31, then
  // rows 32..63 travel through the same bytes for lanes 32..63 (as k_rdo_cand does)
  constexpr bool SPLIT_T = W == 64 && H == 64;
  constexpr int HR = SPLIT_T ? H / 2 : H;
  __shared__ T buf[HR * LSTRIDE];
  const int lane = threadIdx.x;
  const int kcol = r1tx::vtx_1d(tx_type), krow = r1tx::htx_1d(tx_type);
  const bool ud = r1tx::ud_flip(tx_type), lr = r1tx::lr_flip(tx_type);

  // ---- columns ----
  const int ccl = lane / W, c = lane % W;
  const long long ccand = (long long)blockIdx.x * NC + ccl;
  const bool col_live = lane < NC * W && ccand < n;
  T v[H];
  if (col_live) {
    const int16_t *src = in + ccand * (W * H) + c;
#pragma unroll
    for (int r = 0; r < H; r++)
      v[r] = r1tx::shift_fwd((T)src[(ud ? H - 1 - r : r) * W], sh.s[0]);
    r1tx::fwd_1d<H>(v, kcol);
  }
  const int cc = ccl * W + (lr ? W - 1 - c : c);
  // ---- rows ----
  const int cl = lane / H, r = lane % H;
  const long long cand = (long long)blockIdx.x * NC + cl;
  const bool row_live = lane < NC * H && cand < n;
  T u[W];
#pragma unroll
  for (int half = 0; half < (SPLIT_T ? 2 : 1); half++) {
    if (col_live) {
#pragma unroll
      for (int rr = 0; rr < HR; rr++)
        buf[rr * LSTRIDE + cc] = r1tx::shift_fwd(v[half * HR + rr], sh.s[1]);
    }
    __syncthreads();
    if (row_live && r / HR == half) {
#pragma unroll
      for (int k = 0; k < W; k++) u[k] = buf[(r % HR) * LSTRIDE + cl * W + k];
    }
    if (SPLIT_T) __syncthreads();
  }
  if (row_live) {
    r1tx::fwd_1d<W>(u, krow);
    constexpr int OS = H < 32 ? H : 32, WC = W < 32 ? W : 32;
    CT *dst = out + cand * (W * H) + (r >= 32 ? OS * WC : 0) + (r & 31);
#pragma unroll
    for (int cg = 0; cg < W; cg += 32)
#pragma unroll
      for (int k = 0; k < WC; k++)
        __builtin_nontemporal_store((CT)r1tx::shift_fwd(u[k + cg], sh.s[2]), &dst[H * cg + k * OS]);   // streamed out, never re-read here
  }
}

template <int WL, int HL>
int launch(const int16_t *in, void *out, int n, int tx_type, r1tx::Shift3 sh,
           int coeff_bytes, hipStream_t st) {
  constexpr int W = 1 << WL, H = 1 << HL, P = W > H ? W : H, NC = 64 / P;
  const unsigned grid = (unsigned)((n + NC - 1) / NC);
  if (coeff_bytes == 2)
    hipLaunchKernelGGL((k_fwd_tx<WL, HL, int16_t>), dim3(grid), dim3(64), 0, st,
                       in, (int16_t *)out, n, tx_type, sh);
  else
    hipLaunchKernelGGL((k_fwd_tx<WL, HL, int32_t>), dim3(grid), dim3(64), 0, st,
                       in, (int32_t *)out, n, tx_type, sh);
  R1_HIP_CHECK(hipGetLastError());
  return R1_OK;
}

}  // namespace

extern "C" int r1_fwd_txfm_batch(r1_ctx *ctx, const int16_t *residual,
                                 void *coeffs, int n, int tx_size, int tx_type,
                                 int bit_depth, int coeff_bytes, void *stream) {
  R1_REQUIRE(ctx);
  R1_REQUIRE(r1tx::valid_av1_transform(tx_size, tx_type));
  R1_REQUIRE(bit_depth == 8 || bit_depth == 10 || bit_depth == 12);
  R1_REQUIRE(coeff_bytes == 2 || coeff_bytes == 4);
  if (n <= 0) return R1_OK;
  R1_REQUIRE(residual && coeffs);
  hipStream_t st = (hipStream_t)stream;
  const r1tx::Shift3 sh = r1tx::fwd_shift(tx_size, tx_type, bit_depth);
#define R1_TX_CASE(ID, WL, HL) \
  case ID: return launch<WL, HL>(residual, coeffs, n, tx_type, sh, coeff_bytes, st);
  switch (tx_size) {
    R1_TX_CASE(0, 2, 2) R1_TX_CASE(1, 3, 3) R1_TX_CASE(2, 4, 4)
    R1_TX_CASE(3, 5, 5) R1_TX_CASE(4, 6, 6) R1_TX_CASE(5, 2, 3)
    R1_TX_CASE(6, 3, 2) R1_TX_CASE(7, 3, 4) R1_TX_CASE(8, 4, 3)
    R1_TX_CASE(9, 4, 5) R1_TX_CASE(10, 5, 4) R1_TX_CASE(11, 5, 6)
    R1_TX_CASE(12, 6, 5) R1_TX_CASE(13, 2, 4) R1_TX_CASE(14, 4, 2)
    R1_TX_CASE(15, 3, 5) R1_TX_CASE(16, 5, 3) R1_TX_CASE(17, 4, 6)
    R1_TX_CASE(18, 6, 4)
  }
#undef R1_TX_CASE
  return R1_EINVAL;
}
