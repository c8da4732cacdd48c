// lookahead.hip -- whole-frame lookahead cost maps (SURVEY.md 8f "N1")
//   estimate_intra_costs                  src/api/lookahead.rs:30-123
//   estimate_importance_block_difference  src/api/lookahead.rs:125-180
//   estimate_inter_costs (SATD map)       src/api/lookahead.rs:226-268
//
// The reference walks the 8x8 importance blocks of a frame one by one:
// get_intra_edges on the SOURCE plane -> DC_PRED -> get_satd.  Nothing depends
// on a previous block, so the frame is one launch: one LANE per importance
// block.  The lane loads its 8x8 source tile with two unaligned dwordx2 /
// dwordx4 loads per row, subtracts the predictor (pred_dc_128 for every block: see k_intra_costs)
// and runs the 8x8 Hadamard in registers.
#include "common.hpp"
#include "dist_common.hpp"

namespace {

__device__ __forceinline__ void hadamard8(int32_t *d, int stride) {
  int32_t a[8], b[8];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    a[2 * k] = d[(2 * k) * stride] + d[(2 * k + 1) * stride];
    a[2 * k + 1] = d[(2 * k) * stride] - d[(2 * k + 1) * stride];
  }
  b[0] = a[0] + a[2]; b[2] = a[0] - a[2];
  b[1] = a[1] + a[3]; b[3] = a[1] - a[3];
  b[4] = a[4] + a[6]; b[6] = a[4] - a[6];
  b[5] = a[5] + a[7]; b[7] = a[5] - a[7];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    d[k * stride] = b[k] + b[k + 4];
    d[(k + 4) * stride] = b[k] - b[k + 4];
  }
}

// SATD of an 8x8 difference block held in registers (dist.rs:156-221, one tile)
__device__ __forceinline__ uint32_t satd8x8(int32_t *d) {
#pragma unroll
  for (int c = 0; c < 8; c++) hadamard8(d + c, 8);
#pragma unroll
  for (int r = 0; r < 8; r++) hadamard8(d + r * 8, 1);
  uint32_t s = 0;
#pragma unroll
  for (int i = 0; i < 64; i++) s += (uint32_t)iabs32(d[i]);
  return (s + 4) >> 3;
}

template <int BPP>
__global__ __launch_bounds__(64) void k_intra_costs(R1Plane p, int wb, int hb,
                                                    uint32_t *__restrict__ costs) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= wb * hb) return;
  const int bx = b % wb, by = b / wb;
  const int x = bx * 8, y = by * 8;
  const size_t st = (size_t)p.stride * BPP;
  const uint8_t *o = px_addr<BPP>(p, x, y);
  int32_t d[64];
#pragma unroll
  for (int r = 0; r < 8; r++) load_px_row<BPP, 8>(o + r * st, d + r * 8);
  // The predictor: estimate_intra_costs hands predict_intra a "tile" rectangle that starts AT the
  // block (TileRect { x: x * 8, y: y * 8, .. }, lookahead.rs:84-89), and predict_intra takes the
  // PredictionVariant from the block's position relative to that rectangle (predict.rs:212-218):
  // always (0, 0) -> PredictionVariant::NONE -> pred_dc_128 for EVERY block, the frame's interior
  // included.  The edges get_intra_edges gathered are not read.  (Pinned by executing the
  // reference's text: tests/golden/lookahead_ref.npz.)
  const uint32_t dc = 128u << (p.bit_depth - 8);
#pragma unroll
  for (int i = 0; i < 64; i++) d[i] -= (int32_t)dc;
  costs[b] = satd8x8(d);
}

template <int BPP>
__global__ __launch_bounds__(64) void k_inter_costs(R1Plane org, R1Plane ref, int wb, int hb,
                                                    const int16_t *__restrict__ mvs,
                                                    uint32_t *__restrict__ costs) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= wb * hb) return;
  const int bx = b % wb, by = b / wb;
  // reference position in MV units (1/8 pel), truncated toward zero like
  // Rust's `as isize / 8` (lookahead.rs:249-251)
  const int rx = (bx * 64 + mvs[2 * b + 1]) / 8, ry = (by * 64 + mvs[2 * b]) / 8;
  const uint8_t *o = px_addr<BPP>(org, bx * 8, by * 8);
  const uint8_t *q = px_addr<BPP>(ref, rx, ry);
  const size_t so = (size_t)org.stride * BPP, sr = (size_t)ref.stride * BPP;
  int32_t d[64];
#pragma unroll
  for (int r = 0; r < 8; r++) {
    int32_t a[8], c[8];
    load_px_row<BPP, 8>(o + r * so, a);
    load_px_row<BPP, 8>(q + r * sr, c);
#pragma unroll
    for (int i = 0; i < 8; i++) d[r * 8 + i] = a[i] - c[i];
  }
  costs[b] = satd8x8(d);
}

template <int BPP>
__global__ __launch_bounds__(256) void k_imp_diff(R1Plane org, R1Plane ref, int wb, int hb,
                                                  unsigned long long *__restrict__ out) {
  __shared__ unsigned long long part[4];
  const int b = blockIdx.x * 256 + threadIdx.x;
  unsigned long long v = 0;
  if (b < wb * hb) {
    const int bx = b % wb, by = b / wb;
    const uint8_t *o = px_addr<BPP>(org, bx * 8, by * 8), *q = px_addr<BPP>(ref, bx * 8, by * 8);
    const size_t so = (size_t)org.stride * BPP, sr = (size_t)ref.stride * BPP;
    int32_t s1 = 0, s2 = 0;
#pragma unroll
    for (int r = 0; r < 8; r++) {
      int32_t a[8], c[8];
      load_px_row<BPP, 8>(o + r * so, a);
      load_px_row<BPP, 8>(q + r * sr, c);
#pragma unroll
      for (int i = 0; i < 8; i++) { s1 += a[i]; s2 += c[i]; }
    }
    const int32_t dd = (s1 + 32) / 64 - (s2 + 32) / 64;
    v = (unsigned long long)(dd < 0 ? -dd : dd);
  }
  uint32_t lo = (uint32_t)v;              // per-block value < 2^12: 32-bit partial sums suffice
  lo = group_sum<64>(lo);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = lo;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, part[0] + part[1] + part[2] + part[3]);
}

// ActivityMask::from_plane + fill_scales (src/activity.rs:21-66): variance of
// every 8x8 luma block (the plane padded to a multiple of 8: the last blocks
// read the padding, as the reference's aligned rect does), then
// ssim_boost(var, var): the spatial half of the DistortionScale grid that
// r1_dist_scaled_batch and the pixel-domain candidate consume.
template <int BPP>
__global__ __launch_bounds__(256) void k_activity(R1Plane p, int wb, int hb,
                                                  uint32_t *__restrict__ variances,
                                                  uint32_t *__restrict__ scales) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= wb * hb) return;
  const int bx = i % wb, by = i / wb;
  const uint8_t *src = px_addr<BPP>(p, bx * 8, by * 8);
  const size_t st = (size_t)p.stride * BPP;
  unsigned long long sum_s = 0, sum_s2 = 0;
#pragma unroll
  for (int r = 0; r < 8; r++) {
    int32_t a[8];
    load_px_row<BPP, 8>(src + r * st, a);
#pragma unroll
    for (int k = 0; k < 8; k++) {
      sum_s += (uint32_t)a[k];
      sum_s2 += (uint32_t)a[k] * (uint32_t)a[k];
    }
  }
  // variance_8x8 (activity.rs:69-99): u32::try_from(..).unwrap_or(u32::MAX)
  const unsigned long long v = sum_s2 - ((sum_s * sum_s + 32) >> 6);
  const uint32_t var = v > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)v;
  if (variances) variances[i] = var;
  if (scales) scales[i] = r1dist::apply_ssim_boost(1u << 14, var, var, p.bit_depth);
}

}  // namespace

extern "C" int r1_estimate_intra_costs(r1_ctx *ctx, const R1Plane *luma, uint32_t *costs,
                                       void *stream) {
  R1_REQUIRE(ctx && luma && costs);
  R1_REQUIRE(luma->bytes_per_px == 1 || luma->bytes_per_px == 2);
  const int wb = luma->width / 8, hb = luma->height / 8;
  if (wb * hb == 0) return R1_OK;
  const unsigned grid = (unsigned)((wb * hb + 63) / 64);
  hipStream_t st = (hipStream_t)stream;
  if (luma->bytes_per_px == 1)
    hipLaunchKernelGGL((k_intra_costs<1>), dim3(grid), dim3(64), 0, st, *luma, wb, hb, costs);
  else
    hipLaunchKernelGGL((k_intra_costs<2>), dim3(grid), dim3(64), 0, st, *luma, wb, hb, costs);
  R1_HIP_CHECK(hipGetLastError());
  return R1_OK;
}

extern "C" int r1_estimate_inter_costs(r1_ctx *ctx, const R1Plane *org, const R1Plane *ref,
                                       const int16_t *mvs, uint32_t *costs, void *stream) {
  R1_REQUIRE(ctx && org && ref && mvs && costs);
  R1_REQUIRE(org->bytes_per_px == ref->bytes_per_px);
  R1_REQUIRE(org->bytes_per_px == 1 || org->bytes_per_px == 2);
  const int wb = org->width / 8, hb = org->height / 8;
  if (wb * hb == 0) return R1_OK;
  const unsigned grid = (unsigned)((wb * hb + 63) / 64);
  hipStream_t st = (hipStream_t)stream;
  if (org->bytes_per_px == 1)
    hipLaunchKernelGGL((k_inter_costs<1>), dim3(grid), dim3(64), 0, st, *org, *ref, wb, hb, mvs, costs);
  else
    hipLaunchKernelGGL((k_inter_costs<2>), dim3(grid), dim3(64), 0, st, *org, *ref, wb, hb, mvs, costs);
  R1_HIP_CHECK(hipGetLastError());
  return R1_OK;
}

extern "C" int r1_importance_block_difference(r1_ctx *ctx, const R1Plane *org, const R1Plane *ref,
                                              uint64_t *sum_out, void *stream) {
  R1_REQUIRE(ctx && org && ref && sum_out);
  R1_REQUIRE(org->bytes_per_px == ref->bytes_per_px);
  R1_REQUIRE(org->bytes_per_px == 1 || org->bytes_per_px == 2);
  const int wb = org->width / 8, hb = org->height / 8;
  hipStream_t st = (hipStream_t)stream;
  R1_HIP_CHECK(hipMemsetAsync(sum_out, 0, sizeof(uint64_t), st));
  if (wb * hb == 0) return R1_OK;
  const unsigned grid = (unsigned)((wb * hb + 255) / 256);
  if (org->bytes_per_px == 1)
    hipLaunchKernelGGL((k_imp_diff<1>), dim3(grid), dim3(256), 0, st, *org, *ref, wb, hb,
                       (unsigned long long *)sum_out);
  else
    hipLaunchKernelGGL((k_imp_diff<2>), dim3(grid), dim3(256), 0, st, *org, *ref, wb, hb,
                       (unsigned long long *)sum_out);
  R1_HIP_CHECK(hipGetLastError());
  return R1_OK;
}

extern "C" int r1_activity_scales(r1_ctx *ctx, const R1Plane *luma, uint32_t *variances,
                                  uint32_t *scales, void *stream) {
  R1_REQUIRE(ctx && luma && (variances || scales));
  R1_REQUIRE(luma->bytes_per_px == 1 || luma->bytes_per_px == 2);
  R1_REQUIRE((luma->bytes_per_px == 1) == (luma->bit_depth == 8));
  const int wb = (luma->width + 7) / 8, hb = (luma->height + 7) / 8;
  if (wb * hb == 0) return R1_OK;
  const unsigned grid = (unsigned)((wb * hb + 255) / 256);
  hipStream_t st = (hipStream_t)stream;
  if (luma->bytes_per_px == 1)
    hipLaunchKernelGGL((k_activity<1>), dim3(grid), dim3(256), 0, st, *luma, wb, hb, variances, scales);
  else
    hipLaunchKernelGGL((k_activity<2>), dim3(grid), dim3(256), 0, st, *luma, wb, hb, variances, scales);
  R1_HIP_CHECK(hipGetLastError());
  return R1_OK;
}
