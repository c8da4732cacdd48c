// importance.hip -- update_block_importances on device (SURVEY.md 8f "N1";
// reference src/api/internal.rs:911-1068).
//
// After its SATD map (the map of r1_estimate_inter_costs) the reference hands,
// for every 8x8 importance block of the current frame,
//   amount = (intra_cost + future_importance) * (1 - inter_cost / intra_cost) / len
// to the up to four importance blocks of the REFERENCE frame that the block's
// motion-compensated position overlaps, weighted by overlap area -- a
// scatter-add in f32 whose additions happen in the raster order of the source
// blocks (top-left, top-right, bottom-left, bottom-right inside a block).
// f32 addition does not commute with reordering, so atomics would not
// reproduce the reference.  Three steps instead:
//   1  one thread per source block: the four (destination, amount * fraction)
//      pairs, every operation a single IEEE f32 operation (__f*_rn: the compiler
//      must not contract a * b + c, Rust does not);
//   2  a STABLE radix sort of the pairs by destination (rocPRIM through hipCUB):
//      inside a destination the pairs keep their (source, corner) order;
//   3  one thread per destination block: binary search for its run, sequential
//      f32 accumulation in that order.
// The result is bit-identical to the sequential loop for any motion field.
#include <hipcub/hipcub.hpp>

#include "common.hpp"

namespace {

constexpr long long U = 64;   // IMP_BLOCK_SIZE_IN_MV_UNITS: 8 pixels * 8 units per pixel

__global__ __launch_bounds__(256) void k_imp_pairs(const uint32_t *__restrict__ intra_costs,
                                                   const float *__restrict__ future,
                                                   const uint32_t *__restrict__ inter_costs,
                                                   const int16_t *__restrict__ mvs, int w, int h,
                                                   float flen, uint32_t *__restrict__ keys,
                                                   float *__restrict__ vals) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= w * h) return;
  const int x = i % w, y = i / w;
  const long long rx = (long long)x * U + mvs[2 * i + 1], ry = (long long)y * U + mvs[2 * i];
  const float inter = (float)inter_costs[i], intra = (float)intra_costs[i];
  float frac = 0.f;
  if (!(intra <= inter)) frac = __fsub_rn(1.f, __fdiv_rn(inter, intra));
  const float amount = __fdiv_rn(__fmul_rn(__fadd_rn(intra, future[i]), frac), flen);
  // floor to the block grid (the reference's `- (U - 1) if negative` before a truncating division)
  const long long tlx = (rx - (rx < 0 ? U - 1 : 0)) / U * U, tly = (ry - (ry < 0 ? U - 1 : 0)) / U * U;
  const long long ax0 = tlx + U - rx, ax1 = rx - tlx, ay0 = tly + U - ry, ay1 = ry - tly;
#pragma unroll
  for (int c = 0; c < 4; c++) {
    const long long dx = tlx / U + (c & 1), dy = tly / U + (c >> 1);
    const long long area = ((c & 1) ? ax1 : ax0) * ((c >> 1) ? ay1 : ay0);
    const bool in = dx >= 0 && dy >= 0 && dx < w && dy < h;
    keys[4 * i + c] = in ? (uint32_t)(dy * w + dx) : (uint32_t)(w * h);   // off-frame: sorts last
    vals[4 * i + c] = __fmul_rn(amount, __fdiv_rn((float)area, 4096.f));
  }
}

__global__ __launch_bounds__(256) void k_imp_accumulate(const uint32_t *__restrict__ keys,
                                                        const float *__restrict__ vals, int n_pairs,
                                                        int n_blocks, float *__restrict__ ref_imp) {
  const int d = blockIdx.x * 256 + threadIdx.x;
  if (d >= n_blocks) return;
  int lo = 0, hi = n_pairs;            // first pair with key >= d
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (keys[mid] < (uint32_t)d) lo = mid + 1; else hi = mid;
  }
  float acc = ref_imp[d];
  for (int j = lo; j < n_pairs && keys[j] == (uint32_t)d; j++) acc = __fadd_rn(acc, vals[j]);
  ref_imp[d] = acc;
}

struct ImpScratch { size_t keys_in, vals_in, keys_out, vals_out, temp, temp_bytes, total; };

int end_bit_for(int n_blocks) {
  int b = 1;
  while ((1ll << b) <= n_blocks) b++;   // keys go up to n_blocks (the off-frame sentinel)
  return b;
}

hipError_t layout(int n_blocks, ImpScratch &s) {
  const size_t n = (size_t)n_blocks * 4;
  size_t temp = 0;
  const hipError_t e = hipcub::DeviceRadixSort::SortPairs(
      nullptr, temp, (const uint32_t *)nullptr, (uint32_t *)nullptr, (const float *)nullptr,
      (float *)nullptr, (int)n, 0, end_bit_for(n_blocks), (hipStream_t)0);
  auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
  s.keys_in = 0;
  s.vals_in = s.keys_in + up(n * 4);
  s.keys_out = s.vals_in + up(n * 4);
  s.vals_out = s.keys_out + up(n * 4);
  s.temp = s.vals_out + up(n * 4);
  s.temp_bytes = temp;
  s.total = s.temp + up(temp);
  return e;
}

}  // namespace

extern "C" long long r1_update_block_importances_scratch_bytes(int w_in_imp_b, int h_in_imp_b) {
  if (w_in_imp_b <= 0 || h_in_imp_b <= 0 || (long long)w_in_imp_b * h_in_imp_b > (1 << 28)) return -1;
  ImpScratch s;
  if (layout(w_in_imp_b * h_in_imp_b, s) != hipSuccess) return -1;
  return (long long)s.total;
}

extern "C" int r1_update_block_importances(r1_ctx *ctx, const uint32_t *intra_costs,
                                           const float *future_importances,
                                           const uint32_t *inter_costs, const int16_t *mvs,
                                           int w_in_imp_b, int h_in_imp_b, int len,
                                           float *ref_importances, void *scratch,
                                           long long scratch_bytes, void *stream) {
  R1_REQUIRE(ctx);
  R1_REQUIRE(w_in_imp_b >= 0 && h_in_imp_b >= 0 && len >= 1);
  R1_REQUIRE((long long)w_in_imp_b * h_in_imp_b <= (1 << 28));
  const int nb = w_in_imp_b * h_in_imp_b;
  if (nb == 0) return R1_OK;
  R1_REQUIRE(intra_costs && future_importances && inter_costs && mvs && ref_importances && scratch);
  ImpScratch s;
  R1_HIP_CHECK(layout(nb, s));
  R1_REQUIRE(scratch_bytes >= (long long)s.total);
  R1_REQUIRE(((uintptr_t)scratch & 255) == 0);
  hipStream_t st = (hipStream_t)stream;
  uint8_t *b = (uint8_t *)scratch;
  uint32_t *keys_in = (uint32_t *)(b + s.keys_in), *keys_out = (uint32_t *)(b + s.keys_out);
  float *vals_in = (float *)(b + s.vals_in), *vals_out = (float *)(b + s.vals_out);
  const unsigned grid = (unsigned)((nb + 255) / 256);
  hipLaunchKernelGGL(k_imp_pairs, dim3(grid), dim3(256), 0, st, intra_costs, future_importances,
                     inter_costs, mvs, w_in_imp_b, h_in_imp_b, (float)len, keys_in, vals_in);
  size_t temp = s.temp_bytes;
  R1_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(b + s.temp, temp, keys_in, keys_out, vals_in, vals_out,
                                                  nb * 4, 0, end_bit_for(nb), st));
  hipLaunchKernelGGL(k_imp_accumulate, dim3(grid), dim3(256), 0, st, keys_out, vals_out, nb * 4, nb,
                     ref_importances);
  R1_HIP_CHECK(hipGetLastError());
  return R1_OK;
}
