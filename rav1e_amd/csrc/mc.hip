// mc.hip -- batched motion compensation: put_8tap / prep_8tap / mc_avg
// (reference: src/mc.rs:250-479; dispatch tables src/asm/x86/mc.rs:17-78).
//
// Mapping: a block is cut into slabs of P = min(w, 64) columns; one wave
// (workgroup of 64) owns 64 / P slabs.  The wave first stages each slab's
// (h+7) x (P+7) reference window into LDS with unaligned dword loads, then
// lane = (slab, column) runs the separable filter down its column with an
// 8-deep register window (one new LDS row per output row for the vertical
// taps).  Output blocks are dense (stride = w), so a row's P lanes store P
// consecutive samples.
#include "mc_common.hpp"

int r1_mc_fast_launch(bool prep, const R1Plane *ref, int w, int h, const R1McCand *cands, int n,
                      void *dst, hipStream_t st);

namespace {

template <int BPP, bool PREP>
__global__ __launch_bounds__(64) void k_mc(R1Plane ref, int w, int h,
                                           const R1McCand *__restrict__ cands,
                                           int n, void *__restrict__ dst_) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int P = w < 64 ? w : 64;
  const int spc = w / P;      // slabs per candidate
  const int NS = 64 / P;      // slabs per wave
  const int ws = (((P + 7) * BPP + 3) >> 2) << 2;
  const int lane = threadIdx.x;
  const int sl = lane / P, c = lane - sl * P;
  const long long slab = (long long)blockIdx.x * NS + sl;
  const long long cand = slab / spc;
  const int x0 = (int)(slab - cand * spc) * P;
  const bool live = cand < n;
  uint8_t *win = smem + (size_t)sl * (h + 7) * ws;
  R1McCand cd = {};
  if (live) {
    cd = cands[cand];
    r1mc::stage_window<BPP>(win, ws, ref, cd.rx + x0, cd.ry, P, h, c, P);
  }
  __syncthreads();
  if (!live) return;
  const size_t base = (size_t)cand * w * h + x0 + c;
  if constexpr (PREP) {
    int16_t *dst = (int16_t *)dst_ + base;
    r1mc::mc_column<BPP, true, 0>(win, ws, c, w, h, cd.col_frac, cd.row_frac,
                               cd.mode_x, cd.mode_y, ref.bit_depth,
                               [&](int r, int32_t v) { dst[(size_t)r * w] = (int16_t)v; });
  } else if constexpr (BPP == 1) {
    uint8_t *dst = (uint8_t *)dst_ + base;
    r1mc::mc_column<BPP, false, 0>(win, ws, c, w, h, cd.col_frac, cd.row_frac,
                                cd.mode_x, cd.mode_y, ref.bit_depth,
                                [&](int r, int32_t v) { dst[(size_t)r * w] = (uint8_t)v; });
  } else {
    uint16_t *dst = (uint16_t *)dst_ + base;
    r1mc::mc_column<BPP, false, 0>(win, ws, c, w, h, cd.col_frac, cd.row_frac,
                                cd.mode_x, cd.mode_y, ref.bit_depth,
                                [&](int r, int32_t v) { dst[(size_t)r * w] = (uint16_t)v; });
  }
}

template <int BPP>
__global__ __launch_bounds__(256) void k_avg(const int16_t *__restrict__ t1,
                                             const int16_t *__restrict__ t2,
                                             long long total, int bit_depth,
                                             void *__restrict__ dst_) {
  const int ib = r1mc::intermediate_bits(bit_depth);
  const int32_t maxv = (1 << bit_depth) - 1;
  const int32_t bias = bit_depth == 8 ? 0 : 8192 * 2;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total;
       i += (long long)gridDim.x * 256) {
    const int32_t v = r1mc::clamp_px(
        r1mc::round_shift((int32_t)t1[i] + (int32_t)t2[i] + bias, ib + 1), maxv);
    if constexpr (BPP == 1) ((uint8_t *)dst_)[i] = (uint8_t)v;
    else ((uint16_t *)dst_)[i] = (uint16_t)v;
  }
}

int mc_launch(bool prep, const R1Plane *ref, int w, int h, const R1McCand *cands,
              int n, void *dst, hipStream_t st) {
  R1_REQUIRE(ref && (ref->bytes_per_px == 1 || ref->bytes_per_px == 2));
  R1_REQUIRE(r1_is_pow2(w) && w >= 2 && w <= 128);
  R1_REQUIRE(h >= 2 && h <= 128 && (h & 1) == 0);
  if (n <= 0) return R1_OK;
  R1_REQUIRE(cands && dst);
  // block sizes that are transform sizes take the dot4 / dot2 path of the
  // fused kernel (rdo_cand.hip); the slab kernel below covers the rest
  // (w = 2, 128-wide / -high blocks, odd aspect ratios)
  {
    const int rc = r1_mc_fast_launch(prep, ref, w, h, cands, n, dst, st);
    if (rc <= 0) return rc;
  }
  const int bpp = ref->bytes_per_px;
  const int P = w < 64 ? w : 64, NS = 64 / P, spc = w / P;
  const int ws = (((P + 7) * bpp + 3) >> 2) << 2;
  const size_t lds = (size_t)NS * (h + 7) * ws;
  const long long slabs = (long long)n * spc;
  const unsigned grid = (unsigned)((slabs + NS - 1) / NS);
#define R1_MC_GO(BPP, PREP)                                                   \
  hipLaunchKernelGGL((k_mc<BPP, PREP>), dim3(grid), dim3(64), lds, st, *ref, \
                     w, h, cands, n, dst)
  if (bpp == 1) { if (prep) R1_MC_GO(1, true); else R1_MC_GO(1, false); }
  else { if (prep) R1_MC_GO(2, true); else R1_MC_GO(2, false); }
#undef R1_MC_GO
  R1_HIP_CHECK(hipGetLastError());
  return R1_OK;
}

}  // namespace

extern "C" int r1_mc_put_batch(r1_ctx *ctx, const R1Plane *ref, int w, int h,
                               const R1McCand *cands, int n, void *dst,
                               void *stream) {
  R1_REQUIRE(ctx);
  return mc_launch(false, ref, w, h, cands, n, dst, (hipStream_t)stream);
}

extern "C" int r1_mc_prep_batch(r1_ctx *ctx, const R1Plane *ref, int w, int h,
                                const R1McCand *cands, int n, int16_t *tmp,
                                void *stream) {
  R1_REQUIRE(ctx);
  return mc_launch(true, ref, w, h, cands, n, tmp, (hipStream_t)stream);
}

extern "C" int r1_mc_avg_batch(r1_ctx *ctx, const int16_t *tmp1,
                               const int16_t *tmp2, int w, int h, int n,
                               int bit_depth, int bytes_per_px, void *dst,
                               void *stream) {
  R1_REQUIRE(ctx);
  R1_REQUIRE(bytes_per_px == 1 || bytes_per_px == 2);
  R1_REQUIRE(bit_depth == 8 || bit_depth == 10 || bit_depth == 12);
  R1_REQUIRE(w > 0 && h > 0);
  if (n <= 0) return R1_OK;
  R1_REQUIRE(tmp1 && tmp2 && dst);
  const long long total = (long long)n * w * h;
  long long blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  if (bytes_per_px == 1)
    hipLaunchKernelGGL((k_avg<1>), dim3((unsigned)blocks), dim3(256), 0,
                       (hipStream_t)stream, tmp1, tmp2, total, bit_depth, dst);
  else
    hipLaunchKernelGGL((k_avg<2>), dim3((unsigned)blocks), dim3(256), 0,
                       (hipStream_t)stream, tmp1, tmp2, total, bit_depth, dst);
  R1_HIP_CHECK(hipGetLastError());
  return R1_OK;
}
