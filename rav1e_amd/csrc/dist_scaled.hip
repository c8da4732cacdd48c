// dist_scaled.hip -- batched weighted SSE and cdef_dist with the per-block
// DistortionScale bias folded in, i.e. the reference's candidate-level
// pixel-domain distortions
//   sse_wxh        src/rdo.rs:177-224  -> get_weighted_sse  src/dist.rs:234-283
//   cdef_dist_wxh  src/rdo.rs:142-173  -> cdef_dist_kernel  src/dist.rs:302-372
//                                         apply_ssim_boost  src/activity.rs:159-186
//                                         ssim_boost_rsqrt  src/activity.rs:109-145
//   compute_bias = distortion_scale     src/rdo.rs:443-459 (one Q14 scale per
//                  8x8 luma importance block of the frame, resident in HBM)
//   DistortionScale::mul_u64            src/rdo.rs:613-615
//
// Mapping: one LANE per 8x8 tile of a candidate (w, h multiples of 4, so a
// tile is 8 or 4 wide/high); the tile's rows come in with unaligned vector
// loads; all statistics stay in registers; the fixed-point tail (area
// divisors, variance scaling, rsqrt polynomial) runs per lane in u32/u64.
// Tiles of a candidate are consecutive lanes -> segmented wave reduction of
// the u64 partials, one LDS hop for candidates with more than 64 tiles.
#include "common.hpp"
#include "dist_common.hpp"

namespace {

// KIND 2: weighted SSE, KIND 3: cdef_dist
template <int BPP, int KIND, bool RAW = false>
__global__ __launch_bounds__(256) void k_dist_scaled(
    R1Plane org, R1Plane ref, int w, int h, int tw, int tiles, int tpc_log2,
    const R1DistCand *__restrict__ cands, int n, const uint32_t *__restrict__ scales,
    int scale_stride, int xdec, int ydec, unsigned long long *__restrict__ out) {
  __shared__ unsigned long long wave_part[4];
  const int tid = threadIdx.x;
  const long long gt = (long long)blockIdx.x * 256 + tid;
  const int cand = (int)(gt >> tpc_log2);
  const int t = (int)(gt & ((1 << tpc_log2) - 1));
  const bool live = cand < n && t < tiles;
  unsigned long long acc = 0;
  if (live) {
    const R1DistCand c = cands[cand];
    const int tx = t % tw, ty = t / tw;
    const int x0 = tx * 8, y0 = ty * 8;
    const int kw = w - x0 < 8 ? w - x0 : 8, kh = h - y0 < 8 ? h - y0 : 8;
    const uint8_t *po = px_addr<BPP>(org, c.ox + x0, c.oy + y0);
    const uint8_t *pr = px_addr<BPP>(ref, c.rx + x0, c.ry + y0);
    const size_t so = (size_t)org.stride * BPP, sr = (size_t)ref.stride * BPP;
    acc = r1dist::tile_scaled_dist<BPP, KIND>(po, so, pr, sr, kw, kh, c.ox + x0, c.oy + y0, scales,
                                              scale_stride, xdec, ydec, org.bit_depth);
  }
  // segmented u64 reduction over the candidate's tiles
  const int seg = tpc_log2 < 6 ? tpc_log2 : 6;
  for (int m = 1; m < (1 << seg); m <<= 1) {
    const uint32_t lo = __shfl_xor((uint32_t)acc, m, WAVE);
    const uint32_t hi = __shfl_xor((uint32_t)(acc >> 32), m, WAVE);
    acc += ((unsigned long long)hi << 32) | lo;
  }
  if (tpc_log2 > 6) {
    if ((tid & 63) == 0) wave_part[tid >> 6] = acc;
    __syncthreads();
    if (t == 0) {
      acc = 0;
      for (int i = 0; i < (1 << (tpc_log2 - 6)); i++) acc += wave_part[(tid >> 6) + i];
    }
  }
  if (cand < n && t == 0) {
    // get_weighted_sse's tail: den = DistortionScale::new(1, 256).0 = 64
    // RAW: what the reference's WeightedSseFn asm returns (the caller divides, sse.rs:123-131)
    out[cand] = (KIND == 2 && !RAW) ? (acc + 32) / 64 : acc;
  }
}

// cdef_dist_kernel's three values as the reference's CdefDistKernelFn asm returns them
// (src/asm/x86/dist/cdef_dist.rs:18-24,100-118: [svar, dvar, sse] before apply_ssim_boost):
// one lane per candidate, a single tile of w x h <= 8 x 8.
template <int BPP>
__global__ __launch_bounds__(64) void k_cdef_dist_raw(R1Plane org, R1Plane ref, int w, int h,
                                                      const R1DistCand *__restrict__ cands, int n,
                                                      uint32_t *__restrict__ out3) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  const R1DistCand c = cands[i];
  const uint8_t *po = px_addr<BPP>(org, c.ox, c.oy);
  const uint8_t *pr = px_addr<BPP>(ref, c.rx, c.ry);
  const size_t so = (size_t)org.stride * BPP, sr = (size_t)ref.stride * BPP;
  uint32_t sum_s = 0, sum_d = 0, sum_s2 = 0, sum_d2 = 0, sum_sd = 0;
  for (int r = 0; r < h; r++)
    for (int x = 0; x < w; x++) {
      const uint32_t s = (uint32_t)ld_px<BPP>(po + r * so + x * BPP), d = (uint32_t)ld_px<BPP>(pr + r * sr + x * BPP);
      sum_s += s; sum_d += d;
      sum_s2 += s * s; sum_d2 += d * d; sum_sd += s * d;
    }
  const uint32_t sse = sum_d2 + sum_s2 - 2 * sum_sd;
  const unsigned long long div = r1dist::area_divisor(w * h);
  const uint32_t ms = (uint32_t)(((unsigned long long)sum_s * sum_s * div + 8192) >> 14);
  const uint32_t md = (uint32_t)(((unsigned long long)sum_d * sum_d * div + 8192) >> 14);
  uint32_t svar = sum_s2 > ms ? sum_s2 - ms : 0;
  uint32_t dvar = sum_d2 > md ? sum_d2 - md : 0;
  svar = (uint32_t)(((unsigned long long)svar * div + 128) >> 8);
  dvar = (uint32_t)(((unsigned long long)dvar * div + 128) >> 8);
  out3[3 * i] = svar;
  out3[3 * i + 1] = dvar;
  out3[3 * i + 2] = sse;
}

}  // namespace

// internal (ctx.hip's per-table-entry shims): raw weighted SSE with one scale per 4x4 CELL of
// the block (get_weighted_sse's own indexing), and cdef_dist_kernel's raw triple
int r1_internal_wsse_raw(const R1Plane *a, const R1Plane *b, int w, int h, const R1DistCand *cand,
                         const uint32_t *scale, int scale_stride, uint64_t *out, hipStream_t st) {
  const int tw = (w + 7) / 8, th = (h + 7) / 8, tiles = tw * th;
  const int tpc_log2 = r1_ilog2(tiles);
  const unsigned grid = (unsigned)(((1ll << tpc_log2) + 255) / 256);
  // xdec = ydec = 1 makes the scale lookup (x << 1) >> 3 = x >> 2: per 4x4 cell of the block at (0, 0)
  if (a->bytes_per_px == 1)
    hipLaunchKernelGGL((k_dist_scaled<1, 2, true>), dim3(grid), dim3(256), 0, st, *a, *b, w, h, tw, tiles,
                       tpc_log2, cand, 1, scale, scale_stride, 1, 1, (unsigned long long *)out);
  else
    hipLaunchKernelGGL((k_dist_scaled<2, 2, true>), dim3(grid), dim3(256), 0, st, *a, *b, w, h, tw, tiles,
                       tpc_log2, cand, 1, scale, scale_stride, 1, 1, (unsigned long long *)out);
  R1_HIP_CHECK(hipGetLastError());
  return R1_OK;
}
int r1_internal_cdef_dist_raw(const R1Plane *a, const R1Plane *b, int w, int h, const R1DistCand *cand,
                              uint32_t *out3, hipStream_t st) {
  if (a->bytes_per_px == 1)
    hipLaunchKernelGGL((k_cdef_dist_raw<1>), dim3(1), dim3(64), 0, st, *a, *b, w, h, cand, 1, out3);
  else
    hipLaunchKernelGGL((k_cdef_dist_raw<2>), dim3(1), dim3(64), 0, st, *a, *b, w, h, cand, 1, out3);
  R1_HIP_CHECK(hipGetLastError());
  return R1_OK;
}

extern "C" int r1_dist_scaled_batch(r1_ctx *ctx, int kind, const R1Plane *org,
                                    const R1Plane *ref, int w, int h,
                                    const R1DistCand *cands, int n, const uint32_t *scales,
                                    int scale_stride, int xdec, int ydec, uint64_t *out,
                                    void *stream) {
  R1_REQUIRE(ctx && org && ref);
  R1_REQUIRE(kind == R1_DIST_WSSE || kind == R1_DIST_CDEF);
  R1_REQUIRE(org->bytes_per_px == ref->bytes_per_px);
  R1_REQUIRE(org->bytes_per_px == 1 || org->bytes_per_px == 2);
  R1_REQUIRE(w >= 1 && h >= 1 && w <= 128 && h <= 128);
  if (kind == R1_DIST_WSSE) {
    // get_weighted_sse walks whole 4x4 windows (vert_windows(4).step_by(4), dist.rs:248-262): what
    // is left of a clipped block beyond a multiple of 4 is NOT measured -- compute_tx_distortion
    // hands it such sizes at the frame edge (chroma of a frame whose width is 4 mod 8), and the
    // executed reference text (rdo_glue_ref.npz) returns the whole-cell sum, 0 when there is none
    w &= ~3;
    h &= ~3;
    if (w == 0 || h == 0) {
      if (n > 0) {
        R1_REQUIRE(out);
        R1_HIP_CHECK(hipMemsetAsync(out, 0, (size_t)n * sizeof(uint64_t), (hipStream_t)stream));
      }
      return R1_OK;
    }
  }
  // R1_DIST_CDEF: any w, h >= 1 -- cdef_dist_wxh tiles the visible block in 8x8 kernels and hands
  // cdef_dist_kernel whatever is left at the right / bottom (rdo.rs:152-165, AREA_DIVISORS[w * h - 1])
  R1_REQUIRE(xdec >= 0 && xdec <= 1 && ydec >= 0 && ydec <= 1);
  // cdef_dist is only defined on non-subsampled planes (rdo.rs:146-149)
  R1_REQUIRE(kind != R1_DIST_CDEF || (xdec == 0 && ydec == 0));
  R1_REQUIRE(!scales || scale_stride > 0);
  if (n <= 0) return R1_OK;
  R1_REQUIRE(cands && out);
  const int tw = (w + 7) / 8, th = (h + 7) / 8, tiles = tw * th;
  const int tpc_log2 = r1_ilog2(tiles);
  const long long lanes = (long long)n << tpc_log2;
  const unsigned grid = (unsigned)((lanes + 255) / 256);
  hipStream_t st = (hipStream_t)stream;
#define R1_DS_LAUNCH(BPP, KIND)                                                         \
  hipLaunchKernelGGL((k_dist_scaled<BPP, KIND>), dim3(grid), dim3(256), 0, st, *org,   \
                     *ref, w, h, tw, tiles, tpc_log2, cands, n, scales, scale_stride,  \
                     xdec, ydec, (unsigned long long *)out)
  if (org->bytes_per_px == 1) {
    if (kind == R1_DIST_WSSE) R1_DS_LAUNCH(1, 2); else R1_DS_LAUNCH(1, 3);
  } else {
    if (kind == R1_DIST_WSSE) R1_DS_LAUNCH(2, 2); else R1_DS_LAUNCH(2, 3);
  }
#undef R1_DS_LAUNCH
  R1_HIP_CHECK(hipGetLastError());
  return R1_OK;
}
