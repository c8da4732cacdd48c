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
template <int BPP, int KIND>
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
    out[cand] = KIND == 2 ? (acc + 32) / 64 : acc;
  }
}

}  // namespace

extern "C" int r1_dist_scaled_batch(r1_ctx *ctx, int kind, const R1Plane *org,
                                    const R1Plane *ref, int w, int h,
                                    const R1DistCand *cands, int n, const uint32_t *scales,
                                    int scale_stride, int xdec, int ydec, uint64_t *out,
                                    void *stream) {
  R1_REQUIRE(ctx && org && ref);
  R1_REQUIRE(kind == R1_DIST_WSSE || kind == R1_DIST_CDEF);
  R1_REQUIRE(org->bytes_per_px == ref->bytes_per_px);
  R1_REQUIRE(org->bytes_per_px == 1 || org->bytes_per_px == 2);
  R1_REQUIRE(w >= 4 && h >= 4 && w <= 128 && h <= 128 && w % 4 == 0 && h % 4 == 0);
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
