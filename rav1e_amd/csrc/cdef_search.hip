// cdef_search.hip -- the CDEF strength search of rdo_loop_decision
// (reference: src/rdo.rs:2104-2560, the CDEF leg with RestorationFilter::None /
// no restoration unit, rdo.rs:2432-2451 and 2504-2520; rdo_loop_plane_error
// rdo.rs:2027-2093; cdef_analyze_superblock / cdef_filter_superblock
// src/cdef.rs:340-560; cdef_dist_kernel src/dist.rs:302-372, get_weighted_sse 234-283).
//
// The reference filters every superblock of an analysis area once per cdef_index into a
// working copy and then measures the copy against the source, one 8x8 block at a time.
// Here ONE launch does all of that without ever materialising a filtered plane:
//   wave = one 8x8 luma block position (its luma block and both chroma blocks),
//   lane = pixel.  The direction search runs once; the twelve taps of a pixel are read
//   once for each of the (at most two) directions the index set can ask for (its own
//   direction, or 0 when a primary strength is 0) and stay in registers; their min / max
//   (the clamp of cdef.rs:284-292) is formed once per direction; then for every
//   cdef_index only the twelve `constrain` terms are evaluated, and the filtered pixel goes
//   straight into the block's distortion sums: the five moments of cdef_dist_kernel
//   (wave reduction, fixed-point tail + ssim boost + DistortionScale on one lane) for luma,
//   4x4-cell squared errors weighted by the block's DistortionScale for chroma.
//   The per-plane sums of a superblock meet by 64-bit integer atomics (order-free), and a
//   second, tiny kernel applies fi.dist_scale, adds the planes and picks the first minimum.
// The analysis area's borders count as picture edges exactly as on the reference's
// scratch copy (rdo.rs:2277-2284).  Integer arithmetic throughout.
#include "cdef_common.hpp"
#include "dist_common.hpp"

namespace {
using namespace r1cdef;

// constrain (cdef.rs:146-159) with the shift of this (threshold, damping) already formed
__device__ __forceinline__ int32_t constrain_s(int32_t diff, int32_t threshold, int shift) {
  const int32_t ad = diff < 0 ? -diff : diff;
  int32_t mag = threshold - (ad >> shift);
  mag = mag < 0 ? 0 : (mag > ad ? ad : mag);
  return diff < 0 ? -mag : mag;
}
__device__ __forceinline__ int constrain_shift(int threshold, int damping) {
  if (!threshold) return 0;
  const int s = damping - (31 - __clz(threshold));
  return s < 0 ? 0 : s;
}

// the twelve taps of one pixel for direction `dir`: t[0..1] primary k = 0, t[2..5] secondary
// k = 0, t[6..7] primary k = 1, t[8..11] secondary k = 1 (cdef.rs:255-283); mn / mx: the clamp
// range over the centre and the taps that exist
struct Taps { int32_t t[12], mn, mx; };
template <typename RD>
__device__ __forceinline__ Taps load_taps(RD rd, int i, int j, int dir, int32_t x) {
  constexpr uint32_t DY0 = 0x33332221u, DX0 = 0x22233333u, DY1 = 0x44443210u, DX1 = 0x12344444u;
  auto dyx = [&](int d, int k, int &dy, int &dx) {
    const int sh = 4 * d;
    dy = (int)(((k == 0 ? DY0 : DY1) >> sh) & 0xf) - 2;
    dx = (int)(((k == 0 ? DX0 : DX1) >> sh) & 0xf) - 2;
  };
  Taps tp;
  tp.mn = x;
  tp.mx = x;
#pragma unroll
  for (int k = 0; k < 2; k++) {
    int d0y, d0x, d1y, d1x, d2y, d2x;
    dyx(dir, k, d0y, d0x);
    dyx((dir + 2) & 7, k, d1y, d1x);
    dyx((dir + 6) & 7, k, d2y, d2x);
    int32_t *t = tp.t + 6 * k;
    t[0] = rd(i + d0y, j + d0x);
    t[1] = rd(i - d0y, j - d0x);
    t[2] = rd(i + d1y, j + d1x);
    t[3] = rd(i - d1y, j - d1x);
    t[4] = rd(i + d2y, j + d2x);
    t[5] = rd(i - d2y, j - d2x);
#pragma unroll
    for (int q = 0; q < 6; q++) {
      if (t[q] != VERY_LARGE && t[q] > tp.mx) tp.mx = t[q];
      if (t[q] < tp.mn) tp.mn = t[q];
    }
  }
  return tp;
}

__device__ __forceinline__ int32_t filter_from_taps(const Taps &tp, int32_t x, int pri, int sec, int pri_shift,
                                                    int sec_shift, int coeff_shift) {
  const int odd = (pri >> coeff_shift) & 1;
  int32_t sum = 0;
#pragma unroll
  for (int k = 0; k < 2; k++) {
    const int pri_tap = odd ? 3 : (k == 0 ? 4 : 2);
    const int sec_tap = k == 0 ? 2 : 1;
    const int32_t *t = tp.t + 6 * k;
    if (pri) sum += pri_tap * (constrain_s(t[0] - x, pri, pri_shift) + constrain_s(t[1] - x, pri, pri_shift));
    if (sec)
      sum += sec_tap * (constrain_s(t[2] - x, sec, sec_shift) + constrain_s(t[3] - x, sec, sec_shift) +
                        constrain_s(t[4] - x, sec, sec_shift) + constrain_s(t[5] - x, sec, sec_shift));
  }
  const int32_t v = x + ((8 + sum - (sum < 0)) >> 4);
  return v < tp.mn ? tp.mn : (v > tp.mx ? tp.mx : v);
}

struct SearchArgs {
  R1Plane rec[3], src[3];
  const uint8_t *skip_mi;
  int mi_stride, mi_cols, mi_rows;
  const uint32_t *scales;
  int scale_stride;
  R1CdefSearchParams p;
  int n_sbx, n_sby;
  unsigned long long *psum;   // [n_sb][8][3]
};

// geometry of the analysis area a superblock belongs to (rdo.rs:2149-2166, 2184-2190)
struct AreaGeo { int ax0, ay0, sbx, sby, area_w, area_h, blk_cols, blk_rows; };
__device__ __forceinline__ AreaGeo area_of(const SearchArgs &a, int fbx, int fby) {
  AreaGeo g;
  g.ax0 = fbx / a.p.area_sb_w * a.p.area_sb_w;
  g.ay0 = fby / a.p.area_sb_h * a.p.area_sb_h;
  g.sbx = fbx - g.ax0;
  g.sby = fby - g.ay0;
  const int sb_w = min(a.p.area_sb_w, a.n_sbx - g.ax0), sb_h = min(a.p.area_sb_h, a.n_sby - g.ay0);
  const int pixel_w = min(a.p.crop_w - g.ax0 * 64, sb_w * 64), pixel_h = min(a.p.crop_h - g.ay0 * 64, sb_h * 64);
  g.area_w = (pixel_w + 7) >> 3 << 3;
  g.area_h = (pixel_h + 7) >> 3 << 3;
  g.blk_cols = min(sb_w * 16, a.mi_cols - g.ax0 * 16);
  g.blk_rows = min(sb_h * 16, a.mi_rows - g.ay0 * 16);
  return g;
}

// are all 4x4 units of the superblock (inside the area's block grid) skipped?  wave-wide
__device__ __forceinline__ bool sb_all_skip(const SearchArgs &a, const AreaGeo &g, int lane) {
  // lane -> (row = lane >> 2, four units at column 4 * (lane & 3))
  const int y = 16 * g.sby + (lane >> 2), x0 = 16 * g.sbx + 4 * (lane & 3);
  int all = 1;
  if (y < g.blk_rows) {
    const uint8_t *sk = a.skip_mi + (size_t)(g.ay0 * 16 + y) * a.mi_stride + g.ax0 * 16 + x0;
#pragma unroll
    for (int k = 0; k < 4; k++)
      if (x0 + k < g.blk_cols) all &= sk[k] & 1;
  }
  return __all(all);
}

template <int BPP, int XD, int YD>
__global__ __launch_bounds__(64) void k_cdef_search(SearchArgs a) {
  __shared__ int32_t part[128];
  const int lane = threadIdx.x;
  const int gbx = blockIdx.x, gby = blockIdx.y;          // 8x8 luma block of the frame grid
  const int fbx = gbx >> 3, fby = gby >> 3, bx = gbx & 7, by = gby & 7;
  const AreaGeo g = area_of(a, fbx, fby);
  const int mx = g.sbx * 16 + 2 * bx, my = g.sby * 16 + 2 * by;   // area block units
  if (!(mx < g.blk_cols && my < g.blk_rows)) return;
  if (sb_all_skip(a, g, lane)) return;
  const int bd = a.p.bit_depth, coeff_shift = bd - 8;
  const uint8_t *sk = a.skip_mi + (size_t)(g.ay0 * 16 + my) * a.mi_stride + g.ax0 * 16 + mx;
  const int skip = sk[0] & sk[1] & sk[a.mi_stride] & sk[a.mi_stride + 1] & 1;
  const int flx = fbx * 64 + 8 * bx, fly = fby * 64 + 8 * by;   // frame position, luma px
  // edge flags on the area frame (cdef.rs:441-459)
  const int xavail = g.area_w - g.sbx * 64, yavail = g.area_h - g.sby * 64;
  const int edges = ((g.sby > 0 || by > 0) ? HAVE_TOP : 0) | ((g.sbx > 0 || bx > 0) ? HAVE_LEFT : 0) |
                    ((by + 1 < (yavail >> 3)) ? HAVE_BOTTOM : 0) | ((bx + 1 < (xavail >> 3)) ? HAVE_RIGHT : 0);
  int dir = 0;
  uint32_t var = 0;
  if (!skip) {
    const int32_t lum = ldpx<BPP>(px_addr<BPP>(a.rec[0], flx + (lane & 7), fly + (lane >> 3)));
    uint32_t var_v;
    dir = __builtin_amdgcn_readfirstlane(find_dir_wave(lum, coeff_shift, part, var_v));
    var = (uint32_t)__builtin_amdgcn_readfirstlane((int)var_v);
  }
  const uint32_t bias = a.scales ? a.scales[(size_t)(fly >> 3) * a.scale_stride + (flx >> 3)] : (1u << 14);
  unsigned long long *ps = a.psum + (size_t)(fby * a.n_sbx + fbx) * 24;

  // ---- passes: 0 = luma; then the chroma planes (two per pass when subsampled) ----
  constexpr int CXS = 8 >> XD, CYS = 8 >> YD, CPX = CXS * CYS;
  constexpr int NPASS_C = CPX == 64 ? 2 : 1;
  const int npass = a.p.planes == 1 ? 1 : 1 + NPASS_C;
  for (int pass = 0; pass < npass; pass++) {
    const bool luma = pass == 0;
    const int xs = luma ? 8 : CXS, ys = luma ? 8 : CYS, npx = xs * ys;
    const int pl = luma ? 0 : (CPX == 64 ? pass : 1 + lane / CPX);
    const bool act = luma || CPX == 64 || lane < 2 * CPX;
    const int pli = act ? pl : 1;
    const int l = lane % npx, i = l / xs, j = l % xs;
    const R1Plane &rp = a.rec[pli], &sp = a.src[pli];
    const int px = luma ? flx : flx >> XD, py = luma ? fly : fly >> YD;
    const uint8_t *r0 = px_addr<BPP>(rp, px, py);
    const ptrdiff_t rstr = (ptrdiff_t)rp.stride * BPP;
    auto rd = [&](int yy, int xx) -> int32_t {
      const bool ok = (yy >= 0 || (edges & HAVE_TOP)) && (yy < ys || (edges & HAVE_BOTTOM)) &&
                      (xx >= 0 || (edges & HAVE_LEFT)) && (xx < xs || (edges & HAVE_RIGHT));
      return ok ? ldpx<BPP>(r0 + yy * rstr + (ptrdiff_t)xx * BPP) : VERY_LARGE;
    };
    const int32_t x = ldpx<BPP>(r0 + i * rstr + j * BPP);
    const int32_t s = ldpx<BPP>(px_addr<BPP>(sp, px + j, py + i));
    // the directions the index set can select: the block's own (mapped for 4:2:2 chroma), or 0
    const int own = luma ? dir : (XD != YD ? (int)((0x66654207u >> (4 * dir)) & 0xf) : dir);
    Taps ta = {}, tb = {};
    bool need_own = false, need_zero = false;
    for (int idx = 0; idx < a.p.n_idx; idx++) {
      const int st = luma ? a.p.y_strengths[idx] : a.p.uv_strengths[idx];
      if (st / 4 != 0) need_own = true; else need_zero = true;
    }
    if (!skip) {
      if (need_own) ta = load_taps(rd, i, j, own, x);
      if (need_zero) tb = own == 0 && need_own ? ta : load_taps(rd, i, j, 0, x);
    }
    uint32_t m_s = 0, m_s2 = 0;
    if (luma) {
      m_s = group_sum<64>((uint32_t)s);
      m_s2 = group_sum<64>((uint32_t)(s * s));
    }
    for (int idx = 0; idx < a.p.n_idx; idx++) {
      int32_t v = x;
      if (!skip) {
        const int st = luma ? a.p.y_strengths[idx] : a.p.uv_strengths[idx];
        const int pri_raw = st / 4;
        int sec_raw = st % 4;
        sec_raw += sec_raw == 3;
        const int damping = a.p.damping + coeff_shift - (luma ? 0 : 1);
        const int pri = luma ? adjust_strength(pri_raw << coeff_shift, (int)var) : pri_raw << coeff_shift;
        const int sec = sec_raw << coeff_shift;
        const int psh = constrain_shift(pri, damping), ssh = constrain_shift(sec, damping);
        // wave-uniform choice of the tap set
        v = pri_raw != 0 ? filter_from_taps(ta, x, pri, sec, psh, ssh, coeff_shift)
                         : filter_from_taps(tb, x, pri, sec, psh, ssh, coeff_shift);
      }
      if (luma) {
        // cdef_dist_kernel moments over the 64 pixels (dist.rs:316-345); the source's two do not
        // depend on the index (formed once, above the loop)
        const uint32_t m_d = group_sum<64>((uint32_t)v), m_d2 = group_sum<64>((uint32_t)(v * v)),
                       m_sd = group_sum<64>((uint32_t)(s * v));
        if (lane == 0) {
          const unsigned long long d = r1dist::cdef_tile_tail(m_s, m_d, m_s2, m_d2, m_sd, 64, 0, 0, &bias, 0, bd);
          atomicAdd(&ps[idx * 3 + 0], d);
        }
      } else {
        // sse_wxh with a constant bias: 4x4 cells, each (sse * bias + 128) >> 8, the block's
        // sum through get_weighted_sse's (sum + 32) / 64 (rdo.rs:177-224, dist.rs:234-283)
        const int32_t df = s - v;
        uint32_t c = act ? (uint32_t)(df * df) : 0u;
        // cell members: bits 0-1 of j and bits 0-1 of i of the lane index inside the plane
        constexpr int M0 = 1, M1 = 2, M2 = CXS == 8 ? 8 : 4, M3 = CXS == 8 ? 16 : 8;
        if constexpr (CXS == 4) {
          c = group_sum<16>(c);   // lanes 16 k .. 16 k + 15 are one cell: DPP inside the row
        } else {
          c += __shfl_xor(c, M0, 64);
          c += __shfl_xor(c, M1, 64);
          c += __shfl_xor(c, M2, 64);
          c += __shfl_xor(c, M3, 64);
        }
        const bool leader = (j & 3) == 0 && (i & 3) == 0;
        unsigned long long w = leader && act ? ((unsigned long long)c * bias + 128) >> 8 : 0ull;
        // the cells of a plane's block: 1 (4x4), 2 (4x8: rows), 4 (8x8)
        if constexpr (CXS == 8) w += (unsigned long long)__shfl_xor((long long)w, 4, 64);
        if constexpr (CYS == 8) w += (unsigned long long)__shfl_xor((long long)w, CXS == 8 ? 32 : 16, 64);
        if (act && l == 0) atomicAdd(&ps[idx * 3 + pl], (w + 32) >> 6);
      }
    }
  }
}

// one wave per superblock: Distortion * dist_scale per plane, the sum, the first minimum
__global__ __launch_bounds__(64) void k_cdef_search_final(SearchArgs a, unsigned long long *err, int8_t *best) {
  const int fbx = blockIdx.x, fby = blockIdx.y, lane = threadIdx.x;
  const AreaGeo g = area_of(a, fbx, fby);
  const bool skip = sb_all_skip(a, g, lane);
  const size_t sb = (size_t)fby * a.n_sbx + fbx;
  unsigned long long e = 0;
  if (lane < 8 && lane < a.p.n_idx && !skip)
    for (int pl = 0; pl < a.p.planes; pl++)
      e += ((unsigned long long)a.p.dist_scale[pl] * a.psum[sb * 24 + lane * 3 + pl] + 8192) >> 14;
  if (lane < 8) err[sb * 8 + lane] = e;
  // compute_rd_cost with rate 0 is the error as f64 (exact below 2^53): first strict minimum
  int b = 0;
  unsigned long long be = __shfl(e, 0, 64);
  for (int idx = 1; idx < a.p.n_idx; idx++) {
    const unsigned long long ei = __shfl(e, idx, 64);
    if ((double)ei < (double)be) { be = ei; b = idx; }
  }
  if (lane == 0) best[sb] = skip ? (int8_t)-1 : (int8_t)b;
}

}  // namespace

extern "C" long long r1_cdef_strength_search_scratch_bytes(int mi_cols, int mi_rows) {
  return (long long)((mi_cols + 15) / 16) * ((mi_rows + 15) / 16) * 24 * 8;
}

extern "C" int r1_cdef_strength_search(r1_ctx *ctx, const R1Plane *rec, const R1Plane *src,
                                       const uint8_t *skip_mi, int mi_stride, int mi_cols, int mi_rows,
                                       const uint32_t *scales, int scale_stride,
                                       const R1CdefSearchParams *params, uint64_t *err_out,
                                       int8_t *best_out, void *scratch, void *stream) {
  R1_REQUIRE(ctx && rec && src && skip_mi && params && err_out && best_out && scratch);
  const R1CdefSearchParams &p = *params;
  R1_REQUIRE(p.n_idx >= 1 && p.n_idx <= 8 && (p.planes == 1 || p.planes == 3));
  R1_REQUIRE(p.area_sb_w >= 1 && p.area_sb_h >= 1 && p.crop_w > 0 && p.crop_h > 0);
  R1_REQUIRE(p.bit_depth == 8 || p.bit_depth == 10 || p.bit_depth == 12);
  R1_REQUIRE(mi_cols > 0 && mi_rows > 0 && mi_stride >= mi_cols && (mi_cols & 1) == 0 && (mi_rows & 1) == 0);
  R1_REQUIRE(!scales || scale_stride > 0);
  const int np = p.planes;
  for (int k = 0; k < np; k++) {
    R1_REQUIRE(rec[k].bytes_per_px == rec[0].bytes_per_px && src[k].bytes_per_px == rec[0].bytes_per_px);
    R1_REQUIRE(rec[k].data && src[k].data);
  }
  R1_REQUIRE(rec[0].bytes_per_px == 1 || rec[0].bytes_per_px == 2);
  R1_REQUIRE((rec[0].bytes_per_px == 1) == (p.bit_depth == 8));
  R1_REQUIRE(np == 1 || (p.xdec == 1 && p.ydec == 1) || (p.xdec == 1 && p.ydec == 0) ||
             (p.xdec == 0 && p.ydec == 0));
  // the frame is allocated in whole 8x8 blocks (coded frame sizes are padded to 8)
  R1_REQUIRE(mi_cols * 4 <= rec[0].width + 7 && mi_rows * 4 <= rec[0].height + 7);
  R1DeviceGuard guard(ctx);
  SearchArgs a = {};
  for (int k = 0; k < 3; k++) {
    a.rec[k] = rec[k < np ? k : 0];
    a.src[k] = src[k < np ? k : 0];
  }
  a.skip_mi = skip_mi; a.mi_stride = mi_stride; a.mi_cols = mi_cols; a.mi_rows = mi_rows;
  a.scales = scales; a.scale_stride = scale_stride;
  a.p = p;
  a.n_sbx = (mi_cols + 15) / 16;
  a.n_sby = (mi_rows + 15) / 16;
  a.psum = (unsigned long long *)scratch;
  hipStream_t st = (hipStream_t)stream;
  R1_HIP_CHECK(hipMemsetAsync(scratch, 0, (size_t)r1_cdef_strength_search_scratch_bytes(mi_cols, mi_rows), st));
  const dim3 grid(a.n_sbx * 8, a.n_sby * 8);
  const int xd = np == 1 ? 1 : p.xdec, yd = np == 1 ? 1 : p.ydec;
#define R1_CS_LAUNCH(B, X, Y) hipLaunchKernelGGL((k_cdef_search<B, X, Y>), grid, dim3(64), 0, st, a)
#define R1_CS_DEC(B)                                      \
  do {                                                    \
    if (xd == 1 && yd == 1) R1_CS_LAUNCH(B, 1, 1);        \
    else if (xd == 1) R1_CS_LAUNCH(B, 1, 0);              \
    else R1_CS_LAUNCH(B, 0, 0);                           \
  } while (0)
  if (rec[0].bytes_per_px == 1) R1_CS_DEC(1);
  else R1_CS_DEC(2);
#undef R1_CS_DEC
#undef R1_CS_LAUNCH
  R1_HIP_CHECK(hipGetLastError());
  hipLaunchKernelGGL(k_cdef_search_final, dim3(a.n_sbx, a.n_sby), dim3(64), 0, st, a,
                     (unsigned long long *)err_out, best_out);
  R1_HIP_CHECK(hipGetLastError());
  return R1_OK;
}
