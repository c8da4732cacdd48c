// cdef_search.hip -- the CDEF strength search of rdo_loop_decision
// (reference: src/rdo.rs:2104-2560, the CDEF leg with RestorationFilter::None /
// no restoration unit, rdo.rs:2432-2451 and 2504-2520; rdo_loop_plane_error
// rdo.rs:2027-2093; cdef_analyze_superblock / cdef_filter_superblock
// src/cdef.rs:340-560; cdef_dist_kernel src/dist.rs:302-372, get_weighted_sse 234-283).
//
// The reference filters every superblock of an analysis area once per cdef_index into a
// working copy and then measures the copy against the source, one 8x8 block at a time.
// Here no filtered plane is ever materialised: one analysis launch (k_cdef_analyze, a thread per
// 8x8 luma block) + one launch per plane kind (k_cdef_search_pk: luma; both chroma planes) work
// out, for every block and every cdef_index, the filtered pixels in registers (packed pairs, see
// the kernel) and feed them straight into the block's distortion sums: the five moments of
// cdef_dist_kernel for luma, 4x4-cell squared errors weighted by the block's DistortionScale for
// chroma.  The per-plane sums of a superblock meet by 64-bit integer atomics (order-free), and a
// last, tiny kernel applies fi.dist_scale, adds the planes and picks the first minimum.
// The analysis area's borders count as picture edges exactly as on the reference's
// scratch copy (rdo.rs:2277-2284).  Integer arithmetic throughout.
#include "cdef_common.hpp"
#include "dist_common.hpp"

namespace {
using namespace r1cdef;

__device__ __forceinline__ int constrain_shift(int threshold, int damping) {
  if (!threshold) return 0;
  const int s = damping - (31 - __clz(threshold));
  return s < 0 ? 0 : s;
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
  const uint8_t *dirs;        // [n_sby * 8][nbx] of k_cdef_analyze (rec luma)
  const int32_t *vars;
  int nbx;
  // MODE 1 (a later pass of the CDEF leg, restoration choices in play) / MODE 2 (the final pass: the working copy)
  const uint8_t *sb_lrf;      // [n_sb] bit p: plane p of the superblock has a self-guided choice -> the trial is STORED
  const uint8_t *sb_sel;      // [n_sb] (null = all): superblocks this launch evaluates
  R1Plane trial[3];           // the trial planes of cdef_index 0; index i at data + i * trial_idx_bytes
  size_t trial_idx_bytes;
  R1Plane out[3];             // MODE 2
  const int8_t *index_sb;     // MODE 2: [n_sb] cdef_index per superblock, < 0 = not filtered
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
      if (x0 + k < g.blk_cols) all &= sk[k] != 0;
  }
  return __all(all);
}

// ---- the search for one plane (LUMA) or both chroma planes (blockIdx.z), packed pairs.
// Geometry as k_cdef_frame's (cdef.hip): workgroup = 32 x 16 pixels of the plane (always inside one
// superblock), wave = 16 x 8, lane = two horizontally adjacent pixels; here lane = hb*32 + row*4 +
// pair, so that the 32 lanes of an 8-pixel-wide half are one luma block and DPP sums stay inside
// it.  The tile in LDS holds CDEF_VERY_LARGE outside the ANALYSIS AREA's rectangle: that is what
// the reference's edge flags on its scratch copy of the area say block by block.
//  * per block (first lanes of wave 0, once): skip, bias, the six tap offsets of its direction,
//    and for every cdef_index the primary strength after adjust_strength, its shift and taps;
//    the secondary strength / shift of an index do not depend on the block (scalar registers);
//  * the twelve taps are read once per direction set (the block's own direction, and direction 0
//    for an index whose primary strength is 0) and kept in registers;
//  * the secondary sum of an index is reused while the next index has the same secondary strength
//    and direction set (rav1e's presets: 3 evaluations for 8 indices);
//  * nothing per-index runs on a single lane: the block / cell sums of every index go to LDS and
//    ONE pass of the workgroup does all tails (cdef_dist_kernel's fixed point, the weighted-SSE
//    scaling) and the 64-bit atomics, a (block, index) pair per thread.
constexpr int SR_REC = 48;   // dwords per block record
constexpr int ST_STRIDE = 40, ST_ROWS = 20, ST_X0 = 4, ST_Y0 = 2;   // tile: 20 dwords per row, conflict-free for 8 rows x 4 pairs

template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ uint32_t add_dpp_m(uint32_t v) {
  return v + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false);
}
// sum over the 32 lanes of a half-wave; the total is in its last lane (31 / 63)
__device__ __forceinline__ uint32_t half_sum_last(uint32_t v) {
  v = add_dpp<0xB1>(v);            // quad_perm [1,0,3,2]
  v = add_dpp<0x4E>(v);            // quad_perm [2,3,0,1]
  v = add_dpp<0x141>(v);           // row_half_mirror
  v = add_dpp<0x140>(v);           // row_mirror
  return add_dpp_m<0x142, 0xa>(v); // row_bcast15 into rows 1 and 3
}
__device__ __forceinline__ uint32_t swz_xor4(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, (4 << 10) | 0x1f);
}

// MODE 0: the search (first pass).  MODE 1: a later pass -- planes of a superblock whose restoration unit has a
// self-guided choice get their filtered pixels STORED per index (the restoration trial reads them, lrf.hip), the
// others are measured as in MODE 0.  MODE 2: cdef_filter_superblock with the superblock's chosen index into `out`
// (rdo.rs:2546-2560 "keep cdef output up to date"), everything else copied through: the CDEF working copy.
template <int BPP, int XD, int YD, bool LUMA, int MODE = 0>
__global__ __launch_bounds__(256) void k_cdef_search_pk(SearchArgs a) {
  __shared__ __attribute__((aligned(16))) uint16_t tile[ST_ROWS * ST_STRIDE];
  __shared__ __attribute__((aligned(16))) uint32_t rec[32 * SR_REC];
  __shared__ uint32_t acc[32 * 8 * 3 + 16];   // luma: [8 blocks][8 idx][d, d2, sd] + [8][s, s2]; chroma: [32 cells][8 idx]
  constexpr int xs = 8 >> XD, ys = 8 >> YD;
  constexpr int NBX = 32 / xs, NBY = 16 / ys, NB = NBX * NBY;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int pli = LUMA ? 0 : 1 + (int)blockIdx.z;
  const int cs = a.p.bit_depth - 8, bd = a.p.bit_depth;
  const int rx0 = blockIdx.x * 32, ry0 = blockIdx.y * 16;          // plane position of the region
  const int fbx = (rx0 << XD) >> 6, fby = (ry0 << YD) >> 6;        // its superblock
  if (fbx >= a.n_sbx || fby >= a.n_sby) return;
  const AreaGeo g = area_of(a, fbx, fby);
  const bool all_skip = sb_all_skip(a, g, lane);                   // workgroup-uniform
  const int sb_i = fby * a.n_sbx + fbx;
  if (MODE != 2 && all_skip) return;
  if (MODE == 1 && a.sb_sel && !a.sb_sel[sb_i]) return;
  const bool store = MODE == 2 || (MODE == 1 && ((a.sb_lrf[sb_i] >> pli) & 1));   // workgroup-uniform
  const int apply_idx = MODE == 2 ? (all_skip ? -1 : (int)a.index_sb[sb_i]) : 0;
  const R1Plane &rp = a.rec[pli], &sp = a.src[pli];
  const int n_idx = a.p.n_idx;
  const uint8_t *strengths = LUMA ? a.p.y_strengths : a.p.uv_strengths;
  const int damping = a.p.damping + cs - (LUMA ? 0 : 1);
  // cdef_directions (cdef.rs:225-234) packed as nibbles (value + 2)
  constexpr uint32_t DY0 = 0x33332221u, DX0 = 0x22233333u, DY1 = 0x44443210u, DX1 = 0x12344444u;
  auto off = [&](int d, int k) -> uint32_t {
    const int sh4 = 4 * (d & 7);
    const int dy = (int)(((k == 0 ? DY0 : DY1) >> sh4) & 0xf) - 2;
    const int dx = (int)(((k == 0 ? DX0 : DX1) >> sh4) & 0xf) - 2;
    return (uint32_t)((dy * ST_STRIDE + dx) * 2);
  };
  // ---- per-block records
  if (tid < NB) {
    const int gbx = blockIdx.x * NBX + tid % NBX, gby = blockIdx.y * NBY + tid / NBX;   // frame 8x8 grid
    const int bx = gbx & 7, by = gby & 7;
    const int mx = g.sbx * 16 + 2 * bx, my = g.sby * 16 + 2 * by;                        // area block units
    const bool in_grid = mx < g.blk_cols && my < g.blk_rows;
    int skip = 1, dir = 0, var = 0;
    uint32_t bias = 1u << 14;
    if (in_grid) {
      const uint8_t *sk = a.skip_mi + (size_t)(g.ay0 * 16 + my) * a.mi_stride + g.ax0 * 16 + mx;
      const uint32_t s0 = *(const uint16_t *)sk, s1 = *(const uint16_t *)(sk + a.mi_stride);
      dir = a.dirs[(size_t)gby * a.nbx + gbx];
      var = a.vars[(size_t)gby * a.nbx + gbx];
      if (a.scales) bias = a.scales[(size_t)gby * a.scale_stride + gbx];
      skip = r1cdef::skip4(s0, s1);
    }
    const int own = LUMA ? dir : (XD != YD ? (int)((0x66654207u >> (4 * dir)) & 0xf) : dir);
    uint32_t *r = rec + tid * SR_REC;
    r[0] = (in_grid ? 1u : 0u) | (skip ? 0u : 2u);
    r[1] = bias;
#pragma unroll
    for (int k = 0; k < 2; k++) {
      r[2 + k] = off(own, k);
      r[4 + k] = off(own + 2, k);
      r[6 + k] = off(own + 6, k);
    }
    for (int idx = 0; idx < 8; idx++) {
      const int st = idx < n_idx ? strengths[idx] : 0;
      const int pri_raw = st >> 2;
      const int pri = LUMA ? adjust_strength(pri_raw << cs, var) : pri_raw << cs;
      const uint32_t psh = (uint32_t)constrain_shift(pri, damping);
      const int odd = (pri >> cs) & 1;
      r[16 + 4 * idx + 0] = (uint32_t)pri * 0x10001u;
      r[16 + 4 * idx + 1] = psh * 0x10001u;
      r[16 + 4 * idx + 2] = (odd ? 3u : 4u) * 0x10001u;
      r[16 + 4 * idx + 3] = (odd ? 3u : 2u) * 0x10001u;
    }
  }
  // ---- the tile: the area's rectangle in plane pixels, CDEF_VERY_LARGE outside
  {
    const int ax = (g.ax0 * 64) >> XD, ay = (g.ay0 * 64) >> YD;
    const int ax1 = (g.ax0 * 64 + g.area_w) >> XD, ay1 = (g.ay0 * 64 + g.area_h) >> YD;
    if (tid < ST_ROWS * 10) {
      const int ty = tid / 10, tq = tid - ty * 10;
      const int py = ry0 - ST_Y0 + ty, px = rx0 - ST_X0 + 4 * tq;
      uint32_t lo = 0x80008000u, hi = 0x80008000u;
      if (py >= ay && py < ay1 && px >= ax && px < ax1) {
        const uint8_t *gp = px_addr<BPP>(rp, px, py);
        if constexpr (BPP == 1) {
          const uint32_t q = ld_u32(gp);
          lo = __builtin_amdgcn_perm(0, q, 0x0c010c00u);
          hi = __builtin_amdgcn_perm(0, q, 0x0c030c02u);
        } else {
          const U32x2 q = ld_u32x2(gp);
          lo = q.a;
          hi = q.b;
        }
      }
      *(uint2 *)(tile + ty * ST_STRIDE + 4 * tq) = make_uint2(lo, hi);
    }
  }
  __syncthreads();
  // ---- lane -> pixel pair: half-wave hb = 8 columns, row, pair
  const int hb = lane >> 5, row = (lane >> 2) & 7, pq = lane & 3;
  const int lx = (wave & 1) * 16 + hb * 8 + 2 * pq, ly = (wave >> 1) * 8 + row;
  const int blk = (ly / ys) * NBX + lx / xs;
  const uint32_t *rb = rec + blk * SR_REC;
  const uint32_t flags = rb[0];
  const bool in_grid = flags & 1, filt = (flags & 2) != 0 && (MODE != 2 || apply_idx >= 0);
  const uint32_t base = (uint32_t)(((ly + ST_Y0) * ST_STRIDE + lx + ST_X0) * 2);
  Pk x, s;
  x.u = *(const uint32_t *)((const uint8_t *)tile + base);
  s.u = 0;
  if (in_grid && !store) {
    const uint8_t *gp = px_addr<BPP>(sp, rx0 + lx, ry0 + ly);
    if constexpr (BPP == 1) s.u = __builtin_amdgcn_perm(0, (uint32_t) * (const uint16_t *)gp, 0x0c010c00u);
    else s.u = ld_u32(gp);
  }
  typedef __attribute__((address_space(3))) uint16_t LdsU16;
  const uint32_t tbase = (uint32_t)(uintptr_t)(LdsU16 *)tile + base;
  // which direction sets the index set asks for (scalar)
  bool need_own = false, need_zero = false;
  const int idx_lo = MODE == 2 ? (apply_idx < 0 ? 0 : apply_idx) : 0, idx_hi = MODE == 2 ? idx_lo + 1 : n_idx;
  for (int idx = idx_lo; idx < idx_hi; idx++) {
    const int st = strengths[idx];
    if ((st >> 2) != 0) need_own = true;
    else if ((st & 3) != 0) need_zero = true;
  }
  struct TapSet { uint32_t tp[12]; i16x2 mx; u16x2 mn; };
  auto load_set = [&](const uint32_t (&offs)[6]) {
    TapSet ts;
    uint32_t tlo[12], thi[12];
#pragma unroll
    for (int t = 0; t < 6; t++) {
      lds_two(tbase + offs[t], tlo[2 * t], thi[2 * t]);
      lds_two(tbase - offs[t], tlo[2 * t + 1], thi[2 * t + 1]);
    }
    lds_wait(tlo, thi);
    ts.mx = x.s;
    ts.mn = x.v;
#pragma unroll
    for (int t = 0; t < 12; t++) {
      // a skipped block is not filtered: all its taps read as the centre (every term 0, clamp = x)
      Pk p;
      p.u = filt ? (tlo[t] | (thi[t] << 16)) : x.u;
      ts.tp[t] = p.u;
      ts.mx = __builtin_elementwise_max(ts.mx, p.s);
      ts.mn = __builtin_elementwise_min(ts.mn, p.v);
    }
    return ts;
  };
  TapSet own = {}, zero = {};
  if (need_own) {
    const uint32_t offs[6] = {rb[2], rb[3], rb[4], rb[5], rb[6], rb[7]};
    own = load_set(offs);
  }
  if (need_zero) {
    const uint32_t offs[6] = {off(0, 0), off(0, 1), off(2, 0), off(2, 1), off(6, 0), off(6, 1)};
    zero = load_set(offs);
  }
  // the source's two moments of a luma block do not depend on the index
  if (LUMA && !store) {
    Pk one;
    one.u = 0x00010001u;
    const uint32_t m_s = half_sum_last(__builtin_amdgcn_udot2(s.v, one.v, 0u, false));
    const uint32_t m_s2 = half_sum_last(__builtin_amdgcn_udot2(s.v, s.v, 0u, false));
    if ((lane & 31) == 31) {
      acc[32 * 8 * 3 + (wave * 2 + hb) * 2 + 0] = m_s;
      acc[32 * 8 * 3 + (wave * 2 + hb) * 2 + 1] = m_s2;
    }
  }
  int sec_key = -1;
  i16x2 sec_sum = (i16x2)0;
  for (int idx = idx_lo; idx < idx_hi; idx++) {
    const int st = (MODE == 2 && apply_idx < 0) ? 0 : strengths[idx];
    const int pri_raw = st >> 2;
    int sec_raw = st & 3;
    sec_raw += sec_raw == 3;
    Pk v = x;
    if (pri_raw != 0 || sec_raw != 0) {          // scalar
      const bool use_own = pri_raw != 0;
      const TapSet &ts = use_own ? own : zero;
      // secondary taps: tp[4..11] ((dir + 2, k), (dir + 6, k)), weights 2 (k = 0) / 1 (k = 1)
      const int key = sec_raw * 2 + (use_own ? 1 : 0);
      if (key != sec_key) {
        sec_key = key;
        sec_sum = (i16x2)0;
        if (sec_raw != 0) {
          const uint32_t sec = (uint32_t)(sec_raw << cs), ssh = (uint32_t)constrain_shift((int)sec, damping);
          Pk sec2, ssh2;
          sec2.u = sec * 0x10001u;
          ssh2.u = ssh * 0x10001u;
          i16x2 s2 = (i16x2)0, s1 = (i16x2)0;
#pragma unroll
          for (int t = 4; t < 12; t++) {
            Pk p;
            p.u = ts.tp[t];
            const i16x2 c = constrain2(p.s - x.s, sec2.s, ssh2.v);
            if (((t >> 1) & 1) == 0) s2 += c; else s1 += c;
          }
          sec_sum = s2 + s2 + s1;
        }
      }
      i16x2 sum = sec_sum;
      if (pri_raw != 0) {
        const uint4 pr = *(const uint4 *)(rb + 16 + 4 * idx);
        Pk pri2, psh2, pt0, pt1, p0, p1, p2, p3;
        pri2.u = pr.x; psh2.u = pr.y; pt0.u = pr.z; pt1.u = pr.w;
        p0.u = ts.tp[0]; p1.u = ts.tp[1]; p2.u = ts.tp[2]; p3.u = ts.tp[3];
        sum += (constrain2(p0.s - x.s, pri2.s, psh2.v) + constrain2(p1.s - x.s, pri2.s, psh2.v)) * pt0.s;
        sum += (constrain2(p2.s - x.s, pri2.s, psh2.v) + constrain2(p3.s - x.s, pri2.s, psh2.v)) * pt1.s;
      }
      v.s = x.s + ((sum + (sum >> (i16x2)15) + (i16x2)8) >> (i16x2)4);
      v.s = __builtin_elementwise_min(__builtin_elementwise_max(v.s, (i16x2)ts.mn), ts.mx);
    }
    if (MODE != 0 && store) {
      // the filtered pair itself: the trial plane of this index / the working copy
      if (in_grid) {
        const R1Plane &dp = MODE == 2 ? a.out[pli] : a.trial[pli];
        uint8_t *d = (uint8_t *)px_addr<BPP>(dp, rx0 + lx, ry0 + ly) + (MODE == 2 ? (size_t)0 : (size_t)idx * a.trial_idx_bytes);
        if constexpr (BPP == 1) *(uint16_t *)d = (uint16_t)((v.u & 0xffu) | ((v.u >> 8) & 0xff00u));
        else *(uint32_t *)d = v.u;
      }
    } else if constexpr (LUMA) {
      // cdef_dist_kernel's moments of the filtered block (dist.rs:316-345)
      Pk one;
      one.u = 0x00010001u;
      const uint32_t m_d = half_sum_last(__builtin_amdgcn_udot2(v.v, one.v, 0u, false));
      const uint32_t m_d2 = half_sum_last(__builtin_amdgcn_udot2(v.v, v.v, 0u, false));
      const uint32_t m_sd = half_sum_last(__builtin_amdgcn_udot2(s.v, v.v, 0u, false));
      if ((lane & 31) == 31) {
        uint32_t *ap = acc + ((wave * 2 + hb) * 8 + idx) * 3;
        ap[0] = m_d;
        ap[1] = m_d2;
        ap[2] = m_sd;
      }
    } else {
      // squared error of the 4x4 cell: lanes pq & 1 (bit 0), row & 3 (bits 2, 3)
      const i16x2 df = s.s - v.s;
      uint32_t c = (uint32_t)__builtin_amdgcn_sdot2(df, df, 0, false);
      c = add_dpp<0xB1>(c);                 // quad_perm [1,0,3,2]
      c += swz_xor4(c);                     // lane ^ 4
      c = add_dpp<0x128>(c);                // row_ror:8 = lane ^ 8 inside the row
      if ((pq & 1) == 0 && (row & 3) == 0) acc[((ly >> 2) * 8 + (lx >> 2)) * 8 + idx] = c;
    }
  }
  if (MODE != 0 && store) return;                 // workgroup-uniform
  __syncthreads();
  // ---- one (block, index) pair per thread: tails and atomics
  unsigned long long *ps = a.psum + (size_t)(fby * a.n_sbx + fbx) * 24;
  const int tb = tid >> 3, tidx = tid & 7;
  if (tidx >= n_idx) return;
  if constexpr (LUMA) {
    if (tb >= 8) return;
    // luma block tb of the workgroup = (wave, hb): its record
    const int w = tb >> 1, h2 = tb & 1;
    const int blx = (w & 1) * 2 + h2, bly = w >> 1;
    const uint32_t *r = rec + (bly * NBX + blx) * SR_REC;
    if (!(r[0] & 1)) return;
    const uint32_t *ap = acc + (tb * 8 + tidx) * 3;
    const uint32_t bias = r[1];
    const unsigned long long d = r1dist::cdef_tile_tail(acc[32 * 8 * 3 + tb * 2], ap[0], acc[32 * 8 * 3 + tb * 2 + 1], ap[1],
                                                        ap[2], 64, 0, 0, &bias, 0, bd);
    atomicAdd(&ps[tidx * 3 + 0], d);
  } else {
    if (tb >= NB) return;
    const uint32_t *r = rec + tb * SR_REC;
    if (!(r[0] & 1)) return;
    const uint32_t bias = r[1];
    // sse_wxh with a constant bias: 4x4 cells, each (sse * bias + 128) >> 8, the block's sum through
    // get_weighted_sse's (sum + 32) / 64 (rdo.rs:177-224, dist.rs:234-283)
    const int cx0 = (tb % NBX) * (xs / 4), cy0 = (tb / NBX) * (ys / 4);
    unsigned long long w = 0;
#pragma unroll
    for (int cy = 0; cy < ys / 4; cy++)
#pragma unroll
      for (int cx = 0; cx < xs / 4; cx++)
        w += ((unsigned long long)acc[((cy0 + cy) * 8 + cx0 + cx) * 8 + tidx] * bias + 128) >> 8;
    atomicAdd(&ps[tidx * 3 + pli], (w + 32) >> 6);
  }
}

// one wave per superblock: Distortion * dist_scale per plane, the sum, the first minimum
__global__ __launch_bounds__(64) void k_cdef_search_final(SearchArgs a, unsigned long long *err, int8_t *best,
                                                          unsigned long long *err_planes) {
  const int fbx = blockIdx.x, fby = blockIdx.y, lane = threadIdx.x;
  const AreaGeo g = area_of(a, fbx, fby);
  const size_t sb = (size_t)fby * a.n_sbx + fbx;
  const bool skip = sb_all_skip(a, g, lane) || (a.sb_sel && !a.sb_sel[sb]);
  unsigned long long e = 0;
  if (lane < 8 && lane < a.p.n_idx && !skip)
    for (int pl = 0; pl < a.p.planes; pl++) {
      const unsigned long long ep = ((unsigned long long)a.p.dist_scale[pl] * a.psum[sb * 24 + lane * 3 + pl] + 8192) >> 14;
      if (err_planes) err_planes[sb * 24 + lane * 3 + pl] = ep;
      e += ep;
    }
  if (err_planes && lane < 8 && (skip || lane >= a.p.n_idx || a.p.planes == 1))
    for (int pl = (skip || lane >= a.p.n_idx) ? 0 : 1; pl < 3; pl++) err_planes[sb * 24 + lane * 3 + pl] = 0;
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

// sb_lrf[sb] |= 1 << plane for every trial unit
__global__ void k_trial_mark(const R1TrialUnit *__restrict__ units, int n0, int n1, int n2, int n_sb, uint32_t *mask_words) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n0 + n1 + n2) return;
  const int pl = i < n0 ? 0 : (i < n0 + n1 ? 1 : 2);
  const int sb = units[i].sb;
  if (sb >= 0 && sb < n_sb) atomicOr(mask_words + (sb >> 2), (1u << pl) << (8 * (sb & 3)));
}

// what the three entry points share: validation and the launch arguments
int search_args(r1_ctx *ctx, const R1Plane *rec, const R1Plane *src, const uint8_t *skip_mi, int mi_stride,
                int mi_cols, int mi_rows, const uint32_t *scales, int scale_stride, const R1CdefSearchParams *params,
                SearchArgs &a) {
  R1_REQUIRE(ctx && rec && src && skip_mi && params);
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
  a = SearchArgs{};
  for (int k = 0; k < 3; k++) {
    a.rec[k] = rec[k < np ? k : 0];
    a.src[k] = src[k < np ? k : 0];
  }
  a.skip_mi = skip_mi; a.mi_stride = mi_stride; a.mi_cols = mi_cols; a.mi_rows = mi_rows;
  a.scales = scales; a.scale_stride = scale_stride;
  a.p = p;
  a.n_sbx = (mi_cols + 15) / 16;
  a.n_sby = (mi_rows + 15) / 16;
  a.nbx = a.n_sbx * 8;
  return R1_OK;
}

template <int MODE>
int search_launch(const SearchArgs &a, hipStream_t st) {
  const int np = a.p.planes;
  const int xd = np == 1 ? 0 : a.p.xdec, yd = np == 1 ? 0 : a.p.ydec;
  const dim3 grid_y(a.n_sbx * 2, a.n_sby * 4), grid_c((a.n_sbx * 64 >> xd) / 32, (a.n_sby * 64 >> yd) / 16, 2);
#define R1_CS_LAUNCH(B)                                                                                   \
  do {                                                                                                    \
    hipLaunchKernelGGL((k_cdef_search_pk<B, 0, 0, true, MODE>), grid_y, dim3(256), 0, st, a);             \
    if (np == 3) {                                                                                        \
      if (xd == 1 && yd == 1) hipLaunchKernelGGL((k_cdef_search_pk<B, 1, 1, false, MODE>), grid_c, dim3(256), 0, st, a); \
      else if (xd == 1) hipLaunchKernelGGL((k_cdef_search_pk<B, 1, 0, false, MODE>), grid_c, dim3(256), 0, st, a);       \
      else hipLaunchKernelGGL((k_cdef_search_pk<B, 0, 0, false, MODE>), grid_c, dim3(256), 0, st, a);     \
    }                                                                                                     \
  } while (0)
  if (a.rec[0].bytes_per_px == 1) R1_CS_LAUNCH(1);
  else R1_CS_LAUNCH(2);
#undef R1_CS_LAUNCH
  R1_HIP_CHECK(hipGetLastError());
  return R1_OK;
}

// scratch layout: per-superblock sums [n_sb][8][3] u64 | var i32 per 8x8 block | dir u8 per 8x8 block |
// (trial only) plane mask u8 per superblock, padded to 256 B | trial planes [n_idx][Y | U | V]
struct ScratchMap {
  size_t n_sb, psum, vars, dirs, mask, planes, plane_bytes[3], idx_bytes, total;
};
ScratchMap scratch_map(int mi_cols, int mi_rows, int xdec, int ydec, int bpp, int n_idx, int planes) {
  ScratchMap m = {};
  const size_t n_sbx = (mi_cols + 15) / 16, n_sby = (mi_rows + 15) / 16;
  m.n_sb = n_sbx * n_sby;
  m.psum = 0;
  m.vars = m.n_sb * 24 * 8;
  m.dirs = m.vars + m.n_sb * 64 * 4;
  m.mask = m.dirs + m.n_sb * 64;
  m.planes = (m.mask + m.n_sb + 255) & ~(size_t)255;
  for (int k = 0; k < planes; k++) {
    const size_t w = (n_sbx * 64) >> (k ? xdec : 0), h = (n_sby * 64) >> (k ? ydec : 0);
    m.plane_bytes[k] = (w * h * bpp + 255) & ~(size_t)255;
    m.idx_bytes += m.plane_bytes[k];
  }
  m.total = m.planes + m.idx_bytes * (size_t)n_idx;
  return m;
}

}  // namespace

// lrf.hip: the restoration trial of one plane (sgr_tile lives there)
__attribute__((visibility("hidden")))
int r1i_sgr_trial_err_launch(const R1Plane &trial, size_t trial_idx_bytes, const R1Plane &cdef_cur, const R1Plane &src,
                             const R1TrialUnit *units, int n_units, int n_idx, int pli, int xdec, int ydec,
                             const uint32_t *scales, int scale_stride, unsigned long long *psum, int n_sb, hipStream_t st);

// scratch: the per-superblock sums [n_sb][8][3] u64, then (var i32, dir u8) per 8x8 block of the grid
extern "C" long long r1_cdef_strength_search_scratch_bytes(int mi_cols, int mi_rows) {
  const long long n_sb = (long long)((mi_cols + 15) / 16) * ((mi_rows + 15) / 16);
  return n_sb * 24 * 8 + n_sb * 64 * 5;
}

extern "C" int r1_cdef_strength_search(r1_ctx *ctx, const R1Plane *rec, const R1Plane *src,
                                       const uint8_t *skip_mi, int mi_stride, int mi_cols, int mi_rows,
                                       const uint32_t *scales, int scale_stride,
                                       const R1CdefSearchParams *params, uint64_t *err_out,
                                       int8_t *best_out, void *scratch, void *stream) {
  R1_REQUIRE(err_out && best_out && scratch);
  SearchArgs a;
  int rc = search_args(ctx, rec, src, skip_mi, mi_stride, mi_cols, mi_rows, scales, scale_stride, params, a);
  if (rc != R1_OK) return rc;
  R1DeviceGuard guard(ctx);
  a.psum = (unsigned long long *)scratch;
  hipStream_t st = (hipStream_t)stream;
  const size_t n_sb = (size_t)a.n_sbx * a.n_sby;
  R1_HIP_CHECK(hipMemsetAsync(scratch, 0, n_sb * 24 * 8, st));
  // cdef_analyze_superblock once for the frame (a thread per 8x8 block), shared by the planes
  int32_t *vars = (int32_t *)((uint8_t *)scratch + n_sb * 24 * 8);
  uint8_t *dirs = (uint8_t *)(vars + n_sb * 64);
  a.dirs = dirs;
  a.vars = vars;
  rc = cdef_analyze_launch(&rec[0], a.nbx, a.n_sby * 8, mi_cols, mi_rows, dirs, vars, st);
  if (rc != R1_OK) return rc;
  rc = search_launch<0>(a, st);
  if (rc != R1_OK) return rc;
  hipLaunchKernelGGL(k_cdef_search_final, dim3(a.n_sbx, a.n_sby), dim3(64), 0, st, a,
                     (unsigned long long *)err_out, best_out, (unsigned long long *)nullptr);
  R1_HIP_CHECK(hipGetLastError());
  return R1_OK;
}

extern "C" long long r1_cdef_lrf_trial_scratch_bytes(int mi_cols, int mi_rows, int xdec, int ydec,
                                                     int bytes_per_px, int n_idx, int planes) {
  if (mi_cols <= 0 || mi_rows <= 0 || n_idx < 1 || n_idx > 8 || (planes != 1 && planes != 3)) return -1;
  return (long long)scratch_map(mi_cols, mi_rows, xdec, ydec, bytes_per_px, n_idx, planes).total;
}

extern "C" int r1_cdef_lrf_trial_batch(r1_ctx *ctx, const R1Plane *rec, const R1Plane *cdef_cur, const R1Plane *src,
                                       const uint8_t *skip_mi, int mi_stride, int mi_cols, int mi_rows,
                                       const uint32_t *scales, int scale_stride, const R1CdefSearchParams *params,
                                       const R1TrialUnit *units, const int32_t *n_units, const uint8_t *sb_sel,
                                       uint64_t *err_out, uint64_t *err_planes_out, int8_t *best_out, void *scratch,
                                       void *stream) {
  R1_REQUIRE(err_out && best_out && scratch && n_units);
  SearchArgs a;
  int rc = search_args(ctx, rec, src, skip_mi, mi_stride, mi_cols, mi_rows, scales, scale_stride, params, a);
  if (rc != R1_OK) return rc;
  const int np = a.p.planes, bpp = rec[0].bytes_per_px;
  int n_tot = 0;
  for (int k = 0; k < 3; k++) {
    R1_REQUIRE(n_units[k] >= 0 && (k < np || n_units[k] == 0));
    n_tot += n_units[k];
  }
  R1_REQUIRE(n_tot == 0 || (units && cdef_cur));
  if (n_tot)
    for (int k = 0; k < np; k++)
      R1_REQUIRE(cdef_cur[k].data && cdef_cur[k].bytes_per_px == bpp);
  const ScratchMap m = scratch_map(mi_cols, mi_rows, np == 1 ? 0 : a.p.xdec, np == 1 ? 0 : a.p.ydec, bpp, a.p.n_idx, np);
  for (int k = 0; k < np; k++) R1_REQUIRE(m.plane_bytes[k] < (1ull << 32));   // 32-bit byte offsets in lrf.hip's tile loads
  R1DeviceGuard guard(ctx);
  hipStream_t st = (hipStream_t)stream;
  uint8_t *sc = (uint8_t *)scratch;
  a.psum = (unsigned long long *)(sc + m.psum);
  R1_HIP_CHECK(hipMemsetAsync(sc, 0, m.n_sb * 24 * 8, st));
  R1_HIP_CHECK(hipMemsetAsync(sc + m.mask, 0, m.planes - m.mask, st));
  a.vars = (const int32_t *)(sc + m.vars);
  a.dirs = sc + m.dirs;
  a.sb_lrf = sc + m.mask;
  a.sb_sel = sb_sel;
  size_t off = m.planes;
  for (int k = 0; k < np; k++) {
    R1Plane &t = a.trial[k];
    t = R1Plane{};
    t.data = sc + off;
    t.stride = (a.n_sbx * 64) >> (k ? a.p.xdec : 0);
    t.alloc_height = (a.n_sby * 64) >> (k ? a.p.ydec : 0);
    t.width = t.stride;
    t.height = t.alloc_height;
    t.bytes_per_px = bpp;
    t.bit_depth = a.p.bit_depth;
    off += m.plane_bytes[k];
  }
  a.trial_idx_bytes = m.idx_bytes;
  rc = cdef_analyze_launch(&rec[0], a.nbx, a.n_sby * 8, mi_cols, mi_rows, (uint8_t *)(sc + m.dirs), (int32_t *)(sc + m.vars), st);
  if (rc != R1_OK) return rc;
  if (n_tot)
    hipLaunchKernelGGL(k_trial_mark, dim3((n_tot + 255) / 256), dim3(256), 0, st, units, n_units[0], n_units[1],
                       n_units[2], (int)m.n_sb, (uint32_t *)(sc + m.mask));
  rc = search_launch<1>(a, st);
  if (rc != R1_OK) return rc;
  int first = 0;
  for (int k = 0; k < np; k++) {
    if (n_units[k]) {
      rc = r1i_sgr_trial_err_launch(a.trial[k], a.trial_idx_bytes, cdef_cur[k], a.src[k], units + first, n_units[k],
                                    a.p.n_idx, k, k ? a.p.xdec : 0, k ? a.p.ydec : 0, scales, scale_stride, a.psum, (int)m.n_sb, st);
      if (rc != R1_OK) return rc;
    }
    first += n_units[k];
  }
  hipLaunchKernelGGL(k_cdef_search_final, dim3(a.n_sbx, a.n_sby), dim3(64), 0, st, a,
                     (unsigned long long *)err_out, best_out, (unsigned long long *)err_planes_out);
  R1_HIP_CHECK(hipGetLastError());
  return R1_OK;
}

extern "C" int r1_cdef_apply_area(r1_ctx *ctx, const R1Plane *rec, const R1Plane *out, const uint8_t *skip_mi,
                                  int mi_stride, int mi_cols, int mi_rows, const R1CdefSearchParams *params,
                                  const int8_t *index_sb, void *scratch, void *stream) {
  R1_REQUIRE(out && index_sb && scratch);
  SearchArgs a;
  int rc = search_args(ctx, rec, rec, skip_mi, mi_stride, mi_cols, mi_rows, nullptr, 0, params, a);
  if (rc != R1_OK) return rc;
  for (int k = 0; k < a.p.planes; k++) {
    R1_REQUIRE(out[k].data && out[k].data != rec[k].data && out[k].bytes_per_px == rec[0].bytes_per_px);
    R1_REQUIRE(out[k].width >= rec[k].width && out[k].height >= rec[k].height);
    a.out[k] = out[k];
  }
  R1DeviceGuard guard(ctx);
  hipStream_t st = (hipStream_t)stream;
  const size_t n_sb = (size_t)a.n_sbx * a.n_sby;
  int32_t *vars = (int32_t *)((uint8_t *)scratch + n_sb * 24 * 8);
  uint8_t *dirs = (uint8_t *)(vars + n_sb * 64);
  a.dirs = dirs;
  a.vars = vars;
  a.index_sb = index_sb;
  rc = cdef_analyze_launch(&rec[0], a.nbx, a.n_sby * 8, mi_cols, mi_rows, dirs, vars, st);
  if (rc != R1_OK) return rc;
  return search_launch<2>(a, st);
}
