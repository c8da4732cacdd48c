// predict.hip -- batched intra prediction and intra-edge gathering
// (reference: rust::dispatch_predict_intra src/predict.rs:705-784 and the
// kernels it selects, 786-1505; get_intra_edges src/partition.rs:639-898;
// x86 dispatch src/asm/x86/predict.rs).
//
// k_intra_edges: one wave per candidate.  Every entry of the reference's
// IntraEdgeBuffer (left right-aligned ending at index 128, bottom->top;
// top-left at 128; above from 129) has a closed-form source -- a pixel of the
// reconstructed tile, a replicated pixel or a frame-edge base value -- so the
// 257 entries are produced independently by the 64 lanes (no serial fill).
// The partition-tree availability answers (has_top_right / has_bottom_left)
// are encoder state and arrive as flags.
//
// k_intra_predict: lane = (candidate, column), NC = 64 / W candidates per
// wave; the candidate's raw edge is staged in LDS.  Directional modes with the
// intra edge filter build the filtered / upsampled edges in LDS (ping-pong,
// all taps read the unfiltered copy exactly like filter_edge's scratch
// array), then every lane walks its column with the reference's index
// arithmetic.  Rows are stored as W consecutive pixels per candidate
// (coalesced).
#include "common.hpp"
#include "dist_common.hpp"

#define R1_TABLE_QUAL __constant__
#include "intra_tables.inc"

namespace {

enum { DC_PRED = 0, V_PRED, H_PRED, D45_PRED, D135_PRED, D113_PRED, D157_PRED,
       D203_PRED, D67_PRED, SMOOTH_PRED, SMOOTH_V_PRED, SMOOTH_H_PRED, PAETH_PRED,
       UV_CFL_PRED };
constexpr int MAXTX = 64;
constexpr int EDGE_LEN = 4 * MAXTX + 1;

__device__ __forceinline__ int mode_angle(int mode) {
  constexpr int16_t a[9] = {0, 90, 180, 45, 135, 113, 157, 203, 67};
  return mode >= 0 && mode < 9 ? a[mode] : 0;
}
__device__ __forceinline__ int iabs(int v) { return v < 0 ? -v : v; }

// select_ief_strength / select_ief_upsample (predict.rs:1133-1201)
__device__ __forceinline__ int ief_strength(int wh, bool smooth, int delta) {
  const int d = iabs(delta);
  if (smooth) {
    if (wh <= 8) return d >= 64 ? 2 : (d >= 40 ? 1 : 0);
    if (wh <= 16) return d >= 48 ? 2 : (d >= 20 ? 1 : 0);
    if (wh <= 24) return d >= 4 ? 3 : 0;
    return 3;
  }
  if (wh <= 8) return d >= 56 ? 1 : 0;
  if (wh <= 16) return d >= 40 ? 1 : 0;
  if (wh <= 24) return d >= 32 ? 3 : (d >= 16 ? 2 : (d >= 8 ? 1 : 0));
  if (wh <= 32) return d >= 32 ? 3 : (d >= 4 ? 2 : 1);
  return 3;
}
__device__ __forceinline__ bool ief_upsample(int wh, bool smooth, int delta) {
  const int d = iabs(delta);
  if (d == 0 || d >= 40) return false;
  return smooth ? wh <= 8 : wh <= 16;
}

template <int BPP>
__device__ __forceinline__ int32_t ldp(const void *p, size_t i) {
  if constexpr (BPP == 1) return ((const uint8_t *)p)[i];
  else return ((const uint16_t *)p)[i];
}
template <int BPP>
__device__ __forceinline__ void stp(void *p, size_t i, int32_t v) {
  if constexpr (BPP == 1) ((uint8_t *)p)[i] = (uint8_t)v;
  else ((uint16_t *)p)[i] = (uint16_t)v;
}

// ------------------------------------------------------------------ edges
template <int BPP>
__global__ __launch_bounds__(64) void k_intra_edges(
    R1Plane rec, int tile_x, int tile_y, int rect_w, int rect_h, int txw, int txh, int lpc_log2,
    const R1IntraEdgeCand *__restrict__ cands, int n, void *__restrict__ edges,
    int edge_stride, uint8_t *__restrict__ lens) {
  // lpc = lanes per candidate (power of two covering 2*(txw+txh)+1 entries, <= 64)
  const int lpc = 1 << lpc_log2, cpw = 64 >> lpc_log2;
  const int cand = blockIdx.x * cpw + (threadIdx.x >> lpc_log2);
  const int l = threadIdx.x & (lpc - 1);
  if (cand >= n) return;
  const R1IntraEdgeCand cd = cands[cand];
  const int x = cd.x, y = cd.y, bd = rec.bit_depth;
  const int32_t base = 128 << (bd - 8);
  bool needs_left = true, needs_topleft = true, needs_top = true, needs_topright = true,
       needs_bottomleft = true, tl_filter = false;
  if (cd.mode >= 0) {
    int mode = cd.mode;
    if (mode == PAETH_PRED)
      mode = (x == 0 && y == 0) ? DC_PRED : (x == 0 ? V_PRED : (y == 0 ? H_PRED : PAETH_PRED));
    const bool directional = mode >= V_PRED && mode <= D67_PRED;
    const int p_angle = mode_angle(mode) + cd.angle_delta * 3;
    const bool dc_or_cfl = mode == DC_PRED || mode == UV_CFL_PRED;
    needs_left = (!dc_or_cfl || x != 0) || (p_angle > 90 && p_angle != 180);
    needs_topleft = mode == PAETH_PRED || (directional && p_angle != 90 && p_angle != 180);
    needs_top = (!dc_or_cfl || y != 0) || (p_angle != 90 && p_angle < 180);
    needs_topright = directional && p_angle < 90;
    needs_bottomleft = directional && p_angle > 180;
    tl_filter = (cd.flags & 1) && p_angle > 90 && p_angle < 180;
  }
  const uint8_t *t0 = px_addr<BPP>(rec, tile_x, tile_y);
  const size_t st = (size_t)rec.stride;
  auto DST = [&](int yy, int xx) -> int32_t { return ldp<BPP>(t0, (size_t)yy * st + xx); };
  const int th = y + txh > rect_h ? rect_h - y : txh;     // visible rows / cols of the block
  const int tw = x + txw > rect_w ? rect_w - x : txw;
  int tr_avail = 0, bl_avail = 0;
  if (needs_topright && y != 0 && (cd.flags & 2)) {
    tr_avail = rect_w - x - txw;
    tr_avail = tr_avail > txw ? txw : (tr_avail < 0 ? 0 : tr_avail);
  }
  if (needs_bottomleft && x != 0 && (cd.flags & 4)) {
    bl_avail = rect_h - y - txh;
    bl_avail = bl_avail > txh ? txh : (bl_avail < 0 ? 0 : bl_avail);
  }
  // left[i], i = distance below the block's top row (index 127 - i)
  auto left_at = [&](int i) -> int32_t {
    if (i < txh) {
      if (x != 0) return DST(y + (i < th ? i : th - 1), x - 1);
      return y != 0 ? DST(y - 1, 0) : base + 1;
    }
    const int k = i - txh;                                // bottom-left part
    if (k < bl_avail) return DST(y + txh + k, x - 1);
    // replicate left[2*MAX - txh - num_avail] = the entry just above
    const int j = txh + bl_avail - 1;
    if (j >= txh) return DST(y + txh + bl_avail - 1, x - 1);
    if (x != 0) return DST(y + (j < th ? j : th - 1), x - 1);
    return y != 0 ? DST(y - 1, 0) : base + 1;
  };
  auto above_at = [&](int i) -> int32_t {
    if (i < txw) {
      if (y != 0) return DST(y - 1, x + (i < tw ? i : tw - 1));
      return x != 0 ? DST(0, x - 1) : base - 1;
    }
    const int k = i - txw;                                // top-right part
    if (k < tr_avail) return DST(y - 1, x + txw + k);
    const int j = txw + tr_avail - 1;
    if (j >= txw) return DST(y - 1, x + txw + tr_avail - 1);
    if (y != 0) return DST(y - 1, x + (j < tw ? j : tw - 1));
    return x != 0 ? DST(0, x - 1) : base - 1;
  };
  const int init_left = (needs_left ? txh : 0) + (needs_bottomleft ? txw : 0);
  const int init_above = (needs_top ? txw : 0) + (needs_topright ? txh : 0);
  void *e = (uint8_t *)edges + (size_t)cand * edge_stride * BPP;
  // only the initialised range [128 - init_left, 129 + init_above) is written;
  // the reference leaves the rest of the buffer uninitialised as well
  for (int k = 2 * MAXTX - init_left + l; k < 2 * MAXTX + 1 + init_above; k += lpc) {
    int32_t v = 0;
    if (k < 2 * MAXTX) {
      const int i = 2 * MAXTX - 1 - k;
      // bottom-left entries exist only behind a needed left column; the
      // reference never builds one without the other
      if (i < init_left && needs_left) v = left_at(i);
    } else if (k == 2 * MAXTX) {
      if (needs_topleft) {
        v = (x == 0 && y == 0) ? base
            : (y == 0 ? DST(0, x - 1) : (x == 0 ? DST(y - 1, 0) : DST(y - 1, x - 1)));
        if (tl_filter && txw + txh >= 24)
          v = (int32_t)(((uint32_t)left_at(0) * 5 + (uint32_t)v * 6 + (uint32_t)above_at(0) * 5 + 8) >> 4);
      } else {
        v = base;
      }
    } else {
      const int i = k - 2 * MAXTX - 1;
      if (i < init_above && needs_top) v = above_at(i);
    }
    stp<BPP>(e, k, v);
  }
  if (l == 0) {
    lens[2 * cand] = (uint8_t)init_left;
    lens[2 * cand + 1] = (uint8_t)init_above;
  }
}

// ---------------------------------------------------------------- predict
// filter_edge (predict.rs:1203-1233): dst[i] for 1 <= i < size from src
__device__ __forceinline__ int32_t filt5(const uint16_t *src, int i, int size, int strength) {
  constexpr uint8_t K[3][5] = {{0, 4, 8, 4, 0}, {0, 5, 6, 5, 0}, {2, 4, 4, 4, 2}};
  int32_t s = 0;
#pragma unroll
  for (int j = 0; j < 5; j++) {
    int k = i + j - 2;
    k = k < 0 ? 0 : (k > size - 1 ? size - 1 : k);
    s += K[strength - 1][j] * (int32_t)src[k];
  }
  return (s + 8) >> 4;
}

// SATD_OUT: the intra mode pre-screen of src/rdo.rs:1434-1506 in one launch --
// `group` consecutive candidates (the modes of one block) share one edge set
// and one source position; the prediction goes to LDS and only get_satd of it
// against the source block leaves the CU.
// WLT / HLT >= 0 (the pre-screen's square sizes): block size known at compile time -- every row loop
// unrolls, and with SATD_OUT the prediction column never leaves the lane's registers (it used to go
// through LDS and back: one ds_write + one ds_read per pixel, plus the address arithmetic of loops
// with a runtime trip count).
#ifndef R1_PRESCREEN_LOOPED
#define R1_PRESCREEN_LOOPED 1   // A/B switch
#endif
// waves per SIMD the register allocator is asked to make room for in the fixed-size pre-screen
// instantiations: left alone it takes 79 / 113 VGPRs (16x16 / 32x32); asked, 44 / 68 without a spill
// (tools/kres.py): launches 0.136 -> 0.133 and 0.126 -> 0.103 ms
constexpr int intra_waves_hint(bool satd, int wlt) {
  return !satd ? 1 : (wlt == 4 ? 7 : (wlt == 5 ? 5 : 1));   // 8x8 at 6 waves (79 VGPRs): +1.7 % on the launch, left alone
}

template <int BPP, bool SATD_OUT, int WLT = -1, int HLT = -1>
__global__ __launch_bounds__(64, intra_waves_hint(SATD_OUT, WLT)) void k_intra_predict(
    int wl, int hl, const R1IntraCand *__restrict__ cands, int n,
    const void *__restrict__ edges, int edge_stride, const uint8_t *__restrict__ lens,
    const int16_t *__restrict__ ac, int bit_depth, void *__restrict__ dst, R1Plane src,
    const int16_t *__restrict__ pos_xy, int group, uint32_t *__restrict__ satd_out) {
  extern __shared__ uint16_t smem[];
  constexpr bool FIXED = WLT >= 0 && HLT >= 0;
  if constexpr (FIXED) { wl = WLT; hl = HLT; }
  const int W = 1 << wl, H = 1 << hl;
  const int NC = 64 >> wl;
  constexpr bool IN_REGS = SATD_OUT && FIXED;
  int32_t pv[IN_REGS ? (1 << (HLT < 0 ? 0 : HLT)) : 1];   // IN_REGS: the lane's prediction column
  const int FL = 2 * (W + H) + 1;
  const int lane = threadIdx.x;
  const int cl = lane >> wl, c = lane & (W - 1);
  // SATD_OUT: a wave takes ONE member of the group (one mode of the pre-screen) for NC
  // consecutive blocks, so every lane runs the same predictor; with the candidates in
  // list order a 16x16 wave held four different modes and paid for all four branches.
  // LOOPED (the fixed-size pre-screen instantiations): a wave takes ALL the members of the group, one
  // after the other, for its NC blocks -- the blocks' edges are loaded once (they were loaded once per
  // mode: 13 times) and the source columns stay in registers across the modes (8 / 16 rows).
  // 8x8 only: at 16x16 / 32x32 the loop costs registers (131 / 157 VGPRs, 3 waves per SIMD) and the
  // launch gets 15-25 % slower than one wave per (blocks, member); at 8x8 it is 24 % faster
  // (gpurun_out/r04_f: 0.243 -> 0.185 ms for 129600 blocks x 13 modes)
  constexpr bool LOOPED = IN_REGS && R1_PRESCREEN_LOOPED && HLT == 3;
  constexpr bool SRC_REGS = LOOPED && HLT >= 0 && HLT <= 4;
  long long cand = (long long)blockIdx.x * NC + cl;
  bool live = cand < n;
  long long ecand = cand;             // edge set / block of this candidate
  if constexpr (LOOPED) {
    ecand = (long long)blockIdx.x * NC + cl;
    live = ecand < n / group;
    cand = ecand * group;
  } else if constexpr (SATD_OUT) {
    const unsigned bg = blockIdx.x / (unsigned)group, mi = blockIdx.x - bg * (unsigned)group;
    ecand = (long long)bg * NC + cl;
    live = ecand < n / group;
    cand = ecand * group + mi;
  }
  uint16_t *raw = smem + cl * EDGE_LEN;
  uint16_t *work = smem + NC * EDGE_LEN + cl * (4 * FL);   // af0 af1 lf0 lf1
  int left_len = 0, above_len = 0;
  int32_t sv[SRC_REGS ? (1 << (HLT < 0 ? 0 : HLT)) : 1];   // SRC_REGS: the lane's source column
  if constexpr (SRC_REGS) {
#pragma unroll
    for (int r = 0; r < (1 << (HLT < 0 ? 0 : HLT)); r++) sv[r] = 0;
    if (live) {
      const uint8_t *ps = px_addr<BPP>(src, pos_xy[2 * ecand] + c, pos_xy[2 * ecand + 1]);
      const size_t ss = (size_t)src.stride * BPP;
#pragma unroll
      for (int r = 0; r < (1 << (HLT < 0 ? 0 : HLT)); r++) sv[r] = ldp<BPP>(ps + (size_t)r * ss, 0);
    }
  }
  if (live) {
    left_len = lens[2 * ecand];
    above_len = lens[2 * ecand + 1];
    const void *e = (const uint8_t *)edges + (size_t)ecand * edge_stride * BPP;
    // only [128 - left_len, 129 + above_len) is defined (and ever read)
    for (int k = 2 * MAXTX - left_len + c; k < 2 * MAXTX + 1 + above_len; k += W)
      raw[k] = (uint16_t)ldp<BPP>(e, k);
  }
  __syncthreads();
  const long long cand0 = cand;
  for (int it = 0; it < (LOOPED ? group : 1); it++) {
  if constexpr (LOOPED) {
    cand = cand0 + it;
    if (it) __syncthreads();     // the previous member's readers of the work arrays are done
  }
  R1IntraCand cd = {};
  if (live) cd = cands[cand];
  const int mode = cd.mode, variant = cd.variant, angle = cd.angle;
  const int32_t smax = (1 << bit_depth) - 1;
  const uint16_t *above = raw + 2 * MAXTX + 1;
  const int32_t top_left = raw[2 * MAXTX];
  // left pixel beside row r (left_slice[height-1-r])
  auto left_row = [&](int r) -> int32_t { return raw[2 * MAXTX - 1 - r]; };
  void *out = SATD_OUT ? (void *)((uint8_t *)smem + ((NC * (EDGE_LEN + 4 * FL) * 2 + 15) & ~15) +
                                  (size_t)cl * W * H * BPP)
                       : (void *)((uint8_t *)dst + (size_t)cand * W * H * BPP);

  // one predicted pixel of this lane's column (row `i_`)
#define R1_PUT(i_, v_)                                                   \
  do {                                                                   \
    if constexpr (IN_REGS) pv[i_] = (v_);                                \
    else stp<BPP>(out, (size_t)(i_) * W + c, (v_));                      \
  } while (0)
  const bool directional = live && mode >= V_PRED && mode <= D67_PRED &&
                           !(mode == V_PRED && angle == 90) && !(mode == H_PRED && angle == 180);
  const bool enable = directional && cd.ief != 0;
  // ---- edge filter / upsample in LDS (wave-uniform barriers, per-lane predicates)
  int up_a = 0, up_l = 0;
  const uint16_t *aedge = above;      // !enable: raw above, index 0 = above[0]
  uint16_t *af0 = work, *af1 = work + FL, *lf0 = work + 2 * FL, *lf1 = work + 3 * FL;
  const int lb_len = left_len < W + H ? left_len : W + H;
  if (__any(enable)) {
    const bool smooth = cd.ief == 2;
    const int wh = W + H;
    if (enable) {
      const int al = above_len < FL - 1 ? above_len : FL - 1;
      const int ll = lb_len < FL - 1 ? lb_len : FL - 1;
      for (int k = c; k < FL; k += W) {
        af0[k] = k == 0 ? 0 : (k - 1 < al ? above[k - 1] : 0);
        // left_filtered[i] = left[left.len() - i]: i-th pixel downwards from the top
        lf0[k] = k == 0 ? 0 : (k <= ll ? raw[2 * MAXTX - k] : 0);
      }
    }
    __syncthreads();
    int npa = 0, npl = 0, sa = 0, sl = 0;
    if (enable && angle != 90 && angle != 180) {
      if (c == 0) { af0[0] = (uint16_t)top_left; lf0[0] = (uint16_t)top_left; }
      npa = (W < cd.avail_w ? W : cd.avail_w) + (angle < 90 ? H : 0) + 1;
      npl = (H < cd.avail_h ? H : cd.avail_h) + (angle > 180 ? W : 0) + 1;
      sa = ief_strength(wh, smooth, angle - 90);
      sl = ief_strength(wh, smooth, angle - 180);
    }
    __syncthreads();
    if (enable)
      for (int k = c; k < FL; k += W) {
        af1[k] = (sa && k >= 1 && k < npa) ? (uint16_t)filt5(af0, k, npa, sa) : af0[k];
        lf1[k] = (sl && k >= 1 && k < npl) ? (uint16_t)filt5(lf0, k, npl, sl) : lf0[k];
      }
    __syncthreads();
    // upsample_edge (predict.rs:1235-1266): af1/lf1 (filtered) -> af0/lf0 (final)
    if (enable) {
      up_a = ief_upsample(wh, smooth, angle - 90);
      up_l = ief_upsample(wh, smooth, angle - 180);
      const int na = W + (angle < 90 ? H : 0), nl = H + (angle > 180 ? W : 0);
      auto ups = [&](const uint16_t *s, uint16_t *d, int size) {
        auto dup = [&](int i) -> int32_t {
          return i == 0 ? s[0] : (i <= size + 1 ? s[i - 1] : s[size]);
        };
        for (int k = c; k < FL; k += W)     // entries outside [1, 2*size] keep s
          if (k == 0 || k > 2 * size) d[k] = s[k];
        for (int i = c; i < size; i += W) {
          int32_t v = -dup(i) + 9 * dup(i + 1) + 9 * dup(i + 2) - dup(i + 3);
          v = (v + 8) / 16;
          v = v < 0 ? 0 : (v > smax ? smax : v);
          d[2 * i + 1] = (uint16_t)v;
          d[2 * i + 2] = (uint16_t)dup(i + 2);
        }
      };
      if (up_a) ups(af1, af0, na);
      else for (int k = c; k < FL; k += W) af0[k] = af1[k];
      if (up_l) ups(lf1, lf0, nl);
      else for (int k = c; k < FL; k += W) lf0[k] = lf1[k];
    }
    __syncthreads();
    if (enable) aedge = af0;
  }
  if (!live) {
    if constexpr (LOOPED) continue;
    else return;
  }

  // left_edge[k] of the reference (after left_filtered.reverse()) = lf0[FL-1-k];
  // raw case: left_and_left_below_slice[k] = raw[128 - lb_len + k]
  const int l = enable ? FL - 1 : lb_len - 1;
  auto ledge = [&](int k) -> int32_t {
    return enable ? (int32_t)lf0[FL - 1 - k] : (int32_t)raw[2 * MAXTX - lb_len + k];
  };

  if (directional) {
    int dx = 0, dy = 0;
    if (angle < 90) dx = kR1DrIntraDerivative[angle];
    else if (angle > 90 && angle < 180) dx = kR1DrIntraDerivative[180 - angle];
    if (angle > 90 && angle < 180) dy = kR1DrIntraDerivative[angle - 90];
    else if (angle > 180) dy = kR1DrIntraDerivative[270 - angle];
    const int oa = (enable ? 1 : 0) << up_a, ol = (enable ? 1 : 0) << up_l;
    const int j = c;
#pragma unroll
    for (int i = 0; i < H; i++) {
      int32_t v;
      if (angle < 90) {
        const int idx = (i + 1) * dx;
        const int base = (idx >> (6 - up_a)) + (j << up_a);
        const int shift = ((idx << up_a) >> 1) & 31;
        const int mb = (H + W - 1) << up_a;
        if (base < mb)
          v = ((int32_t)aedge[base + oa] * (32 - shift) + (int32_t)aedge[base + 1 + oa] * shift + 16) >> 5;
        else
          v = aedge[mb + oa];
      } else if (angle < 180) {
        int idx = (j << 6) - (i + 1) * dx;
        int base = idx >> (6 - up_a);
        if (base >= -(1 << up_a)) {
          const int shift = ((idx << up_a) >> 1) & 31;
          const int32_t a = (!enable && base < 0) ? top_left : (int32_t)aedge[base + oa];
          const int32_t b = aedge[base + 1 + oa];
          v = (a * (32 - shift) + b * shift + 16) >> 5;
        } else {
          idx = (i << 6) - (j + 1) * dy;
          base = idx >> (6 - up_l);
          const int shift = ((idx << up_l) >> 1) & 31;
          int32_t a, b;
          if (!enable && base < 0) a = top_left;
          else if (base + ol == -2) a = ledge(0);
          else a = ledge(l - (base + ol));
          if (base + ol == -2) b = ledge(1);
          else b = ledge(l - (base + ol + 1));
          v = (a * (32 - shift) + b * shift + 16) >> 5;
        }
      } else {
        const int idx = (j + 1) * dy;
        const int base = (idx >> (6 - up_l)) + (i << up_l);
        const int shift = ((idx << up_l) >> 1) & 31;
        int ia = l - (base + ol), ib = l - (base + ol + 1);
        ia = ia < 0 ? 0 : ia;
        ib = ib < 0 ? 0 : ib;
        v = (ledge(ia) * (32 - shift) + ledge(ib) * shift + 16) >> 5;
      }
      R1_PUT(i, v < 0 ? 0 : (v > smax ? smax : v));
    }
  }
  // ---- non-directional ----
  const int ls_len = left_len < H ? left_len : H;
  if (directional) {
  } else if (mode == V_PRED) {
    const int32_t a = above[c];
#pragma unroll
    for (int r = 0; r < H; r++) R1_PUT(r, a);
  } else if (mode == H_PRED) {
#pragma unroll
    for (int r = 0; r < H; r++) R1_PUT(r, left_row(r));
  } else if (mode == PAETH_PRED) {
    const int32_t rt = above[c];
#pragma unroll
    for (int r = 0; r < H; r++) {
      const int32_t rl = left_row(r);
      const int32_t base = rt + rl - top_left;
      const int32_t pl = iabs(base - rl), pt = iabs(base - rt), ptl = iabs(base - top_left);
      R1_PUT(r, (pl <= pt && pl <= ptl) ? rl : (pt <= ptl ? rt : top_left));
    }
  } else if (mode == SMOOTH_PRED || mode == SMOOTH_V_PRED || mode == SMOOTH_H_PRED) {
    const uint32_t below_pred = raw[2 * MAXTX - ls_len], right_pred = above[W - 1];
    const uint32_t a = above[c], wc = kR1SmWeights[W + c];
#pragma unroll
    for (int r = 0; r < H; r++) {
      const uint32_t lft = (uint32_t)left_row(r), wr = kR1SmWeights[H + r];
      uint32_t p;
      if (mode == SMOOTH_PRED)
        p = (wr * a + (256 - wr) * below_pred + wc * lft + (256 - wc) * right_pred + 256) >> 9;
      else if (mode == SMOOTH_H_PRED)
        p = (wc * lft + (256 - wc) * right_pred + 128) >> 8;
      else
        p = (wr * a + (256 - wr) * below_pred + 128) >> 8;
      R1_PUT(r, (int32_t)p);
    }
  } else {   // DC_PRED / UV_CFL_PRED
    uint32_t avg;
    if (variant == 0) {
      avg = 128u << (bit_depth - 8);
    } else if (variant == 1) {
      uint32_t s = 0;
      for (int i = 0; i < ls_len; i++) s += raw[2 * MAXTX - ls_len + i];
      avg = (s + (uint32_t)(H >> 1)) / (uint32_t)H;
    } else if (variant == 2) {
      uint32_t s = 0;
      for (int i = 0; i < W; i++) s += above[i];
      avg = (s + (uint32_t)(W >> 1)) / (uint32_t)W;
    } else {
      uint32_t s = 0;
      for (int i = 0; i < H; i++) s += raw[2 * MAXTX - ls_len + i];
      for (int i = 0; i < W; i++) s += above[i];
      avg = (s + (uint32_t)((W + H) >> 1)) / (uint32_t)(W + H);
    }
    if (mode == UV_CFL_PRED && angle != 0) {
      const int16_t *acb = ac + cand * (W * H);
#pragma unroll
      for (int r = 0; r < H; r++) {
        const int32_t q6 = (int32_t)(int16_t)angle * (int32_t)acb[r * W + c];
        const int32_t q0 = (iabs(q6) + 32) >> 6;
        const int32_t v = (int32_t)avg + (q6 < 0 ? -q0 : q0);
        R1_PUT(r, v < 0 ? 0 : (v > smax ? smax : v));
      }
    } else {
#pragma unroll
      for (int r = 0; r < H; r++) R1_PUT(r, (int32_t)avg);
    }
  }
#undef R1_PUT
  if constexpr (SATD_OUT) {
    __builtin_amdgcn_wave_barrier();
    const bool small = (W < H ? W : H) == 4;
    const int ts = small ? 4 : 8, ntx = W / ts, nt = ntx * (H / ts);
    uint32_t sum = 0;
    if (!small) {
      // get_satd with 8x8 tiles, lane = column (every lane of the candidate works): the
      // lane re-reads the column it just wrote, takes the vertical Hadamard of each 8-row
      // group on registers and the horizontal one across its 8-lane tile by DPP (the
      // fused candidate kernel's scheme, rdo_cand.hip) -- with one lane per tile a 16x16
      // block kept 4 of its 16 lanes busy for 64 pixels each.
      const int32_t s1 = -(int32_t)((lane ^ (lane >> 2)) & 1);
      const int32_t s2 = -(int32_t)(((lane >> 1) ^ (lane >> 2)) & 1);
      const int bx = pos_xy[2 * ecand], by = pos_xy[2 * ecand + 1];
      const uint8_t *ps = px_addr<BPP>(src, bx + c, by);
      const size_t ss = (size_t)src.stride * BPP;
#pragma unroll
      for (int g = 0; g < H; g += 8) {
        int32_t a[8], b[8], d[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
          int32_t pp;
          if constexpr (IN_REGS) pp = pv[g + k];
          else pp = ldp<BPP>(out, (size_t)(g + k) * W + c);
          if constexpr (SRC_REGS) a[k] = sv[g + k] - pp;
          else a[k] = ldp<BPP>(ps + (size_t)(g + k) * ss, 0) - pp;
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
          b[2 * k] = a[2 * k] + a[2 * k + 1];
          b[2 * k + 1] = a[2 * k] - a[2 * k + 1];
        }
        d[0] = b[0] + b[2]; d[2] = b[0] - b[2];
        d[1] = b[1] + b[3]; d[3] = b[1] - b[3];
        d[4] = b[4] + b[6]; d[6] = b[4] - b[6];
        d[5] = b[5] + b[7]; d[7] = b[5] - b[7];
#pragma unroll
        for (int k = 0; k < 8; k++) {
          // masks {1, 2, 7} span (Z/2)^3: a Walsh-Hadamard transform in a relabelled lane
          // order, same multiset of coefficients; the last stage is folded into the abs,
          // |p + q| + |p - q| = 2 max(|p|, |q|)
          int32_t x = k < 4 ? d[k] + d[k + 4] : d[k - 4] - d[k];
          x = __mul24(x, s1 | 1) + __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xf, 0xf, true);
          x = __mul24(x, s2 | 1) + __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xf, 0xf, true);
          const int32_t ax = iabs(x);
          const int32_t ap = __builtin_amdgcn_update_dpp(0, ax, 0x141, 0xf, 0xf, true);
          sum += (uint32_t)(ax > ap ? ax : ap);
        }
      }
    } else if (c < nt) {
      const int tx = c % ntx, ty = c / ntx;
      const int bx = pos_xy[2 * ecand], by = pos_xy[2 * ecand + 1];
      const uint8_t *po = px_addr<BPP>(src, bx + tx * ts, by + ty * ts);
      const uint8_t *pp = (const uint8_t *)out + ((size_t)ty * ts * W + (size_t)tx * ts) * BPP;
      sum = small ? r1dist::tile_dist<BPP, 4, true>(po, (size_t)src.stride * BPP, pp, (size_t)W * BPP)
                  : r1dist::tile_dist<BPP, 8, true>(po, (size_t)src.stride * BPP, pp, (size_t)W * BPP);
    }
    for (int m = 1; m < W; m <<= 1) sum += __shfl_xor(sum, m, 64);
    const int ln = small ? 2 : 3;
    if (c == 0) satd_out[cand] = (sum + ((1u << ln) >> 1)) >> ln;
  }
  }   // members of the group (LOOPED), else one pass
}

// r1_prescreen_select_batch: one thread per (group, element).  An element's place in the
// stable sort is the number of elements that precede it: smaller key, or equal key and
// smaller index -- no sorting network, the groups have at most 64 members.
__global__ __launch_bounds__(256) void k_prescreen_select(const uint32_t *__restrict__ keys,
                                                          int n_groups, int group, int head, int k,
                                                          uint8_t *__restrict__ out) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long g = t / group;
  const int j = (int)(t - g * group);
  if (g >= n_groups) return;
  const uint32_t *kg = keys + g * group;
  int place = j;
  if (j >= head) {
    const uint32_t kj = kg[j];
    place = head;
    for (int i = head; i < group; i++) {
      const uint32_t ki = kg[i];
      place += (ki < kj) || (ki == kj && i < j);
    }
  }
  if (place < k) out[g * k + place] = (uint8_t)j;
}

// pred_cfl_ac (predict.rs:1020-1063): one wave per candidate
template <int BPP>
__global__ __launch_bounds__(64) void k_cfl_ac(R1Plane luma, int bw, int bh, int xdec, int ydec,
                                               const R1CflAcCand *__restrict__ cands, int n,
                                               int16_t *__restrict__ ac) {
  const int cand = blockIdx.x;
  if (cand >= n) return;
  const R1CflAcCand cd = cands[cand];
  const int mlw = (bw - cd.w_pad * 4) << xdec, mlh = (bh - cd.h_pad * 4) << ydec;
  const int mx = (mlw > 8 ? mlw : 8) - (1 << xdec), my = (mlh > 8 ? mlh : 8) - (1 << ydec);
  const uint8_t *p0 = px_addr<BPP>(luma, cd.x, cd.y);
  const size_t st = (size_t)luma.stride;
  int16_t *out = ac + (size_t)cand * bw * bh;
  int32_t sum = 0;
  for (int i = threadIdx.x; i < bw * bh; i += 64) {
    const int sy = i / bw, sx = i - sy * bw;
    const int ly = sy << ydec, lx = sx << xdec;
    const int y = ly < my ? ly : my, x = lx < mx ? lx : mx;
    int32_t s = ldp<BPP>(p0, y * st + x);
    if (xdec) s += ldp<BPP>(p0, y * st + x + 1);
    if (ydec) s += ldp<BPP>(p0, (y + 1) * st + x) + ldp<BPP>(p0, (y + 1) * st + x + 1);
    s = (int16_t)(s << (3 - xdec - ydec));
    out[i] = (int16_t)s;
    sum += s;
  }
  sum = (int32_t)group_sum<64>((uint32_t)sum);
  const int shift = 31 - __clz(bw * bh);
  const int16_t avg = (int16_t)((sum + (1 << (shift - 1))) >> shift);
  __syncthreads();
  for (int i = threadIdx.x; i < bw * bh; i += 64) out[i] = (int16_t)(out[i] - avg);
}

// rdo_cfl_alpha (src/rdo.rs:1593-1688) for one chroma plane: one wave per block.
// Lane a evaluates alpha = a - 16 (-16 .. 16): pred_cfl_inner (predict.rs:1064-1091)
// of the whole block against the source, plain SSE over the visible part
// (sse_wxh with the default scale is the plain sum); lane 0 then replays the
// reference's sequential selection (alpha 0 first, +-1 .. +-16 with the early
// exit `count < alpha`) on the 33 costs.
template <int BPP>
__global__ __launch_bounds__(64) void k_cfl_alpha(R1Plane src, int wl, int hl,
                                                  const R1CflAlphaCand *__restrict__ cands, int n,
                                                  const void *__restrict__ edges, int edge_stride,
                                                  const uint8_t *__restrict__ lens,
                                                  const int16_t *__restrict__ ac, int bit_depth,
                                                  int16_t *__restrict__ alpha_out,
                                                  unsigned long long *__restrict__ cost_out) {
  __shared__ int16_t s_ac[32 * 32];
  __shared__ uint16_t s_src[32 * 32];
  __shared__ unsigned long long s_cost[33];
  const int W = 1 << wl, H = 1 << hl, lane = threadIdx.x;
  const int cand = blockIdx.x;
  const R1CflAlphaCand cd = cands[cand];
  const int vw = cd.vis_w, vh = cd.vis_h;
  // DC_PRED average of the block's edges (predict.rs:885-933), by variant
  const void *e = (const uint8_t *)edges + (size_t)cand * edge_stride * BPP;
  const int left_len = lens[2 * cand];
  const int ls_len = left_len < H ? left_len : H;
  uint32_t avg;
  if (cd.variant == 0) {
    avg = 128u << (bit_depth - 8);
  } else {
    uint32_t sum = 0;
    if (cd.variant != 2)   // LEFT or BOTH
      for (int i = 0; i < (cd.variant == 1 ? ls_len : H); i++) sum += (uint32_t)ldp<BPP>(e, 2 * MAXTX - ls_len + i);
    if (cd.variant != 1)   // TOP or BOTH
      for (int i = 0; i < W; i++) sum += (uint32_t)ldp<BPP>(e, 2 * MAXTX + 1 + i);
    const uint32_t len = cd.variant == 1 ? (uint32_t)H : (cd.variant == 2 ? (uint32_t)W : (uint32_t)(W + H));
    avg = (sum + (len >> 1)) / len;
  }
  for (int i = lane; i < W * H; i += 64) {
    s_ac[i] = ac[(size_t)cand * W * H + i];
    const int x = i & (W - 1), y = i >> wl;
    s_src[i] = (x < vw && y < vh) ? (uint16_t)ld_px<BPP>(px_addr<BPP>(src, cd.x + x, cd.y + y)) : 0;
  }
  __syncthreads();
  if (lane < 33) {
    const int alpha = lane - 16;
    const int32_t smax = (1 << bit_depth) - 1;
    unsigned long long sse = 0;
    for (int y = 0; y < vh; y++)
      for (int x = 0; x < vw; x++) {
        const int i = (y << wl) + x;
        const int32_t q6 = alpha * (int32_t)s_ac[i];
        const int32_t q0 = ((q6 < 0 ? -q6 : q6) + 32) >> 6;
        int32_t v = (int32_t)avg + (q6 < 0 ? -q0 : q0);
        v = v < 0 ? 0 : (v > smax ? smax : v);
        const int32_t d = (int32_t)s_src[i] - v;
        sse += (unsigned long long)(uint32_t)(d * d);
      }
    s_cost[lane] = sse;
  }
  __syncthreads();
  if (lane == 0) {
    unsigned long long best = s_cost[16];
    int best_a = 0, count = 2;
    for (int a = 1; a <= 16; a++) {
      const unsigned long long cp = s_cost[16 + a], cm = s_cost[16 - a];
      if (cp < best) { best = cp; best_a = a; count += 2; }
      if (cm < best) { best = cm; best_a = -a; count += 2; }
      if (count < a) break;
    }
    alpha_out[cand] = (int16_t)best_a;
    if (cost_out) cost_out[cand] = best;
  }
}

}  // namespace

extern "C" int r1_intra_edges_batch(r1_ctx *ctx, const R1Plane *rec, int tile_x, int tile_y,
                                    int tile_w, int tile_h, int tx_size,
                                    const R1IntraEdgeCand *cands, int n, void *edges,
                                    int edge_stride, uint8_t *lens, void *stream) {
  R1_REQUIRE(ctx && rec);
  R1_REQUIRE(rec->bytes_per_px == 1 || rec->bytes_per_px == 2);
  R1_REQUIRE(tx_size >= 0 && tx_size < 19);
  R1_REQUIRE(edge_stride >= EDGE_LEN);
  R1_REQUIRE(tile_x >= 0 && tile_y >= 0 && tile_w > 0 && tile_h > 0);
  if (n <= 0) return R1_OK;
  R1_REQUIRE(cands && edges && lens);
  // rect_w / rect_h: dst.rect() clipped to the plane (partition.rs:701-704)
  const int rect_w = tile_w < rec->width - tile_x ? tile_w : rec->width - tile_x;
  const int rect_h = tile_h < rec->height - tile_y ? tile_h : rec->height - tile_y;
  static const uint8_t wl[19] = {2, 3, 4, 5, 6, 2, 3, 3, 4, 4, 5, 5, 6, 2, 4, 3, 5, 4, 6};
  static const uint8_t hl[19] = {2, 3, 4, 5, 6, 3, 2, 4, 3, 5, 4, 6, 5, 4, 2, 5, 3, 6, 4};
  hipStream_t st = (hipStream_t)stream;
  const int txw = 1 << wl[tx_size], txh = 1 << hl[tx_size];
  int lpc_log2 = 0;
  while ((1 << lpc_log2) < 2 * (txw + txh) + 1 && lpc_log2 < 6) lpc_log2++;
  // EIGHT lanes per block, whatever its size: what a lane does before it touches an entry (the mode / flag
  // logic, the clipping, the block's address) is the same for every lane of a block and dominates; with 8
  // lanes a wave prepares 8 blocks and walks the 17 .. 129 entries in strides of 8.  Measured against one
  // lane per entry (gpurun_out/r04_g, 4K luma): 4x4 0.097 -> 0.037 ms, 8x8 0.048 -> 0.015, 16x16 0.0165 ->
  // 0.0124, 32x32 0.0121 -> 0.0119.  (-DR1_EDGES_LPC_SHIFT=0 restores one lane per entry, for A/B builds.)
  // (the A/B switch is a build-time macro like the others -- -DR1_EDGES_LPC_SHIFT=0 -- not an environment variable:
  // the shipped library's behaviour does not depend on the process environment)
#ifndef R1_EDGES_LPC_SHIFT
#define R1_EDGES_LPC_SHIFT 3
#endif
  static_assert(R1_EDGES_LPC_SHIFT >= 0 && R1_EDGES_LPC_SHIFT <= 3, "lanes per candidate stay within 8 .. 64");
  lpc_log2 = lpc_log2 - R1_EDGES_LPC_SHIFT < 3 ? 3 : lpc_log2 - R1_EDGES_LPC_SHIFT;
  if (lpc_log2 > 6) lpc_log2 = 6;
  const int cpw = 64 >> lpc_log2;
  const unsigned grid = (unsigned)((n + cpw - 1) / cpw);
  if (rec->bytes_per_px == 1)
    hipLaunchKernelGGL((k_intra_edges<1>), dim3(grid), dim3(64), 0, st, *rec, tile_x, tile_y, rect_w,
                       rect_h, txw, txh, lpc_log2, cands, n, edges, edge_stride, lens);
  else
    hipLaunchKernelGGL((k_intra_edges<2>), dim3(grid), dim3(64), 0, st, *rec, tile_x, tile_y, rect_w,
                       rect_h, txw, txh, lpc_log2, cands, n, edges, edge_stride, lens);
  R1_HIP_CHECK(hipGetLastError());
  return R1_OK;
}

extern "C" int r1_predict_intra_batch(r1_ctx *ctx, int tx_size, const R1IntraCand *cands, int n,
                                      const void *edges, int edge_stride, const uint8_t *lens,
                                      const int16_t *ac, int bit_depth, int bytes_per_px,
                                      void *dst, void *stream) {
  R1_REQUIRE(ctx);
  R1_REQUIRE(tx_size >= 0 && tx_size < 19);
  R1_REQUIRE(bit_depth == 8 || bit_depth == 10 || bit_depth == 12);
  R1_REQUIRE(bytes_per_px == 1 || bytes_per_px == 2);
  R1_REQUIRE((bytes_per_px == 1) == (bit_depth == 8));
  R1_REQUIRE(edge_stride >= EDGE_LEN);
  if (n <= 0) return R1_OK;
  R1_REQUIRE(cands && edges && lens && dst);
  static const uint8_t wl[19] = {2, 3, 4, 5, 6, 2, 3, 3, 4, 4, 5, 5, 6, 2, 4, 3, 5, 4, 6};
  static const uint8_t hl[19] = {2, 3, 4, 5, 6, 3, 2, 4, 3, 5, 4, 6, 5, 4, 2, 5, 3, 6, 4};
  const int W = 1 << wl[tx_size], H = 1 << hl[tx_size];
  const int NC = 64 / W, FL = 2 * (W + H) + 1;
  const size_t lds = (size_t)NC * (EDGE_LEN + 4 * FL) * sizeof(uint16_t);
  const unsigned grid = (unsigned)((n + NC - 1) / NC);
  hipStream_t st = (hipStream_t)stream;
#define R1_PRED_LAUNCH(B, ...)                                                                            \
  hipLaunchKernelGGL((k_intra_predict<B, false, ##__VA_ARGS__>), dim3(grid), dim3(64), lds, st,           \
                     (int)wl[tx_size], (int)hl[tx_size], cands, n, edges, edge_stride, lens, ac,          \
                     bit_depth, dst, R1Plane{}, (const int16_t *)nullptr, 1, (uint32_t *)nullptr)
  // square blocks 4x4 .. 32x32 with the size as a constant (unrolled row loops)
  const int sq = wl[tx_size] == hl[tx_size] ? wl[tx_size] : 0;
  if (bytes_per_px == 1) {
    if (sq == 2) R1_PRED_LAUNCH(1, 2, 2);
    else if (sq == 3) R1_PRED_LAUNCH(1, 3, 3);
    else if (sq == 4) R1_PRED_LAUNCH(1, 4, 4);
    else if (sq == 5) R1_PRED_LAUNCH(1, 5, 5);
    else R1_PRED_LAUNCH(1);
  } else {
    if (sq == 2) R1_PRED_LAUNCH(2, 2, 2);
    else if (sq == 3) R1_PRED_LAUNCH(2, 3, 3);
    else if (sq == 4) R1_PRED_LAUNCH(2, 4, 4);
    else if (sq == 5) R1_PRED_LAUNCH(2, 5, 5);
    else R1_PRED_LAUNCH(2);
  }
#undef R1_PRED_LAUNCH
  R1_HIP_CHECK(hipGetLastError());
  return R1_OK;
}

extern "C" int r1_intra_satd_batch(r1_ctx *ctx, const R1Plane *src, int tx_size,
                                   const R1IntraCand *cands, int n, int group,
                                   const int16_t *pos_xy, const void *edges, int edge_stride,
                                   const uint8_t *lens, const int16_t *ac, uint32_t *satd_out,
                                   void *stream) {
  R1_REQUIRE(ctx && src);
  R1_REQUIRE(tx_size >= 0 && tx_size < 19);
  R1_REQUIRE(src->bit_depth == 8 || src->bit_depth == 10 || src->bit_depth == 12);
  R1_REQUIRE(src->bytes_per_px == 1 || src->bytes_per_px == 2);
  R1_REQUIRE((src->bytes_per_px == 1) == (src->bit_depth == 8));
  R1_REQUIRE(edge_stride >= EDGE_LEN && group >= 1);
  if (n <= 0) return R1_OK;
  R1_REQUIRE(n % group == 0);
  R1_REQUIRE(cands && edges && lens && pos_xy && satd_out);
  static const uint8_t wl[19] = {2, 3, 4, 5, 6, 2, 3, 3, 4, 4, 5, 5, 6, 2, 4, 3, 5, 4, 6};
  static const uint8_t hl[19] = {2, 3, 4, 5, 6, 3, 2, 4, 3, 5, 4, 6, 5, 4, 2, 5, 3, 6, 4};
  const int W = 1 << wl[tx_size], H = 1 << hl[tx_size];
  const int NC = 64 / W, FL = 2 * (W + H) + 1;
  const size_t lds = (((size_t)NC * (EDGE_LEN + 4 * FL) * sizeof(uint16_t) + 15) & ~(size_t)15) +
                     (size_t)NC * W * H * src->bytes_per_px;
  unsigned grid = (unsigned)((n / group + NC - 1) / NC) * (unsigned)group;   // (block group, member)
  // the pre-screen's sizes (luma transform blocks 8x8 .. 32x32) with the block size as a constant; at 8x8 a
  // wave walks the members of its blocks' groups itself
  const int sq = wl[tx_size] == hl[tx_size] ? wl[tx_size] : 0;
  if (R1_PRESCREEN_LOOPED && sq == 3) grid /= (unsigned)group;
  hipStream_t st = (hipStream_t)stream;
#define R1_SATD_LAUNCH(B, ...)                                                                            \
  hipLaunchKernelGGL((k_intra_predict<B, true, ##__VA_ARGS__>), dim3(grid), dim3(64), lds, st,            \
                     (int)wl[tx_size], (int)hl[tx_size], cands, n, edges, edge_stride, lens, ac,          \
                     src->bit_depth, (void *)nullptr, *src, pos_xy, group, satd_out)
  if (src->bytes_per_px == 1) {
    if (sq == 3) R1_SATD_LAUNCH(1, 3, 3);
    else if (sq == 4) R1_SATD_LAUNCH(1, 4, 4);
    else if (sq == 5) R1_SATD_LAUNCH(1, 5, 5);
    else R1_SATD_LAUNCH(1);
  } else {
    if (sq == 3) R1_SATD_LAUNCH(2, 3, 3);
    else if (sq == 4) R1_SATD_LAUNCH(2, 4, 4);
    else if (sq == 5) R1_SATD_LAUNCH(2, 5, 5);
    else R1_SATD_LAUNCH(2);
  }
#undef R1_SATD_LAUNCH
  R1_HIP_CHECK(hipGetLastError());
  return R1_OK;
}

extern "C" int r1_prescreen_select_batch(r1_ctx *ctx, const uint32_t *keys, int n_groups, int group,
                                         int keep_head, int k, uint8_t *idx_out, void *stream) {
  R1_REQUIRE(ctx);
  R1_REQUIRE(group >= 1 && group <= 64 && k >= 1 && k <= group && keep_head >= 0 && keep_head <= k);
  if (n_groups <= 0) return R1_OK;
  R1_REQUIRE(keys && idx_out);
  const long long total = (long long)n_groups * group;
  hipLaunchKernelGGL(k_prescreen_select, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, keys, n_groups, group, keep_head, k, idx_out);
  R1_HIP_CHECK(hipGetLastError());
  return R1_OK;
}

extern "C" int r1_cfl_ac_batch(r1_ctx *ctx, const R1Plane *luma, int bw, int bh, int xdec,
                               int ydec, const R1CflAcCand *cands, int n, int16_t *ac,
                               void *stream) {
  R1_REQUIRE(ctx && luma);
  R1_REQUIRE(luma->bytes_per_px == 1 || luma->bytes_per_px == 2);
  R1_REQUIRE(r1_is_pow2(bw) && r1_is_pow2(bh) && bw >= 4 && bh >= 4 && bw <= 64 && bh <= 64);
  R1_REQUIRE(xdec >= 0 && xdec <= 1 && ydec >= 0 && ydec <= 1 && (ydec == 0 || xdec == 1));
  if (n <= 0) return R1_OK;
  R1_REQUIRE(cands && ac);
  hipStream_t st = (hipStream_t)stream;
  if (luma->bytes_per_px == 1)
    hipLaunchKernelGGL((k_cfl_ac<1>), dim3(n), dim3(64), 0, st, *luma, bw, bh, xdec, ydec, cands, n, ac);
  else
    hipLaunchKernelGGL((k_cfl_ac<2>), dim3(n), dim3(64), 0, st, *luma, bw, bh, xdec, ydec, cands, n, ac);
  R1_HIP_CHECK(hipGetLastError());
  return R1_OK;
}

extern "C" int r1_cfl_alpha_search_batch(r1_ctx *ctx, const R1Plane *src, int tx_size,
                                         const R1CflAlphaCand *cands, int n, const void *edges,
                                         int edge_stride, const uint8_t *lens, const int16_t *ac,
                                         int16_t *alpha_out, uint64_t *cost_out, void *stream) {
  R1_REQUIRE(ctx && src);
  R1_REQUIRE(tx_size >= 0 && tx_size < 19);
  R1_REQUIRE(src->bytes_per_px == 1 || src->bytes_per_px == 2);
  R1_REQUIRE((src->bytes_per_px == 1) == (src->bit_depth == 8));
  R1_REQUIRE(edge_stride >= EDGE_LEN);
  static const uint8_t wl[19] = {2, 3, 4, 5, 6, 2, 3, 3, 4, 4, 5, 5, 6, 2, 4, 3, 5, 4, 6};
  static const uint8_t hl[19] = {2, 3, 4, 5, 6, 3, 2, 4, 3, 5, 4, 6, 5, 4, 2, 5, 3, 6, 4};
  R1_REQUIRE(wl[tx_size] <= 5 && hl[tx_size] <= 5);   // CFL: chroma transforms up to 32x32
  if (n <= 0) return R1_OK;
  R1_REQUIRE(cands && edges && lens && ac && alpha_out);
  hipStream_t st = (hipStream_t)stream;
  if (src->bytes_per_px == 1)
    hipLaunchKernelGGL((k_cfl_alpha<1>), dim3(n), dim3(64), 0, st, *src, (int)wl[tx_size],
                       (int)hl[tx_size], cands, n, edges, edge_stride, lens, ac, src->bit_depth,
                       alpha_out, (unsigned long long *)cost_out);
  else
    hipLaunchKernelGGL((k_cfl_alpha<2>), dim3(n), dim3(64), 0, st, *src, (int)wl[tx_size],
                       (int)hl[tx_size], cands, n, edges, edge_stride, lens, ac, src->bit_depth,
                       alpha_out, (unsigned long long *)cost_out);
  R1_HIP_CHECK(hipGetLastError());
  return R1_OK;
}
