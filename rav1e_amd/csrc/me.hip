// me.hip -- hierarchical motion estimation of whole tiles on the device
// (SURVEY.md 8f "N2"; reference src/me.rs: estimate_tile_motion 153-218,
// estimate_sb_motion 220-282, refine_subsampled_sb_motion 284-322,
// get_subset_predictors 386-534, estimate_motion 536-632,
// refine_subsampled_motion_estimate 634-691, full_pixel_me 693-855,
// get_best_predictor 884-911, fullpel_diamond_search 955-1000, hexagon_search
// 1055-1141, uneven_multi_hex_search 1170-1309, full_search 1464-1510,
// compute_mv_rd 1445-1462, get_mv_rate 1512-1523).
//
// What is parallel and what is not.  The reference walks the superblocks of a
// tile in raster order; a block's predictors are the (already updated) MEStats
// left of and above it plus the (not yet updated) ones right of and below it
// (me.rs:417-457), so the exact dependence graph is a wavefront: SB (x, y)
// needs (x-1, y) and (x, y-1) finished and (x+1, y), (x, y+1) untouched.
//   * one LAUNCH per anti-diagonal of superblocks; the three passes (quarter,
//     half, full resolution) run in the SAME launches, pass q two diagonals
//     behind pass q - 1 (k_me_diag: a pass needs its predecessor finished one
//     diagonal ahead, nothing more) -- the same-diagonal SBs of ALL jobs (tiles
//     x reference frames) run concurrently, grid = (diag length, jobs, passes);
//   * one WORKGROUP (4 waves) per superblock: first the refinement of the
//     previous pass' blocks (4x4 full search, me.rs:663-676), then the pass' own
//     blocks along the anti-diagonals INSIDE the superblock, one wave per block;
//   * inside a wave the candidates of one search step are evaluated together:
//     16-row blocks put 4 candidates x 16 rows on the 64 lanes, 32-row blocks
//     2 x 32; a lane holds its source row in registers, pulls the candidate's
//     reference row with unaligned dword loads (L2-resident: the search window
//     of a block is a few KB) and SADs it with v_sad_u8 / v_sad_u16; the row
//     sums meet in a segmented wave reduction; "first strictly smaller cost
//     wins" of the reference's sequential loops is an argmin with the lower
//     candidate index breaking ties.
// Every search loop of the reference is data dependent (it recentres on the
// best candidate), so a block's search is a chain of such steps; the chip is
// filled by jobs x superblocks-on-the-diagonal x blocks, not by one block.
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <initializer_list>
#include <mutex>
#include <type_traits>
#include <vector>

#include "common.hpp"
#include "dist_common.hpp"
#include "mc_common.hpp"
#include "tx_common.hpp"

namespace {
using r1tx::T;
#include "mc_taps_packed.inc"
#include "cand_helpers.inc"

// One sub-pel candidate of a W x H block (W, H in {8, 16}) inside a 16-lane group, on the fused
// candidate kernel's machinery (rdo_cand.hip): window staged by the group with one round trip,
// lane = column, v_dot4 / v_dot2 column filter, residual against the source block (in LDS for the
// whole search), SATD by the DPP Hadamard (or SAD).  Returns this lane's share; the caller sums the
// group.  The generic path it replaces for these sizes (scalar taps from byte reads, one lane per
// 8x8 Hadamard tile = 4 of 16 lanes busy) cost about three times the instructions.
template <int BPP, int W, int H, int BD, int NL = 16>
__device__ __forceinline__ uint32_t subpel_group_dist(uint8_t *win, const R1Plane &ref, int x, int y, int cf,
                                                      int rf, int fm, int gl, int lane, const uint8_t *src /* LDS: the W x H source block, dense */,
                                                      bool satd, int bit_depth) {
  constexpr int WS = (((W + 7) * BPP + 3) >> 2) << 2;
  r1mc::stage_window_fast<BPP, BPP == 1 ? 0x80808080u : 0u, W, H, NL>(win, WS, ref, x, y, gl);
  __builtin_amdgcn_wave_barrier();
  T v[H];
#pragma unroll
  for (int r = 0; r < H; r++) v[r] = 0;
  if (gl < W) {
    int32_t pred[H];
    if constexpr (BPP == 1) {
      const Taps8 tp = load_taps8<W, H>(cf, rf, fm, fm);
      mc8_column_t<W, H, WS, false>(win, gl, tp, pred, taps_six(tp));
    } else {
      const Taps16 tp = load_taps16<W, H>(cf, rf, fm, fm);
      mc16_column_t<W, H, WS, false>(win, gl, tp, bit_depth, pred, taps_six(tp));
    }
#pragma unroll
    for (int r = 0; r < H; r++) v[r] = ld_px<BPP>(src + (r * W + gl) * BPP) - pred[r];
  }
  __builtin_amdgcn_wave_barrier();   // the window is rewritten by the next candidate of this group
  if (satd) return satd_column<8, H, BD>(v, lane);
  uint32_t s = 0;
#pragma unroll
  for (int r = 0; r < H; r++) s += (uint32_t)iabs32(v[r]);
  return s;
}

constexpr int MI = 4, SB = 64;
constexpr unsigned long long COST_MAX = ~0ull;

struct Msr {   // MotionSearchResult: wave-uniform, replicated in every lane
  int row, col;
  unsigned long long cost;
  uint32_t sad;
};
__device__ __forceinline__ Msr msr_empty() { return Msr{0, 0, COST_MAX, 0xFFFFFFFFu}; }

__device__ __forceinline__ int ilog_abs(int d) {   // ILog::ilog(d.abs())
  const uint32_t a = (uint32_t)(d < 0 ? -d : d);
  return a ? 32 - __clz(a) : 0;
}
__device__ __forceinline__ int div8(int v) { return (v + ((v >> 31) & 7)) >> 3; }   // trunc

// compute_mv_rd's cost (me.rs:1456-1461) from a distortion
struct MvCost {
  uint32_t lambda;
  int allow_hp;
  int pmv_row[2], pmv_col[2];
  __device__ __forceinline__ uint32_t rate1(int row, int col, int k) const {
    const int dr = (int16_t)(row - pmv_row[k]), dc = (int16_t)(col - pmv_col[k]);
    return 2u * (uint32_t)(ilog_abs(allow_hp ? dr : dr >> 1) + ilog_abs(allow_hp ? dc : dc >> 1));
  }
  __device__ __forceinline__ unsigned long long cost(int row, int col, uint32_t dist) const {
    const uint32_t r1 = rate1(row, col, 0), r2 = rate1(row, col, 1) + 1;
    return 256ull * dist + (unsigned long long)(r1 < r2 ? r1 : r2) * lambda;
  }
};

// One block of one wave.  RH = rows per candidate slot (16 or 32): the block
// is at most RH x RH; 64 / RH candidates are evaluated per step.
// KM: candidate batches in flight per search step; 0 = by register budget (three while a batch is
// <= 4 registers, else two: k_me_diag lives on 96 registers); k_me_persist, at two waves per SIMD,
// affords three always (five: equal, eight: slower -- profiles/r02_me_persistent_experiment.md)
template <int BPP, int RH, int KM = 0>
struct Block {
  static constexpr int NCS = 64 / RH, GR = RH / 4, WPG = BPP;   // dwords per 4-px granule
  const uint8_t *ref0;   // (po.x, po.y) of the reference plane
  long sr;               // reference stride, bytes
  int w, h, po_x, po_y;
  int mvx_min, mvx_max, mvy_min, mvy_max;
  MvCost mc;
  int r, slot;               // this lane: row, candidate slot
  uint32_t o[GR * WPG];      // source row (masked)
  uint32_t m[GR * WPG];      // pixel masks of this row: 0 beyond (w, h)

  __device__ __forceinline__ void init(const R1Plane &org, const R1Plane &ref, int lane) {
    r = lane & (RH - 1);
    slot = lane / RH;
    sr = (long)ref.stride * BPP;
    ref0 = px_addr<BPP>(ref, po_x, po_y);
    const uint8_t *op = px_addr<BPP>(org, po_x, po_y) + (long)r * org.stride * BPP;
#pragma unroll
    for (int g = 0; g < GR; g++) {
      int npx = w - 4 * g;
      npx = r < h ? (npx < 0 ? 0 : (npx > 4 ? 4 : npx)) : 0;
      if constexpr (BPP == 1) {
        m[g] = npx >= 4 ? 0xFFFFFFFFu : ((1u << (8 * npx)) - 1u);
        o[g] = m[g] ? ld_u32(op + 4 * g) & m[g] : 0u;
      } else {
        m[2 * g] = npx >= 2 ? 0xFFFFFFFFu : (npx == 1 ? 0xFFFFu : 0u);
        m[2 * g + 1] = npx >= 4 ? 0xFFFFFFFFu : (npx == 3 ? 0xFFFFu : 0u);
        U32x2 v = {0u, 0u};
        if (m[2 * g]) v = ld_u32x2(op + 8 * g);
        o[2 * g] = v.a & m[2 * g];
        o[2 * g + 1] = v.b & m[2 * g + 1];
      }
    }
  }

  // compute_mv_rd of this lane's slot candidate (me.rs:1386-1462) in two halves, so that a
  // search step can have the reference rows of SEVERAL candidate batches in flight before the
  // first SAD: a block's search is a chain of dependent steps and a step is one memory round
  // trip -- batches that do not depend on each other (a predictor list, a search pattern)
  // share one.  check: the MV range test of get_fullpel_mv_rd (full_search calls
  // compute_mv_rd without it).
  __device__ __forceinline__ bool fetch(int row, int col, bool valid, bool check, uint32_t *v) const {
    bool in = valid;
    if (check) in = in && col >= mvx_min && col <= mvx_max && row >= mvy_min && row <= mvy_max;
#pragma unroll
    for (int g = 0; g < GR * WPG; g++) v[g] = 0;
    if (in) {
      const uint8_t *p = ref0 + (long)(div8(row) + r) * sr + (long)div8(col) * BPP;
#pragma unroll
      for (int g = 0; g < GR; g++) {
        if constexpr (BPP == 1) {
          if (m[g]) v[g] = ld_u32(p + 4 * g);
        } else {
          if (m[2 * g]) {
            const U32x2 t = ld_u32x2(p + 8 * g);
            v[2 * g] = t.a;
            v[2 * g + 1] = t.b;
          }
        }
      }
    }
    return in;
  }
  // every lane of the slot returns the same (cost, sad)
  __device__ __forceinline__ void finish(const uint32_t *v, bool in, int row, int col,
                                         unsigned long long &cost, uint32_t &sad) const {
    uint32_t part = 0;
#pragma unroll
    for (int g = 0; g < GR * WPG; g++) {
      if constexpr (BPP == 1) part = __builtin_amdgcn_sad_u8(o[g], v[g] & m[g], part);
      else part = __builtin_amdgcn_sad_u16(o[g], v[g] & m[g], part);
    }
    part = group_sum<RH>(part);   // DPP inside a 16-lane row: no LDS round trips
    cost = in ? mc.cost(row, col, part) : COST_MAX;
    sad = in ? part : 0xFFFFFFFFu;
  }

  // one batch of NCS candidates after its rows have arrived: the slots' costs meet, the
  // lower candidate index wins ties, `if rd.cost < best.rd.cost { best = cand }`
  __device__ __forceinline__ void settle(const uint32_t *v, bool in, int idx, int row, int col, Msr &best,
                                         int *best_idx) const {
    unsigned long long cost;
    uint32_t sad;
    finish(v, in, row, col, cost, sad);
#ifndef R1_ME_SETTLE_SCALAR
#define R1_ME_SETTLE_SCALAR 1   // A/B switch (profiles/r06_ab_notes.md, ab5): 0 = the xor-shuffle tournament
#endif
#if R1_ME_SETTLE_SCALAR
    // Every lane of a slot holds its slot's (cost, sad, idx, row, col): the NCS costs go to SCALAR registers with
    // v_readlane (a few cycles each, no LDS crossbar round trip as a ds_bpermute shuffle is) and the tournament runs
    // on the scalar unit.  idx grows with the slot number, so "the lower index wins ties" is a strict less-than.
    {
      int ws = 0;
      unsigned long long wc = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(cost >> 32), 0) << 32) |
                              (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)cost, 0);
#pragma unroll
      for (int sl = 1; sl < NCS; sl++) {
        const unsigned long long c = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(cost >> 32), sl * RH) << 32) |
                                     (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)cost, sl * RH);
        if (c < wc) { wc = c; ws = sl; }
      }
      const int wl = ws * RH;
      cost = wc;
      idx = __builtin_amdgcn_readlane(idx, wl);
      row = __builtin_amdgcn_readlane(row, wl);
      col = __builtin_amdgcn_readlane(col, wl);
      sad = (uint32_t)__builtin_amdgcn_readlane((int)sad, wl);
    }
#else
#pragma unroll
    for (int s = RH; s < 64; s <<= 1) {
      // (v_permlane16/32_swap instead of these shuffles was tried: no gain, and together with the DPP
      // row sums above it produced wrong lanes at scale -- left alone)
      auto other = [&](uint32_t x) -> uint32_t { return (uint32_t)__shfl_xor((int)x, s, 64); };
      const unsigned long long oc = ((unsigned long long)other((uint32_t)(cost >> 32)) << 32) | other((uint32_t)cost);
      const int oi = (int)other((uint32_t)idx), orow = (int)other((uint32_t)row), ocol = (int)other((uint32_t)col);
      const uint32_t os = other(sad);
      if (oc < cost || (oc == cost && oi < idx)) {
        cost = oc; idx = oi; row = orow; col = ocol; sad = os;
      }
    }
#endif
    if (cost < best.cost) {
      best = Msr{row, col, cost, sad};
      if (best_idx) *best_idx = idx;
    }
  }

  // K batches with their loads issued back to back, settled in candidate order
  template <int K, class Gen>
  __device__ __forceinline__ void step(int base, int n, Gen gen, bool check, Msr &best, int *best_idx) const {
    uint32_t v[K][GR * WPG];
    int idx[K], row[K], col[K];
    bool in[K];
#pragma unroll
    for (int k = 0; k < K; k++) {
      idx[k] = base + k * NCS + slot;
      const bool valid = idx[k] < n;
      row[k] = col[k] = 0;
      if (valid) gen(idx[k], row[k], col[k]);
      in[k] = fetch(row[k], col[k], valid, check, v[k]);
    }
#pragma unroll
    for (int k = 0; k < K; k++) settle(v[k], in[k], idx[k], row[k], col[k], best, best_idx);
  }

  // two independent candidate lists (each at most one batch) evaluated in ONE round trip, each
  // into its own result: the second list is a speculation whose result the caller may drop
  template <class GenA, class GenB>
  __device__ __forceinline__ void scan_pair(int na, GenA gen_a, Msr &best_a, int nb, GenB gen_b, Msr &best_b,
                                            bool check) const {
#ifdef R1_ME_NO_SPEC
    if (true) {
#else
    if (na > NCS || nb > NCS) {
#endif
      scan(na, gen_a, check, best_a, nullptr);
      scan(nb, gen_b, check, best_b, nullptr);
      return;
    }
    uint32_t va[GR * WPG], vb[GR * WPG];
    int ra = 0, ca = 0, rb = 0, cb = 0;
    if (slot < na) gen_a(slot, ra, ca);
    if (slot < nb) gen_b(slot, rb, cb);
    const bool ia = fetch(ra, ca, slot < na, check, va), ib = fetch(rb, cb, slot < nb, check, vb);
    settle(va, ia, slot, ra, ca, best_a, nullptr);
    settle(vb, ib, slot, rb, cb, best_b, nullptr);
  }

  // `for cand in cands { if rd.cost < best.rd.cost { best = cand } }` over
  // n candidates produced by gen(idx, row, col); best_idx: index of the taken one.
  // batches in flight: GR * WPG registers each.  Measured (profiles/r02_me_batch_ab.log): three for
  // 16-row slots of 8-bit pixels (4 registers a batch); two everywhere else -- a third batch of 8 or 16
  // registers spills and made the 10-bit search 18 % slower than two
  static constexpr int KMAX = KM ? KM : ((GR * WPG <= 4) ? 3 : 2);
  template <class Gen>
  __device__ __forceinline__ void scan(int n, Gen gen, bool check, Msr &best, int *best_idx) const {
    int base = 0;
    while (base < n) {
      const int left = n - base;
      if (KMAX >= 3 && left > 2 * NCS) {
        step<KMAX >= 3 ? 3 : 2>(base, n, gen, check, best, best_idx);
        base += (KMAX >= 3 ? 3 : 2) * NCS;
      } else if (KMAX >= 2 && left > NCS) {
        step<2>(base, n, gen, check, best, best_idx);
        base += 2 * NCS;
      } else {
        step<1>(base, n, gen, check, best, best_idx);
        base += NCS;
      }
    }
  }
};

#ifdef R1_ME_PROF
// [0] predictor gather, [1] the candidate scan, [2] the diamond, [3] diamond iterations, [4] searches (non-extensive)
__device__ unsigned long long g_me_fine[8];
#endif
__constant__ int8_t kDiamond[4][2] = {{1, 0}, {0, 1}, {-1, 0}, {0, -1}};   // (row, col)
__constant__ int8_t kHexagon[6][2] = {{-2, 0}, {-1, 2}, {1, 2}, {2, 0}, {1, -2}, {-1, -2}};
__constant__ int8_t kSquare[8][2] = {{1, -1}, {1, 0}, {1, 1}, {0, -1}, {0, 1}, {-1, -1}, {-1, 0}, {-1, 1}};
// UMH_PATTERN as written in the reference (entry 13 repeats entry 7), me.rs:1153-1156
__constant__ int8_t kUmh[16][2] = {{4, -2}, {4, -1}, {4, 0}, {4, 1}, {4, 2}, {2, 3}, {0, 4}, {-2, 3},
                                   {-4, 2}, {-4, 1}, {-4, 0}, {-4, -1}, {-4, -2}, {-2, 3}, {0, -4}, {2, -3}};

template <class B>
__device__ __forceinline__ void fullpel_diamond_search(const B &b, Msr &cur) {
  // me.rs:955-1000: radius 2 until no candidate improves, then radius 1 until none does.  While at
  // radius 2 the four radius-1 candidates of the same centre are fetched in the same round trip:
  // they are what the next step evaluates whenever radius 2 brings no improvement (the common
  // case at the end of every search); otherwise the speculation is dropped.  Same evaluations,
  // same comparisons, same order -- one memory latency less per search.
  int radius_log2 = 1;
  for (;;) {
#ifdef R1_ME_PROF
    if (threadIdx.x == 0) atomicAdd(&g_me_fine[3], 1ull);
#endif
    Msr best = msr_empty();
    const int cr = cur.row, cc = cur.col;
    if (radius_log2 == 1) {
      Msr next = msr_empty();
      b.scan_pair(4, [&](int i, int &row, int &col) {
        row = (int16_t)(cr + (kDiamond[i][0] << 4));
        col = (int16_t)(cc + (kDiamond[i][1] << 4));
      }, best, 4, [&](int i, int &row, int &col) {
        row = (int16_t)(cr + (kDiamond[i][0] << 3));
        col = (int16_t)(cc + (kDiamond[i][1] << 3));
      }, next, true);
      if (cur.cost <= best.cost) {
        radius_log2 = 0;
        if (cur.cost <= next.cost) break;   // the radius-1 step of this centre
        cur = next;
      } else {
        cur = best;
      }
      continue;
    }
    b.scan(4, [&](int i, int &row, int &col) {
      row = (int16_t)(cr + (kDiamond[i][0] << 3));
      col = (int16_t)(cc + (kDiamond[i][1] << 3));
    }, true, best, nullptr);
    if (cur.cost <= best.cost) break;
    cur = best;
  }
}

template <class B>
__device__ __forceinline__ void hexagon_search(const B &b, Msr &cur) {
  int best_idx = 0;
  Msr best = msr_empty();
  {
    const int cr = cur.row, cc = cur.col;
    b.scan(6, [&](int i, int &row, int &col) {
      row = (int16_t)(cr + kHexagon[i][0] * 8);
      col = (int16_t)(cc + kHexagon[i][1] * 8);
    }, true, best, &best_idx);
  }
  while (best.cost < cur.cost) {
    cur = best;
    best = msr_empty();
    const int center = best_idx, cr = cur.row, cc = cur.col;
    int k = 0;
    // the three directions next to the one just taken; k is the visiting order
    b.scan(3, [&](int j, int &row, int &col) {
      const int i = (center + 5 + j) % 6;
      row = (int16_t)(cr + kHexagon[i][0] * 8);
      col = (int16_t)(cc + kHexagon[i][1] * 8);
    }, true, best, &k);
    best_idx = (center + 5 + k) % 6;
  }
  best = msr_empty();
  {
    const int cr = cur.row, cc = cur.col;
    b.scan(8, [&](int i, int &row, int &col) {
      row = (int16_t)(cr + kSquare[i][0] * 8);
      col = (int16_t)(cc + kSquare[i][1] * 8);
    }, true, best, nullptr);
  }
  if (best.cost < cur.cost) cur = best;
}

template <class B>
__device__ __forceinline__ void uneven_multi_hex_search(const B &b, Msr &cur, int me_range) {
  {
    const int cr = cur.row, cc = cur.col;
    const int nh = (me_range + 1) / 2;   // i = 1, 3, .. <= me_range
    b.scan(2 * nh, [&](int k, int &row, int &col) {
      const int i = 2 * (k >> 1) + 1;
      row = (int16_t)(cr + ((k & 1) ? 8 : -8) * i);   // the reference's "horizontal" line steps the row
      col = cc;
    }, true, cur, nullptr);
    const int nv = ((me_range >> 1) + 1) / 2;
    b.scan(2 * nv, [&](int k, int &row, int &col) {
      const int i = 2 * (k >> 1) + 1;
      row = cr;
      col = (int16_t)(cc + ((k & 1) ? 8 : -8) * i);
    }, true, cur, nullptr);
  }
  {   // 5x5: offsets in 1/8 pel as the reference has them (me.rs:1241-1247)
    const int cr = cur.row, cc = cur.col;
    b.scan(24, [&](int k, int &row, int &col) {
      const int j = k >= 12 ? k + 1 : k;   // skip the centre
      row = (int16_t)(cr + j / 5 - 2);
      col = (int16_t)(cc + j % 5 - 2);
    }, true, cur, nullptr);
  }
  {
    const int cr = cur.row, cc = cur.col;
    b.scan(16 * (me_range >> 2), [&](int k, int &row, int &col) {
      const int i = (k >> 4) + 1, p = k & 15;
      row = (int16_t)(cr + kUmh[p][0] * 8 * i);
      col = (int16_t)(cc + kUmh[p][1] * 8 * i);
    }, true, cur, nullptr);
  }
  hexagon_search(b, cur);
}

// full_search (me.rs:1464-1510): rows outer, every `step`-th window
template <class B>
__device__ __forceinline__ Msr full_search(const B &b, int x_lo, int x_hi, int y_lo, int y_hi, int step) {
  Msr best = msr_empty();
  if (x_hi < x_lo || y_hi < y_lo) return best;
  const int nx = (x_hi - x_lo) / step + 1, ny = (y_hi - y_lo) / step + 1;
  b.scan(nx * ny, [&](int k, int &row, int &col) {
    row = (int16_t)(8 * (int16_t)(y_lo + (k / nx) * step - b.po_y));
    col = (int16_t)(8 * (int16_t)(x_lo + (k % nx) * step - b.po_x));
  }, false, best, nullptr);
  return best;
}

struct TileView {
  R1MeStats *stats;
  const R1MeStats *prev;
  int cols_f, rows_f;          // FrameMEStats dims
  int tx, ty, tcols, trows;    // tile origin / size, 4x4 units
  const R1MeStats *rstats = nullptr;   // k_me_persist: the refined statistics (see there)
  __device__ __forceinline__ R1MeStats *at(int y, int x) const {
    return stats + (size_t)(ty + y) * cols_f + tx + x;
  }
};

__device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }
__device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }
__device__ __forceinline__ int iclamp(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// MotionEstimationSubsets (me.rs:364-384) of one wave, in LDS (the lists are
// indexed by lane-dependent candidate numbers): (row, col) pairs.
struct Subsets {
  uint32_t min_sad;
  int has_median, nb, nc;
  int16_t *median, *b, *c, *all;   // 1, <= 5, <= 5, <= 11 pairs
};
constexpr int kSubsetWords = 2 * (1 + 5 + 5 + 11);

// one MEStats entry: written by another wave of THIS workgroup a barrier ago
// (same CU, same L1: workgroup scope is enough -- an agent-scope fence per
// diagonal would write back / invalidate the XCD's L2 and made the 64-job
// launches 3.6x slower) or by another workgroup in an earlier launch (kernel
// boundaries make that visible)
template <bool AGENT = false>
__device__ __forceinline__ unsigned long long load_entry(const R1MeStats *s) {
  if constexpr (AGENT)   // k_me_persist: written by a wave anywhere on the device, in this launch
    return __hip_atomic_load((const unsigned long long *)s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else
    return __hip_atomic_load((const unsigned long long *)s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
template <bool AGENT = false>
__device__ __forceinline__ void load_stats(const R1MeStats *s, int &row, int &col, uint32_t &nsad) {
  const unsigned long long v = load_entry<AGENT>(s);
  row = (int16_t)(v & 0xFFFF);
  col = (int16_t)((v >> 16) & 0xFFFF);
  nsad = (uint32_t)(v >> 32);
}

// process_cand (me.rs:407-414) on a loaded entry
__device__ __forceinline__ void process_cand(unsigned long long v, const int *rng, uint32_t &min_sad,
                                             int16_t *out) {
  const int srow = (int16_t)(v & 0xFFFF), scol = (int16_t)((v >> 16) & 0xFFFF);
  const uint32_t ns = (uint32_t)(v >> 32);
  min_sad = ns < min_sad ? ns : min_sad;
  out[0] = (int16_t)iclamp(div8(srow) * 8, rng[2], rng[3]);
  out[1] = (int16_t)iclamp(div8(scol) * 8, rng[0], rng[1]);
}

// get_subset_predictors (me.rs:386-534).  The up to ten MEStats entries it samples (left, top,
// right, bottom, centre of this frame; the same five of the previous frame) are fetched by ten
// LANES in one load instruction and handed round by shuffles: one memory round trip instead of
// ten dependent ones (a block's search is a latency chain).  Entries of this frame were written
// by another wave of THIS workgroup a barrier ago (same CU, same L1: workgroup scope is enough
// -- an agent-scope fence per diagonal would write back / invalidate the XCD's L2 and made the
// 64-job launches 3.6x slower) or by another workgroup in an earlier launch (kernel boundaries
// make that visible).
template <bool AGENT = false>
__device__ __forceinline__ void get_subset_predictors(const TileView &t, int bx, int by, int pix_w, int pix_h,
                                      const int *rng, int corner, int ssdec, Subsets &s) {
  uint32_t min_sad = 0xFFFFFFFFu;
  s.nb = s.nc = s.has_median = 0;
  const int lane = threadIdx.x & 63;
  const int w = ((pix_w << ssdec) + MI - 1) >> 2, h = ((pix_h << ssdec) + MI - 1) >> 2;
  const int half_w = imin(w >> 1, t.tcols - 1 - bx), half_h = imin(h >> 1, t.trows - 1 - by);
  const int fx = t.tx + bx, fy = t.ty + by;
  const int hw = imin(w >> 1, t.cols_f - 1 - fx), hh = imin(h >> 1, t.rows_f - 1 - fy);
  const bool hp = t.prev != nullptr;
  bool ok[10];
  ok[0] = bx > 0;
  ok[1] = by > 0;
  ok[2] = corner && (corner & 2) && bx + w < t.tcols;
  ok[3] = corner && (corner & 4) && by + h < t.trows;
  ok[4] = corner != 0;
  ok[5] = hp && fx > 0;
  ok[6] = hp && fy > 0;
  ok[7] = hp && fx + w < t.cols_f;
  ok[8] = hp && fy + h < t.rows_f;
  ok[9] = hp;
  // (y, x) of entry `lane` in frame coordinates
  const int ey[10] = {t.ty + by + half_h, t.ty + by - 1, t.ty + by + half_h, t.ty + by + h, t.ty + by + half_h,
                      fy + hh, fy - 1, fy + hh, fy + h, fy + hh};
  const int ex[10] = {t.tx + bx - 1, t.tx + bx + half_w, t.tx + bx + w, t.tx + bx + half_w, t.tx + bx + half_w,
                      fx - 1, fx + hw, fx + w, fx + hw, fx + hw};
  int my_y = 0, my_x = 0;
  bool my_ok = false;
#pragma unroll
  for (int k = 0; k < 10; k++)
    if (lane == k) { my_y = ey[k]; my_x = ex[k]; my_ok = ok[k]; }
  unsigned long long mine = 0;
  if (my_ok) {
    const R1MeStats *base = lane < 5 ? (const R1MeStats *)t.stats : t.prev;
    if constexpr (AGENT) {
      // k_me_persist: the centre, and the right / bottom samples inside this block's own superblock,
      // are the REFINED vectors of the previous pass (second buffer); everything else the live array
      const bool same_sb = ((my_x - t.tx) >> 4) == (bx >> 4) && ((my_y - t.ty) >> 4) == (by >> 4);
      if (lane == 4 || ((lane == 2 || lane == 3) && same_sb)) base = t.rstats;
    }
    mine = load_entry<AGENT && true>(base + (size_t)my_y * t.cols_f + my_x);
  }
  auto entry = [&](int k) -> unsigned long long {
    return ((unsigned long long)(uint32_t)__shfl((int)(mine >> 32), k, 64) << 32) |
           (uint32_t)__shfl((int)(uint32_t)mine, k, 64);
  };
  if (ok[0]) process_cand(entry(0), rng, min_sad, s.b + 2 * s.nb++);
  if (ok[1]) process_cand(entry(1), rng, min_sad, s.b + 2 * s.nb++);
  if (ok[2]) process_cand(entry(2), rng, min_sad, s.b + 2 * s.nb++);
  if (ok[3]) process_cand(entry(3), rng, min_sad, s.b + 2 * s.nb++);
  if (corner) {
    s.has_median = 1;
    process_cand(entry(4), rng, min_sad, s.median);
  } else if (s.nb == 3) {
    // unreachable at INIT (at most left + top), kept for the rule's sake: median of three
    s.has_median = 1;
    for (int k = 0; k < 2; k++) {
      const int a = s.b[k], bb = s.b[2 + k], c = s.b[4 + k];
      s.median[k] = (int16_t)imax(imin(a, bb), imin(imax(a, bb), c));
    }
  }
  s.b[2 * s.nb] = 0;
  s.b[2 * s.nb + 1] = 0;
  s.nb++;
  if (hp) {
#pragma unroll
    for (int k = 5; k < 10; k++)
      if (ok[k]) process_cand(entry(k), rng, min_sad, s.c + 2 * s.nc++);
  }
  s.min_sad = (uint32_t)(((unsigned long long)min_sad * (unsigned long long)(pix_w * pix_h)) >> 14);
  // dec_mv (me.rs:519-532) and all_mvs (me.rs:371-383)
  int n = 0;
  if (s.has_median) {
    s.median[0] >>= ssdec;
    s.median[1] >>= ssdec;
    s.all[0] = s.median[0];
    s.all[1] = s.median[1];
    n = 1;
  }
  for (int i = 0; i < 2 * s.nb; i++) s.all[2 * n + i] = (s.b[i] >>= ssdec);
  n += s.nb;
  for (int i = 0; i < 2 * s.nc; i++) s.all[2 * n + i] = (s.c[i] >>= ssdec);
}

// Inlined, the running best by value.  As an out-of-line function (round 4) every call received the Block BY
// REFERENCE: the 112-byte block state (source rows, masks, MV range, cost model) was written to scratch memory per
// block and read back through flat loads by each of the up to four calls of a step, and the running best (20 bytes)
// was stored and reloaded around every call -- all on the dependent chain of the persistent kernel.
template <class B>
__device__ __forceinline__ Msr try_cands(const B &b, const int16_t *list, int n, const Msr best) {
  Msr r = msr_empty();
#ifdef R1_ME_PROF
  const unsigned long long f0 = wall_clock64();
#endif
  b.scan(n, [&](int i, int &row, int &col) { row = list[2 * i]; col = list[2 * i + 1]; }, true, r,
         nullptr);
#ifdef R1_ME_PROF
  const unsigned long long f1 = wall_clock64();
#endif
  fullpel_diamond_search(b, r);
#ifdef R1_ME_PROF
  if (threadIdx.x == 0) {
    atomicAdd(&g_me_fine[1], f1 - f0);
    atomicAdd(&g_me_fine[2], wall_clock64() - f1);
    atomicAdd(&g_me_fine[4], 1ull);
  }
#endif
  return r.cost < best.cost ? r : best;
}

template <class B, bool AGENT = false>
__device__ __forceinline__ Msr full_pixel_me(const B &b, const TileView &t, const R1MeParams &p, int bx, int by,
                             const int *rng, int corner, bool extensive, int ssdec,
                             int16_t *lds) {
  Subsets s;
  s.median = lds;
  s.b = lds + 2;
  s.c = lds + 12;
  s.all = lds + 22;
#ifdef R1_ME_PROF
  const unsigned long long g0 = wall_clock64();
#endif
  get_subset_predictors<AGENT>(t, bx, by, b.w, b.h, rng, corner, ssdec, s);
#ifdef R1_ME_PROF
  if (threadIdx.x == 0) atomicAdd(&g_me_fine[0], wall_clock64() - g0);
#endif
  Msr best = msr_empty();
  if (!extensive) {
    return try_cands(b, s.all, s.has_median + s.nb + s.nc, best);
  }
  // (min_sad as f32 * 1.2) as u32 + ((w * h) << (bit_depth - 8)), me.rs:773-774
  const uint32_t thresh = (uint32_t)__fmul_rn((float)s.min_sad, 1.2f) +
                          ((uint32_t)(b.w * b.h) << (p.bit_depth - 8));
  if (s.has_median) {
    best = try_cands(b, s.median, 1, best);
    if (best.sad < thresh) return best;
  }
  best = try_cands(b, s.b, s.nb, best);
  if (best.sad < thresh) return best;
  best = try_cands(b, s.c, s.nc, best);
  if (best.sad < thresh) return best;
  uneven_multi_hex_search(b, best, 24);
  if (!p.allow_full_search || best.sad < thresh) return best;
  const int range_x = (192 * p.me_range_scale) >> ssdec, range_y = (64 * p.me_range_scale) >> ssdec;
  const Msr r = full_search(b, b.po_x + imax(-range_x, div8(b.mvx_min)),
                            b.po_x + imin(range_x, div8(b.mvx_max)),
                            b.po_y + imax(-range_y, div8(b.mvy_min)),
                            b.po_y + imin(range_y, div8(b.mvy_max)), 4 >> ssdec);
  return r.cost < best.cost ? r : best;
}

// get_mv_range (me.rs:339-362) >> ssdec (me.rs:563-564)
__device__ __forceinline__ void mv_range(const R1MeParams &p, int fbx, int fby, int blk_w, int blk_h,
                                         int ssdec, int *r) {
  const int border_w = 128 + blk_w * 8, border_h = 128 + blk_h * 8;
  r[0] = imax(-fbx * (8 * MI) - border_w, -(1 << 14) + 1) >> ssdec;
  r[1] = imin(((p.w_in_b - fbx) - blk_w / MI) * (8 * MI) + border_w, (1 << 14) - 1) >> ssdec;
  r[2] = imax(-fby * (8 * MI) - border_h, -(1 << 14) + 1) >> ssdec;
  r[3] = imin(((p.h_in_b - fby) - blk_h / MI) * (8 * MI) + border_h, (1 << 14) - 1) >> ssdec;
}

template <class B>
__device__ __forceinline__ void setup_block(B &b, const R1Plane &org, const R1Plane &ref, const R1MeParams &p,
                                            const TileView &t, int bx, int by, int w, int h,
                                            int ssdec, int lane, int *rng) {
  const int fbx = t.tx + bx, fby = t.ty + by;
  mv_range(p, fbx, fby, w << ssdec, h << ssdec, ssdec, rng);
  b.w = w;
  b.h = h;
  b.po_x = (fbx * MI) >> ssdec;
  b.po_y = (fby * MI) >> ssdec;
  b.mvx_min = rng[0]; b.mvx_max = rng[1]; b.mvy_min = rng[2]; b.mvy_max = rng[3];
  // (a select, not p.lambda[ssdec]: a run-time index into the by-value parameter block sends the whole block to
  // scratch memory -- 112 bytes re-read on every step of the persistent kernel's dependent chain)
  b.mc.lambda = ssdec == 0 ? p.lambda[0] : (ssdec == 1 ? p.lambda[1] : p.lambda[2]);
  b.mc.allow_hp = p.allow_hp;
  // estimate_motion with pmv = None / refine_subsampled_motion_estimate: pmv = [0, 0]
  b.mc.pmv_row[0] = b.mc.pmv_row[1] = b.mc.pmv_col[0] = b.mc.pmv_col[1] = 0;
  b.init(org, ref, lane);
}

template <class B>
__device__ __forceinline__ void setup_block(B &b, const R1MeJob &job, const R1MeParams &p,
                                            const TileView &t, int bx, int by, int w, int h,
                                            int ssdec, int lane, int *rng) {
  setup_block(b, job.org[ssdec], job.ref[ssdec], p, t, bx, by, w, h, ssdec, lane, rng);
}

// save_me_stats (me.rs:324-337) with the normalisation of me.rs:268-270
template <bool AGENT = false, bool WIDE = false>
__device__ __forceinline__ void store_result(const TileView &t, int size_in_b, int bx, int by,
                                             const Msr &r, int w, int h, int ssdec, int lane) {
#ifndef R1_ME_FAST_STORE
#define R1_ME_FAST_STORE 1   // A/B switch (ab5): shifts where the block area / the entry count per row are powers of two
#endif
  // (wave-uniform branches: a 64-bit division and two 32-bit ones by run-time values are ~200 dependent instructions
  // between a search's last compare and the progress word its neighbours wait for)
  const uint32_t wh = (uint32_t)(w * h);
  const uint32_t nsad = (R1_ME_FAST_STORE && (wh & (wh - 1)) == 0)
                            ? (uint32_t)((((unsigned long long)r.sad) << 14) >> (31 - __clz(wh)))
                            : (uint32_t)((((unsigned long long)r.sad) << 14) / (unsigned long long)wh);
  const int nx = imin(bx + size_in_b, t.tcols) - bx, ny = imin(by + size_in_b, t.trows) - by;
  const bool nx_p2 = R1_ME_FAST_STORE && (nx & (nx - 1)) == 0;
  const int nx_l2 = 31 - __clz((unsigned)nx);
  R1MeStats v;
  v.row = (int16_t)(r.row << ssdec);
  v.col = (int16_t)(r.col << ssdec);
  v.normalized_sad = nsad;
  if constexpr (AGENT) {
    const unsigned long long bits = ((unsigned long long)v.normalized_sad << 32) |
                                    ((unsigned long long)(uint16_t)v.col << 16) | (uint16_t)v.row;
    // plain stores: the line stays in THIS XCD's L2, where the job's other waves (all on this
    // XCD, see k_me_persist) read it with L1-bypassing loads
    // (WIDE: the job's waves sit on any XCD -- agent-scope stores, written through)
    for (int i = lane; i < nx * ny; i += 64) {
      const int iy = nx_p2 ? i >> nx_l2 : i / nx, ix = nx_p2 ? i & (nx - 1) : i % nx;
      unsigned long long *d = (unsigned long long *)t.at(by + iy, bx + ix);
      if constexpr (WIDE) __hip_atomic_store(d, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else *d = bits;
    }
  } else {
    for (int i = lane; i < nx * ny; i += 64) {
      const int iy = nx_p2 ? i >> nx_l2 : i / nx, ix = nx_p2 ? i & (nx - 1) : i % nx;
      *t.at(by + iy, bx + ix) = v;
    }
  }
}

// One pass (log2b = 4, 3, 2 <-> ssdec 2, 1, 0) over the superblocks of one
// anti-diagonal of every job.  The three passes run SKEWED in the same launch:
// a superblock of pass q + 1 on diagonal d reads, besides its own area, the
// left / top neighbours (diagonal d - 1, already pass q + 1) and the right /
// bottom neighbours (diagonal d + 1, still pass q: get_subset_predictors samples
// edge midpoints only, me.rs:420-452, never a diagonal neighbour), so it may
// run as soon as pass q has finished diagonal d + 1 -- two launches behind.
// Pass q on diagonal d + 2 meanwhile touches diagonals d + 1 .. d + 3 only.
constexpr int kPassSkew = 2;
#ifdef R1_ME_PROF
// experiment build only: per pass, [0] workgroups, [1] sum of workgroup lifetimes, [2] longest
// workgroup, [3] sum of the refinement phase (100 MHz wall-clock ticks)
__device__ unsigned long long g_me_prof[3][4];
// k_me_persist, search rows: per pass [0] block searches, [1] set-up + wait for the neighbours, [2] the search
// (predictors, candidates, diamond), [3] result stores + publish (100 MHz ticks, summed over the waves)
__device__ unsigned long long g_me_step[3][4];
#endif
#ifndef R1_ME_DIAG_WAVES
#define R1_ME_DIAG_WAVES 5
#endif
template <int BPP>
__global__ __launch_bounds__(256, R1_ME_DIAG_WAVES) void k_me_diag(const R1MeJob *__restrict__ jobs,
                                                 const R1MeParams *__restrict__ pp,
                                                 R1MeStats *const *__restrict__ rbufs, int step) {
  const R1MeParams p = *pp;   // uniform: lives in SGPRs; in device memory so that the launch
                              // arguments (and with them the captured graph) do not depend on it
  // blockIdx.z = role.  0..2: the SEARCH of pass z on diagonal step - kPassSkew * z (pass q works
  // kPassSkew diagonals behind pass q - 1, see the host loop).  3, 4: the REFINEMENT
  // (refine_subsampled_sb_motion) for pass z - 2, ONE DIAGONAL AHEAD of that pass' search.  A
  // superblock's refinement depends on nothing but its own statistics of the previous pass, which
  // are final one launch before its search; done inside the search workgroup it was 12 of its
  // 53 us (the chain that bounds every launch), done here it runs beside the searches of the
  // diagonal before.  Its results must stay invisible to those searches -- their right / bottom
  // predictors in this superblock are the UNREFINED vectors -- so they go to a second buffer
  // (rbufs[job], same geometry as the statistics) that the superblock's own search copies in first.
  const int role = (int)blockIdx.z;
  const bool refine_role = role >= 3;
  const int pass = refine_role ? role - 2 : role;
  const int log2b = 4 - pass, diag = step - kPassSkew * pass + (refine_role ? 1 : 0);
  const R1MeJob &job = jobs[blockIdx.y];
  const int sbw = (job.tile_w + SB - 1) / SB, sbh = (job.tile_h + SB - 1) / SB;
  if (diag < 0) return;
  const int sby = (int)blockIdx.x + imax(0, diag - (sbw - 1)), sbx = diag - sby;
  if (sby >= sbh || sbx < 0 || sbx >= sbw) return;   // workgroup-uniform
  __shared__ int16_t sh_subsets[4][kSubsetWords];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#ifdef R1_ME_PROF
  const unsigned long long prof_t0 = wall_clock64();
  unsigned long long prof_t1 = prof_t0;
#endif
  const bool init = log2b == 4;
  const int ssdec = log2b - 2;
  TileView t{job.stats, job.prev, p.stats_cols, p.stats_rows, job.tile_x / MI, job.tile_y / MI,
             job.tile_w / MI, job.tile_h / MI};
  R1MeStats *const rbuf = rbufs[blockIdx.y];
  const int sb_w = imin(SB, job.tile_w - sbx * SB), sb_h = imin(SB, job.tile_h - sby * SB);

  if (refine_role) {
    // refine_subsampled_sb_motion: the previous pass' blocks at this resolution, into rbuf
    TileView tr = t;
    tr.stats = rbuf;
    const int sz = MI << (log2b + 1);
    const int nbx = (sb_w + sz - 1) / sz, nby = (sb_h + sz - 1) / sz;
    if (wave < nbx * nby) {
      const int x = (wave % nbx) * sz, y = (wave / nbx) * sz;
      const int bx = sbx * 16 + x / MI, by = sby * 16 + y / MI;
      const int w = imin(sz, sb_w - x + (1 << ssdec) - 1) >> ssdec;
      const int h = imin(sz, sb_h - y + (1 << ssdec) - 1) >> ssdec;
      Block<BPP, 32> b;
      int rng[4];
      setup_block(b, job, p, t, bx, by, w, h, ssdec, lane, rng);
      int mvr, mvc;
      uint32_t ns;
      load_stats(t.at(by, bx), mvr, mvc, ns);
      mvr >>= ssdec;
      mvc >>= ssdec;
      const Msr r = full_search(b, b.po_x + imax(div8(mvc) - 1, div8(b.mvx_min)),
                                b.po_x + imin(div8(mvc) + 2, div8(b.mvx_max)),
                                b.po_y + imax(div8(mvr) - 1, div8(b.mvy_min)),
                                b.po_y + imin(div8(mvr) + 2, div8(b.mvy_max)), 1);
      store_result(tr, 1 << (log2b + 1), bx, by, r, w, h, ssdec, lane);
    }
    return;
  }

  if (!init) {
    // the refinement of this superblock (previous launch, role 3 / 4) becomes visible now
    const int bx0 = sbx * 16, by0 = sby * 16;
    const int nx = imin(16, t.tcols - bx0), ny = imin(16, t.trows - by0);
    for (int i = threadIdx.x; i < nx * ny; i += 256) {
      const int y = i / nx, xx = i - y * nx;
      const size_t o = (size_t)(t.ty + by0 + y) * t.cols_f + t.tx + bx0 + xx;
      const unsigned long long v = *(const unsigned long long *)(rbuf + o);
      __hip_atomic_store((unsigned long long *)(t.stats + o), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __syncthreads();   // workgroup-scope release / acquire of the stats just written
#ifdef R1_ME_PROF
    prof_t1 = wall_clock64();
#endif
  }

  // estimate_sb_motion: raster order inside the superblock = anti-diagonals
  const int sz = MI << log2b;
  const int nbx = (sb_w + sz - 1) / sz, nby = (sb_h + sz - 1) / sz;
  for (int d = 0; d < nbx + nby - 1; d++) {
    const int j0 = imax(0, d - (nbx - 1));
    const int jy = j0 + wave, jx = d - jy;
    if (jy < nby && jx >= 0 && jx < nbx) {
      const int x = jx * sz, y = jy * sz;
      const int corner = init ? 0 : (1 | ((x & sz) ? 2 : 0) | ((y & sz) ? 4 : 0));
      const int bx = sbx * 16 + x / MI, by = sby * 16 + y / MI;
      const int w = imin(sz, sb_w - x + (1 << ssdec) - 1) >> ssdec;
      const int h = imin(sz, sb_h - y + (1 << ssdec) - 1) >> ssdec;
      Block<BPP, 16> b;
      int rng[4];
      setup_block(b, job, p, t, bx, by, w, h, ssdec, lane, rng);
      const Msr r = full_pixel_me(b, t, p, bx, by, rng, corner, init, ssdec, sh_subsets[wave]);
      store_result(t, 1 << log2b, bx, by, r, w, h, ssdec, lane);
    }
    __syncthreads();   // workgroup-scope release / acquire of the stats just written
  }
#ifdef R1_ME_PROF
  if (threadIdx.x == 0) {
    const unsigned long long t2 = wall_clock64();
    atomicAdd(&g_me_prof[pass][0], 1ull);
    atomicAdd(&g_me_prof[pass][1], t2 - prof_t0);
    atomicMax(&g_me_prof[pass][2], t2 - prof_t0);
    atomicAdd(&g_me_prof[pass][3], prof_t1 - prof_t0);
  }
#endif
}

// ---------------------------------------------------------------------------
// k_me_persist: the same three passes as ONE launch whose waves hand results over through
// progress counters in memory instead of kernel boundaries (R1MeParams::launch_mode 2 / 3).  A wave WALKS A
// ROW: the blocks of one block row of one pass (or the refinements of one row of the previous
// pass' blocks) from left to right.  The left neighbour is then the wave's own previous block;
// the only same-pass hand-over is the row above, which runs one block ahead -- in the steady
// state its result is already there when it is asked for, so the chain is rows + columns block
// steps (127 for a 960 x 1088 tile at 16 x 16) instead of 7 x 31 superblock-diagonal steps.
// Rows are taken from an atomic counter in an order in which everything a row waits for comes
// earlier (key = bottom edge of the row in 16-pixel cells + a per-pass offset): a wave only waits
// for rows that are already running, so there is no deadlock whatever the residency.
//   a search block (pass q) waits for: the row above having passed it; q > 0: the refinement of
//   its own parent block and of the parents of its right / bottom sample positions when those lie
//   in its own superblock (read refined), their pass q - 1 SEARCH when they lie in the next
//   superblock (read unrefined, from the live array).  A refinement waits for the pass q - 1
//   search of its block.
// Visibility.  PIN (launch_mode 2): every wave of a job sits on the XCD job % 8 (a wave asks
// HW_REG_XCC_ID where it is and takes rows from that XCD's list), so results are plain stores --
// they stay in that XCD's L2 -- read back with L1-bypassing (agent-scope) loads, and the progress
// word is a workgroup-scope store behind s_waitcnt(0).  !PIN (launch_mode 3, fewer jobs than XCDs):
// one list, any wave takes any row; results and progress words are agent-scope stores, written
// through.  Why not an __ATOMIC_RELEASE store / fence: at agent scope gfx950 spells it `buffer_wbl2 sc1`
// + s_waitcnt -- a write-back of the XCD's whole L2 per hand-over (round 2 saw that as a "hang": the
// waits ran out of patience behind it).  Measured in round 4 with every shared entry an agent-scope
// atomic, fence(release) before the progress word and fence(acquire) behind each wait (-DR1_ME_FORMAL=1,
// profiles/r04_me_fence_ab.md): bit-exact, no hang, and 1.3x (1 job) .. 4.1x (64 jobs) SLOWER.  The
// product keeps the ISA-level argument: the statistics are acknowledged by the memory system
// (s_waitcnt vmcnt(0)) before the progress word is issued -- written through (sc1) where the
// readers may sit on another XCD, left in the L2 that all readers share where they are pinned -- and
// every read of shared data is an L1-bypassing atomic load issued after the wait returned.  The C++
// model has no scope between "workgroup" and "agent" to say "this XCD", so the pinned mode stays a
// data race on paper; launch_mode 1 (kernel boundaries) is the formally clean path and the automatic
// fallback (r1_me_status / Context.estimate_frame_motion).
// Residency: TWO waves per SIMD (host: grid 2048) -- a searching wave is a dependent instruction
// chain that wants a VALU slot every ~8 cycles; a third and fourth wave on the SIMD stretch every
// step of a chain without slack -- so the kernel is not held to k_me_diag's 96
// registers and keeps three candidate batches in flight at every pixel size.
// Refined vectors live in the second buffer and are never copied: the samples pick their buffer.
// R1_ME_FORMAL (A/B build switch, profiles/r04_me_fence_ab.md): the hand-over spelled in the language's memory
// model -- every shared statistics entry an agent-scope atomic, __builtin_amdgcn_fence(release, "agent")
// before the progress word, fence(acquire, "agent") behind a successful wait.  On gfx950 the release is
// `buffer_wbl2 sc1` (write back the XCD's L2) and the acquire `buffer_inv sc1`, per block step.
#ifndef R1_ME_FORMAL
#define R1_ME_FORMAL 0
#endif
struct MeRow { uint16_t job; uint8_t kind, pad; uint16_t gy, nb; };   // kind 0..2 search, 3 / 4 refine for pass 1 / 2
struct MePersistArgs {
  const R1MeJob *jobs;
  const R1MeParams *params;
  R1MeStats *const *rbufs;
  const MeRow *rows;            // sorted per XCD: rows of the jobs with job % 8 == xcd (PIN; else one list), in key order
  int n_rows;
  int xoff[9];                  // rows of XCD x: [xoff[x], xoff[x + 1])
  unsigned int *counter;        // [8]: next row of each XCD
  unsigned int *prog;           // per row: epoch << 16 | blocks done
  const unsigned int *foff;     // [job][5]: offset of the job's progress array of each kind
  unsigned int epoch;           // 1 .. 65535
  unsigned int *err;            // set when a wait ran out of patience
  int spin;                     // polls before a wait gives up
};

__device__ __forceinline__ bool me_wait(const unsigned int *f, unsigned int epoch, unsigned int need, int spin) {
  for (int it = 0; it < spin; it++) {
    const unsigned int v = __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((v >> 16) == epoch && (v & 0xFFFFu) >= need) {
#if R1_ME_FORMAL
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
      return true;
    }
    if (it > 64) __builtin_amdgcn_s_sleep(8);
    else if (it > 4) __builtin_amdgcn_s_sleep(1);
  }
  return false;
}

// up to four progress counters polled by four lanes in ONE load per round (a wait is a memory
// round trip even when the counter is already there): a lane with mf != nullptr polls *mf for mn
__device__ __forceinline__ bool me_wait_lanes(const unsigned int *mf, unsigned int mn, unsigned int epoch, int spin) {
  for (int it = 0; it < spin; it++) {
    bool done = true;
    if (mf) {
      const unsigned int v = __hip_atomic_load(mf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      done = (v >> 16) == epoch && (v & 0xFFFFu) >= mn;
    }
    if (__all(done)) {
#if R1_ME_FORMAL
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
      return true;
    }
    if (it > 64) __builtin_amdgcn_s_sleep(8);
    else if (it > 4) __builtin_amdgcn_s_sleep(1);
  }
  return false;
}

// PIN: every job on one XCD (launch_mode 2); !PIN: one row list for the whole device, results and
// progress words written through at agent scope (launch_mode 3: fewer jobs than XCDs)
template <int BPP, bool PIN>
__global__ __launch_bounds__(64, 2) void k_me_persist(MePersistArgs a) {
  __shared__ int16_t sh_subsets[kSubsetWords];
  __shared__ unsigned int sh_item;
  const R1MeParams p = *a.params;
  const int lane = threadIdx.x;
  // every wave of a job sits on ONE XCD (the job's rows are handed out only to waves that find
  // themselves there), so a hand-over never leaves that XCD's L2
  const int xcd = PIN ? (__builtin_amdgcn_s_getreg(6164) & 7) : 0;   // hwreg(HW_REG_XCC_ID, 0, 4)
  // this XCD's slice of the row list, by selects: a.xoff[xcd] -- a run-time index into the kernel's by-value
  // argument block -- made the compiler keep a copy of the whole block in scratch memory
  int x_lo = a.xoff[0], x_hi = a.xoff[1];
#pragma unroll
  for (int k = 1; k < 8; k++)
    if (xcd == k) { x_lo = a.xoff[k]; x_hi = a.xoff[k + 1]; }
  for (;;) {
    if (lane == 0) sh_item = atomicAdd(a.counter + xcd, 1u);
    __syncthreads();
    const unsigned int ii = __builtin_amdgcn_readfirstlane(sh_item) + (unsigned int)x_lo;
    __syncthreads();
    if (ii >= (unsigned int)x_hi) return;
    const MeRow row = a.rows[ii];
    // the row's view of its job BY VALUE (scalar registers): nothing of it is re-read per block
    // behind the stores and atomics of the loop
    const R1MeJob &gjob = a.jobs[row.job];
    struct { R1MeStats *stats; const R1MeStats *prev; int tile_x, tile_y, tile_w, tile_h; } job =
        {gjob.stats, gjob.prev, gjob.tile_x, gjob.tile_y, gjob.tile_w, gjob.tile_h};
    const unsigned int *fo = a.foff + 5 * row.job;
    const bool refine = row.kind >= 3;
    const int pass = refine ? row.kind - 2 : row.kind;         // the pass the row belongs to
    const int log2b = 4 - pass, ssdec = log2b - 2;
    const bool init = log2b == 4;
    const R1Plane org = gjob.org[ssdec], ref = gjob.ref[ssdec];
    TileView t{job.stats, job.prev, p.stats_cols, p.stats_rows, job.tile_x / MI, job.tile_y / MI,
               job.tile_w / MI, job.tile_h / MI};
    t.rstats = a.rbufs[row.job];
    unsigned int *mine = a.prog + fo[row.kind] + row.gy;
    bool ok = true;
    for (int gx = 0; gx < row.nb; gx++) {
      if (refine) {
        // refine_subsampled_motion_estimate of block (gx, gy) of pass `pass - 1`
        const int sz = MI << (log2b + 1);
        const int x = gx * sz, y = row.gy * sz;                // tile px
        const int sbx = x / SB, sby = y / SB;
        const int sb_w = imin(SB, job.tile_w - sbx * SB), sb_h = imin(SB, job.tile_h - sby * SB);
        const int xin = x - sbx * SB, yin = y - sby * SB;
        const int bx = x / MI, by = y / MI;
        const int w = imin(sz, sb_w - xin + (1 << ssdec) - 1) >> ssdec;
        const int h = imin(sz, sb_h - yin + (1 << ssdec) - 1) >> ssdec;
        Block<BPP, 32, 3> b;
        int rng[4];
        setup_block(b, org, ref, p, t, bx, by, w, h, ssdec, lane, rng);
        if (lane == 0) ok = me_wait(a.prog + fo[pass - 1] + row.gy, a.epoch, gx + 1, a.spin) && ok;
        ok = __shfl((int)ok, 0, 64) != 0;
        int mvr, mvc;
        uint32_t ns;
        load_stats<true>(t.at(by, bx), mvr, mvc, ns);
        mvr >>= ssdec;
        mvc >>= ssdec;
        const Msr r = full_search(b, b.po_x + imax(div8(mvc) - 1, div8(b.mvx_min)),
                                  b.po_x + imin(div8(mvc) + 2, div8(b.mvx_max)),
                                  b.po_y + imax(div8(mvr) - 1, div8(b.mvy_min)),
                                  b.po_y + imin(div8(mvr) + 2, div8(b.mvy_max)), 1);
        TileView tr = t;
        tr.stats = (R1MeStats *)t.rstats;
        store_result<true, !PIN || R1_ME_FORMAL>(tr, 1 << (log2b + 1), bx, by, r, w, h, ssdec, lane);
      } else {
        const int sz = MI << log2b;
        const int x = gx * sz, y = row.gy * sz;
        const int sbx = x / SB, sby = y / SB;
        const int sb_w = imin(SB, job.tile_w - sbx * SB), sb_h = imin(SB, job.tile_h - sby * SB);
        const int xin = x - sbx * SB, yin = y - sby * SB;
        const int bx = x / MI, by = y / MI;
        const int w = imin(sz, sb_w - xin + (1 << ssdec) - 1) >> ssdec;
        const int h = imin(sz, sb_h - yin + (1 << ssdec) - 1) >> ssdec;
        // everything that does not depend on the neighbours first: source rows, masks, MV range
#ifdef R1_ME_PROF
        const unsigned long long st0 = wall_clock64();
#endif
        Block<BPP, 16, 3> b;
        int rng[4];
        setup_block(b, org, ref, p, t, bx, by, w, h, ssdec, lane, rng);
        {
          // up to four progress words, lane k polling the k-th: each lane's own (pointer, count) pair is set
          // directly -- lists indexed by a run-time count lived in scratch memory, a store and a load round trip
          // on every step of the chain
          const unsigned int *mf = nullptr;
          unsigned int mn = 0;
          if (row.gy > 0 && lane == 0) { mf = a.prog + fo[pass] + row.gy - 1; mn = gx + 1; }
          if (!init) {
            const int psz = sz * 2;                            // the parents' size, px
            if (lane == 1) { mf = a.prog + fo[2 + pass] + y / psz; mn = x / psz + 1; }   // own parent refined
            // get_subset_predictors' right / bottom sample positions (me.rs:420-452), tile px
            const int wu = ((w << ssdec) + MI - 1) >> 2, hu = ((h << ssdec) + MI - 1) >> 2;   // 4x4 units
            const int half_w = imin(wu >> 1, t.tcols - 1 - bx), half_h = imin(hu >> 1, t.trows - 1 - by);
            if (bx + wu < t.tcols && lane == 2) {
              const int px = (bx + wu) * MI, py = (by + half_h) * MI;
              const bool same = px / SB == sbx && py / SB == sby;
              mf = a.prog + fo[same ? 2 + pass : pass - 1] + py / psz; mn = px / psz + 1;
            }
            if (by + hu < t.trows && lane == 3) {
              const int px = (bx + half_w) * MI, py = (by + hu) * MI;
              const bool same = px / SB == sbx && py / SB == sby;
              mf = a.prog + fo[same ? 2 + pass : pass - 1] + py / psz; mn = px / psz + 1;
            }
          }
          if (row.gy > 0 || !init) ok = me_wait_lanes(mf, mn, a.epoch, a.spin) && ok;
        }
        const int corner = init ? 0 : (1 | ((xin & sz) ? 2 : 0) | ((yin & sz) ? 4 : 0));
#ifdef R1_ME_PROF
        const unsigned long long st1 = wall_clock64();
#endif
        const Msr r = full_pixel_me<Block<BPP, 16, 3>, true>(b, t, p, bx, by, rng, corner, init, ssdec, sh_subsets);
#ifdef R1_ME_PROF
        const unsigned long long st2 = wall_clock64();
#endif
        store_result<true, !PIN || R1_ME_FORMAL>(t, 1 << log2b, bx, by, r, w, h, ssdec, lane);
#ifdef R1_ME_PROF
        __builtin_amdgcn_s_waitcnt(0);
        if (lane == 0) {
          const unsigned long long st3 = wall_clock64();
          atomicAdd(&g_me_step[pass][0], 1ull);
          atomicAdd(&g_me_step[pass][1], st1 - st0);
          atomicAdd(&g_me_step[pass][2], st2 - st1);
          atomicAdd(&g_me_step[pass][3], st3 - st2);
        }
#endif
      }
      // publish: the statistics first (agent-scope stores, acknowledged), then the progress
#if R1_ME_FORMAL
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      if (lane == 0)
        __hip_atomic_store(mine, (a.epoch << 16) | (unsigned int)(gx + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
      asm volatile("" ::: "memory");   // no result store may sink below the wait, no progress store rise above it
      __builtin_amdgcn_s_waitcnt(0);
      asm volatile("" ::: "memory");
      if (lane == 0) {
        const unsigned int word = (a.epoch << 16) | (unsigned int)(gx + 1);
        if constexpr (PIN) __hip_atomic_store(mine, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else __hip_atomic_store(mine, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
#endif
    }
    // the error word is host-mapped pinned memory (one per ring slot): the host reads it where the
    // slot's event is waited for, without a copy (r1_me_status / the slot's reuse)
    if (lane == 0 && !ok) __hip_atomic_store(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// ---------------------------------------------------------------------------
// estimate_motion with pmv = Some(..) (the RDO-time call, src/rdo.rs:1183-1196):
// independent blocks of any BlockSize up to 64x64, full resolution.  One
// WORKGROUP of 4 waves per block:
//   * the source block sits in LDS; in the full-pel steps wave v / slot s takes
//     candidate 4-or-less * v + s of a step (rows of a candidate on max(16, h)
//     lanes), the per-wave winners meet in LDS;
//   * in the sub-pel diamond (me.rs:1311-1383) wave v owns candidate v of the
//     four: it stages the (w+7) x (h+7) reference window in LDS, runs put_8tap
//     (mc_common.hpp) into an LDS tile and takes SATD / SAD of it against the
//     source -- the prediction never exists in HBM.
template <int BPP>
struct WgBlock {
  const uint8_t *ref0;
  long sr;
  int w, h, po_x, po_y;
  int mvx_min, mvx_max, mvy_min, mvy_max;
  MvCost mc;
  int wave, lane, RH, r, slot, ncs;
  const uint8_t *org;             // LDS, row stride w * BPP
  unsigned long long *red;        // LDS, 4 x 3 words: cost, (idx, sad), (row, col)

  __device__ __forceinline__ void eval(int row, int col, bool valid, bool check,
                                       unsigned long long &cost, uint32_t &sad) const {
    bool in = valid;
    if (check) in = in && col >= mvx_min && col <= mvx_max && row >= mvy_min && row <= mvy_max;
    uint32_t part = 0;
    if (in && r < h) {
      const uint8_t *p = ref0 + (long)(div8(row) + r) * sr + (long)div8(col) * BPP;
      const uint8_t *o = org + r * w * BPP;
      for (int g = 0; g < w / 4; g++) {
        if constexpr (BPP == 1) {
          part = __builtin_amdgcn_sad_u8(*(const uint32_t *)(o + 4 * g), ld_u32(p + 4 * g), part);
        } else {
          const U32x2 v = ld_u32x2(p + 8 * g);
          part = __builtin_amdgcn_sad_u16(*(const uint32_t *)(o + 8 * g), v.a, part);
          part = __builtin_amdgcn_sad_u16(*(const uint32_t *)(o + 8 * g + 4), v.b, part);
        }
      }
    }
    for (int s = 1; s < RH; s <<= 1) part += __shfl_xor(part, s, 64);
    cost = in ? mc.cost(row, col, part) : COST_MAX;
    sad = in ? part : 0xFFFFFFFFu;
  }

  // workgroup-wide argmin of (cost, idx): every thread returns the winner
  __device__ __forceinline__ void wg_min(unsigned long long &cost, int &idx, int &row, int &col,
                                         uint32_t &sad) const {
    if (lane == 0) {
      red[3 * wave] = cost;
      red[3 * wave + 1] = ((unsigned long long)(uint32_t)idx << 32) | sad;
      red[3 * wave + 2] = ((unsigned long long)(uint32_t)row << 32) | (uint32_t)col;
    }
    __syncthreads();
    int best = 0;
    for (int v = 1; v < 4; v++) {
      const unsigned long long c = red[3 * v], cb = red[3 * best];
      if (c < cb || (c == cb && (int)(red[3 * v + 1] >> 32) < (int)(red[3 * best + 1] >> 32))) best = v;
    }
    cost = red[3 * best];
    idx = (int)(red[3 * best + 1] >> 32);
    sad = (uint32_t)red[3 * best + 1];
    row = (int)(red[3 * best + 2] >> 32);
    col = (int)(uint32_t)red[3 * best + 2];
    __syncthreads();
  }

  // the shared search code's two-list step (Block::scan_pair): here simply one list after the other
  template <class GenA, class GenB>
  __device__ __forceinline__ void scan_pair(int na, GenA gen_a, Msr &best_a, int nb, GenB gen_b, Msr &best_b,
                                            bool check) const {
    scan(na, gen_a, check, best_a, nullptr);
    scan(nb, gen_b, check, best_b, nullptr);
  }

  template <class Gen>
  __device__ __forceinline__ void scan(int n, Gen gen, bool check, Msr &best, int *best_idx) const {
    for (int base = 0; base < n; base += 4 * ncs) {
      int idx = base + wave * ncs + slot;
      const bool valid = idx < n;
      int row = 0, col = 0;
      if (valid) gen(idx, row, col);
      unsigned long long cost;
      uint32_t sad;
      eval(row, col, valid, check, cost, sad);
      for (int s = RH; s < 64; s <<= 1) {
        const unsigned long long oc =
            ((unsigned long long)(uint32_t)__shfl_xor((int)(cost >> 32), s, 64) << 32) |
            (uint32_t)__shfl_xor((int)(uint32_t)cost, s, 64);
        const int oi = __shfl_xor(idx, s, 64), orow = __shfl_xor(row, s, 64),
                  ocol = __shfl_xor(col, s, 64);
        const uint32_t os = (uint32_t)__shfl_xor((int)sad, s, 64);
        if (oc < cost || (oc == cost && oi < idx)) {
          cost = oc; idx = oi; row = orow; col = ocol; sad = os;
        }
      }
      wg_min(cost, idx, row, col, sad);
      if (cost < best.cost) {
        best = Msr{row, col, cost, sad};
        if (best_idx) *best_idx = idx;
      }
    }
  }

  // this wave: put_8tap of the block at (sx, sy) + fractions into `pred`, then
  // get_satd / get_sad against the source (compute_mv_rd's distortion)
  __device__ __forceinline__ uint32_t predict_dist(const R1Plane &ref, uint8_t *win, uint8_t *pred,
                                                   int sx, int sy, int col_frac, int row_frac,
                                                   int mode, bool use_satd) const {
    // 8-bit 32 / 64-sized blocks: the fused candidate kernel's column filter + DPP SATD, a wave per
    // candidate (64x64 0.27 -> 0.20 ms, 32x32 0.50 -> 0.42 ms for every block of a 4K frame).  The
    // 16-bit variant of the same lost at 32x32 (0.56 -> 0.70 ms: 181 VGPRs, spills, 29 k instructions
    // of code) and stays on the generic path.
    if constexpr (BPP == 1) {
      if ((w == 32 || w == 64) && (h == 32 || h == 64)) {
        uint32_t s = 0;
#define R1_WP(W_, H_) s = subpel_group_dist<1, W_, H_, 8, 64>(win, ref, sx, sy, col_frac, row_frac, mode, lane, lane, org, use_satd, 8)
        if (w == 32 && h == 32) R1_WP(32, 32);
        else if (w == 64 && h == 64) R1_WP(64, 64);
        else if (w == 64) R1_WP(64, 32);
        else R1_WP(32, 64);
#undef R1_WP
        s = group_sum<64>(s);
        return use_satd ? (s + 4u) >> 3 : s;
      }
    }
    const int ws = (((w + 7) * BPP + 3) >> 2) << 2;
    r1mc::stage_window<BPP>(win, ws, ref, sx, sy, w, h, lane, 64);
    __builtin_amdgcn_wave_barrier();
    if (lane < w) {
      if constexpr (BPP == 1)
        r1mc::mc_column<BPP, false, 0>(win, ws, lane, w, h, col_frac, row_frac, mode, mode,
                                       ref.bit_depth,
                                       [&](int rr, int32_t v) { pred[rr * w + lane] = (uint8_t)v; });
      else
        r1mc::mc_column<BPP, false, 0>(win, ws, lane, w, h, col_frac, row_frac, mode, mode,
                                       ref.bit_depth, [&](int rr, int32_t v) {
                                         ((uint16_t *)pred)[rr * w + lane] = (uint16_t)v;
                                       });
    }
    __builtin_amdgcn_wave_barrier();
    const bool small = (w < h ? w : h) == 4;
    const int ts = small ? 4 : 8, ntx = w / ts, nt = ntx * (h / ts);
    uint32_t s = 0;
    if (lane < nt) {
      const int tx = lane % ntx, ty = lane / ntx;
      const size_t off = ((size_t)ty * ts * w + (size_t)tx * ts) * BPP, st = (size_t)w * BPP;
      if (use_satd)
        s = small ? r1dist::tile_dist<BPP, 4, true>(org + off, st, pred + off, st)
                  : r1dist::tile_dist<BPP, 8, true>(org + off, st, pred + off, st);
      else
        s = small ? r1dist::tile_dist<BPP, 4, false>(org + off, st, pred + off, st)
                  : r1dist::tile_dist<BPP, 8, false>(org + off, st, pred + off, st);
    }
    for (int m = 1; m < 64; m <<= 1) s += __shfl_xor(s, m, 64);
    const int ln = small ? 2 : 3;
    return use_satd ? (s + ((1u << ln) >> 1)) >> ln : s;
  }
};

template <int BPP>
__global__ __launch_bounds__(256) void k_me_blocks(R1MeJob job, R1MeParams p,
                                                   const R1MeBlockCand *__restrict__ cands,
                                                   int max_w, int max_h, int use_satd,
                                                   int filter_mode,
                                                   R1MeResult *__restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  __shared__ int16_t sh_subsets[4][kSubsetWords];
  __shared__ unsigned long long sh_red[12];
  const R1MeBlockCand cd = cands[blockIdx.x];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int w = cd.w, h = cd.h;
  if (w > max_w || h > max_h || w < 4 || h < 4 || (w & (w - 1)) || (h & (h - 1))) {
    // not a block this launch was sized for: an empty MotionSearchResult
    if (threadIdx.x == 0) out[blockIdx.x] = R1MeResult{0, 0, 0xFFFFFFFFu, COST_MAX};
    return;
  }
  TileView t{job.stats, job.prev, p.stats_cols, p.stats_rows, job.tile_x / MI, job.tile_y / MI,
             job.tile_w / MI, job.tile_h / MI};
  // LDS: source block | 4 x (window | prediction)
  const int ws = (((w + 7) * BPP + 3) >> 2) << 2;
  const int org_bytes = (w * h * BPP + 15) & ~15, win_bytes = ((h + 7) * ws + 15) & ~15;
  uint8_t *org_l = smem;
  uint8_t *win = smem + org_bytes + wave * (win_bytes + org_bytes);
  uint8_t *pred = win + win_bytes;

  WgBlock<BPP> b;
  int rng[4];
  const int fbx = t.tx + cd.bx, fby = t.ty + cd.by;
  mv_range(p, fbx, fby, w, h, 0, rng);
  b.w = w; b.h = h;
  b.po_x = fbx * MI; b.po_y = fby * MI;
  b.mvx_min = rng[0]; b.mvx_max = rng[1]; b.mvy_min = rng[2]; b.mvy_max = rng[3];
  b.mc.lambda = p.lambda[0];
  b.mc.allow_hp = p.allow_hp;
  for (int k = 0; k < 2; k++) { b.mc.pmv_row[k] = cd.pmv[k][0]; b.mc.pmv_col[k] = cd.pmv[k][1]; }
  const R1Plane &org = job.org[0], &ref = job.ref[0];
  b.sr = (long)ref.stride * BPP;
  b.ref0 = px_addr<BPP>(ref, b.po_x, b.po_y);
  b.wave = wave; b.lane = lane;
  b.RH = h < 16 ? 16 : h;
  b.r = lane % b.RH; b.slot = lane / b.RH; b.ncs = 64 / b.RH;
  b.org = org_l;
  b.red = sh_red;
  {   // source block -> LDS (4-px granules)
    const int gpr = w / 4, ng = gpr * h;
    const uint8_t *o0 = px_addr<BPP>(org, b.po_x, b.po_y);
    for (int i = threadIdx.x; i < ng; i += 256) {
      const int rr = i / gpr, g = i - rr * gpr;
      const uint8_t *src = o0 + (long)rr * org.stride * BPP + g * 4 * BPP;
      if constexpr (BPP == 1) *(uint32_t *)(org_l + rr * w + 4 * g) = ld_u32(src);
      else {
        const U32x2 v = ld_u32x2(src);
        *(uint32_t *)(org_l + (rr * w + 4 * g) * 2) = v.a;
        *(uint32_t *)(org_l + (rr * w + 4 * g) * 2 + 4) = v.b;
      }
    }
  }
  __syncthreads();

  Msr best = full_pixel_me(b, t, p, cd.bx, cd.by, rng, cd.corner, false, 0, sh_subsets[wave]);

  auto in_range = [&](int row, int col) {
    return col >= b.mvx_min && col <= b.mvx_max && row >= b.mvy_min && row <= b.mvy_max;
  };
  if (use_satd) {
    // get_fullpel_mv_rd(best.mv, use_satd) (me.rs:596-613); every wave computes the same
    if (!in_range(best.row, best.col)) {
      best.cost = COST_MAX;
      best.sad = 0xFFFFFFFFu;
    } else {
      const uint32_t d = b.predict_dist(ref, win, pred, b.po_x + div8(best.col),
                                        b.po_y + div8(best.row), 0, 0, filter_mode, true);
      best.sad = d;
      best.cost = b.mc.cost(best.row, best.col, d);
    }
  }
  // subpel_diamond_search: wave v <-> DIAMOND_R1_PATTERN_SUBPEL[v]
  int radius_log2 = 2;
  const int end_log2 = p.allow_hp ? 0 : 1;
  for (;;) {
    int row = (int16_t)(best.row + (kDiamond[wave][0] << radius_log2));
    int col = (int16_t)(best.col + (kDiamond[wave][1] << radius_log2));
    unsigned long long cost = COST_MAX;
    uint32_t sad = 0xFFFFFFFFu;
    if (in_range(row, col)) {
      // get_mv_params (src/predict.rs:284-297): floor offset, 1/16 fraction
      sad = b.predict_dist(ref, win, pred, b.po_x + (col >> 3), b.po_y + (row >> 3),
                           (col << 1) & 15, (row << 1) & 15, filter_mode, use_satd != 0);
      cost = b.mc.cost(row, col, sad);
    }
    int idx = wave;
    b.wg_min(cost, idx, row, col, sad);
    if (best.cost <= cost) {
      if (radius_log2 == end_log2) break;
      radius_log2--;
    } else {
      best = Msr{row, col, cost, sad};
    }
  }
  if (threadIdx.x == 0) {
    R1MeResult r;
    r.row = (int16_t)best.row;
    r.col = (int16_t)best.col;
    r.sad = best.sad;
    r.cost = best.cost;
    out[blockIdx.x] = r;
  }
}

// Blocks up to 16x16: ONE WAVE per block (four independent blocks per
// workgroup, no workgroup barrier anywhere).  Full-pel steps run on the tile
// ME's wave-level engine (source rows in registers, 4 candidates x 16 rows);
// in the sub-pel diamond the four 16-lane groups of the wave each own one
// candidate: window staging, put_8tap (lane = column), SATD / SAD with one lane
// per Hadamard tile, all inside the group; the four costs meet by shuffles.
// PHASE 0: the whole search in one launch (the product path).  PHASE 1 / 2 (round 6 experiment, kept behind
// R1_ME_SMALL_SPLIT): the full-pel search and the sub-pel refinement as two launches -- the result of the first travels
// through `out` (row, col, sad, cost: the whole MotionSearchResult).  What it showed: the full-pel half needs 61 / 64
// VGPRs; the 168 VGPRs + 168 / 196 B of scratch (381 MB of scratch writes per 4K launch, profiles/r05_pmc_frame.json)
// are the sub-pel half's alone (eight inlined (size, bit depth) forms of the fused-candidate column filter + SATD), and
// giving it 223 VGPRs (two workgroups per CU, 0 B scratch) is SLOWER than three with the spills: the launch is a
// latency chain per block like the tile search, the scratch stores are not on it (profiles/r06_ab_notes.md, ab3).
#ifndef R1_ME_SMALL_WAVES
#define R1_ME_SMALL_WAVES(BPP) 3   // A/B: workgroups the register allocator makes room for (x 4 waves)
#endif
#ifndef R1_ME_SMALL_WAVES_P1
#define R1_ME_SMALL_WAVES_P1 4
#endif
#ifndef R1_ME_SMALL_WAVES_P2
#define R1_ME_SMALL_WAVES_P2 3
#endif
template <int BPP, int PHASE>
__global__ __launch_bounds__(256, PHASE == 0 ? R1_ME_SMALL_WAVES(BPP) : (PHASE == 1 ? R1_ME_SMALL_WAVES_P1 : R1_ME_SMALL_WAVES_P2))
void k_me_blocks_small(R1MeJob job, R1MeParams p,
                                                         const R1MeBlockCand *__restrict__ cands,
                                                         int n, int max_w, int max_h, int use_satd,
                                                         int filter_mode,
                                                         R1MeResult *__restrict__ out) {
  constexpr int WS_MAX = (((16 + 7) * BPP + 3) >> 2) << 2;
  constexpr int GROUP_BYTES = ((23 * WS_MAX + 15) & ~15) + 16 * 16 * BPP;   // window + prediction
  __shared__ __attribute__((aligned(16))) uint8_t sh_grp[PHASE == 1 ? 1 : 4][PHASE == 1 ? 1 : 4][PHASE == 1 ? 16 : GROUP_BYTES];
  __shared__ int16_t sh_subsets[PHASE == 2 ? 1 : 4][kSubsetWords];
  __shared__ __attribute__((aligned(16))) uint8_t sh_src[PHASE == 1 ? 1 : 4][PHASE == 1 ? 16 : 16 * 16 * BPP];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // an XCD takes a contiguous run of the block list (common.hpp): callers list blocks in raster order, and the search
  // windows of neighbouring blocks overlap -- dealt round-robin, every XCD's L2 fetched the whole reference
#ifndef R1_ME_SMALL_XCD_RUNS
#define R1_ME_SMALL_XCD_RUNS 1   // A/B switch
#endif
  const long long bi = (long long)(R1_ME_SMALL_XCD_RUNS ? xcd_run_item(blockIdx.x, gridDim.x) : (int)blockIdx.x) * 4 + wave;
  if (bi >= n) return;                         // wave-uniform; no barriers below
  const R1MeBlockCand cd = cands[bi];
  const int w = cd.w, h = cd.h;
  if (w > max_w || h > max_h || w > 16 || h > 16 || w < 4 || h < 4 || (w & (w - 1)) || (h & (h - 1))) {
    if (lane == 0) out[bi] = R1MeResult{0, 0, 0xFFFFFFFFu, COST_MAX};
    return;
  }
  TileView t{job.stats, job.prev, p.stats_cols, p.stats_rows, job.tile_x / MI, job.tile_y / MI,
             job.tile_w / MI, job.tile_h / MI};
  const R1Plane &org = job.org[0], &ref = job.ref[0];
  Block<BPP, 16> b;
  int rng[4];
  const int fbx = t.tx + cd.bx, fby = t.ty + cd.by;
  mv_range(p, fbx, fby, w, h, 0, rng);
  b.w = w; b.h = h;
  b.po_x = fbx * MI; b.po_y = fby * MI;
  b.mvx_min = rng[0]; b.mvx_max = rng[1]; b.mvy_min = rng[2]; b.mvy_max = rng[3];
  b.mc.lambda = p.lambda[0];
  b.mc.allow_hp = p.allow_hp;
  for (int k = 0; k < 2; k++) { b.mc.pmv_row[k] = cd.pmv[k][0]; b.mc.pmv_col[k] = cd.pmv[k][1]; }
  Msr best;
  if constexpr (PHASE != 2) {
    b.init(org, ref, lane);
    best = full_pixel_me(b, t, p, cd.bx, cd.by, rng, cd.corner, false, 0, sh_subsets[PHASE == 2 ? 0 : wave]);
    if constexpr (PHASE == 1) {
      if (lane == 0) {
        R1MeResult r;
        r.row = (int16_t)best.row;
        r.col = (int16_t)best.col;
        r.sad = best.sad;
        r.cost = best.cost;
        out[bi] = r;
      }
      return;
    }
  } else {
    const R1MeResult r = out[bi];      // what PHASE 1 left (the launch before this one on the stream)
    best = Msr{(int)r.row, (int)r.col, r.cost, r.sad};
  }

  auto in_range = [&](int row, int col) {
    return col >= b.mvx_min && col <= b.mvx_max && row >= b.mvy_min && row <= b.mvy_max;
  };
  const bool small = (w < h ? w : h) == 4;
  const int ts = small ? 4 : 8, ntx = w / ts, nt = ntx * (h / ts), ln = small ? 2 : 3;
  const uint8_t *o0 = px_addr<BPP>(org, b.po_x, b.po_y);
  const size_t so = (size_t)org.stride * BPP;
  // distortion of the source block against `pp` (row stride sp), one lane per tile of lanes [0, nt)
  auto block_dist = [&](const uint8_t *pp, size_t sp, int gl, bool satd) -> uint32_t {
    uint32_t s = 0;
    if (gl < nt) {
      const int tx = gl % ntx, ty = gl / ntx;
      const uint8_t *a = o0 + (size_t)ty * ts * so + (size_t)tx * ts * BPP;
      const uint8_t *c = pp + (size_t)ty * ts * sp + (size_t)tx * ts * BPP;
      if (satd) s = small ? r1dist::tile_dist<BPP, 4, true>(a, so, c, sp) : r1dist::tile_dist<BPP, 8, true>(a, so, c, sp);
      else s = small ? r1dist::tile_dist<BPP, 4, false>(a, so, c, sp) : r1dist::tile_dist<BPP, 8, false>(a, so, c, sp);
    }
    return s;
  };
  if (use_satd) {
    // get_fullpel_mv_rd(best.mv, use_satd) (me.rs:596-613): the block at the integer position
    if (!in_range(best.row, best.col)) {
      best.cost = COST_MAX;
      best.sad = 0xFFFFFFFFu;
    } else {
      const uint8_t *rp = px_addr<BPP>(ref, b.po_x + div8(best.col), b.po_y + div8(best.row));
      uint32_t s = block_dist(rp, (size_t)ref.stride * BPP, lane, true);
#pragma unroll
      for (int m = 1; m < 64; m <<= 1) s += __shfl_xor(s, m, 64);
      best.sad = (s + ((1u << ln) >> 1)) >> ln;
      best.cost = b.mc.cost(best.row, best.col, best.sad);
    }
  }
  // subpel_diamond_search: 16-lane group g <-> DIAMOND_R1_PATTERN_SUBPEL[g]
  const int g = lane >> 4, gl = lane & 15;
  uint8_t *win = sh_grp[wave][g];
  uint8_t *pred = win + ((23 * WS_MAX + 15) & ~15);
  const int ws = (((w + 7) * BPP + 3) >> 2) << 2;
  int radius_log2 = 2;
  const int end_log2 = p.allow_hp ? 0 : 1;
  // blocks whose sides are 8 or 16: the source block waits in LDS for the whole search
  const bool fast = (w == 8 || w == 16) && (h == 8 || h == 16);
  uint8_t *srcc = sh_src[wave];
  if (fast) {
    const int wl = w == 16 ? 4 : 3;
    for (int i = lane; i < w * h; i += 64) {
      const int r = i >> wl, c = i & (w - 1);
      if constexpr (BPP == 1) srcc[i] = (uint8_t)ld_px<1>(o0 + (size_t)r * so + c);
      else ((uint16_t *)srcc)[i] = (uint16_t)ld_px<2>(o0 + (size_t)r * so + c * 2);
    }
  }
  __builtin_amdgcn_wave_barrier();
  for (;;) {
    int row = (int16_t)(best.row + (kDiamond[g][0] << radius_log2));
    int col = (int16_t)(best.col + (kDiamond[g][1] << radius_log2));
    const bool ok = in_range(row, col);
    uint32_t s = 0;
    if (fast) {   // wave-uniform: 8 / 16 sizes on the fused-candidate machinery
      if (ok) {
        const int x = b.po_x + (col >> 3), y = b.po_y + (row >> 3), cf = (col << 1) & 15, rf = (row << 1) & 15;
        const bool sd = use_satd != 0;
        const int bd = ref.bit_depth;
#define R1_SP(W_, H_, BD_) s = subpel_group_dist<BPP, W_, H_, BD_>(win, ref, x, y, cf, rf, filter_mode, gl, lane, srcc, sd, bd)
#define R1_SP_BD(W_, H_)                                                  \
  do {                                                                    \
    if constexpr (BPP == 1) R1_SP(W_, H_, 8);                             \
    else if (bd <= 10) R1_SP(W_, H_, 10);                                 \
    else R1_SP(W_, H_, 12);                                               \
  } while (0)
        if (w == 16 && h == 16) R1_SP_BD(16, 16);
        else if (w == 8 && h == 8) R1_SP_BD(8, 8);
        else if (w == 16) R1_SP_BD(16, 8);
        else R1_SP_BD(8, 16);
#undef R1_SP_BD
#undef R1_SP
      }
    } else {
    if (ok) {
      // get_mv_params (src/predict.rs:284-297): floor offset, 1/16 fraction
      r1mc::stage_window<BPP>(win, ws, ref, b.po_x + (col >> 3), b.po_y + (row >> 3), w, h, gl, 16);
    }
    __builtin_amdgcn_wave_barrier();
    if (ok && gl < w) {
      if constexpr (BPP == 1)
        r1mc::mc_column<BPP, false, 0>(win, ws, gl, w, h, (col << 1) & 15, (row << 1) & 15, filter_mode,
                                       filter_mode, ref.bit_depth,
                                       [&](int rr, int32_t v) { pred[rr * w + gl] = (uint8_t)v; });
      else
        r1mc::mc_column<BPP, false, 0>(win, ws, gl, w, h, (col << 1) & 15, (row << 1) & 15, filter_mode,
                                       filter_mode, ref.bit_depth, [&](int rr, int32_t v) {
                                         ((uint16_t *)pred)[rr * w + gl] = (uint16_t)v;
                                       });
    }
    __builtin_amdgcn_wave_barrier();
    if (ok) s = block_dist(pred, (size_t)w * BPP, gl, use_satd != 0);
    }
    s = group_sum<16>(s);
    uint32_t sad = use_satd ? (s + ((1u << ln) >> 1)) >> ln : s;
    unsigned long long cost = ok ? b.mc.cost(row, col, sad) : COST_MAX;
    if (!ok) sad = 0xFFFFFFFFu;
    int idx = g;
#pragma unroll
    for (int m = 16; m < 64; m <<= 1) {
      const unsigned long long oc = ((unsigned long long)(uint32_t)__shfl_xor((int)(cost >> 32), m, 64) << 32) |
                                    (uint32_t)__shfl_xor((int)(uint32_t)cost, m, 64);
      const int oi = __shfl_xor(idx, m, 64), orow = __shfl_xor(row, m, 64), ocol = __shfl_xor(col, m, 64);
      const uint32_t os = (uint32_t)__shfl_xor((int)sad, m, 64);
      if (oc < cost || (oc == cost && oi < idx)) { cost = oc; idx = oi; row = orow; col = ocol; sad = os; }
    }
    if (best.cost <= cost) {
      if (radius_log2 == end_log2) break;
      radius_log2--;
    } else {
      best = Msr{row, col, cost, sad};
    }
  }
  if (lane == 0) {
    R1MeResult r;
    r.row = (int16_t)best.row;
    r.col = (int16_t)best.col;
    r.sad = best.sad;
    r.cost = best.cost;
    out[bi] = r;
  }
}

}  // namespace

#ifdef R1_ME_PROF
extern "C" int r1_debug_me_fine(unsigned long long *out, int reset) {   /* out[8] */
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_me_fine), sizeof(g_me_fine)) != hipSuccess) return -1;
  if (reset) {
    void *p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_me_fine)) != hipSuccess) return -1;
    if (hipMemset(p, 0, sizeof(g_me_fine)) != hipSuccess) return -1;
  }
  return 0;
}
extern "C" int r1_debug_me_step(unsigned long long *out, int reset) {   /* out[3][4] */
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_me_step), sizeof(g_me_step)) != hipSuccess) return -1;
  if (reset) {
    void *p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_me_step)) != hipSuccess) return -1;
    if (hipMemset(p, 0, sizeof(g_me_step)) != hipSuccess) return -1;
  }
  return 0;
}
extern "C" int r1_debug_me_prof(unsigned long long *out, int reset) {   /* out[3][4] */
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_me_prof), sizeof(g_me_prof)) != hipSuccess) return -1;
  if (reset) {
    void *p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_me_prof)) != hipSuccess) return -1;
    if (hipMemset(p, 0, sizeof(g_me_prof)) != hipSuccess) return -1;
  }
  return 0;
}
#endif

// ---- k_me_persist, host side: the item list of a call's geometry (cached per ring slot), the
// flag arrays, one launch ----
namespace {
struct MePersistCache {
  std::vector<int> geo;            // signature: per job tile_w, tile_h
  void *rows = nullptr;            // device: MeRow[n_rows]
  void *foff = nullptr;            // device: uint32[n_jobs][5]
  void *prog = nullptr;            // device: uint32 per row (epoch << 16 | blocks done)
  void *ctl = nullptr;             // device: counter
  unsigned int *err_host = nullptr;   // pinned, host-mapped: set by a wave whose dependency wait ran out
  unsigned int *err_dev = nullptr;    // the same word as the device sees it
  unsigned long long call_id = 0;     // r1_estimate_tile_motion_batch call this slot last served
  int n_rows = 0;
  int xoff[9] = {0};
  unsigned int epoch = 0;
  bool launched = false;
};

// ordering key of a row, in half units of 16-pixel cells: its bottom edge plus a per-pass offset
// chosen so that every row a row waits for has a smaller key
inline int me_row_key(int kind, int gy) {
  static const int c2[3] = {0, 18, 32};
  const int q = kind >= 3 ? kind - 3 : kind, s = 4 >> q;
  return 2 * (gy * s + s - 1) + c2[q] + (kind >= 3 ? 1 : 0);
}

// k_me_persist hands a job's rows out to the waves of ONE XCD (job % 8), which is only right on
// a device whose launches spread over all eight: probed once per context.
__global__ void k_me_xcd_probe(unsigned int *mask) {
  if (threadIdx.x == 0) atomicOr(mask, 1u << (__builtin_amdgcn_s_getreg(6164) & 15));
}

int me_probe_xcds(r1_ctx *ctx, hipStream_t st) {
  unsigned int *d = nullptr, h = 0;
  R1_HIP_CHECK(hipMalloc(&d, sizeof(h)));
  hipError_t e = hipMemsetAsync(d, 0, sizeof(h), st);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_me_xcd_probe, dim3(1024), dim3(64), 0, st, d);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(&h, d, sizeof(h), hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  (void)hipFree(d);
  R1_HIP_CHECK(e);
  // The pinned hand-over (plain result stores that stay in the job's XCD L2, readers bypassing L1)
  // leans on gfx942 / gfx950 cache behaviour, not on the HIP memory model: explicit allow-list on
  // top of the probe.  Anything else takes the unpinned launch (agent-scope write-through).
  hipDeviceProp_t prop;
  R1_HIP_CHECK(hipGetDeviceProperties(&prop, ctx->device));
  const bool arch_ok = !strncmp(prop.gcnArchName, "gfx950", 6) || !strncmp(prop.gcnArchName, "gfx942", 6);
  ctx->me_xcds = (h == 0xFFu && arch_ok) ? 8 : 0;   // anything but exactly XCD 0..7: no pinned launches
  return R1_OK;
}

void me_collect_slot(r1_ctx *ctx, MePersistCache &c);

int me_launch_persistent(r1_ctx *ctx, int slot, const R1MeJob *jobs, int n_jobs, int bpp, const R1MeJob *djobs,
                         const R1MeParams *dparams, R1MeStats *const *drbufs, size_t upload_bytes,
                         hipStream_t st, bool pin) {
  if (!ctx->me_persist[slot]) ctx->me_persist[slot] = new MePersistCache();
  MePersistCache &c = *(MePersistCache *)ctx->me_persist[slot];
  // A dependency wait of the previous call on this slot that ran out of patience (the slot's event
  // has been waited for by the caller of this function): THAT call's statistics are not to be
  // trusted.  It is recorded against that call (r1_me_status reports its id); this call goes ahead.
  me_collect_slot(ctx, c);
  if (!c.err_host) {
    R1_HIP_CHECK(hipHostMalloc((void **)&c.err_host, 64, hipHostMallocMapped));
    *c.err_host = 0;
    hipError_t e = hipHostGetDevicePointer((void **)&c.err_dev, c.err_host, 0);
    if (e != hipSuccess) { (void)hipHostFree(c.err_host); c.err_host = nullptr; R1_HIP_CHECK(e); }
  }
  std::vector<int> geo;
  geo.push_back(pin ? 1 : 0);
  for (int j = 0; j < n_jobs; j++) { geo.push_back(jobs[j].tile_w); geo.push_back(jobs[j].tile_h); }
  if (geo != c.geo || c.epoch >= 65535) {
    // nothing of the old geometry survives a failed rebuild: forget it before freeing
    c.geo.clear();
    c.n_rows = 0;
    for (void **pp : {&c.rows, &c.foff, &c.prog, &c.ctl})
      if (*pp) { (void)hipFree(*pp); *pp = nullptr; }
    std::vector<MeRow> rows;
    std::vector<unsigned int> foff((size_t)n_jobs * 5);
    unsigned int nprog = 0;
    for (int j = 0; j < n_jobs; j++) {
      for (int kind = 0; kind < 5; kind++) {
        const int q = kind >= 3 ? kind - 3 : kind;
        const int nbx = (jobs[j].tile_w + (SB >> q) - 1) / (SB >> q), nby = (jobs[j].tile_h + (SB >> q) - 1) / (SB >> q);
        foff[(size_t)j * 5 + kind] = nprog;
        nprog += (unsigned int)nby;
        for (int gy = 0; gy < nby; gy++)
          rows.push_back(MeRow{(uint16_t)j, (uint8_t)kind, 0, (uint16_t)gy, (uint16_t)nbx});
      }
    }
    const int xmask = pin ? 7 : 0;
    std::stable_sort(rows.begin(), rows.end(), [xmask](const MeRow &a, const MeRow &b) {
      const int xa = a.job & xmask, xb = b.job & xmask;
      if (xa != xb) return xa < xb;
      return me_row_key(a.kind, a.gy) < me_row_key(b.kind, b.gy);
    });
    for (int x = 0; x <= 8; x++) c.xoff[x] = 0;
    for (const MeRow &r : rows) c.xoff[(r.job & xmask) + 1]++;
    for (int x = 0; x < 8; x++) c.xoff[x + 1] += c.xoff[x];
    R1_HIP_CHECK(hipMalloc(&c.rows, rows.size() * sizeof(MeRow)));
    R1_HIP_CHECK(hipMalloc(&c.foff, foff.size() * sizeof(unsigned int)));
    R1_HIP_CHECK(hipMalloc(&c.prog, (size_t)nprog * sizeof(unsigned int)));
    R1_HIP_CHECK(hipMalloc(&c.ctl, 8 * sizeof(unsigned int)));
    R1_HIP_CHECK(hipMemcpy(c.rows, rows.data(), rows.size() * sizeof(MeRow), hipMemcpyHostToDevice));
    R1_HIP_CHECK(hipMemcpy(c.foff, foff.data(), foff.size() * sizeof(unsigned int), hipMemcpyHostToDevice));
    R1_HIP_CHECK(hipMemset(c.prog, 0, (size_t)nprog * sizeof(unsigned int)));
    c.n_rows = (int)rows.size();
    c.epoch = 0;
    c.geo = geo;
  }
  c.epoch++;
  R1_HIP_CHECK(hipMemcpyAsync(ctx->me_jobs[slot], ctx->me_jobs_host[slot], upload_bytes, hipMemcpyHostToDevice, st));
  R1_HIP_CHECK(hipMemsetAsync(c.ctl, 0, 8 * sizeof(unsigned int), st));
  *c.err_host = 0;   // the slot's previous launch has completed (event) and been collected
  MePersistArgs a;
  a.jobs = djobs; a.params = dparams; a.rbufs = drbufs;
  a.rows = (const MeRow *)c.rows; a.n_rows = c.n_rows;
  a.counter = (unsigned int *)c.ctl; a.err = c.err_dev;
  for (int x = 0; x <= 8; x++) a.xoff[x] = c.xoff[x];
  a.prog = (unsigned int *)c.prog; a.foff = (const unsigned int *)c.foff;
  a.epoch = c.epoch;
  a.spin = getenv("R1_ME_PERSISTENT_SPIN") ? atoi(getenv("R1_ME_PERSISTENT_SPIN")) : (1 << 18);
  // TWO waves per SIMD (256 CUs x 4 SIMDs x 2), not as many as fit: a searching wave wants a VALU
  // instruction every ~8 cycles at 2.6-4.4 issue cycles each, so a third and fourth wave on a SIMD
  // stretch every block step of a chain that has no slack (measured, 24 jobs: grid 4096 1.73 ms,
  // 3072 1.60, 2048 1.50, 1536 1.54; 10-bit 2.08 / 1.96 / 1.89 / 2.03, 1024: 2.11).  Rows beyond the grid are taken by
  // the waves that finish theirs, in key order.
  const int gmax = getenv("R1_ME_PERSISTENT_GRID") ? atoi(getenv("R1_ME_PERSISTENT_GRID")) : 2048;
  const int grid = c.n_rows < gmax ? c.n_rows : gmax;
  if (getenv("R1_ME_PERSISTENT_DEBUG")) fprintf(stderr, "k_me_persist: %d rows, grid %d, epoch %u\n", c.n_rows, grid, a.epoch);
  if (bpp == 1 && pin) hipLaunchKernelGGL((k_me_persist<1, true>), dim3(grid), dim3(64), 0, st, a);
  else if (bpp == 1) hipLaunchKernelGGL((k_me_persist<1, false>), dim3(grid), dim3(64), 0, st, a);
  else if (pin) hipLaunchKernelGGL((k_me_persist<2, true>), dim3(grid), dim3(64), 0, st, a);
  else hipLaunchKernelGGL((k_me_persist<2, false>), dim3(grid), dim3(64), 0, st, a);
  R1_HIP_CHECK(hipGetLastError());
  R1_HIP_CHECK(hipEventRecord(ctx->me_done[slot], st));
  c.launched = true;
  c.call_id = ctx->me_calls;
  if (getenv("R1_ME_PERSISTENT_CHECK")) {   // debugging aid: synchronous check of this very call
    R1_HIP_CHECK(hipStreamSynchronize(st));
    const unsigned int e = *(volatile unsigned int *)c.err_host;
    if (getenv("R1_ME_PERSISTENT_DEBUG")) fprintf(stderr, "k_me_persist: rows %d, err %u\n", c.n_rows, e);
    c.launched = false;
    if (e) { r1_set_error("k_me_persist: a dependency wait timed out"); return R1_EHIP; }
  }
  return R1_OK;
}
}  // namespace

// the error word of a slot whose launch has completed: recorded against the call it served
namespace {
void me_collect_slot(r1_ctx *ctx, MePersistCache &c) {
  if (!c.launched || !c.err_host) return;
  c.launched = false;
  if (*(volatile unsigned int *)c.err_host) {
    ctx->me_failed++;
    if (!ctx->me_first_failed) ctx->me_first_failed = c.call_id;
  }
}
}  // namespace

// Results of the persistent tile-ME launches (launch_mode 2 / 3) are valid once this has said so:
// a wave whose dependency wait runs out of patience (a bounded spin, so that a placement the
// protocol did not foresee cannot hang the GPU) goes on with stale predictors and flags the call.
// wait != 0: first waits for every launch enqueued so far.  Returns R1_OK when no call since the
// last r1_me_status has been flagged; R1_ETIMEDOUT otherwise, with *first_failed_call = the 1-based
// index (per context) of the first flagged r1_estimate_tile_motion_batch call -- the caller
// re-issues that call with launch_mode = 1 (the launch-boundary version has no waits).  Flags are
// consumed by the report.  *calls (optional) = calls made on this context so far.
extern "C" int r1_me_status(r1_ctx *ctx, int wait, unsigned long long *first_failed_call,
                            unsigned long long *calls) {
  R1_REQUIRE(ctx);
  std::lock_guard<std::mutex> ring_lock(ctx->me_mu);
  R1DeviceGuard dev_guard(ctx);
  for (int s = 0; s < r1_ctx::kMeSlots; s++) {
    MePersistCache *c = (MePersistCache *)ctx->me_persist[s];
    if (!c || !c->launched || !ctx->me_done[s]) continue;
    if (wait) R1_HIP_CHECK(hipEventSynchronize(ctx->me_done[s]));
    else if (hipEventQuery(ctx->me_done[s]) != hipSuccess) continue;
    me_collect_slot(ctx, *c);
  }
  if (calls) *calls = ctx->me_calls;
  if (first_failed_call) *first_failed_call = ctx->me_first_failed;
  const bool bad = ctx->me_failed != 0;
  if (bad) r1_set_error("k_me_persist: %d call(s) flagged a timed-out dependency wait, first: call %llu; "
                        "re-issue with launch_mode 1", ctx->me_failed, ctx->me_first_failed);
  ctx->me_failed = 0;
  ctx->me_first_failed = 0;
  return bad ? R1_ETIMEDOUT : R1_OK;
}

void r1_me_persist_free(void *cache) {
  MePersistCache *c = (MePersistCache *)cache;
  if (!c) return;
  for (void *p : {c->rows, c->foff, c->prog, c->ctl})
    if (p) (void)hipFree(p);
  if (c->err_host) (void)hipHostFree(c->err_host);
  delete c;
}

extern "C" int r1_estimate_tile_motion_batch(r1_ctx *ctx, const R1MeJob *jobs, int n_jobs,
                                             const R1MeParams *params, void *stream) {
  R1_REQUIRE(ctx && params);
  if (n_jobs <= 0) return R1_OK;
  R1_REQUIRE(jobs);
  R1_REQUIRE(n_jobs <= 256);   // tiles x reference frames of one frame
  R1_REQUIRE(params->bit_depth == 8 || params->bit_depth == 10 || params->bit_depth == 12);
  R1_REQUIRE(params->stats_cols > 0 && params->stats_rows > 0);
  const int bpp = jobs[0].org[0].bytes_per_px;
  R1_REQUIRE(bpp == 1 || bpp == 2);
  int max_sbw = 0, max_sbh = 0;
  for (int j = 0; j < n_jobs; j++) {
    const R1MeJob &b = jobs[j];
    R1_REQUIRE(b.stats);
    R1_REQUIRE(b.tile_x >= 0 && b.tile_y >= 0 && b.tile_w > 0 && b.tile_h > 0);
    R1_REQUIRE(b.tile_x % SB == 0 && b.tile_y % SB == 0 && b.tile_w % MI == 0 && b.tile_h % MI == 0);
    R1_REQUIRE((b.tile_x + b.tile_w) / MI <= params->stats_cols &&
               (b.tile_y + b.tile_h) / MI <= params->stats_rows);
    for (int l = 0; l < 3; l++)
      R1_REQUIRE(b.org[l].data && b.ref[l].data && b.org[l].bytes_per_px == bpp &&
                 b.ref[l].bytes_per_px == bpp);
    const int sbw = (b.tile_w + SB - 1) / SB, sbh = (b.tile_h + SB - 1) / SB;
    max_sbw = sbw > max_sbw ? sbw : max_sbw;
    max_sbh = sbh > max_sbh ? sbh : max_sbh;
  }
  hipStream_t st = (hipStream_t)stream;
  // job descriptors + parameters: caller's memory -> pinned staging -> device, one ring slot per
  // call.  The ring is the one piece of mutable state a context has: concurrent callers (rav1e's
  // per-tile rayon workers share a context) take turns for the enqueue.
  std::lock_guard<std::mutex> ring_lock(ctx->me_mu);
  R1DeviceGuard dev_guard(ctx);
  const size_t jobs_bytes = ((size_t)n_jobs * sizeof(R1MeJob) + 15) & ~(size_t)15;
  const size_t params_bytes = (sizeof(R1MeParams) + 15) & ~(size_t)15;
  const size_t bytes = jobs_bytes + params_bytes + (size_t)n_jobs * sizeof(R1MeStats *);
  const int slot = ctx->me_next;
  if (ctx->me_done[slot]) R1_HIP_CHECK(hipEventSynchronize(ctx->me_done[slot]));
  else R1_HIP_CHECK(hipEventCreateWithFlags(&ctx->me_done[slot], hipEventDisableTiming));
  // Fail-safe for callers that never poll r1_me_status: a persistent launch that has FINISHED with a
  // timed-out dependency wait (stale predictors, non-reference statistics) makes every following call
  // refuse with R1_ETIMEDOUT until r1_me_status has reported -- and thereby consumed -- the flag.
  // Nothing is enqueued and the ring does not advance.
  for (int s = 0; s < r1_ctx::kMeSlots; s++) {
    MePersistCache *pc = (MePersistCache *)ctx->me_persist[s];
    if (!pc || !pc->launched || !ctx->me_done[s]) continue;
    if (s != slot && hipEventQuery(ctx->me_done[s]) != hipSuccess) continue;
    me_collect_slot(ctx, *pc);
  }
  if (ctx->me_failed) {
    r1_set_error("r1_estimate_tile_motion_batch: %d earlier call(s) (first: call %llu) ran with a timed-out "
                 "dependency wait and have not been acknowledged; call r1_me_status and re-issue them with "
                 "launch_mode 1", ctx->me_failed, ctx->me_first_failed);
    return R1_ETIMEDOUT;
  }
  ctx->me_calls++;
  ctx->me_next = (slot + 1) % r1_ctx::kMeSlots;
  if (ctx->me_jobs_bytes[slot] < bytes) {
    if (ctx->me_graph[slot]) (void)hipGraphExecDestroy(ctx->me_graph[slot]);   // it holds the old pointers
    ctx->me_graph[slot] = nullptr;
    if (ctx->me_jobs[slot]) (void)hipFree(ctx->me_jobs[slot]);
    if (ctx->me_jobs_host[slot]) (void)hipHostFree(ctx->me_jobs_host[slot]);
    ctx->me_jobs[slot] = ctx->me_jobs_host[slot] = nullptr;
    ctx->me_jobs_bytes[slot] = 0;
    R1_HIP_CHECK(hipMalloc(&ctx->me_jobs[slot], bytes));
    R1_HIP_CHECK(hipHostMalloc(&ctx->me_jobs_host[slot], bytes, hipHostMallocDefault));
    ctx->me_jobs_bytes[slot] = bytes;
  }
  // the refinement buffers: one MEStats frame per DISTINCT statistics array of the call (the tiles
  // of a frame share theirs), same geometry, so a job's entries sit at the same offsets
  const size_t frame_bytes = (size_t)params->stats_cols * params->stats_rows * sizeof(R1MeStats);
  int uniq_of[256], n_uniq = 0;
  for (int j = 0; j < n_jobs; j++) {
    int u = -1;
    for (int k = 0; k < j && u < 0; k++)
      if (jobs[k].stats == jobs[j].stats) u = uniq_of[k];
    uniq_of[j] = u >= 0 ? u : n_uniq++;
  }
  if (ctx->me_refine_bytes[slot] < frame_bytes * n_uniq) {
    if (ctx->me_graph[slot]) (void)hipGraphExecDestroy(ctx->me_graph[slot]);
    ctx->me_graph[slot] = nullptr;
    if (ctx->me_refine[slot]) (void)hipFree(ctx->me_refine[slot]);
    ctx->me_refine[slot] = nullptr;
    ctx->me_refine_bytes[slot] = 0;
    R1_HIP_CHECK(hipMalloc(&ctx->me_refine[slot], frame_bytes * n_uniq));
    ctx->me_refine_bytes[slot] = frame_bytes * n_uniq;
  }
  memcpy(ctx->me_jobs_host[slot], jobs, (size_t)n_jobs * sizeof(R1MeJob));
  memcpy((uint8_t *)ctx->me_jobs_host[slot] + jobs_bytes, params, sizeof(R1MeParams));
  {
    R1MeStats **rp = (R1MeStats **)((uint8_t *)ctx->me_jobs_host[slot] + jobs_bytes + params_bytes);
    for (int j = 0; j < n_jobs; j++)
      rp[j] = (R1MeStats *)((uint8_t *)ctx->me_refine[slot] + frame_bytes * uniq_of[j]);
  }
  const R1MeJob *djobs = (const R1MeJob *)ctx->me_jobs[slot];
  const R1MeParams *dparams = (const R1MeParams *)((const uint8_t *)ctx->me_jobs[slot] + jobs_bytes);
  R1MeStats *const *drbufs = (R1MeStats *const *)((const uint8_t *)ctx->me_jobs[slot] + jobs_bytes + params_bytes);
  // one persistent launch (k_me_persist) or one launch per superblock diagonal (k_me_diag): the
  // pinned persistent path (2) puts every job on one XCD, so it wants a job per XCD; below that the
  // unpinned one (3: any wave takes any row, results written through at agent scope).  Measured,
  // 8-bit 4K, ms, diagonal launches / pinned / unpinned: 1 job 4.12 / 9.9 / 2.91, 4 jobs 4.80 / 4.7 /
  // 3.96, 8 jobs 1.74 / 1.14 / 1.31, 16 jobs 2.93 / 2.08 / 2.29, 64 jobs 2.03 / 1.69 / 1.77
  static const char *force = getenv("R1_ME_PERSISTENT");   // "0" / "1" / "3": A/B switch for tools/bench_me.py (diagonal / pinned / unpinned)
  int mode = params->launch_mode ? params->launch_mode : (force ? (force[0] == '0' ? 1 : force[0] == '3' ? 3 : 2) : (n_jobs >= 8 ? 2 : 3));
  R1_REQUIRE(mode >= 1 && mode <= 3);
  if (mode == 2) {
    if (ctx->me_xcds < 0) { const int rc = me_probe_xcds(ctx, st); if (rc != R1_OK) return rc; }
    if (ctx->me_xcds != 8) {
      if (params->launch_mode == 2) {
        r1_set_error("r1_estimate_tile_motion_batch: launch_mode 2 needs a device whose launches spread over 8 XCDs");
        return R1_EINVAL;
      }
      mode = 3;
    }
  }
  if (mode >= 2) return me_launch_persistent(ctx, slot, jobs, n_jobs, bpp, djobs, dparams, drbufs, bytes, st, mode == 2);
  const int ndiag = max_sbw + max_sbh - 1;
  const int dlen = max_sbw < max_sbh ? max_sbw : max_sbh;
  // software pipeline over the passes: launch `step` runs diagonal step - 2 q of pass q
  // (grid z = pass); ndiag + 4 launches instead of 3 * ndiag
  const int nsteps = ndiag + 2 * kPassSkew;
  const void *fn = bpp == 1 ? (const void *)k_me_diag<1> : (const void *)k_me_diag<2>;
  static const bool use_graph = !getenv("R1_ME_NO_GRAPH");   // A/B switch for tools/bench_me.py
  if (!use_graph) {
    R1_HIP_CHECK(hipMemcpyAsync(ctx->me_jobs[slot], ctx->me_jobs_host[slot], bytes, hipMemcpyHostToDevice, st));
    for (int step = 0; step < nsteps; step++) {
      if (bpp == 1) hipLaunchKernelGGL(k_me_diag<1>, dim3(dlen, n_jobs, 5), dim3(256), 0, st, djobs, dparams, drbufs, step);
      else hipLaunchKernelGGL(k_me_diag<2>, dim3(dlen, n_jobs, 5), dim3(256), 0, st, djobs, dparams, drbufs, step);
    }
    R1_HIP_CHECK(hipGetLastError());
    R1_HIP_CHECK(hipEventRecord(ctx->me_done[slot], st));
    return R1_OK;
  }
  // The sequence (upload, then one launch per diagonal step, each depending on the one before)
  // is a function of (pixel size, jobs, diagonal length, steps) and of the slot's buffers only --
  // the job contents and the parameters travel through the upload.  It is built once as an
  // explicit hipGraph and replayed with a single hipGraphLaunch per call.
  const long long sig[5] = {bpp, n_jobs, dlen, nsteps, (long long)(size_t)ctx->me_refine[slot]};
  if (!ctx->me_graph[slot] || memcmp(sig, ctx->me_graph_sig[slot], sizeof(sig)) != 0) {
    if (ctx->me_graph[slot]) (void)hipGraphExecDestroy(ctx->me_graph[slot]);
    ctx->me_graph[slot] = nullptr;
    hipGraph_t g;
    R1_HIP_CHECK(hipGraphCreate(&g, 0));
    hipGraphNode_t prev;
    hipError_t e = hipGraphAddMemcpyNode1D(&prev, g, nullptr, 0, ctx->me_jobs[slot], ctx->me_jobs_host[slot],
                                           bytes, hipMemcpyHostToDevice);
    for (int step = 0; step < nsteps && e == hipSuccess; step++) {
      int step_arg = step;
      void *args[4] = {(void *)&djobs, (void *)&dparams, (void *)&drbufs, (void *)&step_arg};
      hipKernelNodeParams kp;
      memset(&kp, 0, sizeof(kp));
      kp.func = (void *)fn;
      kp.gridDim = dim3(dlen, n_jobs, 5);   // roles: three searches, two refinements (k_me_diag)
      kp.blockDim = dim3(256);
      kp.sharedMemBytes = 0;
      kp.kernelParams = args;   // copied at node creation
      hipGraphNode_t node;
      e = hipGraphAddKernelNode(&node, g, &prev, 1, &kp);
      prev = node;
    }
    if (e == hipSuccess) e = hipGraphInstantiate(&ctx->me_graph[slot], g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (e != hipSuccess) {
      ctx->me_graph[slot] = nullptr;
      R1_HIP_CHECK(e);
    }
    memcpy(ctx->me_graph_sig[slot], sig, sizeof(sig));
  }
  R1_HIP_CHECK(hipGraphLaunch(ctx->me_graph[slot], st));
  R1_HIP_CHECK(hipEventRecord(ctx->me_done[slot], st));
  return R1_OK;
}

// k_me_blocks' dynamic LDS at 64x64, 16-bit: source + 4 x (window + prediction) -- above the
// 64 KB default, so the limit is raised ONCE per context (ctx.hip) to this worst case; a
// per-launch setting would race between threads sharing a context.
static constexpr size_t kMeBlocksMaxLds = 8192 + 4 * ((((size_t)(64 + 7) * 144 + 15) & ~(size_t)15) + 8192);

int r1_me_kernel_attrs() {
  R1_HIP_CHECK(hipFuncSetAttribute((const void *)k_me_blocks<1>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMeBlocksMaxLds));
  R1_HIP_CHECK(hipFuncSetAttribute((const void *)k_me_blocks<2>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMeBlocksMaxLds));
  return R1_OK;
}

extern "C" int r1_estimate_motion_batch(r1_ctx *ctx, const R1MeJob *tile, const R1MeParams *params,
                                        const R1MeBlockCand *cands, int n, int max_w, int max_h,
                                        int use_satd, int filter_mode, R1MeResult *out,
                                        void *stream) {
  R1_REQUIRE(ctx && tile && params);
  R1_REQUIRE(params->bit_depth == 8 || params->bit_depth == 10 || params->bit_depth == 12);
  R1_REQUIRE(filter_mode >= 0 && filter_mode <= 3);
  R1_REQUIRE(r1_is_pow2(max_w) && r1_is_pow2(max_h) && max_w >= 4 && max_h >= 4 && max_w <= 64 &&
             max_h <= 64);
  R1_REQUIRE(tile->stats && tile->org[0].data && tile->ref[0].data);
  R1_REQUIRE(tile->tile_x % SB == 0 && tile->tile_y % SB == 0 && tile->tile_w % MI == 0 &&
             tile->tile_h % MI == 0 && tile->tile_w > 0 && tile->tile_h > 0);
  const int bpp = tile->org[0].bytes_per_px;
  R1_REQUIRE((bpp == 1 || bpp == 2) && tile->ref[0].bytes_per_px == bpp);
  if (n <= 0) return R1_OK;
  R1_REQUIRE(cands && out);
  hipStream_t st = (hipStream_t)stream;
  if (max_w <= 16 && max_h <= 16) {   // one wave per block
    const unsigned grid = (unsigned)((n + 3) / 4);
#ifndef R1_ME_SMALL_SPLIT
#define R1_ME_SMALL_SPLIT 0   // 1 = two launches (round 6 A/B, profiles/r06_ab_notes.md ab3: 3-7 % SLOWER -- the scratch
                              // traffic of the one-launch form is not what its waves wait for); env R1_ME_SMALL_SPLIT overrides
#endif
#define R1_SMALL(B, PH) hipLaunchKernelGGL((k_me_blocks_small<B, PH>), dim3(grid), dim3(256), 0, st, *tile, *params, cands, n, \
                                           max_w, max_h, use_satd, filter_mode, out)
    static const int split = [] {
      const char *e = getenv("R1_ME_SMALL_SPLIT");
      return e ? atoi(e) : R1_ME_SMALL_SPLIT;
    }();
    if (split) {
      if (bpp == 1) { R1_SMALL(1, 1); R1_SMALL(1, 2); } else { R1_SMALL(2, 1); R1_SMALL(2, 2); }
    } else {
      if (bpp == 1) R1_SMALL(1, 0); else R1_SMALL(2, 0);
    }
#undef R1_SMALL
    R1_HIP_CHECK(hipGetLastError());
    return R1_OK;
  }
  // LDS for the largest block of the batch: source + 4 x (window + prediction)
  const int ws = (((max_w + 7) * bpp + 3) >> 2) << 2;
  const size_t blk = ((size_t)max_w * max_h * bpp + 15) & ~(size_t)15;
  const size_t lds = blk + 4 * ((((size_t)(max_h + 7) * ws + 15) & ~(size_t)15) + blk);
  R1_REQUIRE(lds <= kMeBlocksMaxLds);
  if (bpp == 1) {
    hipLaunchKernelGGL(k_me_blocks<1>, dim3(n), dim3(256), lds, st, *tile, *params, cands, max_w,
                       max_h, use_satd, filter_mode, out);
  } else {
    hipLaunchKernelGGL(k_me_blocks<2>, dim3(n), dim3(256), lds, st, *tile, *params, cands, max_w,
                       max_h, use_satd, filter_mode, out);
  }
  R1_HIP_CHECK(hipGetLastError());
  return R1_OK;
}
