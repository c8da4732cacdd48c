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
//   * one LAUNCH per anti-diagonal of superblocks per pass (three passes:
//     quarter, half, full resolution) -- the same-diagonal SBs of ALL jobs
//     (tiles x reference frames) run concurrently, grid = (diag length, jobs);
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
#include "common.hpp"

namespace {

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

// One block of one wave.  RH = rows per candidate slot (16 or 32): the block
// is at most RH x RH; 64 / RH candidates are evaluated per step.
template <int BPP, int RH>
struct Block {
  static constexpr int NCS = 64 / RH, GR = RH / 4, WPG = BPP;   // dwords per 4-px granule
  const uint8_t *ref0;   // (po.x, po.y) of the reference plane
  long sr;               // reference stride, bytes
  int w, h, po_x, po_y;
  int mvx_min, mvx_max, mvy_min, mvy_max;
  uint32_t lambda;
  int allow_hp;
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

  // compute_mv_rd of this lane's slot candidate (me.rs:1386-1462); every lane
  // of the slot returns the same (cost, sad).  check: the MV range test of
  // get_fullpel_mv_rd (full_search calls compute_mv_rd without it).
  __device__ __forceinline__ void eval(int row, int col, bool valid, bool check,
                                       unsigned long long &cost, uint32_t &sad) const {
    bool in = valid;
    if (check) in = in && col >= mvx_min && col <= mvx_max && row >= mvy_min && row <= mvy_max;
    uint32_t part = 0;
    if (in) {
      const uint8_t *p = ref0 + (long)(div8(row) + r) * sr + (long)div8(col) * BPP;
#pragma unroll
      for (int g = 0; g < GR; g++) {
        if constexpr (BPP == 1) {
          if (m[g]) part = __builtin_amdgcn_sad_u8(o[g], ld_u32(p + 4 * g) & m[g], part);
        } else {
          if (m[2 * g]) {
            const U32x2 v = ld_u32x2(p + 8 * g);
            part = __builtin_amdgcn_sad_u16(o[2 * g], v.a & m[2 * g], part);
            part = __builtin_amdgcn_sad_u16(o[2 * g + 1], v.b & m[2 * g + 1], part);
          }
        }
      }
    }
#pragma unroll
    for (int s = 1; s < RH; s <<= 1) part += __shfl_xor(part, s, 64);
    // pmv = [0, 0] in every caller of this path (estimate_motion with pmv = None,
    // refine_subsampled_motion_estimate): rate1 == rate2, rate = min(r, r + 1) = r
    const int dr = allow_hp ? row : row >> 1, dc = allow_hp ? col : col >> 1;
    const uint32_t rate = 2u * (uint32_t)(ilog_abs((int16_t)dr) + ilog_abs((int16_t)dc));
    cost = in ? 256ull * part + (unsigned long long)rate * lambda : COST_MAX;
    sad = in ? part : 0xFFFFFFFFu;
  }

  // `for cand in cands { if rd.cost < best.rd.cost { best = cand } }` over
  // n candidates produced by gen(idx, row, col); best_idx: index of the taken one.
  template <class Gen>
  __device__ __forceinline__ void scan(int n, Gen gen, bool check, Msr &best, int *best_idx) const {
    for (int base = 0; base < n; base += NCS) {
      int idx = base + slot;
      const bool valid = idx < n;
      int row = 0, col = 0;
      if (valid) gen(idx, row, col);
      unsigned long long cost;
      uint32_t sad;
      eval(row, col, valid, check, cost, sad);
#pragma unroll
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
      if (cost < best.cost) {
        best = Msr{row, col, cost, sad};
        if (best_idx) *best_idx = idx;
      }
    }
  }
};

__constant__ int8_t kDiamond[4][2] = {{1, 0}, {0, 1}, {-1, 0}, {0, -1}};   // (row, col)
__constant__ int8_t kHexagon[6][2] = {{-2, 0}, {-1, 2}, {1, 2}, {2, 0}, {1, -2}, {-1, -2}};
__constant__ int8_t kSquare[8][2] = {{1, -1}, {1, 0}, {1, 1}, {0, -1}, {0, 1}, {-1, -1}, {-1, 0}, {-1, 1}};
// UMH_PATTERN as written in the reference (entry 13 repeats entry 7), me.rs:1153-1156
__constant__ int8_t kUmh[16][2] = {{4, -2}, {4, -1}, {4, 0}, {4, 1}, {4, 2}, {2, 3}, {0, 4}, {-2, 3},
                                   {-4, 2}, {-4, 1}, {-4, 0}, {-4, -1}, {-4, -2}, {-2, 3}, {0, -4}, {2, -3}};

template <class B>
__device__ void fullpel_diamond_search(const B &b, Msr &cur) {
  int radius_log2 = 1;
  for (;;) {
    Msr best = msr_empty();
    const int cr = cur.row, cc = cur.col, sh = 3 + radius_log2;
    b.scan(4, [&](int i, int &row, int &col) {
      row = (int16_t)(cr + (kDiamond[i][0] << sh));
      col = (int16_t)(cc + (kDiamond[i][1] << sh));
    }, true, best, nullptr);
    if (cur.cost <= best.cost) {
      if (radius_log2 == 0) break;
      radius_log2--;
    } else {
      cur = best;
    }
  }
}

template <class B>
__device__ void hexagon_search(const B &b, Msr &cur) {
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
__device__ void uneven_multi_hex_search(const B &b, Msr &cur, int me_range) {
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
__device__ Msr full_search(const B &b, int x_lo, int x_hi, int y_lo, int y_hi, int step) {
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

// one MEStats entry through the L2 (written by another wave of this workgroup
// a barrier ago, or by another workgroup in an earlier launch)
__device__ __forceinline__ void load_stats(const R1MeStats *s, int &row, int &col, uint32_t &nsad) {
  const unsigned long long v = __hip_atomic_load((const unsigned long long *)s, __ATOMIC_RELAXED,
                                                 __HIP_MEMORY_SCOPE_AGENT);
  row = (int16_t)(v & 0xFFFF);
  col = (int16_t)((v >> 16) & 0xFFFF);
  nsad = (uint32_t)(v >> 32);
}

// process_cand (me.rs:407-414)
__device__ __forceinline__ void process_cand(const R1MeStats *s, const int *rng, uint32_t &min_sad,
                                             int16_t *out) {
  int srow, scol;
  uint32_t ns;
  load_stats(s, srow, scol, ns);
  min_sad = ns < min_sad ? ns : min_sad;
  out[0] = (int16_t)iclamp(div8(srow) * 8, rng[2], rng[3]);
  out[1] = (int16_t)iclamp(div8(scol) * 8, rng[0], rng[1]);
}

__device__ void get_subset_predictors(const TileView &t, int bx, int by, int pix_w, int pix_h,
                                      const int *rng, int corner, int ssdec, Subsets &s) {
  uint32_t min_sad = 0xFFFFFFFFu;
  s.nb = s.nc = s.has_median = 0;
  const int w = ((pix_w << ssdec) + MI - 1) >> 2, h = ((pix_h << ssdec) + MI - 1) >> 2;
  const int half_w = imin(w >> 1, t.tcols - 1 - bx), half_h = imin(h >> 1, t.trows - 1 - by);
  if (bx > 0) process_cand(t.at(by + half_h, bx - 1), rng, min_sad, s.b + 2 * s.nb++);
  if (by > 0) process_cand(t.at(by - 1, bx + half_w), rng, min_sad, s.b + 2 * s.nb++);
  if (corner && (corner & 2) && bx + w < t.tcols)
    process_cand(t.at(by + half_h, bx + w), rng, min_sad, s.b + 2 * s.nb++);
  if (corner && (corner & 4) && by + h < t.trows)
    process_cand(t.at(by + h, bx + half_w), rng, min_sad, s.b + 2 * s.nb++);
  if (corner) {
    s.has_median = 1;
    process_cand(t.at(by + half_h, bx + half_w), rng, min_sad, s.median);
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
  if (t.prev) {
    const int fx = t.tx + bx, fy = t.ty + by;
    const int hw = imin(w >> 1, t.cols_f - 1 - fx), hh = imin(h >> 1, t.rows_f - 1 - fy);
#define R1_PREV(y, x) (t.prev + (size_t)(y) * t.cols_f + (x))
    if (fx > 0) process_cand(R1_PREV(fy + hh, fx - 1), rng, min_sad, s.c + 2 * s.nc++);
    if (fy > 0) process_cand(R1_PREV(fy - 1, fx + hw), rng, min_sad, s.c + 2 * s.nc++);
    if (fx + w < t.cols_f) process_cand(R1_PREV(fy + hh, fx + w), rng, min_sad, s.c + 2 * s.nc++);
    if (fy + h < t.rows_f) process_cand(R1_PREV(fy + h, fx + hw), rng, min_sad, s.c + 2 * s.nc++);
    process_cand(R1_PREV(fy + hh, fx + hw), rng, min_sad, s.c + 2 * s.nc++);
#undef R1_PREV
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

template <class B>
__device__ void try_cands(const B &b, const int16_t *list, int n, Msr &best) {
  Msr r = msr_empty();
  b.scan(n, [&](int i, int &row, int &col) { row = list[2 * i]; col = list[2 * i + 1]; }, true, r,
         nullptr);
  fullpel_diamond_search(b, r);
  if (r.cost < best.cost) best = r;
}

template <class B>
__device__ Msr full_pixel_me(const B &b, const TileView &t, const R1MeParams &p, int bx, int by,
                             const int *rng, int corner, bool extensive, int ssdec,
                             int16_t *lds) {
  Subsets s;
  s.median = lds;
  s.b = lds + 2;
  s.c = lds + 12;
  s.all = lds + 22;
  get_subset_predictors(t, bx, by, b.w, b.h, rng, corner, ssdec, s);
  Msr best = msr_empty();
  if (!extensive) {
    try_cands(b, s.all, s.has_median + s.nb + s.nc, best);
    return best;
  }
  // (min_sad as f32 * 1.2) as u32 + ((w * h) << (bit_depth - 8)), me.rs:773-774
  const uint32_t thresh = (uint32_t)__fmul_rn((float)s.min_sad, 1.2f) +
                          ((uint32_t)(b.w * b.h) << (p.bit_depth - 8));
  if (s.has_median) {
    try_cands(b, s.median, 1, best);
    if (best.sad < thresh) return best;
  }
  try_cands(b, s.b, s.nb, best);
  if (best.sad < thresh) return best;
  try_cands(b, s.c, s.nc, best);
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
__device__ __forceinline__ void setup_block(B &b, const R1MeJob &job, const R1MeParams &p,
                                            const TileView &t, int bx, int by, int w, int h,
                                            int ssdec, int lane, int *rng) {
  const int fbx = t.tx + bx, fby = t.ty + by;
  mv_range(p, fbx, fby, w << ssdec, h << ssdec, ssdec, rng);
  b.w = w;
  b.h = h;
  b.po_x = (fbx * MI) >> ssdec;
  b.po_y = (fby * MI) >> ssdec;
  b.mvx_min = rng[0]; b.mvx_max = rng[1]; b.mvy_min = rng[2]; b.mvy_max = rng[3];
  b.lambda = p.lambda[ssdec];
  b.allow_hp = p.allow_hp;
  b.init(job.org[ssdec], job.ref[ssdec], lane);
}

// save_me_stats (me.rs:324-337) with the normalisation of me.rs:268-270
__device__ __forceinline__ void store_result(const TileView &t, int size_in_b, int bx, int by,
                                             const Msr &r, int w, int h, int ssdec, int lane) {
  const uint32_t nsad = (uint32_t)((((unsigned long long)r.sad) << 14) / (unsigned long long)(w * h));
  const int nx = imin(bx + size_in_b, t.tcols) - bx, ny = imin(by + size_in_b, t.trows) - by;
  R1MeStats v;
  v.row = (int16_t)(r.row << ssdec);
  v.col = (int16_t)(r.col << ssdec);
  v.normalized_sad = nsad;
  for (int i = lane; i < nx * ny; i += 64) *t.at(by + i / nx, bx + i % nx) = v;
}

// One pass (log2b = 4, 3, 2 <-> ssdec 2, 1, 0) over the superblocks of one
// anti-diagonal of every job.
template <int BPP>
__global__ __launch_bounds__(256) void k_me_diag(const R1MeJob *__restrict__ jobs, R1MeParams p,
                                                 int log2b, int diag) {
  const R1MeJob &job = jobs[blockIdx.y];
  const int sbw = (job.tile_w + SB - 1) / SB, sbh = (job.tile_h + SB - 1) / SB;
  const int sby = (int)blockIdx.x + imax(0, diag - (sbw - 1)), sbx = diag - sby;
  if (sby >= sbh || sbx < 0 || sbx >= sbw) return;   // workgroup-uniform
  __shared__ int16_t sh_subsets[4][kSubsetWords];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const bool init = log2b == 4;
  const int ssdec = log2b - 2;
  TileView t{job.stats, job.prev, p.stats_cols, p.stats_rows, job.tile_x / MI, job.tile_y / MI,
             job.tile_w / MI, job.tile_h / MI};
  const int sb_w = imin(SB, job.tile_w - sbx * SB), sb_h = imin(SB, job.tile_h - sby * SB);

  if (!init) {
    // refine_subsampled_sb_motion: the previous pass' blocks at this resolution
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
      store_result(t, 1 << (log2b + 1), bx, by, r, w, h, ssdec, lane);
    }
    __threadfence();
    __syncthreads();
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
    __threadfence();
    __syncthreads();
  }
}

}  // namespace

extern "C" int r1_estimate_tile_motion_batch(r1_ctx *ctx, const R1MeJob *jobs, int n_jobs,
                                             const R1MeParams *params, void *stream) {
  R1_REQUIRE(ctx && params);
  if (n_jobs <= 0) return R1_OK;
  R1_REQUIRE(jobs);
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
  // job descriptors: host -> a device copy owned by the context
  const size_t bytes = (size_t)n_jobs * sizeof(R1MeJob);
  if (ctx->me_jobs_bytes < bytes) {
    if (ctx->me_jobs) (void)hipFree(ctx->me_jobs);
    ctx->me_jobs = nullptr;
    ctx->me_jobs_bytes = 0;
    R1_HIP_CHECK(hipMalloc(&ctx->me_jobs, bytes));
    ctx->me_jobs_bytes = bytes;
  }
  R1_HIP_CHECK(hipMemcpyAsync(ctx->me_jobs, jobs, bytes, hipMemcpyHostToDevice, st));
  const int ndiag = max_sbw + max_sbh - 1;
  const int dlen = max_sbw < max_sbh ? max_sbw : max_sbh;
  for (int log2b = 4; log2b >= 2; log2b--)
    for (int d = 0; d < ndiag; d++) {
      if (bpp == 1)
        hipLaunchKernelGGL(k_me_diag<1>, dim3(dlen, n_jobs), dim3(256), 0, st,
                           (const R1MeJob *)ctx->me_jobs, *params, log2b, d);
      else
        hipLaunchKernelGGL(k_me_diag<2>, dim3(dlen, n_jobs), dim3(256), 0, st,
                           (const R1MeJob *)ctx->me_jobs, *params, log2b, d);
    }
  R1_HIP_CHECK(hipGetLastError());
  // the descriptor buffer is reused by the next call on this context
  R1_HIP_CHECK(hipStreamSynchronize(st));
  return R1_OK;
}
