// lrf.hip -- loop restoration: the self-guided (SGRPROJ) stripe filter of a
// whole plane (SURVEY.md 8f "N3", last stage of deblock -> CDEF -> LRF;
// reference src/lrf.rs: sgrproj_sum_finish 345-363, sgrproj_box_ab_* 176-240,
// sgrproj_box_f_r0/_r1/_r2 242-341, VertPaddedIter / HorzPaddedIter 402-524,
// setup_integral_image 530-627, sgrproj_stripe_filter 630-830,
// RestorationState::lrf_filter_frame 1482-1585; the encoder never selects the
// Wiener filter, src/rdo.rs:2508).
//
// The reference walks stripe by stripe and restoration unit by unit, builds an
// integral image of the padded stripe and rolls three / two rows of (a, b)
// intermediates down the stripe.  Every output pixel, however, only depends
// on the padded stripe within 3 pixels of it, so here:
//   * one WORKGROUP per (stripe, 32-column chunk of a restoration unit);
//   * the padded chunk ((32 + 7) x (stripe + 6) pixels: rows outside the
//     stripe from the deblocked plane -- at most two -- then replicated,
//     columns outside the unit real up to 4 / 3 pixels, replicated at the
//     frame edge) is staged into LDS once;
//   * the (a, b) pairs of both passes are computed for the whole chunk by
//     direct 3x3 / 5x5 box sums from LDS (exact: the integral image's wrapping
//     differences are these sums) and parked in LDS packed into one dword
//     (a <= 256: 9 bits, b < 2^21);
//   * every thread then finishes 8 pixels: the weighted (a, b) stencils, the
//     projection with xqd, clamp, store.
// u32 arithmetic wraps where the reference's release build wraps (p * s).
#include <type_traits>

#include "common.hpp"
#include "dist_common.hpp"

namespace {

__constant__ uint16_t kSgrS[16][2] = {{140, 3236}, {112, 2158}, {93, 1618}, {80, 1438}, {70, 1295},
                                      {58, 1177},  {47, 1079},  {37, 996},  {30, 925},  {25, 863},
                                      {0, 2589},   {0, 1618},   {0, 1177},  {0, 925},   {56, 0},
                                      {22, 0}};

constexpr int TW = 32;                 // chunk width
constexpr int SW = TW + 7;             // padded chunk width
constexpr int AW = TW + 2;             // (a, b) columns: centres -1 .. TW

struct LrfGeom {
  int ydec, crop_w, crop_h, stripe_n, unit_size, unit_cols, unit_rows, stripe_height, bd;
  int chunks;   // 32-column chunks across the plane
};

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// a(z) of sgrproj_sum_finish (lrf.rs:352-358): 256 for z >= 255, 1 for z = 0, else ((z << 8) + z / 2) / (z + 1)
// -- 255 quotients, tabulated at compile time instead of an integer division (~40 instructions) per (a, b) pair
struct SgrATable {
  uint16_t v[256];
  constexpr SgrATable() : v() {
    for (int z = 0; z < 256; z++) v[z] = (uint16_t)(z >= 255 ? 256 : (z == 0 ? 1 : ((z << 8) + z / 2) / (z + 1)));
  }
};
__device__ const SgrATable kSgrA = SgrATable();

// Full-rate 24-bit multiplies, spelled out: where an operand is carried around a loop the instruction selector's
// known-bits walk loses the range and __umul24 comes out as the quarter-rate v_mul_lo_u32 (same finding as
// tx_common.hpp's m24).  Callers state the operand ranges.
__device__ __forceinline__ uint32_t mul_u24(uint32_t a, uint32_t b) {
  uint32_t r;
  asm("v_mul_u32_u24_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ uint32_t mad_u24(uint32_t a, uint32_t b, uint32_t c) {
  uint32_t r;
  asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ int32_t mad_i24(int32_t a, int32_t b, int32_t c) {
  int32_t r;
  asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

// Byte offset of pixel (x, y) from the start of a plane's allocation, in 32 bits with the full-rate multiplier: the
// entry points require stride, alloc_height < 2^24 and an allocation below 4 GiB (lrf_plane_ok).  A load at
// `data + offset` then takes the uniform base from SGPRs and needs no 64-bit vector arithmetic (px_addr is a
// quarter-rate 64-bit multiply-add per address: 20 of them per thread and tile here).
template <int BPP>
__device__ __forceinline__ uint32_t px_off(const R1Plane &p, int x, int y) {
  return mad_u24((uint32_t)(p.yorigin + y), (uint32_t)p.stride, (uint32_t)(p.xorigin + x)) * BPP;
}
template <int BPP>
__device__ __forceinline__ uint32_t ld_px_at(const R1Plane &p, uint32_t off) {
  return ld_px<BPP>((const uint8_t *)p.data + off);
}
inline bool lrf_plane_ok(const R1Plane *p) {
  return p->stride > 0 && p->stride < (1 << 24) && p->alloc_height > 0 && p->alloc_height < (1 << 24) &&
         (unsigned long long)p->stride * (unsigned long long)p->alloc_height * (unsigned long long)p->bytes_per_px < (1ull << 32);
}

// sgrproj_sum_finish -> a | b << 9.  Every product but (for bit depth 12) the last has operands below 2^24 whatever
// the bit depth -- the sums are scaled to the 8-bit range first: scaled_ssq <= 25 * 2^16, scaled_sum <= 25 * 2^8,
// p = n * ssq - sum^2 <= n^2 * (255 / 2)^2 + rounding < 2^24 (the min below keeps the full-rate multiplier exact
// even if that bound were wrong: z saturates at 255 from p * s >= 255 << 20 on, and (2^24 - 1) * 22 is past it),
// (256 - a) * sum <= 255 * 25 * 4095 < 2^25 and, for bit depths up to 10 (NARROW), < 2^23
template <bool NARROW>
__device__ __forceinline__ uint32_t sum_finish(uint32_t ssq, uint32_t sum, uint32_t n,
                                               uint32_t one_over_n, uint32_t s, int bd, const uint16_t *atab) {
  const int sh = bd - 8;
  const uint32_t scaled_ssq = (ssq + ((1u << (2 * sh)) >> 1)) >> (2 * sh);
  const uint32_t scaled_sum = (sum + ((1u << sh) >> 1)) >> sh;
  const int32_t d = (int32_t)__umul24(scaled_ssq, n) - (int32_t)mul_u24(scaled_sum, scaled_sum);
  uint32_t p = (uint32_t)(d > 0 ? d : 0);
  p = p < 0xFFFFFFu ? p : 0xFFFFFFu;
  const uint32_t z = (__umul24(p, s) + (1u << 19)) >> 20;
  const uint32_t a = atab[z < 255u ? z : 255u];
  const uint32_t x = mul_u24((1u << 8) - a, sum);
  const uint32_t b = ((NARROW ? __umul24(x, one_over_n) : x * one_over_n) + (1u << 11)) >> 12;
  return a | (b << 9);
}

// One tile of one unit: columns [cx0, cx0 + tw) (absolute), rows [ty0, ty0 + th)
// relative to the unit's top (ty0 even: the radius-2 pass lives on the odd rows).
struct SgrTile {
  int x0, y0, uw, uh;      // the unit (restoration unit x stripe, or an RDO unit)
  int crop_w, crop_h;      // absolute crop of the plane / of the unit
  int cx0, ty0, tw, th;
  // What setup_integral_image sees left of / above the unit (lrf.rs: `cdeffed.x == 0`, `clamp(y, 0, crop - 1)`): it
  // asks where the unit's slice starts IN ITS PLANE.  The frame filter's plane is the frame: lu = 4 unless x0 == 0,
  // top = 2 (rows above exist down to the plane's row 0).  The restoration search's plane is rdo_loop_decision's
  // scratch copy of the AREA it is deciding (rdo.rs:2277-2296: no padding): a unit in the area's first unit column /
  // row sees nothing left of / above itself, wherever the area lies in the frame -- the caller's edge flags.
  int lu, top;             // real columns left of the unit (0 or 4) / real rows above it (0 or 2)
};
// R1SgrSolveUnit::edges -> (lu, top); a flag is void at the plane's own edge
__device__ __forceinline__ void sgr_unit_edges(SgrTile &t, int edges) {
  t.lu = (edges & R1_SGR_EDGE_LEFT) && t.x0 > 0 ? 4 : 0;
  t.top = (edges & R1_SGR_EDGE_ABOVE) && t.y0 > 0 ? 2 : 0;
}

// Stage the padded tile, compute the (a, b) pairs of both passes, then hand every pixel of the tile to
// `emit(x, y, p, f1, f2, extra)` (x, y tile-relative).
//   TROWS   the most rows a tile of this instantiation has (t.th <= TROWS): sizes the three LDS arrays -- the restoration
//           search runs 32-row tiles where that buys a workgroup per CU, the frame filter 64-row stripes
//   NARROW  the caller guarantees bit depth <= 10: sum_finish's last product fits the full-rate 24-bit multiplier
//   extra_p the pixel of this plane under every pixel of the tile (tile pixel (x, y) <-> plane pixel (ex0 + x, ey0 + y))
//           is loaded for the thread's own pixels BEFORE the tile is staged and handed to emit: the load the caller
//           needs per pixel (the source plane of the moments) is in flight behind the whole tile; null: extra = 0
//   flush() called after every four rows of a thread's column and at its end (a caller accumulating products of
//           14-bit differences in 32 bits moves them to its wide sums there)
// Round 5 (ab10): the tile went on an instruction diet -- the kernel sits at ~80 % of the VALU issue rate, so what
// counts is the count: staging walks rows with a fixed column per thread (47 -> ~15 instructions per element, no
// division in the loop), every multiply whose operands are proven below 2^24 is the full-rate v_mul_u32_u24 /
// v_mad_u32_u24 (v_mul_lo_u32 is quarter rate: 5 per (a, b) pair, 2 per pixel), and the stencil loop is unrolled
// over the thread's rows (even / odd rows of the radius-2 pass resolved at compile time, no register rotation).
//   STRICT  only the unit's own pixels come from inside_p: the columns left of it come from outside_p like the rows
//           above it (the CDEF trial of ONE superblock inside an area whose other superblocks keep their current
//           output, rdo.rs:2458-2489)
template <int BPP, int TROWS, bool NARROW, bool STRICT = false, class Emit, class Flush>
__device__ __forceinline__ void sgr_tile(const R1Plane &inside_p, const R1Plane &outside_p,
                                         const SgrTile &t, int set, int bd, const R1Plane *extra_p, int ex0, int ey0,
                                         Emit emit, Flush flush) {
  static_assert(TROWS % 2 == 0 && TROWS <= 64, "row tiles start on even rows");
  __shared__ uint16_t S[TROWS + 6][SW + 1];
  __shared__ uint32_t ab1[TROWS + 2][AW];
  __shared__ uint32_t ab2[TROWS / 2 + 1][AW];
  const int tid = threadIdx.x;
  // a(z): one table lookup per (a, b) pair; a 512-byte copy per workgroup makes it an LDS read
  __shared__ uint16_t atab_s[256];
  if (tid < 128) ((uint32_t *)atab_s)[tid] = ((const uint32_t *)kSgrA.v)[tid];   // visible after the barrier below
  const uint16_t *atab = atab_s;
  const uint32_t s2 = kSgrS[set & 15][0], s1 = kSgrS[set & 15][1];
  // ---- 0: the thread's pixels of the stencil phase (a column x over `per` rows from y0) and the caller's loads ----
  static_assert(TW == 32, "column = tid & 31");
  constexpr int PER_MAX = (((TROWS + 7) >> 3) + 1) & ~1;
  const int x = tid & (TW - 1);
  const int per = (((t.th + 7) >> 3) + 1) & ~1;
  const int y0 = (tid >> 5) * per;                       // 8 segments
  const int ny = x < t.tw ? (t.th - y0 < per ? t.th - y0 : per) : 0;   // <= 0: nothing to do in phase 3
  uint32_t extra[PER_MAX];
  if (extra_p) {
    const uint32_t o0 = px_off<BPP>(*extra_p, ex0 + x, ey0 + y0);
#pragma unroll
    for (int k = 0; k < PER_MAX; k++) extra[k] = k < ny ? ld_px_at<BPP>(*extra_p, o0 + (uint32_t)(k * extra_p->stride * BPP)) : 0u;
  } else {
#pragma unroll
    for (int k = 0; k < PER_MAX; k++) extra[k] = 0u;
  }
  // ---- 1: padded tile -> LDS (VertPaddedIter / HorzPaddedIter, lrf.rs:402-524) ----
  const int h2 = t.uh + (t.uh & 1), th2 = t.th + (t.th & 1);
  {
    constexpr int SROWS = 256 / SW;          // rows per pass: a thread keeps its column
    const int jj = tid / SW, i = tid - jj * SW;   // S[j][i] <-> unit pixel (cx0 - x0 + i - 4, ty0 + j - 4)
    if (jj < SROWS) {
      const int lu = t.lu;
      int ru = (t.crop_w - t.x0) - t.uw;
      ru = ru < 3 ? ru : 3;
      // (never left of the allocation: px_off's unsigned arithmetic would wrap a negative column to +4 GiB)
      const int xa_ = t.x0 + clampi(t.cx0 - t.x0 + i - 4, -lu, t.uw + ru - 1);
      const int xa = xa_ > -inside_p.xorigin ? xa_ : -inside_p.xorigin;
      const bool one_plane = inside_p.data == outside_p.data;   // workgroup-uniform (the search filters a unit in isolation)
      const int rows = th2 + 6;
      constexpr int NPASS = (TROWS + 6 + SROWS - 1) / SROWS;
      uint32_t v[NPASS];                     // every load of the column in flight before the first LDS store
#pragma unroll
      for (int q = 0; q < NPASS; q++) {
        const int j = jj + q * SROWS;
        const int cy = clampi(t.y0 + t.ty0 + j - 4, 0, t.crop_h - 1);   // (rows past the tile clamp to a valid address)
        const int ly_ = clampi(cy, t.y0 - t.top, t.y0 + h2 + 1);
        const int ly = ly_ > -inside_p.yorigin ? ly_ : -inside_p.yorigin;
        const bool inside = ly >= t.y0 && ly < t.y0 + h2 && (!STRICT || xa >= t.x0);
        if (one_plane) v[q] = ld_px_at<BPP>(inside_p, px_off<BPP>(inside_p, xa, ly));
        else v[q] = inside ? ld_px_at<BPP>(inside_p, px_off<BPP>(inside_p, xa, ly)) : ld_px_at<BPP>(outside_p, px_off<BPP>(outside_p, xa, ly));
      }
#pragma unroll
      for (int q = 0; q < NPASS; q++) {
        const int j = jj + q * SROWS;
        if (j < rows) S[j][i] = (uint16_t)v[q];
      }
    }
  }
  __syncthreads();
  // ---- 2: (a, b) of both passes ----
  // A thread owns a COLUMN of (a, b) centres over a segment of rows and slides the box down: per new
  // centre three (five) pixels of one new row (two new rows for the radius-2 pass, whose centres sit on
  // every other row) instead of the whole 3x3 (5x5) box -- 9 -> 3 and 25 -> 10 LDS reads per centre.
  {
    constexpr int NSEG = 256 / AW;          // row segments per column
    const int seg = tid / AW, c = tid - seg * AW;
    if (seg < NSEG && c <= t.tw + 1) {
      if (s1 > 0) {
        const int rows1 = t.th + 2, per1 = (rows1 + NSEG - 1) / NSEG;
        const int r0 = seg * per1, r1 = r0 + per1 < rows1 ? r0 + per1 : rows1;
        auto row3 = [&](int j, uint32_t &sm, uint32_t &sq) {   // S[j][c + 2 .. c + 4]
          const uint32_t v0 = S[j][c + 2], v1 = S[j][c + 3], v2 = S[j][c + 4];
          sm = v0 + v1 + v2;
          sq = __umul24(v0, v0) + __umul24(v1, v1) + __umul24(v2, v2);
        };
        if (r0 < r1) {
          uint32_t sa, qa, sb, qb, sc, qc;
          row3(r0 + 2, sa, qa);
          row3(r0 + 3, sb, qb);
          for (int r = r0; r < r1; r++) {   // centre (c - 1, r - 1): S rows r + 2 .. r + 4
            row3(r + 4, sc, qc);
            ab1[r][c] = sum_finish<NARROW>(qa + qb + qc, sa + sb + sc, 9, 455, s1, bd, atab);
            sa = sb; qa = qb; sb = sc; qb = qc;
          }
        }
      }
      if (s2 > 0) {
        const int nr = th2 / 2 + 1, per2 = (nr + NSEG - 1) / NSEG;
        const int r0 = seg * per2, r1 = r0 + per2 < nr ? r0 + per2 : nr;
        auto row5 = [&](int j, uint32_t &sm, uint32_t &sq) {   // S[j][c + 1 .. c + 5]
          sm = 0; sq = 0;
#pragma unroll
          for (int dx = 0; dx < 5; dx++) {
            const uint32_t v = S[j][c + 1 + dx];
            sm += v;
            sq += __umul24(v, v);
          }
        };
        if (r0 < r1) {
          uint32_t m1, q1, m2, q2, m3, q3, m4, q4, m5, q5;
          row5(2 * r0 + 1, m1, q1);
          row5(2 * r0 + 2, m2, q2);
          row5(2 * r0 + 3, m3, q3);
          for (int r = r0; r < r1; r++) {   // centre (c - 1, 2 r - 1): S rows 2 r + 1 .. 2 r + 5
            row5(2 * r + 4, m4, q4);
            row5(2 * r + 5, m5, q5);
            ab2[r][c] = sum_finish<NARROW>(q1 + q2 + q3 + q4 + q5, m1 + m2 + m3 + m4 + m5, 25, 164, s2, bd, atab);
            m1 = m3; q1 = q3; m2 = m4; q2 = q4; m3 = m5; q3 = q5;
          }
        }
      }
    }
  }
  __syncthreads();
  // ---- 3: the weighted stencils ----
  // A thread owns a pixel COLUMN over a segment of rows (an even number of them: the radius-2 pass pairs
  // rows) and walks down: the 3x3 stencil of the radius-1 pass is (3 4 3) on its outer rows and (4 4 4)
  // on the middle one, so a row of (a, b) pairs is read once and its two horizontal forms kept; the
  // radius-2 pass reads one row of pairs per TWO pixel rows.  Weight sums: A <= 32 * 256, p < 2^12: 24-bit products.
  if (ny > 0) {
    auto row1 = [&](int j, uint32_t &oa, uint32_t &ob, uint32_t &ma, uint32_t &mb) {   // ab1 row j at x .. x + 2
      const uint32_t v0 = ab1[j][x], v1 = ab1[j][x + 1], v2 = ab1[j][x + 2];
      const uint32_t a0 = v0 & 511u, a1 = v1 & 511u, a2 = v2 & 511u;
      const uint32_t b0 = v0 >> 9, b1 = v1 >> 9, b2 = v2 >> 9;
      const uint32_t as = a0 + a2, bs = b0 + b2;
      oa = 3u * as + 4u * a1;
      ob = 3u * bs + 4u * b1;
      ma = 4u * (as + a1);
      mb = 4u * (bs + b1);
    };
    auto row2 = [&](int r, uint32_t &ha, uint32_t &hb) {   // ab2 row r at x .. x + 2: (5 6 5)
      const uint32_t v0 = ab2[r][x], v1 = ab2[r][x + 1], v2 = ab2[r][x + 2];
      ha = 5u * ((v0 & 511u) + (v2 & 511u)) + 6u * (v1 & 511u);
      hb = 5u * ((v0 >> 9) + (v2 >> 9)) + 6u * (v1 >> 9);
    };
    // one straight-line body per (radius-1 pass on, radius-2 pass on): the parameter set is workgroup-uniform
    auto stencil = [&](auto has1, auto has2) {
      constexpr bool H1 = decltype(has1)::value, H2 = decltype(has2)::value;
      uint32_t oa0 = 0, ob0 = 0, oa1 = 0, ob1 = 0, ma1 = 0, mb1 = 0, dump_a, dump_b;
      if constexpr (H1) {
        row1(y0, oa0, ob0, dump_a, dump_b);
        row1(y0 + 1, oa1, ob1, ma1, mb1);
      }
      uint32_t ha0 = 0, hb0 = 0;
      if constexpr (H2) row2(y0 / 2, ha0, hb0);
#pragma unroll
      for (int k = 0; k < PER_MAX; k += 2) {
        if (k < ny) {
          const int y = y0 + k;                                   // an even row and, below, the odd row after it
          const uint32_t pe = S[y + 4][x + 4];
          uint32_t f1 = pe << 4, f2 = pe << 4, ha1 = 0, hb1 = 0;   // sgrproj_box_f_r0 once per row pair: the odd row
          if constexpr (H1) {                                     // reuses the even row's value
            uint32_t oa2, ob2, ma2, mb2;
            row1(y + 2, oa2, ob2, ma2, mb2);
            f1 = mad_u24(oa0 + ma1 + oa2, pe, ob0 + mb1 + ob2 + (1u << 8)) >> 9;
            oa0 = oa1; ob0 = ob1;
            oa1 = oa2; ob1 = ob2; ma1 = ma2; mb1 = mb2;
          }
          if constexpr (H2) {
            row2(y / 2 + 1, ha1, hb1);
            f2 = mad_u24(ha0 + ha1, pe, hb0 + hb1 + (1u << 8)) >> 9;
          }
          emit(x, y, pe, f1, f2, extra[k]);
          if (k + 1 < ny) {
            const uint32_t po = S[y + 5][x + 4];
            if constexpr (H1) {
              uint32_t oa2, ob2, ma2, mb2;
              row1(y + 3, oa2, ob2, ma2, mb2);
              f1 = mad_u24(oa0 + ma1 + oa2, po, ob0 + mb1 + ob2 + (1u << 8)) >> 9;
              oa0 = oa1; ob0 = ob1;
              oa1 = oa2; ob1 = ob2; ma1 = ma2; mb1 = mb2;
            } else {
              f1 = po << 4;
            }
            if constexpr (H2) {
              f2 = mad_u24(ha1, po, hb1 + (1u << 7)) >> 8;
              ha0 = ha1; hb0 = hb1;
            }
            emit(x, y + 1, po, f1, f2, extra[k + 1]);
          }
          if ((k & 2) != 0) flush();
        }
      }
      flush();
    };
    if (s1 > 0 && s2 > 0) stencil(std::true_type(), std::true_type());
    else if (s1 > 0) stencil(std::true_type(), std::false_type());
    else if (s2 > 0) stencil(std::false_type(), std::true_type());
    else stencil(std::false_type(), std::false_type());
  }
}

template <int BPP>
__global__ __launch_bounds__(256) void k_lrf_sgr(R1Plane cdeffed, R1Plane deblocked, R1Plane out,
                                                 LrfGeom g, const R1LrfUnit *__restrict__ units) {
  const int si = blockIdx.y, chunk = blockIdx.x;
  // stripe geometry (lrf.rs:1507-1517)
  int y0, sh_;
  if (si == 0) {
    y0 = 0;
    sh_ = (64 - 8) >> g.ydec;
  } else {
    y0 = (si * 64 - 8) >> g.ydec;
    const int rest = g.crop_h - y0;
    sh_ = (64 >> g.ydec) < rest ? (64 >> g.ydec) : rest;
  }
  if (sh_ <= 0) return;
  // unit of this chunk (the last unit stretches to the crop width)
  const int cx0 = chunk * TW;
  if (cx0 >= g.crop_w) return;
  int rux = cx0 / g.unit_size;
  rux = rux < g.unit_cols - 1 ? rux : g.unit_cols - 1;
  const int x0 = rux * g.unit_size;
  const int uw = rux == g.unit_cols - 1 ? g.crop_w - x0 : g.unit_size;
  int ruy = si * g.stripe_height / g.unit_size;
  ruy = ruy < g.unit_rows - 1 ? ruy : g.unit_rows - 1;
  const R1LrfUnit u = units[ruy * g.unit_cols + rux];
  if (u.filter != 3) return;   // RESTORE_NONE: `out` already holds the CDEF output
  SgrTile t;
  t.x0 = x0; t.y0 = y0; t.uw = uw; t.uh = sh_;
  t.lu = x0 == 0 ? 0 : 4; t.top = 2;
  t.crop_w = g.crop_w; t.crop_h = g.crop_h;
  t.cx0 = cx0; t.ty0 = 0;
  t.tw = (x0 + uw - cx0) < TW ? (x0 + uw - cx0) : TW;
  t.th = sh_;
  const int w0 = u.xqd[0], w1 = u.xqd[1], w2 = 128 - w0 - w1;
  const int32_t pmax = (1 << g.bd) - 1;
  sgr_tile<BPP, 64, BPP == 1>(cdeffed, deblocked, t, u.set, BPP == 1 ? 8 : g.bd, nullptr, 0, 0,
                              [&](int x, int y, uint32_t p, uint32_t f1, uint32_t f2, uint32_t) {
    // apply_filter (lrf.rs:796-815)
    const int32_t v = w0 * (int32_t)f2 + w1 * (int32_t)(p << 4) + w2 * (int32_t)f1;
    const int32_t s = (v + (1 << 10)) >> 11;
    const int32_t o = s < 0 ? 0 : (s > pmax ? pmax : s);
    uint8_t *d = (uint8_t *)px_addr<BPP>(out, cx0 + x, y0 + y);
    if constexpr (BPP == 1) *d = (uint8_t)o;
    else *(uint16_t *)d = (uint16_t)o;
  }, [] {});
}

// sgrproj_solve's moments (lrf.rs:1010-1054): grid.x = tiles of the largest
// unit, grid.y = (unit, set) pairs; five i64 sums per pair, accumulated with
// atomics (integer sums: exact in any order, like the reference's f64
// accumulation of per-line i64 sums, which never leaves the exact range).
template <int BPP>
__global__ __launch_bounds__(256) void k_sgr_moments(R1Plane cdeffed, R1Plane input,
                                                     const R1SgrSolveUnit *__restrict__ units,
                                                     long long *__restrict__ acc) {
  __shared__ long long part[4][5];
  const R1SgrSolveUnit u = units[blockIdx.y];
  const int ntx = (u.w + TW - 1) / TW, nty = (u.h + 63) / 64;
  if ((int)blockIdx.x >= ntx * nty || u.set > 15) return;   // workgroup-uniform (set 255: r1_lrf_search_batch's "no filter")
  SgrTile t;
  t.x0 = u.x; t.y0 = u.y; t.uw = u.w; t.uh = u.h;
  sgr_unit_edges(t, u.edges);
  t.crop_w = u.x + u.w; t.crop_h = u.y + u.h;   // hard-clipped to the unit (rdo.rs:2651-2666)
  const int tx = blockIdx.x % ntx, ty = blockIdx.x / ntx;
  t.cx0 = u.x + tx * TW;
  t.ty0 = ty * 64;
  t.tw = (u.w - tx * TW) < TW ? (u.w - tx * TW) : TW;
  t.th = (u.h - ty * 64) < 64 ? (u.h - ty * 64) : 64;
  long long m[5] = {0, 0, 0, 0, 0};
  sgr_tile<BPP, 64, BPP == 1>(cdeffed, cdeffed, t, u.set, BPP == 1 ? 8 : cdeffed.bit_depth, &input, t.cx0, u.y + t.ty0,
                [&](int x, int y, uint32_t p, uint32_t f1, uint32_t f2, uint32_t in_px) {
    const int32_t uu = (int32_t)(p << 4);
    const long long sv = ((int32_t)in_px << 4) - uu;
    const long long g2 = (int32_t)f2 - uu, g1 = (int32_t)f1 - uu;
    m[0] += g2 * g2; m[1] += g1 * g1; m[2] += g1 * g2; m[3] += g2 * sv; m[4] += g1 * sv;
  }, [] {});
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 5; k++) {
    uint32_t lo = (uint32_t)m[k], hi = (uint32_t)((unsigned long long)m[k] >> 32);
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
      const unsigned long long o = ((unsigned long long)(uint32_t)__shfl_xor((int)hi, s, 64) << 32) |
                                   (uint32_t)__shfl_xor((int)lo, s, 64);
      const unsigned long long v = (((unsigned long long)hi << 32) | lo) + o;
      lo = (uint32_t)v;
      hi = (uint32_t)(v >> 32);
    }
    if (lane == 0) part[wave][k] = (long long)(((unsigned long long)hi << 32) | lo);
  }
  __syncthreads();
  if (threadIdx.x < 5) {
    const long long v = part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] +
                        part[3][threadIdx.x];
    if (v) atomicAdd((unsigned long long *)acc + blockIdx.y * 5 + threadIdx.x, (unsigned long long)v);
  }
}

// the 2x2 solve in IEEE doubles, operation for operation (lrf.rs:1057-1095): m = the five moments
// (h00, h11, h01, c0, c1) of a w x h unit
__device__ __forceinline__ void sgr_solve_xqd(const long long *m, int w, int h, int set, int8_t *xqd) {
  if (set > 15) {   // no parameter set: no weights
    xqd[0] = xqd[1] = 0;
    return;
  }
  const uint32_t s2 = kSgrS[set & 15][0], s1 = kSgrS[set & 15][1];
  const double nn = __dmul_rn((double)w, (double)h);
  const double h00 = __ddiv_rn((double)m[0], nn), h11 = __ddiv_rn((double)m[1], nn);
  const double h01 = __ddiv_rn((double)m[2], nn);
  const double sc = __ddiv_rn(128.0, nn);
  const double c0 = __dmul_rn((double)m[3], sc), c1 = __dmul_rn((double)m[4], sc);
  double xq0 = 0., xq1 = 0.;
  if (s2 == 0) {
    if (h11 != 0.) xq1 = round(__ddiv_rn(c1, h11));
  } else if (s1 == 0) {
    if (h00 != 0.) xq0 = round(__ddiv_rn(c0, h00));
  } else {
    const double det = __fma_rn(h00, h11, -__dmul_rn(h01, h01));
    if (det != 0.) {
      xq0 = round(__ddiv_rn(__fma_rn(h11, c0, -__dmul_rn(h01, c1)), det));
      xq1 = round(__ddiv_rn(__fma_rn(h00, c1, -__dmul_rn(h01, c0)), det));
    }
  }
  auto sat = [](double v) -> long long {   // `as i32`
    if (v != v) return 0;
    return v > 2147483647. ? 2147483647ll : (v < -2147483648. ? -2147483648ll : (long long)v);
  };
  const long long q0 = sat(xq0), q1 = sat(xq1);
  const long long x0 = q0 < -96 ? -96 : (q0 > 31 ? 31 : q0);
  const long long t = 128 - x0 - q1;
  xqd[0] = (int8_t)x0;
  xqd[1] = (int8_t)(t < -32 ? -32 : (t > 95 ? 95 : t));
}

__global__ void k_sgr_solve(const R1SgrSolveUnit *__restrict__ units, const long long *__restrict__ acc,
                            int n, int8_t *__restrict__ xqd) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const R1SgrSolveUnit u = units[i];
  sgr_solve_xqd(acc + (size_t)i * 5, u.w, u.h, u.set, xqd + 2 * i);
}

// rdo_loop_plane_error's term for one block of the unit (rdo.rs:2060-2088): `test` = the filtered unit
// in LDS (row stride TS pixels), (px, py) = the block's position in the plane
template <int BPP, bool CHROMA, int TS, typename PT>
__device__ __forceinline__ unsigned long long lrf_block_err(const R1Plane &src, const PT *test, int px, int py,
                                                            int bw, int bh, int xdec, int ydec,
                                                            const uint32_t *__restrict__ scales, int scale_stride,
                                                            int bd) {
  const uint8_t *po = px_addr<BPP>(src, px, py);
  const size_t so = (size_t)src.stride * BPP;
  if constexpr (!CHROMA) {
    uint32_t sum_s = 0, sum_d = 0, sum_s2 = 0, sum_d2 = 0, sum_sd = 0;
    for (int r = 0; r < 8; r++)
      for (int i = 0; i < 8; i++) {
        const uint32_t sv = (uint32_t)ld_px<BPP>(po + r * so + (size_t)i * BPP), dv = test[r * TS + i];
        sum_s += sv; sum_d += dv;
        sum_s2 += sv * sv; sum_d2 += dv * dv; sum_sd += sv * dv;
      }
    // RawDistortion(cdef_dist_kernel) * bias: the tail multiplies by the block's DistortionScale
    return r1dist::cdef_tile_tail<0>(sum_s, sum_d, sum_s2, sum_d2, sum_sd, 64, px, py, scales, scale_stride, bd);
  } else {
    // sse_wxh with one bias for the block: get_weighted_sse over its 4x4 cells (dist.rs:234-283)
    const uint32_t sc = scales ? scales[(size_t)((py << ydec) >> 3) * scale_stride + ((px << xdec) >> 3)]
                               : (1u << 14);
    unsigned long long sum = 0;
    for (int cy = 0; cy < bh; cy += 4)
      for (int cx = 0; cx < bw; cx += 4) {
        uint32_t cell = 0;
        for (int r = 0; r < 4; r++)
          for (int i = 0; i < 4; i++) {
            const int32_t d = (int32_t)ld_px<BPP>(po + (cy + r) * so + (size_t)(cx + i) * BPP) -
                              (int32_t)test[(cy + r) * TS + cx + i];
            cell += (uint32_t)(d * d);
          }
        sum += ((unsigned long long)cell * sc + 128) >> 8;
      }
    return (sum + 32) >> 6;
  }
}

// sum of a 64-bit value over the workgroup (256 threads); valid in thread 0
__device__ __forceinline__ unsigned long long wg_sum_u64(unsigned long long v, unsigned long long *part4) {
  uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
#pragma unroll
  for (int sft = 1; sft < 64; sft <<= 1) {
    const unsigned long long o = ((unsigned long long)(uint32_t)__shfl_xor((int)hi, sft, 64) << 32) |
                                 (uint32_t)__shfl_xor((int)lo, sft, 64);
    const unsigned long long t = (((unsigned long long)hi << 32) | lo) + o;
    lo = (uint32_t)t;
    hi = (uint32_t)(t >> 32);
  }
  __syncthreads();   // part4 may still be read from a previous use
  if ((threadIdx.x & 63) == 0) part4[threadIdx.x >> 6] = ((unsigned long long)hi << 32) | lo;
  __syncthreads();
  return part4[0] + part4[1] + part4[2] + part4[3];
}

// The restoration leg of rdo_loop_decision, per (unit, set) pair (src/rdo.rs:2575-2763): the unit
// filtered with the weights k_sgr_solve just wrote -- sgrproj_stripe_filter on the unit's OWN padded
// image (hard-clipped like the solve), never stored -- and rdo_loop_plane_error of the result against
// the source (rdo.rs:2027-2093): per 8x8-luma block cdef_dist_kernel * bias (luma) or sse_wxh with
// |_, _| bias on (8 >> xdec) x (8 >> ydec) pixels (chroma).  Same tiling as k_sgr_moments; a tile's
// filtered pixels go to LDS, one thread per block sums its block, the workgroup's total is added to
// the pair's plane sum.  set = 255: the "no filter option" (the unit of lrf_in as it is).
template <int BPP, bool CHROMA>
__global__ __launch_bounds__(256) void k_sgr_unit_err(R1Plane lrf_in, R1Plane src,
                                                      const R1SgrSolveUnit *__restrict__ units,
                                                      const int8_t *__restrict__ xqd, int xdec, int ydec,
                                                      const uint32_t *__restrict__ scales, int scale_stride,
                                                      unsigned long long *__restrict__ acc) {
  __shared__ uint16_t F[64][TW];
  __shared__ unsigned long long part[4];
  const R1SgrSolveUnit u = units[blockIdx.y];
  const int ntx = (u.w + TW - 1) / TW, nty = (u.h + 63) / 64;
  if ((int)blockIdx.x >= ntx * nty) return;   // workgroup-uniform
  SgrTile t;
  t.x0 = u.x; t.y0 = u.y; t.uw = u.w; t.uh = u.h;
  sgr_unit_edges(t, u.edges);
  t.crop_w = u.x + u.w; t.crop_h = u.y + u.h;   // hard-clipped to the unit (rdo.rs:2651-2666)
  const int tx = blockIdx.x % ntx, ty = blockIdx.x / ntx;
  t.cx0 = u.x + tx * TW;
  t.ty0 = ty * 64;
  t.tw = (u.w - tx * TW) < TW ? (u.w - tx * TW) : TW;
  t.th = (u.h - ty * 64) < 64 ? (u.h - ty * 64) : 64;
  const int bd = lrf_in.bit_depth;
  if (u.set > 15) {
    for (int e = threadIdx.x; e < t.th * TW; e += 256) {
      const int y = e / TW, x = e % TW;   // TW is a constant: no runtime division
      if (x < t.tw) F[y][x] = (uint16_t)ld_px<BPP>(px_addr<BPP>(lrf_in, t.cx0 + x, u.y + t.ty0 + y));
    }
  } else {
    const int w0 = xqd[2 * blockIdx.y], w1 = xqd[2 * blockIdx.y + 1], w2 = 128 - w0 - w1;
    const int32_t pmax = (1 << bd) - 1;
    sgr_tile<BPP, 64, BPP == 1>(lrf_in, lrf_in, t, u.set, bd, nullptr, 0, 0,
                                [&](int x, int y, uint32_t p, uint32_t f1, uint32_t f2, uint32_t) {
      // apply_filter (lrf.rs:796-815)
      const int32_t v = w0 * (int32_t)f2 + w1 * (int32_t)(p << 4) + w2 * (int32_t)f1;
      const int32_t sft = (v + (1 << 10)) >> 11;
      F[y][x] = (uint16_t)(sft < 0 ? 0 : (sft > pmax ? pmax : sft));
    }, [] {});
  }
  __syncthreads();
  const int bw = CHROMA ? 8 >> xdec : 8, bh = CHROMA ? 8 >> ydec : 8;
  const int nbx = t.tw / bw, nby = t.th / bh;
  unsigned long long mine = 0;
  if ((int)threadIdx.x < nbx * nby) {
    const int by = (int)threadIdx.x / nbx, bx = (int)threadIdx.x - by * nbx;
    mine = lrf_block_err<BPP, CHROMA, TW>(src, &F[by * bh][bx * bw], t.cx0 + bx * bw, u.y + t.ty0 + by * bh, bw, bh,
                                          xdec, ydec, scales, scale_stride, bd);
  }
  const unsigned long long v = wg_sum_u64(mine, part);
  if (threadIdx.x == 0 && v) atomicAdd(acc + blockIdx.y, v);
}

// The same leg in ONE launch for units up to 64 x 64 pixels (the 64x64 luma / 32x32 chroma units of the
// speed settings the encoder ships): a workgroup owns a (unit, set) pair, the two filter outputs of every
// pixel stay in LDS between the moments and the projection, so the box filters run once.
// PACK: both filter outputs of a pixel in one dword (f <= 16 * 1023 + rounding: up to 10 bits; at 12 bits an
// all-white unit reaches 65588)
// Occupancy (ab9, ab10): with 32-row tiles an 8-bit workgroup holds 31 KB of LDS and 96 VGPRs -- five per CU -- and a
// 16-bit one 35 KB -- four (three with the 64-row tile of round 4).  The 8-bit kernels are asked for five: with the unit's
// edge flags (t.lu / t.top) they would settle at 106 VGPRs otherwise; the request costs 12 B of scratch
template <int BPP, bool CHROMA, bool PACK>
__global__ __launch_bounds__(256, BPP == 1 ? 5 : 1) void k_lrf_search_unit(R1Plane lrf_in, R1Plane src,
                                                         const R1SgrSolveUnit *__restrict__ units, int xdec, int ydec,
                                                         const uint32_t *__restrict__ scales, int scale_stride,
                                                         uint32_t dist_scale, int8_t *__restrict__ xqd_out,
                                                         unsigned long long *__restrict__ err_out) {
  typedef typename std::conditional<BPP == 1, uint8_t, uint16_t>::type PT;
  __shared__ uint32_t F1[64][64];
  __shared__ uint32_t F2[PACK ? 1 : 64][64];         // PACK: f1 | f2 << 16 in F1
  __shared__ __attribute__((aligned(16))) PT P[64][64];   // the unit's pixels, then the filtered unit
  __shared__ long long mpart[4][5];
  __shared__ unsigned long long epart[4];
  __shared__ int8_t xq[2];
  // Consecutive workgroup ids go to the eight XCDs in turn, each with its own L2.  Callers list the parameter sets of
  // a unit next to each other (rdo_loop_decision's loop order): handing an XCD a CONTIGUOUS run of pairs keeps the nine
  // launches that read the same unit -- its pixels and the source's -- on one L2 (before: every set of a unit fetched
  // it again, 135 MB a luma launch for 17 MB of planes; the tile loads are a quarter of a wave's life)
#ifndef R1_LRF_XCD_RUNS
#define R1_LRF_XCD_RUNS 1   // A/B switch
#endif
  const int pair = R1_LRF_XCD_RUNS ? xcd_run_item(blockIdx.x, gridDim.x) : (int)blockIdx.x;   // common.hpp
  const R1SgrSolveUnit u = units[pair];
  const int bd = BPP == 1 ? 8 : lrf_in.bit_depth;
  if (u.w > 64 || u.h > 64 || u.w <= 0 || u.h <= 0) {   // not what max_w / max_h promised: no result
    if (threadIdx.x == 0) {
      err_out[pair] = ~0ull;
      xqd_out[2 * pair] = xqd_out[2 * pair + 1] = 0;
    }
    return;
  }
  if (u.set > 15) {
    for (int e = threadIdx.x; e < 64 * u.h; e += 256) {
      const int y = e >> 6, x = e & 63;   // rows of 64: no runtime division
      if (x < u.w) P[y][x] = (PT)ld_px<BPP>(px_addr<BPP>(lrf_in, u.x + x, u.y + y));
    }
    if (threadIdx.x == 0) xqd_out[2 * pair] = xqd_out[2 * pair + 1] = 0;
  } else {
    long long m[5] = {0, 0, 0, 0, 0};
    // rows per tile: 32 -- the tile arrays are 9 KB smaller than with 64 and one more workgroup fits a CU at either
    // pixel width; the two extra tiles of a 64-row luma unit cost less than that buys since the tile's fixed part shrank
    // (r05_ab_notes.md ab9 / ab10)
#ifndef R1_LRF_SEARCH_TROWS
#define R1_LRF_SEARCH_TROWS 32
#endif
    constexpr int TR = R1_LRF_SEARCH_TROWS;
    const int ntx = (u.w + TW - 1) / TW;
    for (int ty = 0; ty < u.h; ty += TR)
    for (int tx = 0; tx < ntx; tx++) {
      SgrTile t;
      t.x0 = u.x; t.y0 = u.y; t.uw = u.w; t.uh = u.h;
      sgr_unit_edges(t, u.edges);
      t.crop_w = u.x + u.w; t.crop_h = u.y + u.h;   // clipped to the unit on the right and below (rdo.rs:2651-2666)
      t.cx0 = u.x + tx * TW;
      t.ty0 = ty;
      t.tw = (u.w - tx * TW) < TW ? (u.w - tx * TW) : TW;
      t.th = (u.h - ty) < TR ? (u.h - ty) : TR;
      // PACK (bit depth <= 10): f - u and s - u are 14-bit-and-a-sign differences of Q4 pixels (|f - u| <= 16 * 1023 +
      // rounding), their products < 2^28.1: four of them fit an int32, so the moments of a thread's four rows are
      // gathered with the full-rate 24-bit multiply-add and widened once per four rows instead of five quarter-rate
      // 64-bit multiply-adds per pixel
      int32_t a32[5] = {0, 0, 0, 0, 0};
      sgr_tile<BPP, TR, PACK>(lrf_in, lrf_in, t, u.set, bd, &src, u.x + tx * TW, u.y + ty,
                              [&](int x, int yt, uint32_t p, uint32_t f1, uint32_t f2, uint32_t src_px) {
        const int X = tx * TW + x, y = ty + yt;
        if constexpr (!PACK) { F1[y][X] = f1; F2[y][X] = f2; }
        else F1[y][X] = f1 | (f2 << 16);
        P[y][X] = (PT)p;
        const int32_t uu = (int32_t)(p << 4);
        if constexpr (PACK) {
          const int32_t sv = ((int32_t)src_px << 4) - uu, g2 = (int32_t)f2 - uu, g1 = (int32_t)f1 - uu;
          a32[0] = mad_i24(g2, g2, a32[0]); a32[1] = mad_i24(g1, g1, a32[1]); a32[2] = mad_i24(g1, g2, a32[2]);
          a32[3] = mad_i24(g2, sv, a32[3]); a32[4] = mad_i24(g1, sv, a32[4]);
        } else {
          const long long sv = ((int32_t)src_px << 4) - uu;
          const long long g2 = (int32_t)f2 - uu, g1 = (int32_t)f1 - uu;
          m[0] += g2 * g2; m[1] += g1 * g1; m[2] += g1 * g2; m[3] += g2 * sv; m[4] += g1 * sv;
        }
      }, [&] {
        if constexpr (PACK) {
#pragma unroll
          for (int k = 0; k < 5; k++) { m[k] += a32[k]; a32[k] = 0; }
        }
      });
      __syncthreads();   // the tile's LDS is staged again by the next one
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 5; k++) {
      uint32_t lo = (uint32_t)m[k], hi = (uint32_t)((unsigned long long)m[k] >> 32);
#pragma unroll
      for (int sft = 1; sft < 64; sft <<= 1) {
        const unsigned long long o = ((unsigned long long)(uint32_t)__shfl_xor((int)hi, sft, 64) << 32) |
                                     (uint32_t)__shfl_xor((int)lo, sft, 64);
        const unsigned long long v = (((unsigned long long)hi << 32) | lo) + o;
        lo = (uint32_t)v;
        hi = (uint32_t)(v >> 32);
      }
      if (lane == 0) mpart[wave][k] = (long long)(((unsigned long long)hi << 32) | lo);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      long long tot[5];
      for (int k = 0; k < 5; k++) tot[k] = mpart[0][k] + mpart[1][k] + mpart[2][k] + mpart[3][k];
      sgr_solve_xqd(tot, u.w, u.h, u.set, xq);
      xqd_out[2 * pair] = xq[0];
      xqd_out[2 * pair + 1] = xq[1];
    }
    __syncthreads();
    const int w0 = xq[0], w1 = xq[1], w2 = 128 - w0 - w1;
    const int32_t pmax = (1 << bd) - 1;
    for (int e = threadIdx.x; e < 64 * u.h; e += 256) {
      const int y = e >> 6, x = e & 63;
      if (x >= u.w) continue;
      uint32_t f1, f2;
      if constexpr (!PACK) { f1 = F1[y][x]; f2 = F2[y][x]; }
      else { f1 = F1[y][x] & 0xFFFFu; f2 = F1[y][x] >> 16; }
      // apply_filter (lrf.rs:796-815)
      const int32_t v = w0 * (int32_t)f2 + w1 * (int32_t)((uint32_t)P[y][x] << 4) + w2 * (int32_t)f1;
      const int32_t sft = (v + (1 << 10)) >> 11;
      P[y][x] = (PT)(sft < 0 ? 0 : (sft > pmax ? pmax : sft));
    }
  }
  __syncthreads();
  const int bw = CHROMA ? 8 >> xdec : 8, bh = CHROMA ? 8 >> ydec : 8;
  const int nbx = u.w / bw, nby = u.h / bh;
  unsigned long long mine = 0;
#ifndef R1_LRF_COOP_ERR
#define R1_LRF_COOP_ERR 1   // A/B switch: 0 = one thread per block (round 4)
#endif
  // rdo_loop_plane_error (rdo.rs:2027-2093), the whole workgroup on it.  Round 4 gave a block to a thread: 64 of
  // the 256 threads looped over 64 pixels each -- one-pixel global loads of the source at a stride of a plane
  // row -- while the other waves waited at the barrier.  Now a thread owns a ROW SEGMENT of a block (the 8 lanes
  // of a unit row read 64 contiguous source pixels), the rows of a block meet by xor-shuffles inside their wave
  // (a wave covers exactly one row of blocks), the five sums of every block are parked in LDS and ONE wave runs
  // the 64 fixed-point tails (ssim boost, 64-bit arithmetic) side by side instead of one after the other.
  if constexpr (R1_LRF_COOP_ERR && !CHROMA) {
    uint32_t(*bs)[5] = (uint32_t(*)[5]) & F1[0][0];   // 64 x 5 sums over the filter outputs, which are dead by now
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int half = 0; half < 2; half++) {
      const int y = half * 32 + wave * 8 + (lane >> 3), xs = lane & 7;    // unit row, 8-pixel segment of it
      uint32_t sum_s = 0, sum_d = 0, sum_s2 = 0, sum_d2 = 0, sum_sd = 0;
      if (xs < nbx && y < nby * 8) {
        const uint8_t *po = px_addr<BPP>(src, u.x + xs * 8, u.y + y);
        // eight source pixels in one load where the segment is aligned (units start at multiples of 8 pixels in
        // every configuration the encoder ships; anything else takes the pixel-by-pixel path), eight filtered
        // pixels in one LDS read
        PT sv8[8], dv8[8];
        if (((uintptr_t)po & (8 * BPP - 1)) == 0) {
          if constexpr (BPP == 1) *(uint2 *)sv8 = *(const uint2 *)po;
          else *(uint4 *)sv8 = *(const uint4 *)po;
        } else {
#pragma unroll
          for (int i = 0; i < 8; i++) sv8[i] = (PT)ld_px<BPP>(po + (size_t)i * BPP);
        }
        if constexpr (BPP == 1) *(uint2 *)dv8 = *(const uint2 *)&P[y][xs * 8];
        else *(uint4 *)dv8 = *(const uint4 *)&P[y][xs * 8];
#pragma unroll
        for (int i = 0; i < 8; i++) {
          const uint32_t sv = sv8[i], dv = dv8[i];
          sum_s += sv; sum_d += dv;
          sum_s2 += sv * sv; sum_d2 += dv * dv; sum_sd += sv * dv;
        }
      }
#pragma unroll
      for (int m = 8; m < 64; m <<= 1) {     // the 8 rows of a block: lanes 8 apart
        sum_s += __shfl_xor(sum_s, m, 64); sum_d += __shfl_xor(sum_d, m, 64);
        sum_s2 += __shfl_xor(sum_s2, m, 64); sum_d2 += __shfl_xor(sum_d2, m, 64);
        sum_sd += __shfl_xor(sum_sd, m, 64);
      }
      if (lane < 8) {
        uint32_t *b = bs[(half * 4 + wave) * 8 + lane];
        b[0] = sum_s; b[1] = sum_d; b[2] = sum_s2; b[3] = sum_d2; b[4] = sum_sd;
      }
    }
    __syncthreads();
    if (threadIdx.x < 64) {
      const int by = threadIdx.x >> 3, bx = threadIdx.x & 7;
      if (bx < nbx && by < nby) {
        const uint32_t *b = bs[threadIdx.x];
        // RawDistortion(cdef_dist_kernel) * bias: the tail multiplies by the block's DistortionScale
        mine = r1dist::cdef_tile_tail<0>(b[0], b[1], b[2], b[3], b[4], 64, u.x + bx * 8, u.y + by * 8, scales,
                                         scale_stride, bd);
      }
    }
#ifndef R1_LRF_COOP_ERR_CHROMA
#define R1_LRF_COOP_ERR_CHROMA 0   // measured (profiles/r05_ab_notes.md, ab4): the chroma form LOSES 5-6 % -- off
#endif
  } else if constexpr (R1_LRF_COOP_ERR_CHROMA && CHROMA) {
    if (bw == 4 && bh == 4) {
      // 4:2:0: a block is one 4x4 cell of get_weighted_sse (dist.rs:234-283) with the block's bias.  A thread owns
      // a 4-pixel row segment; a wave pass covers four unit rows = one row of cells; rows meet by xor-shuffles.
      const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
      for (int pass = 0; pass < 4; pass++) {
        const int y = (pass * 4 + wave) * 4 + (lane >> 4), xs = lane & 15;
        uint32_t cell = 0;
        const bool in = xs < nbx && y < nby * 4;
        if (in) {
          const uint8_t *po = px_addr<BPP>(src, u.x + xs * 4, u.y + y);
#pragma unroll
          for (int i = 0; i < 4; i++) {
            const int32_t d = (int32_t)ld_px<BPP>(po + (size_t)i * BPP) - (int32_t)P[y][xs * 4 + i];
            cell += (uint32_t)(d * d);
          }
        }
        cell += __shfl_xor(cell, 16, 64);
        cell += __shfl_xor(cell, 32, 64);
        if (in && lane < 16) {
          const int px = u.x + xs * 4, py = u.y + y;
          const uint32_t sc = scales ? scales[(size_t)((py << ydec) >> 3) * scale_stride + ((px << xdec) >> 3)] : (1u << 14);
          mine += ((((unsigned long long)cell * sc + 128) >> 8) + 32) >> 6;
        }
      }
    } else {
      for (int b = threadIdx.x; b < 16 * nby; b += 256) {
        const int by = b >> 4, bx = b & 15;
        if (bx >= nbx) continue;
        mine += lrf_block_err<BPP, CHROMA, 64>(src, &P[by * bh][bx * bw], u.x + bx * bw, u.y + by * bh, bw, bh, xdec, ydec,
                                               scales, scale_stride, bd);
      }
    }
  } else
  for (int b = threadIdx.x; b < 16 * nby; b += 256) {   // rows of 16 block slots (a unit is at most 16 blocks wide)
    const int by = b >> 4, bx = b & 15;
    if (bx >= nbx) continue;
    mine += lrf_block_err<BPP, CHROMA, 64>(src, &P[by * bh][bx * bw], u.x + bx * bw, u.y + by * bh, bw, bh, xdec, ydec,
                                           scales, scale_stride, bd);
  }
  const unsigned long long v = wg_sum_u64(mine, epart);
  // Distortion * fi.dist_scale[pli] (rdo.rs:2092; DistortionScale::mul_u64, rdo.rs:613-615)
  if (threadIdx.x == 0) err_out[pair] = ((unsigned long long)dist_scale * v + 8192) >> 14;
}

// A later pass of rdo_loop_decision's CDEF leg (rdo.rs:2407-2530): the superblock's trial output (cdef_search.hip,
// MODE 1: the plane `trial` of cdef_index blockIdx.z) restored with the unit's CURRENT choice -- setup_integral_image
// on the superblock alone (crop = the superblock; left / above it the area's working copy `cdef_cur` where the edge
// flags say so), sgrproj_stripe_filter with the chosen (set, xqd) -- and rdo_loop_plane_error of the restored
// superblock against the source, added to the (superblock, index, plane) sum the CDEF kernels use.
#ifndef R1_TRIAL_TROWS
#define R1_TRIAL_TROWS 64   // rows per tile (A/B: 32)
#endif
template <int BPP, bool CHROMA>
__global__ __launch_bounds__(256) void k_sgr_trial_err(R1Plane trial, size_t trial_idx_bytes, R1Plane cdef_cur, R1Plane src,
                                                       const R1TrialUnit *__restrict__ units, int pli, int xdec, int ydec,
                                                       const uint32_t *__restrict__ scales, int scale_stride,
                                                       unsigned long long *__restrict__ psum, int n_sb) {
  constexpr int TR = R1_TRIAL_TROWS;
  __shared__ __attribute__((aligned(16))) uint16_t F[TR][TW];
  __shared__ unsigned long long part[4];
  const R1TrialUnit u = units[blockIdx.y];
  const int idx = blockIdx.z;
  // a unit that is not what the header promises (a superblock's visible rectangle inside the plane, a known parameter
  // set, a superblock of this frame) is skipped, never read or accumulated: workgroup-uniform
  if (u.w <= 0 || u.h <= 0 || u.w > 64 || u.h > 64 || u.set > 15 || u.sb < 0 || u.sb >= n_sb || u.x < 0 || u.y < 0 ||
      u.x + u.w > trial.width || u.y + u.h > trial.height || u.x + u.w > src.width + 7 || u.y + u.h > src.height + 7)
    return;
  const int ntx = (u.w + TW - 1) / TW, nty = (u.h + TR - 1) / TR;
  if ((int)blockIdx.x >= ntx * nty) return;
  const int tx = (int)blockIdx.x % ntx, ty = (int)blockIdx.x / ntx;
  trial.data = (uint8_t *)trial.data + (size_t)idx * trial_idx_bytes;
  SgrTile t;
  t.x0 = u.x; t.y0 = u.y; t.uw = u.w; t.uh = u.h;
  sgr_unit_edges(t, u.edges);
  t.crop_w = u.x + u.w; t.crop_h = u.y + u.h;   // hard-clipped to the superblock (rdo.rs:2458-2466)
  t.cx0 = u.x + tx * TW;
  t.ty0 = ty * TR;
  t.tw = (u.w - tx * TW) < TW ? (u.w - tx * TW) : TW;
  t.th = (u.h - ty * TR) < TR ? (u.h - ty * TR) : TR;
  const int bd = BPP == 1 ? 8 : src.bit_depth;
  const int w0 = u.xqd[0], w1 = u.xqd[1], w2 = 128 - w0 - w1;
  const int32_t pmax = (1 << bd) - 1;
  sgr_tile<BPP, TR, BPP == 1, true>(trial, cdef_cur, t, u.set, bd, nullptr, 0, 0,
                                    [&](int x, int y, uint32_t p, uint32_t f1, uint32_t f2, uint32_t) {
    // apply_filter (lrf.rs:796-815)
    const int32_t v = w0 * (int32_t)f2 + w1 * (int32_t)(p << 4) + w2 * (int32_t)f1;
    const int32_t sft = (v + (1 << 10)) >> 11;
    F[y][x] = (uint16_t)(sft < 0 ? 0 : (sft > pmax ? pmax : sft));
  }, [] {});
  __syncthreads();
  const int bw = CHROMA ? 8 >> xdec : 8, bh = CHROMA ? 8 >> ydec : 8;
  const int nbx = t.tw / bw, nby = t.th / bh;
  unsigned long long mine = 0;
#ifndef R1_TRIAL_COOP
#define R1_TRIAL_COOP 1   // A/B switch: 0 = one thread per 8x8 block
#endif
  if constexpr (R1_TRIAL_COOP && !CHROMA) {
    // rdo_loop_plane_error of the tile with the whole workgroup (as k_lrf_search_unit does): a thread owns an
    // 8-pixel row segment of a block -- 16 contiguous source bytes, 16 bytes of LDS -- the 8 rows of a block meet by
    // xor-shuffles (lanes 4 apart), the block sums are parked in LDS and 32 threads run the fixed-point tails
    __shared__ uint32_t bs[TR / 2][5];
    const int y = (int)threadIdx.x >> 2, xs = (int)threadIdx.x & 3;
    uint32_t sum_s = 0, sum_d = 0, sum_s2 = 0, sum_d2 = 0, sum_sd = 0;
    if (xs < nbx && y < nby * 8 && y < TR) {
      typedef typename std::conditional<BPP == 1, uint8_t, uint16_t>::type PT;
      const uint8_t *po = px_addr<BPP>(src, t.cx0 + xs * 8, u.y + t.ty0 + y);
      PT sv8[8];
      if (((uintptr_t)po & (8 * BPP - 1)) == 0) {
        if constexpr (BPP == 1) *(uint2 *)sv8 = *(const uint2 *)po;
        else *(uint4 *)sv8 = *(const uint4 *)po;
      } else {
#pragma unroll
        for (int i = 0; i < 8; i++) sv8[i] = (PT)ld_px<BPP>(po + (size_t)i * BPP);
      }
      uint16_t dv8[8];
      *(uint4 *)dv8 = *(const uint4 *)&F[y][xs * 8];
#pragma unroll
      for (int i = 0; i < 8; i++) {
        const uint32_t sv = sv8[i], dv = dv8[i];
        sum_s += sv; sum_d += dv;
        sum_s2 += sv * sv; sum_d2 += dv * dv; sum_sd += sv * dv;
      }
    }
#pragma unroll
    for (int m = 4; m < 32; m <<= 1) {     // the 8 rows of a block: lanes 4 apart
      sum_s += __shfl_xor(sum_s, m, 64); sum_d += __shfl_xor(sum_d, m, 64);
      sum_s2 += __shfl_xor(sum_s2, m, 64); sum_d2 += __shfl_xor(sum_d2, m, 64);
      sum_sd += __shfl_xor(sum_sd, m, 64);
    }
    if ((y & 7) == 0 && y < TR) {
      uint32_t *b = bs[(y >> 3) * 4 + xs];
      b[0] = sum_s; b[1] = sum_d; b[2] = sum_s2; b[3] = sum_d2; b[4] = sum_sd;
    }
    __syncthreads();
    if (threadIdx.x < TR / 2) {
      const int by = (int)threadIdx.x >> 2, bx = (int)threadIdx.x & 3;
      if (bx < nbx && by < nby) {
        const uint32_t *b = bs[threadIdx.x];
        mine = r1dist::cdef_tile_tail<0>(b[0], b[1], b[2], b[3], b[4], 64, t.cx0 + bx * 8, u.y + t.ty0 + by * 8, scales, scale_stride, bd);
      }
    }
  } else
  if ((int)threadIdx.x < nbx * nby) {
    const int by = (int)threadIdx.x / nbx, bx = (int)threadIdx.x - by * nbx;
    mine = lrf_block_err<BPP, CHROMA, TW>(src, &F[by * bh][bx * bw], t.cx0 + bx * bw, u.y + t.ty0 + by * bh, bw, bh,
                                          xdec, ydec, scales, scale_stride, bd);
  }
  const unsigned long long v = wg_sum_u64(mine, part);
  if (threadIdx.x == 0 && v) atomicAdd(psum + (size_t)u.sb * 24 + idx * 3 + pli, v);
}

// Distortion * fi.dist_scale[pli] (rdo.rs:2092; DistortionScale::mul_u64, rdo.rs:613-615)
__global__ void k_lrf_err_finish(const unsigned long long *__restrict__ acc, int n, uint32_t dist_scale,
                                 unsigned long long *__restrict__ err) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) err[i] = ((unsigned long long)dist_scale * acc[i] + 8192) >> 14;
}

}  // namespace

// called by cdef_search.hip (r1_cdef_lrf_trial_batch); not part of the C ABI
__attribute__((visibility("hidden")))
int r1i_sgr_trial_err_launch(const R1Plane &trial, size_t trial_idx_bytes, const R1Plane &cdef_cur, const R1Plane &src,
                             const R1TrialUnit *units, int n_units, int n_idx, int pli, int xdec, int ydec,
                             const uint32_t *scales, int scale_stride, unsigned long long *psum, int n_sb, hipStream_t st) {
  R1_REQUIRE(lrf_plane_ok(&trial) && lrf_plane_ok(&cdef_cur) && lrf_plane_ok(&src));
  R1_REQUIRE(trial.bytes_per_px == src.bytes_per_px && cdef_cur.bytes_per_px == src.bytes_per_px);
  const dim3 grid((64 / TW) * (64 / R1_TRIAL_TROWS), n_units, n_idx);
#define R1_TRIAL(BPP, CH)                                                                                         \
  hipLaunchKernelGGL((k_sgr_trial_err<BPP, CH>), grid, dim3(256), 0, st, trial, trial_idx_bytes, cdef_cur, src, units, \
                     pli, xdec, ydec, scales, scale_stride, psum, n_sb)
  if (src.bytes_per_px == 1) {
    if (pli) R1_TRIAL(1, true); else R1_TRIAL(1, false);
  } else {
    if (pli) R1_TRIAL(2, true); else R1_TRIAL(2, false);
  }
#undef R1_TRIAL
  R1_HIP_CHECK(hipGetLastError());
  return R1_OK;
}

extern "C" int r1_lrf_sgrproj_plane(r1_ctx *ctx, const R1Plane *cdeffed, const R1Plane *deblocked,
                                    const R1Plane *out, int ydec, int crop_w, int crop_h,
                                    int frame_height, int unit_size, int unit_cols, int unit_rows,
                                    int stripe_height, const R1LrfUnit *units, void *stream) {
  R1_REQUIRE(ctx && cdeffed && deblocked && out && units);
  R1_REQUIRE(lrf_plane_ok(cdeffed) && lrf_plane_ok(deblocked));   // 32-bit byte offsets in the tile loads
  R1_REQUIRE(cdeffed->bytes_per_px == deblocked->bytes_per_px &&
             cdeffed->bytes_per_px == out->bytes_per_px);
  R1_REQUIRE(cdeffed->bytes_per_px == 1 || cdeffed->bytes_per_px == 2);
  R1_REQUIRE((cdeffed->bytes_per_px == 1) == (cdeffed->bit_depth == 8));
  R1_REQUIRE(cdeffed->data != out->data);   // the filter reads CDEF output around what it writes
  R1_REQUIRE(ydec >= 0 && ydec <= 1 && crop_w > 0 && crop_h > 0 && frame_height > 0);
  R1_REQUIRE(unit_size >= 32 && unit_size <= 256 && unit_size % 32 == 0);
  R1_REQUIRE(unit_cols > 0 && unit_rows > 0 && (stripe_height == 64 || stripe_height == 32));
  R1_REQUIRE((unit_cols - 1) * unit_size < crop_w);
  LrfGeom g;
  g.ydec = ydec;
  g.crop_w = crop_w;
  g.crop_h = crop_h;
  g.stripe_n = (frame_height + 7) / 64 + 1;
  g.unit_size = unit_size;
  g.unit_cols = unit_cols;
  g.unit_rows = unit_rows;
  g.stripe_height = stripe_height;
  g.bd = cdeffed->bit_depth;
  g.chunks = (crop_w + TW - 1) / TW;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(g.chunks, g.stripe_n);
  if (cdeffed->bytes_per_px == 1)
    hipLaunchKernelGGL(k_lrf_sgr<1>, grid, dim3(256), 0, st, *cdeffed, *deblocked, *out, g, units);
  else
    hipLaunchKernelGGL(k_lrf_sgr<2>, grid, dim3(256), 0, st, *cdeffed, *deblocked, *out, g, units);
  R1_HIP_CHECK(hipGetLastError());
  return R1_OK;
}

extern "C" int r1_sgrproj_solve_batch(r1_ctx *ctx, const R1Plane *cdeffed, const R1Plane *input,
                                      const R1SgrSolveUnit *units, int n, int max_w, int max_h,
                                      int64_t *moments_scratch, int8_t *xqd_out, void *stream) {
  R1_REQUIRE(ctx && cdeffed && input);
  R1_REQUIRE(lrf_plane_ok(cdeffed) && lrf_plane_ok(input));
  R1_REQUIRE(cdeffed->bytes_per_px == input->bytes_per_px && cdeffed->bit_depth == input->bit_depth);
  R1_REQUIRE(cdeffed->bytes_per_px == 1 || cdeffed->bytes_per_px == 2);
  R1_REQUIRE((cdeffed->bytes_per_px == 1) == (cdeffed->bit_depth == 8));
  R1_REQUIRE(max_w > 0 && max_h > 0 && max_w <= 384 && max_h <= 384);
  if (n <= 0) return R1_OK;
  R1_REQUIRE(units && moments_scratch && xqd_out);
  hipStream_t st = (hipStream_t)stream;
  R1_HIP_CHECK(hipMemsetAsync(moments_scratch, 0, (size_t)n * 5 * sizeof(int64_t), st));
  const dim3 grid(((max_w + TW - 1) / TW) * ((max_h + 63) / 64), n);
  if (cdeffed->bytes_per_px == 1)
    hipLaunchKernelGGL(k_sgr_moments<1>, grid, dim3(256), 0, st, *cdeffed, *input, units,
                       (long long *)moments_scratch);
  else
    hipLaunchKernelGGL(k_sgr_moments<2>, grid, dim3(256), 0, st, *cdeffed, *input, units,
                       (long long *)moments_scratch);
  hipLaunchKernelGGL(k_sgr_solve, dim3((n + 127) / 128), dim3(128), 0, st, units,
                     (const long long *)moments_scratch, n, xqd_out);
  R1_HIP_CHECK(hipGetLastError());
  return R1_OK;
}

// The restoration leg of rdo_loop_decision for one plane, everything but the rate (see
// k_sgr_unit_err): units as for r1_sgrproj_solve_batch, plus set = 255 for the "no filter option".
extern "C" int r1_lrf_search_batch(r1_ctx *ctx, const R1Plane *lrf_in, const R1Plane *src,
                                   const R1SgrSolveUnit *units, int n, int max_w, int max_h, int is_chroma,
                                   int xdec, int ydec, const uint32_t *scales, int scale_stride,
                                   uint32_t dist_scale, int64_t *scratch, int8_t *xqd_out,
                                   uint64_t *err_out, void *stream) {
  R1_REQUIRE(ctx && lrf_in && src);
  R1_REQUIRE(lrf_plane_ok(lrf_in) && lrf_plane_ok(src));
  R1_REQUIRE(lrf_in->bytes_per_px == src->bytes_per_px && lrf_in->bit_depth == src->bit_depth);
  R1_REQUIRE(lrf_in->bytes_per_px == 1 || lrf_in->bytes_per_px == 2);
  R1_REQUIRE((lrf_in->bytes_per_px == 1) == (lrf_in->bit_depth == 8));
  R1_REQUIRE(max_w > 0 && max_h > 0 && max_w <= 384 && max_h <= 384);
  R1_REQUIRE(xdec >= 0 && xdec <= 1 && ydec >= 0 && ydec <= 1 && (is_chroma || (!xdec && !ydec)));
  R1_REQUIRE(!scales || scale_stride > 0);
  if (n <= 0) return R1_OK;
  R1_REQUIRE(units && scratch && xqd_out && err_out);
  hipStream_t st = (hipStream_t)stream;
  if (max_w <= 64 && max_h <= 64) {
    // one launch: a workgroup per pair keeps the filter outputs in LDS between the solve and the projection
#define R1_LRF_UNIT(BPP, CH, PK)                                                                                  \
  hipLaunchKernelGGL((k_lrf_search_unit<BPP, CH, PK>), dim3(n), dim3(256), 0, st, *lrf_in, *src, units, xdec, ydec, \
                     scales, scale_stride, dist_scale, xqd_out, (unsigned long long *)err_out)
    if (lrf_in->bytes_per_px == 1) {
      if (is_chroma) R1_LRF_UNIT(1, true, true); else R1_LRF_UNIT(1, false, true);
    } else if (lrf_in->bit_depth <= 10) {
      if (is_chroma) R1_LRF_UNIT(2, true, true); else R1_LRF_UNIT(2, false, true);
    } else {
      if (is_chroma) R1_LRF_UNIT(2, true, false); else R1_LRF_UNIT(2, false, false);
    }
#undef R1_LRF_UNIT
    R1_HIP_CHECK(hipGetLastError());
    return R1_OK;
  }
  // larger units: moments, solve, then the box filters again for the error
  // scratch: 5 moments per pair, then the pair's plane sum
  R1_HIP_CHECK(hipMemsetAsync(scratch, 0, (size_t)n * 6 * sizeof(int64_t), st));
  unsigned long long *acc = (unsigned long long *)scratch + (size_t)n * 5;
  const dim3 grid(((max_w + TW - 1) / TW) * ((max_h + 63) / 64), n);
  if (lrf_in->bytes_per_px == 1)
    hipLaunchKernelGGL(k_sgr_moments<1>, grid, dim3(256), 0, st, *lrf_in, *src, units, (long long *)scratch);
  else
    hipLaunchKernelGGL(k_sgr_moments<2>, grid, dim3(256), 0, st, *lrf_in, *src, units, (long long *)scratch);
  hipLaunchKernelGGL(k_sgr_solve, dim3((n + 127) / 128), dim3(128), 0, st, units, (const long long *)scratch, n,
                     xqd_out);
#define R1_LRF_ERR(BPP, CH)                                                                                  \
  hipLaunchKernelGGL((k_sgr_unit_err<BPP, CH>), grid, dim3(256), 0, st, *lrf_in, *src, units,              \
                     (const int8_t *)xqd_out, xdec, ydec, scales, scale_stride, acc)
  if (lrf_in->bytes_per_px == 1) {
    if (is_chroma) R1_LRF_ERR(1, true); else R1_LRF_ERR(1, false);
  } else {
    if (is_chroma) R1_LRF_ERR(2, true); else R1_LRF_ERR(2, false);
  }
#undef R1_LRF_ERR
  hipLaunchKernelGGL(k_lrf_err_finish, dim3((n + 127) / 128), dim3(128), 0, st, acc, n, dist_scale,
                     (unsigned long long *)err_out);
  R1_HIP_CHECK(hipGetLastError());
  return R1_OK;
}

