// deblock.hip -- deblocking filter and its level search (SURVEY.md 8f "N3";
// reference src/deblock.rs: deblock_adjusted_level 23-69, deblock_size 95-130,
// deblock_level 133-143, the narrow / wide filters 147-306, the level
// thresholds nhev4 / mask4 / mask6 / flat6 / mask8 / flat8 / flat14_outer,
// deblock_size{4,6,8,14}_inner, sse_size{4,6,8,14}, filter_{v,h}_edge,
// sse_{v,h}_edge, deblock_plane 1294-1459, sse_plane 1461-1542, sse_optimize
// 1553-1617).
//
// The reference walks the edges of a plane in an interleaved order (vertical
// edges lead the horizontal ones by one row and two columns, deblock.rs:
// 1336-1430).  That order is equivalent to "all vertical edges, then all
// horizontal edges" (AV1 spec 7.14.2), and inside one pass the lines are
// independent: an edge of filter size s reads s/2 and changes < s/2 samples on
// each side, and s <= 4 * min(transform extent on either side), so two edges
// bounding the same transform block never touch the same samples.  Hence two
// launches per plane, one LANE PER EDGE LINE, in place:
//   * vertical pass: lane = (edge column, pixel row); consecutive lanes are
//     consecutive 4x4 columns of the same pixel row (overlapping 14-byte
//     windows of one row: the same cache lines);
//   * horizontal pass: lane = (pixel column, edge row); every tap load is one
//     coalesced row access across the wave.
// The level search (sse_plane) runs the same walk on the unfiltered
// reconstruction, forms the up to five candidate outputs of a line once, and
// adds their SSE differences at the line's two level thresholds into a
// 2 x 65-entry i64 tally: LDS atomics per workgroup, one global atomic per
// entry per workgroup.
#include "common.hpp"

namespace {

constexpr int MAX_LF = 63;

struct Geom {
  const R1DeblockBlock *blocks;
  int stride, cols, rows;   // blocks array stride; filtered extent in 4x4 luma units
  int pli, xdec, ydec, bd;
};

__device__ __forceinline__ int iabs(int v) { return v < 0 ? -v : v; }
__device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }
__device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

__device__ __forceinline__ R1DeblockBlock load_block(const Geom &g, int y, int x) {
  // 8-byte entries: one load
  const unsigned long long v = *(const unsigned long long *)(g.blocks + (size_t)y * g.stride + x);
  R1DeblockBlock b;
  __builtin_memcpy(&b, &v, 8);
  return b;
}
__device__ __forceinline__ int tx_mi(const R1DeblockBlock &b, int pli, bool width) {
  const int v = pli == 0 ? b.tx_log2 : b.uvtx_log2;
  return 1 << (width ? (v & 7) : ((v >> 3) & 7));
}

__device__ int adjusted_level(const R1DeblockState &d, const R1DeblockBlock &b, int pli, bool vertical) {
  const int idx = pli == 0 ? (vertical ? 0 : 1) : pli + 1;
  int level;
  if (d.block_deltas_enabled) {
    const int delta = (d.block_delta_multi ? b.deltas[idx] : b.deltas[0]) << d.block_delta_shift;
    level = (uint8_t)clampi((int8_t)(delta + (int8_t)d.levels[idx]), 0, MAX_LF);
  } else {
    level = d.levels[idx];
  }
  if (d.deltas_enabled) {
    const int mode_type = (b.flags >> 2) & 1, ref = (b.flags >> 3) & 7, intra = (b.flags >> 1) & 1;
    const int l5 = level >> 5;
    return clampi(level + (d.ref_deltas[ref] << l5) + (intra ? 0 : d.mode_deltas[mode_type] << l5), 0,
                  MAX_LF);
  }
  return level;
}

// tx edge / block edge / skip tests of filter_{v,h}_edge + deblock_size; returns
// the filter size (0: nothing to do).  size_vertical: the `vertical` handed to
// deblock_size -- sse_h_edge of the reference passes `true` (deblock.rs:1258).
__device__ int edge_size(const Geom &g, int bx, int by, bool vertical, bool size_vertical,
                         R1DeblockBlock &b, R1DeblockBlock &prev) {
  b = load_block(g, by, bx);
  const bool tx_edge = vertical ? (((bx >> g.xdec) & (tx_mi(b, g.pli, true) - 1)) == 0)
                                : (((by >> g.ydec) & (tx_mi(b, g.pli, false) - 1)) == 0);
  if (!tx_edge) return 0;
  prev = vertical ? load_block(g, by | g.ydec, (bx | g.xdec) - (1 << g.xdec))
                  : load_block(g, (by | g.ydec) - (1 << g.ydec), bx | g.xdec);
  const int n4 = vertical ? (1 << (b.n4_log2 & 7)) : (1 << ((b.n4_log2 >> 3) & 7));
  const bool block_edge = ((vertical ? bx : by) & (n4 - 1)) == 0;
  const bool skip = b.flags & 1, pskip = prev.flags & 1, intra = b.flags & 2, pintra = prev.flags & 2;
  if (!(block_edge || !skip || !pskip || intra || pintra)) return 0;
  const int n = imin(tx_mi(b, g.pli, size_vertical), tx_mi(prev, g.pli, size_vertical)) << 2;
  return imin(g.pli == 0 ? 14 : 6, n);
}

// One line across an edge: p[k] = p_k, q[k] = q_k (k < size / 2 valid).
struct Line {
  int p[7], q[7];
};

template <int BPP>
__device__ __forceinline__ void load_line(Line &l, const uint8_t *at_q0, long step, int h) {
#pragma unroll
  for (int k = 0; k < 7; k++) {
    l.p[k] = l.q[k] = 0;
    if (k < h) {
      l.q[k] = ld_px<BPP>(at_q0 + k * step);
      l.p[k] = ld_px<BPP>(at_q0 - (k + 1) * step);
    }
  }
}

__device__ __forceinline__ int limit_to_level(int v, int sh) { return (v + (1 << sh) - 1) >> sh; }
__device__ __forceinline__ int blimit_to_level(int v, int sh) { return (((v + (1 << sh) - 1) >> sh) - 2) / 3; }
__device__ __forceinline__ int thresh_to_level(int v, int sh) { return ((v + (1 << sh) - 1) >> sh) << 4; }

__device__ __forceinline__ int nhev4(const Line &l, int sh) {
  return thresh_to_level(imax(iabs(l.p[1] - l.p[0]), iabs(l.q[1] - l.q[0])), sh);
}
// mask4 / mask6 / mask8 (n = 2, 3, 4 taps per side)
__device__ __forceinline__ int mask_n(const Line &l, int n, int sh) {
  int m = 0;
#pragma unroll
  for (int i = 1; i < 4; i++)
    if (i < n) m = imax(m, imax(iabs(l.p[i] - l.p[i - 1]), iabs(l.q[i] - l.q[i - 1])));
  return imax(limit_to_level(m, sh),
              blimit_to_level(iabs(l.p[0] - l.q[0]) * 2 + iabs(l.p[1] - l.q[1]) / 2, sh));
}
// flat6 / flat8 (taps 1 .. hi) and flat14_outer (taps 4 .. 6)
__device__ __forceinline__ int flat_inner(const Line &l, int hi) {
  int m = 0;
#pragma unroll
  for (int i = 1; i < 4; i++)
    if (i <= hi) m = imax(m, imax(iabs(l.p[i] - l.p[0]), iabs(l.q[i] - l.q[0])));
  return m;
}
__device__ __forceinline__ int flat_outer(const Line &l) {
  int m = 0;
#pragma unroll
  for (int i = 4; i < 7; i++) m = imax(m, imax(iabs(l.p[i] - l.p[0]), iabs(l.q[i] - l.q[0])));
  return m;
}

// the candidate filters; outputs in the same p / q indexing (only the changed taps)
__device__ __forceinline__ void narrow(const Line &l, int sh, bool four, Line &o) {
  const int lo = -128 << sh, hi = (128 << sh) - 1, mx = (256 << sh) - 1;
  const int p1 = l.p[1], p0 = l.p[0], q0 = l.q[0], q1 = l.q[1];
  const int f0 = four ? 0 : clampi(p1 - q1, lo, hi);
  const int f1 = clampi(f0 + 3 * (q0 - p0) + 4, lo, hi) >> 3;
  const int f2 = clampi(f0 + 3 * (q0 - p0) + 3, lo, hi) >> 3;
  o = l;
  o.p[0] = clampi(p0 + f2, 0, mx);
  o.q[0] = clampi(q0 - f1, 0, mx);
  if (four) {
    const int f3 = (f1 + 1) >> 1;
    o.p[1] = clampi(p1 + f3, 0, mx);
    o.q[1] = clampi(q1 - f3, 0, mx);
  }
}
__device__ __forceinline__ void wide6(const Line &l, Line &o) {
  const int p2 = l.p[2], p1 = l.p[1], p0 = l.p[0], q0 = l.q[0], q1 = l.q[1], q2 = l.q[2];
  o = l;
  o.p[1] = (p2 * 3 + p1 * 2 + p0 * 2 + q0 + 4) >> 3;
  o.p[0] = (p2 + p1 * 2 + p0 * 2 + q0 * 2 + q1 + 4) >> 3;
  o.q[0] = (p1 + p0 * 2 + q0 * 2 + q1 * 2 + q2 + 4) >> 3;
  o.q[1] = (p0 + q0 * 2 + q1 * 2 + q2 * 3 + 4) >> 3;
}
__device__ __forceinline__ void wide8(const Line &l, Line &o) {
  const int p3 = l.p[3], p2 = l.p[2], p1 = l.p[1], p0 = l.p[0], q0 = l.q[0], q1 = l.q[1], q2 = l.q[2],
            q3 = l.q[3];
  o = l;
  o.p[2] = (p3 * 3 + p2 * 2 + p1 + p0 + q0 + 4) >> 3;
  o.p[1] = (p3 * 2 + p2 + p1 * 2 + p0 + q0 + q1 + 4) >> 3;
  o.p[0] = (p3 + p2 + p1 + p0 * 2 + q0 + q1 + q2 + 4) >> 3;
  o.q[0] = (p2 + p1 + p0 + q0 * 2 + q1 + q2 + q3 + 4) >> 3;
  o.q[1] = (p1 + p0 + q0 + q1 * 2 + q2 + q3 * 2 + 4) >> 3;
  o.q[2] = (p0 + q0 + q1 + q2 * 2 + q3 * 3 + 4) >> 3;
}
__device__ __forceinline__ void wide14(const Line &l, Line &o) {
  const int p6 = l.p[6], p5 = l.p[5], p4 = l.p[4], p3 = l.p[3], p2 = l.p[2], p1 = l.p[1], p0 = l.p[0];
  const int q0 = l.q[0], q1 = l.q[1], q2 = l.q[2], q3 = l.q[3], q4 = l.q[4], q5 = l.q[5], q6 = l.q[6];
  o = l;
  o.p[5] = (p6 * 7 + p5 * 2 + p4 * 2 + p3 + p2 + p1 + p0 + q0 + 8) >> 4;
  o.p[4] = (p6 * 5 + p5 * 2 + p4 * 2 + p3 * 2 + p2 + p1 + p0 + q0 + q1 + 8) >> 4;
  o.p[3] = (p6 * 4 + p5 + p4 * 2 + p3 * 2 + p2 * 2 + p1 + p0 + q0 + q1 + q2 + 8) >> 4;
  o.p[2] = (p6 * 3 + p5 + p4 + p3 * 2 + p2 * 2 + p1 * 2 + p0 + q0 + q1 + q2 + q3 + 8) >> 4;
  o.p[1] = (p6 * 2 + p5 + p4 + p3 + p2 * 2 + p1 * 2 + p0 * 2 + q0 + q1 + q2 + q3 + q4 + 8) >> 4;
  o.p[0] = (p6 + p5 + p4 + p3 + p2 + p1 * 2 + p0 * 2 + q0 * 2 + q1 + q2 + q3 + q4 + q5 + 8) >> 4;
  o.q[0] = (p5 + p4 + p3 + p2 + p1 + p0 * 2 + q0 * 2 + q1 * 2 + q2 + q3 + q4 + q5 + q6 + 8) >> 4;
  o.q[1] = (p4 + p3 + p2 + p1 + p0 + q0 * 2 + q1 * 2 + q2 * 2 + q3 + q4 + q5 + q6 * 2 + 8) >> 4;
  o.q[2] = (p3 + p2 + p1 + p0 + q0 + q1 * 2 + q2 * 2 + q3 * 2 + q4 + q5 + q6 * 3 + 8) >> 4;
  o.q[3] = (p2 + p1 + p0 + q0 + q1 + q2 * 2 + q3 * 2 + q4 * 2 + q5 + q6 * 4 + 8) >> 4;
  o.q[4] = (p1 + p0 + q0 + q1 + q2 + q3 * 2 + q4 * 2 + q5 * 2 + q6 * 5 + 8) >> 4;
  o.q[5] = (p0 + q0 + q1 + q2 + q3 + q4 * 2 + q5 * 2 + q6 * 7 + 8) >> 4;
}

__device__ __forceinline__ int mask_of(const Line &l, int size, int sh) {
  return mask_n(l, size == 4 ? 2 : (size == 6 ? 3 : 4), sh);
}

// lane -> (bx, by, line) of the vertical / horizontal pass; false if outside
__device__ __forceinline__ bool locate(const Geom &g, long long tid, bool vertical, int &bx, int &by,
                                       int &i) {
  const int ncx = g.cols >> g.xdec, ncy = g.rows >> g.ydec;   // 4x4 units of this plane
  if (vertical) {
    const int x = (int)(tid % ncx), yrow = (int)(tid / ncx);
    if (yrow >= ncy * 4 || x == 0) return false;
    bx = x << g.xdec;
    by = (yrow >> 2) << g.ydec;
    i = yrow & 3;
  } else {
    const int xcol = (int)(tid % (ncx * 4)), e = (int)(tid / (ncx * 4));
    if (e >= ncy || e == 0) return false;
    bx = (xcol >> 2) << g.xdec;
    by = e << g.ydec;
    i = xcol & 3;
  }
  return true;
}

template <int BPP>
__device__ __forceinline__ void deblock_line(const R1Plane &plane, const Geom &g, const R1DeblockState &d,
                                             bool vertical, long long tid) {
  int bx, by, i;
  if (!locate(g, tid, vertical, bx, by, i)) return;
  R1DeblockBlock b, prev;
  const int size = edge_size(g, bx, by, vertical, vertical, b, prev);
  if (!size) return;
  int level = adjusted_level(d, b, g.pli, vertical);
  if (level == 0) level = adjusted_level(d, prev, g.pli, vertical);
  if (level == 0) return;
  const int px = (bx >> g.xdec) * 4 + (vertical ? 0 : i), py = (by >> g.ydec) * 4 + (vertical ? i : 0);
  uint8_t *q0 = (uint8_t *)px_addr<BPP>(plane, px, py);
  const long step = vertical ? BPP : (long)plane.stride * BPP;
  const int h = size >> 1, sh = g.bd - 8;
  Line l, o;
  load_line<BPP>(l, q0, step, h);
  if (mask_of(l, size, sh) > level) return;
  const int flat = 1 << sh;
  if (size != 4 && flat_inner(l, size == 6 ? 2 : 3) <= flat) {
    if (size == 6) wide6(l, o);
    else if (size == 14 && flat_outer(l) <= flat) wide14(l, o);
    else wide8(l, o);
  } else {
    narrow(l, sh, nhev4(l, sh) <= level, o);
  }
  // taps the reference writes back: 4 -> all, 6 -> p1..q1, 8 -> p2..q2, 14 -> p5..q5
  const int nw = size == 4 ? 2 : h - 1;
#pragma unroll
  for (int k = 0; k < 6; k++) {
    if (k < nw) {
      if constexpr (BPP == 1) {
        q0[k * step] = (uint8_t)o.q[k];
        q0[-(k + 1) * step] = (uint8_t)o.p[k];
      } else {
        *(uint16_t *)(q0 + k * step) = (uint16_t)o.q[k];
        *(uint16_t *)(q0 - (k + 1) * step) = (uint16_t)o.p[k];
      }
    }
  }
}

template <int BPP>
__global__ __launch_bounds__(256) void k_deblock(R1Plane plane, Geom g, R1DeblockState d, int vertical_) {
  deblock_line<BPP>(plane, g, d, vertical_ != 0, (long long)blockIdx.x * 256 + threadIdx.x);
}

// deblock_filter_frame (deblock.rs:1544-1551): one pass of all three planes in one launch,
// blockIdx.y = plane (the planes are independent; a chroma plane has a quarter of the lines and
// its surplus workgroups leave at once)
struct DeblockPlanes { R1Plane p[3]; Geom g[3]; };
template <int BPP>
__global__ __launch_bounds__(256) void k_deblock_frame(DeblockPlanes s, R1DeblockState d, int vertical_,
                                                       unsigned active) {
  const int pl = blockIdx.y;
  if (!((active >> pl) & 1)) return;
  deblock_line<BPP>(s.p[pl], s.g[pl], d, vertical_ != 0, (long long)blockIdx.x * 256 + threadIdx.x);
}

__device__ __forceinline__ long long sse_lines(const Line &a, const Line &b, int nw) {
  int s = 0;   // stride_sse sums in i32
#pragma unroll
  for (int k = 0; k < 6; k++)
    if (k < nw) s += (a.p[k] - b.p[k]) * (a.p[k] - b.p[k]) + (a.q[k] - b.q[k]) * (a.q[k] - b.q[k]);
  return s;
}

template <int BPP>
__device__ __forceinline__ void deblock_sse_walk(const R1Plane &rec, const R1Plane &src, const Geom &g,
                                                 bool vertical, long long total,
                                                 long long *__restrict__ tally_out) {
  __shared__ unsigned long long tally[MAX_LF + 2];
  for (int k = threadIdx.x; k < MAX_LF + 2; k += 256) tally[k] = 0;
  __syncthreads();
  long long none_sum = 0;   // tally[0] gets every line's sse_none: one add per wave
  // grid-stride: a workgroup flushes its tally once, however many lines it walks
  for (long long tid = (long long)blockIdx.x * 256 + threadIdx.x; tid < total;
       tid += (long long)gridDim.x * 256) {
  int bx, by, i;
  R1DeblockBlock b, prev;
  int size = 0;
  if (locate(g, tid, vertical, bx, by, i)) size = edge_size(g, bx, by, vertical, true, b, prev);
  if (size) {
    const int px = (bx >> g.xdec) * 4 + (vertical ? 0 : i), py = (by >> g.ydec) * 4 + (vertical ? i : 0);
    const long step = vertical ? BPP : (long)rec.stride * BPP;
    const long sstep = vertical ? BPP : (long)src.stride * BPP;
    const int h = size >> 1, sh = g.bd - 8, nw = size == 4 ? 2 : h - 1;
    Line l, s, o;
    load_line<BPP>(l, px_addr<BPP>(rec, px, py), step, h);
    load_line<BPP>(s, px_addr<BPP>(src, px, py), sstep, h);
    const int flat = 1 << sh;
    const int mask = clampi(mask_of(l, size, sh), 1, MAX_LF + 1);
    const int nhev = clampi(nhev4(l, sh), mask, MAX_LF + 1);
    const bool flatp = size != 4 && flat_inner(l, size == 6 ? 2 : 3) <= flat;
    const bool flat14p = size == 14 && flat_outer(l) <= flat;
    const long long sse_none = sse_lines(s, l, nw);
    long long at_mask, at_nhev = 0;
    if (flatp) {
      long long w = sse_none;
      if (mask <= MAX_LF) {
        if (size == 6) wide6(l, o);
        else if (flat14p) wide14(l, o);
        else wide8(l, o);
        w = sse_lines(s, o, nw);
      }
      at_mask = w - sse_none;
    } else {
      long long n2 = sse_none, n4 = sse_none;
      if (nhev != mask) {
        narrow(l, sh, false, o);
        n2 = sse_lines(s, o, nw);
      }
      if (nhev <= MAX_LF) {
        narrow(l, sh, true, o);
        n4 = sse_lines(s, o, nw);
      }
      at_mask = n2 - sse_none;
      at_nhev = n4 - n2;
    }
    none_sum += sse_none;
    if (at_mask) atomicAdd(&tally[mask], (unsigned long long)at_mask);
    if (at_nhev) atomicAdd(&tally[nhev], (unsigned long long)at_nhev);
  }
  }
  {
    uint32_t lo = (uint32_t)none_sum, hi = (uint32_t)((unsigned long long)none_sum >> 32);
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
      const unsigned long long o = ((unsigned long long)(uint32_t)__shfl_xor((int)hi, m, 64) << 32) |
                                   (uint32_t)__shfl_xor((int)lo, m, 64);
      const unsigned long long t = (((unsigned long long)hi << 32) | lo) + o;
      lo = (uint32_t)t;
      hi = (uint32_t)(t >> 32);
    }
    if ((threadIdx.x & 63) == 0 && (lo | hi)) atomicAdd(&tally[0], ((unsigned long long)hi << 32) | lo);
  }
  __syncthreads();
  for (int k = threadIdx.x; k < MAX_LF + 2; k += 256)
    if (tally[k]) atomicAdd((unsigned long long *)tally_out + k, tally[k]);
}

template <int BPP>
__global__ __launch_bounds__(256) void k_deblock_sse(R1Plane rec, R1Plane src, Geom g, int vertical_,
                                                     long long total,
                                                     long long *__restrict__ tally_out) {
  deblock_sse_walk<BPP>(rec, src, g, vertical_ != 0, total, tally_out);
}

// the level search of all three planes and both edge directions in one launch:
// blockIdx.y = 2 * plane + (0 vertical, 1 horizontal); every (plane, direction) reads the same
// unfiltered reconstruction, so the six walks are independent
struct DeblockSsePlanes { R1Plane rec[3], src[3]; Geom g[3]; long long total[6]; long long *out[6]; };
template <int BPP>
__global__ __launch_bounds__(256) void k_deblock_sse_frame(DeblockSsePlanes s) {
  const int pl = blockIdx.y >> 1;
  deblock_sse_walk<BPP>(s.rec[pl], s.src[pl], s.g[pl], (blockIdx.y & 1) == 0, s.total[blockIdx.y],
                        s.out[blockIdx.y]);
}

int make_geom(Geom &g, const R1Plane *p, int pli, int xdec, int ydec, const R1DeblockBlock *blocks,
              int blocks_stride, int blocks_cols, int blocks_rows, int crop_w, int crop_h) {
  R1_REQUIRE(p && blocks);
  R1_REQUIRE(pli >= 0 && pli <= 2 && xdec >= 0 && xdec <= 1 && ydec >= 0 && ydec <= 1);
  R1_REQUIRE(pli != 0 || (xdec == 0 && ydec == 0));
  R1_REQUIRE(p->bytes_per_px == 1 || p->bytes_per_px == 2);
  R1_REQUIRE((p->bytes_per_px == 1) == (p->bit_depth == 8));
  R1_REQUIRE(blocks_cols > 0 && blocks_rows > 0 && blocks_stride >= blocks_cols && crop_w > 0 &&
             crop_h > 0);
  const int mc = (crop_w + 3) >> 2, mr = (crop_h + 3) >> 2;
  g.blocks = blocks;
  g.stride = blocks_stride;
  g.cols = (((blocks_cols < mc ? blocks_cols : mc) + ((1 << xdec) >> 1)) >> xdec) << xdec;
  g.rows = (((blocks_rows < mr ? blocks_rows : mr) + ((1 << ydec) >> 1)) >> ydec) << ydec;
  g.pli = pli;
  g.xdec = xdec;
  g.ydec = ydec;
  g.bd = p->bit_depth;
  // the rounding above can step one 4x4 column / row past the block array
  R1_REQUIRE(g.cols <= blocks_stride && g.rows <= blocks_rows);
  return R1_OK;
}

long long pass_threads(const Geom &g, bool vertical) {
  const long long ncx = g.cols >> g.xdec, ncy = g.rows >> g.ydec;
  return vertical ? ncx * ncy * 4 : ncx * 4 * ncy;
}

}  // namespace

extern "C" int r1_deblock_plane(r1_ctx *ctx, const R1DeblockState *state, const R1Plane *plane, int pli,
                                int xdec, int ydec, const R1DeblockBlock *blocks, int blocks_stride,
                                int blocks_cols, int blocks_rows, int crop_w, int crop_h,
                                void *stream) {
  R1_REQUIRE(ctx && state);
  Geom g;
  const int rc = make_geom(g, plane, pli, xdec, ydec, blocks, blocks_stride, blocks_cols, blocks_rows,
                           crop_w, crop_h);
  if (rc != R1_OK) return rc;
  // deblock_plane's early outs (deblock.rs:1302-1319)
  if (pli == 0 ? (state->levels[0] == 0 && state->levels[1] == 0) : state->levels[pli + 1] == 0)
    return R1_OK;
  hipStream_t st = (hipStream_t)stream;
  for (int pass = 0; pass < 2; pass++) {
    const bool vertical = pass == 0;
    const long long n = pass_threads(g, vertical);
    if (n <= 0) continue;
    const unsigned grid = (unsigned)((n + 255) / 256);
    if (plane->bytes_per_px == 1)
      hipLaunchKernelGGL(k_deblock<1>, dim3(grid), dim3(256), 0, st, *plane, g, *state, (int)vertical);
    else
      hipLaunchKernelGGL(k_deblock<2>, dim3(grid), dim3(256), 0, st, *plane, g, *state, (int)vertical);
  }
  R1_HIP_CHECK(hipGetLastError());
  return R1_OK;
}

extern "C" int r1_deblock_sse_plane(r1_ctx *ctx, const R1Plane *rec, const R1Plane *src, int pli,
                                    int xdec, int ydec, const R1DeblockBlock *blocks,
                                    int blocks_stride, int blocks_cols, int blocks_rows, int crop_w,
                                    int crop_h, int64_t *v_tally, int64_t *h_tally, void *stream) {
  R1_REQUIRE(ctx && src && v_tally && h_tally);
  Geom g;
  const int rc = make_geom(g, rec, pli, xdec, ydec, blocks, blocks_stride, blocks_cols, blocks_rows,
                           crop_w, crop_h);
  if (rc != R1_OK) return rc;
  R1_REQUIRE(src->bytes_per_px == rec->bytes_per_px && src->bit_depth == rec->bit_depth);
  hipStream_t st = (hipStream_t)stream;
  for (int pass = 0; pass < 2; pass++) {
    const bool vertical = pass == 0;
    const long long n = pass_threads(g, vertical);
    if (n <= 0) continue;
    unsigned grid = (unsigned)((n + 255) / 256);
    grid = grid > 1024 ? 1024 : grid;   // 4 workgroups per CU
    long long *out = (long long *)(vertical ? v_tally : h_tally);
    if (rec->bytes_per_px == 1)
      hipLaunchKernelGGL(k_deblock_sse<1>, dim3(grid), dim3(256), 0, st, *rec, *src, g, (int)vertical, n,
                         out);
    else
      hipLaunchKernelGGL(k_deblock_sse<2>, dim3(grid), dim3(256), 0, st, *rec, *src, g, (int)vertical, n,
                         out);
  }
  R1_HIP_CHECK(hipGetLastError());
  return R1_OK;
}

extern "C" int r1_deblock_frame(r1_ctx *ctx, const R1DeblockState *state, const R1Plane *planes,
                                int xdec, int ydec, const R1DeblockBlock *blocks, int blocks_stride,
                                int blocks_cols, int blocks_rows, int crop_w, int crop_h,
                                void *stream) {
  R1_REQUIRE(ctx && state && planes);
  DeblockPlanes s;
  unsigned active = 0;
  for (int pli = 0; pli < 3; pli++) {
    const int rc = make_geom(s.g[pli], planes + pli, pli, pli ? xdec : 0, pli ? ydec : 0, blocks,
                             blocks_stride, blocks_cols, blocks_rows, crop_w, crop_h);
    if (rc != R1_OK) return rc;
    R1_REQUIRE(planes[pli].bytes_per_px == planes[0].bytes_per_px);
    s.p[pli] = planes[pli];
    // deblock_plane's early outs (deblock.rs:1302-1319)
    const bool off = pli == 0 ? (state->levels[0] == 0 && state->levels[1] == 0)
                              : state->levels[pli + 1] == 0;
    if (!off) active |= 1u << pli;
  }
  if (!active) return R1_OK;
  hipStream_t st = (hipStream_t)stream;
  for (int pass = 0; pass < 2; pass++) {
    const bool vertical = pass == 0;
    long long n = 0;
    for (int pli = 0; pli < 3; pli++) {
      const long long np = pass_threads(s.g[pli], vertical);
      n = np > n ? np : n;
    }
    if (n <= 0) continue;
    const dim3 grid((unsigned)((n + 255) / 256), 3);
    if (planes[0].bytes_per_px == 1)
      hipLaunchKernelGGL(k_deblock_frame<1>, grid, dim3(256), 0, st, s, *state, (int)vertical, active);
    else
      hipLaunchKernelGGL(k_deblock_frame<2>, grid, dim3(256), 0, st, s, *state, (int)vertical, active);
  }
  R1_HIP_CHECK(hipGetLastError());
  return R1_OK;
}

extern "C" int r1_deblock_sse_frame(r1_ctx *ctx, const R1Plane *rec, const R1Plane *src, int xdec,
                                    int ydec, const R1DeblockBlock *blocks, int blocks_stride,
                                    int blocks_cols, int blocks_rows, int crop_w, int crop_h,
                                    int64_t *tallies, void *stream) {
  R1_REQUIRE(ctx && rec && src && tallies);
  DeblockSsePlanes s;
  long long nmax = 0;
  for (int pli = 0; pli < 3; pli++) {
    const int rc = make_geom(s.g[pli], rec + pli, pli, pli ? xdec : 0, pli ? ydec : 0, blocks,
                             blocks_stride, blocks_cols, blocks_rows, crop_w, crop_h);
    if (rc != R1_OK) return rc;
    R1_REQUIRE(src[pli].bytes_per_px == rec[pli].bytes_per_px && src[pli].bit_depth == rec[pli].bit_depth &&
               rec[pli].bytes_per_px == rec[0].bytes_per_px);
    s.rec[pli] = rec[pli];
    s.src[pli] = src[pli];
    for (int dir = 0; dir < 2; dir++) {
      s.total[2 * pli + dir] = pass_threads(s.g[pli], dir == 0);
      s.out[2 * pli + dir] = (long long *)tallies + (size_t)(2 * pli + dir) * (MAX_LF + 2);
      nmax = s.total[2 * pli + dir] > nmax ? s.total[2 * pli + dir] : nmax;
    }
  }
  if (nmax <= 0) return R1_OK;
  unsigned gx = (unsigned)((nmax + 255) / 256);
  gx = gx > 512 ? 512 : gx;   // 6 x 512 workgroups, grid-stride inside
  hipStream_t st = (hipStream_t)stream;
  if (rec[0].bytes_per_px == 1)
    hipLaunchKernelGGL(k_deblock_sse_frame<1>, dim3(gx, 6), dim3(256), 0, st, s);
  else
    hipLaunchKernelGGL(k_deblock_sse_frame<2>, dim3(gx, 6), dim3(256), 0, st, s);
  R1_HIP_CHECK(hipGetLastError());
  return R1_OK;
}

// sse_optimize's tail for one plane (deblock.rs:1584-1614), host arithmetic
extern "C" int r1_deblock_pick_levels(const int64_t *v_tally, const int64_t *h_tally, int pli,
                                      uint8_t *levels_out) {
  R1_REQUIRE(v_tally && h_tally && levels_out && pli >= 0 && pli <= 2);
  int64_t v[MAX_LF + 1], h[MAX_LF + 1];
  for (int i = 0; i <= MAX_LF; i++) {
    v[i] = v_tally[i] + (i ? v[i - 1] : 0);
    h[i] = h_tally[i] + (i ? h[i - 1] : 0);
  }
  if (pli == 0) {
    int bv = 0, bh = 0;
    for (int i = 1; i <= MAX_LF; i++) {
      if (v[bv] > v[i]) bv = i;
      if (h[bh] > h[i]) bh = i;
    }
    levels_out[0] = (uint8_t)bv;
    levels_out[1] = (uint8_t)bh;
  } else {
    int b = 0;
    for (int i = 1; i <= MAX_LF; i++)
      if (v[b] + h[b] > v[i] + h[i]) b = i;
    levels_out[0] = (uint8_t)b;
  }
  return R1_OK;
}
