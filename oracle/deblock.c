/*
 * deblock.c -- CPU oracle: deblocking filter and its level search
 * (SURVEY.md 8f "N3").  TEST INFRASTRUCTURE ONLY (see r1_oracle.h).
 *
 * Restates src/deblock.rs of the reference:
 *   deblock_adjusted_level 23-69     deblock_left / deblock_up 72-92
 *   deblock_size 95-130     deblock_level 133-143
 *   filter_narrow2_4 147-176, filter_narrow4_4 198-225, filter_wide6_4 248-259,
 *   filter_wide8_6 262-275, filter_wide14_12 288-306
 *   limit / blimit / thresh <-> level 334-361, nhev4 364, mask4 369, mask6 475,
 *   flat6 491, mask8 628, flat8 654, flat14_outer 830
 *   deblock_size{4,6,8,14}_inner 377, 499, 670, 846 (+ the v / h appliers)
 *   sse_size{4,6,8,14} 420, 552, 733, 934
 *   filter_v_edge 1099, sse_v_edge 1145, filter_h_edge 1195, sse_h_edge 1241
 *   deblock_plane 1294-1459 (in the reference's own interleaved edge order)
 *   sse_plane 1461-1542, sse_optimize 1553-1617
 *
 * Pinning: the reference has no vectors for this file.  The FILTER pass is a
 * normative AV1 decoder process, so tests/golden/gen_deblock_golden.py holds an
 * independent model in the specification's formulation (boolean limit /
 * blimit / thresh masks, the generic tap-window wide filter, all vertical edges
 * before all horizontal ones); this file matched its vectors.  The level
 * search (sse_*) is an encoder heuristic: pinned by a brute-force model that
 * filters every edge at every level 0..63 (same script).
 *
 * Block info wire format (r1o_deblock_block, one per 4x4 luma block, row-major
 * `blocks_stride` wide) -- what the reference reads from `Block`
 * (src/context/block_unit.rs): transform / block extents in 4x4 units as log2,
 * the chroma transform from BlockSize::largest_chroma_tx_size evaluated by the
 * host, skip / intra / mode-type flags, reference index, per-block deltas.
 */
#include <stdlib.h>
#include <string.h>

#include "r1_oracle.h"

#define MAX_LF 63

static int iabs(int v) { return v < 0 ? -v : v; }
static int imax(int a, int b) { return a > b ? a : b; }
static int imin(int a, int b) { return a < b ? a : b; }
static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* ---- filters: p[] holds the taps across the edge, p[n/2 - 1] = p0, p[n/2] = q0 ---- */
static void narrow2_4(int p1, int p0, int q0, int q1, int shift, int *o) {
  const int lo = -128 << shift, hi = (128 << shift) - 1, mx = (256 << shift) - 1;
  const int f0 = clampi(p1 - q1, lo, hi);
  const int f1 = clampi(f0 + 3 * (q0 - p0) + 4, lo, hi) >> 3;
  const int f2 = clampi(f0 + 3 * (q0 - p0) + 3, lo, hi) >> 3;
  o[0] = p1;
  o[1] = clampi(p0 + f2, 0, mx);
  o[2] = clampi(q0 - f1, 0, mx);
  o[3] = q1;
}

static void narrow4_4(int p1, int p0, int q0, int q1, int shift, int *o) {
  const int lo = -128 << shift, hi = (128 << shift) - 1, mx = (256 << shift) - 1;
  const int f1 = clampi(3 * (q0 - p0) + 4, lo, hi) >> 3;
  const int f2 = clampi(3 * (q0 - p0) + 3, lo, hi) >> 3;
  const int f3 = (f1 + 1) >> 1;
  o[0] = clampi(p1 + f3, 0, mx);
  o[1] = clampi(p0 + f2, 0, mx);
  o[2] = clampi(q0 - f1, 0, mx);
  o[3] = clampi(q1 - f3, 0, mx);
}

static void wide6_4(const int *t, int *o) { /* t = p2 p1 p0 q0 q1 q2 */
  const int p2 = t[0], p1 = t[1], p0 = t[2], q0 = t[3], q1 = t[4], q2 = t[5];
  o[0] = (p2 * 3 + p1 * 2 + p0 * 2 + q0 + 4) >> 3;
  o[1] = (p2 + p1 * 2 + p0 * 2 + q0 * 2 + q1 + 4) >> 3;
  o[2] = (p1 + p0 * 2 + q0 * 2 + q1 * 2 + q2 + 4) >> 3;
  o[3] = (p0 + q0 * 2 + q1 * 2 + q2 * 3 + 4) >> 3;
}

static void wide8_6(const int *t, int *o) { /* t = p3 .. q3 */
  const int p3 = t[0], p2 = t[1], p1 = t[2], p0 = t[3], q0 = t[4], q1 = t[5], q2 = t[6], q3 = t[7];
  o[0] = (p3 * 3 + p2 * 2 + p1 + p0 + q0 + 4) >> 3;
  o[1] = (p3 * 2 + p2 + p1 * 2 + p0 + q0 + q1 + 4) >> 3;
  o[2] = (p3 + p2 + p1 + p0 * 2 + q0 + q1 + q2 + 4) >> 3;
  o[3] = (p2 + p1 + p0 + q0 * 2 + q1 + q2 + q3 + 4) >> 3;
  o[4] = (p1 + p0 + q0 + q1 * 2 + q2 + q3 * 2 + 4) >> 3;
  o[5] = (p0 + q0 + q1 + q2 * 2 + q3 * 3 + 4) >> 3;
}

static void wide14_12(const int *t, int *o) { /* t = p6 .. q6 */
  const int p6 = t[0], p5 = t[1], p4 = t[2], p3 = t[3], p2 = t[4], p1 = t[5], p0 = t[6], q0 = t[7],
            q1 = t[8], q2 = t[9], q3 = t[10], q4 = t[11], q5 = t[12], q6 = t[13];
  o[0] = (p6 * 7 + p5 * 2 + p4 * 2 + p3 + p2 + p1 + p0 + q0 + 8) >> 4;
  o[1] = (p6 * 5 + p5 * 2 + p4 * 2 + p3 * 2 + p2 + p1 + p0 + q0 + q1 + 8) >> 4;
  o[2] = (p6 * 4 + p5 + p4 * 2 + p3 * 2 + p2 * 2 + p1 + p0 + q0 + q1 + q2 + 8) >> 4;
  o[3] = (p6 * 3 + p5 + p4 + p3 * 2 + p2 * 2 + p1 * 2 + p0 + q0 + q1 + q2 + q3 + 8) >> 4;
  o[4] = (p6 * 2 + p5 + p4 + p3 + p2 * 2 + p1 * 2 + p0 * 2 + q0 + q1 + q2 + q3 + q4 + 8) >> 4;
  o[5] = (p6 + p5 + p4 + p3 + p2 + p1 * 2 + p0 * 2 + q0 * 2 + q1 + q2 + q3 + q4 + q5 + 8) >> 4;
  o[6] = (p5 + p4 + p3 + p2 + p1 + p0 * 2 + q0 * 2 + q1 * 2 + q2 + q3 + q4 + q5 + q6 + 8) >> 4;
  o[7] = (p4 + p3 + p2 + p1 + p0 + q0 * 2 + q1 * 2 + q2 * 2 + q3 + q4 + q5 + q6 * 2 + 8) >> 4;
  o[8] = (p3 + p2 + p1 + p0 + q0 + q1 * 2 + q2 * 2 + q3 * 2 + q4 + q5 + q6 * 3 + 8) >> 4;
  o[9] = (p2 + p1 + p0 + q0 + q1 + q2 * 2 + q3 * 2 + q4 * 2 + q5 + q6 * 4 + 8) >> 4;
  o[10] = (p1 + p0 + q0 + q1 + q2 + q3 * 2 + q4 * 2 + q5 * 2 + q6 * 5 + 8) >> 4;
  o[11] = (p0 + q0 + q1 + q2 + q3 + q4 * 2 + q5 * 2 + q6 * 7 + 8) >> 4;
}

static int limit_to_level(int limit, int shift) { return (limit + (1 << shift) - 1) >> shift; }
static int blimit_to_level(int blimit, int shift) {
  return (((blimit + (1 << shift) - 1) >> shift) - 2) / 3;
}
static int thresh_to_level(int thresh, int shift) {
  return ((thresh + (1 << shift) - 1) >> shift) << 4;
}

/* taps centred on the edge: c[-1] = p0, c[0] = q0 */
static int nhev4(const int *c, int shift) {
  return thresh_to_level(imax(iabs(c[-2] - c[-1]), iabs(c[1] - c[0])), shift);
}
/* mask4 / mask6 / mask8: n = taps per side entering the limit test (2, 3, 4) */
static int maskn(const int *c, int n, int shift) {
  int m = 0;
  for (int i = 1; i < n; i++) m = imax(m, imax(iabs(c[-i - 1] - c[-i]), iabs(c[i] - c[i - 1])));
  return imax(limit_to_level(m, shift),
              blimit_to_level(iabs(c[-1] - c[0]) * 2 + iabs(c[-2] - c[1]) / 2, shift));
}
/* flat6 / flat8: taps 1 .. n-1 against p0 / q0; flat14_outer: taps lo .. hi */
static int flat_range(const int *c, int lo, int hi) {
  int m = 0;
  for (int i = lo; i <= hi; i++) m = imax(m, imax(iabs(c[-i - 1] - c[-1]), iabs(c[i] - c[0])));
  return m;
}

/* The candidate outputs of one line of an edge of `size` taps.  Outputs are
 * written for the positions the reference's deblock_size*_inner returns:
 * size 4 -> taps [0,4), 6 -> [1,5), 8 -> [1,7), 14 -> [1,13) of the line. */
enum { K_NONE, K_NARROW2, K_NARROW4, K_WIDE, K_WIDE14 };
static int out_len(int size) { return size == 4 ? 4 : size == 6 ? 4 : size == 8 ? 6 : 12; }
static int out_off(int size) { return size == 4 ? 0 : 1; }

static void apply_kind(const int *t, int size, int kind, int shift, int *o) {
  const int n = out_len(size), off = out_off(size), h = size / 2;
  for (int i = 0; i < n; i++) o[i] = t[off + i];
  int x[4];
  switch (kind) {
    case K_NARROW2:
    case K_NARROW4:
      if (kind == K_NARROW2) narrow2_4(t[h - 2], t[h - 1], t[h], t[h + 1], shift, x);
      else narrow4_4(t[h - 2], t[h - 1], t[h], t[h + 1], shift, x);
      for (int i = 0; i < 4; i++) o[h - 2 - off + i] = x[i];
      break;
    case K_WIDE:
      if (size == 6) wide6_4(t, o);
      else if (size == 8) wide8_6(t, o);
      else wide8_6(t + 3, o + 3); /* filter_wide8_12: p5 p4 p3 | wide8 | q3 q4 q5 */
      break;
    case K_WIDE14:
      wide14_12(t, o);
      break;
    default:
      break;
  }
}

/* deblock_size{4,6,8,14}_inner: returns 0 if the line is left alone */
static int filter_line(const int *t, int size, int level, int bd, int *o) {
  const int shift = bd - 8, h = size / 2;
  const int *c = t + h;
  const int mask = size == 4 ? maskn(c, 2, shift) : size == 6 ? maskn(c, 3, shift) : maskn(c, 4, shift);
  if (mask > level) return 0;
  const int flat = 1 << shift;
  int kind;
  if (size == 4) {
    kind = nhev4(c, shift) <= level ? K_NARROW4 : K_NARROW2;
  } else {
    const int inner = size == 6 ? flat_range(c, 1, 2) : flat_range(c, 1, 3);
    if (inner <= flat) {
      kind = (size == 14 && flat_range(c, 4, 6) <= flat) ? K_WIDE14 : K_WIDE;
    } else {
      kind = nhev4(c, shift) <= level ? K_NARROW4 : K_NARROW2;
    }
  }
  apply_kind(t, size, kind, shift, o);
  return 1;
}

static int64_t sse_of(const int *a, const int *b, int n) {
  int32_t s = 0; /* stride_sse sums in i32 before widening */
  for (int i = 0; i < n; i++) s += (a[i] - b[i]) * (a[i] - b[i]);
  return s;
}

/* sse_size{4,6,8,14} for one line: t = reconstructed taps, s = source taps */
static void sse_line(const int *t, const int *s, int size, int bd, int64_t *tally) {
  const int shift = bd - 8, h = size / 2, n = out_len(size), off = out_off(size);
  const int *c = t + h;
  const int flat = 1 << shift;
  int none[12], nar2[12], nar4[12], wide[12], wide14[12];
  apply_kind(t, size, K_NONE, shift, none);
  apply_kind(t, size, K_NARROW2, shift, nar2);
  apply_kind(t, size, K_NARROW4, shift, nar4);
  const int *a = s + off;
  int mask = size == 4 ? maskn(c, 2, shift) : size == 6 ? maskn(c, 3, shift) : maskn(c, 4, shift);
  mask = clampi(mask, 1, MAX_LF + 1);
  const int nhev = clampi(nhev4(c, shift), mask, MAX_LF + 1);
  const int64_t sse_none = sse_of(a, none, n);
  int flatp = 0, flat14p = 0;
  if (size != 4) {
    flatp = (size == 6 ? flat_range(c, 1, 2) : flat_range(c, 1, 3)) <= flat;
    apply_kind(t, size, K_WIDE, shift, wide);
    if (size == 14) {
      flat14p = flat_range(c, 4, 6) <= flat;
      apply_kind(t, size, K_WIDE14, shift, wide14);
    }
  }
  const int64_t sse_wide = (flatp && !flat14p && mask <= MAX_LF) ? sse_of(a, wide, n) : sse_none;
  const int64_t sse_wide14 = (flatp && flat14p && mask <= MAX_LF) ? sse_of(a, wide14, n) : sse_none;
  const int64_t sse_n2 = (!flatp && nhev != mask) ? sse_of(a, nar2, n) : sse_none;
  const int64_t sse_n4 = (!flatp && nhev <= MAX_LF) ? sse_of(a, nar4, n) : sse_none;
  tally[0] += sse_none;
  tally[mask] -= sse_none;
  if (flatp) {
    tally[mask] += flat14p ? sse_wide14 : sse_wide;
  } else {
    tally[mask] += sse_n2;
    tally[nhev] -= sse_n2;
    tally[nhev] += sse_n4;
  }
}

/* ---- block info ---- */
typedef struct {
  const r1o_deblock_block *b;
  int stride, cols, rows;
} blocks_t;

static const r1o_deblock_block *blk(const blocks_t *bl, int y, int x) {
  return &bl->b[(size_t)y * bl->stride + x];
}
static int tx_w_mi(const r1o_deblock_block *b, int pli) {
  return 1 << ((pli == 0 ? b->tx_log2 : b->uvtx_log2) & 7);
}
static int tx_h_mi(const r1o_deblock_block *b, int pli) {
  return 1 << (((pli == 0 ? b->tx_log2 : b->uvtx_log2) >> 3) & 7);
}
static int n4_w(const r1o_deblock_block *b) { return 1 << (b->n4_log2 & 7); }
static int n4_h(const r1o_deblock_block *b) { return 1 << ((b->n4_log2 >> 3) & 7); }
static int is_skip(const r1o_deblock_block *b) { return b->flags & 1; }
static int is_intra(const r1o_deblock_block *b) { return (b->flags >> 1) & 1; }

static int adjusted_level(const r1o_deblock_state *d, const r1o_deblock_block *b, int pli, int vertical) {
  const int idx = pli == 0 ? !vertical : pli + 1;
  int level;
  if (d->block_deltas_enabled) {
    const int delta = (d->block_delta_multi ? b->deltas[idx] : b->deltas[0]) << d->block_delta_shift;
    level = (uint8_t)clampi((int8_t)(delta + (int8_t)d->levels[idx]), 0, MAX_LF);
  } else {
    level = d->levels[idx];
  }
  if (d->deltas_enabled) {
    const int mode_type = (b->flags >> 2) & 1, ref = (b->flags >> 3) & 7;
    const int l5 = level >> 5;
    return clampi(level + (d->ref_deltas[ref] << l5) + (is_intra(b) ? 0 : d->mode_deltas[mode_type] << l5),
                  0, MAX_LF);
  }
  return level;
}

static int deblock_size(const r1o_deblock_block *b, const r1o_deblock_block *prev, int pli, int vertical,
                        int block_edge) {
  if (!(block_edge || !is_skip(b) || !is_skip(prev) || is_intra(b) || is_intra(prev))) return 0;
  const int tx_n = vertical ? tx_w_mi(b, pli) : tx_h_mi(b, pli);
  const int prev_n = vertical ? tx_w_mi(prev, pli) : tx_h_mi(prev, pli);
  return imin(pli == 0 ? 14 : 6, imin(tx_n, prev_n) << 2);
}

static int deblock_level(const r1o_deblock_state *d, const r1o_deblock_block *b, const r1o_deblock_block *prev,
                         int pli, int vertical) {
  const int level = adjusted_level(d, b, pli, vertical);
  return level == 0 ? adjusted_level(d, prev, pli, vertical) : level;
}

static int rd_px(const r1o_plane *p, int x, int y) {
  const size_t i = (size_t)(p->yorigin + y) * p->stride + p->xorigin + x;
  return p->bytes_per_px == 1 ? ((const uint8_t *)p->data)[i] : ((const uint16_t *)p->data)[i];
}
static void wr_px(const r1o_plane *p, int x, int y, int v) {
  const size_t i = (size_t)(p->yorigin + y) * p->stride + p->xorigin + x;
  if (p->bytes_per_px == 1) ((uint8_t *)p->data)[i] = (uint8_t)v;
  else ((uint16_t *)p->data)[i] = (uint16_t)v;
}

/* edge geometry shared by the filter and the sse walk; returns the filter size or 0.
 * size_vertical: the `vertical` argument handed to deblock_size (sse_h_edge of
 * the reference passes `true`, src/deblock.rs:1258 -- kept). */
static int edge_size(const blocks_t *bl, int bx, int by, int pli, int xdec, int ydec, int vertical,
                     int size_vertical, const r1o_deblock_block **b_out, const r1o_deblock_block **prev_out) {
  const r1o_deblock_block *b = blk(bl, by, bx);
  const int tx_edge = vertical ? (((bx >> xdec) & (tx_w_mi(b, pli) - 1)) == 0)
                               : (((by >> ydec) & (tx_h_mi(b, pli) - 1)) == 0);
  if (!tx_edge) return 0;
  const r1o_deblock_block *prev = vertical ? blk(bl, by | ydec, (bx | xdec) - (1 << xdec))
                                           : blk(bl, (by | ydec) - (1 << ydec), bx | xdec);
  const int block_edge = vertical ? ((bx & (n4_w(b) - 1)) == 0) : ((by & (n4_h(b) - 1)) == 0);
  *b_out = b;
  *prev_out = prev;
  return deblock_size(b, prev, pli, size_vertical, block_edge);
}

static void filter_edge(const r1o_deblock_state *d, const blocks_t *bl, int bx, int by, const r1o_plane *p,
                        int pli, int bd, int xdec, int ydec, int vertical) {
  const r1o_deblock_block *b, *prev;
  const int size = edge_size(bl, bx, by, pli, xdec, ydec, vertical, vertical, &b, &prev);
  if (!size) return;
  const int level = deblock_level(d, b, prev, pli, vertical);
  if (!level) return;
  const int px = (bx >> xdec) * 4, py = (by >> ydec) * 4, h = size / 2;
  for (int i = 0; i < 4; i++) {
    int t[14], o[12];
    for (int k = 0; k < size; k++) t[k] = vertical ? rd_px(p, px - h + k, py + i) : rd_px(p, px + i, py - h + k);
    if (filter_line(t, size, level, bd, o)) {
      const int n = out_len(size), off = out_off(size);
      for (int k = 0; k < n; k++) {
        if (vertical) wr_px(p, px - h + off + k, py + i, o[k]);
        else wr_px(p, px + i, py - h + off + k, o[k]);
      }
    }
  }
}

static void plane_extent(const blocks_t *bl, int crop_w, int crop_h, int xdec, int ydec, int *cols, int *rows) {
  *cols = ((imin(bl->cols, (crop_w + 3) >> 2) + ((1 << xdec) >> 1)) >> xdec) << xdec;
  *rows = ((imin(bl->rows, (crop_h + 3) >> 2) + ((1 << ydec) >> 1)) >> ydec) << ydec;
}

/* deblock_plane, edge for edge in the reference's order (vertical edges lead
 * the horizontal ones by one row and two columns) */
int r1o_deblock_plane(const r1o_deblock_state *d, const r1o_plane *p, int pli, int xdec, int ydec,
                      const r1o_deblock_block *blocks, int blocks_stride, int blocks_cols, int blocks_rows,
                      int crop_w, int crop_h, int bd) {
  if (xdec > 1 || ydec > 1 || pli < 0 || pli > 2) return -1;
  if (pli == 0 ? (d->levels[0] == 0 && d->levels[1] == 0) : d->levels[pli + 1] == 0) return 0;
  const blocks_t bl = { blocks, blocks_stride, blocks_cols, blocks_rows };
  int cols, rows;
  plane_extent(&bl, crop_w, crop_h, xdec, ydec, &cols, &rows);
  const int sx = 1 << xdec, sy = 1 << ydec;
#define V(x, y) filter_edge(d, &bl, x, y, p, pli, bd, xdec, ydec, 1)
#define H(x, y) filter_edge(d, &bl, x, y, p, pli, bd, xdec, ydec, 0)
  if (rows > 0) {
    for (int x = sx; x < cols; x += sx) V(x, 0);
    if (rows > sy)
      for (int x = sx; x < cols; x += sx) V(x, sy);
  }
  for (int y = 2 * sy; y < rows; y += sy) {
    if (cols > sx) V(sx, y);
    for (int x = 2 * sx; x < cols; x += sx) {
      V(x, y);
      H(x - 2 * sx, y - sy);
    }
    if (cols >= 2 * sx) H(cols - 2 * sx, y - sy);
    if (cols >= sx) H(cols - sx, y - sy);
  }
  if (rows > sy)
    for (int x = 0; x < cols; x += sx) H(x, rows - sy);
#undef V
#undef H
  return 0;
}

static void sse_edge(const blocks_t *bl, int bx, int by, const r1o_plane *rec, const r1o_plane *src, int pli,
                     int bd, int xdec, int ydec, int vertical, int64_t *tally) {
  const r1o_deblock_block *b, *prev;
  const int size = edge_size(bl, bx, by, pli, xdec, ydec, vertical, 1, &b, &prev);
  if (!size) return;
  const int px = (bx >> xdec) * 4, py = (by >> ydec) * 4, h = size / 2;
  for (int i = 0; i < 4; i++) {
    int t[14], s[14];
    for (int k = 0; k < size; k++) {
      t[k] = vertical ? rd_px(rec, px - h + k, py + i) : rd_px(rec, px + i, py - h + k);
      s[k] = vertical ? rd_px(src, px - h + k, py + i) : rd_px(src, px + i, py - h + k);
    }
    sse_line(t, s, size, bd, tally);
  }
}

/* sse_plane: v_tally / h_tally, MAX_LOOP_FILTER + 2 entries each, accumulated into */
int r1o_deblock_sse_plane(const r1o_plane *rec, const r1o_plane *src, int pli, int xdec, int ydec,
                          const r1o_deblock_block *blocks, int blocks_stride, int blocks_cols,
                          int blocks_rows, int crop_w, int crop_h, int bd, int64_t *v_tally,
                          int64_t *h_tally) {
  if (xdec > 1 || ydec > 1) return -1;
  const blocks_t bl = { blocks, blocks_stride, blocks_cols, blocks_rows };
  int cols, rows;
  plane_extent(&bl, crop_w, crop_h, xdec, ydec, &cols, &rows);
  const int sx = 1 << xdec, sy = 1 << ydec;
  for (int x = sx; x < cols; x += sx) sse_edge(&bl, x, 0, rec, src, pli, bd, xdec, ydec, 1, v_tally);
  for (int y = sy; y < rows; y += sy) {
    sse_edge(&bl, 0, y, rec, src, pli, bd, xdec, ydec, 0, h_tally);
    for (int x = sx; x < cols; x += sx) {
      sse_edge(&bl, x, y, rec, src, pli, bd, xdec, ydec, 1, v_tally);
      sse_edge(&bl, x, y, rec, src, pli, bd, xdec, ydec, 0, h_tally);
    }
  }
  return 0;
}

/* sse_optimize's tail for one plane (src/deblock.rs:1584-1614): prefix sums, then
 * the first minimum.  Luma: out[0] = vertical, out[1] = horizontal level;
 * chroma: out[0] = the common level. */
void r1o_deblock_pick_levels(int64_t *v_tally, int64_t *h_tally, int pli, uint8_t *out) {
  for (int i = 1; i <= MAX_LF; i++) {
    v_tally[i] += v_tally[i - 1];
    h_tally[i] += h_tally[i - 1];
  }
  if (pli == 0) {
    int bv = 0, bh = 0;
    for (int i = 1; i <= MAX_LF; i++) {
      if (v_tally[bv] > v_tally[i]) bv = i;
      if (h_tally[bh] > h_tally[i]) bh = i;
    }
    out[0] = (uint8_t)bv;
    out[1] = (uint8_t)bh;
  } else {
    int b = 0;
    for (int i = 1; i <= MAX_LF; i++)
      if (v_tally[b] + h_tally[b] > v_tally[i] + h_tally[i]) b = i;
    out[0] = (uint8_t)b;
  }
}
