/*
 * me.c -- CPU oracle: hierarchical motion estimation of one tile against one
 * reference (SURVEY.md 8f "N2").  TEST INFRASTRUCTURE ONLY (see r1_oracle.h).
 *
 * Restates, function by function, src/me.rs of the reference:
 *   estimate_tile_motion 153-218     estimate_sb_motion 220-282
 *   refine_subsampled_sb_motion 284-322     save_me_stats 324-337
 *   get_mv_range 339-362     get_subset_predictors 386-534
 *   estimate_motion 536-632 (pmv = None: no sub-pel step)
 *   refine_subsampled_motion_estimate 634-691     full_pixel_me 693-855
 *   get_best_predictor 884-911     fullpel_diamond_search 955-1000
 *   hexagon_search 1055-1141     uneven_multi_hex_search 1170-1309
 *   get_fullpel_mv_rd 1386-1409     compute_mv_rd 1445-1462
 *   full_search 1464-1510     get_mv_rate 1512-1523
 *   sub_pixel_me 857-882     subpel_diamond_search 1311-1383
 *   get_subpel_mv_rd 1411-1443 (+ PredictionMode::get_mv_params src/predict.rs:284-297)
 * MotionVector arithmetic: src/mc.rs:28-100 (i16 components, 1/8 pel).
 *
 * Pinning: the reference holds no test vectors for me.rs (no #[test] in the file) and cannot be
 * built here (Rust); the vectors come from EXECUTING its source text (tools/rustlite):
 * tests/golden/me_ref.npz (gen_me_ref.py -- 14 tile searches, 37 block searches with the sub-pel
 * diamond), reproduced entry by entry in tests/test_oracle_me_ref.py.  The structural properties
 * in tests/test_oracle_me.py (zero-motion and pure-translation recovery, monotone cost, range
 * clamping) and the second restatement tests/me_model.py stay as independent checks.
 *
 * Planes: index 0 = full resolution, 1 = half, 2 = quarter (FrameState
 * input_hres / input_qres, src/encoder.rs:412-413, produced by v_frame's
 * Plane::downsampled; they are INPUTS here).  Stats: FrameMEStats
 * (src/me.rs:31-79), one MEStats per 4x4 luma block, row-major, `cols` wide.
 */
#include <limits.h>
#include <stdlib.h>
#include <string.h>

#include "r1_oracle.h"

#define MI 4
#define SB 64
#define MV_LOW (-(1 << 14))
#define MV_UPP (1 << 14)

typedef struct { int16_t row, col; } mv_t;
typedef struct { mv_t mv; uint64_t cost; uint32_t sad; } msr_t; /* MotionSearchResult */

typedef struct {
  const r1o_plane *org, *ref;
  int hbd, po_x, po_y, w, h, allow_hp;
  uint32_t lambda;
  mv_t pmv[2];
  int mvx_min, mvx_max, mvy_min, mvy_max;
} mectx;

static msr_t msr_empty(void) {
  msr_t r = { { 0, 0 }, UINT64_MAX, UINT32_MAX };
  return r;
}

static const uint8_t *px(const r1o_plane *p, int x, int y) {
  return (const uint8_t *)p->data +
         ((size_t)(p->yorigin + y) * p->stride + (size_t)(p->xorigin + x)) * p->bytes_per_px;
}

static int ilog16(int v) { /* ILog::ilog on i16: bits needed, 0 for v <= 0 */
  int n = 0;
  while (v > 0) { n++; v >>= 1; }
  return n;
}

static uint32_t diff_to_rate(int16_t diff, int allow_hp) {
  const int d = allow_hp ? diff : (diff >> 1);
  return 2u * (uint32_t)ilog16(d < 0 ? -d : d);
}

static uint32_t get_mv_rate(mv_t a, mv_t b, int allow_hp) {
  return diff_to_rate((int16_t)(a.row - b.row), allow_hp) +
         diff_to_rate((int16_t)(a.col - b.col), allow_hp);
}

/* compute_mv_rd with the SAD of the block at an integer position */
static void mv_rd_at(const mectx *c, mv_t cand, int rx, int ry, uint64_t *cost, uint32_t *sad) {
  const uint32_t s = r1o_get_sad(px(c->org, c->po_x, c->po_y), c->org->stride, px(c->ref, rx, ry),
                                 c->ref->stride, c->w, c->h, c->hbd);
  const uint32_t r1 = get_mv_rate(cand, c->pmv[0], c->allow_hp);
  const uint32_t r2 = get_mv_rate(cand, c->pmv[1], c->allow_hp);
  const uint32_t rate = r1 < r2 + 1 ? r1 : r2 + 1;
  *cost = 256ull * s + (uint64_t)rate * c->lambda;
  *sad = s;
}

static void fullpel_mv_rd(const mectx *c, mv_t cand, uint64_t *cost, uint32_t *sad) {
  if (cand.col < c->mvx_min || cand.col > c->mvx_max || cand.row < c->mvy_min ||
      cand.row > c->mvy_max) {
    *cost = UINT64_MAX;
    *sad = UINT32_MAX;
    return;
  }
  mv_rd_at(c, cand, c->po_x + cand.col / 8, c->po_y + cand.row / 8, cost, sad);
}

static mv_t mv_add(mv_t a, int drow, int dcol) {
  mv_t r = { (int16_t)(a.row + drow), (int16_t)(a.col + dcol) };
  return r;
}

/* `if rd.cost < best.rd.cost { best = cand }` */
static void take_if_better(const mectx *c, mv_t cand, msr_t *best) {
  uint64_t cost;
  uint32_t sad;
  fullpel_mv_rd(c, cand, &cost, &sad);
  if (cost < best->cost) {
    best->mv = cand;
    best->cost = cost;
    best->sad = sad;
  }
}

static msr_t get_best_predictor(const mectx *c, const mv_t *pred, int n) {
  msr_t best = msr_empty();
  for (int i = 0; i < n; i++) take_if_better(c, pred[i], &best);
  return best;
}

static const int8_t DIAMOND[4][2] = { { 1, 0 }, { 0, 1 }, { -1, 0 }, { 0, -1 } }; /* (row, col) */

static void fullpel_diamond_search(const mectx *c, msr_t *cur) {
  int radius_log2 = 1;
  for (;;) {
    msr_t best = msr_empty();
    for (int i = 0; i < 4; i++)
      take_if_better(c, mv_add(cur->mv, (DIAMOND[i][0] * 8) << radius_log2,
                               (DIAMOND[i][1] * 8) << radius_log2), &best);
    if (cur->cost <= best.cost) {
      if (radius_log2 == 0) break;
      radius_log2--;
    } else {
      *cur = best;
    }
  }
}

static const int8_t HEXAGON[6][2] = { { -2, 0 }, { -1, 2 }, { 1, 2 }, { 2, 0 }, { 1, -2 }, { -1, -2 } };
static const int8_t SQUARE[8][2] = { { 1, -1 }, { 1, 0 }, { 1, 1 }, { 0, -1 },
                                     { 0, 1 }, { -1, -1 }, { -1, 0 }, { -1, 1 } };

static void hexagon_search(const mectx *c, msr_t *cur) {
  int best_idx = 0;
  msr_t best = msr_empty();
  for (int i = 0; i < 6; i++) {
    uint64_t cost;
    uint32_t sad;
    const mv_t cand = mv_add(cur->mv, HEXAGON[i][0] * 8, HEXAGON[i][1] * 8);
    fullpel_mv_rd(c, cand, &cost, &sad);
    if (cost < best.cost) {
      best_idx = i;
      best.mv = cand; best.cost = cost; best.sad = sad;
    }
  }
  while (best.cost < cur->cost) {
    *cur = best;
    best = msr_empty();
    const int center_idx = best_idx;
    for (int off = 5; off <= 7; off++) {
      const int i = (center_idx + off) % 6;
      uint64_t cost;
      uint32_t sad;
      const mv_t cand = mv_add(cur->mv, HEXAGON[i][0] * 8, HEXAGON[i][1] * 8);
      fullpel_mv_rd(c, cand, &cost, &sad);
      if (cost < best.cost) {
        best_idx = i;
        best.mv = cand; best.cost = cost; best.sad = sad;
      }
    }
  }
  best = msr_empty();
  for (int i = 0; i < 8; i++)
    take_if_better(c, mv_add(cur->mv, SQUARE[i][0] * 8, SQUARE[i][1] * 8), &best);
  if (best.cost < cur->cost) *cur = best;
}

static const int8_t UMH[16][2] = { /* (row, col) */
  { 4, -2 }, { 4, -1 }, { 4, 0 }, { 4, 1 }, { 4, 2 }, { 2, 3 }, { 0, 4 }, { -2, 3 },
  { -4, 2 }, { -4, 1 }, { -4, 0 }, { -4, -1 }, { -4, -2 }, { -2, 3 }, { 0, -4 }, { 2, -3 } };

static void uneven_multi_hex_search(const mectx *c, msr_t *cur, int me_range) {
  mv_t center = cur->mv;
  /* the "horizontal" line of the reference steps the ROW (me.rs:1196-1199) */
  for (int i = 1; i <= me_range; i += 2)
    for (int s = -1; s <= 1; s += 2) take_if_better(c, mv_add(center, s * 8 * i, 0), cur);
  for (int i = 1; i <= (me_range >> 1); i += 2)
    for (int s = -1; s <= 1; s += 2) take_if_better(c, mv_add(center, 0, s * 8 * i), cur);
  /* 5x5 around the best: offsets are NOT scaled to full pel (me.rs:1241-1247) */
  center = cur->mv;
  for (int row = -2; row <= 2; row++)
    for (int col = -2; col <= 2; col++) {
      if (row == 0 && col == 0) continue;
      take_if_better(c, mv_add(center, row, col), cur);
    }
  center = cur->mv;
  for (int i = 1; i <= (me_range >> 2); i++)
    for (int k = 0; k < 16; k++)
      take_if_better(c, mv_add(center, UMH[k][0] * 8 * i, UMH[k][1] * 8 * i), cur);
  hexagon_search(c, cur);
}

/* full_search: windows of the search region, rows outer, every `step`-th */
static msr_t full_search(const mectx *c, int x_lo, int x_hi, int y_lo, int y_hi, int step) {
  msr_t best = msr_empty();
  for (int y = y_lo; y <= y_hi; y += step)
    for (int x = x_lo; x <= x_hi; x += step) {
      const mv_t mv = { (int16_t)(8 * (int16_t)(y - c->po_y)), (int16_t)(8 * (int16_t)(x - c->po_x)) };
      uint64_t cost;
      uint32_t sad;
      mv_rd_at(c, mv, x, y, &cost, &sad);
      if (cost < best.cost) {
        best.mv = mv; best.cost = cost; best.sad = sad;
      }
    }
  return best;
}

static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }
static int iclamp(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

static void get_mv_range(int w_in_b, int h_in_b, int bx, int by, int blk_w, int blk_h, int *r) {
  const int border_w = 128 + blk_w * 8, border_h = 128 + blk_h * 8;
  const int x_min = -bx * (8 * MI) - border_w;
  const int x_max = ((w_in_b - bx) - blk_w / MI) * (8 * MI) + border_w;
  const int y_min = -by * (8 * MI) - border_h;
  const int y_max = ((h_in_b - by) - blk_h / MI) * (8 * MI) + border_h;
  r[0] = imax(x_min, MV_LOW + 1);
  r[1] = imin(x_max, MV_UPP - 1);
  r[2] = imax(y_min, MV_LOW + 1);
  r[3] = imin(y_max, MV_UPP - 1);
}

typedef struct {
  uint32_t min_sad;
  int has_median;
  mv_t median;
  int nb, nc;
  mv_t b[5], c[5];
} subsets_t;

typedef struct {
  const r1o_me_params *p;
  r1o_me_stats *stats;       /* frame array */
  const r1o_me_stats *prev;  /* previous frame's array for this reference, or NULL */
  int tx, ty, tcols, trows;  /* tile origin / size in 4x4 units */
} tilectx;

static r1o_me_stats *tstat(const tilectx *t, int y, int x) {
  return &t->stats[(size_t)(t->ty + y) * t->p->stats_cols + t->tx + x];
}

static mv_t process_cand(const r1o_me_stats *s, const int *rng, uint32_t *min_sad) {
  if (s->normalized_sad < *min_sad) *min_sad = s->normalized_sad;
  const int col = (s->col / 8) * 8, row = (s->row / 8) * 8; /* quantize_to_fullpel */
  mv_t r = { (int16_t)iclamp(row, rng[2], rng[3]), (int16_t)iclamp(col, rng[0], rng[1]) };
  return r;
}

static int cmp16(const void *a, const void *b) { return *(const int16_t *)a - *(const int16_t *)b; }

/* corner: 0 = INIT, else CORNER { right = bit 1, bottom = bit 2 } (bit 0 set) */
static subsets_t get_subset_predictors(const tilectx *t, int bx, int by, int pix_w, int pix_h,
                                       const int *rng, int corner, int ssdec) {
  subsets_t s;
  memset(&s, 0, sizeof s);
  uint32_t min_sad = UINT32_MAX;
  const int w = ((pix_w << ssdec) + MI - 1) >> 2, h = ((pix_h << ssdec) + MI - 1) >> 2;
  const int half_w = imin(w >> 1, t->tcols - 1 - bx), half_h = imin(h >> 1, t->trows - 1 - by);
  if (bx > 0) s.b[s.nb++] = process_cand(tstat(t, by + half_h, bx - 1), rng, &min_sad);
  if (by > 0) s.b[s.nb++] = process_cand(tstat(t, by - 1, bx + half_w), rng, &min_sad);
  if (corner && (corner & 2) && bx + w < t->tcols)
    s.b[s.nb++] = process_cand(tstat(t, by + half_h, bx + w), rng, &min_sad);
  if (corner && (corner & 4) && by + h < t->trows)
    s.b[s.nb++] = process_cand(tstat(t, by + h, bx + half_w), rng, &min_sad);
  if (corner) {
    s.has_median = 1;
    s.median = process_cand(tstat(t, by + half_h, bx + half_w), rng, &min_sad);
  } else if (s.nb == 3) {
    int16_t rows[3], cols[3];
    for (int i = 0; i < 3; i++) { rows[i] = s.b[i].row; cols[i] = s.b[i].col; }
    qsort(rows, 3, sizeof rows[0], cmp16);
    qsort(cols, 3, sizeof cols[0], cmp16);
    s.has_median = 1;
    s.median.row = rows[1];
    s.median.col = cols[1];
  }
  s.b[s.nb].row = 0;
  s.b[s.nb].col = 0;
  s.nb++;
  if (t->prev) {
    const int fx = t->tx + bx, fy = t->ty + by, pc = t->p->stats_cols, pr = t->p->stats_rows;
    const int hw = imin(w >> 1, pc - 1 - fx), hh = imin(h >> 1, pr - 1 - fy);
#define PREV(y, x) (&t->prev[(size_t)(y) * pc + (x)])
    if (fx > 0) s.c[s.nc++] = process_cand(PREV(fy + hh, fx - 1), rng, &min_sad);
    if (fy > 0) s.c[s.nc++] = process_cand(PREV(fy - 1, fx + hw), rng, &min_sad);
    if (fx + w < pc) s.c[s.nc++] = process_cand(PREV(fy + hh, fx + w), rng, &min_sad);
    if (fy + h < pr) s.c[s.nc++] = process_cand(PREV(fy + h, fx + hw), rng, &min_sad);
    s.c[s.nc++] = process_cand(PREV(fy + hh, fx + hw), rng, &min_sad);
#undef PREV
  }
  s.min_sad = (uint32_t)(((uint64_t)min_sad * (uint64_t)(pix_w * pix_h)) >> 14);
  if (s.has_median) { s.median.col >>= ssdec; s.median.row >>= ssdec; }
  for (int i = 0; i < s.nb; i++) { s.b[i].col >>= ssdec; s.b[i].row >>= ssdec; }
  for (int i = 0; i < s.nc; i++) { s.c[i].col >>= ssdec; s.c[i].row >>= ssdec; }
  return s;
}

static void try_cands(const mectx *c, const mv_t *pred, int n, msr_t *best) {
  msr_t r = get_best_predictor(c, pred, n);
  fullpel_diamond_search(c, &r);
  if (r.cost < best->cost) *best = r;
}

static msr_t full_pixel_me(const tilectx *t, const mectx *c, int bx, int by, const int *rng,
                           int corner, int extensive, int ssdec) {
  const subsets_t s = get_subset_predictors(t, bx, by, c->w, c->h, rng, corner, ssdec);
  msr_t best = msr_empty();
  if (!extensive) {
    mv_t all[11];
    int n = 0;
    if (s.has_median) all[n++] = s.median;
    for (int i = 0; i < s.nb; i++) all[n++] = s.b[i];
    for (int i = 0; i < s.nc; i++) all[n++] = s.c[i];
    try_cands(c, all, n, &best);
    return best;
  }
  const uint32_t thresh = (uint32_t)((float)s.min_sad * 1.2f) +
                          ((uint32_t)(c->w * c->h) << (t->p->bit_depth - 8));
  if (s.has_median) {
    try_cands(c, &s.median, 1, &best);
    if (best.sad < thresh) return best;
  }
  try_cands(c, s.b, s.nb, &best);
  if (best.sad < thresh) return best;
  try_cands(c, s.c, s.nc, &best);
  if (best.sad < thresh) return best;
  uneven_multi_hex_search(c, &best, 24);
  if (!t->p->allow_full_search || best.sad < thresh) return best;
  {
    const int range_x = (192 * t->p->me_range_scale) >> ssdec;
    const int range_y = (64 * t->p->me_range_scale) >> ssdec;
    const int x_lo = c->po_x + imax(-range_x, c->mvx_min / 8);
    const int x_hi = c->po_x + imin(range_x, c->mvx_max / 8);
    const int y_lo = c->po_y + imax(-range_y, c->mvy_min / 8);
    const int y_hi = c->po_y + imin(range_y, c->mvy_max / 8);
    mectx z = *c; /* pmv = [0, 0] for the full search */
    memset(z.pmv, 0, sizeof z.pmv);
    const msr_t r = full_search(&z, x_lo, x_hi, y_lo, y_hi, 4 >> ssdec);
    return r.cost < best.cost ? r : best;
  }
}

static void block_ctx(const tilectx *t, const r1o_plane *org3, const r1o_plane *ref3, int bx, int by,
                      int w, int h, int ssdec, mectx *c, int *rng) {
  const int fbx = t->tx + bx, fby = t->ty + by;
  get_mv_range(t->p->w_in_b, t->p->h_in_b, fbx, fby, w << ssdec, h << ssdec, rng);
  /* estimate_motion shadows the range with its decimated version (me.rs:563-564)
   * before full_pixel_me: the predictors are clamped against that one too */
  for (int i = 0; i < 4; i++) rng[i] >>= ssdec;
  memset(c, 0, sizeof *c);
  c->org = &org3[ssdec];
  c->ref = &ref3[ssdec];
  c->hbd = c->org->bytes_per_px == 2;
  c->po_x = (fbx * MI) >> ssdec;
  c->po_y = (fby * MI) >> ssdec;
  c->w = w;
  c->h = h;
  c->allow_hp = t->p->allow_hp;
  c->lambda = t->p->lambda[ssdec];
  c->mvx_min = rng[0]; c->mvx_max = rng[1]; c->mvy_min = rng[2]; c->mvy_max = rng[3];
}

static void save_me_stats(const tilectx *t, int size_in_b, int bx, int by, mv_t mv, uint32_t nsad) {
  const int xe = imin(bx + size_in_b, t->tcols), ye = imin(by + size_in_b, t->trows);
  for (int y = by; y < ye; y++)
    for (int x = bx; x < xe; x++) {
      r1o_me_stats *s = tstat(t, y, x);
      s->row = mv.row; s->col = mv.col; s->normalized_sad = nsad;
    }
}

static void store_result(const tilectx *t, int log2b, int bx, int by, msr_t r, int w, int h,
                         int ssdec) {
  mv_t mv = { (int16_t)(r.mv.row << ssdec), (int16_t)(r.mv.col << ssdec) };
  const uint32_t nsad = (uint32_t)((((uint64_t)r.sad) << 14) / (uint64_t)(w * h));
  save_me_stats(t, 1 << log2b, bx, by, mv, nsad);
}

int r1o_estimate_tile_motion(const r1o_plane *org3, const r1o_plane *ref3, const r1o_me_params *p,
                             r1o_me_stats *stats, const r1o_me_stats *prev) {
  if (p->tile_x % SB || p->tile_y % SB || p->tile_w % MI || p->tile_h % MI) return -1;
  tilectx t = { p, stats, prev, p->tile_x / MI, p->tile_y / MI, p->tile_w / MI, p->tile_h / MI };
  const int sbw = (p->tile_w + SB - 1) / SB, sbh = (p->tile_h + SB - 1) / SB;
  for (int log2b = 4; log2b >= 2; log2b--) {
    const int init = log2b == 4, ssdec = log2b - 2;
    for (int sby = 0; sby < sbh; sby++)
      for (int sbx = 0; sbx < sbw; sbx++) {
        const int sb_w = imin(SB, p->tile_w - sbx * SB), sb_h = imin(SB, p->tile_h - sby * SB);
        if (!init) { /* new_subsampling: refine the previous pass' blocks at this resolution */
          const int sz = MI << (log2b + 1);
          for (int y = 0; y < sb_h; y += sz)
            for (int x = 0; x < sb_w; x += sz) {
              const int bx = sbx * 16 + x / MI, by = sby * 16 + y / MI;
              const int w = imin(sz, sb_w - x + (1 << ssdec) - 1) >> ssdec;
              const int h = imin(sz, sb_h - y + (1 << ssdec) - 1) >> ssdec;
              mectx c;
              int rng[4];
              block_ctx(&t, org3, ref3, bx, by, w, h, ssdec, &c, rng);
              const r1o_me_stats *s0 = tstat(&t, by, bx);
              const int mvc = (int16_t)(s0->col >> ssdec), mvr = (int16_t)(s0->row >> ssdec);
              const int x_lo = c.po_x + imax(mvc / 8 - 1, c.mvx_min / 8);
              const int x_hi = c.po_x + imin(mvc / 8 + 2, c.mvx_max / 8);
              const int y_lo = c.po_y + imax(mvr / 8 - 1, c.mvy_min / 8);
              const int y_hi = c.po_y + imin(mvr / 8 + 2, c.mvy_max / 8);
              const msr_t r = full_search(&c, x_lo, x_hi, y_lo, y_hi, 1);
              store_result(&t, log2b + 1, bx, by, r, w, h, ssdec);
            }
        }
        const int sz = MI << log2b;
        for (int y = 0; y < sb_h; y += sz)
          for (int x = 0; x < sb_w; x += sz) {
            const int corner = init ? 0 : (1 | ((x & sz) ? 2 : 0) | ((y & sz) ? 4 : 0));
            const int bx = sbx * 16 + x / MI, by = sby * 16 + y / MI;
            const int w = imin(sz, sb_w - x + (1 << ssdec) - 1) >> ssdec;
            const int h = imin(sz, sb_h - y + (1 << ssdec) - 1) >> ssdec;
            mectx c;
            int rng[4];
            block_ctx(&t, org3, ref3, bx, by, w, h, ssdec, &c, rng);
            const msr_t r = full_pixel_me(&t, &c, bx, by, rng, corner, init, ssdec);
            store_result(&t, log2b, bx, by, r, w, h, ssdec);
          }
      }
  }
  return 0;
}

/* ---- estimate_motion with pmv = Some(..): the RDO-time call (src/rdo.rs:1183-1196):
 * full_pixel_me at full resolution, then the SATD re-cost and the sub-pel
 * diamond (me.rs:594-626).  Blocks are independent: stats are only read. ---- */
static void pred_rd(const mectx *c, mv_t cand, const void *pred, int pstride, int use_satd,
                    uint64_t *cost, uint32_t *sad) {
  const void *o = px(c->org, c->po_x, c->po_y);
  const uint32_t s = use_satd ? r1o_get_satd(o, c->org->stride, pred, pstride, c->w, c->h, c->hbd)
                              : r1o_get_sad(o, c->org->stride, pred, pstride, c->w, c->h, c->hbd);
  const uint32_t r1 = get_mv_rate(cand, c->pmv[0], c->allow_hp);
  const uint32_t r2 = get_mv_rate(cand, c->pmv[1], c->allow_hp);
  const uint32_t rate = r1 < r2 + 1 ? r1 : r2 + 1;
  *cost = 256ull * s + (uint64_t)rate * c->lambda;
  *sad = s;
}

static int in_range(const mectx *c, mv_t m) {
  return !(m.col < c->mvx_min || m.col > c->mvx_max || m.row < c->mvy_min || m.row > c->mvy_max);
}

static void subpel_mv_rd(const mectx *c, mv_t cand, int use_satd, int mode, int bit_depth,
                         uint64_t *cost, uint32_t *sad) {
  if (!in_range(c, cand)) {
    *cost = UINT64_MAX;
    *sad = UINT32_MAX;
    return;
  }
  int mc_w = 1;
  while (mc_w < c->w) mc_w <<= 1; /* w.next_power_of_two() */
  const int mc_h = (c->h + 1) & ~1;
  uint16_t tmp[128 * 128];
  /* get_mv_params: floor offsets, 1/16 fractions (luma: xdec = ydec = 0) */
  const int row_off = cand.row >> 3, col_off = cand.col >> 3;
  const int row_frac = (cand.row << 1) & 15, col_frac = (cand.col << 1) & 15;
  r1o_put_8tap(tmp, mc_w, px(c->ref, c->po_x + col_off, c->po_y + row_off), c->ref->stride, mc_w,
               mc_h, col_frac, row_frac, mode, mode, bit_depth, c->hbd);
  pred_rd(c, cand, tmp, mc_w, use_satd, cost, sad);
}

int r1o_estimate_motion_batch(const r1o_plane *org3, const r1o_plane *ref3, const r1o_me_params *p,
                              const r1o_me_stats *stats, const r1o_me_stats *prev,
                              const r1o_me_block *blk, int n, int use_satd, int filter_mode,
                              r1o_me_result *out) {
  if (p->tile_x % SB || p->tile_y % SB || p->tile_w % MI || p->tile_h % MI) return -1;
  const tilectx t = { p, (r1o_me_stats *)stats, prev, p->tile_x / MI, p->tile_y / MI,
                      p->tile_w / MI, p->tile_h / MI };
#pragma omp parallel for schedule(dynamic, 8)
  for (int i = 0; i < n; i++) {
    mectx c;
    int rng[4];
    block_ctx(&t, org3, ref3, blk[i].bx, blk[i].by, blk[i].w, blk[i].h, 0, &c, rng);
    for (int k = 0; k < 2; k++) {
      c.pmv[k].row = blk[i].pmv[k][0];
      c.pmv[k].col = blk[i].pmv[k][1];
    }
    msr_t best = full_pixel_me(&t, &c, blk[i].bx, blk[i].by, rng, blk[i].corner, 0, 0);
    if (use_satd) { /* me.rs:596-613: get_fullpel_mv_rd(best.mv, use_satd) */
      if (!in_range(&c, best.mv)) {
        best.cost = UINT64_MAX;
        best.sad = UINT32_MAX;
      } else {
        pred_rd(&c, best.mv, px(c.ref, c.po_x + best.mv.col / 8, c.po_y + best.mv.row / 8),
                c.ref->stride, 1, &best.cost, &best.sad);
      }
    }
    int radius_log2 = 2;
    const int end_log2 = p->allow_hp ? 0 : 1;
    for (;;) {
      msr_t bc = msr_empty();
      for (int k = 0; k < 4; k++) {
        const mv_t cand = mv_add(best.mv, DIAMOND[k][0] << radius_log2, DIAMOND[k][1] << radius_log2);
        uint64_t cost;
        uint32_t sad;
        subpel_mv_rd(&c, cand, use_satd, filter_mode, p->bit_depth, &cost, &sad);
        if (cost < bc.cost) {
          bc.mv = cand; bc.cost = cost; bc.sad = sad;
        }
      }
      if (best.cost <= bc.cost) {
        if (radius_log2 == end_log2) break;
        radius_log2--;
      } else {
        best = bc;
      }
    }
    out[i].row = best.mv.row;
    out[i].col = best.mv.col;
    out[i].sad = best.sad;
    out[i].cost = best.cost;
  }
  return 0;
}
