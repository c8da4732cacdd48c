/*
 * loop_decision.c -- CPU oracle: rdo_loop_decision with BOTH filters on, the parts the first-pass
 * functions (cdef.c r1o_cdef_strength_search, lrf.c r1o_lrf_search_unit) do not cover.
 * TEST INFRASTRUCTURE ONLY (see r1_oracle.h).
 *
 * Restates src/rdo.rs of the reference:
 *   the CDEF leg of a LATER pass, 2377-2560: for every superblock of an analysis area that is not
 *   completely skipped, every cdef_index: cdef_filter_superblock into the area's CDEF working copy
 *   (2400-2408), then per plane (2410-2520) -- a restoration unit with RestorationFilter::Sgrproj
 *   over the superblock: setup_integral_image on THAT superblock of the working copy, crop = stripe =
 *   its visible size (2458-2470), sgrproj_stripe_filter with the unit's (set, xqd) into the
 *   restoration working copy (2471-2483), rdo_loop_plane_error of the restored superblock (2484-2495);
 *   RestorationFilter::None / no unit: rdo_loop_plane_error of the CDEF output (2432-2443, 2504-2516);
 *   the final pass of a superblock with its chosen index, 2546-2560 ("keep cdef output up to date").
 * cdef_filter_superblock's edge logic: src/cdef.rs:405-560 (as in cdef.c).
 *
 * Pinning: tests/golden/loop_decision_ref.npz `ldb*` -- rdo_loop_decision executed whole through
 * tools/rustlite, every plane error of every pass recorded (tests/test_loop_decision_ref.py replays
 * the trace through these functions).
 */
#include <stdlib.h>
#include <string.h>

#include "r1_oracle.h"

enum { HAVE_LEFT = 1, HAVE_RIGHT = 2, HAVE_TOP = 4, HAVE_BOTTOM = 8 };

typedef struct {
  int ax0, ay0, sbx, sby, area_w, area_h, blk_cols, blk_rows;
} area_geo;

static area_geo area_of(const r1o_cdef_search_params *p, int mi_cols, int mi_rows, int fbx, int fby) {
  const int n_sbx = (mi_cols + 15) / 16, n_sby = (mi_rows + 15) / 16;
  area_geo g;
  g.ax0 = fbx / p->area_sb_w * p->area_sb_w;
  g.ay0 = fby / p->area_sb_h * p->area_sb_h;
  g.sbx = fbx - g.ax0;
  g.sby = fby - g.ay0;
  const int sb_w = p->area_sb_w < n_sbx - g.ax0 ? p->area_sb_w : n_sbx - g.ax0;
  const int sb_h = p->area_sb_h < n_sby - g.ay0 ? p->area_sb_h : n_sby - g.ay0;
  const int crop_w = p->crop_w - g.ax0 * 64, crop_h = p->crop_h - g.ay0 * 64;
  const int pixel_w = crop_w < sb_w * 64 ? crop_w : sb_w * 64;
  const int pixel_h = crop_h < sb_h * 64 ? crop_h : sb_h * 64;
  g.area_w = (pixel_w + 7) >> 3 << 3;
  g.area_h = (pixel_h + 7) >> 3 << 3;
  g.blk_cols = sb_w * 16 < mi_cols - g.ax0 * 16 ? sb_w * 16 : mi_cols - g.ax0 * 16;
  g.blk_rows = sb_h * 16 < mi_rows - g.ay0 * 16 ? sb_h * 16 : mi_rows - g.ay0 * 16;
  return g;
}

static int sb_all_skip(const area_geo *g, const uint8_t *skip_mi, int mi_stride) {
  int all_skip = 1;
  for (int y = 16 * g->sby; y < 16 * g->sby + 16 && y < g->blk_rows; y++)
    for (int x = 16 * g->sbx; x < 16 * g->sbx + 16 && x < g->blk_cols; x++)
      all_skip &= skip_mi[(size_t)(g->ay0 * 16 + y) * mi_stride + g->ax0 * 16 + x] != 0;
  return all_skip;
}

static uint8_t *px_ptr(const r1o_plane *p, int x, int y) {
  return (uint8_t *)p->data + ((size_t)(p->yorigin + y) * p->stride + p->xorigin + x) * p->bytes_per_px;
}

/* cdef_filter_superblock (cdef.rs:405-560) of superblock (fbx, fby) with cdef_index idx: rec -> dst
 * (whole-frame planes, same coordinates).  Skipped 8x8 blocks are left as dst holds them. */
static void cdef_filter_sb(const r1o_plane *rec, const r1o_plane *dst, const uint8_t *skip_mi, int mi_stride,
                           const r1o_cdef_search_params *p, const area_geo *g, int idx) {
  const int bd = p->bit_depth, coeff_shift = bd - 8, hbd = rec[0].bytes_per_px == 2;
  const int ys = p->y_strengths[idx], uvs = p->uv_strengths[idx];
  const int pri_y = ys / 4, pri_uv = uvs / 4;
  int sec_y = ys % 4, sec_uv = uvs % 4;
  if (sec_y == 3) sec_y++;
  if (sec_uv == 3) sec_uv++;
  const int in_xoff = g->sbx * 64, in_yoff = g->sby * 64;
  const int xavail = g->area_w - in_xoff, yavail = g->area_h - in_yoff;
  const int have_top = g->sby > 0 ? HAVE_TOP : 0, have_left = g->sbx > 0 ? HAVE_LEFT : 0;
  int edges = have_top | HAVE_BOTTOM;
  for (int by = 0; by < 8; by++) {
    if (by + 1 >= (yavail >> 3)) edges &= ~HAVE_BOTTOM;
    edges &= ~HAVE_LEFT;
    edges |= have_left;
    edges |= HAVE_RIGHT;
    for (int bx = 0; bx < 8; bx++) {
      if (bx + 1 >= (xavail >> 3)) edges &= ~HAVE_RIGHT;
      const int mx = g->sbx * 16 + 2 * bx, my = g->sby * 16 + 2 * by;
      if (mx < g->blk_cols && my < g->blk_rows) {
        const uint8_t *sk = skip_mi + (size_t)(g->ay0 * 16 + my) * mi_stride + g->ax0 * 16 + mx;
        const int skip = sk[0] && sk[1] && sk[mi_stride] && sk[mi_stride + 1];
        const int flx = g->ax0 * 64 + in_xoff + 8 * bx, fly = g->ay0 * 64 + in_yoff + 8 * by;
        if (!skip) {
          uint32_t var = 0;
          const int dir = r1o_cdef_find_dir(px_ptr(&rec[0], flx, fly), rec[0].stride, &var, coeff_shift, hbd);
          for (int pl = 0; pl < p->planes; pl++) {
            const int xdec = pl ? p->xdec : 0, ydec = pl ? p->ydec : 0;
            int lpri, lsec, ldamp = p->damping + coeff_shift, ldir;
            if (pl == 0) {
              lpri = r1o_cdef_adjust_strength(pri_y << coeff_shift, (int)var);
              lsec = sec_y << coeff_shift;
              ldir = pri_y != 0 ? dir : 0;
            } else {
              static const uint8_t UVDIR[8] = {7, 0, 2, 4, 5, 6, 6, 6};
              lpri = pri_uv << coeff_shift;
              lsec = sec_uv << coeff_shift;
              ldamp -= 1;
              ldir = pri_uv != 0 ? (xdec != ydec ? UVDIR[dir] : dir) : 0;
            }
            r1o_cdef_filter_block(px_ptr(&dst[pl], flx >> xdec, fly >> ydec), dst[pl].stride,
                                  px_ptr(&rec[pl], flx >> xdec, fly >> ydec), rec[pl].stride, lpri, lsec, ldir, ldamp,
                                  bd, xdec, ydec, edges, hbd);
          }
        }
      }
      edges |= HAVE_LEFT;
    }
    edges |= HAVE_TOP;
  }
}

static void copy_sb(const r1o_plane *from, const r1o_plane *to, const r1o_cdef_search_params *p, int fbx, int fby,
                    int mi_cols, int mi_rows) {
  for (int pl = 0; pl < p->planes; pl++) {
    const int xdec = pl ? p->xdec : 0, ydec = pl ? p->ydec : 0;
    const int x0 = (fbx * 64) >> xdec, y0 = (fby * 64) >> ydec;
    int w = 64 >> xdec, h = 64 >> ydec;
    const int fw = (mi_cols * 4) >> xdec, fh = (mi_rows * 4) >> ydec;   /* the block grid: whole 8x8 blocks */
    if (x0 + w > fw) w = fw - x0;
    if (y0 + h > fh) h = fh - y0;
    for (int y = 0; y < h; y++)
      memcpy(px_ptr(&to[pl], x0, y0 + y), px_ptr(&from[pl], x0, y0 + y), (size_t)w * from[pl].bytes_per_px);
  }
}

/* The CDEF working copy of every area: out = rec with every superblock filtered by index_sb[sb]
 * (< 0 or completely skipped: left as rec).  out: whole-frame planes, distinct from rec. */
int r1o_cdef_apply_area(const r1o_plane *rec, const r1o_plane *out, const uint8_t *skip_mi, int mi_stride,
                        int mi_cols, int mi_rows, const r1o_cdef_search_params *p, const int8_t *index_sb) {
  if (p->n_idx < 1 || p->n_idx > 8 || (p->planes != 1 && p->planes != 3)) return -1;
  const int n_sbx = (mi_cols + 15) / 16, n_sby = (mi_rows + 15) / 16;
  for (int fby = 0; fby < n_sby; fby++)
    for (int fbx = 0; fbx < n_sbx; fbx++) {
      const area_geo g = area_of(p, mi_cols, mi_rows, fbx, fby);
      copy_sb(rec, out, p, fbx, fby, mi_cols, mi_rows);
      const int idx = index_sb[fby * n_sbx + fbx];
      if (idx < 0 || idx >= p->n_idx || sb_all_skip(&g, skip_mi, mi_stride)) continue;
      cdef_filter_sb(rec, out, skip_mi, mi_stride, p, &g, idx);
    }
  return 0;
}

/* One pass of the CDEF leg with the restoration units' current choices in play.
 *   rec: the deblocked frame; work: the CDEF working copy (r1o_cdef_apply_area) -- MODIFIED while a
 *   trial runs and put back afterwards, so the caller's copy is unchanged on return; src: the source.
 *   units: n_units[0] luma entries, then U, then V: the superblocks under a self-guided choice (x, y,
 *   w, h = the superblock's visible rectangle in plane pixels, edges = R1O_SGR_EDGE_*, sb = fby * n_sbx + fbx).
 *   sb_sel (NULL = all): superblocks to evaluate.
 *   err: [n_sb][8], err_planes: [n_sb][8][3] (may be NULL), best: [n_sb] (-1 = skipped / not selected). */
int r1o_cdef_lrf_trial(const r1o_plane *rec, const r1o_plane *work, const r1o_plane *src, const uint8_t *skip_mi,
                       int mi_stride, int mi_cols, int mi_rows, const uint32_t *scales, int scale_stride,
                       const r1o_cdef_search_params *p, const r1o_trial_unit *units, const int32_t *n_units,
                       const uint8_t *sb_sel, uint64_t *err, uint64_t *err_planes, int8_t *best) {
  if (p->n_idx < 1 || p->n_idx > 8 || (p->planes != 1 && p->planes != 3)) return -1;
  if (p->area_sb_w < 1 || p->area_sb_h < 1) return -1;
  const int n_sbx = (mi_cols + 15) / 16, n_sby = (mi_rows + 15) / 16, bd = p->bit_depth;
  memset(err, 0, sizeof(uint64_t) * 8 * (size_t)n_sbx * n_sby);
  if (err_planes) memset(err_planes, 0, sizeof(uint64_t) * 24 * (size_t)n_sbx * n_sby);
  void *restored = malloc(64 * 64 * 2);
  r1o_plane saved[3];   /* whole-frame scratch planes with work's geometry: the superblock under trial, saved */
  for (int pl = 0; pl < p->planes; pl++) {
    saved[pl] = work[pl];
    saved[pl].data = malloc((size_t)work[pl].stride * work[pl].alloc_height * work[pl].bytes_per_px);
  }
  for (int fby = 0; fby < n_sby; fby++)
    for (int fbx = 0; fbx < n_sbx; fbx++) {
      const int sb = fby * n_sbx + fbx;
      const area_geo g = area_of(p, mi_cols, mi_rows, fbx, fby);
      best[sb] = -1;
      if (sb_all_skip(&g, skip_mi, mi_stride) || (sb_sel && !sb_sel[sb])) continue;
      /* the unit of each plane over this superblock, if any */
      const r1o_trial_unit *un[3] = {NULL, NULL, NULL};
      int first = 0;
      for (int pl = 0; pl < p->planes; pl++) {
        for (int i = 0; i < n_units[pl]; i++)
          if (units[first + i].sb == sb) un[pl] = &units[first + i];
        first += n_units[pl];
      }
      uint64_t *eo = err + 8 * (size_t)sb;
      copy_sb(work, &saved[0], p, fbx, fby, mi_cols, mi_rows);
      for (int idx = 0; idx < p->n_idx; idx++) {
        /* the trial's CDEF output goes into the working copy: skipped blocks keep the deblocked pixels */
        copy_sb(rec, work, p, fbx, fby, mi_cols, mi_rows);
        cdef_filter_sb(rec, work, skip_mi, mi_stride, p, &g, idx);
        uint64_t e = 0;
        for (int pl = 0; pl < p->planes; pl++) {
          const int xdec = pl ? p->xdec : 0, ydec = pl ? p->ydec : 0;
          uint64_t sum;
          if (un[pl]) {
            const r1o_trial_unit *u = un[pl];
            if (r1o_sgr_filter_rect(&work[pl], u->x, u->y, u->w, u->h, u->set, u->xqd, u->edges, bd, restored)) {
              free(restored);
              for (int q = 0; q < p->planes; q++) free(saved[q].data);
              return -1;
            }
            sum = r1o_loop_plane_error_rect(&src[pl], restored, u->w, u->x, u->y, u->w, u->h, pl != 0, xdec, ydec, scales,
                                            scale_stride, bd);
          } else {
            /* rdo_loop_plane_error over the blocks of the superblock inside the block grid (rdo.rs:2040-2043) */
            const int x0 = (fbx * 64) >> xdec, y0 = (fby * 64) >> ydec;
            int w = 64 >> xdec, h = 64 >> ydec;
            const int gw = ((g.ax0 * 16 + g.blk_cols) * 4) >> xdec, gh = ((g.ay0 * 16 + g.blk_rows) * 4) >> ydec;
            if (x0 + w > gw) w = gw - x0;
            if (y0 + h > gh) h = gh - y0;
            sum = r1o_loop_plane_error_rect(&src[pl], px_ptr(&work[pl], x0, y0), work[pl].stride, x0, y0, w, h, pl != 0,
                                            xdec, ydec, scales, scale_stride, bd);
          }
          const uint64_t ep = ((uint64_t)p->dist_scale[pl] * sum + 8192) >> 14;   /* Distortion * dist_scale */
          if (err_planes) err_planes[(size_t)sb * 24 + idx * 3 + pl] = ep;
          e += ep;
        }
        eo[idx] = e;
      }
      copy_sb(&saved[0], work, p, fbx, fby, mi_cols, mi_rows);   /* the caller's working copy as it was */
      int b = 0;
      for (int idx = 1; idx < p->n_idx; idx++)
        if ((double)eo[idx] < (double)eo[b]) b = idx;
      best[sb] = (int8_t)b;
    }
  free(restored);
  for (int pl = 0; pl < p->planes; pl++) free(saved[pl].data);
  return 0;
}
