/*
 * oracle/lookahead.c -- TEST INFRASTRUCTURE (see r1_oracle.h).
 * The lookahead cost maps (SURVEY.md 8f "N1"), composed from the scalar
 * restatements exactly as the reference composes its own kernels:
 *   estimate_intra_costs                  src/api/lookahead.rs:30-123
 *     (get_intra_edges -> DC_PRED predict_intra -> get_satd per 8x8 block; the variant of the
 *      prediction is NONE for every block, see the call)
 *   estimate_importance_block_difference  src/api/lookahead.rs:125-180
 *   estimate_inter_costs (the SATD map)   src/api/lookahead.rs:226-268
 *   update_block_importances              src/api/internal.rs:911-1068
 *     (its SATD map is estimate_inter_costs' map; the f32 propagation follows)
 *     (the motion vectors come from compute_motion_vectors -- motion search is
 *      a separate "next" row -- and are an input here)
 */
#include <stdlib.h>
#include <string.h>

#include "r1_oracle.h"

static inline const uint8_t *pat(const r1o_plane *p, int x, int y) {
  return (const uint8_t *)p->data +
         ((size_t)(p->yorigin + y) * p->stride + (size_t)(p->xorigin + x)) * p->bytes_per_px;
}

/* costs: (height/8) * (width/8) entries, row-major */
void r1o_estimate_intra_costs(const r1o_plane *plane, int bit_depth, uint32_t *costs) {
  const int hbd = plane->bytes_per_px == 2, bpp = plane->bytes_per_px;
  const int wb = plane->width / 8, hb = plane->height / 8;
#pragma omp parallel for schedule(static)
  for (int y = 0; y < hb; y++)
    for (int x = 0; x < wb; x++) {
      uint16_t edge16[257];
      uint8_t edge8[257];
      void *edge = hbd ? (void *)edge16 : (void *)edge8;
      int lens[2];
      /* dst = plane.as_region(): the tile is the whole plane */
      r1o_get_intra_edges(edge, lens, pat(plane, 0, 0), plane->stride, x * 8, y * 8, plane->width,
                          plane->height, 1 /* TX_8X8 */, bit_depth, 0 /* DC_PRED */, 0, 0, 0, 0, hbd);
      uint16_t pred16[64];
      uint8_t pred8[64];
      void *pred = hbd ? (void *)pred16 : (void *)pred8;
      /* predict_intra's (x, y) is the block position RELATIVE TO tile_rect, and the reference passes
       * TileRect { x: x * 8, y: y * 8, .. } (lookahead.rs:84-89): (0, 0) for every block, i.e.
       * PredictionVariant::NONE -> pred_dc_128 (predict.rs:212-218) */
      r1o_predict_intra(0, 0, 0, pred, 8, 1, bit_depth, NULL, 0, 0, 0, edge, lens[0], lens[1],
                        8, 8, hbd);
      costs[y * wb + x] = r1o_get_satd(pat(plane, x * 8, y * 8), plane->stride, pred, 8, 8, 8, hbd);
      (void)bpp;
    }
}

/* returns the integer sum of |mean_org - mean_ref| over the importance blocks
 * (the reference divides by the block count in f64 afterwards) */
uint64_t r1o_importance_block_difference(const r1o_plane *org, const r1o_plane *ref) {
  const int hbd = org->bytes_per_px == 2;
  const int wb = org->width / 8, hb = org->height / 8;
  uint64_t total = 0;
  for (int y = 0; y < hb; y++)
    for (int x = 0; x < wb; x++) {
      int64_t so = 0, sr = 0;
      for (int i = 0; i < 8; i++)
        for (int j = 0; j < 8; j++) {
          const uint8_t *a = pat(org, x * 8 + j, y * 8 + i), *b = pat(ref, x * 8 + j, y * 8 + i);
          so += hbd ? *(const uint16_t *)a : *a;
          sr += hbd ? *(const uint16_t *)b : *b;
        }
      const int64_t d = (so + 32) / 64 - (sr + 32) / 64;
      total += (uint64_t)(d < 0 ? -d : d);
    }
  return total;
}

/* mvs: (row, col) int16 pairs in 1/8 pel, one per importance block (the
 * reference reads stats[y*2][x*2].mv).  costs: per-block SATD. */
void r1o_estimate_inter_costs(const r1o_plane *org, const r1o_plane *ref, const int16_t *mvs,
                              uint32_t *costs) {
  const int hbd = org->bytes_per_px == 2;
  const int wb = org->width / 8, hb = org->height / 8;
#pragma omp parallel for schedule(static)
  for (int y = 0; y < hb; y++)
    for (int x = 0; x < wb; x++) {
      const int64_t rx = (int64_t)x * 64 + mvs[2 * (y * wb + x) + 1];
      const int64_t ry = (int64_t)y * 64 + mvs[2 * (y * wb + x)];
      /* `as isize / 8`: Rust integer division truncates toward zero */
      const int px = (int)(rx / 8), py = (int)(ry / 8);
      costs[y * wb + x] = r1o_get_satd(pat(org, x * 8, y * 8), org->stride, pat(ref, px, py),
                                       ref->stride, 8, 8, hbd);
    }
}

/* ActivityMask::from_plane + fill_scales (src/activity.rs:21-66), variance_8x8 69-99 */
void r1o_activity_scales(const r1o_plane *luma, uint32_t *variances, uint32_t *scales) {
  const int wb = (luma->width + 7) / 8, hb = (luma->height + 7) / 8;
  for (int by = 0; by < hb; by++)
    for (int bx = 0; bx < wb; bx++) {
      uint16_t sum_cols[8] = { 0 };
      uint32_t sum2_cols[8] = { 0 };
      for (int j = 0; j < 8; j++)
        for (int k = 0; k < 8; k++) {
          const size_t i = (size_t)(luma->yorigin + by * 8 + j) * luma->stride + luma->xorigin + bx * 8 + k;
          const uint16_t s = luma->bytes_per_px == 1 ? ((const uint8_t *)luma->data)[i]
                                                     : ((const uint16_t *)luma->data)[i];
          sum_cols[k] = (uint16_t)(sum_cols[k] + s);
          sum2_cols[k] += (uint32_t)s * s;
        }
      uint64_t sum_s = 0, sum_s2 = 0;
      for (int k = 0; k < 8; k++) { sum_s += sum_cols[k]; sum_s2 += sum2_cols[k]; }
      const uint64_t v = sum_s2 - ((sum_s * sum_s + 32) >> 6);
      const uint32_t var = v > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)v;
      if (variances) variances[by * wb + bx] = var;
      if (scales) scales[by * wb + bx] = r1o_apply_ssim_boost(1u << 14, var, var, luma->bit_depth);
    }
}

/* update_block_importances (src/api/internal.rs:911-1068) after its get_satd:
 * every 8x8 importance block of the current frame hands
 *   (intra_cost + future_importance) * (1 - inter_cost / intra_cost) / len
 * to the (up to) four importance blocks of the reference frame that its motion-
 * compensated position overlaps, by overlap area.  All arithmetic is f32, one
 * IEEE operation at a time (Rust never contracts a * b + c), and the additions
 * into `ref_importances` happen in the raster order of the source blocks,
 * top-left / top-right / bottom-left / bottom-right within a block.
 * intra_costs / inter_costs / future_importances / ref_importances: w x h maps;
 * mvs: (row, col) in 1/8 pel per block (me_stats[2y][2x].mv). */
void r1o_update_block_importances(const uint32_t *intra_costs, const float *future_importances,
                                  const uint32_t *inter_costs, const int16_t *mvs, int w, int h,
                                  int len, float *ref_importances) {
  const int64_t U = 64; /* IMP_BLOCK_SIZE_IN_MV_UNITS = 8 px * 8 units */
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      const int i = y * w + x;
      const int64_t rx = (int64_t)x * U + mvs[2 * i + 1], ry = (int64_t)y * U + mvs[2 * i];
      const float inter = (float)inter_costs[i], intra = (float)intra_costs[i];
      volatile float frac = 0.f; /* volatile: no re-association, no contraction */
      if (!(intra <= inter)) {
        volatile float q = inter / intra;
        frac = 1.f - q;
      }
      volatile float sum = intra + future_importances[i];
      volatile float prod = sum * frac;
      const float amount = prod / (float)len;
      const int64_t tlx = (rx - (rx < 0 ? U - 1 : 0)) / U * U, tly = (ry - (ry < 0 ? U - 1 : 0)) / U * U;
      const int64_t bx[4] = { tlx, tlx + U, tlx, tlx + U }, by[4] = { tly, tly, tly + U, tly + U };
      const int64_t wx[4] = { tlx + U - rx, rx + U - (tlx + U), tlx + U - rx, rx + U - (tlx + U) };
      const int64_t wy[4] = { tly + U - ry, tly + U - ry, ry + U - (tly + U), ry + U - (tly + U) };
      for (int c = 0; c < 4; c++) {
        const int64_t dx = bx[c] / U, dy = by[c] / U;
        if (dx >= 0 && dy >= 0 && dx < w && dy < h) {
          const float fraction = (float)(wx[c] * wy[c]) / (float)(U * U);
          volatile float add = amount * fraction;
          volatile float acc = ref_importances[dy * w + dx] + add;
          ref_importances[dy * w + dx] = acc;
        }
      }
    }
}
