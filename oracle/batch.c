/*
 * oracle/batch.c -- TEST INFRASTRUCTURE (see r1_oracle.h).
 * Host-memory mirrors of the batch C ABI: each candidate is evaluated by the
 * scalar restatements (dist.c, mc.c, fwd_tx.c), i.e. exactly what the
 * reference's Rust path would compute call by call
 * (compute_mv_rd src/me.rs:1445-1454; predict_inter_single
 * src/predict.rs:304-331; encode_tx_block src/encoder.rs:1533-1552).
 * OpenMP over candidates stands in for rav1e's rayon-over-tiles when this is
 * timed as bench.py's cpu_baseline.
 */
#include <omp.h>
#include <string.h>

#include "r1_oracle.h"

void r1o_set_threads(int n) { omp_set_num_threads(n > 0 ? n : 1); }

static inline const uint8_t *at(const r1o_plane *p, int x, int y) {
  return (const uint8_t *)p->data +
         ((size_t)(p->yorigin + y) * p->stride + (size_t)(p->xorigin + x)) *
             p->bytes_per_px;
}

int r1o_dist_batch(int kind, const r1o_plane *org, const r1o_plane *ref, int w,
                   int h, const r1o_dist_cand *c, int n, uint32_t *out) {
  const int hbd = org->bytes_per_px == 2;
  if (ref->bytes_per_px != org->bytes_per_px) return -1;
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n; i++) {
    const void *o = at(org, c[i].ox, c[i].oy), *r = at(ref, c[i].rx, c[i].ry);
    out[i] = kind == 0 ? r1o_get_sad(o, org->stride, r, ref->stride, w, h, hbd)
                       : r1o_get_satd(o, org->stride, r, ref->stride, w, h, hbd);
  }
  return 0;
}

int r1o_fwd_txfm_batch(const int16_t *residual, void *coeffs, int n,
                       int tx_size, int tx_type, int bit_depth,
                       int coeff_bytes) {
  if (!r1o_valid_av1_transform(tx_size, tx_type)) return -1;
  const int w = r1o_tx_width(tx_size), h = r1o_tx_height(tx_size);
  const size_t area = (size_t)w * h;
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n; i++)
    r1o_forward_transform(residual + i * area,
                          (uint8_t *)coeffs + i * area * coeff_bytes, w,
                          tx_size, tx_type, bit_depth, coeff_bytes == 4);
  return 0;
}

int r1o_mc_put_batch(const r1o_plane *ref, int w, int h, const r1o_mc_cand *c,
                     int n, void *dst) {
  const int hbd = ref->bytes_per_px == 2;
  const size_t area = (size_t)w * h;
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n; i++)
    r1o_put_8tap((uint8_t *)dst + i * area * ref->bytes_per_px, w,
                 at(ref, c[i].rx, c[i].ry), ref->stride, w, h, c[i].col_frac,
                 c[i].row_frac, c[i].mode_x, c[i].mode_y, ref->bit_depth, hbd);
  return 0;
}

int r1o_mc_prep_batch(const r1o_plane *ref, int w, int h, const r1o_mc_cand *c,
                      int n, int16_t *tmp) {
  const int hbd = ref->bytes_per_px == 2;
  const size_t area = (size_t)w * h;
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n; i++)
    r1o_prep_8tap(tmp + i * area, at(ref, c[i].rx, c[i].ry), ref->stride, w, h,
                  c[i].col_frac, c[i].row_frac, c[i].mode_x, c[i].mode_y,
                  ref->bit_depth, hbd);
  return 0;
}

int r1o_mc_avg_batch(const int16_t *t1, const int16_t *t2, int w, int h, int n,
                     int bit_depth, int bytes_per_px, void *dst) {
  const size_t area = (size_t)w * h;
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n; i++)
    r1o_mc_avg((uint8_t *)dst + i * area * bytes_per_px, w, t1 + i * area,
               t2 + i * area, w, h, bit_depth, bytes_per_px == 2);
  return 0;
}

int r1o_rdo_cand_batch(const r1o_plane *org, const r1o_plane *ref, int w, int h,
                       int tx_size, const r1o_rdo_cand *c, int n,
                       uint32_t *sad_out, uint32_t *satd_out, void *coeffs,
                       void *pred_out) {
  const int hbd = org->bytes_per_px == 2, bpp = org->bytes_per_px;
  const int cb = hbd ? 4 : 2;
  const size_t area = (size_t)w * h;
  if (coeffs && (r1o_tx_width(tx_size) != w || r1o_tx_height(tx_size) != h))
    return -1;
  int bad = 0;
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n; i++) {
    uint16_t pred16[128 * 128]; /* large enough for u8 or u16 blocks */
    int16_t resid[64 * 64];
    void *pred = pred16;
    r1o_put_8tap(pred, w, at(ref, c[i].rx, c[i].ry), ref->stride, w, h,
                 c[i].col_frac, c[i].row_frac, c[i].mode_x, c[i].mode_y,
                 ref->bit_depth, hbd);
    const void *o = at(org, c[i].ox, c[i].oy);
    if (sad_out) sad_out[i] = r1o_get_sad(o, org->stride, pred, w, w, h, hbd);
    if (satd_out) satd_out[i] = r1o_get_satd(o, org->stride, pred, w, w, h, hbd);
    if (pred_out) memcpy((uint8_t *)pred_out + i * area * bpp, pred, area * bpp);
    if (coeffs) {
      r1o_diff(resid, o, org->stride, pred, w, w, h, hbd);
      if (r1o_forward_transform(resid, (uint8_t *)coeffs + i * area * cb, w,
                                tx_size, c[i].tx_type, org->bit_depth, hbd))
        bad = 1;
    }
  }
  return bad ? -1 : 0;
}
