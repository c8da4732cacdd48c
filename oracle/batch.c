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
#include <stdlib.h>
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

/* One inter RDO candidate end to end, the composition encode_tx_block runs
 * for RDOType::TxDistEstRate (src/encoder.rs:1533-1650): motion-compensated
 * prediction (put_8tap) -> residual -> forward_transform -> quantize ->
 * dequantize -> transform-domain distortion -> estimate_rate.  The
 * per-candidate tx_type selects both the 1-D kernels and the scan order.
 * qcoeffs (optional): dense coded-area blocks, coefficient size 2 (8-bit) or 4. */
int r1o_rdo_full_cand_batch(const r1o_plane *org, const r1o_plane *ref, int w, int h,
                            int tx_size, const r1o_rdo_cand *c, int n, int qindex,
                            int is_intra, int dc_delta_q, int ac_delta_q,
                            uint32_t *sad_out, uint32_t *satd_out, uint16_t *eob_out,
                            uint64_t *tx_dist_out, uint64_t *est_rate_out,
                            void *qcoeffs_out) {
  const int hbd = org->bytes_per_px == 2;
  const int cb = hbd ? 4 : 2;
  if (r1o_tx_width(tx_size) != w || r1o_tx_height(tx_size) != h) return -1;
  const int cw = w < 32 ? w : 32, ch = h < 32 ? h : 32;
  const size_t carea = (size_t)cw * ch;
  int bad = 0;
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n; i++) {
    uint16_t pred16[64 * 64];
    int16_t resid[64 * 64];
    int32_t co[64 * 64], qc[32 * 32], rc[32 * 32]; /* room for either width */
    void *pred = pred16;
    r1o_put_8tap(pred, w, at(ref, c[i].rx, c[i].ry), ref->stride, w, h,
                 c[i].col_frac, c[i].row_frac, c[i].mode_x, c[i].mode_y,
                 ref->bit_depth, hbd);
    const void *o = at(org, c[i].ox, c[i].oy);
    if (sad_out) sad_out[i] = r1o_get_sad(o, org->stride, pred, w, w, h, hbd);
    if (satd_out) satd_out[i] = r1o_get_satd(o, org->stride, pred, w, w, h, hbd);
    r1o_diff(resid, o, org->stride, pred, w, w, h, hbd);
    if (r1o_forward_transform(resid, co, w, tx_size, c[i].tx_type, org->bit_depth, hbd)) {
      bad = 1;
      continue;
    }
    const int eob = r1o_quantize(co, qc, tx_size, c[i].tx_type, qindex, org->bit_depth,
                                 is_intra, dc_delta_q, ac_delta_q, hbd);
    if (eob < 0) {
      bad = 1;
      continue;
    }
    eob_out[i] = (uint16_t)eob;
    r1o_dequantize(qc, rc, tx_size, qindex, org->bit_depth, dc_delta_q, ac_delta_q, hbd);
    tx_dist_out[i] = r1o_tx_domain_distortion(co, rc, tx_size, hbd);
    if (est_rate_out) est_rate_out[i] = r1o_estimate_rate(qindex, tx_size, tx_dist_out[i]);
    if (qcoeffs_out) memcpy((uint8_t *)qcoeffs_out + i * carea * cb, qc, carea * cb);
  }
  return bad ? -1 : 0;
}

/* The default-configuration RDO evaluation of one inter transform block
 * (tune = Psychovisual or need_recon_pixel: encode_tx_block's pixel path,
 * src/encoder.rs:1533-1661, + compute_distortion, src/rdo.rs:254-340):
 * put_8tap -> diff -> forward_transform -> quantize -> dequantize ->
 * inverse_transform_add -> sse_wxh (kind 2) / cdef_dist_wxh (kind 3) of the
 * reconstruction against the source, with the DistortionScale grid. */
int r1o_rdo_pixel_cand_batch(const r1o_plane *org, const r1o_plane *ref, int w, int h,
                             int tx_size, const r1o_rdo_cand *c, int n, int qindex,
                             int is_intra, int dc_delta_q, int ac_delta_q, int kind,
                             const uint32_t *scales, int scale_stride, int xdec, int ydec,
                             uint32_t *sad_out, uint32_t *satd_out, uint16_t *eob_out,
                             uint64_t *dist_out, void *qcoeffs_out, void *rec_out,
                             const void *pred_in) {
  /* pred_in (dense n x h x w pixels) replaces put_8tap of `ref` (intra
   * predictions, compound averages); kind 0: transform-domain distortion */
  const int hbd = org->bytes_per_px == 2, bpp = org->bytes_per_px;
  const int cb = hbd ? 4 : 2;
  if (r1o_tx_width(tx_size) != w || r1o_tx_height(tx_size) != h) return -1;
  const int cw = w < 32 ? w : 32, ch = h < 32 ? h : 32;
  const size_t carea = (size_t)cw * ch, area = (size_t)w * h;
  int bad = 0;
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n; i++) {
    uint16_t pred16[64 * 64];
    int16_t resid[64 * 64];
    int32_t co[64 * 64], qc[32 * 32], rc[32 * 32];
    void *pred = pred16;
    if (pred_in)
      memcpy(pred, (const uint8_t *)pred_in + i * area * bpp, area * bpp);
    else
      r1o_put_8tap(pred, w, at(ref, c[i].rx, c[i].ry), ref->stride, w, h,
                   c[i].col_frac, c[i].row_frac, c[i].mode_x, c[i].mode_y,
                   ref->bit_depth, hbd);
    const void *o = at(org, c[i].ox, c[i].oy);
    if (sad_out) sad_out[i] = r1o_get_sad(o, org->stride, pred, w, w, h, hbd);
    if (satd_out) satd_out[i] = r1o_get_satd(o, org->stride, pred, w, w, h, hbd);
    r1o_diff(resid, o, org->stride, pred, w, w, h, hbd);
    if (r1o_forward_transform(resid, co, w, tx_size, c[i].tx_type, org->bit_depth, hbd)) {
      bad = 1;
      continue;
    }
    const int eob = r1o_quantize(co, qc, tx_size, c[i].tx_type, qindex, org->bit_depth,
                                 is_intra, dc_delta_q, ac_delta_q, hbd);
    if (eob < 0) {
      bad = 1;
      continue;
    }
    eob_out[i] = (uint16_t)eob;
    r1o_dequantize(qc, rc, tx_size, qindex, org->bit_depth, dc_delta_q, ac_delta_q, hbd);
    if (qcoeffs_out) memcpy((uint8_t *)qcoeffs_out + i * carea * cb, qc, carea * cb);
    if (kind == 0) {
      dist_out[i] = r1o_tx_domain_distortion(co, rc, tx_size, hbd);
      continue;
    }
    /* the prediction buffer becomes the reconstruction */
    if (r1o_inverse_transform_add(rc, pred, w, tx_size, c[i].tx_type, org->bit_depth, hbd, hbd)) {
      bad = 1;
      continue;
    }
    /* the reconstruction as a one-block plane for the scaled-distortion glue */
    r1o_plane rp = { pred, w, h, w, h, 0, 0, bpp, org->bit_depth };
    const r1o_dist_cand dc = { c[i].ox, c[i].oy, 0, 0 };
    r1o_dist_scaled_batch(kind, org, &rp, w, h, &dc, 1, scales, scale_stride, xdec, ydec,
                          &dist_out[i]);
    if (rec_out) memcpy((uint8_t *)rec_out + i * area * bpp, pred, area * bpp);
  }
  return bad ? -1 : 0;
}

/* get_tx_set (src/context/transform_unit.rs:123-148) with TxSize::sqr / sqr_up
 * (src/transform/mod.rs:220-239) on the sides, then the row of av1_tx_used
 * (transform_unit.rs:37-44) as a bit mask over TxType; rav1e_only keeps the
 * entries of RAV1E_TX_TYPES (src/transform/mod.rs:28-44).  This is the filter
 * of rdo_tx_type_decision's loop (src/rdo.rs:1732-1736). */
uint32_t r1o_tx_type_mask(int tx_size, int is_inter, int use_reduced_set, int rav1e_only) {
  static const uint8_t used[6][16] = {
      {1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0},   /* TX_SET_DCTONLY */
      {1, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0},   /* TX_SET_INTER_3 */
      {1, 1, 1, 1, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0},   /* TX_SET_INTRA_2 */
      {1, 1, 1, 1, 0, 0, 0, 0, 0, 1, 1, 1, 0, 0, 0, 0},   /* TX_SET_INTRA_1 */
      {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0},   /* TX_SET_INTER_2 */
      {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1}};  /* TX_SET_INTER_1 */
  static const uint8_t rav1e_types[7] = {0, 1, 2, 3, 9, 10, 11};
  if (tx_size < 0 || tx_size >= 19) return 0;
  const int w = r1o_tx_width(tx_size), h = r1o_tx_height(tx_size);
  const int sqr_up = w > h ? w : h, sqr = w < h ? w : h;
  int set;
  if (sqr_up > 32) set = 0;
  else if (is_inter) set = (use_reduced_set || sqr_up == 32) ? 1 : (sqr == 16 ? 4 : 5);
  else if (sqr_up == 32) set = 0;
  else set = (use_reduced_set || sqr == 16) ? 2 : 3;
  uint32_t m = 0;
  if (rav1e_only) {
    for (int k = 0; k < 7; k++)
      if (used[set][rav1e_types[k]]) m |= 1u << rav1e_types[k];
  } else {
    for (int t = 0; t < 16; t++)
      if (used[set][t]) m |= 1u << t;
  }
  return m;
}

/* rdo_tx_type_decision's loop body (src/rdo.rs:1731-1810) for one prediction per
 * candidate: for every TxType t of the mask in ascending order the evaluation of
 * r1o_rdo_pixel_cand_batch (kind 2 / 3) or r1o_rdo_full_cand_batch (kind 0) with
 * tx_type = t on the SAME prediction (the reference re-runs motion_compensate per
 * type; the prediction does not depend on t).  Slot (i, j) = i * nt + j. */
int r1o_rdo_txsearch_batch(const r1o_plane *org, const r1o_plane *ref, const void *pred_in,
                           int w, int h, int tx_size, const r1o_rdo_cand *c, int n,
                           uint32_t tx_type_mask, int qindex, int is_intra, int dc_delta_q,
                           int ac_delta_q, int kind, const uint32_t *scales, int scale_stride,
                           int xdec, int ydec, uint32_t *sad_out, uint32_t *satd_out,
                           uint16_t *eob_out, uint64_t *dist_out, uint64_t *est_rate_out,
                           void *qcoeffs_out, void *rec_out) {
  const int hbd = org->bytes_per_px == 2, bpp = org->bytes_per_px;
  const int cb = hbd ? 4 : 2;
  const int cw = w < 32 ? w : 32, ch = h < 32 ? h : 32;
  const size_t carea = (size_t)cw * ch, area = (size_t)w * h;
  int nt = 0, types[16];
  for (int t = 0; t < 16; t++)
    if ((tx_type_mask >> t) & 1) types[nt++] = t;
  if (!nt || (tx_type_mask >> 16)) return -1;
  int bad = 0;
  for (int j = 0; j < nt; j++) {
    r1o_rdo_cand *cj = (r1o_rdo_cand *)malloc(sizeof(*cj) * (size_t)(n > 0 ? n : 1));
    uint16_t *eob = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)(n > 0 ? n : 1));
    uint64_t *dist = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(n > 0 ? n : 1));
    uint8_t *qc = qcoeffs_out ? (uint8_t *)malloc(carea * cb * (size_t)(n > 0 ? n : 1)) : NULL;
    uint8_t *rec = rec_out ? (uint8_t *)malloc(area * bpp * (size_t)(n > 0 ? n : 1)) : NULL;
    for (int i = 0; i < n; i++) {
      cj[i] = c[i];
      cj[i].tx_type = (uint8_t)types[j];
    }
    if (r1o_rdo_pixel_cand_batch(org, ref, w, h, tx_size, cj, n, qindex, is_intra, dc_delta_q,
                                 ac_delta_q, kind, scales, scale_stride, xdec, ydec,
                                 j == 0 ? sad_out : NULL, j == 0 ? satd_out : NULL, eob, dist, qc, rec,
                                 pred_in))
      bad = 1;
    for (int i = 0; i < n; i++) {
      const size_t s = (size_t)i * nt + j;
      eob_out[s] = eob[i];
      dist_out[s] = dist[i];
      if (est_rate_out && kind == 0) est_rate_out[s] = r1o_estimate_rate(qindex, tx_size, dist[i]);
      if (qc) memcpy((uint8_t *)qcoeffs_out + s * carea * cb, qc + (size_t)i * carea * cb, carea * cb);
      if (rec) memcpy((uint8_t *)rec_out + s * area * bpp, rec + (size_t)i * area * bpp, area * bpp);
    }
    free(cj); free(eob); free(dist); free(qc); free(rec);
  }
  return bad ? -1 : 0;
}

/* sse_wxh / cdef_dist_wxh (src/rdo.rs:142-224) over a candidate list, with
 * compute_bias = distortion_scale (src/rdo.rs:443-459): one Q14
 * DistortionScale per 8x8 LUMA importance block of the frame,
 * scales[(luma_y >> 3) * scale_stride + (luma_x >> 3)]; NULL = default 1<<14.
 * kind 2: weighted SSE (any plane decimation), kind 3: cdef_dist (luma). */
int r1o_dist_scaled_batch(int kind, const r1o_plane *org, const r1o_plane *ref,
                          int w, int h, const r1o_dist_cand *c, int n,
                          const uint32_t *scales, int scale_stride, int xdec,
                          int ydec, uint64_t *out) {
  const int hbd = org->bytes_per_px == 2;
  if (ref->bytes_per_px != org->bytes_per_px) return -1;
  if (kind != 2 && kind != 3) return -1;
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n; i++) {
    const void *o = at(org, c[i].ox, c[i].oy), *r = at(ref, c[i].rx, c[i].ry);
    if (kind == 2) {
      /* sse_wxh: per-4x4-cell scale buffer, stride = next_pow2(ceil(w/4)) */
      uint32_t buf[32 * 32];
      const int nbw = (w + 3) / 4, nbh = (h + 3) / 4;
      int bs = 1;
      while (bs < nbw) bs <<= 1;
      for (int by = 0; by < nbh; by++)
        for (int bx = 0; bx < nbw; bx++) {
          const int lx = (c[i].ox + bx * 4) << xdec, ly = (c[i].oy + by * 4) << ydec;
          buf[by * bs + bx] =
              scales ? scales[(size_t)(ly >> 3) * scale_stride + (lx >> 3)] : (1u << 14);
        }
      out[i] = r1o_get_weighted_sse(o, org->stride, r, ref->stride, buf, bs, w, h, hbd);
    } else {
      uint64_t sum = 0;
      const int bpp = org->bytes_per_px;
      for (int y = 0; y < h; y += 8)
        for (int x = 0; x < w; x += 8) {
          const int kh = h - y < 8 ? h - y : 8, kw = w - x < 8 ? w - x : 8;
          const uint64_t v = r1o_cdef_dist_kernel(
              (const uint8_t *)o + ((size_t)y * org->stride + x) * bpp, org->stride,
              (const uint8_t *)r + ((size_t)y * ref->stride + x) * bpp, ref->stride, kw, kh,
              org->bit_depth, hbd);
          const uint64_t sc =
              scales ? scales[(size_t)((c[i].oy + y) >> 3) * scale_stride + ((c[i].ox + x) >> 3)]
                     : (1u << 14);
          sum += (sc * v + (1u << 14 >> 1)) >> 14;
        }
      out[i] = sum;
    }
  }
  return 0;
}
