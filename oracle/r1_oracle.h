/*
 * r1_oracle.h -- CPU oracle for the rav1e block-kernel hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a scalar, plain-C restatement of the
 * reference's pure-Rust kernels (RAV1E_CPU_TARGET=rust semantics).  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may link
 * or call it, and only as the checker / the timed CPU baseline.  The product
 * (rav1e_amd/csrc, librav1e_hip.so) never includes or links anything here.
 *
 * Pinning status (DESIGN.md section 4 has the table): every function a GPU test compares against
 * reproduces either known answers the reference itself holds (get_sad / get_satd: src/dist.rs:418-500;
 * 4x4 intra prediction: src/predict.rs:1523-1693; estimate_rate: src/rdo.rs:2749) or vectors made by
 * EXECUTING the reference's source text in the build container (tools/rustlite,
 * tests/golden/gen_NAME_ref.py -> tests/golden/NAME_ref.npz; tests/test_oracle_NAME_ref.py).  Exception:
 * Plane::pad / Plane::downsampled restate the un-vendored v_frame 0.3.9 crate (oracle/plane.c).
 *
 * Conventions: strides are in ELEMENTS (not bytes); `hbd` != 0 means the
 * pixel type is u16 (rav1e Pixel = u16), else u8.  Pointers address the
 * top-left sample of the block (the PlaneRegion / PlaneSlice origin).
 */
#ifndef R1_ORACLE_H
#define R1_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- dist (src/dist.rs) ---- */
uint32_t r1o_get_sad(const void *org, ptrdiff_t org_stride, const void *ref,
                     ptrdiff_t ref_stride, int w, int h, int hbd);
uint32_t r1o_get_satd(const void *org, ptrdiff_t org_stride, const void *ref,
                      ptrdiff_t ref_stride, int w, int h, int hbd);
uint64_t r1o_get_weighted_sse(const void *src1, ptrdiff_t stride1,
                              const void *src2, ptrdiff_t stride2,
                              const uint32_t *scale, size_t scale_stride,
                              int w, int h, int hbd);
uint32_t r1o_cdef_dist_kernel(const void *src, ptrdiff_t src_stride,
                              const void *dst, ptrdiff_t dst_stride, int w,
                              int h, int bit_depth, int hbd);
uint32_t r1o_apply_ssim_boost(uint32_t input, uint32_t svar, uint32_t dvar,
                              int bit_depth);
/* rdo.rs glue: cdef_dist_wxh / sse_wxh with a per-8x8 (resp. per-4x4-cell)
 * DistortionScale grid (Q14, row stride in entries). */
uint64_t r1o_cdef_dist_wxh(const void *src1, ptrdiff_t stride1,
                           const void *src2, ptrdiff_t stride2, int w, int h,
                           int bit_depth, int hbd, const uint32_t *bias8x8,
                           size_t bias_stride);

/* ---- mc (src/mc.rs) ---- */
void r1o_put_8tap(void *dst, ptrdiff_t dst_stride, const void *src,
                  ptrdiff_t src_stride, int w, int h, int col_frac,
                  int row_frac, int mode_x, int mode_y, int bit_depth,
                  int hbd);
void r1o_prep_8tap(int16_t *tmp, const void *src, ptrdiff_t src_stride, int w,
                   int h, int col_frac, int row_frac, int mode_x, int mode_y,
                   int bit_depth, int hbd);
void r1o_mc_avg(void *dst, ptrdiff_t dst_stride, const int16_t *tmp1,
                const int16_t *tmp2, int w, int h, int bit_depth, int hbd);

/* ---- forward transform (src/transform/forward*.rs) ---- */
/* tx_size / tx_type use the reference's enum values (transform/mod.rs:56-123).
 * coeff32 != 0: output is int32_t (T::Coeff for u16 pixels) else int16_t.
 * returns 0, or -1 for an invalid (tx_size, tx_type) pair (reference panics). */
int r1o_forward_transform(const int16_t *input, void *output, size_t stride,
                          int tx_size, int tx_type, int bd, int coeff32);
int r1o_tx_width(int tx_size);
int r1o_tx_height(int tx_size);
int r1o_valid_av1_transform(int tx_size, int tx_type);
/* 1-D kernels exposed for property tests. type: 0..12 = TxfmType order
 * (DCT4,DCT8,DCT16,DCT32,DCT64,ADST4,ADST8,ADST16,Id4,Id8,Id16,Id32,WHT4) */
void r1o_fwd_txfm_1d(int32_t *coeffs, int txfm_type);

/* ---- inverse transform (src/transform/inverse.rs) ---- */
/* cls: 0 DCT 1 ADST 2 FLIPADST 3 IDTX 4 WHT; returns -1 if no such kernel */
int r1o_inv_txfm_1d(int32_t *coeffs, int cls, int n, int range_bits);
/* dst holds the prediction on entry, the reconstruction on return */
int r1o_inverse_transform_add(const void *coeffs, void *dst, ptrdiff_t stride,
                              int tx_size, int tx_type, int bd, int coeff32,
                              int hbd);
int r1o_inv_txfm_add_batch(const void *coeffs, int coeff_stride, const void *pred,
                           void *rec, int n, int tx_size, int tx_type,
                           int bit_depth, int coeff_bytes, int bytes_per_px);

/* ---- quantize (src/quantize/mod.rs, src/scan_order.rs) ---- */
void r1o_gen_scan(int kind, int W, int H, uint16_t *scan);
int r1o_scan_kind(int tx_type);
int r1o_get_scan(int tx_size, int tx_type, uint16_t *scan, uint16_t *iscan);
int r1o_get_log_tx_scale(int tx_size);
uint16_t r1o_dc_q(int qindex, int delta_q, int bit_depth);
uint16_t r1o_ac_q(int qindex, int delta_q, int bit_depth);
uint32_t r1o_divu(uint32_t x, uint32_t d);
int r1o_quantize(const void *coeffs, void *qcoeffs, int tx_size, int tx_type,
                 int qindex, int bit_depth, int is_intra, int dc_delta_q,
                 int ac_delta_q, int coeff32);
void r1o_dequantize(const void *qcoeffs, void *rcoeffs, int tx_size, int qindex,
                    int bit_depth, int dc_delta_q, int ac_delta_q, int coeff32);
int r1o_quantize_batch(const void *coeffs, int coeff_stride, int n, int tx_size,
                       int tx_type, int qindex, int bit_depth, int is_intra,
                       int dc_delta_q, int ac_delta_q, int coeff_bytes,
                       void *qcoeffs, uint16_t *eobs, void *rcoeffs);

uint64_t r1o_tx_domain_distortion(const void *coeffs, const void *rcoeffs, int tx_size,
                                  int coeff32);
uint64_t r1o_estimate_rate(int qindex, int tx_size, uint64_t fast_distortion);
int r1o_quantize_rdo_batch(const void *coeffs, int coeff_stride, int n, int tx_size, int tx_type,
                           int qindex, int bit_depth, int is_intra, int dc_delta_q,
                           int ac_delta_q, int coeff_bytes, void *qcoeffs, uint16_t *eobs,
                           void *rcoeffs, uint64_t *tx_dist, uint64_t *est_rate);

/* ---- intra prediction (src/predict.rs, src/partition.rs:639-898) ---- */
int r1o_intra_mode_to_angle(int mode);
int r1o_select_ief_strength(int width, int height, int smooth, int angle_delta);
int r1o_select_ief_upsample(int width, int height, int smooth, int angle_delta);
int r1o_dispatch_predict_intra(int mode, int variant, void *dst, ptrdiff_t stride,
                               int tx_size, int bit_depth, const int16_t *ac, int angle,
                               int ief, const void *edge, int left_len, int above_len,
                               int avail_w, int avail_h, int hbd);
int r1o_predict_intra(int mode, int x, int y, void *dst, ptrdiff_t stride, int tx_size,
                      int bit_depth, const int16_t *ac, int angle_delta, int alpha, int ief,
                      const void *edge, int left_len, int above_len, int avail_w, int avail_h,
                      int hbd);
void r1o_pred_cfl_ac(int16_t *ac, const void *luma, ptrdiff_t stride, int bw, int bh,
                     int w_pad, int h_pad, int xdec, int ydec, int hbd);
void r1o_get_intra_edges(void *edge, int lens[2], const void *tile, ptrdiff_t stride, int x,
                         int y, int rect_w, int rect_h, int tx_size, int bit_depth, int mode,
                         int enable_ief, int angle_delta, int has_tr, int has_bl, int hbd);

/* ---- CDEF (src/cdef.rs) ---- */
int r1o_cdef_find_dir(const void *img, ptrdiff_t stride, uint32_t *var, int coeff_shift,
                      int hbd);
void r1o_cdef_filter_block(void *dst, ptrdiff_t dstride, const void *input, ptrdiff_t istride,
                           int pri_strength, int sec_strength, int dir, int damping,
                           int bit_depth, int xdec, int ydec, int edges, int hbd);
int r1o_cdef_adjust_strength(int strength, int var);

/* encoder.rs:1355 diff */
void r1o_diff(int16_t *dst, const void *src1, ptrdiff_t stride1,
              const void *src2, ptrdiff_t stride2, int w, int h, int hbd);

/* ---- batch drivers (oracle/batch.c): host-memory mirrors of the product's
 * batch C ABI (include/rav1e_amd.h), looping the scalar kernels above over
 * candidate lists with OpenMP.  Plane/candidate structs have the same layout
 * as R1Plane / R1DistCand / R1McCand / R1RdoCand so tests can reuse one
 * ctypes definition; `data` is a HOST pointer here. ---- */
typedef struct {
  void *data;
  int32_t stride, alloc_height, width, height, xorigin, yorigin;
  int32_t bytes_per_px, bit_depth;
} r1o_plane;
typedef struct { int16_t ox, oy, rx, ry; } r1o_dist_cand;
typedef struct { int16_t rx, ry; uint8_t col_frac, row_frac, mode_x, mode_y; } r1o_mc_cand;
typedef struct {
  int16_t ox, oy, rx, ry;
  uint8_t col_frac, row_frac, mode_x, mode_y, tx_type, reserved[3];
} r1o_rdo_cand;

void r1o_cdef_filter_tile_plane(const r1o_plane *luma, const r1o_plane *in, const r1o_plane *out,
                                int p, int xdec, int ydec, int tile_w, int tile_h,
                                const uint8_t *skip_mi, int mi_stride, int mi_cols,
                                int mi_rows, const uint8_t *cdef_index_sb, int sb_stride,
                                const uint8_t *y_strengths, const uint8_t *uv_strengths,
                                int damping, int bit_depth);
/* CDEF strength search of rdo_loop_decision (src/rdo.rs:2104-2560, CDEF leg without a
 * restoration filter; see oracle/cdef.c) */
typedef struct {
  uint8_t y_strengths[8], uv_strengths[8];
  int32_t damping, bit_depth, n_idx, planes, xdec, ydec;
  int32_t crop_w, crop_h;          /* fi.width / fi.height */
  int32_t area_sb_w, area_sb_h;    /* analysis area (largest restoration unit) in superblocks */
  uint32_t dist_scale[3];          /* fi.dist_scale[pli], Q14 */
} r1o_cdef_search_params;
int r1o_cdef_strength_search(const r1o_plane *rec, const r1o_plane *src, const uint8_t *skip_mi,
                             int mi_stride, int mi_cols, int mi_rows, const uint32_t *scales,
                             int scale_stride, const r1o_cdef_search_params *p, uint64_t *err,
                             int8_t *best);
/* lookahead cost maps (src/api/lookahead.rs:30-268) */
void r1o_estimate_intra_costs(const r1o_plane *plane, int bit_depth, uint32_t *costs);
uint64_t r1o_importance_block_difference(const r1o_plane *org, const r1o_plane *ref);
void r1o_estimate_inter_costs(const r1o_plane *org, const r1o_plane *ref, const int16_t *mvs,
                              uint32_t *costs);
void r1o_update_block_importances(const uint32_t *intra_costs, const float *future_importances,
                                  const uint32_t *inter_costs, const int16_t *mvs, int w, int h,
                                  int len, float *ref_importances);
/* hierarchical motion estimation of one tile against one reference
 * (src/me.rs:153-335, see oracle/me.c).  org3 / ref3: [full, half, quarter]
 * resolution planes; stats: FrameMEStats of this reference (in/out), prev: the
 * previous frame's (EPZS subset C) or NULL.  Pinned by me_ref.npz (executed src/me.rs text). */
typedef struct { int16_t row, col; uint32_t normalized_sad; } r1o_me_stats;
typedef struct {
  int32_t w_in_b, h_in_b;                 /* fi.w_in_b / fi.h_in_b: frame size in 4x4 units */
  int32_t stats_cols, stats_rows;         /* FrameMEStats::cols / rows */
  int32_t tile_x, tile_y, tile_w, tile_h; /* luma px; x, y multiples of 64 */
  int32_t bit_depth, allow_hp, allow_full_search, me_range_scale;
  uint32_t lambda[3];                     /* by ssdec (me.rs:175-177 evaluated by the host) */
} r1o_me_params;
int r1o_estimate_tile_motion(const r1o_plane *org3, const r1o_plane *ref3, const r1o_me_params *p,
                             r1o_me_stats *stats, const r1o_me_stats *prev);
/* estimate_motion(.., Some(pmv), corner, extensive = false, ssdec = 0, None) of
 * src/rdo.rs:1183-1196 over independent blocks of one tile: full-pel search
 * from the tile's MEStats, SATD re-cost, sub-pel diamond (me.rs:536-632).
 * bx, by: tile-relative 4x4 units; w, h: block size in px; corner: 0 = INIT,
 * else 1 | right << 1 | bottom << 2; pmv[k] = (row, col). */
typedef struct {
  int16_t bx, by;
  uint8_t w, h, corner, reserved;
  int16_t pmv[2][2];
} r1o_me_block;
typedef struct { int16_t row, col; uint32_t sad; uint64_t cost; } r1o_me_result;
int r1o_estimate_motion_batch(const r1o_plane *org3, const r1o_plane *ref3, const r1o_me_params *p,
                              const r1o_me_stats *stats, const r1o_me_stats *prev,
                              const r1o_me_block *blk, int n, int use_satd, int filter_mode,
                              r1o_me_result *out);
/* ---- deblocking filter + level search (src/deblock.rs, see oracle/deblock.c) ---- */
typedef struct {
  uint8_t tx_log2;   /* luma transform: log2(width_mi) | log2(height_mi) << 3 */
  uint8_t uvtx_log2; /* bsize.largest_chroma_tx_size(xdec, ydec), same packing */
  uint8_t n4_log2;   /* block: log2(n4_w) | log2(n4_h) << 3 */
  uint8_t flags;     /* 1 skip, 2 ref_frames[0] == INTRA_FRAME, 4 mode_type
                        (mode >= NEARESTMV && != GLOBALMV && != GLOBAL_GLOBALMV),
                        bits 3-5 ref_frames[0].to_index() */
  int8_t deltas[4];  /* deblock_deltas */
} r1o_deblock_block;
typedef struct {     /* DeblockState (src/encoder.rs) */
  uint8_t levels[4];
  uint8_t sharpness; /* carried, unused by the reference's filters (always 0) */
  uint8_t deltas_enabled, block_deltas_enabled, block_delta_shift, block_delta_multi;
  int8_t ref_deltas[8], mode_deltas[2];
  uint8_t reserved[5];
} r1o_deblock_state;
int r1o_deblock_plane(const r1o_deblock_state *d, const r1o_plane *p, int pli, int xdec, int ydec,
                      const r1o_deblock_block *blocks, int blocks_stride, int blocks_cols, int blocks_rows,
                      int crop_w, int crop_h, int bd);
int r1o_deblock_sse_plane(const r1o_plane *rec, const r1o_plane *src, int pli, int xdec, int ydec,
                          const r1o_deblock_block *blocks, int blocks_stride, int blocks_cols,
                          int blocks_rows, int crop_w, int crop_h, int bd, int64_t *v_tally,
                          int64_t *h_tally);
void r1o_deblock_pick_levels(int64_t *v_tally, int64_t *h_tally, int pli, uint8_t *out);
/* ---- loop restoration: self-guided stripe filter (src/lrf.rs, see oracle/lrf.c) ---- */
typedef struct {
  uint8_t filter; /* RESTORE_NONE 0, RESTORE_SGRPROJ 3 (src/lrf.rs:34-37) */
  uint8_t set;    /* index into SGRPROJ_PARAMS_S */
  int8_t xqd[2];
} r1o_lrf_unit;
int r1o_lrf_filter_plane(const r1o_plane *cdeffed, const r1o_plane *deblocked, const r1o_plane *out,
                         int ydec, int crop_w, int crop_h, int frame_h, int unit_size, int unit_cols,
                         int unit_rows, int stripe_height, const r1o_lrf_unit *units, int bd);
/* edges of a unit of the restoration search: which of its left / upper neighbourhood exists in the
 * area rdo_loop_decision is working on (oracle/lrf.c, setup_integral_image) */
#define R1O_SGR_EDGE_LEFT 1
#define R1O_SGR_EDGE_ABOVE 2
int r1o_lrf_search_unit(const r1o_plane *lrf_in, const r1o_plane *src, int x0, int y0, int w, int h,
                        int set, int edges, int is_chroma, int xdec, int ydec, const uint32_t *scales,
                        int scale_stride, uint32_t dist_scale, int bd, int8_t *xqd_out,
                        uint64_t *err_out);
uint64_t r1o_loop_plane_error_rect(const r1o_plane *src, const void *test, int tstride, int x0, int y0,
                                   int w, int h, int is_chroma, int xdec, int ydec, const uint32_t *scales,
                                   int scale_stride, int bd);
int r1o_sgr_filter_rect(const r1o_plane *plane, int x0, int y0, int w, int h, int set, const int8_t *xqd,
                        int edges, int bd, void *out_px);
/* loop_decision.c: rdo_loop_decision with both filters on (src/rdo.rs:2377-2560) */
typedef struct r1o_trial_unit {
  int16_t x, y, w, h;
  uint8_t set, edges;
  int8_t xqd[2];
  int32_t sb;
} r1o_trial_unit;
int r1o_cdef_apply_area(const r1o_plane *rec, const r1o_plane *out, const uint8_t *skip_mi, int mi_stride,
                        int mi_cols, int mi_rows, const r1o_cdef_search_params *p, const int8_t *index_sb);
int r1o_cdef_lrf_trial(const r1o_plane *rec, const r1o_plane *work, const r1o_plane *src, const uint8_t *skip_mi,
                       int mi_stride, int mi_cols, int mi_rows, const uint32_t *scales, int scale_stride,
                       const r1o_cdef_search_params *p, const r1o_trial_unit *units, const int32_t *n_units,
                       const uint8_t *sb_sel, uint64_t *err, uint64_t *err_planes, int8_t *best);
void r1o_sgrproj_solve(const r1o_plane *cdeffed, const r1o_plane *input, int x0, int y0, int w, int h,
                       int set, int edges, int bd, int8_t *xqd_out);
void r1o_activity_scales(const r1o_plane *luma, uint32_t *variances, uint32_t *scales);
/* plane.c: v_frame 0.3.9 Plane::pad / Plane::downsampled (see the header of plane.c) */
void r1o_plane_pad(const r1o_plane *p, int w, int h, int xdec, int ydec);
int r1o_plane_downsample(const r1o_plane *src, const r1o_plane *dst, int frame_w, int frame_h,
                         int dst_xdec, int dst_ydec);
void r1o_set_threads(int n);
/* mc.c: get_filter (src/mc.rs:238-247) for fast_cand.c */
const int16_t *r1o_get_filter(int mode, int frac, int length);
/* fast_cand.c: the compiler-vectorised CPU-baseline leg of bench.py (square blocks, DCT_DCT) */
int r1o_fast_rdo_cand_batch(const r1o_plane *org, const r1o_plane *ref, int n_px, int tx_size,
                            const r1o_rdo_cand *c, int n, int threads, uint32_t *sad_out,
                            uint32_t *satd_out, void *coeffs);
int r1o_dist_batch(int kind, const r1o_plane *org, const r1o_plane *ref, int w,
                   int h, const r1o_dist_cand *c, int n, uint32_t *out);
int r1o_dist_scaled_batch(int kind, const r1o_plane *org, const r1o_plane *ref,
                          int w, int h, const r1o_dist_cand *c, int n,
                          const uint32_t *scales, int scale_stride, int xdec,
                          int ydec, uint64_t *out);
int r1o_fwd_txfm_batch(const int16_t *residual, void *coeffs, int n,
                       int tx_size, int tx_type, int bit_depth, int coeff_bytes);
int r1o_mc_put_batch(const r1o_plane *ref, int w, int h, const r1o_mc_cand *c,
                     int n, void *dst);
int r1o_mc_prep_batch(const r1o_plane *ref, int w, int h, const r1o_mc_cand *c,
                      int n, int16_t *tmp);
int r1o_mc_avg_batch(const int16_t *t1, const int16_t *t2, int w, int h, int n,
                     int bit_depth, int bytes_per_px, void *dst);
int r1o_rdo_cand_batch(const r1o_plane *org, const r1o_plane *ref, int w, int h,
                       int tx_size, const r1o_rdo_cand *c, int n,
                       uint32_t *sad_out, uint32_t *satd_out, void *coeffs,
                       void *pred_out);
int r1o_rdo_full_cand_batch(const r1o_plane *org, const r1o_plane *ref, int w, int h,
                            int tx_size, const r1o_rdo_cand *c, int n, int qindex,
                            int is_intra, int dc_delta_q, int ac_delta_q,
                            uint32_t *sad_out, uint32_t *satd_out, uint16_t *eob_out,
                            uint64_t *tx_dist_out, uint64_t *est_rate_out,
                            void *qcoeffs_out);
int r1o_rdo_pixel_cand_batch(const r1o_plane *org, const r1o_plane *ref, int w, int h,
                             int tx_size, const r1o_rdo_cand *c, int n, int qindex,
                             int is_intra, int dc_delta_q, int ac_delta_q, int kind,
                             const uint32_t *scales, int scale_stride, int xdec, int ydec,
                             uint32_t *sad_out, uint32_t *satd_out, uint16_t *eob_out,
                             uint64_t *dist_out, void *qcoeffs_out, void *rec_out,
                             const void *pred_in);
uint32_t r1o_tx_type_mask(int tx_size, int is_inter, int use_reduced_set, int rav1e_only);
int r1o_rdo_txsearch_batch(const r1o_plane *org, const r1o_plane *ref, const void *pred_in,
                           int w, int h, int tx_size, const r1o_rdo_cand *c, int n,
                           uint32_t tx_type_mask, int qindex, int is_intra, int dc_delta_q,
                           int ac_delta_q, int kind, const uint32_t *scales, int scale_stride,
                           int xdec, int ydec, uint32_t *sad_out, uint32_t *satd_out,
                           uint16_t *eob_out, uint64_t *dist_out, uint64_t *est_rate_out,
                           void *qcoeffs_out, void *rec_out);

#ifdef __cplusplus
}
#endif
#endif
