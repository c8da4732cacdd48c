/*
 * rav1e_amd.h -- C ABI of librav1e_hip.so, the MI355X (gfx950) block-kernel
 * backend for rav1e's per-block RDO inner loop.
 *
 * This is the drop-in boundary: exactly what a new `CpuFeatureLevel::HIP`
 * row of rav1e's `src/asm` dispatch tables would bind over FFI
 * (reference dispatch surface: src/cpu_features/x86.rs:97-158; per-family
 * tables cited at each entry point below).  Two layers:
 *
 *  (1) BATCH API (the product).  The reference calls its kernels one block at
 *      a time (src/me.rs:1445-1454, src/rdo.rs:1328-1352, src/encoder.rs:
 *      1404-1661); a GPU wants thousands of independent candidates per
 *      launch, so the host enqueues descriptor arrays and flushes.  All
 *      pointers are DEVICE pointers unless stated; the caller owns every
 *      buffer; nothing is retained after a call returns; calls are
 *      asynchronous on `stream` (a hipStream_t passed as void*, NULL = the
 *      default stream).  Return value: 0 ok, negative R1_E* on error; results
 *      are undefined on error.  Thread-safe: no global mutable state (the one
 *      per-context resource, the job-descriptor ring of the tile ME, is
 *      serialised internally).
 *
 *  (2) PER-CALL COMPAT SHIMS with the reference's exact asm signatures
 *      (host pointers, byte strides, values returned directly).  They stage,
 *      launch and synchronise -- for plumbing / `check_asm`-style parity
 *      only, never fast.
 *
 * Enum integer values are the reference's (they index its dispatch tables):
 *   BlockSize  src/partition.rs:130-153      TxSize  src/transform/mod.rs:101-123
 *   TxType     src/transform/mod.rs:56-74    FilterMode  src/mc.rs:100-106
 */
#ifndef RAV1E_AMD_H
#define RAV1E_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define R1_OK 0
#define R1_EINVAL (-1)   /* bad argument / descriptor (reference: assert! panic) */
#define R1_EHIP (-2)     /* HIP runtime error; see r1_last_error() */
#define R1_ECOMM (-3)    /* RCCL error (or no RCCL library to load); see r1_last_error() */
#define R1_ENOMEM (-4)
#define R1_ETIMEDOUT (-5) /* r1_me_status: a persistent tile-ME launch flagged a timed-out dependency wait */

/* ---- Plane<T> view (v_frame 0.3.9 PlaneConfig layout; reference use:
 * src/tiling/plane_region.rs:185 -- element (x,y) lives at
 * data[(yorigin + y) * stride + xorigin + x]).  `data` is a device pointer to
 * the start of the allocation; stride is in ELEMENTS. */
typedef struct R1Plane {
  void *data;
  int32_t stride;
  int32_t alloc_height;
  int32_t width, height;     /* visible size */
  int32_t xorigin, yorigin;  /* padding left / above pixel (0,0) */
  int32_t bytes_per_px;      /* 1 (Pixel = u8) or 2 (Pixel = u16) */
  int32_t bit_depth;         /* 8, 10 or 12 */
} R1Plane;

/* rav1e FilterMode (src/mc.rs:100-106) */
enum { R1_FILTER_REGULAR = 0, R1_FILTER_SMOOTH = 1, R1_FILTER_SHARP = 2,
       R1_FILTER_BILINEAR = 3 };

/* distortion kinds of r1_dist_batch */
enum { R1_DIST_SAD = 0, R1_DIST_SATD = 1,
       /* r1_dist_scaled_batch: */ R1_DIST_WSSE = 2, R1_DIST_CDEF = 3 };

/* One distortion candidate: block of the launch's (w,h) at (ox,oy) in the
 * org plane against (rx,ry) in the ref plane (full-pel, plane coordinates;
 * may lie in the padding as the reference's MV clamp allows,
 * src/me.rs:339-362). */
typedef struct R1DistCand {
  int16_t ox, oy, rx, ry;
} R1DistCand;

/* One motion-compensation candidate: full-pel position of the block in the
 * ref plane plus 1/16-pel fractions, i.e. the outputs of get_mv_params
 * (src/predict.rs:284-297). */
typedef struct R1McCand {
  int16_t rx, ry;
  uint8_t col_frac, row_frac; /* 0..15 */
  uint8_t mode_x, mode_y;     /* FilterMode */
} R1McCand;

/* One fused RDO candidate (mc -> dist -> diff -> forward transform). */
typedef struct R1RdoCand {
  int16_t ox, oy;             /* block position in the org (source) plane */
  int16_t rx, ry;             /* full-pel position in the ref plane */
  uint8_t col_frac, row_frac; /* 0..15 */
  uint8_t mode_x, mode_y;     /* FilterMode */
  uint8_t tx_type;            /* TxType for the residual transform */
  uint8_t reserved[3];
} R1RdoCand;

typedef struct r1_ctx r1_ctx;

/* ---- context ---- */
int r1_ctx_create(int device, r1_ctx **out);
void r1_ctx_destroy(r1_ctx *ctx);
const char *r1_last_error(void);
/* ABI version, bumped on any incompatible change.
 * 4 (round 4): r1_estimate_tile_motion_batch refuses with R1_ETIMEDOUT while a flagged persistent launch
 *    has not been acknowledged through r1_me_status; r1_cdef_filter_frame_plane checks luma->bit_depth
 *    against params->bit_depth and the _dirs variant wants 8-aligned tile_w / tile_h; skip_mi bytes are
 *    bools (any non-zero value) in the CDEF filter and the strength search alike.
 * 5 (round 5): + r1_rdo_txsearch_batch, r1_tx_type_mask (additions only; nothing of 4 changed).
 * 6 (round 5): R1SgrSolveUnit.reserved[0] became `edges` (R1_SGR_EDGE_*): what r1_sgrproj_solve_batch /
 *    r1_lrf_search_batch see left of / above a unit is the caller's statement of the unit's place in its
 *    rdo_loop_decision area, no longer the unit's place in the frame (0 = nothing, the case of one unit per
 *    plane and area); the restoration entry points refuse planes of 4 GiB and more.  MIGRATION from 5: a
 *    caller that left reserved = 0 keeps its results only where a plane has ONE unit per area; with several
 *    units per area (64-pixel luma units under 128-pixel chroma units, qindex > 160) it must now set `edges`
 *     -- 0 means "the unit alone", no longer "as the unit lies in the frame".
 * 7 (round 6): + r1_cdef_lrf_trial_batch, r1_cdef_lrf_trial_scratch_bytes, r1_cdef_apply_area, R1TrialUnit,
 *    r1_comm_plane_pool_open / _close (additions only; nothing of 6 changed).
 * Additions never changed an existing signature: the peer-store entry points (r1_comm_push_*, round 4) and
 * r1_comm_push_frame (round 5) were added under versions 4 and 5 respectively. */
int r1_abi_version(void);

/* ---- dist:: (reference: src/dist.rs get_sad 31, get_satd 156; dispatch
 * tables SAD_FNS/SATD_FNS/_HBD src/asm/x86/dist/mod.rs:483-729).
 * out[i] = get_sad/get_satd(org@cand[i], ref@cand[i], w, h).  w,h <= 128 and
 * multiples of 4 (SATD uses 4x4 Hadamards when min(w,h)==4 else 8x8). */
int r1_dist_batch(r1_ctx *ctx, int kind, const R1Plane *org,
                  const R1Plane *ref, int w, int h, const R1DistCand *cands,
                  int n, uint32_t *out, void *stream);

/* ---- candidate-level pixel-domain distortion with the DistortionScale bias
 * (reference: sse_wxh src/rdo.rs:177-224 -> get_weighted_sse src/dist.rs:234-283
 * [kind R1_DIST_WSSE]; cdef_dist_wxh src/rdo.rs:142-173 -> cdef_dist_kernel
 * src/dist.rs:302-372 + apply_ssim_boost src/activity.rs:159-186
 * [kind R1_DIST_CDEF]; x86 tables src/asm/x86/dist/{sse,cdef_dist}.rs).
 * scales: the frame's distortion_scales grid (src/rdo.rs:443-459), one Q14
 * u32 per 8x8 LUMA importance block, entry [(luma_y >> 3) * scale_stride +
 * (luma_x >> 3)] where luma_xy = (plane_xy << dec) of each 4x4 cell (WSSE) or
 * 8x8 kernel (CDEF) -- exactly what the compute_bias closure at
 * src/rdo.rs:283-303 looks up.  NULL = DistortionScale::default() (1 << 14).
 * out[i] = the reference's Distortion (u64), before `* fi.dist_scale[p]`.
 * w, h <= 128: the visible block size after frame clipping (clip_visible_bsize, src/rdo.rs:228-251),
 * any size >= 1.  R1_DIST_CDEF: the last kernel of a row / column is kernel_w x kernel_h < 8x8.
 * R1_DIST_WSSE: -- like get_weighted_sse, only whole 4x4 cells are measured (a
 * clipped chroma block may be 2 wide: its distortion is 0, as in the reference). */
int r1_dist_scaled_batch(r1_ctx *ctx, int kind, const R1Plane *org,
                         const R1Plane *ref, int w, int h,
                         const R1DistCand *cands, int n, const uint32_t *scales,
                         int scale_stride, int xdec, int ydec, uint64_t *out,
                         void *stream);

/* ---- transform::forward (reference: src/transform/forward.rs:71-161;
 * x86 entry src/asm/x86/transform/forward.rs:444-447).
 * residual: n dense blocks of w*h int16 (row-major, stride = tx width, the
 * only call site's layout, src/encoder.rs:1544-1552).
 * coeffs: n dense blocks of w*h coefficients in the reference's transposed,
 * 32x32-chunked order; int16 when coeff_bytes == 2 (Pixel = u8) or int32
 * when 4 (Pixel = u16).  Invalid (tx_size, tx_type) -> R1_EINVAL. */
int r1_fwd_txfm_batch(r1_ctx *ctx, const int16_t *residual, void *coeffs,
                      int n, int tx_size, int tx_type, int bit_depth,
                      int coeff_bytes, void *stream);

/* ---- transform::inverse (reference: inverse_transform_add,
 * src/transform/inverse.rs:1633-1705; InvTxfmFunc tables
 * src/asm/x86/transform/inverse.rs, wrapper src/asm/shared/transform/inverse.rs:30-36).
 * coeffs: block i at coeffs + i*coeff_stride (elements); only the first
 * min(w,32)*min(h,32) entries are read, in the forward transform's transposed
 * order (so the output of r1_fwd_txfm_batch / r1_quantize_batch's rcoeffs can
 * be passed unchanged).  int16 when bytes_per_px == 1, int32 when 2
 * (T::Coeff).  pred / rec: n dense w*h pixel blocks; rec may alias pred (the
 * reference adds in place). */
int r1_inv_txfm_add_batch(r1_ctx *ctx, const void *coeffs, int coeff_stride,
                          const void *pred, void *rec, int n, int tx_size,
                          int tx_type, int bit_depth, int bytes_per_px,
                          void *stream);

/* ---- quantize:: (reference: QuantizationContext::{update,quantize}
 * src/quantize/mod.rs:219-355, dequantize 363-384 with its dispatch
 * src/asm/x86/quantize.rs; scan orders src/scan_order.rs). */
typedef struct R1QuantParams {
  uint8_t qindex;       /* base_q_idx of the block */
  uint8_t bit_depth;    /* 8, 10, 12 */
  uint8_t is_intra;     /* selects the rounding biases (mod.rs:258-265) */
  int8_t dc_delta_q, ac_delta_q;
  uint8_t reserved[3];
} R1QuantParams;
/* block i reads coeffs + i*coeff_stride (>= coded area entries, the layout
 * r1_fwd_txfm_batch writes); qcoeffs / rcoeffs: dense coded-area blocks
 * (min(w,32)*min(h,32)), fully written (zeros beyond eob); eobs[i] = the
 * reference's return value.  rcoeffs may be NULL (no fused dequantize).
 * coeff_bytes 2 (T::Coeff = i16) or 4 (i32).  tx_type 16 (WHT) -> R1_EINVAL
 * (the reference's scan table has 16 columns). */
int r1_quantize_batch(r1_ctx *ctx, const void *coeffs, int coeff_stride, int n,
                      int tx_size, int tx_type, const R1QuantParams *params,
                      int coeff_bytes, void *qcoeffs, uint16_t *eobs,
                      void *rcoeffs, void *stream);
/* The RDO form of the same call (SURVEY.md 8f "N4", first step): additionally the
 * transform-domain distortion of encode_tx_block (src/encoder.rs:1616-1640;
 * coeffs must then hold the full w*h forward-transform output, coeff_stride >=
 * w*h) and, when est_rate is non-NULL, estimate_rate(qindex, tx_size, tx_dist)
 * from RDO_RATE_TABLE (src/rdo.rs:127-139) -- what RDOType::TxDistEstRate
 * uses to rank transform candidates without the entropy coder. */
int r1_quantize_rdo_batch(r1_ctx *ctx, const void *coeffs, int coeff_stride, int n,
                          int tx_size, int tx_type, const R1QuantParams *params,
                          int coeff_bytes, void *qcoeffs, uint16_t *eobs,
                          void *rcoeffs, uint64_t *tx_dist, uint64_t *est_rate,
                          void *stream);
int r1_dequantize_batch(r1_ctx *ctx, const void *qcoeffs, int n, int tx_size,
                        const R1QuantParams *params, int coeff_bytes,
                        void *rcoeffs, void *stream);

/* ---- mc:: (reference: src/mc.rs put_8tap 250, prep_8tap 360, mc_avg 454;
 * dispatch tables PUT_FNS/PREP_FNS/AVG_FNS src/asm/x86/mc.rs:17-78,371-382).
 * put: dst = n dense w*h pixel blocks (same pixel type as `ref`).
 * prep: tmp = n dense w*h int16 blocks.  avg: dst from two prep outputs.
 * w power of two in 2..128 (this ABI: >= 4), h even. */
int r1_mc_put_batch(r1_ctx *ctx, const R1Plane *ref, int w, int h,
                    const R1McCand *cands, int n, void *dst, void *stream);
int r1_mc_prep_batch(r1_ctx *ctx, const R1Plane *ref, int w, int h,
                     const R1McCand *cands, int n, int16_t *tmp, void *stream);
int r1_mc_avg_batch(r1_ctx *ctx, const int16_t *tmp1, const int16_t *tmp2,
                    int w, int h, int n, int bit_depth, int bytes_per_px,
                    void *dst, void *stream);
/* The same put_8tap / prep_8tap (prep != 0) with the horizontal 8-tap pass on the matrix
 * cores (banded-Toeplitz v_mfma_i32_16x16x32_i8; csrc/mc_mfma.hip): 8-bit planes, square
 * blocks 8 / 16 / 32 / 64.  Bit-identical to r1_mc_put_batch / r1_mc_prep_batch; kept as a
 * separate entry point so that both formulations can be timed side by side
 * (tools/bench_mc_mfma.py).  Other sizes / bit depths: R1_EINVAL. */
int r1_mc_batch_mfma(r1_ctx *ctx, int prep, const R1Plane *ref, int w, int h,
                     const R1McCand *cands, int n, void *dst, void *stream);

/* ---- predict:: (reference: get_intra_edges src/partition.rs:639-898,
 * PredictionMode::predict_intra src/predict.rs:205-249, dispatch_predict_intra
 * 705-784 with its kernels 786-1505, pred_cfl_ac 1020-1063; x86 dispatch
 * src/asm/x86/predict.rs).  PredictionMode values 0..13 are the reference's
 * (src/predict.rs:75-89). */
#define R1_INTRA_EDGE_LEN 257  /* IntraEdgeBuffer: 4 * MAX_TX_SIZE + 1 (partition.rs:600) */

/* Inputs of get_intra_edges for one transform block.  (x, y): position
 * relative to the tile (`po`).  flags: bit0 enable_intra_edge_filter, bit1 =
 * has_top_right(..), bit2 = has_bottom_left(..) -- those two are decisions on
 * the partition tree (encoder state, src/partition.rs:400-560) and stay on
 * the host.  mode < 0 = None (every edge is needed). */
typedef struct R1IntraEdgeCand {
  int16_t x, y;
  int8_t mode;
  int8_t angle_delta;   /* IntraParam::AngleDelta, else 0 */
  uint8_t flags;
  uint8_t reserved;
} R1IntraEdgeCand;
/* edges: n buffers of `edge_stride` (>= 257) pixels in the reference's layout:
 * left right-aligned ending at index 128 (bottom -> top), top-left at 128,
 * above from 129.  lens[2i], lens[2i+1] = init_left, init_above of
 * IntraEdge::new.  Tile = (tile_x, tile_y, tile_w, tile_h) in plane pixels. */
int r1_intra_edges_batch(r1_ctx *ctx, const R1Plane *rec, int tile_x, int tile_y,
                         int tile_w, int tile_h, int tx_size,
                         const R1IntraEdgeCand *cands, int n, void *edges,
                         int edge_stride, uint8_t *lens, void *stream);

/* Arguments of dispatch_predict_intra for one block (after predict_intra's
 * PAETH / CFL remaps, which depend only on the block position). */
typedef struct R1IntraCand {
  uint8_t mode;        /* PredictionMode 0..13 */
  uint8_t variant;     /* PredictionVariant: 0 NONE, 1 LEFT, 2 TOP, 3 BOTH */
  int16_t angle;       /* p_angle; alpha for UV_CFL_PRED */
  uint8_t ief;         /* ief_params: 0 None, 1 Some(no smooth neighbour), 2 Some(smooth) */
  uint8_t avail_w;     /* min(tx width, plane width - block x)  (predict.rs:1346) */
  uint8_t avail_h;     /* min(tx height, plane height - block y) */
  uint8_t reserved;
} R1IntraCand;
/* dst: n dense w*h pixel blocks.  ac: n dense w*h int16 (CFL only, else NULL). */
int r1_predict_intra_batch(r1_ctx *ctx, int tx_size, const R1IntraCand *cands, int n,
                           const void *edges, int edge_stride, const uint8_t *lens,
                           const int16_t *ac, int bit_depth, int bytes_per_px,
                           void *dst, void *stream);

/* pred_cfl_ac: (x, y) = luma position of the block in plane pixels; bw x bh =
 * chroma block size; w_pad / h_pad in 4-pixel units (luma_ac, predict.rs:667-702). */
typedef struct R1CflAcCand {
  int16_t x, y;
  uint8_t w_pad, h_pad;
  uint8_t reserved[2];
} R1CflAcCand;
int r1_cfl_ac_batch(r1_ctx *ctx, const R1Plane *luma, int bw, int bh, int xdec,
                    int ydec, const R1CflAcCand *cands, int n, int16_t *ac,
                    void *stream);

/* rdo_cfl_alpha (src/rdo.rs:1593-1688) for ONE chroma plane: the alpha in
 * -16 .. 16 minimising the SSE of UV_CFL_PRED against the source over the
 * visible part of the block, with the reference's search order and early exit.
 * (x, y): block position in `src` (the chroma INPUT plane); variant: the
 * PredictionVariant of the DC average (0 NONE, 1 LEFT, 2 TOP, 3 BOTH); vis_w /
 * vis_h: clip_visible_bsize.  edges / lens: r1_intra_edges_batch of the
 * reconstructed chroma plane (mode UV_CFL_PRED), ac: r1_cfl_ac_batch, one
 * entry per candidate.  alpha_out: int16 per candidate; cost_out (optional):
 * its SSE. */
typedef struct R1CflAlphaCand {
  int16_t x, y;
  uint8_t variant, vis_w, vis_h, reserved;
} R1CflAlphaCand;
int r1_cfl_alpha_search_batch(r1_ctx *ctx, const R1Plane *src, int tx_size,
                              const R1CflAlphaCand *cands, int n, const void *edges,
                              int edge_stride, const uint8_t *lens, const int16_t *ac,
                              int16_t *alpha_out, uint64_t *cost_out, void *stream);

/* ---- cdef:: (reference: cdef_find_dir src/cdef.rs:84-143, cdef_filter_block
 * 198-298, cdef_filter_superblock / cdef_filter_tile 405-625; x86 dispatch
 * src/asm/x86/cdef.rs:83-110).  Edge flags = the reference's CDEF_HAVE_*. */
enum { R1_CDEF_HAVE_LEFT = 1, R1_CDEF_HAVE_RIGHT = 2, R1_CDEF_HAVE_TOP = 4,
       R1_CDEF_HAVE_BOTTOM = 8, R1_CDEF_HAVE_ALL = 15 };
typedef struct R1CdefDirCand { int16_t x, y; } R1CdefDirCand;   /* 8x8 luma block, plane px */
int r1_cdef_find_dir_batch(r1_ctx *ctx, const R1Plane *luma, const R1CdefDirCand *cands,
                           int n, uint8_t *dir_out, int32_t *var_out, void *stream);
/* One cdef_filter_block call: block at (x, y) of plane `in` (size (8>>xdec) x
 * (8>>ydec)), written to the same position of `out` (a different plane). */
typedef struct R1CdefBlockCand {
  int16_t x, y;
  int16_t pri_strength, sec_strength;  /* already shifted / adjusted, as passed to the asm */
  uint8_t dir, damping, edges, reserved;
} R1CdefBlockCand;
int r1_cdef_filter_block_batch(r1_ctx *ctx, const R1Plane *in, const R1Plane *out,
                               int xdec, int ydec, const R1CdefBlockCand *cands, int n,
                               void *stream);
/* cdef_filter_tile for plane p of the whole frame (the reference's only call,
 * src/encoder.rs:3301-3321): skip test over the four 4x4 blocks of every 8x8,
 * direction search on luma, adjust_strength, chroma direction map, edge
 * flags, filter or copy.  skip_mi: Block::skip per 4x4 luma unit (TileBlocks), a bool per byte
 * (any non-zero value = skipped; the strength search reads it the same way);
 * cdef_index_sb: per 64x64 superblock; params = the FrameInvariants fields
 * cdef_filter_superblock reads.  luma->bit_depth must equal params->bit_depth (R1_EINVAL). */
typedef struct R1CdefParams {
  uint8_t y_strengths[8], uv_strengths[8];   /* fi.cdef_y_strengths / cdef_uv_strengths */
  uint8_t damping;                           /* fi.cdef_damping */
  uint8_t bit_depth;
  uint8_t reserved[2];
} R1CdefParams;
int r1_cdef_filter_frame_plane(r1_ctx *ctx, const R1Plane *luma, const R1Plane *in,
                               const R1Plane *out, int p, int xdec, int ydec,
                               int tile_w, int tile_h, const uint8_t *skip_mi,
                               int mi_stride, int mi_cols, int mi_rows,
                               const uint8_t *cdef_index_sb, int sb_stride,
                               const R1CdefParams *params, void *stream);

/* The same in two steps, as the reference does it per superblock (cdef_analyze_superblock once,
 * src/cdef.rs:340-373, then cdef_filter_superblock for each plane, 405-560): the analysis
 * writes (dir, var) of every 8x8 luma block, raster order over the grid of
 * nbx = 8*ceil(tile_w/64) by nby = 8*ceil(tile_h/64) blocks (r1_cdef_analyze_blocks() entries;
 * blocks outside mi_cols x mi_rows are not written; skipped blocks are analysed too, their
 * entries are never read), and three filter calls share it.  r1_cdef_filter_frame_plane is
 * analysis + one plane with stream-ordered scratch.  tile_w / tile_h here = the luma PLANE size as
 * v_frame pads it (multiples of 8; r1_cdef_filter_frame_plane_dirs has no luma plane to take the
 * picture limits from and rejects anything else with R1_EINVAL). */
long long r1_cdef_analyze_blocks(int tile_w, int tile_h);
int r1_cdef_analyze_frame(r1_ctx *ctx, const R1Plane *luma, int tile_w, int tile_h,
                          int mi_cols, int mi_rows, uint8_t *dir_out, int32_t *var_out,
                          void *stream);
int r1_cdef_filter_frame_plane_dirs(r1_ctx *ctx, const uint8_t *dirs, const int32_t *vars,
                                    const R1Plane *in, const R1Plane *out, int p, int xdec,
                                    int ydec, int tile_w, int tile_h, const uint8_t *skip_mi,
                                    int mi_stride, int mi_cols, int mi_rows,
                                    const uint8_t *cdef_index_sb, int sb_stride,
                                    const R1CdefParams *params, void *stream);

/* CDEF strength search: the CDEF leg of rdo_loop_decision (src/rdo.rs:2104-2560) when no
 * restoration filter is in play (RestorationFilter::None / no restoration unit,
 * rdo.rs:2432-2451, 2504-2520).  The reference cuts the deblocked reconstruction into
 * analysis areas of area_sb_w x area_sb_h superblocks (the largest restoration unit; 1 x 1
 * with restoration off), takes a scratch copy of each (rdo.rs:2277-2284: the AREA's borders are
 * picture edges for CDEF) and, for every superblock that is not completely skipped, tries
 * cdef_index 0 .. n_idx-1: cdef_filter_superblock (src/cdef.rs:405-560), then
 * rdo_loop_plane_error (rdo.rs:2027-2093: cdef_dist_kernel * bias per 8x8 luma block, sse_wxh
 * with the bias per chroma block, each plane's sum * fi.dist_scale[pli]) and keeps the first
 * index of smallest compute_rd_cost(rate 0, err).  One launch evaluates every (superblock,
 * index) of the frame; no filtered plane is written.
 *   rec / src: `planes` whole-frame planes (rec deblocked; areas at multiples of the area
 *   size), skip_mi: Block::skip per 4x4 luma unit, scales: coded_frame_data.distortion_scales
 *   (Q14, per 8x8 luma block; NULL = DistortionScale::default()).
 *   err_out: [n_sby][n_sbx][8] ScaledDistortion (0 for skipped superblocks / unused indices),
 *   best_out: [n_sby][n_sbx], -1 = skipped; n_sb* = ceil(mi / 16).  scratch: DEVICE,
 *   r1_cdef_strength_search_scratch_bytes() bytes (zeroed here).  All pointers DEVICE. */
typedef struct R1CdefSearchParams {
  uint8_t y_strengths[8], uv_strengths[8];   /* fi.cdef_y_strengths / cdef_uv_strengths */
  int32_t damping, bit_depth;                /* fi.cdef_damping, fi.sequence.bit_depth */
  int32_t n_idx;                             /* 1 << fi.cdef_bits */
  int32_t planes;                            /* 1 (Cs400) or 3 */
  int32_t xdec, ydec;                        /* chroma decimation: (1,1), (1,0) or (0,0) */
  int32_t crop_w, crop_h;                    /* fi.width, fi.height */
  int32_t area_sb_w, area_sb_h;
  uint32_t dist_scale[3];                    /* fi.dist_scale[pli] (DistortionScale, Q14) */
} R1CdefSearchParams;
long long r1_cdef_strength_search_scratch_bytes(int mi_cols, int mi_rows);
int r1_cdef_strength_search(r1_ctx *ctx, const R1Plane *rec, const R1Plane *src,
                            const uint8_t *skip_mi, int mi_stride, int mi_cols, int mi_rows,
                            const uint32_t *scales, int scale_stride,
                            const R1CdefSearchParams *params, uint64_t *err_out,
                            int8_t *best_out, void *scratch, void *stream);

/* ---- rdo_loop_decision with BOTH filters on: the later passes of its CDEF leg, and the working copy
 * (src/rdo.rs:2366-2574; BASELINE configs[3]: speed 4 enables cdef and lrf,
 * src/api/config/speedsettings.rs:78-79,168-171).  The reference alternates the two legs until no
 * choice changes.  From the second pass on a restoration unit may hold a self-guided choice, and every
 * CDEF trial of a superblock under it then goes (rdo.rs:2407-2530)
 *   cdef_filter_superblock(index)  ->  per plane: setup_integral_image on THAT SUPERBLOCK of the working
 *   copy (crop = the superblock: hard-clipped right and below; 4 columns left / 2 rows above from the
 *   working copy when the superblock is not first in its area) -> sgrproj_stripe_filter(set, xqd of the
 *   unit's current choice) -> rdo_loop_plane_error of the RESTORED superblock,
 * planes without such a choice as in the first pass (error of the CDEF output itself).
 *
 * r1_cdef_lrf_trial_batch: ONE call per pass for every (superblock, index) of the frame.
 *   rec / src / skip_mi / scales / params / err_out / best_out: as r1_cdef_strength_search (with no
 *     units the two calls return the same numbers);
 *   units (DEVICE): the superblocks whose restoration unit holds a self-guided choice -- n_units[0]
 *     luma entries, then n_units[1] of U, then n_units[2] of V (n_units: HOST, 3 ints).  (x, y, w, h):
 *     the superblock's visible rectangle in pixels of that plane (w = vis_width, h = vis_height,
 *     rdo.rs:2415-2428; multiples of 8 >> dec), set / xqd: the choice, edges: R1_SGR_EDGE_LEFT when the
 *     superblock is not in column 0 of its area, R1_SGR_EDGE_ABOVE when not in row 0, sb = fby * n_sbx + fbx;
 *   cdef_cur: the area working copies as r1_cdef_apply_area leaves them (read only where an edge flag is
 *     set; any valid planes otherwise);
 *   sb_sel (DEVICE, n_sb bytes, NULL = all): superblocks to evaluate.  In an area of several
 *     superblocks a trial reads its left / upper neighbours' CURRENT output, which the same pass may
 *     just have changed: such a host walks the area positions in raster order -- one call per position
 *     with that position selected, r1_cdef_apply_area in between; areas of one superblock (64-pixel
 *     luma units, every qindex <= 160) need one call;
 *   err_planes_out (optional): [n_sb][8][3] the per-plane ScaledDistortion terms of err_out;
 *   scratch: r1_cdef_lrf_trial_scratch_bytes() bytes -- it holds the trial output of every index as
 *     whole planes (n_idx x frame), which the restoration trial reads back.
 * The rate of a trial is the same for every index of a superblock (rdo.rs:2444-2456, 2499-2503): the
 * host adds it to err_out before comparing costs when it wants the reference's f64 rounding;
 * best_out compares the errors alone.
 *
 * r1_cdef_apply_area: cdef_filter_superblock with index_sb[sb] for every superblock into `out` -- the
 * CDEF working copy of every area at once (rdo.rs:2546-2560: "keep cdef output up to date"), the input
 * of the restoration leg (rdo.rs:2575-2582).  Superblocks with index < 0 or completely skipped, and
 * skipped 8x8 blocks, are copied from rec.  Areas' borders are picture edges as in the search.  Only
 * pixels of 8x8 blocks inside the block grid are written.  scratch: r1_cdef_strength_search_scratch_bytes(). */
typedef struct R1TrialUnit {
  int16_t x, y, w, h;
  uint8_t set, edges;
  int8_t xqd[2];
  int32_t sb;
} R1TrialUnit;
long long r1_cdef_lrf_trial_scratch_bytes(int mi_cols, int mi_rows, int xdec, int ydec,
                                          int bytes_per_px, int n_idx, int planes);
int r1_cdef_lrf_trial_batch(r1_ctx *ctx, const R1Plane *rec, const R1Plane *cdef_cur, const R1Plane *src,
                            const uint8_t *skip_mi, int mi_stride, int mi_cols, int mi_rows,
                            const uint32_t *scales, int scale_stride, const R1CdefSearchParams *params,
                            const R1TrialUnit *units, const int32_t *n_units, const uint8_t *sb_sel,
                            uint64_t *err_out, uint64_t *err_planes_out, int8_t *best_out, void *scratch,
                            void *stream);
int r1_cdef_apply_area(r1_ctx *ctx, const R1Plane *rec, const R1Plane *out, const uint8_t *skip_mi,
                       int mi_stride, int mi_cols, int mi_rows, const R1CdefSearchParams *params,
                       const int8_t *index_sb, void *scratch, void *stream);

/* ---- lookahead cost maps (SURVEY.md 8f "N1"; reference:
 * estimate_intra_costs src/api/lookahead.rs:30-123,
 * estimate_importance_block_difference 125-180, the SATD map of
 * estimate_inter_costs 226-268).  One launch per frame; the 8x8 importance
 * blocks are independent.  costs: (height/8) * (width/8) u32, row-major.
 * mvs: (row, col) int16 pairs in 1/8 pel per importance block
 * (stats[y*2][x*2].mv; the motion search itself is not part of this call).
 * sum_out: device u64 = sum over blocks of |mean(org) - mean(ref)| (the
 * reference divides by the block count in f64). */
int r1_estimate_intra_costs(r1_ctx *ctx, const R1Plane *luma, uint32_t *costs, void *stream);
int r1_estimate_inter_costs(r1_ctx *ctx, const R1Plane *org, const R1Plane *ref,
                            const int16_t *mvs, uint32_t *costs, void *stream);
int r1_importance_block_difference(r1_ctx *ctx, const R1Plane *org, const R1Plane *ref,
                                   uint64_t *sum_out, void *stream);

/* ---- frame glue: planes stay resident in HBM between the block stages ----
 * r1_plane_pad: Plane::pad(w, h) (v_frame 0.3.9) as FramePad::pad calls it per plane
 * (src/frame/mod.rs:76-86) on the reconstruction before it becomes a reference
 * (src/api/internal.rs:1436): the visible area of (w + xdec) >> xdec by (h + ydec) >> ydec
 * pixels is replicated over the whole allocation (left / right to the stride, above / below to
 * alloc_height).  w, h: FRAME size in luma pixels; xdec, ydec: the plane's decimation.
 * r1_plane_downsample: Plane::downsampled(frame_w, frame_h) (v_frame 0.3.9), the call that
 * makes FrameState::input_hres / input_qres (src/encoder.rs:476-477): dst (caller-allocated,
 * visible size ((src.width + 1) / 2, (src.height + 1) / 2), half the padding) receives the
 * rounded 2x2 box average, then dst.pad(frame_w, frame_h) with dst's decimation (1 for the
 * half-, 2 for the quarter-resolution plane) -- one launch for both.  src must be padded (odd
 * sizes read one pixel into its border), as in the reference. */
int r1_plane_pad(r1_ctx *ctx, const R1Plane *plane, int w, int h, int xdec, int ydec, void *stream);
int r1_plane_downsample(r1_ctx *ctx, const R1Plane *src, const R1Plane *dst, int frame_w, int frame_h,
                        int dst_xdec, int dst_ydec, void *stream);

/* ActivityMask::from_plane + fill_scales (src/activity.rs:21-66): the spatial
 * DistortionScale of every 8x8 luma block, ssim_boost(variance, variance) in
 * Q14 -- the producer of the `scales` grid r1_dist_scaled_batch and
 * r1_rdo_pixel_cand_batch consume (the encoder multiplies it with the
 * lookahead's spatiotemporal scale, src/api/internal.rs).  ceil(w/8) x
 * ceil(h/8) entries, row-major; either output may be NULL. */
int r1_activity_scales(r1_ctx *ctx, const R1Plane *luma, uint32_t *variances, uint32_t *scales,
                       void *stream);

/* update_block_importances (SURVEY.md 8f "N1"; src/api/internal.rs:911-1068)
 * after its SATD map, which is the map of r1_estimate_inter_costs with
 * mvs[i] = me_stats[2y][2x].mv: every 8x8 importance block i of the current
 * frame adds  (intra_cost + future_importance) * (1 - inter_cost / intra_cost) / len
 * (0 when intra_cost <= inter_cost), split by overlap area, to the importance
 * blocks of the reference frame under its motion-compensated position.  All
 * maps are w_in_imp_b x h_in_imp_b, row-major, DEVICE; ref_importances is
 * updated in place.  f32 throughout, one IEEE operation at a time and summed
 * in the reference's order (source blocks in raster order): the result is
 * bit-identical to the sequential loop.  scratch: DEVICE, 256-byte aligned,
 * at least r1_update_block_importances_scratch_bytes(w, h) bytes (pair lists,
 * per-destination counts / offsets: 76 bytes per importance block; < 0 on
 * error), caller-owned like every buffer. */
long long r1_update_block_importances_scratch_bytes(int w_in_imp_b, int h_in_imp_b);
int r1_update_block_importances(r1_ctx *ctx, const uint32_t *intra_costs,
                                const float *future_importances, const uint32_t *inter_costs,
                                const int16_t *mvs, int w_in_imp_b, int h_in_imp_b, int len,
                                float *ref_importances, void *scratch, long long scratch_bytes,
                                void *stream);

/* ---- intra mode pre-screen (SURVEY.md 8f "N1"; src/rdo.rs:1434-1506): for
 * every block the candidate modes are predicted from ONE edge set
 * (get_intra_edges with IntraParam::None) and ranked by get_satd against the
 * source block.  cands: n = blocks * group entries, the `group` modes of block
 * b at [b*group, (b+1)*group); edges / lens (r1_intra_edges_batch layout) and
 * pos_xy (x, y of the block in `src`, int16 pairs) have one entry per BLOCK.
 * The predictions stay in LDS; satd_out[n].  The probability ordering and the
 * final sort (rdo.rs:1424-1428, 1504) are entropy-coder state and stay with
 * the caller. */
int r1_intra_satd_batch(r1_ctx *ctx, const R1Plane *src, int tx_size, const R1IntraCand *cands,
                        int n, int group, const int16_t *pos_xy, const void *edges,
                        int edge_stride, const uint8_t *lens, const int16_t *ac,
                        uint32_t *satd_out, void *stream);

/* The selection step of both mode pre-screens (SURVEY.md 8f "N1": "returning
 * sorted candidate lists"): per group of `group` keys (the SATDs of one block's
 * candidates, in the caller's order -- for the intra pre-screen that is the
 * probability order of src/rdo.rs:1424-1428) the first `keep_head` candidates
 * keep their places, the rest are STABLY sorted by key, and the first k of the
 * resulting list are returned as indices into the group:
 *   intra  modes[num_modes_rdo / 2..].sort_by_key(satd); take(num_modes_rdo)
 *          (src/rdo.rs:1504-1509): keep_head = num_modes_rdo / 2, k = num_modes_rdo
 *   inter  sorted.sort_by_key(satd); take(num_modes_rdo) (src/rdo.rs:1352-1357):
 *          keep_head = 0
 * keys: n_groups * group (device), idx_out: n_groups * k bytes (device).
 * 1 <= k <= group <= 64, 0 <= keep_head <= k. */
int r1_prescreen_select_batch(r1_ctx *ctx, const uint32_t *keys, int n_groups, int group,
                              int keep_head, int k, uint8_t *idx_out, void *stream);

/* ---- hierarchical motion estimation of whole tiles (SURVEY.md 8f "N2").
 * Replaces estimate_tile_motion (src/me.rs:153-218) for one (tile, reference
 * frame) pair per job: the three passes (quarter, half, full resolution) of
 * estimate_sb_motion / refine_subsampled_sb_motion with full_pixel_me
 * (predictor subsets, diamond search; at the first pass the thresholded
 * subset / uneven-multi-hexagon / optional full search cascade), writing the
 * reference's FrameMEStats.  Jobs are independent (the reference runs tiles on
 * rayon workers, src/encoder.rs encode_tile_group, and loops the reference
 * frames inside, me.rs:190-199) and execute concurrently; inside a job the
 * superblocks go in anti-diagonal wavefronts -- the exact dependence order of
 * the reference's raster walk (see rav1e_amd/csrc/me.hip).
 *
 * R1MeStats = MEStats (src/me.rs:31-35): mv in 1/8 pel, SAD normalised to a
 * 128x128 block.  stats: the FrameMEStats array of this reference frame
 * (stats_cols x stats_rows entries, one per 4x4 luma block, device memory,
 * in/out: the first pass reads what the previous frame left there exactly as
 * the reference does); prev: the previous frame's array for the same
 * reference index (EPZS subset C, me.rs:477-514) or NULL.
 * org / ref: [0] full, [1] half, [2] quarter resolution luma (FrameState
 * input_hres / input_qres, src/encoder.rs:412-413; ReferenceFrame input_hres /
 * input_qres): device planes with the reference's padding (>= 16 px + block
 * at each resolution, which the MV range of me.rs:339-362 assumes).
 * lambda[ssdec]: (fi.me_lambda * 256 / (1 << 2*ssdec) * (ssdec == 0 ? 0.5 :
 * 0.125)) as u32, me.rs:175-177 -- f64 arithmetic, evaluated by the host.
 * `jobs` is HOST memory (pointers inside are device pointers) and is consumed
 * before the call returns; the work itself is only ENQUEUED on `stream`
 * as ONE persistent launch whose waves walk block rows and hand results over
 * through progress words in memory (two waves per SIMD).  launch_mode 2: every job
 * is pinned to one XCD so that the hand-overs stay in that XCD's L2 -- wants >= 8
 * jobs to use the chip and a device whose launches spread over 8 XCDs (probed once
 * per context; forcing mode 2 elsewhere is R1_EINVAL).  launch_mode 3: not pinned
 * -- any wave takes any row, results written through at agent scope.  launch_mode 1:
 * the launch-boundary version (superblock columns + rows + 3 launches replayed as
 * one hipGraph, the three passes skewed inside them), kept as the cross-check.
 * launch_mode 0 takes 2 from 8 jobs on, else 3.  A dependency wait of a persistent
 * launch that runs out of patience is not a hang: the call is flagged (r1_me_status,
 * below), and once a flagged launch has finished every following call of this
 * function returns R1_ETIMEDOUT -- nothing enqueued -- until r1_me_status has
 * reported the flag.  At most 256 jobs per call (tiles x reference frames of
 * one frame).  The context keeps one scratch MEStats frame per distinct
 * `stats` array of a call (the refinements of a pass are computed one diagonal
 * ahead of its searches and must stay invisible to them until then). */
typedef struct R1MeStats {
  int16_t row, col;
  uint32_t normalized_sad;
} R1MeStats;
typedef struct R1MeParams {
  int32_t w_in_b, h_in_b;            /* fi.w_in_b, fi.h_in_b (4x4 units) */
  int32_t stats_cols, stats_rows;    /* FrameMEStats::cols, rows */
  int32_t bit_depth;
  int32_t allow_hp;                  /* fi.allow_high_precision_mv */
  int32_t allow_full_search;         /* speed_settings.motion.me_allow_full_search */
  int32_t me_range_scale;            /* fi.me_range_scale */
  uint32_t lambda[3];
  int32_t launch_mode;               /* r1_estimate_tile_motion_batch: 0 = choose (below), 1 = one
                                      * launch per superblock diagonal, 2 / 3 = one persistent launch,
                                      * jobs pinned to an XCD each / not pinned */
} R1MeParams;
typedef struct R1MeJob {
  R1Plane org[3], ref[3];
  R1MeStats *stats;
  const R1MeStats *prev;
  int32_t tile_x, tile_y;            /* luma px, multiples of 64 */
  int32_t tile_w, tile_h;            /* luma px, multiples of 4 */
} R1MeJob;
int r1_estimate_tile_motion_batch(r1_ctx *ctx, const R1MeJob *jobs, int n_jobs,
                                  const R1MeParams *params, void *stream);
/* The persistent launches (launch_mode 2 / 3) hand results between waves through memory; a wave
 * whose dependency wait runs out of patience (a bounded spin: nothing can hang the GPU) carries on
 * with stale predictors and flags the CALL.  The statistics of a call are valid once r1_me_status
 * has returned R1_OK after it: wait != 0 first waits for every launch enqueued so far (wait == 0
 * only looks at launches that have finished).  R1_ETIMEDOUT: *first_failed_call (optional) is the
 * 1-based index, per context, of the first flagged r1_estimate_tile_motion_batch call -- re-issue
 * it with launch_mode = 1 (launch boundaries instead of waits).  Flags are consumed by the report;
 * while one is pending, r1_estimate_tile_motion_batch refuses with R1_ETIMEDOUT (a caller that never
 * polls cannot go on consuming non-reference statistics silently).  *calls (optional): calls made on this context so far. */
int r1_me_status(r1_ctx *ctx, int wait, unsigned long long *first_failed_call,
                 unsigned long long *calls);

/* estimate_motion with a predicted MV (the RDO-time call, src/rdo.rs:1183-1196:
 * estimate_motion(fi, ts, w, h, tile_bo, ref, Some(pmv), corner, false, 0, None),
 * src/me.rs:536-632) over n independent blocks of one (tile, reference) pair:
 * full_pixel_me at full resolution from the tile's MEStats (read only),
 * get_fullpel_mv_rd with SATD when use_satd (speed_settings.motion
 * .use_satd_subpel), then subpel_diamond_search (1/2 -> 1/4 -> 1/8 pel when
 * allow_hp; put_8tap of fi.default_filter = filter_mode + get_satd / get_sad).
 * tile: HOST descriptor (planes [0] are used; stats / prev as above).
 * cands, out: DEVICE arrays.  bx, by: tile-relative position in 4x4 units;
 * w, h: a BlockSize up to 64x64; corner: 0 = MVSamplingMode::INIT, else
 * 1 | right << 1 | bottom << 2; pmv[k] = (row, col) in 1/8 pel.
 * max_w, max_h: the largest block of the batch (sizes the LDS of the launch;
 * a candidate that exceeds it, or is not a power-of-two BlockSize, returns the
 * reference's MotionSearchResult::empty(): cost = u64::MAX, sad = u32::MAX).
 * out[i] = MotionSearchResult { mv, rd { cost, sad } }. */
typedef struct R1MeBlockCand {
  int16_t bx, by;
  uint8_t w, h, corner, reserved;
  int16_t pmv[2][2];
} R1MeBlockCand;
typedef struct R1MeResult {
  int16_t row, col;
  uint32_t sad;
  uint64_t cost;
} R1MeResult;
int r1_estimate_motion_batch(r1_ctx *ctx, const R1MeJob *tile, const R1MeParams *params,
                             const R1MeBlockCand *cands, int n, int max_w, int max_h,
                             int use_satd, int filter_mode, R1MeResult *out, void *stream);

/* ---- deblocking filter and its level search (SURVEY.md 8f "N3"; reference
 * src/deblock.rs: deblock_plane 1294-1459 / deblock_filter_frame 1544-1551,
 * sse_plane 1461-1542, sse_optimize 1553-1617, deblock_filter_optimize 1620).
 * R1DeblockBlock: what the filter reads from the reference's per-4x4 `Block`
 * (src/context/block_unit.rs), one entry per 4x4 luma block of the frame,
 * row-major with `blocks_stride` entries per row, DEVICE memory:
 *   tx_log2   = log2(txsize.width_mi()) | log2(txsize.height_mi()) << 3
 *   uvtx_log2 = the same for bsize.largest_chroma_tx_size(xdec, ydec)
 *   n4_log2   = log2(n4_w) | log2(n4_h) << 3
 *   flags     = skip | (ref_frames[0] == INTRA_FRAME) << 1
 *               | (mode >= NEARESTMV && mode != GLOBALMV && mode != GLOBAL_GLOBALMV) << 2
 *               | ref_frames[0].to_index() << 3
 *   deltas    = deblock_deltas
 * R1DeblockState = DeblockState (levels: Y vertical, Y horizontal, U, V).
 * r1_deblock_plane filters `plane` IN PLACE (pli 0..2; luma: xdec = ydec = 0),
 * crop_w / crop_h in luma pixels as the reference's callers pass them
 * (fi.width, fi.height).
 * r1_deblock_sse_plane ADDS the level-search tallies of one plane into
 * v_tally / h_tally (device, MAX_LOOP_FILTER + 2 = 65 int64 each, zeroed by
 * the caller); r1_deblock_pick_levels is sse_optimize's tail on HOST copies of
 * them: luma -> levels_out[0..1] = (vertical, horizontal), chroma ->
 * levels_out[0].  Like the reference's sse_h_edge (deblock.rs:1258) the
 * horizontal tallies size their filters from transform WIDTHS; near the top and
 * bottom of the frame such a line may read up to 7 rows of the planes' padding. */
typedef struct R1DeblockBlock {
  uint8_t tx_log2, uvtx_log2, n4_log2, flags;
  int8_t deltas[4];
} R1DeblockBlock;
typedef struct R1DeblockState {
  uint8_t levels[4];
  uint8_t sharpness;            /* carried; the reference's filters never read it */
  uint8_t deltas_enabled, block_deltas_enabled, block_delta_shift, block_delta_multi;
  int8_t ref_deltas[8], mode_deltas[2];
  uint8_t reserved[5];
} R1DeblockState;
int r1_deblock_plane(r1_ctx *ctx, const R1DeblockState *state, const R1Plane *plane, int pli,
                     int xdec, int ydec, const R1DeblockBlock *blocks, int blocks_stride,
                     int blocks_cols, int blocks_rows, int crop_w, int crop_h, void *stream);
int r1_deblock_sse_plane(r1_ctx *ctx, const R1Plane *rec, const R1Plane *src, int pli, int xdec,
                         int ydec, const R1DeblockBlock *blocks, int blocks_stride, int blocks_cols,
                         int blocks_rows, int crop_w, int crop_h, int64_t *v_tally,
                         int64_t *h_tally, void *stream);
int r1_deblock_pick_levels(const int64_t *v_tally, const int64_t *h_tally, int pli,
                           uint8_t *levels_out);
/* The same for all three planes of a 4:x:x frame at once (deblock_filter_frame /
 * deblock_filter_optimize, src/deblock.rs:1544-1551, 1620-1668): planes / rec /
 * src point at THREE R1Plane structs (Y, U, V; host memory), chroma decimation
 * xdec / ydec.  r1_deblock_frame: two launches (all vertical edges of the three
 * planes, then all horizontal ones) instead of six.  r1_deblock_sse_frame: one
 * launch for the six (plane, direction) tallies; `tallies` = 6 x 65 int64
 * (DEVICE, zeroed by the caller), entry 2 * pli + (0 vertical, 1 horizontal),
 * each as r1_deblock_sse_plane fills it. */
int r1_deblock_frame(r1_ctx *ctx, const R1DeblockState *state, const R1Plane *planes, int xdec,
                     int ydec, const R1DeblockBlock *blocks, int blocks_stride, int blocks_cols,
                     int blocks_rows, int crop_w, int crop_h, void *stream);
int r1_deblock_sse_frame(r1_ctx *ctx, const R1Plane *rec, const R1Plane *src, int xdec, int ydec,
                         const R1DeblockBlock *blocks, int blocks_stride, int blocks_cols,
                         int blocks_rows, int crop_w, int crop_h, int64_t *tallies, void *stream);

/* ---- loop restoration, self-guided filter (SURVEY.md 8f "N3", last stage of
 * the post-filter chain; reference RestorationState::lrf_filter_frame
 * src/lrf.rs:1482-1585 -> setup_integral_image 530 + sgrproj_stripe_filter
 * 630; the encoder never selects Wiener, src/rdo.rs:2508).  One plane per call:
 * cdeffed = the CDEF output, deblocked = the frame before CDEF (rows outside a
 * 64-row stripe are taken from it, as the decoder does), out = a plane that
 * already holds a copy of the CDEF output (units without a filter are left as
 * they are; must not alias cdeffed).  crop_w / crop_h: visible size of THIS
 * plane ((fi.width + xdec_round) >> xdec ..), frame_height: fi.height (luma;
 * sets the stripe count), ydec: the plane's vertical decimation, unit_size /
 * unit_cols / unit_rows / stripe_height: RestorationPlaneConfig, units:
 * unit_rows x unit_cols entries (DEVICE), filter = RESTORE_NONE 0 or
 * RESTORE_SGRPROJ 3, set = index into SGRPROJ_PARAMS_S, xqd as coded.
 * A unit with R1_SGR_EDGE_LEFT reads 4 columns left of its x, one with R1_SGR_EDGE_ABOVE 2 rows above its y:
 * such a unit must lie at x >= 4 (y >= 2) or the plane must carry that much origin padding (xorigin >= 4,
 * yorigin >= 2; rav1e's planes carry 88 / 44) -- the kernels clamp the read to the allocation's first
 * column / row otherwise, which is not what the reference reads.
 * All three restoration entry points address pixels with 32-bit byte offsets:
 * R1_EINVAL for an input plane whose allocation (stride * alloc_height *
 * bytes_per_px) reaches 4 GiB or whose stride / alloc_height reach 2^24. */
typedef struct R1LrfUnit {
  uint8_t filter, set;
  int8_t xqd[2];
} R1LrfUnit;
int r1_lrf_sgrproj_plane(r1_ctx *ctx, const R1Plane *cdeffed, const R1Plane *deblocked,
                         const R1Plane *out, int ydec, int crop_w, int crop_h, int frame_height,
                         int unit_size, int unit_cols, int unit_rows, int stripe_height,
                         const R1LrfUnit *units, void *stream);
/* sgrproj_solve (src/lrf.rs:847-1096) as the restoration search calls it
 * (rdo_loop_decision, src/rdo.rs:2651-2676): for each (restoration unit,
 * parameter set) pair the least-squares projection weights xqd of the
 * self-guided filter of `cdeffed` towards `input` (the source frame), the
 * unit hard-clipped at its right / bottom edge, its left / upper neighbourhood
 * as the unit's `edges` say (below), no stripes.  units (DEVICE):
 * (x, y, w, h) in plane pixels, w <= max_w, h <= max_h (<= 384); xqd_out:
 * 2 int8 per pair; moments_scratch: 5 int64 per pair (device; zeroed here).
 * The moments are exact integers, the 2x2 solve is done in IEEE doubles with
 * the reference's operation order (including its two fused multiply-adds),
 * so the weights are bit-identical. */
typedef struct R1SgrSolveUnit {
  int16_t x, y, w, h;
  uint8_t set;
  uint8_t edges;      /* R1_SGR_EDGE_* (ABI 6; was reserved = 0) */
  uint8_t reserved[2];
} R1SgrSolveUnit;
/* What setup_integral_image (src/lrf.rs:530-630) sees outside the unit.  rdo_loop_decision filters
 * a unit on its scratch copy of the AREA it is deciding -- the largest restoration unit of the three
 * planes, in superblocks (src/rdo.rs:2119-2141, 2277-2296) -- and that copy has no pixels outside the
 * area.  The right and bottom edges are always clipped to the unit (crop = the unit).  On the left
 * the filter reads 4 real columns, above 2 real rows, IF the unit's slice does not start in column 0
 * / row 0 of the area copy, i.e. if the plane has more than one unit per area (a 64-pixel luma unit
 * beside 128-pixel chroma units) and this is not the first:
 *   edges = (unit x in plane pixels) % (area width in plane pixels) != 0 ? R1_SGR_EDGE_LEFT : 0
 *         | (unit y ...) % (area height ...) != 0 ? R1_SGR_EDGE_ABOVE : 0
 * 0 (every unit of the 64 / 32-pixel configuration: one unit per plane and area) = the unit alone.
 * A flag is ignored at the plane's own left / top edge.  Found by executing rdo_loop_decision itself
 * (tests/golden/gen_loop_decision_ref.py); up to ABI 5 the kernels decided from the unit's position
 * in the FRAME, which matches the reference only for areas at the frame's left / top edge. */
enum { R1_SGR_EDGE_LEFT = 1, R1_SGR_EDGE_ABOVE = 2 };
int r1_sgrproj_solve_batch(r1_ctx *ctx, const R1Plane *cdeffed, const R1Plane *input,
                           const R1SgrSolveUnit *units, int n, int max_w, int max_h,
                           int64_t *moments_scratch, int8_t *xqd_out, void *stream);
/* The restoration-filter leg of rdo_loop_decision for ONE plane, everything but the entropy coder's
 * rate (src/rdo.rs:2575-2763).  For each (restoration unit, parameter set) pair:
 *   xqd = sgrproj_solve on the unit (as above);
 *   the unit filtered with those weights -- sgrproj_stripe_filter on the unit's own integral image
 *   (setup_integral_image with crop = the unit, rdo.rs:2651-2666, 2688-2702), never stored;
 *   err = rdo_loop_plane_error of that against `src` (rdo.rs:2027-2093): over the 8x8-luma blocks
 *   of the unit  cdef_dist_kernel * bias  (luma, is_chroma 0)  or  sse_wxh with |_, _| bias on
 *   (8 >> xdec) x (8 >> ydec) pixels (is_chroma 1), bias = scales[(luma y >> 3) * scale_stride +
 *   (luma x >> 3)] (NULL: the default scale), summed and multiplied by dist_scale (fi.dist_scale[pli],
 *   Q14).
 * units[i].set = 255: the "no filter option" (rdo.rs:2617-2643): err of lrf_in itself, xqd (0, 0).
 * Units start on superblock boundaries; w % (8 >> xdec) == 0 and h % (8 >> ydec) == 0 (a visible
 * frame that is a multiple of 8 luma pixels: the reference's last blocks otherwise read its working
 * copy beyond what the filter wrote; partial blocks are left out here).  scratch: 6 int64 per pair
 * (device).  The host adds cw.fc.count_lrf_switchable's rate to err and keeps the cheapest choice
 * (compute_rd_cost, rdo.rs:718-723). */
int r1_lrf_search_batch(r1_ctx *ctx, const R1Plane *lrf_in, const R1Plane *src,
                        const R1SgrSolveUnit *units, int n, int max_w, int max_h, int is_chroma,
                        int xdec, int ydec, const uint32_t *scales, int scale_stride,
                        uint32_t dist_scale, int64_t *scratch, int8_t *xqd_out, uint64_t *err_out,
                        void *stream);

/* ---- fused RDO candidate: the headline path.  For each candidate:
 *   pred   = put_8tap(ref @ (rx,ry), fracs, modes)              (src/mc.rs:250)
 *   sad    = get_sad(org @ (ox,oy), pred)       if sad_out     (src/dist.rs:31)
 *   satd   = get_satd(org @ (ox,oy), pred)      if satd_out    (src/dist.rs:156)
 *   resid  = diff(org, pred)                                    (src/encoder.rs:1355)
 *   coeffs = forward_transform(resid, tx_size, tx_type)  if coeffs
 * in one launch with the block staged in LDS; pred never touches HBM unless
 * pred_out is non-NULL.  tx_size must be the block's own size (w x h <= 64). */
int r1_rdo_cand_batch(r1_ctx *ctx, const R1Plane *org, const R1Plane *ref,
                      int w, int h, int tx_size, const R1RdoCand *cands, int n,
                      uint32_t *sad_out, uint32_t *satd_out, void *coeffs,
                      void *pred_out, void *stream);

/* ---- full RDO candidate (SURVEY.md 8f "N4"): the fused candidate carried
 * through the quantizer, i.e. what encode_tx_block computes per transform block
 * for RDOType::TxDistEstRate (src/encoder.rs:1533-1650) with an inter
 * prediction in front (src/rdo.rs:1073 rdo_tx_size_type -> motion_compensate ->
 * encode_tx_block):
 *   pred, sad, satd, resid, coeffs     as r1_rdo_cand_batch
 *   eob      = quantize(coeffs -> qcoeffs; scan order of the candidate's tx_type)
 *                                                   (src/quantize/mod.rs:282-361)
 *   rcoeffs  = dequantize(qcoeffs)                  (src/quantize/mod.rs:363-384)
 *   tx_dist  = sum (coeffs - rcoeffs)^2, rounded and shifted by tx_scale
 *                                                   (src/encoder.rs:1615-1650)
 *   est_rate = estimate_rate(qindex, tx_size, tx_dist)       (src/rdo.rs:127-139)
 * One launch; prediction, residual, coefficients and rcoeffs never leave the
 * CU: 14 bytes per candidate go to HBM.  eob_out, tx_dist_out required;
 * sad_out, satd_out, est_rate_out optional; qcoeffs_out (optional): dense
 * coded-area blocks (min(w,32)*min(h,32) int16 for 8-bit / int32 otherwise);
 * coeffs (optional): the unquantized w*h coefficients as r1_rdo_cand_batch.
 * tx_type >= 16 (WHT) has no scan order and is rejected by the reference's
 * table bounds; here the result for such a candidate is unspecified. */
int r1_rdo_full_cand_batch(r1_ctx *ctx, const R1Plane *org, const R1Plane *ref, int w,
                           int h, int tx_size, const R1RdoCand *cands, int n,
                           const R1QuantParams *params, uint32_t *sad_out,
                           uint32_t *satd_out, uint16_t *eob_out, uint64_t *tx_dist_out,
                           uint64_t *est_rate_out, void *qcoeffs_out, void *coeffs,
                           void *stream);

/* ---- the default-configuration RDO evaluation of an inter transform block in
 * ONE launch: what luma_chroma_mode_rdo / rdo_tx_size_type make encode_tx_block
 * and compute_distortion do per candidate when the distortion is taken in the
 * pixel domain (tune = Psychovisual, or need_recon_pixel; src/encoder.rs:
 * 1533-1661, src/rdo.rs:254-340):
 *   pred, sad, satd, resid, coeffs, eob, qcoeffs     as r1_rdo_full_cand_batch
 *   rcoeffs = dequantize(qcoeffs)
 *   rec     = inverse_transform_add(rcoeffs, pred)    (src/transform/inverse.rs:1633)
 *   dist    = sse_wxh (R1_DIST_WSSE) / cdef_dist_wxh (R1_DIST_CDEF) of rec against
 *             the source block, with the DistortionScale grid of
 *             r1_dist_scaled_batch (same `scales` layout, xdec / ydec of the plane)
 * Prediction, residual, coefficients and reconstruction live in registers /
 * LDS; per candidate 10 bytes (eob, dist) + the optional outputs reach HBM.
 * qcoeffs_out: what the entropy coder of the host needs for the real rate
 * (RDOType::PixelDistRealRate); rec_out: dense w*h reconstructions. */
int r1_rdo_pixel_cand_batch(r1_ctx *ctx, const R1Plane *org, const R1Plane *ref, int w, int h,
                            int tx_size, const R1RdoCand *cands, int n,
                            const R1QuantParams *params, int dist_kind, const uint32_t *scales,
                            int scale_stride, int xdec, int ydec, uint32_t *sad_out,
                            uint32_t *satd_out, uint16_t *eob_out, uint64_t *dist_out,
                            void *qcoeffs_out, void *rec_out, void *stream);

/* The same chains for a prediction that is NOT a single-reference put_8tap:
 * `pred` holds n dense w*h predictions (r1_predict_intra_batch output for the
 * intra mode / tx-type decision of src/rdo.rs:1507-1600 and rdo_tx_size_type,
 * r1_mc_avg_batch output for compound modes); the candidates' rx / ry /
 * fractions are ignored, (ox, oy, tx_type) are used.  dist_kind 0: dist_out =
 * the transform-domain distortion of r1_rdo_full_cand_batch (rec_out must be
 * NULL); R1_DIST_WSSE / R1_DIST_CDEF: the pixel-domain leg of
 * r1_rdo_pixel_cand_batch. */
int r1_rdo_pred_cand_batch(r1_ctx *ctx, const R1Plane *org, const void *pred, int w, int h,
                           int tx_size, const R1RdoCand *cands, int n,
                           const R1QuantParams *params, int dist_kind, const uint32_t *scales,
                           int scale_stride, int xdec, int ydec, uint32_t *sad_out,
                           uint32_t *satd_out, uint16_t *eob_out, uint64_t *dist_out,
                           void *qcoeffs_out, void *rec_out, void *stream);

/* ---- the transform-type search of one prediction in ONE launch (ABI 5): rdo_tx_type_decision
 * (src/rdo.rs:1701-1817) evaluates every TxType of RAV1E_TX_TYPES (src/transform/mod.rs:28-44) that the
 * block's tx set allows (av1_tx_used[get_tx_set(..)], src/context/transform_unit.rs:37-44, 123-148) on the
 * SAME prediction -- it even re-runs motion_compensate per type (rdo.rs:1738-1742) -- through
 * write_tx_tree / write_tx_blocks -> encode_tx_block (src/encoder.rs:1404-1661) and compute_distortion /
 * compute_tx_distortion.  Here the prediction (put_8tap of `ref`, or the dense `pred` buffer of
 * r1_rdo_pred_cand_batch: exactly one of the two is non-NULL), the residual, SAD / SATD and the staged
 * source block are made ONCE per candidate; then, for every set bit t of tx_type_mask in ascending order
 * (slot j = the j-th set bit; nt = popcount(tx_type_mask)):
 *   forward_transform(t) -> quantize -> dequantize -> [inverse_transform_add(t) -> sse_wxh / cdef_dist_wxh]
 * with the residual re-formed on the CU.  The candidates' own tx_type field is ignored.
 *   dist_kind 0:                 dist_out = transform-domain distortion, est_rate_out (optional) as
 *                                r1_rdo_full_cand_batch; rec_out must be NULL
 *   R1_DIST_WSSE / R1_DIST_CDEF: the pixel-domain leg of r1_rdo_pixel_cand_batch; est_rate_out must be NULL
 * Outputs: sad_out / satd_out (optional): n words; eob_out, dist_out (, est_rate_out): n * nt entries,
 * entry [i * nt + j] = candidate i, slot j; qcoeffs_out (optional): n * nt dense coded-area blocks;
 * rec_out (optional): n * nt dense w*h reconstructions.  The host adds the rate of each slot's qcoeffs
 * and keeps the cheapest (compute_rd_cost, rdo.rs:718-723), applying the reference's early exit after the
 * first type itself (rdo.rs:1793-1802: a pure saving there, it changes no result).
 * Sizes up to 16x16 (up to 7 types here, 16 in AV1) run the fan-out kernel.  Sizes with a 32-point side
 * (DCT_DCT, + IDTX for inter blocks) and with a 64-point side (TX_SET_DCTONLY: tx_type_mask must be 1) run one
 * plain launch per type with the type forced -- two waves per SIMD of the fan-out kernel lose against two
 * launches at four (measured) -- same results, same slots.
 * R1_EINVAL: an empty mask, bits beyond the 16 TxTypes, a type the size has no kernel for (tx_type_mask must be a
 * subset of r1_tx_type_mask(tx_size, 1, 0, 0): the inter sets are the largest), both or neither of ref / pred. */
int r1_rdo_txsearch_batch(r1_ctx *ctx, const R1Plane *org, const R1Plane *ref, const void *pred, int w,
                          int h, int tx_size, const R1RdoCand *cands, int n, uint32_t tx_type_mask,
                          const R1QuantParams *params, int dist_kind, const uint32_t *scales,
                          int scale_stride, int xdec, int ydec, uint32_t *sad_out, uint32_t *satd_out,
                          uint16_t *eob_out, uint64_t *dist_out, uint64_t *est_rate_out,
                          void *qcoeffs_out, void *rec_out, void *stream);
/* The mask of that loop: bit t set = TxType t is in av1_tx_used[get_tx_set(tx_size, is_inter,
 * use_reduced_set)]; rav1e_types_only != 0 keeps only RAV1E_TX_TYPES (DCT_DCT, ADST_DCT, DCT_ADST,
 * ADST_ADST, IDTX, V_DCT, H_DCT = 0x0E0F).  0 for an invalid tx_size.  Host arithmetic, no device work. */
uint32_t r1_tx_type_mask(int tx_size, int is_inter, int use_reduced_set, int rav1e_types_only);

/* ---- multi-GPU: the tile-boundary exchange and the reference-frame all-gather (RCCL over
 * xGMI), SURVEY.md 8(e).  One process (or host thread) per GPU, tile r on rank r.  Rendezvous:
 * rank 0 calls r1_comm_unique_id and hands the 128 bytes to the other ranks by any channel the
 * host has; every rank then calls r1_comm_create with the device of its context current.
 * All calls enqueue on `stream`; planes / buffers are device memory. */
typedef struct r1_comm r1_comm;
typedef struct R1HaloXfer {
  int32_t peer;            /* rank on the other side */
  int32_t dir;             /* 0: send this rectangle of my plane, 1: receive into it */
  int32_t x0, y0, x1, y1;  /* plane pixels, visible-area coordinates, half-open */
} R1HaloXfer;
/* RCCL is loaded on first use (dlopen), not linked: a process that never calls r1_comm_* needs no
 * RCCL at all.  Which library: $R1_RCCL_LIBRARY if set; else a librccl already loaded into the
 * process (a PyTorch process then has ONE RCCL, torch's own); else librccl.so.1 by the loader's
 * search path, then /opt/rocm/lib/librccl.so.1.  r1_comm_library() names the one in use (NULL before
 * the first r1_comm_* call or when none could be loaded; that case is R1_ECOMM). */
const char *r1_comm_library(void);
int r1_comm_unique_id(uint8_t *id128);
int r1_comm_create(r1_ctx *ctx, int rank, int world, const uint8_t *id128, r1_comm **out);
void r1_comm_destroy(r1_comm *comm);
int r1_comm_rank(const r1_comm *comm);
int r1_comm_world(const r1_comm *comm);
/* every rank contributes bytes_per_rank bytes; recv = the contributions in rank order
 * (a plane: each rank's slab of whole rows -> the whole allocation on every rank) */
int r1_comm_allgather(r1_comm *comm, const void *send, void *recv, size_t bytes_per_rank,
                      void *stream);
/* rank r owns rects4[4r .. 4r+3] = (x0, y0, x1, y1) of `plane`; afterwards every rank holds
 * every tile (pack, one all-gather of equal slots, unpack): the whole reconstructed frame as
 * the next reference */
int r1_comm_allgather_tiles(r1_comm *comm, const R1Plane *plane, const int32_t *rects4, void *stream);
/* rectangles of `plane` to / from neighbouring ranks in one grouped send / receive; the two
 * sides derive matching rectangles from the tile grid (rav1e_amd.tiles.tile_halo_plan =
 * (my tile) ^ (peer's tile + halo)), so nothing is negotiated */
int r1_comm_exchange_halos(r1_comm *comm, const R1Plane *plane, const R1HaloXfer *xfers, int n,
                           void *stream);

/* ---- the same exchange as direct peer stores (added in round 4 under ABI 4; current version: r1_abi_version()).  xGMI is a load / store fabric: once a
 * peer's plane is mapped into this process, a kernel stores this rank's rectangles straight into
 * it -- N - 1 links at once, no staging copy, nothing to unpack.  Planes have the same geometry on
 * every rank.  INVARIANT: the destination plane is in no peer's live reference set -- nobody may still
 * read it (motion search / compensation of a later frame included) when the stores start; the
 * hand-shake (r1_comm_barrier) only orders the stores of THIS step before the kernels behind it.
 * The reconstruction of a frame is a new buffer (src/encoder.rs:3322) that then sits in up to 8
 * reference slots (encoder.rs rec -> ref_frames): a host keeps a pool of (live reference slots + 1)
 * planes per rank, maps every plane of the pool ONCE at start-up (r1_comm_plane_pool_open: one blocking
 * collective for the whole pool; not a per-frame call) and stores a new reconstruction only into the plane that
 * left every rank's reference set.  bench.py / rav1e_amd.tiles.TileRing model the two-plane case (the
 * previous frame as the only reference): one reader, so rotating two is enough THERE and only there. */
typedef struct R1IpcMem {
  uint8_t handle[64];      /* hipIpcMemHandle_t of the allocation */
  uint64_t offset;         /* of the exported range inside it */
  uint64_t bytes;
} R1IpcMem;
typedef struct R1PushRect {
  int32_t peer;            /* index into peer_data */
  int32_t x0, y0, x1, y1;  /* plane pixels, visible-area coordinates, half-open */
} R1PushRect;
/* device memory of this process (any offset inside a hipMalloc allocation) as 80 bytes another
 * process maps with r1_ipc_open; the bytes travel by whatever channel the host has */
int r1_ipc_export(r1_ctx *ctx, const void *ptr, size_t bytes, R1IpcMem *out);
/* before mapping anything: 1 = the context's device can address every other GPU this process
 * sees (hipDeviceCanAccessPeer), 0 = some GPU refuses (stay on r1_comm_exchange_halos /
 * r1_comm_allgather_tiles), -1 = unknown (the process sees one GPU only) */
int r1_ipc_peer_access(r1_ctx *ctx);
int r1_ipc_open(r1_ctx *ctx, const R1IpcMem *mem, void **ptr);
int r1_ipc_close(r1_ctx *ctx, void *ptr);
/* stores rects[i] of `plane` into the same rectangle of the plane at peer_data[rects[i].peer]
 * (one launch per 16 rectangles, enqueued on `stream`; no hand-shake) */
int r1_push_rects(r1_ctx *ctx, const R1Plane *plane, void *const *peer_data, int n_peers,
                  const R1PushRect *rects, int n, void *stream);
/* with a communicator: every rank's plane mapped on every rank (peer_data: `world` entries,
 * [rank] = plane->data; the exports travel in one all-gather; blocking, once per plane) */
int r1_comm_open_peer_planes(r1_comm *comm, r1_ctx *ctx, const R1Plane *plane, void **peer_data);
int r1_comm_close_peer_planes(r1_comm *comm, r1_ctx *ctx, void **peer_data);
/* the POOL of a host (live reference slots + 1 planes, see above) in ONE blocking collective: n_planes
 * (<= 64) planes of the same geometry on every rank; peer_data: n_planes * world entries,
 * [p * world + r] = plane p of rank r ([p * world + rank] = planes[p].data).  peer_data + p * world is what
 * the r1_comm_push_* calls take for plane p.  (ABI 7) */
int r1_comm_plane_pool_open(r1_comm *comm, r1_ctx *ctx, const R1Plane *planes, int n_planes, void **peer_data);
int r1_comm_plane_pool_close(r1_comm *comm, r1_ctx *ctx, int n_planes, void **peer_data);
/* stream-ordered hand-shake (a 4-byte all-reduce): what follows on `stream` starts after every
 * rank's stream reached this call */
int r1_comm_barrier(r1_comm *comm, void *stream);
/* r1_comm_allgather_tiles / r1_comm_exchange_halos as peer stores + r1_comm_barrier (of the
 * xfers list only the dir == 0 entries are used: a rank's receives are its peers' stores) */
int r1_comm_push_tile(r1_comm *comm, r1_ctx *ctx, const R1Plane *plane, void *const *peer_data,
                      const int32_t *rects4, void *stream);
/* both legs of a frame behind ONE hand-shake: the halo stores, the tile stores, r1_comm_barrier (added in round 5 under ABI 5) */
int r1_comm_push_frame(r1_comm *comm, r1_ctx *ctx, const R1Plane *plane, void *const *peer_data,
                       const R1HaloXfer *xfers, int n, const int32_t *rects4, void *stream);
int r1_comm_push_halos(r1_comm *comm, r1_ctx *ctx, const R1Plane *plane, void *const *peer_data,
                       const R1HaloXfer *xfers, int n, void *stream);

/* ---- per-call compat shims: reference asm signatures, HOST pointers ----
 * SadFn / SatdFn (src/asm/x86/dist/mod.rs:21-43): strides in BYTES. */
uint32_t rav1e_sad_hip(const uint8_t *src, ptrdiff_t src_stride,
                       const uint8_t *dst, ptrdiff_t dst_stride, int w, int h);
uint32_t rav1e_satd_hip(const uint8_t *src, ptrdiff_t src_stride,
                        const uint8_t *dst, ptrdiff_t dst_stride, int w, int h);
uint32_t rav1e_sad_hbd_hip(const uint16_t *src, ptrdiff_t src_stride,
                           const uint16_t *dst, ptrdiff_t dst_stride, int w,
                           int h);
uint32_t rav1e_satd_hbd_hip(const uint16_t *src, ptrdiff_t src_stride,
                            const uint16_t *dst, ptrdiff_t dst_stride, int w,
                            int h, uint32_t bdmax);
/* PutFn / PutHBDFn (src/asm/x86/mc.rs:17-38); `src` must be readable 3 px
 * before and 4 px after the block in both dimensions (mc.rs:121-123).
 * mode_x/mode_y select the table entry the reference indexes with
 * get_2d_mode_idx (src/asm/x86/mc.rs:82-84). */
void rav1e_put_8tap_hip(uint8_t *dst, ptrdiff_t dst_stride, const uint8_t *src,
                        ptrdiff_t src_stride, int w, int h, int mx, int my,
                        int mode_x, int mode_y);
void rav1e_put_8tap_hbd_hip(uint16_t *dst, ptrdiff_t dst_stride,
                            const uint16_t *src, ptrdiff_t src_stride, int w,
                            int h, int mx, int my, int mode_x, int mode_y,
                            int bitdepth_max);
/* forward_transform (Rust-generic entry, src/asm/x86/transform/forward.rs:444) */
int rav1e_fwd_txfm_hip(const int16_t *input, void *output, size_t stride,
                       int tx_size, int tx_type, int bd, int coeff_bytes);

/* InvTxfmFunc / InvTxfmHBDFunc (src/asm/shared/transform/inverse.rs:15-19)
 * with the dispatch-table indices (tx_size, tx_type) explicit; dst holds the
 * prediction on entry; strides in BYTES; eob is ignored like in the Rust path. */
int rav1e_inv_txfm_add_hip(uint8_t *dst, ptrdiff_t dst_stride, const int16_t *coeff,
                           int eob, int tx_size, int tx_type);
int rav1e_inv_txfm_add_hbd_hip(uint16_t *dst, ptrdiff_t dst_stride, const int32_t *coeff,
                               int eob, int bitdepth_max, int tx_size, int tx_type);
/* CdefDirLBDFn / CdefDirHBDFn, CdefFilterFn / CdefFilterHBDFn
 * (src/asm/x86/cdef.rs:16-37,184-191); `tmp` points at the block inside the
 * reference's padded u16 tile (2 pixels of padding on every side). */
int rav1e_cdef_dir_hip(const uint8_t *img, ptrdiff_t stride, uint32_t *var);
int rav1e_cdef_dir_hbd_hip(const uint16_t *img, ptrdiff_t stride, uint32_t *var,
                           int bitdepth_max);
void rav1e_cdef_filter_hip(uint8_t *dst, ptrdiff_t dst_stride, const uint16_t *tmp,
                           ptrdiff_t tmp_stride, int pri_strength, int sec_strength,
                           int dir, int damping, int xdec, int ydec);
void rav1e_cdef_filter_hbd_hip(uint16_t *dst, ptrdiff_t dst_stride, const uint16_t *tmp,
                               ptrdiff_t tmp_stride, int pri_strength, int sec_strength,
                               int dir, int damping, int bitdepth_max, int xdec, int ydec);
/* intra prediction: the asm entry points take the pointer to the top-left
 * element of the edge buffer (src/asm/x86/predict.rs:20-36); mode / variant
 * select the table entry, ief = 0 none / 1 edge filter / 2 + smooth neighbour
 * (the flags the reference folds into `angle`, predict.rs:301-303). */
int rav1e_ipred_hip(void *dst, ptrdiff_t dst_stride, const void *topleft, int width,
                    int height, int angle, int mode, int variant, int ief, int left_len,
                    int above_len, int avail_w, int avail_h, const int16_t *ac,
                    int bit_depth);

#ifdef __cplusplus
}
#endif
#endif /* RAV1E_AMD_H */
