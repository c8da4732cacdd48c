// ctx.hip -- context, error reporting and the per-call compat shims that
// carry the reference's asm signatures (host pointers in, values out).
#include <stdarg.h>
#include <string.h>

#include <mutex>

#include "common.hpp"

static thread_local char g_err[512] = "";

void r1_set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char *r1_last_error(void) { return g_err; }
// 3: round-3 additions (r1_me_status, r1_comm_library, the predict:: dispatch symbols), R1_ENOMEM /
// R1_ETIMEDOUT got values of their own, R1MeParams.reserved became launch_mode (round 2)
extern "C" int r1_abi_version(void) { return 7; }

extern "C" int r1_ctx_create(int device, r1_ctx **out) {
  R1_REQUIRE(out);
  int count = 0;
  R1_HIP_CHECK(hipGetDeviceCount(&count));
  R1_REQUIRE(device >= 0 && device < count);
  R1_HIP_CHECK(hipSetDevice(device));
  r1_ctx *c = new r1_ctx();
  c->device = device;
  c->stage = nullptr;
  c->stage_bytes = 0;
  c->pinned = nullptr;
  c->pinned_bytes = 0;
  if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) {
    delete c;
    r1_set_error("hipStreamCreate failed");
    return R1_EHIP;
  }
  c->scan_dev = nullptr;
  for (int k = 0; k < r1_ctx::kMeSlots; k++) {
    c->me_jobs[k] = c->me_jobs_host[k] = nullptr;
    c->me_jobs_bytes[k] = 0;
    c->me_done[k] = nullptr;
    c->me_graph[k] = nullptr;
    c->me_refine[k] = nullptr;
    c->me_refine_bytes[k] = 0;
    c->me_persist[k] = nullptr;
  }
  c->me_next = 0;
  if (r1_scan_tables_create(c) != R1_OK || r1_me_kernel_attrs() != R1_OK) {
    r1_scan_tables_destroy(c);
    (void)hipStreamDestroy(c->own_stream);
    delete c;
    return R1_EHIP;
  }
  *out = c;
  return R1_OK;
}

extern "C" void r1_ctx_destroy(r1_ctx *c) {
  if (!c) return;
  if (c->stage) (void)hipFree(c->stage);
  if (c->pinned) (void)hipHostFree(c->pinned);
  for (int k = 0; k < r1_ctx::kMeSlots; k++) {
    if (c->me_done[k]) {
      (void)hipEventSynchronize(c->me_done[k]);
      (void)hipEventDestroy(c->me_done[k]);
    }
    if (c->me_graph[k]) (void)hipGraphExecDestroy(c->me_graph[k]);
    if (c->me_refine[k]) (void)hipFree(c->me_refine[k]);
    r1_me_persist_free(c->me_persist[k]);
    if (c->me_jobs[k]) (void)hipFree(c->me_jobs[k]);
    if (c->me_jobs_host[k]) (void)hipHostFree(c->me_jobs_host[k]);
  }
  r1_cdef_scratch_free(c);
  r1_scan_tables_destroy(c);
  (void)hipStreamDestroy(c->own_stream);
  delete c;
}

// ---------------------------------------------------------------------------
// compat shims: one lazily created context, one lock (the reference calls its
// kernels from rayon workers concurrently; these shims serialise -- they are
// plumbing, not the product).
// ---------------------------------------------------------------------------
namespace {
std::mutex g_mu;
r1_ctx *g_ctx = nullptr;

r1_ctx *shim_ctx() {
  if (!g_ctx && r1_ctx_create(0, &g_ctx) != R1_OK) {
    fprintf(stderr, "rav1e_amd: no HIP device for compat shim: %s\n", g_err);
    abort();  // reference kernels are infallible; there is no CPU fallback here
  }
  return g_ctx;
}

void *stage(r1_ctx *c, size_t bytes) {
  if (c->stage_bytes < bytes) {
    if (c->stage) (void)hipFree(c->stage);
    size_t cap = bytes < (1u << 20) ? (1u << 20) : bytes * 2;
    if (hipMalloc(&c->stage, cap) != hipSuccess) {
      fprintf(stderr, "rav1e_amd: hipMalloc(%zu) failed\n", cap);
      abort();
    }
    c->stage_bytes = cap;
  }
  return c->stage;
}

#define SHIM_HIP(expr)                                                       \
  do {                                                                       \
    hipError_t e_ = (expr);                                                  \
    if (e_ != hipSuccess) {                                                  \
      fprintf(stderr, "rav1e_amd shim: %s -> %s\n", #expr,                   \
              hipGetErrorString(e_));                                        \
      abort();                                                               \
    }                                                                        \
  } while (0)

inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

uint32_t dist_shim(int kind, const void *src, ptrdiff_t ss, const void *dst,
                   ptrdiff_t ds, int w, int h, int bpp, int bd) {
  std::lock_guard<std::mutex> lk(g_mu);
  r1_ctx *c = shim_ctx();
  const size_t row = (size_t)w * bpp, blk = align256(row * h);
  uint8_t *d = (uint8_t *)stage(c, 2 * blk + 512);
  hipStream_t st = c->own_stream;
  SHIM_HIP(hipMemcpy2DAsync(d, row, src, ss, row, h, hipMemcpyHostToDevice, st));
  SHIM_HIP(hipMemcpy2DAsync(d + blk, row, dst, ds, row, h, hipMemcpyHostToDevice, st));
  R1DistCand cand = {0, 0, 0, 0};
  SHIM_HIP(hipMemcpyAsync(d + 2 * blk, &cand, sizeof(cand), hipMemcpyHostToDevice, st));
  R1Plane a = {d, w, h, w, h, 0, 0, bpp, bd};
  R1Plane b = {d + blk, w, h, w, h, 0, 0, bpp, bd};
  uint32_t *res = (uint32_t *)(d + 2 * blk + 256);
  if (r1_dist_batch(c, kind, &a, &b, w, h, (const R1DistCand *)(d + 2 * blk), 1,
                    res, st) != R1_OK) {
    fprintf(stderr, "rav1e_amd shim: %s\n", g_err);
    abort();
  }
  uint32_t out = 0;
  SHIM_HIP(hipMemcpyAsync(&out, res, 4, hipMemcpyDeviceToHost, st));
  SHIM_HIP(hipStreamSynchronize(st));
  return out;
}

void put_shim(void *dst, ptrdiff_t ds, const void *src, ptrdiff_t ss, int w,
              int h, int mx, int my, int mode_x, int mode_y, int bpp, int bd) {
  std::lock_guard<std::mutex> lk(g_mu);
  r1_ctx *c = shim_ctx();
  const int ww = w + 7, wh = h + 7;
  const size_t wrow = (size_t)ww * bpp, wblk = align256(wrow * wh);
  const size_t orow = (size_t)w * bpp, oblk = align256(orow * h);
  uint8_t *d = (uint8_t *)stage(c, wblk + oblk + 256);
  hipStream_t st = c->own_stream;
  const uint8_t *s0 = (const uint8_t *)src - 3 * ss - 3 * bpp;
  SHIM_HIP(hipMemcpy2DAsync(d, wrow, s0, ss, wrow, wh, hipMemcpyHostToDevice, st));
  R1McCand cand = {0, 0, (uint8_t)mx, (uint8_t)my, (uint8_t)mode_x, (uint8_t)mode_y};
  SHIM_HIP(hipMemcpyAsync(d + wblk + oblk, &cand, sizeof(cand), hipMemcpyHostToDevice, st));
  R1Plane p = {d, ww, wh, w, h, 3, 3, bpp, bd};
  if (r1_mc_put_batch(c, &p, w, h, (const R1McCand *)(d + wblk + oblk), 1,
                      d + wblk, st) != R1_OK) {
    fprintf(stderr, "rav1e_amd shim: %s\n", g_err);
    abort();
  }
  SHIM_HIP(hipMemcpy2DAsync(dst, ds, d + wblk, orow, orow, h, hipMemcpyDeviceToHost, st));
  SHIM_HIP(hipStreamSynchronize(st));
}

int bd_from_max(int bitdepth_max) { return bitdepth_max >= 4095 ? 12 : (bitdepth_max >= 1023 ? 10 : 8); }
}  // namespace

extern "C" uint32_t rav1e_sad_hip(const uint8_t *src, ptrdiff_t ss,
                                  const uint8_t *dst, ptrdiff_t ds, int w, int h) {
  return dist_shim(R1_DIST_SAD, src, ss, dst, ds, w, h, 1, 8);
}
extern "C" uint32_t rav1e_satd_hip(const uint8_t *src, ptrdiff_t ss,
                                   const uint8_t *dst, ptrdiff_t ds, int w, int h) {
  return dist_shim(R1_DIST_SATD, src, ss, dst, ds, w, h, 1, 8);
}
extern "C" uint32_t rav1e_sad_hbd_hip(const uint16_t *src, ptrdiff_t ss,
                                      const uint16_t *dst, ptrdiff_t ds, int w,
                                      int h) {
  return dist_shim(R1_DIST_SAD, src, ss, dst, ds, w, h, 2, 10);
}
extern "C" uint32_t rav1e_satd_hbd_hip(const uint16_t *src, ptrdiff_t ss,
                                       const uint16_t *dst, ptrdiff_t ds, int w,
                                       int h, uint32_t bdmax) {
  return dist_shim(R1_DIST_SATD, src, ss, dst, ds, w, h, 2, bd_from_max((int)bdmax));
}
extern "C" void rav1e_put_8tap_hip(uint8_t *dst, ptrdiff_t ds, const uint8_t *src,
                                   ptrdiff_t ss, int w, int h, int mx, int my,
                                   int mode_x, int mode_y) {
  put_shim(dst, ds, src, ss, w, h, mx, my, mode_x, mode_y, 1, 8);
}
extern "C" void rav1e_put_8tap_hbd_hip(uint16_t *dst, ptrdiff_t ds,
                                       const uint16_t *src, ptrdiff_t ss, int w,
                                       int h, int mx, int my, int mode_x,
                                       int mode_y, int bitdepth_max) {
  put_shim(dst, ds, src, ss, w, h, mx, my, mode_x, mode_y, 2, bd_from_max(bitdepth_max));
}

extern "C" int rav1e_fwd_txfm_hip(const int16_t *input, void *output,
                                  size_t stride, int tx_size, int tx_type, int bd,
                                  int coeff_bytes) {
  static const uint8_t wl[19] = {2, 3, 4, 5, 6, 2, 3, 3, 4, 4, 5, 5, 6, 2, 4, 3, 5, 4, 6};
  static const uint8_t hl[19] = {2, 3, 4, 5, 6, 3, 2, 4, 3, 5, 4, 6, 5, 4, 2, 5, 3, 6, 4};
  if (tx_size < 0 || tx_size >= 19) return R1_EINVAL;
  std::lock_guard<std::mutex> lk(g_mu);
  r1_ctx *c = shim_ctx();
  const int w = 1 << wl[tx_size], h = 1 << hl[tx_size];
  const size_t ib = align256((size_t)w * h * 2), ob = (size_t)w * h * coeff_bytes;
  uint8_t *d = (uint8_t *)stage(c, ib + ob);
  hipStream_t st = c->own_stream;
  SHIM_HIP(hipMemcpy2DAsync(d, (size_t)w * 2, input, stride * 2, (size_t)w * 2, h,
                            hipMemcpyHostToDevice, st));
  int rc = r1_fwd_txfm_batch(c, (const int16_t *)d, d + ib, 1, tx_size, tx_type, bd,
                             coeff_bytes, st);
  if (rc != R1_OK) return rc;
  SHIM_HIP(hipMemcpyAsync(output, d + ib, ob, hipMemcpyDeviceToHost, st));
  SHIM_HIP(hipStreamSynchronize(st));
  return R1_OK;
}

// ---- inverse transform shims: InvTxfmFunc / InvTxfmHBDFunc
// (src/asm/shared/transform/inverse.rs:15-19: dst, dst_stride in BYTES... the
// reference passes plane_cfg.stride in elements as isize and the asm scales it;
// here strides are in BYTES like the other shims) with the table indices
// (tx_size, tx_type) as explicit arguments.  `eob` is accepted and ignored
// (the reference's Rust path ignores it too, inverse.rs:1636).
namespace {
int inv_shim(void *dst, ptrdiff_t ds, const void *coeff, int tx_size, int tx_type, int bpp,
             int bd) {
  static const uint8_t wl[19] = {2, 3, 4, 5, 6, 2, 3, 3, 4, 4, 5, 5, 6, 2, 4, 3, 5, 4, 6};
  static const uint8_t hl[19] = {2, 3, 4, 5, 6, 3, 2, 4, 3, 5, 4, 6, 5, 4, 2, 5, 3, 6, 4};
  if (tx_size < 0 || tx_size >= 19) return R1_EINVAL;
  std::lock_guard<std::mutex> lk(g_mu);
  r1_ctx *c = shim_ctx();
  const int w = 1 << wl[tx_size], h = 1 << hl[tx_size];
  const int area = (w < 32 ? w : 32) * (h < 32 ? h : 32);
  const size_t cbytes = (size_t)area * (bpp == 1 ? 2 : 4), cb = align256(cbytes);
  const size_t prow = (size_t)w * bpp, pb = align256(prow * h);
  uint8_t *d = (uint8_t *)stage(c, cb + pb);
  hipStream_t st = c->own_stream;
  SHIM_HIP(hipMemcpyAsync(d, coeff, cbytes, hipMemcpyHostToDevice, st));
  SHIM_HIP(hipMemcpy2DAsync(d + cb, prow, dst, ds, prow, h, hipMemcpyHostToDevice, st));
  const int rc = r1_inv_txfm_add_batch(c, d, area, d + cb, d + cb, 1, tx_size, tx_type, bd, bpp, st);
  if (rc != R1_OK) return rc;
  SHIM_HIP(hipMemcpy2DAsync(dst, ds, d + cb, prow, prow, h, hipMemcpyDeviceToHost, st));
  SHIM_HIP(hipStreamSynchronize(st));
  return R1_OK;
}
}  // namespace

extern "C" int rav1e_inv_txfm_add_hip(uint8_t *dst, ptrdiff_t dst_stride, const int16_t *coeff,
                                      int eob, int tx_size, int tx_type) {
  (void)eob;
  return inv_shim(dst, dst_stride, coeff, tx_size, tx_type, 1, 8);
}
extern "C" int rav1e_inv_txfm_add_hbd_hip(uint16_t *dst, ptrdiff_t dst_stride,
                                          const int32_t *coeff, int eob, int bitdepth_max,
                                          int tx_size, int tx_type) {
  (void)eob;
  return inv_shim(dst, dst_stride, coeff, tx_size, tx_type, 2, bd_from_max(bitdepth_max));
}

// ---- CDEF shims: CdefDirLBDFn / CdefDirHBDFn (src/asm/x86/cdef.rs:184-191)
// and CdefFilterFn / CdefFilterHBDFn (cdef.rs:16-37).  The filter takes the
// reference's pre-padded u16 tile (CDEF_VERY_LARGE where nothing exists), so
// every halo pixel "exists" and the sentinel does the rest -- the reference's
// own fast path (cdef.rs:236-296).
namespace {
int cdef_dir_shim(const void *img, ptrdiff_t stride, uint32_t *var, int bpp, int bd) {
  std::lock_guard<std::mutex> lk(g_mu);
  r1_ctx *c = shim_ctx();
  const size_t row = (size_t)8 * bpp;
  uint8_t *d = (uint8_t *)stage(c, 1024);
  hipStream_t st = c->own_stream;
  SHIM_HIP(hipMemcpy2DAsync(d, row, img, stride, row, 8, hipMemcpyHostToDevice, st));
  R1CdefDirCand cand = {0, 0};
  SHIM_HIP(hipMemcpyAsync(d + 256, &cand, sizeof(cand), hipMemcpyHostToDevice, st));
  R1Plane p = {d, 8, 8, 8, 8, 0, 0, bpp, bd};
  if (r1_cdef_find_dir_batch(c, &p, (const R1CdefDirCand *)(d + 256), 1, d + 512,
                             (int32_t *)(d + 516), st) != R1_OK) {
    fprintf(stderr, "rav1e_amd shim: %s\n", g_err);
    abort();
  }
  uint8_t res[8];
  SHIM_HIP(hipMemcpyAsync(res, d + 512, 8, hipMemcpyDeviceToHost, st));
  SHIM_HIP(hipStreamSynchronize(st));
  int32_t v;
  memcpy(&v, res + 4, 4);
  *var = (uint32_t)v;
  return res[0];
}

void cdef_filter_shim(void *dst, ptrdiff_t ds, const uint16_t *tmp, ptrdiff_t tmp_stride_bytes,
                      int pri, int sec, int dir, int damping, int xdec, int ydec, int bd,
                      int dst_bpp) {
  std::lock_guard<std::mutex> lk(g_mu);
  r1_ctx *c = shim_ctx();
  const int xs = 8 >> xdec, ys = 8 >> ydec, tw = xs + 4, th = ys + 4;
  const size_t trow = (size_t)tw * 2, tb = align256(trow * th);
  uint8_t *d = (uint8_t *)stage(c, 2 * tb + 256);
  hipStream_t st = c->own_stream;
  const uint8_t *t0 = (const uint8_t *)tmp - 2 * tmp_stride_bytes - 2 * 2;   // padding's top-left
  SHIM_HIP(hipMemcpy2DAsync(d, trow, t0, tmp_stride_bytes, trow, th, hipMemcpyHostToDevice, st));
  R1CdefBlockCand cand = {0, 0, (int16_t)pri, (int16_t)sec, (uint8_t)dir, (uint8_t)damping,
                          R1_CDEF_HAVE_ALL, 0};
  SHIM_HIP(hipMemcpyAsync(d + 2 * tb, &cand, sizeof(cand), hipMemcpyHostToDevice, st));
  // both planes are u16 views with the block at (0,0) and a 2-pixel origin
  R1Plane in = {d, tw, th, xs, ys, 2, 2, 2, bd};
  R1Plane out = {d + tb, tw, th, xs, ys, 2, 2, 2, bd};
  if (r1_cdef_filter_block_batch(c, &in, &out, xdec, ydec, (const R1CdefBlockCand *)(d + 2 * tb), 1,
                                 st) != R1_OK) {
    fprintf(stderr, "rav1e_amd shim: %s\n", g_err);
    abort();
  }
  uint16_t res[8 * 8];
  SHIM_HIP(hipMemcpy2DAsync(res, (size_t)xs * 2, d + tb + (2 * tw + 2) * 2, trow, (size_t)xs * 2, ys,
                            hipMemcpyDeviceToHost, st));
  SHIM_HIP(hipStreamSynchronize(st));
  for (int i = 0; i < ys; i++)
    for (int j = 0; j < xs; j++) {
      if (dst_bpp == 1) ((uint8_t *)dst)[i * ds + j] = (uint8_t)res[i * xs + j];
      else *(uint16_t *)((uint8_t *)dst + i * ds + j * 2) = res[i * xs + j];
    }
}
}  // namespace

extern "C" int rav1e_cdef_dir_hip(const uint8_t *img, ptrdiff_t stride, uint32_t *var) {
  return cdef_dir_shim(img, stride, var, 1, 8);
}
extern "C" int rav1e_cdef_dir_hbd_hip(const uint16_t *img, ptrdiff_t stride, uint32_t *var,
                                      int bitdepth_max) {
  return cdef_dir_shim(img, stride, var, 2, bd_from_max(bitdepth_max));
}
extern "C" void rav1e_cdef_filter_hip(uint8_t *dst, ptrdiff_t dst_stride, const uint16_t *tmp,
                                      ptrdiff_t tmp_stride, int pri_strength, int sec_strength,
                                      int dir, int damping, int xdec, int ydec) {
  cdef_filter_shim(dst, dst_stride, tmp, tmp_stride, pri_strength, sec_strength, dir, damping, xdec,
                   ydec, 8, 1);
}
extern "C" void rav1e_cdef_filter_hbd_hip(uint16_t *dst, ptrdiff_t dst_stride, const uint16_t *tmp,
                                          ptrdiff_t tmp_stride, int pri_strength, int sec_strength,
                                          int dir, int damping, int bitdepth_max, int xdec,
                                          int ydec) {
  cdef_filter_shim(dst, dst_stride, tmp, tmp_stride, pri_strength, sec_strength, dir, damping, xdec,
                   ydec, bd_from_max(bitdepth_max), 2);
}

// ---- intra prediction shim: the reference's asm entry points take the
// pointer to the top-left element of the IntraEdgeBuffer
// (src/asm/x86/predict.rs:20-36: dst, stride, topleft, width, height, angle).
// The table indices of the dispatch (mode, variant) and the two facts the asm
// receives folded into `angle` flags (edge filter on / smooth neighbour,
// predict.rs:301-303) are explicit here.
extern "C" int rav1e_ipred_hip(void *dst, ptrdiff_t dst_stride, const void *topleft, int width,
                               int height, int angle, int mode, int variant, int ief,
                               int left_len, int above_len, int avail_w, int avail_h,
                               const int16_t *ac, int bit_depth) {
  int tx_size = -1;
  static const uint8_t wl[19] = {2, 3, 4, 5, 6, 2, 3, 3, 4, 4, 5, 5, 6, 2, 4, 3, 5, 4, 6};
  static const uint8_t hl[19] = {2, 3, 4, 5, 6, 3, 2, 4, 3, 5, 4, 6, 5, 4, 2, 5, 3, 6, 4};
  for (int t = 0; t < 19; t++)
    if ((1 << wl[t]) == width && (1 << hl[t]) == height) tx_size = t;
  if (tx_size < 0 || left_len < 0 || left_len > 128 || above_len < 0 || above_len > 128)
    return R1_EINVAL;
  const int bpp = bit_depth == 8 ? 1 : 2;
  std::lock_guard<std::mutex> lk(g_mu);
  r1_ctx *c = shim_ctx();
  const size_t eb = align256((size_t)R1_INTRA_EDGE_LEN * bpp);
  const size_t prow = (size_t)width * bpp, pb = align256(prow * height);
  const size_t ab = align256((size_t)width * height * 2);
  uint8_t *d = (uint8_t *)stage(c, eb + pb + ab + 512);
  hipStream_t st = c->own_stream;
  // rebuild the 257-entry buffer around the top-left pointer
  SHIM_HIP(hipMemsetAsync(d, 0, eb, st));
  SHIM_HIP(hipMemcpyAsync(d + (size_t)(128 - left_len) * bpp,
                          (const uint8_t *)topleft - (size_t)left_len * bpp,
                          (size_t)(left_len + 1 + above_len) * bpp, hipMemcpyHostToDevice, st));
  R1IntraCand cand = {(uint8_t)mode, (uint8_t)variant, (int16_t)angle, (uint8_t)ief,
                      (uint8_t)(avail_w > 64 ? 64 : avail_w), (uint8_t)(avail_h > 64 ? 64 : avail_h), 0};
  uint8_t lens[2] = {(uint8_t)left_len, (uint8_t)above_len};
  uint8_t *meta = d + eb + pb + ab;
  SHIM_HIP(hipMemcpyAsync(meta, &cand, sizeof(cand), hipMemcpyHostToDevice, st));
  SHIM_HIP(hipMemcpyAsync(meta + 64, lens, 2, hipMemcpyHostToDevice, st));
  if (ac)
    SHIM_HIP(hipMemcpyAsync(d + eb + pb, ac, (size_t)width * height * 2, hipMemcpyHostToDevice, st));
  const int rc = r1_predict_intra_batch(c, tx_size, (const R1IntraCand *)meta, 1, d,
                                        R1_INTRA_EDGE_LEN, meta + 64,
                                        ac ? (const int16_t *)(d + eb + pb) : nullptr, bit_depth,
                                        bpp, d + eb, st);
  if (rc != R1_OK) return rc;
  SHIM_HIP(hipMemcpy2DAsync(dst, dst_stride, d + eb, prow, prow, height, hipMemcpyDeviceToHost, st));
  SHIM_HIP(hipStreamSynchronize(st));
  return R1_OK;
}

// ---------------------------------------------------------------------------
// per-table-entry dispatch symbols (include/rav1e_amd_dispatch.h, generated by
// tools/gen_dispatch.py): the reference's exact asm argument lists.  They sit on the
// generic shims above plus the few below.
// ---------------------------------------------------------------------------
int r1_internal_wsse_raw(const R1Plane *a, const R1Plane *b, int w, int h, const R1DistCand *cand,
                         const uint32_t *scale, int scale_stride, uint64_t *out, hipStream_t st);
int r1_internal_cdef_dist_raw(const R1Plane *a, const R1Plane *b, int w, int h, const R1DistCand *cand,
                              uint32_t *out3, hipStream_t st);
namespace {
#define SHIM_OK(expr)                                                  \
  do {                                                                 \
    if ((expr) != R1_OK) {                                             \
      fprintf(stderr, "rav1e_amd shim: %s -> %s\n", #expr, g_err);     \
      abort();                                                         \
    }                                                                  \
  } while (0)

void inv_shim_abort(void *dst, ptrdiff_t ds, const void *coeff, int tx_size, int tx_type, int bpp, int bd) {
  SHIM_OK(inv_shim(dst, ds, coeff, tx_size, tx_type, bpp, bd));
}

// two host blocks -> device, as planes with the block at (0, 0)
struct TwoBlocks { uint8_t *d; size_t blk; R1Plane a, b; };
TwoBlocks upload_pair(r1_ctx *c, const void *src, ptrdiff_t ss, const void *dst, ptrdiff_t ds, int w, int h,
                      int bpp, int bd, size_t extra, hipStream_t st) {
  const size_t row = (size_t)w * bpp, blk = align256(row * h);
  uint8_t *d = (uint8_t *)stage(c, 2 * blk + 512 + extra);
  SHIM_HIP(hipMemcpy2DAsync(d, row, src, ss, row, h, hipMemcpyHostToDevice, st));
  SHIM_HIP(hipMemcpy2DAsync(d + blk, row, dst, ds, row, h, hipMemcpyHostToDevice, st));
  R1DistCand cand = {0, 0, 0, 0};
  SHIM_HIP(hipMemcpyAsync(d + 2 * blk, &cand, sizeof(cand), hipMemcpyHostToDevice, st));
  TwoBlocks t = {d, blk, {d, w, h, w, h, 0, 0, bpp, bd}, {d + blk, w, h, w, h, 0, 0, bpp, bd}};
  return t;
}

// WeightedSseFn (src/asm/x86/dist/sse.rs:18-34): the raw sum over 4x4 cells of
// (cell_sse * scale + 128) >> 8; `scale_stride` in BYTES like every asm stride (sse.rs:113)
uint64_t wsse_shim(const void *src, ptrdiff_t ss, const void *dst, ptrdiff_t ds, const uint32_t *scale,
                   ptrdiff_t scale_stride_bytes, int w, int h, int bpp) {
  std::lock_guard<std::mutex> lk(g_mu);
  r1_ctx *c = shim_ctx();
  hipStream_t st = c->own_stream;
  const int cw = w / 4, chh = h / 4;
  const size_t sbytes = align256((size_t)cw * chh * 4);
  TwoBlocks t = upload_pair(c, src, ss, dst, ds, w, h, bpp, bpp == 1 ? 8 : 10, sbytes, st);
  uint8_t *sd = t.d + 2 * t.blk + 512;
  SHIM_HIP(hipMemcpy2DAsync(sd, (size_t)cw * 4, scale, scale_stride_bytes, (size_t)cw * 4, chh,
                            hipMemcpyHostToDevice, st));
  uint64_t *res = (uint64_t *)(t.d + 2 * t.blk + 256);
  SHIM_OK(r1_internal_wsse_raw(&t.a, &t.b, w, h, (const R1DistCand *)(t.d + 2 * t.blk), (const uint32_t *)sd,
                               cw, res, st));
  uint64_t out = 0;
  SHIM_HIP(hipMemcpyAsync(&out, res, 8, hipMemcpyDeviceToHost, st));
  SHIM_HIP(hipStreamSynchronize(st));
  return out;
}

// CdefDistKernelFn (src/asm/x86/dist/cdef_dist.rs:18-24): ret_ptr[3] = {svar, dvar, sse}
void cdef_dist_kernel_shim(const void *src, ptrdiff_t ss, const void *dst, ptrdiff_t ds, int w, int h, int bpp,
                           uint32_t *ret) {
  std::lock_guard<std::mutex> lk(g_mu);
  r1_ctx *c = shim_ctx();
  hipStream_t st = c->own_stream;
  TwoBlocks t = upload_pair(c, src, ss, dst, ds, w, h, bpp, bpp == 1 ? 8 : 10, 0, st);
  uint32_t *res = (uint32_t *)(t.d + 2 * t.blk + 256);
  SHIM_OK(r1_internal_cdef_dist_raw(&t.a, &t.b, w, h, (const R1DistCand *)(t.d + 2 * t.blk), res, st));
  SHIM_HIP(hipMemcpyAsync(ret, res, 12, hipMemcpyDeviceToHost, st));
  SHIM_HIP(hipStreamSynchronize(st));
}

// PrepFn / PrepHBDFn (src/asm/x86/mc.rs:40-59): tmp = dense w*h int16
void prep_shim(int16_t *tmp, const void *src, ptrdiff_t ss, int w, int h, int mx, int my, int mode_x, int mode_y,
               int bpp, int bd) {
  std::lock_guard<std::mutex> lk(g_mu);
  r1_ctx *c = shim_ctx();
  const int ww = w + 7, wh = h + 7;
  const size_t wrow = (size_t)ww * bpp, wblk = align256(wrow * wh);
  const size_t oblk = align256((size_t)w * h * 2);
  uint8_t *d = (uint8_t *)stage(c, wblk + oblk + 256);
  hipStream_t st = c->own_stream;
  const uint8_t *s0 = (const uint8_t *)src - 3 * ss - 3 * bpp;
  SHIM_HIP(hipMemcpy2DAsync(d, wrow, s0, ss, wrow, wh, hipMemcpyHostToDevice, st));
  R1McCand cand = {0, 0, (uint8_t)mx, (uint8_t)my, (uint8_t)mode_x, (uint8_t)mode_y};
  SHIM_HIP(hipMemcpyAsync(d + wblk + oblk, &cand, sizeof(cand), hipMemcpyHostToDevice, st));
  R1Plane p = {d, ww, wh, w, h, 3, 3, bpp, bd};
  SHIM_OK(r1_mc_prep_batch(c, &p, w, h, (const R1McCand *)(d + wblk + oblk), 1, (int16_t *)(d + wblk), st));
  SHIM_HIP(hipMemcpyAsync(tmp, d + wblk, (size_t)w * h * 2, hipMemcpyDeviceToHost, st));
  SHIM_HIP(hipStreamSynchronize(st));
}

// AvgFn / AvgHBDFn (src/asm/x86/mc.rs:61-78)
void avg_shim(void *dst, ptrdiff_t ds, const int16_t *t1, const int16_t *t2, int w, int h, int bpp, int bd) {
  std::lock_guard<std::mutex> lk(g_mu);
  r1_ctx *c = shim_ctx();
  const size_t tb = align256((size_t)w * h * 2), ob = align256((size_t)w * h * bpp);
  uint8_t *d = (uint8_t *)stage(c, 2 * tb + ob);
  hipStream_t st = c->own_stream;
  SHIM_HIP(hipMemcpyAsync(d, t1, (size_t)w * h * 2, hipMemcpyHostToDevice, st));
  SHIM_HIP(hipMemcpyAsync(d + tb, t2, (size_t)w * h * 2, hipMemcpyHostToDevice, st));
  SHIM_OK(r1_mc_avg_batch(c, (const int16_t *)d, (const int16_t *)(d + tb), w, h, 1, bd, bpp, d + 2 * tb, st));
  SHIM_HIP(hipMemcpy2DAsync(dst, ds, d + 2 * tb, (size_t)w * bpp, (size_t)w * bpp, h, hipMemcpyDeviceToHost, st));
  SHIM_HIP(hipStreamSynchronize(st));
}

// DequantizeFn (src/asm/x86/quantize.rs:22-31): i16 coefficients, coded area of tx_size
void dequant_shim(int qindex, const int16_t *coeffs, int16_t *rcoeffs, int tx_size, int bit_depth, int dc_delta_q,
                  int ac_delta_q) {
  static const uint8_t wl[19] = {2, 3, 4, 5, 6, 2, 3, 3, 4, 4, 5, 5, 6, 2, 4, 3, 5, 4, 6};
  static const uint8_t hl[19] = {2, 3, 4, 5, 6, 3, 2, 4, 3, 5, 4, 6, 5, 4, 2, 5, 3, 6, 4};
  if (tx_size < 0 || tx_size >= 19) abort();
  std::lock_guard<std::mutex> lk(g_mu);
  r1_ctx *c = shim_ctx();
  const int w = 1 << wl[tx_size], h = 1 << hl[tx_size];
  const int area = (w < 32 ? w : 32) * (h < 32 ? h : 32);
  const size_t cb = align256((size_t)area * 2);
  uint8_t *d = (uint8_t *)stage(c, 2 * cb);
  hipStream_t st = c->own_stream;
  SHIM_HIP(hipMemcpyAsync(d, coeffs, (size_t)area * 2, hipMemcpyHostToDevice, st));
  R1QuantParams qp = {};
  qp.qindex = qindex;
  qp.bit_depth = bit_depth;
  qp.is_intra = 0;
  qp.dc_delta_q = dc_delta_q;
  qp.ac_delta_q = ac_delta_q;
  SHIM_OK(r1_dequantize_batch(c, d, 1, tx_size, &qp, 2, d + cb, st));
  SHIM_HIP(hipMemcpyAsync(rcoeffs, d + cb, (size_t)area * 2, hipMemcpyDeviceToHost, st));
  SHIM_HIP(hipStreamSynchronize(st));
}

// The reference's ipred entry points (src/asm/x86/predict.rs:21-236): dst, stride in BYTES,
// pointer to the top-left element of the IntraEdgeBuffer, and for the directional zones the
// edge-filter flags folded into `angle` (bit 10 enable, bit 9 smooth neighbour, :301-303).  What
// the asm does not receive -- how far the edges were initialised, how much of the block lies
// inside the frame -- is taken as the asm takes it: the edges are valid as far as the mode reads
// them (left h / above w; the zone's far edge w + h), the block lies inside the frame, z2's
// filter counts are clipped with the dx / dy it is given (:306-316).  mode / variant are the
// constants of the symbol (tools/gen_dispatch.py); mode 13 = UV_CFL_PRED with `angle_arg` = alpha.
void ipred_shim(void *dst, ptrdiff_t stride, const void *topleft, int w, int h, int angle_arg, int mode,
                int variant, int dx, int dy, const int16_t *ac, int bpp, int bd) {
  const bool directional = mode == 3 || mode == 4 || mode == 7;
  int angle = 0, ief = 0, left = h, above = w, aw = w, ah = h;
  if (directional) {
    angle = angle_arg & 511;
    ief = ((angle_arg >> 10) & 1) ? 1 + ((angle_arg >> 9) & 1) : 0;
    if (mode == 3) above = w + h > 128 ? 128 : w + h;
    if (mode == 7) left = w + h > 128 ? 128 : w + h;
    if (mode == 4) {
      if (dx > 0 && dx < aw) aw = dx;
      if (dy > 0 && dy < ah) ah = dy;
    }
  } else if (mode == 1) {
    angle = 90;
  } else if (mode == 2) {
    angle = 180;
  } else if (mode == 13) {
    angle = angle_arg;   // alpha
  }
  SHIM_OK(rav1e_ipred_hip(dst, stride, topleft, w, h, angle, mode, variant, ief, left, above, aw, ah, ac, bd));
  (void)bpp;
}

// rav1e_ipred_cfl_ac_{420,422,444} (src/asm/x86/predict.rs:142-186, wrapper :873-927): ac = dense
// width x height int16, src = the luma block under it, w_pad / h_pad in 4-pixel units of the
// chroma block.  The luma pixels read are those of rust::pred_cfl_ac (src/predict.rs:1020-1063).
void cfl_ac_shim(int16_t *ac, const void *src, ptrdiff_t stride, int w_pad, int h_pad, int w, int h, int xdec,
                 int ydec, int bpp, int bd) {
  std::lock_guard<std::mutex> lk(g_mu);
  r1_ctx *c = shim_ctx();
  hipStream_t st = c->own_stream;
  int lw = (w - 4 * w_pad) << xdec, lh = (h - 4 * h_pad) << ydec;
  if (lw < 8) lw = 8;
  if (lh < 8) lh = 8;
  const size_t row = (size_t)lw * bpp, blk = align256(row * lh), ab = align256((size_t)w * h * 2);
  uint8_t *d = (uint8_t *)stage(c, blk + ab + 256);
  SHIM_HIP(hipMemcpy2DAsync(d, row, src, stride, row, lh, hipMemcpyHostToDevice, st));
  R1CflAcCand cand = {0, 0, (uint8_t)w_pad, (uint8_t)h_pad, {0, 0}};
  SHIM_HIP(hipMemcpyAsync(d + blk + ab, &cand, sizeof(cand), hipMemcpyHostToDevice, st));
  R1Plane p = {d, lw, lh, lw, lh, 0, 0, bpp, bd};
  SHIM_OK(r1_cfl_ac_batch(c, &p, w, h, xdec, ydec, (const R1CflAcCand *)(d + blk + ab), 1, (int16_t *)(d + blk), st));
  SHIM_HIP(hipMemcpyAsync(ac, d + blk, (size_t)w * h * 2, hipMemcpyDeviceToHost, st));
  SHIM_HIP(hipStreamSynchronize(st));
}
}  // namespace

#include "dispatch_gen.inc"
