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
extern "C" int r1_abi_version(void) { return 1; }

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
  if (r1_scan_tables_create(c) != R1_OK) {
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
