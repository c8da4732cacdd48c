/*
 * oracle/plane.c -- TEST INFRASTRUCTURE (see r1_oracle.h).
 * The two frame-glue operations that keep planes resident between the per-block stages:
 *   Plane::pad          called by FramePad::pad, src/frame/mod.rs:76-86, at src/api/internal.rs:1436
 *                       (the reconstruction's borders are replicated before it becomes a reference)
 *   Plane::downsampled  called at src/encoder.rs:476-477 (half- and quarter-resolution inputs of
 *                       the hierarchical motion search)
 * Both live in the third-party crate v_frame (Cargo.lock pins 0.3.9), which is NOT under
 * /root/reference: they are restated here from the crate's published algorithm --
 *   pad(w, h): width = (w + xdec) >> xdec, height likewise; every row of the visible area gets
 *     its first pixel replicated over [0, xorigin) and its last over [xorigin + width, stride);
 *     then the first (already widened) row is copied to every row above, the last to every row
 *     below, down to alloc_height;
 *   downsampled(frame_w, frame_h): new plane of ((width + 1) / 2, (height + 1) / 2), decimation
 *     + 1, padding / 2; each pixel = (a + b + c + d + 2) >> 2 of the 2x2 source quad (for odd
 *     sizes the quad reaches one pixel into the source's padding); then new.pad(frame_w, frame_h).
 * PARITY UNPINNED against the crate's source (absent); pinned only through the reference's call
 * sites and the crate's documented unit-test vectors reproduced in tests/test_oracle_plane.py.
 */
#include <string.h>

#include "r1_oracle.h"

void r1o_plane_pad(const r1o_plane *p, int w, int h, int xdec, int ydec) {
  const int bpp = p->bytes_per_px;
  const size_t stride = (size_t)p->stride;
  const int width = (w + xdec) >> xdec, height = (h + ydec) >> ydec;
  uint8_t *d = (uint8_t *)p->data;
  for (int y = 0; y < height; y++) {
    uint8_t *row = d + (size_t)(p->yorigin + y) * stride * bpp;
    for (int x = 0; x < p->xorigin; x++) memcpy(row + (size_t)x * bpp, row + (size_t)p->xorigin * bpp, bpp);
    for (int x = p->xorigin + width; x < p->stride; x++)
      memcpy(row + (size_t)x * bpp, row + (size_t)(p->xorigin + width - 1) * bpp, bpp);
  }
  for (int y = 0; y < p->yorigin; y++)
    memcpy(d + (size_t)y * stride * bpp, d + (size_t)p->yorigin * stride * bpp, stride * bpp);
  for (int y = p->yorigin + height; y < p->alloc_height; y++)
    memcpy(d + (size_t)y * stride * bpp, d + (size_t)(p->yorigin + height - 1) * stride * bpp, stride * bpp);
}

int r1o_plane_downsample(const r1o_plane *src, const r1o_plane *dst, int frame_w, int frame_h,
                         int dst_xdec, int dst_ydec) {
  const int width = (src->width + 1) / 2, height = (src->height + 1) / 2;
  if (dst->width != width || dst->height != height || dst->bytes_per_px != src->bytes_per_px) return -1;
  /* the crate asserts that the quads stay inside the source allocation */
  if (width * 2 > src->stride - src->xorigin || height * 2 > src->alloc_height - src->yorigin) return -1;
  const int hbd = src->bytes_per_px == 2;
  for (int y = 0; y < height; y++)
    for (int x = 0; x < width; x++) {
      uint32_t sum = 0;
      for (int dy = 0; dy < 2; dy++)
        for (int dx = 0; dx < 2; dx++) {
          const size_t i = (size_t)(src->yorigin + 2 * y + dy) * src->stride + src->xorigin + 2 * x + dx;
          sum += hbd ? ((const uint16_t *)src->data)[i] : ((const uint8_t *)src->data)[i];
        }
      const uint32_t avg = (sum + 2) >> 2;
      const size_t o = (size_t)(dst->yorigin + y) * dst->stride + dst->xorigin + x;
      if (hbd) ((uint16_t *)dst->data)[o] = (uint16_t)avg;
      else ((uint8_t *)dst->data)[o] = (uint8_t)avg;
    }
  r1o_plane_pad(dst, frame_w, frame_h, dst_xdec, dst_ydec);
  return 0;
}
