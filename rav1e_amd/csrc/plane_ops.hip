// plane_ops.hip -- the frame glue that keeps planes resident in HBM between the block stages:
//   r1_plane_pad         Plane::pad           via FramePad::pad src/frame/mod.rs:76-86, called on the
//                                             reconstruction at src/api/internal.rs:1436
//   r1_plane_downsample  Plane::downsampled   called at src/encoder.rs:476-477 (half / quarter
//                                             resolution inputs of the hierarchical motion search)
// (both are v_frame 0.3.9 functions; their algorithm is restated in oracle/plane.c's header).
//
// Border replication has no ordering in it once it is written as a gather: the value of ANY
// element of the allocation after pad(w, h) is the visible pixel at the coordinates clamped to
// [0, width) x [0, height).  So one launch covers the whole border (and, for the downsample, the
// whole new plane including its border): one workgroup per allocation row, 16-byte chunks.
// HBM-bound, and small: a 4K 8-bit border is 1.3 MB, a half-resolution plane 2.3 MB.
#include "common.hpp"

namespace {

template <int BPP> struct PxT;
template <> struct PxT<1> { typedef uint8_t t; };
template <> struct PxT<2> { typedef uint16_t t; };

// one workgroup per allocation row; a thread per 16-byte chunk of the row (VEC) or per pixel
template <int BPP, bool VEC>
__global__ __launch_bounds__(256) void k_plane_pad(R1Plane p, int width, int height) {
  typedef typename PxT<BPP>::t px;
  constexpr int PER = VEC ? 16 / BPP : 1;
  const int row = blockIdx.x;
  const int y = row - p.yorigin;
  const bool inside_rows = y >= 0 && y < height;
  const int cy = y < 0 ? 0 : (y >= height ? height - 1 : y);
  px *dst = (px *)p.data + (size_t)row * p.stride;
  const px *src = (const px *)p.data + (size_t)(p.yorigin + cy) * p.stride + p.xorigin;
  for (int c = threadIdx.x * PER; c < p.stride; c += 256 * PER) {
    const int x0 = c - p.xorigin;
    if (VEC && x0 >= 0 && x0 + PER <= width) {
      // visible columns: rows of the visible area keep them, rows above / below copy them
      if (!inside_rows) *(uint4 *)(dst + c) = *(const uint4 *)(src + x0);
      continue;
    }
    for (int k = 0; k < PER && c + k < p.stride; k++) {
      const int x = x0 + k;
      if (inside_rows && x >= 0 && x < width) continue;
      const int cx = x < 0 ? 0 : (x >= width ? width - 1 : x);
      dst[c + k] = src[cx];
    }
  }
}

// dst(X, Y) over the whole allocation of dst = box average of the source quad at the coordinates
// clamped to the padded-from size (pad_w, pad_h): the downsample and its pad() in one pass
template <int BPP, bool VEC>
__global__ __launch_bounds__(256) void k_plane_downsample(R1Plane s, R1Plane d, int pad_w, int pad_h) {
  typedef typename PxT<BPP>::t px;
  constexpr int PER = 8;   // outputs per thread
  const int row = blockIdx.x;
  const int y = row - d.yorigin;
  const int cy = y < 0 ? 0 : (y >= pad_h ? pad_h - 1 : y);
  px *dst = (px *)d.data + (size_t)row * d.stride;
  const px *s0 = (const px *)s.data + (size_t)(s.yorigin + 2 * cy) * s.stride + s.xorigin;
  const px *s1 = s0 + s.stride;
  for (int c = threadIdx.x * PER; c < d.stride; c += 256 * PER) {
    const int x0 = c - d.xorigin;
    if (VEC && x0 >= 0 && x0 + PER <= pad_w) {
      if (BPP == 1) {
        const uint4 a = *(const uint4 *)(s0 + 2 * x0), b = *(const uint4 *)(s1 + 2 * x0);
        const uint32_t aw[4] = {a.x, a.y, a.z, a.w}, bw[4] = {b.x, b.y, b.z, b.w};
        uint32_t o[2] = {0, 0};
#pragma unroll
        for (int k = 0; k < 4; k++) {   // a dword holds two quads' columns
          const uint32_t q0 = (aw[k] & 0xFF) + ((aw[k] >> 8) & 0xFF) + (bw[k] & 0xFF) + ((bw[k] >> 8) & 0xFF);
          const uint32_t q1 = ((aw[k] >> 16) & 0xFF) + (aw[k] >> 24) + ((bw[k] >> 16) & 0xFF) + (bw[k] >> 24);
          o[k >> 1] |= (((q0 + 2) >> 2) | (((q1 + 2) >> 2) << 8)) << (16 * (k & 1));
        }
        *(uint2 *)(dst + c) = make_uint2(o[0], o[1]);
      } else {
        uint32_t o[4];
#pragma unroll
        for (int h = 0; h < 2; h++) {
          const uint4 a = *(const uint4 *)(s0 + 2 * x0 + 8 * h), b = *(const uint4 *)(s1 + 2 * x0 + 8 * h);
          const uint32_t aw[4] = {a.x, a.y, a.z, a.w}, bw[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
          for (int k = 0; k < 4; k++) {   // a dword holds one quad's two columns
            const uint32_t q = (aw[k] & 0xFFFF) + (aw[k] >> 16) + (bw[k] & 0xFFFF) + (bw[k] >> 16);
            const uint32_t v = (q + 2) >> 2;
            if (k & 1) o[2 * h + (k >> 1)] |= v << 16;
            else o[2 * h + (k >> 1)] = v;
          }
        }
        *(uint4 *)(dst + c) = make_uint4(o[0], o[1], o[2], o[3]);
      }
      continue;
    }
    for (int k = 0; k < PER && c + k < d.stride; k++) {
      const int x = x0 + k;
      const int cx = x < 0 ? 0 : (x >= pad_w ? pad_w - 1 : x);
      const uint32_t q = (uint32_t)s0[2 * cx] + s0[2 * cx + 1] + s1[2 * cx] + s1[2 * cx + 1];
      dst[c + k] = (px)((q + 2) >> 2);
    }
  }
}

inline bool aligned16(const R1Plane *p) {
  return ((uintptr_t)p->data % 16) == 0 && ((size_t)p->stride * p->bytes_per_px) % 16 == 0 &&
         ((size_t)p->xorigin * p->bytes_per_px) % 16 == 0;
}

}  // namespace

extern "C" int r1_plane_pad(r1_ctx *ctx, const R1Plane *plane, int w, int h, int xdec, int ydec, void *stream) {
  R1_REQUIRE(ctx && plane && plane->data);
  R1_REQUIRE(plane->bytes_per_px == 1 || plane->bytes_per_px == 2);
  R1_REQUIRE(xdec >= 0 && xdec <= 2 && ydec >= 0 && ydec <= 2);
  const int width = (w + xdec) >> xdec, height = (h + ydec) >> ydec;
  R1_REQUIRE(width >= 1 && height >= 1);
  R1_REQUIRE(plane->xorigin + width <= plane->stride && plane->yorigin + height <= plane->alloc_height);
  hipStream_t st = (hipStream_t)stream;
  const bool vec = aligned16(plane);
  const dim3 grid(plane->alloc_height), block(256);
  if (plane->bytes_per_px == 1) {
    if (vec) hipLaunchKernelGGL((k_plane_pad<1, true>), grid, block, 0, st, *plane, width, height);
    else hipLaunchKernelGGL((k_plane_pad<1, false>), grid, block, 0, st, *plane, width, height);
  } else {
    if (vec) hipLaunchKernelGGL((k_plane_pad<2, true>), grid, block, 0, st, *plane, width, height);
    else hipLaunchKernelGGL((k_plane_pad<2, false>), grid, block, 0, st, *plane, width, height);
  }
  R1_HIP_CHECK(hipGetLastError());
  return R1_OK;
}

extern "C" int r1_plane_downsample(r1_ctx *ctx, const R1Plane *src, const R1Plane *dst, int frame_w,
                                   int frame_h, int dst_xdec, int dst_ydec, void *stream) {
  R1_REQUIRE(ctx && src && dst && src->data && dst->data && src->data != dst->data);
  R1_REQUIRE(src->bytes_per_px == dst->bytes_per_px && (src->bytes_per_px == 1 || src->bytes_per_px == 2));
  R1_REQUIRE(dst_xdec >= 1 && dst_xdec <= 2 && dst_ydec >= 1 && dst_ydec <= 2);
  const int width = (src->width + 1) / 2, height = (src->height + 1) / 2;
  R1_REQUIRE(dst->width == width && dst->height == height);
  // the crate's own assertions: every quad lies inside the source allocation
  R1_REQUIRE(width * 2 <= src->stride - src->xorigin && height * 2 <= src->alloc_height - src->yorigin);
  const int pad_w = (frame_w + dst_xdec) >> dst_xdec, pad_h = (frame_h + dst_ydec) >> dst_ydec;
  R1_REQUIRE(pad_w >= 1 && pad_h >= 1 && pad_w <= width && pad_h <= height);
  R1_REQUIRE(dst->xorigin + width <= dst->stride && dst->yorigin + height <= dst->alloc_height);
  hipStream_t st = (hipStream_t)stream;
  const bool vec = aligned16(src) && aligned16(dst) && dst->xorigin % 8 == 0;
  const dim3 grid(dst->alloc_height), block(256);
  if (src->bytes_per_px == 1) {
    if (vec) hipLaunchKernelGGL((k_plane_downsample<1, true>), grid, block, 0, st, *src, *dst, pad_w, pad_h);
    else hipLaunchKernelGGL((k_plane_downsample<1, false>), grid, block, 0, st, *src, *dst, pad_w, pad_h);
  } else {
    if (vec) hipLaunchKernelGGL((k_plane_downsample<2, true>), grid, block, 0, st, *src, *dst, pad_w, pad_h);
    else hipLaunchKernelGGL((k_plane_downsample<2, false>), grid, block, 0, st, *src, *dst, pad_w, pad_h);
  }
  R1_HIP_CHECK(hipGetLastError());
  return R1_OK;
}
