// common.hpp -- shared host/device helpers of librav1e_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/rav1e_amd.h"

struct r1_ctx {
  int device;
  hipStream_t own_stream;  // used by the compat shims
  void *stage;             // device staging for the compat shims
  size_t stage_bytes;
  void *pinned;
  size_t pinned_bytes;
};

void r1_set_error(const char *fmt, ...);

#define R1_HIP_CHECK(expr)                                              \
  do {                                                                  \
    hipError_t e_ = (expr);                                             \
    if (e_ != hipSuccess) {                                             \
      r1_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr,        \
                   hipGetErrorString(e_));                              \
      return R1_EHIP;                                                   \
    }                                                                   \
  } while (0)

#define R1_REQUIRE(cond)                                                \
  do {                                                                  \
    if (!(cond)) {                                                      \
      r1_set_error("%s:%d: requirement failed: %s", __FILE__, __LINE__, \
                   #cond);                                              \
      return R1_EINVAL;                                                 \
    }                                                                   \
  } while (0)

static inline bool r1_is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }
static inline int r1_ilog2(int v) {
  int l = 0;
  while ((1 << l) < v) l++;
  return l;
}

#define WAVE 64

// ---- device helpers ----
// Byte address of pixel (x, y) of a plane; coordinates may be negative or
// beyond width/height as long as they stay inside the padded allocation.
template <int BPP>
__device__ __forceinline__ const uint8_t *px_addr(const R1Plane &p, int x,
                                                  int y) {
  return (const uint8_t *)p.data +
         ((size_t)(p.yorigin + y) * (size_t)p.stride + (size_t)(p.xorigin + x)) *
             BPP;
}

// Unaligned vector loads: gfx950 runs in unaligned-access mode, so these
// compile to single global_load_dword{,x2,x4} at any byte address.
struct __attribute__((packed)) U32x1 { uint32_t a; };
struct __attribute__((packed)) U32x2 { uint32_t a, b; };
struct __attribute__((packed)) U32x4 { uint32_t a, b, c, d; };

__device__ __forceinline__ uint32_t ld_u32(const uint8_t *p) {
  U32x1 v;
  __builtin_memcpy(&v, p, 4);
  return v.a;
}
__device__ __forceinline__ U32x2 ld_u32x2(const uint8_t *p) {
  U32x2 v;
  __builtin_memcpy(&v, p, 8);
  return v;
}
__device__ __forceinline__ U32x4 ld_u32x4(const uint8_t *p) {
  U32x4 v;
  __builtin_memcpy(&v, p, 16);
  return v;
}

// Load N consecutive pixels starting at byte address p into int32 lanes.
template <int BPP, int N>
__device__ __forceinline__ void load_px_row(const uint8_t *p, int32_t *out) {
  static_assert(N == 4 || N == 8, "row of 4 or 8 pixels");
  if constexpr (BPP == 1 && N == 4) {
    uint32_t v = ld_u32(p);
#pragma unroll
    for (int i = 0; i < 4; i++) out[i] = (v >> (8 * i)) & 0xff;
  } else if constexpr (BPP == 1 && N == 8) {
    U32x2 v = ld_u32x2(p);
#pragma unroll
    for (int i = 0; i < 4; i++) {
      out[i] = (v.a >> (8 * i)) & 0xff;
      out[4 + i] = (v.b >> (8 * i)) & 0xff;
    }
  } else if constexpr (BPP == 2 && N == 4) {
    U32x2 v = ld_u32x2(p);
    out[0] = v.a & 0xffff; out[1] = v.a >> 16;
    out[2] = v.b & 0xffff; out[3] = v.b >> 16;
  } else {
    U32x4 v = ld_u32x4(p);
    out[0] = v.a & 0xffff; out[1] = v.a >> 16;
    out[2] = v.b & 0xffff; out[3] = v.b >> 16;
    out[4] = v.c & 0xffff; out[5] = v.c >> 16;
    out[6] = v.d & 0xffff; out[7] = v.d >> 16;
  }
}

template <int BPP>
__device__ __forceinline__ int32_t ld_px(const uint8_t *p) {
  if constexpr (BPP == 1) return *p;
  else return *(const uint16_t *)p;
}

__device__ __forceinline__ int32_t iabs32(int32_t v) { return v < 0 ? -v : v; }

// sum over a power-of-two group of lanes (group size <= 64); every lane of the
// group ends up with the group total.
template <int GROUP>
__device__ __forceinline__ uint32_t group_sum(uint32_t v) {
#pragma unroll
  for (int m = 1; m < GROUP; m <<= 1) v += __shfl_xor(v, m, WAVE);
  return v;
}
