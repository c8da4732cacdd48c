// dist_common.hpp -- the per-tile SAD / SATD arithmetic shared by dist.hip
// (candidates from HBM) and me.hip (sub-pel candidates predicted into LDS).
// Reference: get_sad src/dist.rs:31-52, get_satd 156-221, hadamard4_1d 61,
// hadamard8_1d 84, hadamard2d 122; get_weighted_sse 234-283, cdef_dist_kernel
// 302-372, apply_ssim_boost src/activity.rs:109-186 (see dist_scaled.hip).
#pragma once
#include "common.hpp"

namespace r1dist {

template <int N>
__device__ __forceinline__ void hadamard_1d(int32_t *d, int stride) {
  // butterfly order of dist.rs:71-78 / 95-117
  if constexpr (N == 4) {
    int32_t a0 = d[0] + d[stride], a1 = d[0] - d[stride];
    int32_t a2 = d[2 * stride] + d[3 * stride], a3 = d[2 * stride] - d[3 * stride];
    d[0] = a0 + a2; d[stride] = a1 + a3;
    d[2 * stride] = a0 - a2; d[3 * stride] = a1 - a3;
  } else {
    int32_t a[8], b[8];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      a[2 * k] = d[(2 * k) * stride] + d[(2 * k + 1) * stride];
      a[2 * k + 1] = d[(2 * k) * stride] - d[(2 * k + 1) * stride];
    }
    b[0] = a[0] + a[2]; b[2] = a[0] - a[2];
    b[1] = a[1] + a[3]; b[3] = a[1] - a[3];
    b[4] = a[4] + a[6]; b[6] = a[4] - a[6];
    b[5] = a[5] + a[7]; b[7] = a[5] - a[7];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      d[k * stride] = b[k] + b[k + 4];
      d[(k + 4) * stride] = b[k] - b[k + 4];
    }
  }
}

// Sum of |Hadamard(org - ref)| (SATD) or |org - ref| (SAD) over one TSxTS tile.
template <int BPP, int TS, bool SATD>
__device__ __forceinline__ uint32_t tile_dist(const uint8_t *po, size_t so,
                                              const uint8_t *pr, size_t sr) {
  int32_t d[TS * TS];
#pragma unroll
  for (int r = 0; r < TS; r++) {
    int32_t o[TS], q[TS];
    load_px_row<BPP, TS>(po + r * so, o);
    load_px_row<BPP, TS>(pr + r * sr, q);
#pragma unroll
    for (int c = 0; c < TS; c++) d[r * TS + c] = o[c] - q[c];
  }
  if constexpr (SATD) {
    // vertical then horizontal (hadamard2d, dist.rs:122-139)
#pragma unroll
    for (int c = 0; c < TS; c++) hadamard_1d<TS>(d + c, TS);
#pragma unroll
    for (int r = 0; r < TS; r++) hadamard_1d<TS>(d + r * TS, 1);
  }
  uint32_t s = 0;
#pragma unroll
  for (int i = 0; i < TS * TS; i++) s += (uint32_t)iabs32(d[i]);
  return s;
}

// ---- candidate-level pixel-domain distortions with the DistortionScale bias ----
// round(2^14 / (1 + x)): the reference's AREA_DIVISORS (dist.rs:290-297)
__device__ __forceinline__ uint32_t area_divisor(int area) {
  return (16384u + (uint32_t)(area >> 1)) / (uint32_t)area;
}

__device__ __forceinline__ uint32_t apply_ssim_boost(uint32_t input, uint32_t svar32,
                                                     uint32_t dvar32, int bit_depth) {
  const int coeff_shift = bit_depth - 8;
  const uint64_t svar = svar32 >> (2 * coeff_shift), dvar = dvar32 >> (2 * coeff_shift);
  const uint64_t C1 = 3355, C2 = 16128, C3 = 12338;
  const uint64_t RATIO = (((C1 << 15) / C3) + 1) >> 1;
  // ssim_boost_rsqrt: 1/sqrt(x) in Q(rshift) by a quadratic on the normalised mantissa
  const uint64_t x = C1 * C1 + svar * dvar;
  const int k = (63 - __builtin_clzll(x)) >> 1;
  const int s = 2 * k - 14;
  const uint16_t t = (uint16_t)(s > 0 ? x >> s : x << -s);
  const int rshift = (uint8_t)(14 + ((s + 16) >> 1));
  const int32_t nn = (int32_t)t - 32768;
  const int32_t inner = -13490 + ((nn * 6711) >> 15);
  const int32_t rsqrt = 23557 + ((nn * inner) >> 15);
  const uint64_t norm = (uint16_t)rsqrt;
  return (uint32_t)(((uint64_t)input * (((RATIO * (svar + dvar + C2)) * norm) >> 14)) >> rshift);
}

template <int BPP>
__device__ __forceinline__ void load_row(const uint8_t *p, int kw, int32_t *out) {
  if (kw == 8) {
    load_px_row<BPP, 8>(p, out);
  } else if (kw == 4) {
    load_px_row<BPP, 4>(p, out);
#pragma unroll
    for (int i = 4; i < 8; i++) out[i] = 0;
  } else {
    // a block cut by the edge of a frame whose size is not a multiple of 4 (cdef_dist_wxh hands
    // cdef_dist_kernel any kernel_w x kernel_h, rdo.rs:152-165): pixel by pixel, nothing past kw is read
#pragma unroll
    for (int i = 0; i < 8; i++) out[i] = i < kw ? (int32_t)ld_px<BPP>(p + i * BPP) : 0;
  }
}


// cdef_dist_kernel's fixed-point tail (dist.rs:350-372) + the ssim boost + the
// DistortionScale of the 8x8 importance block at (px, py), from the five sums
// of a tile of `area` pixels.
// BDT: the bit depth when the caller knows it at compile time (8 / 10: sum^2 fits 32 bits), 0 otherwise.
template <int BDT = 0>
__device__ __forceinline__ unsigned long long cdef_tile_tail(
    uint32_t sum_s, uint32_t sum_d, uint32_t sum_s2, uint32_t sum_d2, uint32_t sum_sd, int area, int px,
    int py, const uint32_t *__restrict__ scales, int scale_stride, int bit_depth) {
  const uint32_t sse = sum_d2 + sum_s2 - 2 * sum_sd;
  uint32_t svar, dvar;
  if ((area & (area - 1)) == 0) {
    // area = 2^la (every tile of 4 / 8 pixel sides): AREA_DIVISORS[area - 1] = 2^(14 - la) exactly, so
    //   (x * div + 2^13) >> 14 = (x + area / 2) >> la        and        (v * div + 128) >> 8 = v << (6 - la)
    // (dist.rs:352-367; "when w and h are powers of two, this can be done via shifting", the reference says)
    // -- six 64-bit multiplies of the general form become shifts
    const int la = 31 - __builtin_clz((unsigned)area);
    uint32_t ms, md;
    if constexpr (BDT == 8 || BDT == 10) {        // sum <= 64 * 1023: the square fits 32 bits
      ms = (sum_s * sum_s + (uint32_t)(area >> 1)) >> la;
      md = (sum_d * sum_d + (uint32_t)(area >> 1)) >> la;
    } else {
      ms = (uint32_t)(((unsigned long long)sum_s * sum_s + (unsigned long long)(area >> 1)) >> la);
      md = (uint32_t)(((unsigned long long)sum_d * sum_d + (unsigned long long)(area >> 1)) >> la);
    }
    svar = (sum_s2 > ms ? sum_s2 - ms : 0) << (6 - la);
    dvar = (sum_d2 > md ? sum_d2 - md : 0) << (6 - la);
  } else {
    const unsigned long long div = area_divisor(area);
    const uint32_t ms = (uint32_t)(((unsigned long long)sum_s * sum_s * div + 8192) >> 14);
    const uint32_t md = (uint32_t)(((unsigned long long)sum_d * sum_d * div + 8192) >> 14);
    svar = sum_s2 > ms ? sum_s2 - ms : 0;
    dvar = sum_d2 > md ? sum_d2 - md : 0;
    svar = (uint32_t)(((unsigned long long)svar * div + 128) >> 8);
    dvar = (uint32_t)(((unsigned long long)dvar * div + 128) >> 8);
  }
  const unsigned long long v = apply_ssim_boost(sse, svar, dvar, bit_depth);
  const unsigned long long sc =
      scales ? scales[(size_t)(py >> 3) * scale_stride + (px >> 3)] : (1u << 14);
  return (sc * v + 8192) >> 14;
}

// One 8x8 (or edge 4-wide / 4-high) tile of sse_wxh (KIND 2: four 4x4 cells,
// each weighted by the DistortionScale of its importance block, rdo.rs:177-224
// -> dist.rs:234-283) or of cdef_dist_wxh (KIND 3: cdef_dist_kernel + ssim
// boost + scale, rdo.rs:142-173).  (px, py): plane position of the tile in the
// SOURCE plane (selects the scale entries).  The weighted-SSE partials still
// need get_weighted_sse's final (sum + 32) / 64.
template <int BPP, int KIND>
__device__ __forceinline__ unsigned long long tile_scaled_dist(
    const uint8_t *po, size_t so, const uint8_t *pr, size_t sr, int kw, int kh, int px, int py,
    const uint32_t *__restrict__ scales, int scale_stride, int xdec, int ydec, int bit_depth) {
  unsigned long long acc = 0;
  if constexpr (KIND == 2) {
    // four 4x4 cells: [cy][cx]
    uint32_t cell[2][2] = {{0, 0}, {0, 0}};
#pragma unroll
    for (int r = 0; r < 8; r++) {
      if (r < kh) {
        int32_t a[8], b[8];
        load_row<BPP>(po + r * so, kw, a);
        load_row<BPP>(pr + r * sr, kw, b);
#pragma unroll
        for (int i = 0; i < 8; i++) {
          const int32_t d = a[i] - b[i];
          cell[r >> 2][i >> 2] += (uint32_t)(d * d);
        }
      }
    }
#pragma unroll
    for (int cy = 0; cy < 2; cy++)
#pragma unroll
      for (int cx = 0; cx < 2; cx++) {
        if (cy * 4 < kh && cx * 4 < kw) {
          const int lx = (px + cx * 4) << xdec, ly = (py + cy * 4) << ydec;
          const uint32_t sc =
              scales ? scales[(size_t)(ly >> 3) * scale_stride + (lx >> 3)] : (1u << 14);
          acc += ((unsigned long long)cell[cy][cx] * sc + 128) >> 8;
        }
      }
  } else {
    uint32_t sum_s = 0, sum_d = 0, sum_s2 = 0, sum_d2 = 0, sum_sd = 0;
#pragma unroll
    for (int r = 0; r < 8; r++) {
      if (r < kh) {
        int32_t a[8], b[8];
        load_row<BPP>(po + r * so, kw, a);
        load_row<BPP>(pr + r * sr, kw, b);
#pragma unroll
        for (int i = 0; i < 8; i++) {
          const uint32_t s = (uint32_t)a[i], d = (uint32_t)b[i];
          sum_s += s; sum_d += d;
          sum_s2 += s * s; sum_d2 += d * d; sum_sd += s * d;
        }
      }
    }
    acc = cdef_tile_tail(sum_s, sum_d, sum_s2, sum_d2, sum_sd, kw * kh, px, py, scales, scale_stride,
                         bit_depth);
  }
  return acc;
}

}  // namespace r1dist
