// dist_common.hpp -- the per-tile SAD / SATD arithmetic shared by dist.hip
// (candidates from HBM) and me.hip (sub-pel candidates predicted into LDS).
// Reference: get_sad src/dist.rs:31-52, get_satd 156-221, hadamard4_1d 61,
// hadamard8_1d 84, hadamard2d 122.
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

}  // namespace r1dist
