// cdef_common.hpp -- what the CDEF kernels share (cdef.hip: the filter entry points;
// cdef_search.hip: the strength search of rdo_loop_decision): the missing-pixel sentinel and edge
// flags of cdef_filter_block (src/cdef.rs:33-38, 161-196), the wave-wide direction search
// (cdef_find_dir, src/cdef.rs:84-143) and adjust_strength (src/cdef.rs:313-321).
#pragma once
#include "common.hpp"

namespace r1cdef {

constexpr int VERY_LARGE = 0x8000;
enum { HAVE_LEFT = 1, HAVE_RIGHT = 2, HAVE_TOP = 4, HAVE_BOTTOM = 8 };

template <int BPP>
__device__ __forceinline__ int32_t ldpx(const uint8_t *p) {
  if constexpr (BPP == 1) return *p;
  else return *(const uint16_t *)p;
}

// lane = pixel (i = lane >> 3, j = lane & 7); returns dir (all lanes), var via ref.
// Every lane adds its pixel into the 8 x 16 partial-sum table in LDS (ds_add:
// integer adds, order-free; measured cheaper than a gather formulation because
// the kernel is VALU-bound), then all 128 table entries are squared and
// weighted in parallel (840 / line length, cdef.rs:110-133) and summed inside
// 16-lane groups.  i32 adds wrap, so the order does not matter.
__device__ __forceinline__ int find_dir_wave(int32_t pixel /* of lane (i = lane >> 3, j = lane & 7) */,
                                             int coeff_shift, int32_t *part /* [128] LDS */,
                                             uint32_t &var) {
  const int lane = threadIdx.x & 63;
  const int i = lane >> 3, j = lane & 7;
  part[lane] = 0;
  part[lane + 64] = 0;
  __builtin_amdgcn_wave_barrier();
  const int32_t x = (pixel >> coeff_shift) - 128;
  atomicAdd(&part[0 * 16 + i + j], x);
  atomicAdd(&part[1 * 16 + i + j / 2], x);
  atomicAdd(&part[2 * 16 + i], x);
  atomicAdd(&part[3 * 16 + 3 + i - j / 2], x);
  atomicAdd(&part[4 * 16 + 7 + i - j], x);
  atomicAdd(&part[5 * 16 + 3 - i / 2 + j], x);
  atomicAdd(&part[6 * 16 + j], x);
  atomicAdd(&part[7 * 16 + i / 2 + j], x);
  __builtin_amdgcn_wave_barrier();
  auto weight = [](int d, int m) -> int32_t {
    constexpr int32_t DIV[9] = {0, 840, 420, 280, 210, 168, 140, 120, 105};
    if (d == 2 || d == 6) return m < 8 ? 105 : 0;
    if (d == 0 || d == 4) return m < 7 ? DIV[m + 1] : (m == 7 ? 105 : (m < 15 ? DIV[15 - m] : 0));
    return m < 3 ? DIV[2 * m + 2] : (m < 8 ? 105 : (m < 11 ? DIV[22 - 2 * m] : 0));
  };
  int32_t costs[8];
#pragma unroll
  for (int half = 0; half < 2; half++) {
    const int d = half * 4 + (lane >> 4), m = lane & 15;
    const int32_t pv = part[d * 16 + m];
    int32_t t = pv * pv * weight(d, m);
#pragma unroll
    for (int sft = 1; sft < 16; sft <<= 1) t += __shfl_xor(t, sft, 64);
#pragma unroll
    for (int q = 0; q < 4; q++) costs[half * 4 + q] = __shfl(t, q * 16, 64);
  }
  int best = 0;
  int32_t best_cost = costs[0];

#pragma unroll
  for (int d = 1; d < 8; d++)
    if (costs[d] > best_cost) { best_cost = costs[d]; best = d; }
  int32_t orth = costs[0];
#pragma unroll
  for (int d = 1; d < 8; d++)
    if (d == ((best + 4) & 7)) orth = costs[d];
  var = (uint32_t)((best_cost - orth) >> 10);
  return best;
}

__device__ __forceinline__ int adjust_strength(int strength, int var) {
  const int v6 = var >> 6;
  int i = 0;
  if (v6 != 0) {
    i = 31 - __clz(v6);
    i = i < 12 ? i : 12;
  }
  return var != 0 ? (strength * (4 + i) + 8) >> 4 : 0;
}

}  // namespace r1cdef
