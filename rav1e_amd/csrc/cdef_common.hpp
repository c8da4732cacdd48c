// cdef_common.hpp -- what the CDEF kernels share (cdef.hip: the filter entry points;
// cdef_search.hip: the strength search of rdo_loop_decision): the frame analysis kernel (a thread
// per 8x8 block), the packed-pair filter arithmetic, the missing-pixel sentinel and edge
// flags of cdef_filter_block (src/cdef.rs:33-38, 161-196), the wave-wide direction search
// (cdef_find_dir, src/cdef.rs:84-143) and adjust_strength (src/cdef.rs:313-321).
#pragma once
#include "common.hpp"

namespace r1cdef {

constexpr int VERY_LARGE = 0x8000;

// Block::skip of the four 4x4 units of an 8x8 block (src/cdef.rs:443-449: `skip = a.skip && b.skip
// && ...` on Rust bools).  s0 / s1: the two skip bytes of the upper / lower row as one u16 each.
// A byte is a bool: any non-zero value is true -- the filter kernel and the strength search must
// agree on that, whatever the host stores.
__device__ __forceinline__ int skip4(uint32_t s0, uint32_t s1) {
  return ((s0 & 0xffu) != 0) & ((s0 & 0xff00u) != 0) & ((s1 & 0xffu) != 0) & ((s1 & 0xff00u) != 0);
}
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

// ---- cdef_analyze_superblock for the whole frame (cdef.rs:340-373): one THREAD per 8x8 luma
// block, the 64 pixels in registers.  A wave per block spends most of its instructions moving
// partial sums between lanes (ds_add by 64 lanes: ~150 instructions per pixel); here every add is
// an add (about 300 adds + 170 multiplies per block, < 10 instructions per pixel) and the lanes of
// a wave read neighbouring blocks, i.e. whole rows.  The direction of a skipped block is never
// read, so it is computed for every block of the grid.
template <int BPP>
__global__ __launch_bounds__(64) void k_cdef_analyze(R1Plane luma, int nbx, int nby, int mi_cols,
                                                     int mi_rows, uint8_t *__restrict__ dir_out,
                                                     int32_t *__restrict__ var_out) {
  const int gbx = blockIdx.x * 64 + threadIdx.x, gby = blockIdx.y;
  if (gbx >= nbx || gby >= nby || gbx * 2 >= mi_cols || gby * 2 >= mi_rows) return;
  const int cs = luma.bit_depth - 8;
  int32_t x[8][8];
  const uint8_t *p0 = px_addr<BPP>(luma, gbx * 8, gby * 8);
#pragma unroll
  for (int i = 0; i < 8; i++) {
    load_px_row<BPP, 8>(p0 + (size_t)i * luma.stride * BPP, x[i]);
#pragma unroll
    for (int j = 0; j < 8; j++) x[i][j] = (x[i][j] >> cs) - 128;
  }
  // pair sums shared by the half-slope directions: h = two neighbours of a row, v = of a column
  int32_t h[8][4], v[4][8];
#pragma unroll
  for (int i = 0; i < 8; i++)
#pragma unroll
    for (int m = 0; m < 4; m++) h[i][m] = x[i][2 * m] + x[i][2 * m + 1];
#pragma unroll
  for (int q = 0; q < 4; q++)
#pragma unroll
    for (int j = 0; j < 8; j++) v[q][j] = x[2 * q][j] + x[2 * q + 1][j];
  int32_t part[8][15];
#pragma unroll
  for (int d = 0; d < 8; d++)
#pragma unroll
    for (int m = 0; m < 15; m++) part[d][m] = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
#pragma unroll
    for (int j = 0; j < 8; j++) {
      part[0][i + j] += x[i][j];
      part[4][7 + i - j] += x[i][j];
    }
#pragma unroll
    for (int m = 0; m < 4; m++) {
      part[1][i + m] += h[i][m];
      part[3][3 + i - m] += h[i][m];
      part[2][i] += h[i][m];
    }
  }
#pragma unroll
  for (int q = 0; q < 4; q++)
#pragma unroll
    for (int j = 0; j < 8; j++) {
      part[5][3 - q + j] += v[q][j];
      part[7][q + j] += v[q][j];
      part[6][j] += v[q][j];
    }
  // cost = sum of squared line sums * 840 / line length (cdef.rs:110-133); |line sum| <= 1024 and
  // its square * 840 < 2^30: the 24-bit multiplier is exact
  constexpr int32_t DIV[9] = {0, 840, 420, 280, 210, 168, 140, 120, 105};
  int32_t cost[8];
#pragma unroll
  for (int d = 0; d < 8; d++) {
    int32_t c = 0;
    if (d == 2 || d == 6) {
      int32_t sq = 0;
#pragma unroll
      for (int m = 0; m < 8; m++) sq += __mul24(part[d][m], part[d][m]);
      c = sq * 105;
    } else if (d == 0 || d == 4) {
#pragma unroll
      for (int m = 0; m < 7; m++)
        c += __mul24(__mul24(part[d][m], part[d][m]) + __mul24(part[d][14 - m], part[d][14 - m]), DIV[m + 1]);
      c += __mul24(__mul24(part[d][7], part[d][7]), 105);
    } else {
      int32_t sq = 0;
#pragma unroll
      for (int m = 3; m < 8; m++) sq += __mul24(part[d][m], part[d][m]);
      c = sq * 105;
#pragma unroll
      for (int m = 0; m < 3; m++)
        c += __mul24(__mul24(part[d][m], part[d][m]) + __mul24(part[d][10 - m], part[d][10 - m]), DIV[2 * m + 2]);
    }
    cost[d] = c;
  }
  int best = 0;
  int32_t best_cost = cost[0];
#pragma unroll
  for (int d = 1; d < 8; d++)
    if (cost[d] > best_cost) { best_cost = cost[d]; best = d; }   // first maximum (cdef.rs:64-73)
  int32_t orth = cost[4];
#pragma unroll
  for (int d = 1; d < 8; d++)
    if (best == d) orth = cost[(d + 4) & 7];
  dir_out[(size_t)gby * nbx + gbx] = (uint8_t)best;
  var_out[(size_t)gby * nbx + gbx] = (best_cost - orth) >> 10;
}

typedef int16_t i16x2 __attribute__((ext_vector_type(2)));
typedef uint16_t u16x2 __attribute__((ext_vector_type(2)));
union Pk {
  uint32_t u;
  i16x2 s;
  u16x2 v;
};
// the two pixels at LDS byte address a / a + 2 (a is only 2-byte aligned: a 32-bit read of an odd
// pixel position is an order of magnitude slower, and the compiler would merge two 16-bit reads into
// one -- hence asm; d16_hi loads do not keep the other half on this chip (SRAM ECC), hence two
// registers and one v_lshl_or to pair them).  The caller waits (lds_wait) before it looks at them.
__device__ __forceinline__ void lds_two(uint32_t a, uint32_t &lo, uint32_t &hi) {
  asm volatile("ds_read_u16 %0, %2\n\tds_read_u16 %1, %2 offset:2" : "=&v"(lo), "=&v"(hi) : "v"(a) : "memory");
}
__device__ __forceinline__ void lds_wait(uint32_t (&lo)[12], uint32_t (&hi)[12]) {
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(lo[0]), "+v"(lo[1]), "+v"(lo[2]), "+v"(lo[3]), "+v"(lo[4]), "+v"(lo[5]), "+v"(lo[6]),
                 "+v"(lo[7]), "+v"(lo[8]), "+v"(lo[9]), "+v"(lo[10]), "+v"(lo[11]), "+v"(hi[0]), "+v"(hi[1]),
                 "+v"(hi[2]), "+v"(hi[3]), "+v"(hi[4]), "+v"(hi[5]), "+v"(hi[6]), "+v"(hi[7]), "+v"(hi[8]),
                 "+v"(hi[9]), "+v"(hi[10]), "+v"(hi[11])
               :: "memory");
}

__device__ __forceinline__ i16x2 constrain2(i16x2 d, i16x2 thr, u16x2 sh) {
  const i16x2 ad = __builtin_elementwise_max(d, -d);
  i16x2 m = thr - (i16x2)((u16x2)ad >> sh);
  m = __builtin_elementwise_max(m, (i16x2)0);
  return __builtin_elementwise_max(__builtin_elementwise_min(d, m), -m);
}

inline int cdef_analyze_launch(const R1Plane *luma, int nbx, int nby, int mi_cols, int mi_rows,
                        uint8_t *dirs, int32_t *vars, hipStream_t st) {
  const dim3 grid((nbx + 63) / 64, nby);
  if (luma->bytes_per_px == 1)
    hipLaunchKernelGGL((k_cdef_analyze<1>), grid, dim3(64), 0, st, *luma, nbx, nby, mi_cols, mi_rows, dirs, vars);
  else
    hipLaunchKernelGGL((k_cdef_analyze<2>), grid, dim3(64), 0, st, *luma, nbx, nby, mi_cols, mi_rows, dirs, vars);
  R1_HIP_CHECK(hipGetLastError());
  return R1_OK;
}

}  // namespace r1cdef
