// mc_mfma.hip -- put_8tap / prep_8tap (8-bit) with the HORIZONTAL 8-tap pass on the
// matrix cores (reference: src/mc.rs:250-451).  BASELINE.json's north_star asks for "MFMA
// only for the batched put/prep 8-tap separable convolution in mc.rs"; this is that
// variant, kept beside the dot4 path (rdo_cand.hip mc8_column) so that both can be timed
// on the same box (tools/bench_mc_mfma.py, profiles/r02_mc_mfma_*).
//
// Horizontal pass as a banded-Toeplitz product on v_mfma_i32_16x16x32_i8:
//   D[16 rows][16 cols] = A[16 rows][32 K] * B[32 K][16 cols] + C
// The 16 output columns of one MFMA are TWO independent groups of 8 columns ("sub-tiles"),
// each with its own 16-byte K range:
//   A[i][16 s + k]  = biased window byte (p - 128) of sub-tile s, window row 16 t + i,
//                     byte 8 g + k            (g = column group of 8, k = 0..15)
//   B[16 s + k][8 s + m] = half-tap h[k - m] (0 <= k - m < 8), 0 elsewhere
//   C = 8192 + 2 (+32 for put): the -128 bias of the pixels and the rounding of
//       mid = (sum t*p + 4) >> 3 = (sum h*(p-128) + 8192 + 2) >> 2   (mc8_column's algebra)
// so D >> 2 IS the i16 intermediate of mc.rs:314-326 for 16 rows x 8 columns of two
// different windows -- two 8x8 candidates per instruction, or two column groups of a wider
// block.  8 of the 32 K entries of a column are non-zero (25 % of the MACs are useful).
// D's layout (col = lane & 15, rows 4 (lane >> 4) + reg) is column-major in groups of four
// rows, so the intermediates go to LDS as packed i16 pairs and the vertical pass -- lane =
// column, v_dot2_i32_i16 on row pairs, exactly mc8_column's -- reads its column back.
//
// VALU work per intermediate: shift + half a pack (was: 2 v_alignbyte + 2 v_dot4 + shift).
#include "mc_common.hpp"

namespace {
#include "mc_taps_packed.inc"

typedef int v4i __attribute__((ext_vector_type(4)));
typedef short v2s __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int32_t dot2_seed0(uint32_t a, uint32_t b) {
  int32_t r;
  asm("v_dot2_i32_i16 %0, %1, %2, 0" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ int32_t dot2_seed64(uint32_t a, uint32_t b) {
  int32_t r;
  asm("v_dot2_i32_i16 %0, %1, %2, 64" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

template <int WL, int HL, bool PREP>
__global__ __launch_bounds__(64) void k_mc_mfma(R1Plane ref, const R1McCand *__restrict__ cands,
                                                int n, void *__restrict__ dst) {
  constexpr int W = 1 << WL, H = 1 << HL;
  constexpr int P = W > H ? W : H, NC = 64 / P;
  constexpr int NT = (H + 7 + 15) / 16, HM = NT * 16;   // intermediate rows per column (padded)
  constexpr int G = W / 8;                              // column groups of 8
  constexpr int WSM = W + 8;                            // window row stride, multiple of 8
  constexpr int WIN = HM * WSM;                         // rows >= H + 7 are never staged: their
                                                        // intermediates are garbage nobody reads
  constexpr int ST = NC * NT * G;                       // sub-tiles of the wave, order (cand, t, g)
  static_assert(ST % 2 == 0, "two sub-tiles per MFMA");
  __shared__ __attribute__((aligned(16))) uint8_t win_s[NC * WIN];
  __shared__ __attribute__((aligned(16))) int16_t mid_s[NC * W * HM];

  const int lane = threadIdx.x;
  const int cl = lane / P, c = lane % P;
  const long long cand0 = (long long)blockIdx.x * NC;
  const long long cand = cand0 + cl;
  const bool live = cand < n;
  R1McCand cd = {};
  if (live) {
    cd = cands[cand];
    r1mc::stage_window<1, 0x80808080u>(win_s + cl * WIN, WSM, ref, cd.rx, cd.ry, W, H, c, P);
  }
  // B fragments: this lane's output column n = lane & 15 belongs to sub-tile (n >> 3) of each
  // MFMA; its K rows are non-zero only in the matching half of K (kg >> 1 == n >> 3).
  const int kg = lane >> 4, s = kg >> 1, half = kg & 1;
  const int sb = (lane >> 3) & 1, m = lane & 7;
  const uint32_t bmask = s == sb ? 0xffffffffu : 0u;
  constexpr int NB = W >= 16 ? NC : ST / 2;   // distinct B fragments a lane needs
  uint32_t b0[NB], b1[NB];
#pragma unroll
  for (int x = 0; x < NB; x++) {
    // W >= 16: fragment x belongs to candidate x (both sub-tiles of an MFMA are column groups
    // of the same candidate); W == 8: MFMA x pairs candidates 2x and 2x + 1
    const int ca = W >= 16 ? x : (2 * x + sb) / NT;
    R1McCand cx = {};
    if (cand0 + ca < n) cx = cands[cand0 + ca];
    const int fxi = cx.mode_x;   // W >= 8: never the 4-tap sets (mc.rs:241-246)
    const uint32_t *tb = &kTapB[fxi][cx.col_frac][m][2 * half];
    b0[x] = tb[0] & bmask;
    b1[x] = tb[1] & bmask;
  }
  __syncthreads();

  // ---- horizontal pass: ST / 2 MFMAs ----
  constexpr int32_t bias = 8192 + 2 + (PREP ? 0 : 32);
  const v4i cbias = {bias, bias, bias, bias};
  const int i = lane & 15;
  const uint32_t smask = (uint32_t)-s, sbmask = (uint32_t)-sb;
#pragma unroll
  for (int q = 0; q < ST / 2; q++) {
    // sub-tiles 2q and 2q + 1, decoded at compile time
    const int st0 = 2 * q, st1 = 2 * q + 1;
    const int g0 = st0 % G, t0 = (st0 / G) % NT, c0 = st0 / (G * NT);
    const int g1 = st1 % G, t1 = (st1 / G) % NT, c1 = st1 / (G * NT);
    const int a_off0 = c0 * WIN + 16 * t0 * WSM + 8 * g0;
    const int a_off1 = c1 * WIN + 16 * t1 * WSM + 8 * g1;
    const int aoff = a_off0 + (int)(smask & (uint32_t)(a_off1 - a_off0)) + i * WSM + 8 * half;
    const long a = *(const long *)(win_s + aoff);
    const int bx = W >= 16 ? c0 : q;
    const long b = (long)(((unsigned long long)b1[bx] << 32) | b0[bx]);
    const v4i d = __builtin_amdgcn_mfma_i32_16x16x32_i8(a, b, cbias, 0, 0, 0);
    // D: column n of sub-tile sb, rows 16 t + 4 kg .. + 3
    const int m_off0 = (c0 * W + 8 * g0) * HM + 16 * t0;
    const int m_off1 = (c1 * W + 8 * g1) * HM + 16 * t1;
    const int moff = m_off0 + (int)(sbmask & (uint32_t)(m_off1 - m_off0)) + m * HM + 4 * kg;
    const uint32_t lo = __builtin_amdgcn_perm((uint32_t)(d[1] >> 2), (uint32_t)(d[0] >> 2), 0x05040100u);
    const uint32_t hi = __builtin_amdgcn_perm((uint32_t)(d[3] >> 2), (uint32_t)(d[2] >> 2), 0x05040100u);
    *(uint2 *)(mid_s + moff) = make_uint2(lo, hi);
  }
  __syncthreads();
  if (!(live && c < W)) return;

  // ---- vertical pass: lane = column, packed row pairs from LDS (mc8_column's second half) ----
  const int fyi = cd.mode_y;     // H >= 8
  uint32_t ty[4], tz[5];
#pragma unroll
  for (int j = 0; j < 4; j++) ty[j] = kTapI16[fyi][cd.row_frac][j];
  tz[0] = ty[0] << 16;
#pragma unroll
  for (int j = 1; j < 4; j++) tz[j] = __builtin_amdgcn_alignbit(ty[j], ty[j - 1], 16);
  tz[4] = ty[3] >> 16;
  const uint32_t *mp = (const uint32_t *)(mid_s + (cl * W + c) * HM);
  int32_t pred[H];
  uint32_t pk[5];
#pragma unroll
  for (int j = 0; j < 4; j++) pk[j] = mp[j];
#pragma unroll
  for (int j = 0; j < H / 2; j++) {
    pk[4] = mp[j + 4];
    int32_t a0 = PREP ? dot2_seed64(pk[0], ty[0]) : dot2_seed0(pk[0], ty[0]);
    int32_t a1 = PREP ? dot2_seed64(pk[0], tz[0]) : dot2_seed0(pk[0], tz[0]);
#pragma unroll
    for (int k = 1; k < 4; k++)
      a0 = __builtin_amdgcn_sdot2(__builtin_bit_cast(v2s, pk[k]), __builtin_bit_cast(v2s, ty[k]), a0, false);
#pragma unroll
    for (int k = 1; k < 5; k++)
      a1 = __builtin_amdgcn_sdot2(__builtin_bit_cast(v2s, pk[k]), __builtin_bit_cast(v2s, tz[k]), a1, false);
    if constexpr (PREP) {
      pred[2 * j] = a0 >> 7;
      pred[2 * j + 1] = a1 >> 7;
    } else {
      a0 >>= 11;
      a1 >>= 11;
      pred[2 * j] = a0 < 0 ? 0 : (a0 > 255 ? 255 : a0);
      pred[2 * j + 1] = a1 < 0 ? 0 : (a1 > 255 ? 255 : a1);
    }
#pragma unroll
    for (int k = 0; k < 4; k++) pk[k] = pk[k + 1];
  }
  if constexpr (PREP) {
    uint16_t *pp = (uint16_t *)dst + (size_t)cand * W * H + c;
#pragma unroll
    for (int r = 0; r < H; r++) pp[(size_t)r * W] = (uint16_t)pred[r];
  } else {
    uint8_t *pp = (uint8_t *)dst + (size_t)cand * W * H + c;
#pragma unroll
    for (int r = 0; r < H; r++) pp[(size_t)r * W] = (uint8_t)pred[r];
  }
}

template <int WL, int HL>
int launch(bool prep, const R1Plane &ref, const R1McCand *cands, int n, void *dst, hipStream_t st) {
  constexpr int W = 1 << WL, H = 1 << HL, P = W > H ? W : H, NC = 64 / P;
  const unsigned grid = (unsigned)((n + NC - 1) / NC);
  if (prep)
    hipLaunchKernelGGL((k_mc_mfma<WL, HL, true>), dim3(grid), dim3(64), 0, st, ref, cands, n, dst);
  else
    hipLaunchKernelGGL((k_mc_mfma<WL, HL, false>), dim3(grid), dim3(64), 0, st, ref, cands, n, dst);
  R1_HIP_CHECK(hipGetLastError());
  return R1_OK;
}

}  // namespace

// Experimental twin of r1_mc_put_batch / r1_mc_prep_batch (same arguments, same results):
// 8-bit planes, square blocks 8 / 16 / 32 / 64.  prep != 0 selects prep_8tap.
extern "C" int r1_mc_batch_mfma(r1_ctx *ctx, int prep, const R1Plane *ref, int w, int h,
                                const R1McCand *cands, int n, void *dst, void *stream) {
  R1_REQUIRE(ctx && ref && ref->bytes_per_px == 1 && ref->bit_depth == 8);
  R1_REQUIRE(w == h && (w == 8 || w == 16 || w == 32 || w == 64));
  if (n <= 0) return R1_OK;
  R1_REQUIRE(cands && dst);
  hipStream_t st = (hipStream_t)stream;
  switch (w) {
    case 8: return launch<3, 3>(prep != 0, *ref, cands, n, dst, st);
    case 16: return launch<4, 4>(prep != 0, *ref, cands, n, dst, st);
    case 32: return launch<5, 5>(prep != 0, *ref, cands, n, dst, st);
    default: return launch<6, 6>(prep != 0, *ref, cands, n, dst, st);
  }
}
