// dist.hip -- batched SAD / SATD (reference: src/dist.rs get_sad 31-52,
// get_satd 156-221; dispatch tables src/asm/x86/dist/mod.rs:483-729).
//
// Mapping (gfx950, wave = 64): one LANE owns one TSxTS tile of one candidate
// (TS = 4 when min(w,h) == 4, else 8 -- the reference's Hadamard size rule,
// dist.rs:166).  The lane pulls its TS rows of org and ref with unaligned
// vector loads (global_load_dword / x2 / x4), keeps the TSxTS difference in
// registers, runs both Hadamard passes in registers (no cross-lane traffic)
// and reduces |coeff|.  The tiles of one candidate are consecutive lanes, so
// the per-candidate total is a segmented wave reduction (DPP/ds_bpermute),
// plus one LDS hop for candidates with more than 64 tiles (>= 64x128).
// Integer adds are exact in any order, so the result is bit-identical to the
// reference's serial sum.
#include "common.hpp"
#include "dist_common.hpp"

namespace {
using r1dist::tile_dist;

// tiles-per-candidate tpc = (w/TS)*(h/TS) is a power of two.
template <int BPP, int TS, bool SATD>
__global__ __launch_bounds__(256) void k_dist(R1Plane org, R1Plane ref,
                                              int wt_log2, int tpc_log2,
                                              const R1DistCand *__restrict__ cands,
                                              int n, uint32_t *__restrict__ out) {
  __shared__ uint32_t wave_part[4];
  const int tid = threadIdx.x;
  // XCD-aware: workgroup i runs on XCD i % 8; XCD x takes the x-th contiguous eighth of the tile list, so the
  // K candidates of a block (same source tile, overlapping reference tiles) meet in ONE L2 (grid = multiple of 8)
  const unsigned wg = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const long long gt = (long long)wg * 256 + tid;  // global tile id
  const int cand = (int)(gt >> tpc_log2);
  const int t = (int)(gt & ((1 << tpc_log2) - 1));
  const bool live = cand < n;
  uint32_t s = 0;
  if (live) {
    const R1DistCand c = cands[cand];
    const int tx = t & ((1 << wt_log2) - 1), ty = t >> wt_log2;
    const uint8_t *po = px_addr<BPP>(org, c.ox + tx * TS, c.oy + ty * TS);
    const uint8_t *pr = px_addr<BPP>(ref, c.rx + tx * TS, c.ry + ty * TS);
    s = tile_dist<BPP, TS, SATD>(po, (size_t)org.stride * BPP, pr,
                                 (size_t)ref.stride * BPP);
  }
  constexpr int LN = TS == 4 ? 2 : 3;
  if (tpc_log2 <= 6) {
    // segmented reduction inside the wave
    for (int m = 1; m < (1 << tpc_log2); m <<= 1) s += __shfl_xor(s, m, WAVE);
    if (live && t == 0) out[cand] = SATD ? (s + ((1u << LN) >> 1)) >> LN : s;
  } else {
    // 128 or 256 tiles per candidate: wave totals meet in LDS
    s = group_sum<64>(s);
    if ((tid & 63) == 0) wave_part[tid >> 6] = s;
    __syncthreads();
    const int waves_per_cand = 1 << (tpc_log2 - 6);  // 2 or 4
    if (live && t == 0) {
      uint32_t tot = 0;
      const int w0 = tid >> 6;
      for (int i = 0; i < waves_per_cand; i++) tot += wave_part[w0 + i];
      out[cand] = SATD ? (tot + ((1u << LN) >> 1)) >> LN : tot;
    }
  }
}

template <int BPP, int TS>
int launch_dist(int kind, const R1Plane &org, const R1Plane &ref, int w, int h,
                const R1DistCand *cands, int n, uint32_t *out, hipStream_t st) {
  const int wt_log2 = r1_ilog2(w / TS), tpc_log2 = wt_log2 + r1_ilog2(h / TS);
  const long long tiles = (long long)n << tpc_log2;
  const unsigned grid = ((unsigned)((tiles + 255) / 256) + 7u) & ~7u;   // whole rounds over the 8 XCDs
  if (kind == R1_DIST_SAD)
    hipLaunchKernelGGL((k_dist<BPP, TS, false>), dim3(grid), dim3(256), 0, st,
                       org, ref, wt_log2, tpc_log2, cands, n, out);
  else
    hipLaunchKernelGGL((k_dist<BPP, TS, true>), dim3(grid), dim3(256), 0, st,
                       org, ref, wt_log2, tpc_log2, cands, n, out);
  R1_HIP_CHECK(hipGetLastError());
  return R1_OK;
}

}  // namespace

extern "C" int r1_dist_batch(r1_ctx *ctx, int kind, const R1Plane *org,
                             const R1Plane *ref, int w, int h,
                             const R1DistCand *cands, int n, uint32_t *out,
                             void *stream) {
  R1_REQUIRE(ctx && org && ref);
  R1_REQUIRE(kind == R1_DIST_SAD || kind == R1_DIST_SATD);
  R1_REQUIRE(org->bytes_per_px == ref->bytes_per_px);
  R1_REQUIRE(org->bytes_per_px == 1 || org->bytes_per_px == 2);
  R1_REQUIRE(r1_is_pow2(w) && r1_is_pow2(h) && w >= 4 && h >= 4 && w <= 128 &&
             h <= 128);
  if (n <= 0) return R1_OK;
  R1_REQUIRE(cands && out);
  hipStream_t st = (hipStream_t)stream;
  const bool small = (w < h ? w : h) == 4;
  // SAD has no tile-size rule; use the widest tile the block allows.
  if (org->bytes_per_px == 1)
    return small ? launch_dist<1, 4>(kind, *org, *ref, w, h, cands, n, out, st)
                 : launch_dist<1, 8>(kind, *org, *ref, w, h, cands, n, out, st);
  return small ? launch_dist<2, 4>(kind, *org, *ref, w, h, cands, n, out, st)
               : launch_dist<2, 8>(kind, *org, *ref, w, h, cands, n, out, st);
}
