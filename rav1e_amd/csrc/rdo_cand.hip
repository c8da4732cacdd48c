// rdo_cand.hip -- the fused RDO candidate kernel: for every candidate
//   pred  = put_8tap(ref)            (src/mc.rs:250-353)
//   sad   = get_sad(org, pred)       (src/dist.rs:31-52)
//   satd  = get_satd(org, pred)      (src/dist.rs:156-221)
//   resid = org - pred               (diff, src/encoder.rs:1355-1381)
//   coeff = forward_transform(resid) (src/transform/forward.rs:71-161)
// in ONE launch; the prediction and the residual never leave the CU.
// This is the reference's per-candidate call chain
// (predict_inter_single src/predict.rs:304-331 -> compute_mv_rd
// src/me.rs:1445-1454 / rdo.rs:1328-1352 -> encode_tx_block
// src/encoder.rs:1533-1552) restructured as a batch.
//
// Mapping (wave = 64, one wave per workgroup): a wave owns NC = 64 / max(W,H)
// candidates.  Phase A stages each candidate's (H+7)x(W+7) reference window
// in LDS.  Phase B: lane = (candidate, column) runs the separable 8-tap filter
// down its column (8-deep register window), subtracts from the source pixels
// and keeps the whole residual COLUMN in registers.  SAD is a lane sum; SATD
// runs the vertical Hadamard on 8 registers and the horizontal one across the
// 8 neighbouring lanes with wave shuffles.  Phase C runs the column transform
// on the same registers, transposes through LDS (odd stride, aliasing the
// dead window), and phase D runs the row transform with lane = (candidate,
// row), storing coefficients in the reference's transposed 32x32-chunk order.
#include "mc_common.hpp"
#include <type_traits>
#include "tx_common.hpp"

namespace {
using r1tx::T;

// Horizontal (cross-lane) Hadamard over groups of TS adjacent lanes, applied
// to one register.  Lane pairs (l, l^m): the lower lane keeps a+b, the upper
// a-b -- the butterfly of dist.rs:55-57 with the data spread over lanes.
template <int TS>
__device__ __forceinline__ int32_t hadamard_lanes(int32_t x, int lane) {
#pragma unroll
  for (int m = 1; m < TS; m <<= 1) {
    const int32_t p = __shfl_xor(x, m, WAVE);
    x = (lane & m) ? p - x : x + p;
  }
  return x;
}

template <int BPP, int WL, int HL, typename CT>
__global__ __launch_bounds__(64) void k_rdo_cand(
    R1Plane org, R1Plane ref, const R1RdoCand *__restrict__ cands, int n,
    uint32_t *__restrict__ sad_out, uint32_t *__restrict__ satd_out,
    CT *__restrict__ coeffs, void *__restrict__ pred_out, r1tx::Shift3 sh) {
  constexpr int W = 1 << WL, H = 1 << HL;
  constexpr int P = W > H ? W : H, NC = 64 / P;
  constexpr int TS = (W < H ? W : H) == 4 ? 4 : 8;
  constexpr int WS = (((W + 7) * BPP + 3) >> 2) << 2;   // window row stride
  constexpr int WIN_BYTES = NC * (H + 7) * WS;
  constexpr int LSTRIDE = NC * W + 1;
  constexpr int TXB_BYTES = H * LSTRIDE * 4;
  constexpr int LDS_BYTES = WIN_BYTES > TXB_BYTES ? WIN_BYTES : TXB_BYTES;
  __shared__ __attribute__((aligned(16))) uint8_t smem[LDS_BYTES];
  T *buf = (T *)smem;

  const int lane = threadIdx.x;
  const int cl = lane / P, c = lane % P;
  const long long cand = (long long)blockIdx.x * NC + cl;
  const bool live = cand < n;
  R1RdoCand cd = {};
  if (live) cd = cands[cand];

  // ---- A: stage the reference window ----
  uint8_t *win = smem + cl * (H + 7) * WS;
  if (live) r1mc::stage_window<BPP>(win, WS, ref, cd.rx, cd.ry, W, H, c, P);
  __syncthreads();

  // ---- B: prediction column, residual, SAD / SATD ----
  T v[H];
#pragma unroll
  for (int r = 0; r < H; r++) v[r] = 0;
  uint32_t sad = 0;
  const bool col_live = live && c < W;
  if (col_live) {
    const uint8_t *po = px_addr<BPP>(org, cd.ox + c, cd.oy);
    const size_t so = (size_t)org.stride * BPP;
    uint8_t *pp = pred_out ? (uint8_t *)pred_out + ((size_t)cand * W * H + c) * BPP
                           : nullptr;
    r1mc::mc_column<BPP, false, H>(
        win, WS, c, W, H, cd.col_frac, cd.row_frac, cd.mode_x, cd.mode_y,
        ref.bit_depth, [&](int r, int32_t p) {
          const int32_t o = ld_px<BPP>(po + r * so);
          v[r] = o - p;
          if (pp) {
            if constexpr (BPP == 1) pp[(size_t)r * W] = (uint8_t)p;
            else *(uint16_t *)(pp + (size_t)r * W * 2) = (uint16_t)p;
          }
        });
#pragma unroll
    for (int r = 0; r < H; r++) sad += (uint32_t)iabs32(v[r]);
  }
  if (sad_out) {
    const uint32_t s = group_sum<P>(sad);
    if (live && c == 0) sad_out[cand] = s;
  }
  if (satd_out) {
    uint32_t acc = 0;
#pragma unroll
    for (int g = 0; g < H / TS; g++) {
      int32_t a[TS];
#pragma unroll
      for (int k = 0; k < TS; k++) a[k] = v[g * TS + k];
      // vertical pass on the lane's own TS rows (dist.rs:126-131)
      if constexpr (TS == 4) {
        const int32_t a0 = a[0] + a[1], a1 = a[0] - a[1];
        const int32_t a2 = a[2] + a[3], a3 = a[2] - a[3];
        a[0] = a0 + a2; a[1] = a1 + a3; a[2] = a0 - a2; a[3] = a1 - a3;
      } else {
        int32_t b[8], d[8];
#pragma unroll
        for (int k = 0; k < 4; k++) {
          b[2 * k] = a[2 * k] + a[2 * k + 1];
          b[2 * k + 1] = a[2 * k] - a[2 * k + 1];
        }
        d[0] = b[0] + b[2]; d[2] = b[0] - b[2];
        d[1] = b[1] + b[3]; d[3] = b[1] - b[3];
        d[4] = b[4] + b[6]; d[6] = b[4] - b[6];
        d[5] = b[5] + b[7]; d[7] = b[5] - b[7];
#pragma unroll
        for (int k = 0; k < 4; k++) {
          a[k] = d[k] + d[k + 4];
          a[k + 4] = d[k] - d[k + 4];
        }
      }
      // horizontal pass across the TS lanes of the tile (dist.rs:132-138)
#pragma unroll
      for (int k = 0; k < TS; k++)
        acc += (uint32_t)iabs32(hadamard_lanes<TS>(a[k], lane));
    }
    const uint32_t s = group_sum<P>(acc);
    constexpr int LN = TS == 4 ? 2 : 3;
    if (live && c == 0) satd_out[cand] = (s + ((1u << LN) >> 1)) >> LN;
  }
  if (!coeffs) return;   // wave-uniform: kernel argument

  // ---- C: column transform on the residual registers ----
  __syncthreads();  // every lane is done reading the window; LDS becomes buf
  const int tx_type = cd.tx_type;
  if (col_live) {
    if (r1tx::ud_flip(tx_type)) {
#pragma unroll
      for (int r = 0; r < H / 2; r++) {
        const T t = v[r]; v[r] = v[H - 1 - r]; v[H - 1 - r] = t;
      }
    }
#pragma unroll
    for (int r = 0; r < H; r++) v[r] = r1tx::shift_fwd(v[r], sh.s[0]);
    r1tx::fwd_1d<H>(v, r1tx::vtx_1d(tx_type));
    const int cc = cl * W + (r1tx::lr_flip(tx_type) ? W - 1 - c : c);
#pragma unroll
    for (int r = 0; r < H; r++)
      buf[r * LSTRIDE + cc] = r1tx::shift_fwd(v[r], sh.s[1]);
  }
  __syncthreads();
  // ---- D: row transform, transposed store ----
  {
    const int cl2 = lane / P, r = lane % P;   // P lanes per candidate again
    const long long cand2 = (long long)blockIdx.x * NC + cl2;
    if (cand2 < n && r < H) {
      const int tt = cands[cand2].tx_type;
      T u[W];
#pragma unroll
      for (int k = 0; k < W; k++) u[k] = buf[r * LSTRIDE + cl2 * W + k];
      r1tx::fwd_1d<W>(u, r1tx::htx_1d(tt));
      constexpr int OS = H < 32 ? H : 32, WC = W < 32 ? W : 32;
      CT *dst = coeffs + cand2 * (W * H) + (r >= 32 ? OS * WC : 0) + (r & 31);
#pragma unroll
      for (int cg = 0; cg < W; cg += 32)
#pragma unroll
        for (int k = 0; k < WC; k++)
          dst[H * cg + k * OS] = (CT)r1tx::shift_fwd(u[k + cg], sh.s[2]);
    }
  }
}

template <int BPP, int WL, int HL>
int launch(const R1Plane &org, const R1Plane &ref, const R1RdoCand *cands, int n,
           uint32_t *sad, uint32_t *satd, void *coeffs, void *pred,
           r1tx::Shift3 sh, hipStream_t st) {
  constexpr int W = 1 << WL, H = 1 << HL, P = W > H ? W : H, NC = 64 / P;
  typedef typename std::conditional<BPP == 1, int16_t, int32_t>::type CT;
  const unsigned grid = (unsigned)((n + NC - 1) / NC);
  hipLaunchKernelGGL((k_rdo_cand<BPP, WL, HL, CT>), dim3(grid), dim3(64), 0, st,
                     org, ref, cands, n, sad, satd, (CT *)coeffs, pred, sh);
  R1_HIP_CHECK(hipGetLastError());
  return R1_OK;
}

}  // namespace

extern "C" int r1_rdo_cand_batch(r1_ctx *ctx, const R1Plane *org,
                                 const R1Plane *ref, int w, int h, int tx_size,
                                 const R1RdoCand *cands, int n,
                                 uint32_t *sad_out, uint32_t *satd_out,
                                 void *coeffs, void *pred_out, void *stream) {
  R1_REQUIRE(ctx && org && ref);
  R1_REQUIRE(org->bytes_per_px == ref->bytes_per_px);
  R1_REQUIRE(org->bytes_per_px == 1 || org->bytes_per_px == 2);
  R1_REQUIRE(org->bit_depth == ref->bit_depth);
  R1_REQUIRE(tx_size >= 0 && tx_size < 19);
  R1_REQUIRE((1 << r1tx::kTxWLog2[tx_size]) == w &&
             (1 << r1tx::kTxHLog2[tx_size]) == h);
  if (n <= 0) return R1_OK;
  R1_REQUIRE(cands);
  hipStream_t st = (hipStream_t)stream;
  // the per-candidate tx_type selects the 1-D kernels on the device; the
  // shifts depend only on (tx_size, bit depth) for every non-WHT type
  const r1tx::Shift3 sh = r1tx::fwd_shift(tx_size, 0, org->bit_depth);
#define R1_RC_CASE(ID, WL, HL)                                                \
  case ID:                                                                    \
    return org->bytes_per_px == 1                                             \
               ? launch<1, WL, HL>(*org, *ref, cands, n, sad_out, satd_out,  \
                                   coeffs, pred_out, sh, st)                  \
               : launch<2, WL, HL>(*org, *ref, cands, n, sad_out, satd_out,  \
                                   coeffs, pred_out, sh, st);
  switch (tx_size) {
    R1_RC_CASE(0, 2, 2) R1_RC_CASE(1, 3, 3) R1_RC_CASE(2, 4, 4)
    R1_RC_CASE(3, 5, 5) R1_RC_CASE(4, 6, 6) R1_RC_CASE(5, 2, 3)
    R1_RC_CASE(6, 3, 2) R1_RC_CASE(7, 3, 4) R1_RC_CASE(8, 4, 3)
    R1_RC_CASE(9, 4, 5) R1_RC_CASE(10, 5, 4) R1_RC_CASE(11, 5, 6)
    R1_RC_CASE(12, 6, 5) R1_RC_CASE(13, 2, 4) R1_RC_CASE(14, 4, 2)
    R1_RC_CASE(15, 3, 5) R1_RC_CASE(16, 5, 3) R1_RC_CASE(17, 4, 6)
    R1_RC_CASE(18, 6, 4)
  }
#undef R1_RC_CASE
  return R1_EINVAL;
}
