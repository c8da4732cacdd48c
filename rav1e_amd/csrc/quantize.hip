// quantize.hip -- batched quantize + dequantize
// (reference: QuantizationContext::{update,quantize} src/quantize/mod.rs:219-355,
// rust::dequantize 363-384, scan orders src/scan_order.rs, coded size
// av1_get_coded_tx_size src/context/mod.rs:95-102; call site
// src/encoder.rs:1556-1606).
//
// The AC loop of the reference is a serial recurrence over scan order: the
// rounding offset of coefficient i depends on `level_mode`, which depends on
// the quantized values before it (mod.rs:317-339).  `level_mode` has two
// states, so coefficient i is a function {0,1} -> {0,1} (plus one output per
// state) and function composition is associative: the recurrence becomes a
// prefix scan over 2-bit function codes (quant_common.hpp: quantize_group).
//
// Mapping (wave = 64): G = min(64, coded area) lanes own one block, 64/G
// blocks per wave, 4 waves per workgroup.
//  1 the block's coded coefficients are staged into LDS with coalesced loads;
//  2 lane l owns the contiguous run [l*NPL, (l+1)*NPL) of scan positions:
//    gathers lds[scan[i]] into registers, eob-1 = max scan index with
//    |c| >= deadzone (group max);
//  3 DC by the group's lane 0; each lane evaluates both candidate outcomes
//    (level_mode 0 / 1) of its run and composes the run's transition function
//    sequentially, one Hillis-Steele scan inside the group composes the lanes,
//    a replay pass picks the results, which overwrite lds[scan[i]] in place;
//  4 the group writes qcoeffs (and rcoeffs = dequantize(qcoeffs)) back with
//    coalesced stores; positions >= eob are zero (the reference relies on a
//    pre-zeroed buffer, encoder.rs:1518-1521).
// Exact u32 division by the quantizer uses the reference's own reciprocal
// form divu_pair (mod.rs:129-157), computed on the host per launch.
#include <stdlib.h>

#include <vector>

#include "tx_common.hpp"
#include "quant_common.hpp"

namespace {

template <typename CT, int GL, int NPL, bool DIST>
__global__ __launch_bounds__(256) void k_quantize(
    const CT *__restrict__ coeffs, int coeff_stride, int n, int area,
    const uint16_t *__restrict__ scan, r1q::QParams qp, CT *__restrict__ qcoeffs,
    uint16_t *__restrict__ eobs, CT *__restrict__ rcoeffs, int full_area, int tx_size,
    int q_bin, unsigned long long *__restrict__ tx_dist,
    unsigned long long *__restrict__ est_rate) {
  constexpr int G = 1 << GL, BPW = 64 / G;     // lanes per block, blocks per wave
  __shared__ int32_t lds[4][BPW * G * NPL];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int g = lane >> GL, l = lane & (G - 1);
  const long long blk = ((long long)blockIdx.x * 4 + wave) * BPW + g;
  const bool live = blk < n;
  int32_t *mine = lds[wave] + g * (G * NPL);
  // 1: stage the wave's BPW blocks (consecutive lanes = consecutive coefficients)
  constexpr int AREA = G * NPL;
  const long long blk0 = ((long long)blockIdx.x * 4 + wave) * BPW;
#pragma unroll
  for (int k = 0; k < NPL; k++) {
    const int e = k * 64 + lane, bl = e / AREA, idx = e % AREA;
    if (blk0 + bl < n) lds[wave][e] = (int32_t)coeffs[(blk0 + bl) * coeff_stride + idx];
  }
  __builtin_amdgcn_wave_barrier();
  // 2-3: scan-order gather, eob, DC, AC prefix scan (quant_common.hpp)
  unsigned long long tail = 0;
  if constexpr (DIST) {
    if (live) {   // coefficients beyond the coded area (64-point sizes): rcoeff = 0
      const CT *src = coeffs + blk * coeff_stride;
      for (int i = area + l; i < full_area; i += G) {
        const int32_t c = (int32_t)src[i];
        tail += (unsigned long long)(long long)(int32_t)((uint32_t)c * (uint32_t)c);
      }
    }
  }
  int eob = 0;
  unsigned long long dist = 0;
  r1q::quantize_group<CT, GL, NPL, DIST>(mine, g << GL, l, live, scan, qp, tail, eob, dist);
  if (live && l == 0) {
    eobs[blk] = (uint16_t)eob;
    if constexpr (DIST) {
      tx_dist[blk] = dist;
      if (est_rate) est_rate[blk] = r1q::estimate_rate(q_bin, tx_size, dist);
    }
  }
  __builtin_amdgcn_wave_barrier();
  // 4: write back (coalesced, dense blocks), dequantize on the way (mod.rs:372-383)
  const int32_t off = (1 << qp.lts) - 1;
#pragma unroll
  for (int k = 0; k < NPL; k++) {
    const int e = k * 64 + lane, bl = e / AREA, idx = e % AREA;
    if (blk0 + bl < n) {
      const int32_t q = (int32_t)(CT)lds[wave][e];
      qcoeffs[(blk0 + bl) * AREA + idx] = (CT)q;
      if (rcoeffs) {
        const uint32_t quant = idx == 0 ? qp.dc_q : qp.ac_q;
        rcoeffs[(blk0 + bl) * AREA + idx] =
            (CT)((int32_t)((uint32_t)q * quant + (uint32_t)((q >> 31) & off)) >> qp.lts);
      }
    }
  }
}

template <typename CT>
__global__ __launch_bounds__(256) void k_dequantize(const CT *__restrict__ q,
                                                    CT *__restrict__ r, long long total,
                                                    int area, uint32_t dc_q, uint32_t ac_q,
                                                    int lts) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int32_t c = (int32_t)q[i];
  const uint32_t quant = (i % area) == 0 ? dc_q : ac_q;
  const int32_t off = (1 << lts) - 1;
  r[i] = (CT)((int32_t)((uint32_t)c * quant + (uint32_t)((c >> 31) & off)) >> lts);
}

using r1q::coded_dim;
using r1q::log_tx_scale;

// scan order rule (see oracle/quantize.c header; verified against the
// reference's 42 literal tables by tests/golden/gen_quant_golden.py)
void gen_scan(int kind, int W, int H, uint16_t *scan) {
  if (kind == 2) {
    for (int i = 0; i < W * H; i++) scan[i] = (uint16_t)i;
  } else if (kind == 1) {
    for (int r = 0; r < H; r++)
      for (int c = 0; c < W; c++) scan[r * W + c] = (uint16_t)(c * H + r);
  } else {
    int k = 0;
    for (int d = 0; d < W + H - 1; d++) {
      const bool down = W > H || (W == H && d % 2 == 0);
      const int r0 = d < W ? 0 : d - W + 1, r1 = d < H ? d : H - 1;
      if (down)
        for (int r = r1; r >= r0; r--) scan[k++] = (uint16_t)((d - r) * H + r);
      else
        for (int r = r0; r <= r1; r++) scan[k++] = (uint16_t)((d - r) * H + r);
    }
  }
}

template <typename CT, bool DIST>
int launch_q(r1_ctx *ctx, const void *coeffs, int coeff_stride, int n, int tx_size,
             int kind, const r1q::QParams &qp, void *q, uint16_t *eobs, void *r,
             int q_bin, uint64_t *tx_dist, uint64_t *est_rate, hipStream_t st) {
  const int full_area = 1 << (r1tx::kTxWLog2[tx_size] + r1tx::kTxHLog2[tx_size]);
  const int area = coded_dim(r1tx::kTxWLog2[tx_size]) * coded_dim(r1tx::kTxHLog2[tx_size]);
  const uint16_t *scan = ctx->scan_dev + ctx->scan_off[tx_size][kind];
#define R1_Q_LAUNCH(GL, NPL)                                                          \
  do {                                                                                \
    constexpr int BPWG = 4 * (64 >> GL);                                              \
    const unsigned grid = (unsigned)((n + BPWG - 1) / BPWG);                          \
    hipLaunchKernelGGL((k_quantize<CT, GL, NPL, DIST>), dim3(grid), dim3(256), 0, st, \
                       (const CT *)coeffs, coeff_stride, n, area, scan, qp, (CT *)q,  \
                       eobs, (CT *)r, full_area, tx_size, q_bin,                      \
                       (unsigned long long *)tx_dist, (unsigned long long *)est_rate); \
  } while (0)
  switch (area) {
    // at least 4 coefficients per lane: the per-block reductions and the cross-lane scan
    // are paid per lane group, the run is worked off sequentially
    case 16: R1_Q_LAUNCH(2, 4); break;
    case 32: R1_Q_LAUNCH(3, 4); break;
    case 64: R1_Q_LAUNCH(4, 4); break;
    case 128: R1_Q_LAUNCH(5, 4); break;
    case 256: R1_Q_LAUNCH(6, 4); break;
    case 512: R1_Q_LAUNCH(6, 8); break;
    case 1024: R1_Q_LAUNCH(6, 16); break;
    default: return R1_EINVAL;
  }
#undef R1_Q_LAUNCH
  R1_HIP_CHECK(hipGetLastError());
  return R1_OK;
}

}  // namespace

int r1_scan_tables_create(r1_ctx *c) {
  std::vector<uint16_t> all;
  for (int ts = 0; ts < 19; ts++) {
    const int W = coded_dim(r1tx::kTxWLog2[ts]), H = coded_dim(r1tx::kTxHLog2[ts]);
    for (int kind = 0; kind < 3; kind++) {
      c->scan_off[ts][kind] = (uint32_t)all.size();
      all.resize(all.size() + (size_t)W * H);
      gen_scan(kind, W, H, all.data() + c->scan_off[ts][kind]);
    }
  }
  R1_HIP_CHECK(hipMalloc((void **)&c->scan_dev, all.size() * sizeof(uint16_t)));
  R1_HIP_CHECK(hipMemcpy(c->scan_dev, all.data(), all.size() * sizeof(uint16_t),
                         hipMemcpyHostToDevice));
  return R1_OK;
}

void r1_scan_tables_destroy(r1_ctx *c) {
  if (c->scan_dev) (void)hipFree(c->scan_dev);
  c->scan_dev = nullptr;
}

namespace {
int quantize_common(r1_ctx *ctx, const void *coeffs, int coeff_stride, int n, int tx_size,
                    int tx_type, const R1QuantParams *p, int coeff_bytes, void *qcoeffs,
                    uint16_t *eobs, void *rcoeffs, uint64_t *tx_dist, uint64_t *est_rate,
                    bool with_dist, void *stream) {
  R1_REQUIRE(ctx && p);
  // av1_scan_orders has TX_TYPES = 16 columns: WHT_WHT would index out of it
  R1_REQUIRE(r1tx::valid_av1_transform(tx_size, tx_type) && tx_type < 16);
  R1_REQUIRE(p->bit_depth == 8 || p->bit_depth == 10 || p->bit_depth == 12);
  R1_REQUIRE(coeff_bytes == 2 || coeff_bytes == 4);
  const int area = coded_dim(r1tx::kTxWLog2[tx_size]) * coded_dim(r1tx::kTxHLog2[tx_size]);
  const int full_area = 1 << (r1tx::kTxWLog2[tx_size] + r1tx::kTxHLog2[tx_size]);
  R1_REQUIRE(coeff_stride >= (with_dist ? full_area : area));
  if (n <= 0) return R1_OK;
  R1_REQUIRE(coeffs && qcoeffs && eobs);
  R1_REQUIRE(!with_dist || tx_dist);
  const r1q::QParams qp = r1q::make_qparams(*p, tx_size, coeff_bytes);
  const int kind = tx_type < 10 ? 0 : ((tx_type & 1) ? 2 : 1);
  const int q_bin = p->qindex / 32;   // RDO_QUANT_DIV
  hipStream_t st = (hipStream_t)stream;
  if (with_dist)
    return coeff_bytes == 2
               ? launch_q<int16_t, true>(ctx, coeffs, coeff_stride, n, tx_size, kind, qp, qcoeffs,
                                         eobs, rcoeffs, q_bin, tx_dist, est_rate, st)
               : launch_q<int32_t, true>(ctx, coeffs, coeff_stride, n, tx_size, kind, qp, qcoeffs,
                                         eobs, rcoeffs, q_bin, tx_dist, est_rate, st);
  return coeff_bytes == 2
             ? launch_q<int16_t, false>(ctx, coeffs, coeff_stride, n, tx_size, kind, qp, qcoeffs,
                                        eobs, rcoeffs, q_bin, nullptr, nullptr, st)
             : launch_q<int32_t, false>(ctx, coeffs, coeff_stride, n, tx_size, kind, qp, qcoeffs,
                                        eobs, rcoeffs, q_bin, nullptr, nullptr, st);
}
}  // namespace

extern "C" int r1_quantize_batch(r1_ctx *ctx, const void *coeffs, int coeff_stride, int n,
                                 int tx_size, int tx_type, const R1QuantParams *p,
                                 int coeff_bytes, void *qcoeffs, uint16_t *eobs,
                                 void *rcoeffs, void *stream) {
  return quantize_common(ctx, coeffs, coeff_stride, n, tx_size, tx_type, p, coeff_bytes, qcoeffs,
                         eobs, rcoeffs, nullptr, nullptr, false, stream);
}

extern "C" int r1_quantize_rdo_batch(r1_ctx *ctx, const void *coeffs, int coeff_stride, int n,
                                     int tx_size, int tx_type, const R1QuantParams *p,
                                     int coeff_bytes, void *qcoeffs, uint16_t *eobs,
                                     void *rcoeffs, uint64_t *tx_dist, uint64_t *est_rate,
                                     void *stream) {
  return quantize_common(ctx, coeffs, coeff_stride, n, tx_size, tx_type, p, coeff_bytes, qcoeffs,
                         eobs, rcoeffs, tx_dist, est_rate, true, stream);
}

extern "C" int r1_dequantize_batch(r1_ctx *ctx, const void *qcoeffs, int n, int tx_size,
                                   const R1QuantParams *p, int coeff_bytes, void *rcoeffs,
                                   void *stream) {
  R1_REQUIRE(ctx && p);
  R1_REQUIRE(tx_size >= 0 && tx_size < 19);
  R1_REQUIRE(p->bit_depth == 8 || p->bit_depth == 10 || p->bit_depth == 12);
  R1_REQUIRE(coeff_bytes == 2 || coeff_bytes == 4);
  if (n <= 0) return R1_OK;
  R1_REQUIRE(qcoeffs && rcoeffs);
  const int area = coded_dim(r1tx::kTxWLog2[tx_size]) * coded_dim(r1tx::kTxHLog2[tx_size]);
  const uint32_t dcq = r1q::dc_q(*p), acq = r1q::ac_q(*p);
  const long long total = (long long)n * area;
  const unsigned grid = (unsigned)((total + 255) / 256);
  hipStream_t st = (hipStream_t)stream;
  if (coeff_bytes == 2)
    hipLaunchKernelGGL((k_dequantize<int16_t>), dim3(grid), dim3(256), 0, st,
                       (const int16_t *)qcoeffs, (int16_t *)rcoeffs, total, area, dcq, acq,
                       log_tx_scale(tx_size));
  else
    hipLaunchKernelGGL((k_dequantize<int32_t>), dim3(grid), dim3(256), 0, st,
                       (const int32_t *)qcoeffs, (int32_t *)rcoeffs, total, area, dcq, acq,
                       log_tx_scale(tx_size));
  R1_HIP_CHECK(hipGetLastError());
  return R1_OK;
}
