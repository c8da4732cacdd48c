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
// wave-level prefix scan over 2-bit function codes.
//
// Mapping (wave = 64): G = min(64, coded area) lanes own one block, 64/G
// blocks per wave, 4 waves per workgroup.
//  1 the block's coded coefficients are staged into LDS with coalesced loads;
//  2 lane l visits scan positions l, l+G, ..: gathers lds[scan[i]] into
//    registers, eob-1 = max scan index with |c| >= deadzone (group max);
//  3 DC by the group's lane 0; per chunk of G scan positions both candidate
//    outcomes (level_mode 0 / 1) are evaluated, the 2-bit transition codes are
//    composed by a Hillis-Steele scan inside the group, the carry-in state
//    picks the result, which overwrites lds[scan[i]] in place;
//  4 the group writes qcoeffs (and rcoeffs = dequantize(qcoeffs)) back with
//    coalesced stores; positions >= eob are zero (the reference relies on a
//    pre-zeroed buffer, encoder.rs:1518-1521).
// Exact u32 division by the quantizer uses the reference's own reciprocal
// form divu_pair (mod.rs:129-157), computed on the host per launch.
#include <stdlib.h>

#include <vector>

#include "tx_common.hpp"
#include "quant_tables.inc"
#define R1_TABLE_QUAL __constant__
#include "rate_table.inc"

namespace {

struct QParams {
  uint32_t dc_q, ac_q;
  uint32_t dc_a, dc_b, dc_s;   // divu_gen(dc_q)
  uint32_t ac_a, ac_b, ac_s;   // divu_gen(ac_q)
  uint32_t dc_offset, ac_offset0, ac_offset1;
  int32_t deadzone;            // already cast to the coefficient type
  int32_t lts;                 // log_tx_scale
};

__device__ __forceinline__ uint32_t divu_pair(uint32_t x, uint32_t a, uint32_t b,
                                              uint32_t s) {
  return (uint32_t)((((uint64_t)a * x + b) >> 32) >> s);
}

// estimate_rate (src/rdo.rs:127-139): piecewise-linear lookup in RDO_RATE_TABLE
__device__ __forceinline__ unsigned long long estimate_rate(int q_bin, int tx_size,
                                                            unsigned long long fd) {
  unsigned long long down = fd / 2000;
  down = down > 48 ? 48 : down;
  const unsigned long long up = down + 1;
  const long long x0 = (long long)(down * 2000);
  const long long y0 = kR1RdoRateTable[q_bin][tx_size][down], y1 = kR1RdoRateTable[q_bin][tx_size][up];
  const long long slope = ((y1 - y0) * 256) / 2000;
  const long long v = y0 + ((((long long)fd - x0) * slope) >> 8);
  return v < 0 ? 0ull : (unsigned long long)v;
}

// DIST: also the transform-domain distortion of encode_tx_block
// (src/encoder.rs:1616-1640: sum (coeff - rcoeff)^2 over the coded area + sum
// coeff^2 beyond it, rounding shift by 2 * (3 - log_tx_scale)) and the table
// rate estimate for it.
template <typename CT, int GL, int NPL, bool DIST>
__global__ __launch_bounds__(256) void k_quantize(
    const CT *__restrict__ coeffs, int coeff_stride, int n, int area,
    const uint16_t *__restrict__ scan, QParams qp, CT *__restrict__ qcoeffs,
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
  // 1: stage (coalesced)
  if (live) {
    const CT *src = coeffs + blk * coeff_stride;
#pragma unroll
    for (int k = 0; k < NPL; k++) mine[k * G + l] = (int32_t)src[k * G + l];
  }
  __builtin_amdgcn_wave_barrier();
  // 2: gather in scan order, eob search
  int32_t cv[NPL];
  uint16_t pos[NPL];
  int eob_m1 = 0;
#pragma unroll
  for (int k = 0; k < NPL; k++) {
    pos[k] = scan[k * G + l];
    cv[k] = live ? mine[pos[k]] : 0;
    // T::abs() wraps at T::MIN (mod.rs:296: c.abs() on T::Coeff)
    const int32_t a = (int32_t)(CT)(cv[k] < 0 ? (CT)(0 - (uint32_t)cv[k]) : (CT)cv[k]);
    if (a >= qp.deadzone) eob_m1 = k * G + l;   // increasing in k: the max survives
  }
#pragma unroll
  for (int m = 1; m < G; m <<= 1) {
    const int o = __shfl_xor(eob_m1, m, 64);
    eob_m1 = o > eob_m1 ? o : eob_m1;
  }
  // 3: DC (lane 0 of the group holds scan position 0 = coefficient 0)
  int32_t q0 = 0;
  {
    const int32_t c = (int32_t)((uint32_t)cv[0] << qp.lts);
    const uint32_t a = c < 0 ? 0u - (uint32_t)c : (uint32_t)c;
    const uint32_t v = divu_pair(a + qp.dc_offset, qp.dc_a, qp.dc_b, qp.dc_s);
    q0 = (int32_t)(CT)(c < 0 ? -(int32_t)v : (int32_t)v);
  }
  q0 = __shfl(q0, g << GL, 64);
  const int eob = eob_m1 > 0 ? eob_m1 + 1 : (q0 != 0);
  int carry = 1;   // level_mode starts at 1
  unsigned long long dist = 0;
#pragma unroll
  for (int k = 0; k < NPL; k++) {
    const int i = k * G + l;
    // wave-uniform early out is not possible per group; predicate instead
    const bool act = i >= 1 && i < eob;
    const int32_t c = (int32_t)((uint32_t)cv[k] << qp.lts);
    const uint32_t a = c < 0 ? 0u - (uint32_t)c : (uint32_t)c;
    const uint32_t level0 = divu_pair(a, qp.ac_a, qp.ac_b, qp.ac_s);
    const uint32_t thr = (level0 + 1) * qp.ac_q;
    const uint32_t up0 = a + qp.ac_offset0 >= thr, up1 = a + qp.ac_offset1 >= thr;
    // level_mode 0: offset1 iff level0 > 1; level_mode 1: offset1 iff level0 > 0
    const uint32_t aq0 = level0 + (level0 > 1 ? up1 : up0);
    const uint32_t aq1 = level0 + (level0 > 0 ? up1 : up0);
    // transitions (mod.rs:331-335): 0 -> (aq > 1), 1 -> (aq != 0)
    uint32_t F = act ? ((aq0 > 1 ? 1u : 0u) | (aq1 != 0 ? 2u : 0u)) : 2u;
#pragma unroll
    for (int d = 1; d < G; d <<= 1) {
      const uint32_t p = __shfl_up(F, d, G);
      if (l >= d) F = ((F >> (p & 1)) & 1) | (((F >> ((p >> 1) & 1)) & 1) << 1);
    }
    uint32_t E = __shfl_up(F, 1, G);
    if (l == 0) E = 2u;
    const int mode = (E >> carry) & 1;
    const uint32_t last = __shfl(F, (g << GL) + G - 1, 64);
    carry = (last >> carry) & 1;
    const uint32_t aq = mode ? aq1 : aq0;
    int32_t q = act ? (c < 0 ? -(int32_t)aq : (int32_t)aq) : 0;
    if (i == 0) q = q0;
    if (live) mine[pos[k]] = q;
    if constexpr (DIST) {
      const int32_t qt = (int32_t)(CT)q;
      const uint32_t quant = pos[k] == 0 ? qp.dc_q : qp.ac_q;
      const int32_t off = (1 << qp.lts) - 1;
      const int32_t r = (int32_t)(CT)((int32_t)((uint32_t)qt * quant + (uint32_t)((qt >> 31) & off)) >> qp.lts);
      const int32_t dd = (int32_t)((uint32_t)cv[k] - (uint32_t)r);
      // `(c * c) as u64`: i32 product (wrapping), sign-extended
      if (live) dist += (unsigned long long)(long long)(int32_t)((uint32_t)dd * (uint32_t)dd);
    }
  }
  if constexpr (DIST) {
    if (live) {   // coefficients beyond the coded area (64-point sizes): rcoeff = 0
      const CT *src = coeffs + blk * coeff_stride;
      for (int i = area + l; i < full_area; i += G) {
        const int32_t c = (int32_t)src[i];
        dist += (unsigned long long)(long long)(int32_t)((uint32_t)c * (uint32_t)c);
      }
    }
#pragma unroll
    for (int m = 1; m < G; m <<= 1) {
      const uint32_t lo = __shfl_xor((uint32_t)dist, m, 64);
      const uint32_t hi = __shfl_xor((uint32_t)(dist >> 32), m, 64);
      dist += ((unsigned long long)hi << 32) | lo;
    }
    if (live && l == 0) {
      const int bits = 2 * (3 - qp.lts);
      const unsigned long long d = (dist + (1ull << (bits - 1))) >> bits;
      tx_dist[blk] = d;
      if (est_rate) est_rate[blk] = estimate_rate(q_bin, tx_size, d);
    }
  }
  __builtin_amdgcn_wave_barrier();
  // 4: write back (coalesced), dequantize on the way (mod.rs:372-383)
  if (live) {
    CT *qd = qcoeffs + blk * area;
    CT *rd = rcoeffs ? rcoeffs + blk * area : nullptr;
    const int32_t off = (1 << qp.lts) - 1;
#pragma unroll
    for (int k = 0; k < NPL; k++) {
      const int idx = k * G + l;
      const int32_t q = (int32_t)(CT)mine[idx];
      qd[idx] = (CT)q;
      if (rd) {
        const uint32_t quant = idx == 0 ? qp.dc_q : qp.ac_q;
        rd[idx] = (CT)((int32_t)((uint32_t)q * quant + (uint32_t)((q >> 31) & off)) >> qp.lts);
      }
    }
    if (l == 0) eobs[blk] = (uint16_t)eob;
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

void divu_gen(uint32_t d, uint32_t *a, uint32_t *b, uint32_t *s) {
  const unsigned m = 31 - (unsigned)__builtin_clz(d);
  if ((d & (d - 1)) == 0) {
    *a = 0xFFFFFFFFu; *b = 0xFFFFFFFFu;
  } else {
    const uint64_t t = (1ull << (m + 32)) / d;
    const uint64_t r = (t * d + d) & 0xFFFFFFFFull;
    if (r <= (1ull << m)) { *a = (uint32_t)t + 1; *b = 0; }
    else { *a = (uint32_t)t; *b = (uint32_t)t; }
  }
  *s = m;
}

int coded_dim(int log2) { return log2 > 5 ? 32 : 1 << log2; }
int bd_class(int bd) { int b = (bd ^ 8) >> 1; return b < 2 ? b : 2; }
int clampq(int q) { return q < 0 ? 0 : (q > 255 ? 255 : q); }
int log_tx_scale(int tx_size) {
  const int area = 1 << (r1tx::kTxWLog2[tx_size] + r1tx::kTxHLog2[tx_size]);
  return (area > 256) + (area > 1024);
}

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
             int kind, const QParams &qp, void *q, uint16_t *eobs, void *r,
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
    case 16: R1_Q_LAUNCH(4, 1); break;
    case 32: R1_Q_LAUNCH(5, 1); break;
    case 64: R1_Q_LAUNCH(6, 1); break;
    case 128: R1_Q_LAUNCH(6, 2); break;
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
  QParams qp;
  const int bc = bd_class(p->bit_depth);
  qp.dc_q = kR1DcQLookup[bc][clampq(p->qindex + p->dc_delta_q)];
  qp.ac_q = kR1AcQLookup[bc][clampq(p->qindex + p->ac_delta_q)];
  divu_gen(qp.dc_q, &qp.dc_a, &qp.dc_b, &qp.dc_s);
  divu_gen(qp.ac_q, &qp.ac_a, &qp.ac_b, &qp.ac_s);
  const bool intra = p->is_intra != 0;
  qp.dc_offset = qp.dc_q * (intra ? 109 : 108) / 256;
  qp.ac_offset0 = qp.ac_q * (intra ? 98 : 97) / 256;
  qp.ac_offset1 = qp.ac_q * (intra ? 109 : 108) / 256;
  const uint32_t off_eob = qp.ac_q * (intra ? 88 : 44) / 256;
  qp.lts = log_tx_scale(tx_size);
  const uint32_t dz = (qp.ac_q - off_eob + (1u << qp.lts) - 1) >> qp.lts;
  qp.deadzone = coeff_bytes == 2 ? (int32_t)(int16_t)dz : (int32_t)dz;
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
  const int bc = bd_class(p->bit_depth);
  const uint32_t dcq = kR1DcQLookup[bc][clampq(p->qindex + p->dc_delta_q)];
  const uint32_t acq = kR1AcQLookup[bc][clampq(p->qindex + p->ac_delta_q)];
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
