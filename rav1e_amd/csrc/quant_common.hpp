// quant_common.hpp -- the quantizer as a device function shared by
// quantize.hip (coefficients from HBM) and rdo_cand.hip (coefficients straight
// from the fused forward transform, never leaving the CU).
//
// Restates (reference file:line):
//   QuantizationContext::update / quantize   src/quantize/mod.rs:219-355
//   dequantize                               src/quantize/mod.rs:363-384
//   divu_gen / divu_pair                     src/quantize/mod.rs:129-157
//   transform-domain distortion              src/encoder.rs:1616-1640
//   estimate_rate, RDO_RATE_TABLE            src/rdo.rs:127-139, src/rdo_tables.rs
#pragma once
#include "tx_common.hpp"

namespace r1q {

#include "quant_tables.inc"
#define R1_TABLE_QUAL __constant__
#include "rate_table.inc"
#undef R1_TABLE_QUAL

struct QParams {
  uint32_t dc_q, ac_q;
  uint32_t dc_a, dc_b, dc_s;   // divu_gen(dc_q)
  uint32_t ac_a, ac_b, ac_s;   // divu_gen(ac_q)
  uint32_t dc_offset, ac_offset0, ac_offset1;
  int32_t deadzone;            // already cast to the coefficient type
  int32_t lts;                 // log_tx_scale
};

// ---- host side -----------------------------------------------------------
inline void divu_gen(uint32_t d, uint32_t *a, uint32_t *b, uint32_t *s) {
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
inline int coded_dim(int log2) { return log2 > 5 ? 32 : 1 << log2; }
inline int bd_class(int bd) { int b = (bd ^ 8) >> 1; return b < 2 ? b : 2; }
inline int clampq(int q) { return q < 0 ? 0 : (q > 255 ? 255 : q); }
inline int log_tx_scale(int tx_size) {
  const int area = 1 << (r1tx::kTxWLog2[tx_size] + r1tx::kTxHLog2[tx_size]);
  return (area > 256) + (area > 1024);
}
inline uint32_t dc_q(const R1QuantParams &p) {
  return kR1DcQLookup[bd_class(p.bit_depth)][clampq(p.qindex + p.dc_delta_q)];
}
inline uint32_t ac_q(const R1QuantParams &p) {
  return kR1AcQLookup[bd_class(p.bit_depth)][clampq(p.qindex + p.ac_delta_q)];
}
// QuantizationContext::update (mod.rs:219-265) for one (tx size, coefficient type)
inline QParams make_qparams(const R1QuantParams &p, int tx_size, int coeff_bytes) {
  QParams qp;
  qp.dc_q = dc_q(p);
  qp.ac_q = ac_q(p);
  divu_gen(qp.dc_q, &qp.dc_a, &qp.dc_b, &qp.dc_s);
  divu_gen(qp.ac_q, &qp.ac_a, &qp.ac_b, &qp.ac_s);
  const bool intra = p.is_intra != 0;
  qp.dc_offset = qp.dc_q * (intra ? 109 : 108) / 256;
  qp.ac_offset0 = qp.ac_q * (intra ? 98 : 97) / 256;
  qp.ac_offset1 = qp.ac_q * (intra ? 109 : 108) / 256;
  const uint32_t off_eob = qp.ac_q * (intra ? 88 : 44) / 256;
  qp.lts = log_tx_scale(tx_size);
  const uint32_t dz = (qp.ac_q - off_eob + (1u << qp.lts) - 1) >> qp.lts;
  qp.deadzone = coeff_bytes == 2 ? (int32_t)(int16_t)dz : (int32_t)dz;
  return qp;
}

// ---- device side ---------------------------------------------------------
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

// G = 1 << GL lanes of a wave (the group's lane 0 is wave lane `g0`) own one
// block whose coded coefficients sit in LDS `mine` as int32 values of CT
// (transposed layout).  Lane l visits scan positions l, l+G, ..: gather,
// eob-1 = max scan index with |c| >= deadzone, DC by lane 0, AC through the
// 2-state prefix scan (see quantize.hip header).  On return `mine` holds the
// quantized coefficients, eob the reference's return value and -- DIST -- dist
// the transform-domain distortion (coded part + `tail` = the caller's partial
// sum of squares beyond the coded area), already rounded and shifted.
template <typename CT, int GL, int NPL, bool DIST>
__device__ __forceinline__ void quantize_group(int32_t *mine, int g0, int l, bool live,
                                               const uint16_t *__restrict__ scan,
                                               const QParams &qp, unsigned long long tail,
                                               int &eob_out, unsigned long long &dist_out) {
  constexpr int G = 1 << GL;
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
  // DC (lane 0 of the group holds scan position 0 = coefficient 0)
  int32_t q0 = 0;
  {
    const int32_t c = (int32_t)((uint32_t)cv[0] << qp.lts);
    const uint32_t a = c < 0 ? 0u - (uint32_t)c : (uint32_t)c;
    const uint32_t v = divu_pair(a + qp.dc_offset, qp.dc_a, qp.dc_b, qp.dc_s);
    q0 = (int32_t)(CT)(c < 0 ? -(int32_t)v : (int32_t)v);
  }
  q0 = __shfl(q0, g0, 64);
  const int eob = eob_m1 > 0 ? eob_m1 + 1 : (q0 != 0);
  int carry = 1;   // level_mode starts at 1
  unsigned long long dist = tail;
#pragma unroll
  for (int k = 0; k < NPL; k++) {
    const int i = k * G + l;
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
    const uint32_t last = __shfl(F, g0 + G - 1, 64);
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
  eob_out = eob;
  if constexpr (DIST) {
#pragma unroll
    for (int m = 1; m < G; m <<= 1) {
      const uint32_t lo = __shfl_xor((uint32_t)dist, m, 64);
      const uint32_t hi = __shfl_xor((uint32_t)(dist >> 32), m, 64);
      dist += ((unsigned long long)hi << 32) | lo;
    }
    const int bits = 2 * (3 - qp.lts);
    dist_out = (dist + (1ull << (bits - 1))) >> bits;
  }
}

}  // namespace r1q
