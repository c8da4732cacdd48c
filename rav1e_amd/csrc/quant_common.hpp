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
  // i16 coefficients (8-bit pixels): |c << lts| <= 2^17, so floor(a / ac_q) is
  // (a * ac_m24) >> ac_s24 with 24-bit multiplies (m = floor(2^s / q) + 1,
  // s = 18 + ceil(log2 q): exact for every a < 2^18; swept over every table q in
  // tests/test_oracle_quant.py::test_narrow_division_magic_is_exact)
  uint32_t ac_m24, ac_s24;
  // i32 coefficients that come from PIXEL residuals (the fused kernels): |c << lts| <= 2^17 / 2^19 / 2^21 at 8 / 10 /
  // 12 bits (L1 norms of the transform networks, tools/tx_range.py::shifted_coefficient_bound, tests/test_tx_range.py),
  // so a + offset < 2^22 and floor(a / ac_q) = (a * ac_m22) >> ac_s22 with the same 24-bit multipliers
  // (m = floor(2^s / q) + 1 < 2^23, s = 22 + ceil(log2 q) -- up to 37: past 32 the quotient is the high word shifted)
  uint32_t ac_m22, ac_s22;
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
// floor(a / q) = (a * m) >> s for every a < 2^nbits: m = floor(2^s / q) + 1, s = nbits + ceil(log2 q)
inline void narrow_magic(uint32_t q, uint32_t *m, uint32_t *s, unsigned nbits = 18) {
  unsigned L = 0;
  while ((1u << L) < q) L++;
  *s = nbits + L;
  *m = (uint32_t)((1ull << *s) / q) + 1;
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
  narrow_magic(qp.ac_q, &qp.ac_m24, &qp.ac_s24);
  narrow_magic(qp.ac_q, &qp.ac_m22, &qp.ac_s22, 22);
  return qp;
}

// ---- device side ---------------------------------------------------------
__device__ __forceinline__ uint32_t divu_pair(uint32_t x, uint32_t a, uint32_t b,
                                              uint32_t s) {
  return (uint32_t)((((uint64_t)a * x + b) >> 32) >> s);
}

// bits 32..47 of the 48-bit product of two 24-bit operands (m: wave-uniform)
__device__ __forceinline__ uint32_t umulhi24(uint32_t m, uint32_t a) {
  uint32_t r;
  asm("v_mul_hi_u32_u24 %0, %1, %2" : "=v"(r) : "s"(m), "v"(a));
  return r;
}

// low 32 bits of the product of two sign-extended 24-bit operands, as the
// hardware defines it (__mul24 promises the compiler "no signed overflow",
// which turns the sign extension of a wrapped square into a zero extension)
__device__ __forceinline__ int32_t mul24_wrap(int32_t a, int32_t b) {
  int32_t r;
  asm("v_mul_i32_i24 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// estimate_rate (src/rdo.rs:127-139): piecewise-linear lookup in RDO_RATE_TABLE
__device__ __forceinline__ unsigned long long estimate_rate(int q_bin, int tx_size,
                                                            unsigned long long fd) {
  // down = min(fd / 2000, 48): clamp first, then a 32-bit reciprocal
  // (ceil(2^32 / 2000) = 2147484 is exact below 6.1e6)
  const uint32_t f = fd > 97999ull ? 97999u : (uint32_t)fd;
  const uint32_t down = __umulhi(f, 2147484u);
  const long long x0 = (long long)(down * 2000u);
  const int32_t y0 = (int32_t)kR1RdoRateTable[q_bin][tx_size][down],
                y1 = (int32_t)kR1RdoRateTable[q_bin][tx_size][down + 1];
  const int32_t slope = ((y1 - y0) * 256) / 2000;   // |y| <= 99999: 32 bits are enough
  const long long v = y0 + ((((long long)fd - x0) * slope) >> 8);
  return v < 0 ? 0ull : (unsigned long long)v;
}

// Scan positions l*NPL .. l*NPL+NPL-1 of one table, as packed u16 pairs
// (vector loads: the run of a lane is contiguous in the table).
template <int NPL>
struct ScanRun {
  uint32_t w[NPL >= 2 ? NPL / 2 : 1];
  __device__ __forceinline__ void load(const uint16_t *__restrict__ scan, int l) {
    const uint16_t *p = scan + l * NPL;
    if constexpr (NPL >= 8) {
#pragma unroll
      for (int j = 0; j < NPL / 8; j++) {
        const uint4 v = ((const uint4 *)p)[j];
        w[4 * j] = v.x; w[4 * j + 1] = v.y; w[4 * j + 2] = v.z; w[4 * j + 3] = v.w;
      }
    } else if constexpr (NPL == 4) {
      const uint2 v = *(const uint2 *)p;
      w[0] = v.x; w[1] = v.y;
    } else if constexpr (NPL == 2) {
      w[0] = *(const uint32_t *)p;
    } else {
      w[0] = *p;
    }
  }
  __device__ __forceinline__ uint32_t at(int k) const {
    return (k & 1) ? w[k >> 1] >> 16 : w[k >> 1] & 0xFFFFu;
  }
};

// G = 1 << GL lanes of a wave (the group's lane 0 is wave lane `g0`) own one
// block whose coded coefficients sit in LDS `mine` as int32 values of CT
// (transposed layout).  Lane l owns the contiguous run of scan positions
// [l*NPL, (l+1)*NPL): gather, eob-1 = max scan index with |c| >= deadzone, DC
// by lane 0, then the AC loop of mod.rs:311-336.  Its only serial dependence
// is level_mode (one bit), so every element is a function {0,1} -> {0,1}
// (reset / keep / set, see pass 1).  A lane composes its run sequentially, one
// log2(G)-step scan composes the lanes, and a replay pass picks the level per
// element.  On return `mine` holds the quantized coefficients, eob the
// reference's return value and -- DIST -- dist the transform-domain distortion
// (coded part + `tail` = the caller's partial sum of squares beyond the coded
// area), already rounded and shifted.
// LTS: log_tx_scale when the caller knows it at compile time (the fused kernels:
// it follows from the block size), -1 = qp.lts.
// MID (i32 coefficients only): the caller guarantees |c << lts| + ac_offset < 2^22 (coefficients of a pixel residual,
// see QParams::ac_m22): the AC division and the dequantizer run on the full-rate 24-bit multipliers instead of the
// quarter-rate 64-bit multiply-add / v_mul_lo_u32 of the general path (quantize.hip: coefficients from HBM, any value).
template <typename CT, int GL, int NPL, bool DIST, int LTS = -1, bool MID = false>
__device__ __forceinline__ void quantize_group(int32_t *mine, int g0, int l, bool live,
                                               const uint16_t *__restrict__ scan,
                                               const QParams &qp, unsigned long long tail,
                                               int &eob_out, unsigned long long &dist_out) {
  constexpr int G = 1 << GL;
  // i16 coefficients: every product below fits the 24-bit multipliers
  // (v_mul_u32_u24 / v_mul_hi_u32_u24: full rate; v_mul_lo_u32 / v_mad_u64_u32
  // are quarter rate).  The i16 <-> 8-bit coupling is the reference's own
  // (T::Coeff, src/util/mod.rs) and is enforced at the entry points.
  constexpr bool NARROW = sizeof(CT) == 2;
  // short runs keep pass 1's result of every element in a register (A0 << 1 | dA);
  // long ones park it in the element's own LDS slot and pass 2 reads it back
  constexpr bool KEEP = NPL <= 16;
  const int lts = LTS >= 0 ? LTS : qp.lts;
  // LDS accesses are unconditional: a dead group owns its (unused) slice of
  // the tile all the same and nothing of it reaches HBM
  (void)live;
  int32_t cv[NPL];
  ScanRun<NPL> pos;
  pos.load(scan, l);
  int eob_m1 = 0;
#pragma unroll
  for (int k = 0; k < NPL; k++) {
    cv[k] = mine[pos.at(k)];
    // T::abs() wraps at T::MIN (mod.rs:296: c.abs() on T::Coeff)
    const int32_t a = (int32_t)(CT)(cv[k] < 0 ? (CT)(0 - (uint32_t)cv[k]) : (CT)cv[k]);
    if (a >= qp.deadzone) eob_m1 = l * NPL + k;   // increasing in k: the max survives
  }
#pragma unroll
  for (int m = 1; m < G; m <<= 1) {
    const int o = __shfl_xor(eob_m1, m, 64);
    eob_m1 = o > eob_m1 ? o : eob_m1;
  }
  // DC (lane 0 of the group holds scan position 0 = coefficient 0)
  int32_t q0 = 0;
  {
    const int32_t c = (int32_t)((uint32_t)cv[0] << lts);
    const uint32_t a = c < 0 ? 0u - (uint32_t)c : (uint32_t)c;
    const uint32_t v = divu_pair(a + qp.dc_offset, qp.dc_a, qp.dc_b, qp.dc_s);
    q0 = (int32_t)(CT)(c < 0 ? -(int32_t)v : (int32_t)v);
  }
  q0 = __shfl(q0, g0, 64);
  const int eob = eob_m1 > 0 ? eob_m1 + 1 : (q0 != 0);

  // pass 1.  With level0 = a / q and rem = a % q the reference's two rounding
  // candidates are A0 = level0 + (rem + offset0 >= q), A1 = level0 + (rem +
  // offset1 >= q) (A0 <= A1 <= A0 + 1), and going through mod.rs:317-336 case
  // by case (level0 = 0, 1, >= 2 against level_mode = 0, 1) the whole AC step
  // collapses to
  //     level_mode' = (min(A0, 2) + level_mode) >> 1,   |q| = level_mode' ? A1 : A0
  // so an element is the pair (A0, dA = A1 - A0), kept as pk = A0 << 1 | dA
  // (A0 < 2^30: q >= 4), and its transition function is min(A0, 2): 0 = reset,
  // 1 = keep, 2 = set.  A run of elements composes to its LAST element that is
  // not "keep" (or to "keep"): one compare-and-select per element.  Elements at
  // or past eob are 0 (level 0; nothing after them reads level_mode), the DC
  // slot is "keep".  The sign is not carried: pass 2 takes it from cv again.
  const int lim = eob - l * NPL;        // elements k < lim of this run are below eob
  const uint32_t need0 = qp.ac_q - qp.ac_offset0, need1 = qp.ac_q - qp.ac_offset1;
  uint32_t pks[KEEP ? NPL : 1];
  uint32_t st = 2u;                     // "keep" (A0 = 1)
#pragma unroll
  for (int k = 0; k < NPL; k++) {
    const int32_t c = (int32_t)((uint32_t)cv[k] << lts);
    const uint32_t a = c < 0 ? 0u - (uint32_t)c : (uint32_t)c;
    uint32_t level0, rem;
    if constexpr (NARROW) {
      const uint32_t lo = __umul24(a, qp.ac_m24), hi = umulhi24(qp.ac_m24, a);
      level0 = __builtin_amdgcn_alignbit(hi, lo, qp.ac_s24);   // (hi:lo) >> s, s < 32
      rem = a - __umul24(level0, qp.ac_q);
    } else if constexpr (MID) {
      const uint32_t lo = __umul24(a, qp.ac_m22), hi = umulhi24(qp.ac_m22, a);
      // (hi:lo) >> s; s is wave-uniform: 24 .. 37
      level0 = qp.ac_s22 < 32 ? __builtin_amdgcn_alignbit(hi, lo, qp.ac_s22) : hi >> (qp.ac_s22 - 32);
      rem = a - __umul24(level0, qp.ac_q);
    } else {
      level0 = divu_pair(a, qp.ac_a, qp.ac_b, qp.ac_s);
      rem = a - level0 * qp.ac_q;
    }
    const uint32_t A0 = level0 + (rem >= need0 ? 1u : 0u);
    const uint32_t dA = (rem - need1 < need0 - need1) ? 1u : 0u;   // need1 <= rem < need0
    uint32_t pk = (A0 << 1) | dA;
    pk = (k < lim) ? pk : 0u;
    if (k == 0) pk = l == 0 ? 2u : pk;
    st = ((pk | 1u) == 3u) ? st : pk;   // A0 == 1: keep
    if constexpr (KEEP) pks[k] = pk;
    else mine[pos.at(k)] = (int32_t)pk;
  }
  // the run's function as (level_mode after it when entered with 0) | (... with 1) << 1
  const uint32_t ts = (st >> 1) < 2u ? (st >> 1) : 2u;
  uint32_t L = (ts >> 1) | (((ts + 1u) >> 1) << 1);
  // the lanes' functions, composed in lane order (inclusive), then the
  // level_mode entering this lane's run (level_mode starts at 1)
#pragma unroll
  for (int d = 1; d < G; d <<= 1) {
    const uint32_t p = __shfl_up(L, d, G);
    if (l >= d) L = ((L >> (p & 1)) & 1) | (((L >> ((p >> 1) & 1)) & 1) << 1);
  }
  uint32_t E = __shfl_up(L, 1, G);
  if (l == 0) E = 2u;
  uint32_t mode = (E >> 1) & 1;

  // pass 2: replay
  unsigned long long dist = tail;
  const int32_t off = (1 << lts) - 1;
#pragma unroll
  for (int k = 0; k < NPL; k++) {
    const uint32_t pix = pos.at(k);
    uint32_t pk;
    if constexpr (KEEP) pk = pks[k];
    else pk = (uint32_t)mine[pix];
    const uint32_t A0 = pk >> 1;
    mode = ((A0 < 2u ? A0 : 2u) + mode) >> 1;
    const uint32_t mag = A0 + (pk & mode);        // mode is 0 / 1: picks dA
    // copysign(abs_qcoeff, coeff): the sign of the SHIFTED coefficient (mod.rs:318,338); an i16
    // coefficient cannot lose its sign to a shift by <= 2
    const int32_t sg = NARROW ? cv[k] >> 31 : (int32_t)((uint32_t)cv[k] << lts) >> 31;
    int32_t q = (int32_t)((mag ^ (uint32_t)sg) - (uint32_t)sg);
    if (k == 0 && l == 0) q = q0;
    mine[pix] = q;
    if constexpr (DIST) {
      const int32_t qt = (int32_t)(CT)q;
      // scan position 0 is coefficient 0 in every scan order
      const uint32_t quant = (k == 0 && l == 0) ? qp.dc_q : qp.ac_q;
      int32_t r, dd, sq;
      if constexpr (NARROW) {
        r = (int32_t)(CT)((__mul24(qt, (int32_t)quant) + ((qt >> 31) & off)) >> lts);
        dd = cv[k] - r;                 // both i16: 17 bits
        sq = mul24_wrap(dd, dd);        // low 32 bits = the wrapping i32 product
      } else if constexpr (MID) {
        // |q| <= 2^22 / 4, quant < 2^15, |dd| < 2^22: the low 32 bits of the 24-bit products ARE the wrapping i32 products
        r = (int32_t)((uint32_t)mul24_wrap(qt, (int32_t)quant) + (uint32_t)((qt >> 31) & off)) >> lts;
        dd = (int32_t)((uint32_t)cv[k] - (uint32_t)r);
        sq = mul24_wrap(dd, dd);
      } else {
        r = (int32_t)((uint32_t)qt * quant + (uint32_t)((qt >> 31) & off)) >> lts;
        dd = (int32_t)((uint32_t)cv[k] - (uint32_t)r);
        sq = (int32_t)((uint32_t)dd * (uint32_t)dd);
      }
      // `(c * c) as u64`: i32 product (wrapping), sign-extended
      dist += (unsigned long long)(long long)sq;
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
    const int bits = 2 * (3 - lts);
    dist_out = (dist + (1ull << (bits - 1))) >> bits;
  }
}

}  // namespace r1q
