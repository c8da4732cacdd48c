// mc_common.hpp -- AV1 sub-pel interpolation filters and the put/prep
// arithmetic shared by mc.hip and rdo_cand.hip.
//
// Restates (reference file:line):
//   SUBPEL_FILTERS     src/mc.rs:110-219 (AV1 spec tables: regular, smooth,
//                      sharp, bilinear, 4-tap regular, 4-tap smooth)
//   get_filter         src/mc.rs:238-247
//   put_8tap           src/mc.rs:250-353     prep_8tap  src/mc.rs:360-451
//   mc_avg             src/mc.rs:454-479     PREP_BIAS  src/mc.rs:355-357
#pragma once
#include "common.hpp"

namespace r1mc {

// taps as int8 pairs would not hold 128 (the frac-0 row), which the kernels
// never multiply with (frac 0 takes the copy / 1-D paths), so int16 it is.
__constant__ int16_t kSubpel[6][16][8] = {
    {{0, 0, 0, 128, 0, 0, 0, 0}, {0, 2, -6, 126, 8, -2, 0, 0},
     {0, 2, -10, 122, 18, -4, 0, 0}, {0, 2, -12, 116, 28, -8, 2, 0},
     {0, 2, -14, 110, 38, -10, 2, 0}, {0, 2, -14, 102, 48, -12, 2, 0},
     {0, 2, -16, 94, 58, -12, 2, 0}, {0, 2, -14, 84, 66, -12, 2, 0},
     {0, 2, -14, 76, 76, -14, 2, 0}, {0, 2, -12, 66, 84, -14, 2, 0},
     {0, 2, -12, 58, 94, -16, 2, 0}, {0, 2, -12, 48, 102, -14, 2, 0},
     {0, 2, -10, 38, 110, -14, 2, 0}, {0, 2, -8, 28, 116, -12, 2, 0},
     {0, 0, -4, 18, 122, -10, 2, 0}, {0, 0, -2, 8, 126, -6, 2, 0}},
    {{0, 0, 0, 128, 0, 0, 0, 0}, {0, 2, 28, 62, 34, 2, 0, 0},
     {0, 0, 26, 62, 36, 4, 0, 0}, {0, 0, 22, 62, 40, 4, 0, 0},
     {0, 0, 20, 60, 42, 6, 0, 0}, {0, 0, 18, 58, 44, 8, 0, 0},
     {0, 0, 16, 56, 46, 10, 0, 0}, {0, -2, 16, 54, 48, 12, 0, 0},
     {0, -2, 14, 52, 52, 14, -2, 0}, {0, 0, 12, 48, 54, 16, -2, 0},
     {0, 0, 10, 46, 56, 16, 0, 0}, {0, 0, 8, 44, 58, 18, 0, 0},
     {0, 0, 6, 42, 60, 20, 0, 0}, {0, 0, 4, 40, 62, 22, 0, 0},
     {0, 0, 4, 36, 62, 26, 0, 0}, {0, 0, 2, 34, 62, 28, 2, 0}},
    {{0, 0, 0, 128, 0, 0, 0, 0}, {-2, 2, -6, 126, 8, -2, 2, 0},
     {-2, 6, -12, 124, 16, -6, 4, -2}, {-2, 8, -18, 120, 26, -10, 6, -2},
     {-4, 10, -22, 116, 38, -14, 6, -2}, {-4, 10, -22, 108, 48, -18, 8, -2},
     {-4, 10, -24, 100, 60, -20, 8, -2}, {-4, 10, -24, 90, 70, -22, 10, -2},
     {-4, 12, -24, 80, 80, -24, 12, -4}, {-2, 10, -22, 70, 90, -24, 10, -4},
     {-2, 8, -20, 60, 100, -24, 10, -4}, {-2, 8, -18, 48, 108, -22, 10, -4},
     {-2, 6, -14, 38, 116, -22, 10, -4}, {-2, 6, -10, 26, 120, -18, 8, -2},
     {-2, 4, -6, 16, 124, -12, 6, -2}, {0, 2, -2, 8, 126, -6, 2, -2}},
    {{0, 0, 0, 128, 0, 0, 0, 0}, {0, 0, 0, 120, 8, 0, 0, 0},
     {0, 0, 0, 112, 16, 0, 0, 0}, {0, 0, 0, 104, 24, 0, 0, 0},
     {0, 0, 0, 96, 32, 0, 0, 0}, {0, 0, 0, 88, 40, 0, 0, 0},
     {0, 0, 0, 80, 48, 0, 0, 0}, {0, 0, 0, 72, 56, 0, 0, 0},
     {0, 0, 0, 64, 64, 0, 0, 0}, {0, 0, 0, 56, 72, 0, 0, 0},
     {0, 0, 0, 48, 80, 0, 0, 0}, {0, 0, 0, 40, 88, 0, 0, 0},
     {0, 0, 0, 32, 96, 0, 0, 0}, {0, 0, 0, 24, 104, 0, 0, 0},
     {0, 0, 0, 16, 112, 0, 0, 0}, {0, 0, 0, 8, 120, 0, 0, 0}},
    {{0, 0, 0, 128, 0, 0, 0, 0}, {0, 0, -4, 126, 8, -2, 0, 0},
     {0, 0, -8, 122, 18, -4, 0, 0}, {0, 0, -10, 116, 28, -6, 0, 0},
     {0, 0, -12, 110, 38, -8, 0, 0}, {0, 0, -12, 102, 48, -10, 0, 0},
     {0, 0, -14, 94, 58, -10, 0, 0}, {0, 0, -12, 84, 66, -10, 0, 0},
     {0, 0, -12, 76, 76, -12, 0, 0}, {0, 0, -10, 66, 84, -12, 0, 0},
     {0, 0, -10, 58, 94, -14, 0, 0}, {0, 0, -10, 48, 102, -12, 0, 0},
     {0, 0, -8, 38, 110, -12, 0, 0}, {0, 0, -6, 28, 116, -10, 0, 0},
     {0, 0, -4, 18, 122, -8, 0, 0}, {0, 0, -2, 8, 126, -4, 0, 0}},
    {{0, 0, 0, 128, 0, 0, 0, 0}, {0, 0, 30, 62, 34, 2, 0, 0},
     {0, 0, 26, 62, 36, 4, 0, 0}, {0, 0, 22, 62, 40, 4, 0, 0},
     {0, 0, 20, 60, 42, 6, 0, 0}, {0, 0, 18, 58, 44, 8, 0, 0},
     {0, 0, 16, 56, 46, 10, 0, 0}, {0, 0, 14, 54, 48, 12, 0, 0},
     {0, 0, 12, 52, 52, 12, 0, 0}, {0, 0, 12, 48, 54, 14, 0, 0},
     {0, 0, 10, 46, 56, 16, 0, 0}, {0, 0, 8, 44, 58, 18, 0, 0},
     {0, 0, 6, 42, 60, 20, 0, 0}, {0, 0, 4, 40, 62, 22, 0, 0},
     {0, 0, 4, 36, 62, 26, 0, 0}, {0, 0, 2, 34, 62, 30, 0, 0}}};

// mc.rs:238-247: blocks whose filtered dimension is <= 4 use the 4-tap sets.
__device__ __forceinline__ const int16_t *get_filter(int mode, int frac,
                                                     int length) {
  const int idx = (mode == R1_FILTER_BILINEAR || length > 4)
                      ? mode
                      : (mode < 1 ? mode : 1) + 4;
  return kSubpel[idx][frac];
}

__device__ __forceinline__ int32_t round_shift(int32_t v, int b) {
  return (v + ((1 << b) >> 1)) >> b;
}
__device__ __forceinline__ int32_t clamp_px(int32_t v, int32_t maxv) {
  return v < 0 ? 0 : (v > maxv ? maxv : v);
}
__device__ __forceinline__ int intermediate_bits(int bit_depth) {
  return bit_depth == 12 ? 2 : 4;
}

// ---- staging of the reference window into LDS ----
// A "slab" is up to 64 adjacent columns of one candidate block.  Its window is
// rows [ry-3, ry+h+4) x columns [rx-3, rx+P+4) of the reference plane -- the
// exact read footprint the reference documents (src/asm/x86/mc.rs:121-123).
// Rows are stored with a stride of `ws` bytes (multiple of 4).  `nl` lanes
// (lane index `l`) cooperate; global reads are unaligned dword loads, the
// last partial dword of a row is read bytewise so nothing outside the
// documented footprint is touched.
// XORM is xor-ed into every staged dword (0x80808080 turns u8 pixels into the
// biased i8 operands of v_dot4_i32_i8).
template <int BPP, uint32_t XORM = 0u>
__device__ __forceinline__ void stage_window(uint8_t *win, int ws,
                                             const R1Plane &ref, int rx, int ry,
                                             int P, int h, int l, int nl) {
  const int row_bytes = (P + 7) * BPP;
  const int nd = (row_bytes + 3) >> 2;
  const int total = (h + 7) * nd;
  const size_t gstride = (size_t)ref.stride * BPP;
  const uint8_t *g0 = px_addr<BPP>(ref, rx - 3, ry - 3);
  // batches of 4 independent loads per lane so that the global-load latency
  // is paid once per batch, not once per dword
  for (int i0 = l; i0 < total; i0 += 4 * nl) {
    uint32_t v[4];
    int off[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int i = i0 + u * nl;
      v[u] = 0;
      off[u] = -1;
      if (i < total) {
        const int r = i / nd, d = i - r * nd;
        // the last dword of a row may be partial: read the 4 bytes that END at
        // the row end and shift, so nothing outside the footprint is touched
        const int over = d * 4 + 4 - row_bytes;
        const int back = over > 0 ? over : 0;
        v[u] = (ld_u32(g0 + r * gstride + d * 4 - back) >> (8 * back)) ^ XORM;
        off[u] = r * ws + d * 4;
      }
    }
#pragma unroll
    for (int u = 0; u < 4; u++)
      if (off[u] >= 0) *(uint32_t *)(win + off[u]) = v[u];
  }
}

// Compile-time variant for the fused kernel: the whole window of a candidate
// (TOT dwords per lane) is requested before anything is written to LDS, so a
// wave pays ONE global-memory round trip for its window instead of one per
// batch of four loads.
template <int BPP, uint32_t XORM, int P, int H, int NL>
__device__ __forceinline__ void stage_window_ct(uint8_t *win, int ws, const R1Plane &ref,
                                                int rx, int ry, int l) {
  constexpr int ROW_BYTES = (P + 7) * BPP;
  constexpr int ND = (ROW_BYTES + 3) >> 2;
  constexpr int TOTAL = (H + 7) * ND;
  constexpr int PER = (TOTAL + NL - 1) / NL;
  const size_t gstride = (size_t)ref.stride * BPP;
  const uint8_t *g0 = px_addr<BPP>(ref, rx - 3, ry - 3);
  uint32_t v[PER];
  int off[PER];
#pragma unroll
  for (int u = 0; u < PER; u++) {
    const int i = l + u * NL;
    v[u] = 0;
    off[u] = -1;
    if (i < TOTAL) {
      const int r = i / ND, d = i - r * ND;
      const int over = d * 4 + 4 - ROW_BYTES;
      const int back = over > 0 ? over : 0;
      v[u] = (ld_u32(g0 + r * gstride + d * 4 - back) >> (8 * back)) ^ XORM;
      off[u] = r * ws + d * 4;
    }
  }
#pragma unroll
  for (int u = 0; u < PER; u++)
    if (off[u] >= 0) *(uint32_t *)(win + off[u]) = v[u];
}

// Row-mapped variant (used when a candidate has at least one lane per dword of
// a window row): lane l owns dword d = l % ND of rows r0 + u * RP, r0 = l / ND.
// All index arithmetic is done once per lane; per load only a 32-bit offset
// add remains, the LDS offset of pass u is an immediate.  (The generic
// element-mapped loop above spends ~17 VALU instructions per dword on
// div / mod / address math -- 23 % of the fused 8x8 kernel's time.)
template <int BPP, uint32_t XORM, int P, int H, int NL>
__device__ __forceinline__ void stage_window_rows(uint8_t *win, int ws, const R1Plane &ref,
                                                  int rx, int ry, int l) {
  constexpr int ROW_BYTES = (P + 7) * BPP;
  constexpr int ND = (ROW_BYTES + 3) >> 2;
  constexpr int RP = NL / ND;                       // rows per pass
  static_assert(RP >= 1, "needs a lane per dword of a row");
  constexpr int NR = H + 7;
  constexpr int PASSES = (NR + RP - 1) / RP;
  constexpr int OVER = ND * 4 - ROW_BYTES;          // bytes of the last dword beyond the row
  const int r0 = l / ND, d = l - r0 * ND;
  const bool lane_on = l < RP * ND;
  // the last dword of a row is read ending at the row end and shifted down, so
  // nothing outside the documented footprint is touched
  const int back = d == ND - 1 ? OVER : 0;
  const uint32_t gstride = (uint32_t)ref.stride * BPP;
  const uint8_t *base = (const uint8_t *)ref.data;
  uint32_t goff = ((uint32_t)(ref.yorigin + ry - 3 + r0) * (uint32_t)ref.stride +
                   (uint32_t)(ref.xorigin + rx - 3)) * BPP + (uint32_t)(d * 4 - back);
  uint32_t v[PASSES];
#pragma unroll
  for (int u = 0; u < PASSES; u++) {
    v[u] = 0;
    if (lane_on && r0 + u * RP < NR) v[u] = ld_u32(base + goff);
    goff += RP * gstride;
  }
  uint8_t *w0 = win + r0 * ws + d * 4;
  const uint32_t sh = 8u * (uint32_t)back;
#pragma unroll
  for (int u = 0; u < PASSES; u++)
    if (lane_on && r0 + u * RP < NR) *(uint32_t *)(w0 + u * RP * ws) = (v[u] >> sh) ^ XORM;
}

// 8-byte variant of the row mapping: lane l owns the 8-byte chunk ch = l % NDV
// of rows r0 + u * RP.  The last chunk of a row is read ENDING at the row end
// (it overlaps its neighbour) so the documented footprint is respected
// exactly, and shifted into place.  Half as many VMEM instructions as the
// dword version: for small blocks the kernel is bound by the number of
// scattered load instructions, not by bytes.
template <int BPP, uint32_t XORM, int P, int H, int NL>
__device__ __forceinline__ void stage_window_rows8(uint8_t *win, int ws, const R1Plane &ref,
                                                   int rx, int ry, int l) {
  constexpr int ROW_BYTES = (P + 7) * BPP;
  static_assert(ROW_BYTES >= 8, "window rows are at least 11 bytes");
  constexpr int NDV = (ROW_BYTES + 7) >> 3;         // 8-byte chunks per row
  constexpr int RP = NL / NDV;                      // rows per pass
  static_assert(RP >= 1, "needs a lane per chunk of a row");
  constexpr int NR = H + 7;
  constexpr int PASSES = (NR + RP - 1) / RP;
  constexpr int OVER = NDV * 8 - ROW_BYTES;         // 0..7: overlap of the last chunk
  constexpr int WSR = ((ROW_BYTES + 3) >> 2) << 2;  // bytes of an LDS row that may be written
  const int r0 = l / NDV, ch = l - r0 * NDV;
  const bool lane_on = l < RP * NDV;
  const int back = ch == NDV - 1 ? OVER : 0;
  const uint32_t gstride = (uint32_t)ref.stride * BPP;
  const uint8_t *base = (const uint8_t *)ref.data;
  uint32_t goff = ((uint32_t)(ref.yorigin + ry - 3 + r0) * (uint32_t)ref.stride +
                   (uint32_t)(ref.xorigin + rx - 3)) * BPP + (uint32_t)(ch * 8 - back);
  U32x2 v[PASSES];
#pragma unroll
  for (int u = 0; u < PASSES; u++) {
    v[u].a = 0; v[u].b = 0;
    if (lane_on && r0 + u * RP < NR) v[u] = ld_u32x2(base + goff);
    goff += RP * gstride;
  }
  uint8_t *w0 = win + r0 * ws + ch * 8;
  const uint32_t sh = 8u * (uint32_t)back;
  const bool second = ch * 8 + 4 < WSR;             // the chunk's upper dword lies inside the row
#pragma unroll
  for (int u = 0; u < PASSES; u++)
    if (lane_on && r0 + u * RP < NR) {
      const uint64_t q = ((((uint64_t)v[u].b) << 32) | v[u].a) >> sh;
      *(uint32_t *)(w0 + u * RP * ws) = (uint32_t)q ^ XORM;
      if (second) *(uint32_t *)(w0 + u * RP * ws + 4) = (uint32_t)(q >> 32) ^ XORM;
    }
}

// Picks the row-mapped staging when the geometry allows it.
template <int BPP, uint32_t XORM, int P, int H, int NL>
__device__ __forceinline__ void stage_window_fast(uint8_t *win, int ws, const R1Plane &ref,
                                                  int rx, int ry, int l) {
  constexpr int ND = ((P + 7) * BPP + 3) >> 2;
  constexpr int NDV = ((P + 7) * BPP + 7) >> 3;
  // 8-byte chunks pay for the few-lanes-per-candidate shapes (measured: 8x8 -4.6 %,
  // larger sizes neutral to slightly worse)
  if constexpr (NL <= 16 && NL >= NDV) stage_window_rows8<BPP, XORM, P, H, NL>(win, ws, ref, rx, ry, l);
  else if constexpr (NL >= ND) stage_window_rows<BPP, XORM, P, H, NL>(win, ws, ref, rx, ry, l);
  else stage_window_ct<BPP, XORM, P, H, NL>(win, ws, ref, rx, ry, l);
}

// The same three mappings with the global loads and the LDS writes as separate steps, so that a
// kernel can issue EVERYTHING it needs from memory (source block, window, tap tables) before it
// waits for any of it -- written as one call, the source's LDS write sits between the source's
// loads and the window's, and the wave pays two dependent round trips instead of one.
#ifndef R1_WIN_WIDE_STORE
#define R1_WIN_WIDE_STORE 1   // A/B switch (tools/build_variant.sh)
#endif
template <int BPP, uint32_t XORM, int P, int H, int NL>
struct WindowStage {
  static constexpr int ROW_BYTES = (P + 7) * BPP;
  static constexpr int ND = (ROW_BYTES + 3) >> 2;
  static constexpr int NDV = (ROW_BYTES + 7) >> 3;
  static constexpr int NR = H + 7;
  static constexpr int MODE = (NL <= 16 && NL >= NDV) ? 8 : (NL >= ND ? 4 : 0);   // as stage_window_fast
  // rows8 / rows
  static constexpr int CH = MODE == 8 ? NDV : ND;
  static constexpr int RP = MODE ? NL / CH : 1;
  static constexpr int PASSES = MODE ? (NR + RP - 1) / RP : 1;
  // element-mapped
  static constexpr int TOTAL = NR * ND;
  static constexpr int PER = (TOTAL + NL - 1) / NL;
  static constexpr int NV = MODE == 8 ? 2 * PASSES : (MODE == 4 ? PASSES : PER);
  uint32_t v[NV];
  int r0, ch, back;
  bool lane_on;
  int l;

  __device__ __forceinline__ void load(const R1Plane &ref, int rx, int ry, int lane) {
    l = lane;
    const uint32_t gstride = (uint32_t)ref.stride * BPP;
    const uint8_t *base = (const uint8_t *)ref.data;
    if constexpr (MODE != 0) {
      constexpr int CB = MODE;                                   // chunk bytes: 8 or 4
      constexpr int OVER = CH * CB - ROW_BYTES;                 // overlap of the last chunk of a row
      r0 = l / CH;
      ch = l - r0 * CH;
      lane_on = l < RP * CH;
      back = ch == CH - 1 ? OVER : 0;
      uint32_t goff = ((uint32_t)(ref.yorigin + ry - 3 + r0) * (uint32_t)ref.stride +
                       (uint32_t)(ref.xorigin + rx - 3)) * BPP + (uint32_t)(ch * CB - back);
#pragma unroll
      for (int u = 0; u < PASSES; u++) {
        if constexpr (MODE == 8) {
          v[2 * u] = 0; v[2 * u + 1] = 0;
          if (lane_on && r0 + u * RP < NR) { const U32x2 t = ld_u32x2(base + goff); v[2 * u] = t.a; v[2 * u + 1] = t.b; }
        } else {
          v[u] = 0;
          if (lane_on && r0 + u * RP < NR) v[u] = ld_u32(base + goff);
        }
        goff += RP * gstride;
      }
    } else {
      const uint8_t *g0 = px_addr<BPP>(ref, rx - 3, ry - 3);
#pragma unroll
      for (int u = 0; u < PER; u++) {
        const int i = l + u * NL;
        v[u] = 0;
        if (i < TOTAL) {
          const int r = i / ND, d = i - r * ND;
          const int over = d * 4 + 4 - ROW_BYTES;
          const int bk = over > 0 ? over : 0;
          v[u] = (ld_u32(g0 + (size_t)r * gstride + d * 4 - bk) >> (8 * bk)) ^ XORM;
        }
      }
    }
  }

  __device__ __forceinline__ void store(uint8_t *win, int ws) const {
    if constexpr (MODE == 8) {
      constexpr int WSR = ((ROW_BYTES + 3) >> 2) << 2;          // bytes of an LDS row that may be written
      uint8_t *w0 = win + r0 * ws + ch * 8;
      const uint32_t sh = 8u * (uint32_t)back;
      const bool second = ch * 8 + 4 < WSR;
      // every chunk's second dword lies inside the row (P = 8 at either pixel width, P = 4 at 16 bits):
      // one ds_write_b64 per pass instead of two ds_write_b32 -- half the LDS write instructions of the
      // staging, and a 16-lane group of a b64 store covers two candidates' chunks instead of four
      // (the b32 pairs of neighbouring candidates met on the same banks: SQ_LDS_BANK_CONFLICT was
      // 18 % / 30 % of the LDS cycles of the 8-bit / 10-bit 8x8 launch, all of it here)
      constexpr bool WIDE = R1_WIN_WIDE_STORE && (CH - 1) * 8 + 4 < WSR;
#pragma unroll
      for (int u = 0; u < PASSES; u++)
        if (lane_on && r0 + u * RP < NR) {
          const uint64_t q = ((((uint64_t)v[2 * u + 1]) << 32) | v[2 * u]) >> sh;
          if constexpr (WIDE) {
            *(uint2 *)(w0 + u * RP * ws) = make_uint2((uint32_t)q ^ XORM, (uint32_t)(q >> 32) ^ XORM);
          } else {
            *(uint32_t *)(w0 + u * RP * ws) = (uint32_t)q ^ XORM;
            if (second) *(uint32_t *)(w0 + u * RP * ws + 4) = (uint32_t)(q >> 32) ^ XORM;
          }
        }
    } else if constexpr (MODE == 4) {
      uint8_t *w0 = win + r0 * ws + ch * 4;
      const uint32_t sh = 8u * (uint32_t)back;
#pragma unroll
      for (int u = 0; u < PASSES; u++)
        if (lane_on && r0 + u * RP < NR) *(uint32_t *)(w0 + u * RP * ws) = (v[u] >> sh) ^ XORM;
    } else {
#pragma unroll
      for (int u = 0; u < PER; u++) {
        const int i = l + u * NL;
        if (i < TOTAL) {
          const int r = i / ND, d = i - r * ND;
          *(uint32_t *)(win + r * ws + d * 4) = v[u];
        }
      }
    }
  }
};

// One column of put_8tap / prep_8tap from a staged window.  `c` is the column
// inside the slab, `w`/`h` the full block size (they select the 4-tap filter
// variants).  emit(r, value) receives each output sample: the clamped pixel
// for put, the int16 intermediate for prep.
// HT > 0 fixes the height at compile time so that every row loop unrolls and
// emit() may target a register array (used by the fused RDO kernel).
template <int BPP, bool PREP, int HT, typename Emit>
__device__ __forceinline__ void mc_column(const uint8_t *win, int ws, int c,
                                          int w, int h_rt, int col_frac,
                                          int row_frac, int mode_x, int mode_y,
                                          int bit_depth, Emit emit) {
  const int h = HT > 0 ? HT : h_rt;
  const int ib = intermediate_bits(bit_depth);
  const int32_t maxv = (1 << bit_depth) - 1;
  const int32_t bias = bit_depth == 8 ? 0 : 8192;  // PREP_BIAS
  const uint8_t *col = win + c * BPP;
  if (col_frac == 0 && row_frac == 0) {
#pragma unroll
    for (int r = 0; r < h; r++) {
      const int32_t p = ld_px<BPP>(col + (r + 3) * ws + 3 * BPP);
      emit(r, PREP ? (int32_t)(int16_t)((int16_t)(p << ib) - (int16_t)bias) : p);
    }
  } else if (col_frac == 0) {
    const int16_t *yf = get_filter(mode_y, row_frac, h);
    int32_t f[8], t[8];
#pragma unroll
    for (int k = 0; k < 8; k++) f[k] = yf[k];
#pragma unroll
    for (int k = 0; k < 7; k++) t[k + 1] = ld_px<BPP>(col + k * ws + 3 * BPP);
#pragma unroll
    for (int r = 0; r < h; r++) {
#pragma unroll
      for (int k = 0; k < 7; k++) t[k] = t[k + 1];
      t[7] = ld_px<BPP>(col + (r + 7) * ws + 3 * BPP);
      int32_t s = 0;
#pragma unroll
      for (int k = 0; k < 8; k++) s += f[k] * t[k];
      emit(r, PREP ? (int32_t)(int16_t)(round_shift(s, 7 - ib) - bias)
                   : clamp_px(round_shift(s, 7), maxv));
    }
  } else if (row_frac == 0) {
    const int16_t *xf = get_filter(mode_x, col_frac, w);
    int32_t f[8];
#pragma unroll
    for (int k = 0; k < 8; k++) f[k] = xf[k];
#pragma unroll
    for (int r = 0; r < h; r++) {
      int32_t s = 0;
#pragma unroll
      for (int k = 0; k < 8; k++) s += f[k] * ld_px<BPP>(col + (r + 3) * ws + k * BPP);
      emit(r, PREP ? (int32_t)(int16_t)(round_shift(s, 7 - ib) - bias)
                   : clamp_px(round_shift(round_shift(s, 7 - ib), ib), maxv));
    }
  } else {
    const int16_t *xf = get_filter(mode_x, col_frac, w);
    const int16_t *yf = get_filter(mode_y, row_frac, h);
    int32_t fx[8], fy[8], m[8];
#pragma unroll
    for (int k = 0; k < 8; k++) { fx[k] = xf[k]; fy[k] = yf[k]; }
    auto hrow = [&](int rr) -> int32_t {
      int32_t s = 0;
#pragma unroll
      for (int k = 0; k < 8; k++) s += fx[k] * ld_px<BPP>(col + rr * ws + k * BPP);
      return (int32_t)(int16_t)round_shift(s, 7 - ib);  // i16 intermediate
    };
#pragma unroll
    for (int k = 0; k < 7; k++) m[k + 1] = hrow(k);
#pragma unroll
    for (int r = 0; r < h; r++) {
#pragma unroll
      for (int k = 0; k < 7; k++) m[k] = m[k + 1];
      m[7] = hrow(r + 7);
      int32_t s = 0;
#pragma unroll
      for (int k = 0; k < 8; k++) s += fy[k] * m[k];
      emit(r, PREP ? (int32_t)(int16_t)(round_shift(s, 7) - bias)
                   : clamp_px(round_shift(s, 7 + ib), maxv));
    }
  }
}

}  // namespace r1mc
