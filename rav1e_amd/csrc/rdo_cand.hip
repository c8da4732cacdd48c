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
// candidates.
//  A  every lane pulls its source column into registers (H loads in flight)
//     while the wave stages each candidate's (H+7)x(W+7) reference window in
//     LDS (batched unaligned dword loads).
//  B  lane = (candidate, column) filters its column.  8-bit pixels: three
//     aligned LDS dwords per window row, v_alignbyte to the lane's byte
//     phase, pixels biased by -128 so that the horizontal 8 taps are two
//     v_dot4_i32_i8; the i16 intermediates are packed in pairs and the
//     vertical 8 taps are 4 (even rows) or 5 (odd rows) v_dot2_i32_i16.
//     All four (col_frac==0?, row_frac==0?) cases of mc.rs:264-352 go through
//     this one code path: a 128-valued tap at phase 0 reproduces the copy and
//     1-D paths bit for bit (see the derivation at mc8_column).
//     The residual COLUMN stays in registers.  SAD is a lane sum.  SATD: the
//     vertical Hadamard on 8 registers, the horizontal one across the 8
//     neighbouring lanes with DPP (quad_perm / row_half_mirror) -- no LDS.
//  C  column transform on the same registers (24-bit multiplies, exact here),
//     transpose through LDS (odd stride, aliasing the dead window).
//  D  lane = (candidate, row): row transform, stores in the reference's
//     transposed 32x32-chunk coefficient order.
#include <cstdlib>
#include <type_traits>

#include "dist_common.hpp"
#include "itx_common.hpp"
#include "mc_common.hpp"
#include "quant_common.hpp"
#include "tx_common.hpp"

#ifdef R1_PHASE_PROF
// experiment build only (make prof): wall-clock cycles a wave spends in each phase of the
// headline kernel, summed over all waves; read back by r1_debug_phase_prof().
// every 64th workgroup writes its own row: no atomics, so the probes do not queue up
__device__ unsigned long long g_phase[4096][8];
#define R1_PROF(i)                                                          \
  do {                                                                      \
    const unsigned long long t_ = __builtin_readcyclecounter();             \
    if (QM == 0 && threadIdx.x == 0 && (blockIdx.x & 63) == 0 && (blockIdx.x >> 6) < 4096) \
      g_phase[blockIdx.x >> 6][i] = t_ - tprev_;                            \
    tprev_ = t_;                                                            \
  } while (0)
#define R1_PROF_INIT unsigned long long tprev_ = __builtin_readcyclecounter()
#else
#define R1_PROF(i) do {} while (0)
#define R1_PROF_INIT do {} while (0)
#endif

// which instantiations stage the source block in LDS (see k_rdo_cand); overridable for A/B builds
#ifndef R1_SRC_LDS_POLICY
#define R1_SRC_LDS_POLICY(BPP, P) ((BPP) == 1 || (P) <= 32)
#endif
// (Round 3 tried a software pipeline over two candidate groups per wave -- next group's loads in
// flight under this group's arithmetic.  Measured, profiles/r03_ab_notes.md ab3: -2.3 % on the 8-bit
// 8x8 launch, a LOSS everywhere else (the launches are VALU-issue bound; the registers of the loads
// in flight cost more occupancy than the hidden round trip is worth), and the machine scheduler did
// not terminate on the QM = 2 instantiations of that loop.  Not kept.)
// Translation units.  k_rdo_cand has 19 sizes x 3 bit depths x 3 QM variants = 171 instantiations;
// compiled in one piece they take minutes of one core.  The Makefile compiles this
// file ten times: nine slices (-DR1_RDO_TU_BD=8|10|12 -DR1_RDO_TU_QM=0|1|2: the kernel and one
// r1_rdo_slice_b*_q* launcher each) in parallel, and once without either macro (k_mc_fast, the
// dispatch and the entry points).  Experiment builds (-DR1_HEADLINE_ONLY) are one unit.
#if defined(R1_HEADLINE_ONLY)
#define R1_RDO_SLICE_TU
#define R1_RDO_DISPATCH_TU
#elif defined(R1_RDO_TU_BD)
#define R1_RDO_SLICE_TU
#else
#define R1_RDO_DISPATCH_TU
#endif

// which instantiations run their dead candidate slots unmasked (see k_rdo_cand)
#ifndef R1_TB16_POLICY
#define R1_TB16_POLICY(BD, WL, HL) ((BD) == 10 && (WL) == 5 && (HL) == 5)
#endif
#ifndef R1_TX_TILE_I16
#define R1_TX_TILE_I16 1
#endif
// the coefficient blocks leave with non-temporal stores (see the store loops of k_rdo_cand)
#ifndef R1_NT_STORE
#define R1_NT_STORE 1
#endif
#ifndef R1_XCD_REMAP
#define R1_XCD_REMAP 1   // A/B switch (see k_rdo_cand)
#endif
#ifndef R1_SRC_KEEP
#define R1_SRC_KEEP 1   // A/B switch (see k_rdo_cand)
#endif
#ifndef R1_SRC_PAD
#define R1_SRC_PAD 1   // A/B switch: the padded source-block stride in LDS (see k_rdo_cand)
#endif
// which instantiations keep the source chunks in registers across the filter and stage them over the
// dead window afterwards (see k_rdo_cand)
#ifndef R1_SRC_LATE_POLICY
#define R1_SRC_LATE_POLICY(BD, P, QM) ((BD) != 8 && ((P) == 32 || ((P) == 16 && (QM) == 0)))
#endif
#ifndef R1_UNMASK_POLICY
#define R1_UNMASK_POLICY(BD, WL, HL) (!((BD) == 8 && (WL) == 6 && (HL) == 6))
#endif
// which instantiations send the coefficients through LDS for 16-byte stores
#ifndef R1_WIDE_STORE_POLICY
#define R1_WIDE_STORE_POLICY(P) ((P) <= 16)
#endif

namespace {
using r1tx::T;

#include "mc_taps_packed.inc"

#include "cand_helpers.inc"

// QUANT (the "full" candidate): the coefficients do not go to HBM
// (unless `coeffs` is also given) but through the quantizer in place --
// quantize + dequantize + transform-domain distortion + estimate_rate, i.e.
// encode_tx_block's RDOType::TxDistEstRate evaluation (src/encoder.rs:1533-1650)
// -- and only (eob, distortion, rate) leave the CU.
}  // namespace
struct RdoQuantArgs {
  r1q::QParams qp;
  const uint16_t *scan[3];   // av1_scan_orders[tx_size]: default / mrow / mcol
  int tx_size, q_bin;
  uint16_t *eob;
  unsigned long long *tx_dist, *est_rate;
  void *qcoeffs;             // optional: dense coded-area blocks
  // QM == 2 (pixel-domain leg): dequantize -> inverse transform -> reconstruct ->
  // sse_wxh / cdef_dist_wxh against the source with the DistortionScale grid
  // (encode_tx_block with need_recon_pixel / compute_distortion, src/rdo.rs:254-340)
  int dist_kind, inv_shift;
  const uint32_t *scales;
  int scale_stride, xdec, ydec;
  unsigned long long *pix_dist;
  void *rec;                 // optional: dense w*h reconstructions
  // prediction from a dense buffer (n x h x w pixels: intra predictions, compound
  // averages) instead of put_8tap of the reference plane
  const void *pred_in;
  // MT (transform-type search fan-out, rdo_tx_type_decision src/rdo.rs:1701-1817): every candidate is
  // carried through the chain once per set bit of tx_mask (bit t = TxType t, ascending), on ONE
  // prediction / residual; result slot of (candidate i, j-th set bit) = i * nt + j, nt = popcount
  uint32_t tx_mask;
  int nt;
  // plain (non-MT) kernels: tx_mask != 0 forces the type of every candidate to its lowest set bit and the
  // results go to slot `slot` of nt (sizes with a 32-point side: one launch per type, see r1_rdo_txsearch_batch)
  int slot;
};
namespace {
using r1tx::T;

// QM: 0 = coefficients to HBM (headline), 1 = + quantizer, tx-domain distortion,
// rate (N4), 2 = + quantizer, inverse transform, pixel-domain distortion.
// Waves per SIMD the register allocator is asked to make room for (0 = no request).  Only
// where the kernel sits a few registers above an allocation step (512 / n, in eights) and
// the step costs no spill worth mentioning -- measured, see DESIGN.md 5.1 "occupancy".
constexpr int rdo_waves_hint(int bd, int wl, int hl, int qm, bool mt = false) {
  // the type-search instantiations (see the MT loop of k_rdo_cand): their own steps
  if (mt) {
#ifdef R1_HINT_MT
    if (R1_HINT_MT_COND) return R1_HINT_MT;
#endif
    // what the straight-line kernels are asked for leaves the loop with 50-350 B of scratch per lane, and at
    // thousands of waves in flight that is traffic to the Infinity Cache: same-box A/B (r05_ab_notes.md, ab1)
    // 16x16 fan-out 0.754 -> 0.610 ms (8-bit), 0.855 -> 0.602 (10-bit), 10-bit 8x8 0.655 -> 0.553 at the steps below
#ifndef R1_MT_H8
#define R1_MT_H8 (qm == 2 ? (bd == 8 ? 7 : 6) : 1)
#endif
#ifndef R1_MT_H16
#define R1_MT_H16 (qm == 2 ? 4 : 1)
#endif
    if (wl <= 3 && hl <= 3) return R1_MT_H8;
    return R1_MT_H16;
  }
#ifdef R1_HINT_8X8
  if (wl == 3 && hl == 3 && qm == 0) return R1_HINT_8X8;   // A/B: the pipelined 8x8 kernel sits at 69 (8-bit)
#endif
#ifdef R1_HINT_16X16
  if (wl == 4 && hl == 4 && qm == 0) return R1_HINT_16X16;
#endif
#ifdef R1_HINT_64_HBD
  if (wl == 6 && hl == 6 && qm == 0 && bd != 8) return R1_HINT_64_HBD;   // A/B: 10-bit 64x64 sits at 165 (3 waves); 4 = 152 B of spills, launch 0.281 -> 0.365 ms
#endif
  if (wl == 5 && hl == 5 && qm == 0 && bd != 8) return 6;   // 89 -> 80 VGPRs, no spill: 5 -> 6 waves, launch -1.7 % (ab7)
#ifdef R1_HINT_X   /* A/B: -DR1_HINT_X=5 '-DR1_HINT_X_COND=(bd==8&&wl==6&&hl==6&&qm==0)' */
  if (R1_HINT_X_COND) return R1_HINT_X;
#endif
#ifdef R1_HINT_Y
  if (R1_HINT_Y_COND) return R1_HINT_Y;
#endif
  // the pixel-domain chain sat a few registers above an allocation step at three sizes; asked for the
  // step, the allocator gets there without a spill worth mentioning (same-box, r04_ab_notes.md ab8:
  // 8-bit 290.3 -> 295.0 k, 10-bit 274.2 -> 283.5 k)
  if (wl == 3 && hl == 3 && qm == 2) return 8;   // 73 / 74 VGPRs -> 58 / 62: 6 -> 8 waves, launch -2.2 / -3.7 % (10-bit at 7: 72 VGPRs, -1.6 %; ab13)
  if (wl == 4 && hl == 4 && qm == 2) return bd == 8 ? 6 : 5;   // 105 / 107 -> 80 + 44 B scratch / 94: 4 -> 6 / 5 waves, -6 / -5.1 % (10-bit at 6: 68 B of scratch, +1 %; ab13)
  if (wl == 5 && hl == 5 && qm == 2) return 4;              // 8-bit 132 -> 128; 10-bit 131 -> 128 (8 B of scratch): 3 -> 4 waves, -5.4 %
  if (wl == 5 && hl == 5 && qm == 1) return 5;              // 8-bit 97 -> 96; 10-bit 120 -> 96 (20 B of scratch): 4 -> 5 waves, launch -4 % (ab10)
  if (wl == 6 && hl == 6 && qm == 2) return bd == 8 ? 4 : 3;   // 8-bit: 168 -> 128 + 64 B of scratch, launch -2.9 %; 10-bit at 4: +14 % (ab12) -> 3: 181 -> 168
  return 1;
}

template <int BD, int WL, int HL, typename CT, int QM, bool MT = false>
__global__ __launch_bounds__(64, rdo_waves_hint(BD, WL, HL, QM, MT)) void k_rdo_cand(
    R1Plane org, R1Plane ref, const R1RdoCand *__restrict__ cands, int n,
    uint32_t *__restrict__ sad_out, uint32_t *__restrict__ satd_out,
    CT *__restrict__ coeffs, void *__restrict__ pred_out, RdoQuantArgs qa) {
  constexpr int BPP = BD == 8 ? 1 : 2;
  // forward-transform shifts of this (size, bit depth): immediates
  constexpr int SH0 = r1tx::fwd_shift_ct(WL, HL, BD, 0), SH1 = r1tx::fwd_shift_ct(WL, HL, BD, 1),
                SH2 = r1tx::fwd_shift_ct(WL, HL, BD, 2);
  constexpr int W = 1 << WL, H = 1 << HL;
  constexpr int P = W > H ? W : H, NC = 64 / P;
  constexpr int TS = (W < H ? W : H) == 4 ? 4 : 8;
  constexpr int WS = (((W + 7) * BPP + 3) >> 2) << 2;   // window row stride
  constexpr int WIN_BYTES = NC * (H + 7) * WS;
  // The transpose tile holds the column pass's outputs after shift[1]: bounded by 16353 at 8-bit
  // (every size and type) and by 23214 at 10-bit with sides up to 32 (tools/tx_range.py: pixel range,
  // shift[0], the L1 gain of the column network, shift[1]), so int16 holds them exactly.  Used for
  // 10-bit 32x32 only, together with SRC_LATE below: tile 8320 -> 4224 B, window + source 10336 ->
  // 6240 B, 4 -> 5 waves per SIMD, launch 0.252 -> 0.237 ms (profiles/r04_ab_notes.md).  At 8-bit
  // 32x32 the same change (5 -> 6 waves) made the launch 1.5 % SLOWER -- that kernel is not short of
  // waves -- and is off.  Row stride 66 int16 = 33 dwords: a candidate's row lanes read 32 banks.
  // The type search's shared tile (COLSHARE, below) is int16 at every bit depth: with both sides <= 16 the column
  // pass's output after shift[1] is bounded by 8193 / 16433 / 16445 at 8 / 10 / 12 bits (tests/test_tx_range.py).
#ifndef R1_MT_COLSHARE
#define R1_MT_COLSHARE 1
#endif
  constexpr bool TB16 = R1_TX_TILE_I16 && (R1_TB16_POLICY(BD, WL, HL) || (R1_MT_COLSHARE && MT && W <= 16 && H <= 16));
  typedef typename std::conditional<TB16, int16_t, T>::type TB;
  constexpr int LSTRIDE = NC * W + (TB16 ? 2 : 1);
  constexpr int ISTRIDE = NC * W + 1;       // the inverse transform's row buffer (QM == 2): int32
  // 64x64: the transpose goes through LDS in two halves of 32 rows (8.3 KB instead of
  // 16.6 KB per wave).  At 16.6 KB the CU held 9 waves where the registers allow 12, and
  // this kernel lives on occupancy: a wave issues one instruction per ~10 cycles whatever
  // the size, so the SIMD's throughput is proportional to the waves it holds.
  constexpr bool SPLIT_T = W == 64 && H == 64;
  constexpr int TXB_ROWS = SPLIT_T ? 32 : H;
  constexpr int TXB_BYTES = TXB_ROWS * LSTRIDE * (int)sizeof(TB);
  constexpr int IRB_BYTES = QM == 2 ? (H < 32 ? H : 32) * ISTRIDE * 4 : 0;
  // The quantizer's coded-area tile, one per candidate, P dwords of padding between candidates: at the bare
  // stride (64 / 128 / 256 dwords for 8x8 .. 16x16) the NC candidates of a lane group wrote, gathered and read
  // back the same banks (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE 0.36 / 0.26 of the pixel chain's 8x8 / 16x16
  // launches, profiles/r04_v5_pmc_pixel_summary.json); the forward kernel's TPAD, carried over (R1_QT_PAD: A/B)
#ifndef R1_QT_PAD
#define R1_QT_PAD 1
#endif
  // (two candidates per wave sit in different 32-lane groups and never meet in a bank: no padding there --
  // with it the 10-bit 32x32 launch was 2.8 % slower, r05_ab_notes.md ab2)
  constexpr int QT_PAD = (R1_QT_PAD && NC > 2) ? P : 0;
  constexpr int QT_STRIDE = (W < 32 ? W : 32) * (H < 32 ? H : 32) + QT_PAD;
  constexpr int QT_BYTES = QM != 0 ? NC * QT_STRIDE * 4 : 0;
  // QM == 2 (pixel-domain leg): only the coded area (32 x 32 of a 64-point side) is quantized, and there is no
  // `tail` energy to sum (encoder.rs:1617-1640 computes it only when rdo_type.needs_tx_dist()), so vertical
  // frequencies >= 32 are never read: the column pass does not store them -- the compiler then prunes the
  // fdct64 network down to the outputs that are (its upper-half outputs are dead) -- and the row pass runs
  // on rows 0 .. 31 only; horizontal frequencies >= 32 die the same way inside the row lanes (R1_PRUNE64: A/B)
#ifndef R1_PRUNE64
#define R1_PRUNE64 1
#endif
  constexpr int HU = (R1_PRUNE64 && QM == 2 && H > 32) ? 32 : H;   // vertical frequencies that are used
  constexpr int REC_BYTES = QM == 2 ? NC * W * H * BPP : 0;
  // The source block is staged in LDS next to the window (16-byte row chunks: H*W*BPP/1024
  // load instructions per wave instead of H one-pixel-per-lane loads) and read back column by
  // column AFTER the motion compensation: the H source registers are not live across the
  // filter any more.  Not for 64-wide 16-bit blocks: + 8 KB of LDS would cost a wave per SIMD.
  constexpr bool SRC_LDS = R1_SRC_LDS_POLICY(BPP, P);
  // SRC_LATE: the source chunks wait in registers (SPASS x 4 VGPRs) while the window is filtered and
  // go to LDS afterwards, OVER the dead window -- window + source side by side (10336 B at 10-bit
  // 32x32) held the CU at 15 waves (4 per SIMD after rounding); with the source over the window the
  // footprint is the window's 6240 B and the ~93 VGPRs allow 5.
  // 16-bit 16x16, headline only (the pixel chain keeps its source block in LDS for the distortion,
  // SRC_KEEP below): 6592 -> 4416 B, 6 -> 8 waves, launch 0.2255 -> 0.217 ms (r04_ab_notes.md, ab7)
  constexpr bool SRC_LATE = SRC_LDS && R1_SRC_LATE_POLICY(BD, P, QM);
  constexpr int SRC_ROW = W * BPP;
  constexpr int WIN_PAD = (WIN_BYTES + 15) & ~15;
  // A candidate's source block starts max(16, row bytes) past a multiple of its own size: with the bare
  // stride (16 / 32 / 64 / 128 dwords) the column reads of the NC candidates of a lane group hit the SAME
  // banks with different addresses -- 2-way at 8-bit 8x8 and at 16x16, 4-way at 10-bit 8x8: this, not the
  // window staging, was the SQ_LDS_BANK_CONFLICT of those launches (0.18 / 0.30 of the LDS cycles)
  constexpr int SRC_CSTRIDE = H * SRC_ROW + (NC > 1 && R1_SRC_PAD ? (SRC_ROW > 16 ? SRC_ROW : 16) : 0);
  constexpr int SRC_BYTES = SRC_LDS ? NC * SRC_CSTRIDE : 0;
  // SRC_KEEP (pixel-domain chain, blocks up to 16 rows): the staged source block sits BEHIND the work area
  // that the later phases alias (transpose tile, quantizer tile, row buffer), so the distortion at the end of
  // the chain reads its source column from LDS again instead of issuing H more global loads per lane
  constexpr bool SRC_KEEP = R1_SRC_KEEP && QM == 2 && H <= 16 && SRC_LDS && !SRC_LATE;
  constexpr int WS_BYTES = SRC_KEEP ? WIN_PAD
                                    : (SRC_LATE ? (WIN_PAD > SRC_BYTES ? WIN_PAD : SRC_BYTES) : WIN_PAD + SRC_BYTES);
  constexpr int LDS_A0 = WS_BYTES > TXB_BYTES ? WS_BYTES : TXB_BYTES;
  constexpr int LDS_A = LDS_A0 > IRB_BYTES ? LDS_A0 : IRB_BYTES;
  constexpr int LDS_B = QT_BYTES > REC_BYTES ? QT_BYTES : REC_BYTES;
  constexpr int LDS_WORK = ((LDS_A > LDS_B ? LDS_A : LDS_B) + 15) & ~15;
  constexpr int SRC_OFF = SRC_KEEP ? LDS_WORK : (SRC_LATE ? 0 : WIN_PAD);
  // MT, COLSHARE: the transposed output of the column pass in a tile of its own behind everything else -- the later
  // phases of a type alias the work area, and the types that share a column kernel (the seven RAV1E types use three:
  // DCT x3, ADST x2, identity x2) all read their rows from this one tile (see the type loop)
  constexpr bool COLSHARE = R1_MT_COLSHARE && MT && !SPLIT_T;
  constexpr int TKEEP_OFF = (LDS_WORK + (SRC_KEEP ? SRC_BYTES : 0) + 15) & ~15;
  constexpr int LDS_BYTES = COLSHARE ? TKEEP_OFF + H * LSTRIDE * (int)sizeof(TB) : LDS_WORK + (SRC_KEEP ? SRC_BYTES : 0);
  __shared__ __attribute__((aligned(16))) uint8_t smem[LDS_BYTES];
  T *buf = (T *)smem;
  TB *tbuf = (TB *)smem;
  TB *tkeep = COLSHARE ? (TB *)(smem + TKEEP_OFF) : tbuf;

  R1_PROF_INIT;
  // Workgroup -> candidate group, XCD-aware.  The dispatcher deals workgroups round-robin over the 8 XCDs
  // (workgroup i runs on XCD i % 8), each with its own L2.  Consecutive candidate groups belong to the same
  // block (the K candidates of a block sit next to each other in the list and share the source block and
  // most of their reference windows): in dispatch order they would land on 8 different L2s and each would
  // fetch the window rows again.  So XCD x takes the x-th contiguous eighth of the list: workgroup i works on
  // group (i % 8) * (grid / 8) + i / 8 (the host rounds the grid up to a multiple of 8; groups past the
  // list end return).  Same-box A/B: profiles/r04_ab_notes.md, ab5.
#if R1_XCD_REMAP
  const unsigned wg = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  if ((long long)wg * NC >= (long long)n) return;
#else
  const unsigned wg = blockIdx.x;
#endif
  const int lane = threadIdx.x;
  const int cl = lane / P, c = lane % P;
  // n < 2^31 candidates: the liveness test and the lane-local parts of every address are 32-bit;
  // what is 64-bit is the workgroup's base (wg * per-workgroup bytes), which the scalar
  // unit computes
  const int cand_i = (int)wg * NC + cl;
  const long long cand = cand_i;
  // Only STORES look at whether this lane's candidate exists (live_st).  The dead slots of the
  // launch's last wave load and compute the launch's last candidate once more: no masked regions,
  // no zero-initialised registers for the lanes that would have sat out (35 v_mov of the 8x8
  // kernel's 766 VALU instructions), every wave runs the same straight line.
  // Measured (profiles/r03_ab_notes.md, ab4): -3 % at 8x8, -1 % at 16x16 / 32x32, +2 % on the 10-bit
  // step; the 8-bit 64x64 instantiation alone loses (121 -> 143 VGPRs, 4 -> 3 waves per SIMD) and
  // keeps its masked regions.
  constexpr bool UNMASK = R1_UNMASK_POLICY(BD, WL, HL);
  const bool live_st = cand_i < n;
  const bool live = UNMASK || live_st;
  const int cl_ld = live_st ? cl : n - 1 - (int)wg * NC;     // >= 0: the wave's first candidate exists
  const long long cand_ld = live_st ? cand : (long long)n - 1;
  R1RdoCand cd = {};
  if (live) cd = (cands + (size_t)wg * NC)[cl_ld];
#ifdef R1_PHASE_PROF
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  R1_PROF(5);   // A0: descriptor round trip
#endif
  constexpr bool QUANT = QM != 0;
  // QM == 2 keeps the prediction column (packed pixels) for the reconstruction
  constexpr int PPK = QM == 2 ? (H * BPP + 3) / 4 : 1;
  uint32_t ppk[PPK];
#pragma unroll
  for (int k = 0; k < PPK; k++) ppk[k] = 0;

  // ---- A: source block (registers, or LDS in wide chunks), reference window into LDS ----
  T v[H];
#pragma unroll
  for (int r = 0; r < H; r++) v[r] = 0;
  const bool col_live = live && c < W;
  const uint8_t *src_l = smem + SRC_OFF + cl * SRC_CSTRIDE + c * BPP;
  // A.1: every global load the wave needs goes out before it waits for any of them -- source
  // block, reference window, tap tables depend on the descriptor only (one round trip behind it,
  // not three)
  constexpr int CHS = SRC_ROW >= 16 ? 16 : SRC_ROW;      // source bytes per lane per pass
  constexpr int CPR = SRC_ROW / CHS;                      // chunks per source row
  constexpr int RPP = P / CPR;                            // rows per pass (P lanes per candidate)
  constexpr int SPASS = SRC_LDS ? (H + RPP - 1) / RPP : 1;
  const int srow = c / CPR, sch = c - srow * CPR;
  U32x4 q[SPASS];
  if constexpr (SRC_LDS) {
    // planes are far below 4 GB: a 32-bit byte offset from the allocation's start
    const uint8_t *po = (const uint8_t *)org.data +
                        (((uint32_t)(org.yorigin + cd.oy) * (uint32_t)org.stride + (uint32_t)(org.xorigin + cd.ox)) * BPP +
                         (uint32_t)(sch * CHS));
    const uint32_t so = (uint32_t)org.stride * BPP;
#pragma unroll
    for (int u = 0; u < SPASS; u++) {
      const int rr = srow + u * RPP;
      q[u] = U32x4{0, 0, 0, 0};
      if (live && rr < H) {
        if constexpr (CHS == 16) q[u] = ld_u32x4(po + rr * so);
        else if constexpr (CHS == 8) { const U32x2 t = ld_u32x2(po + rr * so); q[u].a = t.a; q[u].b = t.b; }
        else q[u].a = ld_u32(po + rr * so);
      }
    }
  } else if (col_live) {
    const uint8_t *po = px_addr<BPP>(org, cd.ox + c, cd.oy);
    const size_t so = (size_t)org.stride * BPP;
#pragma unroll
    for (int r = 0; r < H; r++) v[r] = ld_px<BPP>(po + r * so);
  }
  uint8_t *win = smem + cl * (H + 7) * WS;
  const bool from_ref = live && !qa.pred_in;   // !pred_in is wave-uniform: kernel argument
  r1mc::WindowStage<BPP, BPP == 1 ? 0x80808080u : 0u, W, H, P> wst;
  if (from_ref) wst.load(ref, cd.rx, cd.ry, c);
  typename std::conditional<BPP == 1, Taps8, Taps16>::type tp = {};
  if (from_ref) {
    if constexpr (BPP == 1) tp = load_taps8<W, H>(cd.col_frac, cd.row_frac, cd.mode_x, cd.mode_y);
    else tp = load_taps16<W, H>(cd.col_frac, cd.row_frac, cd.mode_x, cd.mode_y);
  }
  // every filter of the wave with zero outer taps (anything but SHARP): the short column filter
  const bool six = taps_six(tp);
  // A.2: into LDS
  auto stage_source = [&]() {
    if (live) {
      uint8_t *sd = smem + SRC_OFF + cl * SRC_CSTRIDE + sch * CHS;
#pragma unroll
      for (int u = 0; u < SPASS; u++) {
        const int rr = srow + u * RPP;
        if (rr < H) {
          if constexpr (CHS == 16) *(uint4 *)(sd + rr * SRC_ROW) = make_uint4(q[u].a, q[u].b, q[u].c, q[u].d);
          else if constexpr (CHS == 8) *(uint2 *)(sd + rr * SRC_ROW) = make_uint2(q[u].a, q[u].b);
          else *(uint32_t *)(sd + rr * SRC_ROW) = q[u].a;
        }
      }
    }
  };
  if constexpr (SRC_LDS && !SRC_LATE) stage_source();
  if (from_ref) wst.store(win, WS);
#ifdef R1_PHASE_PROF
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  R1_PROF(6);   // A1: source column + window round trip (+ LDS writes issued)
#endif
  __syncthreads();
  R1_PROF(0);   // A: descriptor, source column, window staged

  // ---- B: prediction column, residual, SAD / SATD ----
  uint32_t sad_acc = 0;
  if constexpr (BPP == 1) {
    if (col_live) {
      int32_t pred[H];
      if (qa.pred_in) {
        const uint8_t *pi = (const uint8_t *)qa.pred_in + (size_t)cand_ld * W * H + c;
#pragma unroll
        for (int r = 0; r < H; r++) pred[r] = pi[(size_t)r * W];
      } else {
        mc8_column_t<W, H, WS, false>(win, c, tp, pred, six);
      }
      if (pred_out && live_st) {
        uint8_t *pp = (uint8_t *)pred_out + (size_t)cand * W * H + c;
#pragma unroll
        for (int r = 0; r < H; r++) pp[(size_t)r * W] = (uint8_t)pred[r];
      }
      if constexpr (QM == 2) {
#pragma unroll
        for (int r = 0; r < H; r++) ppk[r >> 2] |= (uint32_t)pred[r] << (8 * (r & 3));
      }
      if constexpr (SRC_LDS) {
#pragma unroll
        for (int r = 0; r < H; r++) {
          const uint32_t sp = src_l[r * SRC_ROW];
          sad_acc = sad_u32(sp, (uint32_t)pred[r], sad_acc);   // |source - prediction| summed in one op
          v[r] = (T)sp - pred[r];
        }
      } else {
#pragma unroll
        for (int r = 0; r < H; r++) v[r] -= pred[r];
      }
    }
  } else {
    int32_t pred[H];
    if (col_live) {
      if (qa.pred_in) {
        const uint16_t *pi = (const uint16_t *)qa.pred_in + (size_t)cand_ld * W * H + c;
#pragma unroll
        for (int r = 0; r < H; r++) pred[r] = pi[(size_t)r * W];
      } else {
        mc16_column_t<W, H, WS, false>(win, c, tp, BD, pred, six);
      }
      if (pred_out && live_st) {
        uint16_t *pp = (uint16_t *)pred_out + (size_t)cand * W * H + c;
#pragma unroll
        for (int r = 0; r < H; r++) pp[(size_t)r * W] = (uint16_t)pred[r];
      }
      if constexpr (QM == 2) {
#pragma unroll
        for (int r = 0; r < H; r++) ppk[r >> 1] |= (uint32_t)pred[r] << (16 * (r & 1));
      }
    }
    if constexpr (SRC_LATE) {
      __syncthreads();   // every lane has filtered its column: the window is dead
      stage_source();
      __syncthreads();
    }
    if (col_live) {
      if constexpr (SRC_LDS) {
#pragma unroll
        for (int r = 0; r < H; r++) {
          const uint32_t sp = *(const uint16_t *)(src_l + r * SRC_ROW);
          sad_acc = sad_u32(sp, (uint32_t)pred[r], sad_acc);
          v[r] = (T)sp - pred[r];
        }
      } else {
#pragma unroll
        for (int r = 0; r < H; r++) v[r] -= pred[r];
      }
    }
  }
  R1_PROF(1);   // B1: motion compensation + residual
  if (sad_out) {
    uint32_t sad = sad_acc;
    if constexpr (!SRC_LDS) {
#pragma unroll
      for (int r = 0; r < H; r++) sad += (uint32_t)iabs32(v[r]);
    }
    const uint32_t s = group_sum<P>(sad);
    // (non-temporal here too was tried: no difference, gpurun_out/r04_ab4 -- 8 bytes per candidate)
    if (live_st && c == 0) (sad_out + (size_t)wg * NC)[cl] = s;
  }
  if (satd_out) {
    const uint32_t s = group_sum<P>(satd_column<TS, H, BD>(v, lane));
    constexpr int LN = TS == 4 ? 2 : 3;
    if (live_st && c == 0) (satd_out + (size_t)wg * NC)[cl] = (s + ((1u << LN) >> 1)) >> LN;
  }
  R1_PROF(2);   // B2: SAD + SATD
  if (!QUANT && !coeffs) return;   // wave-uniform: kernel argument

  // ---- MT: the transform-type fan-out (rdo_tx_type_decision, src/rdo.rs:1701-1817).  The reference runs
  // motion_compensate + write_tx_tree + compute_distortion once per type of RAV1E_TX_TYPES
  // (src/transform/mod.rs:28-44) that the block's tx set allows, on the SAME prediction; here phases A / B ran
  // once and C .. H loop over the set bits of the launch's mask (a kernel argument: wave-uniform, every
  // 1-D kernel switch below is a scalar branch).  What an iteration needs again is the residual column: where
  // the source block stays in LDS (SRC_KEEP) it is re-formed from there and the packed prediction the
  // reconstruction keeps anyway -- no register is live across the loop for it; elsewhere a register copy.
  constexpr bool MT_RECOMP = MT && SRC_KEEP;
  T vkeep[MT && !MT_RECOMP ? H : 1];
  if constexpr (MT && !MT_RECOMP) {
#pragma unroll
    for (int r = 0; r < H; r++) vkeep[r] = v[r];
  }
  // TAIL_DEFER (type search of an 8x8 block under cdef_dist: one 8x8 kernel per candidate, eight lanes per
  // candidate, at most seven types): see the end of the kernel
#ifndef R1_MT_TAIL_DEFER
#define R1_MT_TAIL_DEFER 1
#endif
  constexpr bool TAIL_DEFER = R1_MT_TAIL_DEFER && MT && QM == 2 && W == 8 && H == 8;
  // (a mask of more than eight types -- the full AV1 inter set has sixteen -- keeps its tails inside the loop)
  const bool tail_defer = TAIL_DEFER && qa.dist_kind == R1_DIST_CDEF && qa.nt <= 8;   // wave-uniform
  uint32_t tail_keep[5] = {0, 0, 0, 0, 0};
  // The loop: groups of types that share the column pass (same vertical 1-D kernel and the same flips; without
  // COLSHARE every type is a group of its own), and inside a group the types in ascending order.  The result slot of
  // a type is its rank in the launch's mask, whatever order the groups come in.
  uint32_t rem = MT ? qa.tx_mask : 1u;
  bool fresh = true;   // v still holds the residual as phase B left it
  do {
  int t0 = 0;
  uint32_t gmask = 1u;
  if constexpr (MT) {
    t0 = (int)__builtin_ctz(rem);
    gmask = 1u << t0;
    if constexpr (COLSHARE) {
      auto colkey = [](int t) { return r1tx::vtx_1d(t) | ((int)r1tx::ud_flip(t) << 4) | ((int)r1tx::lr_flip(t) << 5); };
      const int k0 = colkey(t0);
      for (uint32_t m = rem & (rem - 1); m != 0; m &= m - 1) {   // scalar: the mask is a kernel argument
        const int t = (int)__builtin_ctz(m);
        if (colkey(t) == k0) gmask |= 1u << t;
      }
    }
    rem &= ~gmask;
    if (!fresh) {   // wave-uniform
      if constexpr (MT_RECOMP) {
        if (col_live) {
#pragma unroll
          for (int r = 0; r < H; r++) {
            const T sp = BPP == 1 ? (T)src_l[r * SRC_ROW] : (T) * (const uint16_t *)(src_l + r * SRC_ROW);
            const T pr = BPP == 1 ? (T)((ppk[r >> 2] >> (8 * (r & 3))) & 0xFF)
                                  : (T)((ppk[r >> 1] >> (16 * (r & 1))) & 0xFFFF);
            v[r] = sp - pr;
          }
        }
      } else {
#pragma unroll
        for (int r = 0; r < H; r++) v[r] = vkeep[r];
      }
    }
    fresh = false;
  }
  // ---- C: column transform on the residual registers ----
  __syncthreads();  // every lane is done reading the window (MT: the previous type's last phase); LDS becomes buf
  const int tx_col = MT ? t0 : (QM != 0 && qa.tx_mask != 0 ? (int)__builtin_ctz(qa.tx_mask) : (int)cd.tx_type);
  const bool any_ud = __any(live && r1tx::ud_flip(tx_col));
  if (col_live) {
    if (any_ud) {   // wave-uniform: skipped when no candidate of the wave flips
      const bool ud = r1tx::ud_flip(tx_col);
#pragma unroll
      for (int r = 0; r < H / 2; r++) {
        const T t0 = v[r], t1 = v[H - 1 - r];
        v[r] = ud ? t1 : t0;
        v[H - 1 - r] = ud ? t0 : t1;
      }
    }
#pragma unroll
    for (int r = 0; r < H; r++) v[r] = r1tx::shift_fwd_ct<SH0>(v[r]);
    r1tx::fwd_1d_m24<H>(v, r1tx::vtx_1d(tx_col));
    if constexpr (!SPLIT_T) {
      const int cc = cl * W + (r1tx::lr_flip(tx_col) ? W - 1 - c : c);
#pragma unroll
      for (int r = 0; r < HU; r++)
        tkeep[r * LSTRIDE + cc] = (TB)r1tx::shift_fwd_ct<SH1>(v[r]);
    }
  }
  if constexpr (!SPLIT_T) __syncthreads();
  R1_PROF(3);   // C: column transform, transpose written
  do {   // the types of the group: rows from the shared tile, then everything that depends on the type
  const int tx_type = MT ? (int)__builtin_ctz(gmask) : tx_col;
  const int slot = MT ? (int)__builtin_popcount(qa.tx_mask & ((1u << tx_type) - 1u)) : 0;
  // result slot of (candidate, type)
  const long long oslot = MT ? cand * (long long)qa.nt + slot
                             : (QM != 0 && qa.nt != 0 ? cand * (long long)qa.nt + qa.slot : cand);
  // ---- D: row transform, transposed store ----
  // P lanes per candidate again: the lane that filtered column c of candidate cl now owns row c
  // of the same candidate -- its descriptor is still in registers
  const int cl2 = cl, r = c;
  const bool live2 = live;
  const bool row_live = live2 && r < HU;
  const int tt = tx_type;
  constexpr int OS = H < 32 ? H : 32, WC = W < 32 ? W : 32;
  T u[W];
  if constexpr (SPLIT_T) {
    // rows 0..31 travel first and are picked up by lanes 0..31, then rows 32..63 through
    // the same bytes for lanes 32..63 (one candidate per wave here: cl = cl2 = 0)
    const int cc = r1tx::lr_flip(tx_type) ? W - 1 - c : c;
#pragma unroll
    for (int half = 0; half < HU / 32; half++) {
      if (col_live) {
#pragma unroll
        for (int rr = 0; rr < 32; rr++)
          tbuf[rr * LSTRIDE + cc] = (TB)r1tx::shift_fwd_ct<SH1>(v[half * 32 + rr]);
      }
      __syncthreads();
      if (row_live && (r >> 5) == half) {
#pragma unroll
        for (int k = 0; k < W; k++) u[k] = tbuf[(r & 31) * LSTRIDE + k];
      }
      __syncthreads();
    }
  }
  if (row_live) {
    if constexpr (!SPLIT_T) {
#pragma unroll
      for (int k = 0; k < W; k++) u[k] = tkeep[r * LSTRIDE + cl2 * W + k];
    }
    r1tx::fwd_1d_m24<W>(u, r1tx::htx_1d(tt));
#pragma unroll
    for (int k = 0; k < W; k++) {
      u[k] = r1tx::shift_fwd_ct<SH2>(u[k]);
      // `as T::Coeff` (forward.rs:157): the stores below truncate by themselves; only the
      // quantizer variants go on computing with the value
      if constexpr (QUANT) u[k] = (T)(CT)u[k];
    }
  }
  if constexpr (MT) {
    // the type search keeps its coefficients on the CU
  } else
  if constexpr (!R1_WIDE_STORE_POLICY(P)) {
    // large blocks: direct element stores (measured: the LDS detour costs more than the
    // 16-byte stores save at 32x32 and 64x64, profiles/r02_wide_store_ab.log)
    if (coeffs && row_live && live_st) {
      CT *dst = coeffs + (size_t)wg * (NC * W * H) + (cl2 * (W * H) + (r >= 32 ? OS * WC : 0) + (r & 31));
#pragma unroll
      for (int cg = 0; cg < W; cg += 32)
#pragma unroll
        for (int k = 0; k < WC; k++) {
#if R1_NT_STORE
          __builtin_nontemporal_store((CT)u[k + cg], &dst[H * cg + k * OS]);
#else
          dst[H * cg + k * OS] = (CT)u[k + cg];
#endif
        }
    }
  } else
  if (coeffs) {   // wave-uniform: kernel argument
    // The reference's transposed coefficient order (forward.rs:135-159) puts the rows r of a
    // column k next to each other: with lane = row a direct store is one 2- or 4-byte element
    // per lane per instruction -- W (64x64: 128) scattered store instructions per wave, and
    // this kernel runs at the texture-address unit's ~16 cycles per vector-memory instruction
    // (DESIGN.md 5.1).  So the block is assembled in LDS in its final order (the transpose
    // tile is dead: every lane holds its row) and leaves as 16-byte stores, W*sizeof(CT)/16
    // per wave.  64x64 32-bit coefficients (16 KB) go in two halves (k < 32, k >= 32), which
    // are contiguous halves of the output.
    constexpr int ESZ = (int)sizeof(CT);
    constexpr int NP = NC * W * H * ESZ > LDS_WORK ? 2 : 1;
    static_assert(NP == 1 || W == 64, "only the 64-wide blocks are split");
    static_assert(NC * W * H * ESZ / NP <= LDS_WORK, "a pass fits the LDS of the kernel");
    constexpr int EPP = W * H / NP;                 // elements of one candidate per pass
    constexpr int CBY = EPP * ESZ;                  // bytes of one candidate per pass
    constexpr int CH = CBY / P >= 16 ? 16 : CBY / P;   // bytes a lane moves per step
    constexpr int NCH = CBY / P / CH;
    static_assert(CH * NCH * P == CBY && (CH == 16 || CH == 8 || CH == 4), "whole chunks");
    // candidates one row of P elements apart: with the bare stride (a multiple of 32 dwords for
    // 8x8 / 16x16) the NC candidates of a lane group hit the same banks with their element
    // writes (4-way at 8x8: SQ_LDS_BANK_CONFLICT 4.1 M -> 20.7 M per launch when this path came in)
    constexpr int TPAD = NC > 1 ? ((P * ESZ + 15) & ~15) / ESZ : 0;
    static_assert(NC * (EPP + TPAD) * ESZ <= LDS_WORK, "the padded tiles fit the LDS of the kernel");
    static_assert(((EPP + TPAD) * ESZ) % 16 == 0, "16-byte reads stay aligned");
    CT *tile = (CT *)smem + cl2 * (EPP + TPAD);
    uint8_t *gdst = (uint8_t *)(coeffs + (size_t)wg * (NC * W * H)) + cl2 * (W * H * ESZ);
#pragma unroll
    for (int p = 0; p < NP; p++) {
      __syncthreads();   // rows are in registers (pass 0) / the previous half has been copied out
      if (row_live) {
        constexpr int KP = W / NP;
#pragma unroll
        for (int k = p * KP; k < (p + 1) * KP; k++) {
          const int e = (r >= 32 ? OS * WC : 0) + (r & 31) + H * (k & ~31) + (k & 31) * OS - p * EPP;
          tile[e] = (CT)u[k];
        }
      }
      __syncthreads();
      if (live_st) {
        const uint8_t *src = (const uint8_t *)tile + r * CH;
        uint8_t *dst = gdst + p * CBY + r * CH;
#pragma unroll
        for (int j = 0; j < NCH; j++) {
#if R1_NT_STORE
          // The coefficients are not read again by this launch, and a step writes 0.26 GB (8-bit) / 0.53 GB
          // (10-bit) of them per ladder size: written through the L2 as ordinary stores they evict the window
          // rows the K candidates of a block share.  Non-temporal stores (same-box A/B, gpurun_out/r04_ab3):
          // 8-bit 8x8 launch 0.226 -> 0.206 ms, 10-bit 8x8 0.308 -> 0.232, 10-bit 16x16 0.252 -> 0.221; step
          // +3.3 % / +11 %.
          if constexpr (CH == 16) {
            typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
            __builtin_nontemporal_store(*(const u32x4_t *)(src + j * P * CH), (u32x4_t *)(dst + j * P * CH));
          } else if constexpr (CH == 8) {
            typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
            __builtin_nontemporal_store(*(const u32x2_t *)(src + j * P * CH), (u32x2_t *)(dst + j * P * CH));
          } else {
            __builtin_nontemporal_store(*(const uint32_t *)(src + j * P * CH), (uint32_t *)(dst + j * P * CH));
          }
#else
          if constexpr (CH == 16) *(uint4 *)(dst + j * P * CH) = *(const uint4 *)(src + j * P * CH);
          else if constexpr (CH == 8) *(uint2 *)(dst + j * P * CH) = *(const uint2 *)(src + j * P * CH);
          else *(uint32_t *)(dst + j * P * CH) = *(const uint32_t *)(src + j * P * CH);
#endif
        }
      }
    }
  }
  R1_PROF(4);   // D: row transform + stores issued
  if constexpr (QUANT) {
    // ---- E: quantizer on the coded area, in LDS (aliases the transpose tile:
    // every row lane has its row in registers by now) ----
    constexpr int CODED = OS * WC;
    constexpr int PL = WL > HL ? WL : HL;          // log2(P)
    constexpr int NPLQ = CODED / P;
    static_assert(CODED % P == 0 && NPLQ >= 1, "P lanes share the coded area");
    __syncthreads();
    int32_t *tile = (int32_t *)smem + cl2 * QT_STRIDE;
    unsigned long long tail = 0;
    if (row_live) {
#pragma unroll
      for (int k = 0; k < W; k++) {
        if (r < 32 && k < 32) {
          tile[k * OS + r] = u[k];
        } else {   // beyond the coded area: rcoeff = 0 (encoder.rs:1628-1634)
          tail += (unsigned long long)(long long)(int32_t)((uint32_t)u[k] * (uint32_t)u[k]);
        }
      }
    }
    __syncthreads();
    const int kind = tt < 10 ? 0 : ((tt & 1) ? 2 : 1);
    int eob = 0;
    unsigned long long dist = 0;
    // log_tx_scale follows from the block size (quantize/mod.rs:get_log_tx_scale): a constant here
    constexpr int LTS = (W * H > 256) + (W * H > 1024);
#ifndef R1_QUANT_MID
#define R1_QUANT_MID 0   // the 24-bit quantizer for i32 coefficients: exact (tests/test_tx_range.py, GPU-tested at the range
                         // limits) and no faster -- 10-bit pixel chain 283-285 k -> 282 k Mpx/s, the 32x32 launch +5 % slower
                         // (profiles/r06_ab_notes.md, ab4).  Off: the round-5 arithmetic stays the product path.
#endif
    // i32 coefficients here come from a pixel residual: |c << lts| <= 2^21 (quant_common.hpp, QParams::ac_m22)
    constexpr bool QMID = R1_QUANT_MID && sizeof(CT) == 4;
    r1q::quantize_group<CT, PL, NPLQ, QM == 1, LTS, QMID>(tile, cl2 * P, r, live2, qa.scan[kind], qa.qp, tail,
                                                          eob, dist);
    if (live_st && r == 0) {
      qa.eob[oslot] = (uint16_t)eob;
      if constexpr (QM == 1) {
        qa.tx_dist[oslot] = dist;
        if (qa.est_rate) qa.est_rate[oslot] = r1q::estimate_rate(qa.q_bin, qa.tx_size, dist);
      }
    }
    if (qa.qcoeffs) {
      __syncthreads();
      if (live_st) {
        CT *qd = (CT *)qa.qcoeffs + oslot * CODED;
#pragma unroll
        for (int k = 0; k < NPLQ; k++) qd[k * P + r] = (CT)tile[k * P + r];
      }
    }
    if constexpr (QM == 2) {
      // ---- F: dequantize (mod.rs:372-383) + inverse row transform
      // (inverse_transform_add, src/transform/inverse.rs:1633-1705; k_inv_tx) ----
      constexpr int HC = OS;
      constexpr bool RECT1 = (WL > HL ? WL - HL : HL - WL) == 1;
      __syncthreads();
      const bool irow_live = live2 && r < HC;
      T w_[W];
      {
        const int range = BD + 8;
        const T hi = (T)((1 << (range - 1)) - 1), lo = -hi - 1;
        constexpr int32_t off = (1 << LTS) - 1;
        if (irow_live) {
#pragma unroll
          for (int k = 0; k < WC; k++) {
            const int32_t q = (int32_t)(CT)tile[k * OS + r];
            const uint32_t quant = (k == 0 && r == 0) ? qa.qp.dc_q : qa.qp.ac_q;
            const uint32_t prod = QMID ? (uint32_t)r1q::mul24_wrap(q, (int32_t)quant) : (uint32_t)q * quant;
            const T raw = (T)(CT)((int32_t)(prod + (uint32_t)((q >> 31) & off)) >> LTS);
            const T val = RECT1 ? ((T)((uint32_t)raw * 2896u + 2048u) >> 12) : raw;
            w_[k] = r1itx::clamp3(val, lo, hi);
          }
#pragma unroll
          for (int k = WC; k < W; k++) w_[k] = 0;
#ifndef R1_STUB_INVR
          r1itx::inv_1d<W, true>(w_, r1tx::htx_1d(tt), lo, hi);
#endif
        }
      }
      __syncthreads();   // every coefficient has been read: the tile becomes the row buffer
      if (irow_live) {
#pragma unroll
        for (int k = 0; k < W; k++) buf[r * ISTRIDE + cl2 * W + k] = w_[k];
      }
      __syncthreads();
      // ---- G: inverse column transform, reconstruction (lane = column again) ----
      T rc[H];
      if (col_live) {
        const int range = BD + 6 > 16 ? BD + 6 : 16;
        const T hi = (T)((1 << (range - 1)) - 1), lo = -hi - 1;
        const T pmax = (T)((1 << BD) - 1);
#pragma unroll
        for (int rr = 0; rr < HC; rr++) {
          const T x = buf[rr * ISTRIDE + cl * W + c];
          rc[rr] = r1itx::clamp3((x + ((1 << qa.inv_shift) >> 1)) >> qa.inv_shift, lo, hi);
        }
#pragma unroll
        for (int rr = HC; rr < H; rr++) rc[rr] = 0;
#ifndef R1_STUB_INVC
        r1itx::inv_1d<H, true>(rc, r1tx::vtx_1d(tx_type), lo, hi);
#endif
#pragma unroll
        for (int rr = 0; rr < H; rr++) {
          const T pr = BPP == 1 ? (T)((ppk[rr >> 2] >> (8 * (rr & 3))) & 0xFF)
                                : (T)((ppk[rr >> 1] >> (16 * (rr & 1))) & 0xFFFF);
          const T px = pr + ((rc[rr] + 8) >> 4);
          rc[rr] = px < 0 ? 0 : (px > pmax ? pmax : px);
        }
      }
      if (col_live && live_st && qa.rec) {
        if constexpr (BPP == 1) {
          uint8_t *d = (uint8_t *)qa.rec + (size_t)oslot * W * H + c;
#pragma unroll
          for (int rr = 0; rr < H; rr++) d[(size_t)rr * W] = (uint8_t)rc[rr];
        } else {
          uint16_t *d = (uint16_t *)qa.rec + (size_t)oslot * W * H + c;
#pragma unroll
          for (int rr = 0; rr < H; rr++) d[(size_t)rr * W] = (uint16_t)rc[rr];
        }
      }
      unsigned long long acc = 0;
#ifdef R1_STUB_DIST   /* timing experiments only (tools/gpu_r4_h.sh): results are wrong */
      acc = (unsigned long long)(uint32_t)rc[0] + (uint32_t)rc[H - 1];
#else
      constexpr bool COL_DIST = H <= 16;
      if constexpr (COL_DIST) {
        // ---- H (blocks up to 32 rows): sse_wxh / cdef_dist_wxh with lane = column.  The
        // reconstruction column is still in registers, the source column is read
        // again; a tile's 8 (4) lanes meet by xor-shuffles, its first lane runs the
        // fixed-point tail (dist_common.hpp).  With one lane per 8x8 tile (the H = 64
        // path below) an 8x8 candidate keeps 8 of the 64 lanes busy for 64 pixels each.
        constexpr int KW = W < 8 ? W : 8, KH = H < 8 ? H : 8;
        const uint8_t *po = px_addr<BPP>(org, cd.ox + (col_live ? c : 0), cd.oy);
        const size_t so = (size_t)org.stride * BPP;
        if (qa.dist_kind == R1_DIST_CDEF) {
          // the five sums of every tile row first, ONE fixed-point tail afterwards: after the xor-shuffles
          // all KW lanes of a tile hold its sums, so lane j of the group takes tile row j (16-row blocks have
          // two) -- the tail (ssim boost in 64-bit arithmetic) used to run once per tile row with one lane
          // of the group alive
          constexpr int NR = H / KH;
          static_assert(NR <= KW, "a tile group has a lane for every tile row");
          uint32_t S[NR][5];
#pragma unroll
          for (int t = 0; t < NR; t++) {
            const int y0 = t * KH;
            uint32_t sum_s = 0, sum_d = 0, sum_s2 = 0, sum_d2 = 0, sum_sd = 0;
            if (col_live) {
#pragma unroll
              for (int rr = 0; rr < KH; rr++) {
                uint32_t sv;
                if constexpr (SRC_KEEP) sv = BPP == 1 ? (uint32_t)src_l[(y0 + rr) * SRC_ROW]
                                                      : (uint32_t) * (const uint16_t *)(src_l + (y0 + rr) * SRC_ROW);
                else sv = (uint32_t)ld_px<BPP>(po + (y0 + rr) * so);
                const uint32_t dv = (uint32_t)rc[y0 + rr];
                sum_s += sv; sum_d += dv;
                sum_s2 += sv * sv; sum_d2 += dv * dv; sum_sd += sv * dv;
              }
            }
#pragma unroll
            for (int m = 1; m < KW; m <<= 1) {
              sum_s += __shfl_xor(sum_s, m, 64); sum_d += __shfl_xor(sum_d, m, 64);
              sum_s2 += __shfl_xor(sum_s2, m, 64); sum_d2 += __shfl_xor(sum_d2, m, 64);
              sum_sd += __shfl_xor(sum_sd, m, 64);
            }
            S[t][0] = sum_s; S[t][1] = sum_d; S[t][2] = sum_s2; S[t][3] = sum_d2; S[t][4] = sum_sd;
          }
          const int j = c & (KW - 1);
          uint32_t P5[5];
#pragma unroll
          for (int q5 = 0; q5 < 5; q5++) {
            P5[q5] = S[0][q5];
#pragma unroll
            for (int t = 1; t < NR; t++) P5[q5] = j == t ? S[t][q5] : P5[q5];
          }
          if (TAIL_DEFER && tail_defer) {
            // type search, 8x8: lane `slot` of the candidate's eight keeps this type's five sums; the tails run
            // once, behind the loop
            if (j == slot) {
#pragma unroll
              for (int q5 = 0; q5 < 5; q5++) tail_keep[q5] = P5[q5];
            }
          } else if (col_live && j < NR)
            acc += r1dist::cdef_tile_tail<BD>(P5[0], P5[1], P5[2], P5[3], P5[4], KW * KH, cd.ox + c - j,
                                              cd.oy + j * KH, qa.scales, qa.scale_stride, BD);
        } else {
#pragma unroll
          for (int y0 = 0; y0 < H; y0 += 4) {
            uint32_t cell = 0;
            if (col_live) {
#pragma unroll
              for (int rr = 0; rr < 4; rr++) {
                int32_t sv;
                if constexpr (SRC_KEEP) sv = BPP == 1 ? (int32_t)src_l[(y0 + rr) * SRC_ROW]
                                                      : (int32_t) * (const uint16_t *)(src_l + (y0 + rr) * SRC_ROW);
                else sv = ld_px<BPP>(po + (y0 + rr) * so);
                const int32_t d = sv - (int32_t)rc[y0 + rr];
                cell += (uint32_t)(d * d);
              }
            }
            cell += __shfl_xor(cell, 1, 64);
            cell += __shfl_xor(cell, 2, 64);
            if (col_live && (c & 3) == 0) {
              const int lx = (cd.ox + c) << qa.xdec, ly = (cd.oy + y0) << qa.ydec;
              const uint32_t sc =
                  qa.scales ? qa.scales[(size_t)(ly >> 3) * qa.scale_stride + (lx >> 3)] : (1u << 14);
              acc += ((unsigned long long)cell * sc + 128) >> 8;
            }
          }
        }
      } else {
      __syncthreads();   // the row buffer has been read: LDS becomes the reconstruction
      uint8_t *rec_l = smem + cl * (W * H * BPP);
      if (col_live) {
#pragma unroll
        for (int rr = 0; rr < H; rr++) {
          if constexpr (BPP == 1) rec_l[rr * W + c] = (uint8_t)rc[rr];
          else ((uint16_t *)rec_l)[rr * W + c] = (uint16_t)rc[rr];
        }
      }
      __syncthreads();
      // ---- H (64-row blocks): one lane per 8x8 tile (dist_common.hpp) ----
      constexpr int TW8 = (W + 7) / 8, NT8 = TW8 * ((H + 7) / 8);
      static_assert(NT8 <= P, "a candidate's lanes cover its 8x8 tiles");
      if (live && c < NT8) {
        const int x0 = (c % TW8) * 8, y0 = (c / TW8) * 8;
        const int kw = W - x0 < 8 ? W - x0 : 8, kh = H - y0 < 8 ? H - y0 : 8;
        const uint8_t *po = px_addr<BPP>(org, cd.ox + x0, cd.oy + y0);
        const uint8_t *pr = rec_l + (y0 * W + x0) * BPP;
        if (qa.dist_kind == R1_DIST_WSSE)
          acc = r1dist::tile_scaled_dist<BPP, 2>(po, (size_t)org.stride * BPP, pr, (size_t)W * BPP, kw, kh,
                                                 cd.ox + x0, cd.oy + y0, qa.scales, qa.scale_stride,
                                                 qa.xdec, qa.ydec, BD);
        else
          acc = r1dist::tile_scaled_dist<BPP, 3>(po, (size_t)org.stride * BPP, pr, (size_t)W * BPP, kw, kh,
                                                 cd.ox + x0, cd.oy + y0, qa.scales, qa.scale_stride,
                                                 qa.xdec, qa.ydec, BD);
      }
      }
#endif
      if (!(TAIL_DEFER && tail_defer)) {   // wave-uniform
#pragma unroll
        for (int m = 1; m < P; m <<= 1) {
          const uint32_t lo = __shfl_xor((uint32_t)acc, m, 64);
          const uint32_t hi = __shfl_xor((uint32_t)(acc >> 32), m, 64);
          acc += ((unsigned long long)hi << 32) | lo;
        }
        if (live_st && c == 0) qa.pix_dist[oslot] = qa.dist_kind == R1_DIST_WSSE ? (acc + 32) / 64 : acc;
      }
    }
  }
  if constexpr (!MT) break;
  gmask &= gmask - 1;
  } while (gmask != 0);
  if constexpr (!MT) break;
  } while (rem != 0);
  if constexpr (TAIL_DEFER) {
    // the fixed-point tails of cdef_dist_kernel (ssim boost, 64-bit arithmetic, ~120 instructions): inside the loop
    // they ran once per type with ONE lane of a candidate's eight alive; here lane j runs the tail of type j --
    // one pass for all (up to seven) types of the candidate
    if (tail_defer && col_live && c < qa.nt) {
      const unsigned long long d = r1dist::cdef_tile_tail<BD>(tail_keep[0], tail_keep[1], tail_keep[2], tail_keep[3],
                                                              tail_keep[4], 64, cd.ox, cd.oy, qa.scales, qa.scale_stride, BD);
      if (live_st) qa.pix_dist[cand * (long long)qa.nt + c] = d;
    }
  }
}

#ifdef R1_RDO_DISPATCH_TU
// put_8tap / prep_8tap alone on the same machinery (blocks whose size is a
// transform size): window staged with one round trip, dot4 / dot2 columns.
template <int BPP, int WL, int HL, bool PREP>
__global__ __launch_bounds__(64) void k_mc_fast(R1Plane ref, const R1McCand *__restrict__ cands,
                                                int n, void *__restrict__ dst) {
  constexpr int W = 1 << WL, H = 1 << HL;
  constexpr int P = W > H ? W : H, NC = 64 / P;
  constexpr int WS = (((W + 7) * BPP + 3) >> 2) << 2;
  __shared__ __attribute__((aligned(16))) uint8_t smem[NC * (H + 7) * WS];
  const int lane = threadIdx.x;
  const int cl = lane / P, c = lane % P;
  // XCD-aware like k_rdo_cand: XCD x takes the x-th contiguous eighth of the list (grid = multiple of 8)
  const unsigned wg = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const long long cand = (long long)wg * NC + cl;
  const bool live = cand < n;
  R1McCand cd = {};
  if (live) cd = cands[cand];
  uint8_t *win = smem + cl * (H + 7) * WS;
  if (live)
    r1mc::stage_window_fast<BPP, BPP == 1 ? 0x80808080u : 0u, W, H, P>(win, WS, ref, cd.rx, cd.ry, c);
  __syncthreads();
  const bool any_cf0 = __any(live && cd.col_frac == 0);
  if (!(live && c < W)) return;
  int32_t pred[H];
  if constexpr (BPP == 1)
    mc8_column<W, H, WS, PREP>(win, c, cd.col_frac, cd.row_frac, cd.mode_x, cd.mode_y, any_cf0, pred);
  else
    mc16_column<W, H, WS, PREP>(win, c, cd.col_frac, cd.row_frac, cd.mode_x, cd.mode_y,
                                ref.bit_depth, pred);
  // the predictions stream out (non-temporal: they would only push the reference rows out of the L2)
  if constexpr (PREP || BPP == 2) {
    uint16_t *pp = (uint16_t *)dst + (size_t)cand * W * H + c;
#pragma unroll
    for (int r = 0; r < H; r++) __builtin_nontemporal_store((uint16_t)pred[r], &pp[(size_t)r * W]);
  } else {
    uint8_t *pp = (uint8_t *)dst + (size_t)cand * W * H + c;
#pragma unroll
    for (int r = 0; r < H; r++) __builtin_nontemporal_store((uint8_t)pred[r], &pp[(size_t)r * W]);
  }
}

template <int BPP, int WL, int HL>
int launch_mc_fast(bool prep, const R1Plane &ref, const R1McCand *cands, int n, void *dst,
                   hipStream_t st) {
  constexpr int W = 1 << WL, H = 1 << HL, P = W > H ? W : H, NC = 64 / P;
  const unsigned grid = ((unsigned)((n + NC - 1) / NC) + 7u) & ~7u;
  if (prep)
    hipLaunchKernelGGL((k_mc_fast<BPP, WL, HL, true>), dim3(grid), dim3(64), 0, st, ref, cands, n, dst);
  else
    hipLaunchKernelGGL((k_mc_fast<BPP, WL, HL, false>), dim3(grid), dim3(64), 0, st, ref, cands, n, dst);
  R1_HIP_CHECK(hipGetLastError());
  return R1_OK;
}

#endif   // R1_RDO_DISPATCH_TU

#ifdef R1_RDO_SLICE_TU
template <int BD, int WL, int HL, int QM, bool MT>
int launch(const R1Plane &org, const R1Plane &ref, const R1RdoCand *cands, int n,
           uint32_t *sad, uint32_t *satd, void *coeffs, void *pred, const RdoQuantArgs *qa,
           hipStream_t st) {
  constexpr int W = 1 << WL, H = 1 << HL, P = W > H ? W : H, NC = 64 / P;
  typedef typename std::conditional<BD == 8, int16_t, int32_t>::type CT;
  const unsigned groups = (unsigned)((n + NC - 1) / NC);
#if R1_XCD_REMAP
  const unsigned grid = (groups + 7u) & ~7u;     // whole rounds over the 8 XCDs (see the kernel's `wg`)
#else
  const unsigned grid = groups;
#endif
  hipLaunchKernelGGL((k_rdo_cand<BD, WL, HL, CT, QM, MT>), dim3(grid), dim3(64), 0, st,
                     org, ref, cands, n, sad, satd, (CT *)coeffs, pred, qa ? *qa : RdoQuantArgs{});
  R1_HIP_CHECK(hipGetLastError());
  return R1_OK;
}

// one (bit depth, QM) slice: tx_size -> instantiation
template <int BD, int QM, bool MT>
int slice(int tx_size, const R1Plane &org, const R1Plane &ref, const R1RdoCand *cands, int n,
          uint32_t *sad, uint32_t *satd, void *coeffs, void *pred, const RdoQuantArgs *qa,
          hipStream_t st) {
  // R1_RDO_TU_TSMASK: the transform sizes this unit instantiates (the Makefile cuts the slow QM = 2
  // slices into parts by size)
#ifndef R1_RDO_TU_TSMASK
#define R1_RDO_TU_TSMASK 0x7ffff
#endif
  // the type-search slices: sizes up to 16 x 16 (ids 0-2, 5-8, 13, 14).  A 64-point side has TX_SET_DCTONLY
  // (get_tx_set, src/context/transform_unit.rs:123-131) and a 32-point side DCT_DCT (+ IDTX for inter blocks):
  // one or two types, which the plain kernel evaluates at twice the occupancy (same-box A/B,
  // profiles/r05_ab_notes.md: the 32x32 fan-out kernel held 2 waves per SIMD and LOST 13-22 % against two launches)
  constexpr unsigned TSM = MT ? ((R1_RDO_TU_TSMASK) & 0x61E7u) : (unsigned)(R1_RDO_TU_TSMASK);
#define R1_RC_CASE(ID, WL, HL)                                                                   \
  case ID:                                                                                       \
    if constexpr ((TSM >> ID) & 1)                                                               \
      return launch<BD, WL, HL, QM, MT>(org, ref, cands, n, sad, satd, coeffs, pred, qa, st);    \
    else                                                                                         \
      break;
  switch (tx_size) {
#ifdef R1_HEADLINE_ONLY   // experiment builds (tools/build_variant.sh): the headline instantiations only
    R1_RC_CASE(1, 3, 3) R1_RC_CASE(2, 4, 4) R1_RC_CASE(3, 5, 5) R1_RC_CASE(4, 6, 6)
#else
    R1_RC_CASE(0, 2, 2) R1_RC_CASE(1, 3, 3) R1_RC_CASE(2, 4, 4)
    R1_RC_CASE(3, 5, 5) R1_RC_CASE(4, 6, 6) R1_RC_CASE(5, 2, 3)
    R1_RC_CASE(6, 3, 2) R1_RC_CASE(7, 3, 4) R1_RC_CASE(8, 4, 3)
    R1_RC_CASE(9, 4, 5) R1_RC_CASE(10, 5, 4) R1_RC_CASE(11, 5, 6)
    R1_RC_CASE(12, 6, 5) R1_RC_CASE(13, 2, 4) R1_RC_CASE(14, 4, 2)
    R1_RC_CASE(15, 3, 5) R1_RC_CASE(16, 5, 3) R1_RC_CASE(17, 4, 6)
    R1_RC_CASE(18, 6, 4)
#endif
  }
#undef R1_RC_CASE
  return R1_EINVAL;
}
#endif

}  // namespace

#ifdef R1_PHASE_PROF
extern "C" int r1_debug_phase_prof(unsigned long long *out, int reset) {   /* out[4096][8] */
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phase), 4096 * 64) != hipSuccess) return -1;
  if (reset) {
    void *p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_phase)) != hipSuccess) return -1;
    if (hipMemset(p, 0, 4096 * 64) != hipSuccess) return -1;
  }
  return 0;
}
#endif

#define R1_SLICE_ARGS                                                                         \
  int tx_size, const R1Plane &org, const R1Plane &ref, const R1RdoCand *cands, int n,         \
      uint32_t *sad, uint32_t *satd, void *coeffs, void *pred, const RdoQuantArgs *qa, hipStream_t st
#define R1_SLICE_NAME2(B, Q) r1_rdo_slice_b##B##_q##Q
#define R1_SLICE_NAME(B, Q) R1_SLICE_NAME2(B, Q)
#if defined(R1_RDO_TU_BD) && !defined(R1_HEADLINE_ONLY)
// slice numbers 0..2 = QM; 3 / 4 = the type-search (MT) instantiations of QM 1 / 2
int R1_SLICE_NAME(R1_RDO_TU_BD, R1_RDO_TU_QM)(R1_SLICE_ARGS) {
  return slice<R1_RDO_TU_BD, (R1_RDO_TU_QM >= 3 ? R1_RDO_TU_QM - 2 : R1_RDO_TU_QM), (R1_RDO_TU_QM >= 3)>(
      tx_size, org, ref, cands, n, sad, satd, coeffs, pred, qa, st);
}
#endif

#ifdef R1_RDO_DISPATCH_TU
#ifndef R1_HEADLINE_ONLY
int r1_rdo_slice_b8_q0(R1_SLICE_ARGS);  int r1_rdo_slice_b8_q1(R1_SLICE_ARGS);  int r1_rdo_slice_b8_q2(R1_SLICE_ARGS);
int r1_rdo_slice_b10_q0(R1_SLICE_ARGS); int r1_rdo_slice_b10_q1(R1_SLICE_ARGS); int r1_rdo_slice_b10_q2(R1_SLICE_ARGS);
int r1_rdo_slice_b12_q0(R1_SLICE_ARGS); int r1_rdo_slice_b12_q1(R1_SLICE_ARGS); int r1_rdo_slice_b12_q2(R1_SLICE_ARGS);
int r1_rdo_slice_b8_q3(R1_SLICE_ARGS);  int r1_rdo_slice_b8_q4(R1_SLICE_ARGS);
int r1_rdo_slice_b10_q3(R1_SLICE_ARGS); int r1_rdo_slice_b10_q4(R1_SLICE_ARGS);
int r1_rdo_slice_b12_q3(R1_SLICE_ARGS); int r1_rdo_slice_b12_q4(R1_SLICE_ARGS);
#endif
// Used by r1_mc_put_batch / r1_mc_prep_batch (mc.hip) for block sizes that are
// transform sizes; returns 1 when (w, h) is not one of them.
int r1_mc_fast_launch(bool prep, const R1Plane *ref, int w, int h, const R1McCand *cands, int n,
                      void *dst, hipStream_t st) {
  int ts = -1;
  for (int t = 0; t < 19; t++)
    if ((1 << r1tx::kTxWLog2[t]) == w && (1 << r1tx::kTxHLog2[t]) == h) ts = t;
  if (ts < 0) return 1;
#ifdef R1_HEADLINE_ONLY
  return 1;
#else
#define R1_MF_CASE(ID, WL, HL)                                                         \
  case ID:                                                                             \
    return ref->bytes_per_px == 1 ? launch_mc_fast<1, WL, HL>(prep, *ref, cands, n, dst, st) \
                                  : launch_mc_fast<2, WL, HL>(prep, *ref, cands, n, dst, st);
  switch (ts) {
    R1_MF_CASE(0, 2, 2) R1_MF_CASE(1, 3, 3) R1_MF_CASE(2, 4, 4)
    R1_MF_CASE(3, 5, 5) R1_MF_CASE(4, 6, 6) R1_MF_CASE(5, 2, 3)
    R1_MF_CASE(6, 3, 2) R1_MF_CASE(7, 3, 4) R1_MF_CASE(8, 4, 3)
    R1_MF_CASE(9, 4, 5) R1_MF_CASE(10, 5, 4) R1_MF_CASE(11, 5, 6)
    R1_MF_CASE(12, 6, 5) R1_MF_CASE(13, 2, 4) R1_MF_CASE(14, 4, 2)
    R1_MF_CASE(15, 3, 5) R1_MF_CASE(16, 5, 3) R1_MF_CASE(17, 4, 6)
    R1_MF_CASE(18, 6, 4)
  }
#undef R1_MF_CASE
  return 1;
#endif
}

namespace {
int rdo_dispatch(r1_ctx *ctx, const R1Plane *org, const R1Plane *ref, int w, int h, int tx_size,
                 const R1RdoCand *cands, int n, uint32_t *sad_out, uint32_t *satd_out,
                 void *coeffs, void *pred_out, const RdoQuantArgs *qa, void *stream, bool mt = false) {
  const bool from_pred = qa && qa->pred_in;
  R1_REQUIRE(ctx && org && (ref || from_pred));
  const R1Plane no_ref = {};
  if (from_pred) ref = &no_ref;
  R1_REQUIRE(from_pred || org->bytes_per_px == ref->bytes_per_px);
  R1_REQUIRE(org->bytes_per_px == 1 || org->bytes_per_px == 2);
  R1_REQUIRE(from_pred || org->bit_depth == ref->bit_depth);
  R1_REQUIRE((org->bytes_per_px == 1) == (org->bit_depth == 8));
  R1_REQUIRE(tx_size >= 0 && tx_size < 19);
  R1_REQUIRE((1 << r1tx::kTxWLog2[tx_size]) == w &&
             (1 << r1tx::kTxHLog2[tx_size]) == h);
  if (n <= 0) return R1_OK;
  R1_REQUIRE(cands);
  hipStream_t st = (hipStream_t)stream;
  // the per-candidate tx_type selects the 1-D kernels on the device; the
  // shifts depend only on (tx_size, bit depth) for every non-WHT type and are
  // compile-time constants of the instantiation
  const int bd = org->bit_depth;
  R1_REQUIRE(bd == 8 || bd == 10 || bd == 12);
  const int qm = !qa ? 0 : (qa->pix_dist ? 2 : 1);
#ifdef R1_HEADLINE_ONLY
  if (qm != 0 || bd == 12) return R1_EINVAL;
  return bd == 8 ? slice<8, 0, false>(tx_size, *org, *ref, cands, n, sad_out, satd_out, coeffs, pred_out, nullptr, st)
                 : slice<10, 0, false>(tx_size, *org, *ref, cands, n, sad_out, satd_out, coeffs, pred_out, nullptr, st);
#else
  typedef int (*SliceFn)(R1_SLICE_ARGS);
  static const SliceFn kSlices[3][5] = {
      {r1_rdo_slice_b8_q0, r1_rdo_slice_b8_q1, r1_rdo_slice_b8_q2, r1_rdo_slice_b8_q3, r1_rdo_slice_b8_q4},
      {r1_rdo_slice_b10_q0, r1_rdo_slice_b10_q1, r1_rdo_slice_b10_q2, r1_rdo_slice_b10_q3, r1_rdo_slice_b10_q4},
      {r1_rdo_slice_b12_q0, r1_rdo_slice_b12_q1, r1_rdo_slice_b12_q2, r1_rdo_slice_b12_q3, r1_rdo_slice_b12_q4}};
  R1_REQUIRE(!mt || (qa && qa->tx_mask != 0 && qm != 0 && !coeffs));
  return kSlices[(bd - 8) / 2][mt ? qm + 2 : qm](tx_size, *org, *ref, cands, n, sad_out, satd_out, coeffs, pred_out,
                                                 qa, st);
#endif
}
}  // namespace

extern "C" int r1_rdo_cand_batch(r1_ctx *ctx, const R1Plane *org,
                                 const R1Plane *ref, int w, int h, int tx_size,
                                 const R1RdoCand *cands, int n,
                                 uint32_t *sad_out, uint32_t *satd_out,
                                 void *coeffs, void *pred_out, void *stream) {
  return rdo_dispatch(ctx, org, ref, w, h, tx_size, cands, n, sad_out, satd_out, coeffs, pred_out,
                      nullptr, stream);
}

extern "C" int r1_rdo_full_cand_batch(r1_ctx *ctx, const R1Plane *org, const R1Plane *ref, int w,
                                      int h, int tx_size, const R1RdoCand *cands, int n,
                                      const R1QuantParams *params, uint32_t *sad_out,
                                      uint32_t *satd_out, uint16_t *eob_out,
                                      uint64_t *tx_dist_out, uint64_t *est_rate_out,
                                      void *qcoeffs_out, void *coeffs, void *stream) {
  R1_REQUIRE(ctx && org && params && eob_out && tx_dist_out);
  R1_REQUIRE(tx_size >= 0 && tx_size < 19);
  R1_REQUIRE(params->bit_depth == org->bit_depth);
  RdoQuantArgs qa = {};
  qa.qp = r1q::make_qparams(*params, tx_size, org->bytes_per_px == 1 ? 2 : 4);
  for (int k = 0; k < 3; k++) qa.scan[k] = ctx->scan_dev + ctx->scan_off[tx_size][k];
  qa.tx_size = tx_size;
  qa.q_bin = params->qindex / 32;   // RDO_QUANT_DIV
  qa.eob = eob_out;
  qa.tx_dist = (unsigned long long *)tx_dist_out;
  qa.est_rate = (unsigned long long *)est_rate_out;
  qa.qcoeffs = qcoeffs_out;
  return rdo_dispatch(ctx, org, ref, w, h, tx_size, cands, n, sad_out, satd_out, coeffs, nullptr,
                      &qa, stream);
}

extern "C" int r1_rdo_pixel_cand_batch(r1_ctx *ctx, const R1Plane *org, const R1Plane *ref, int w,
                                       int h, int tx_size, const R1RdoCand *cands, int n,
                                       const R1QuantParams *params, int dist_kind,
                                       const uint32_t *scales, int scale_stride, int xdec, int ydec,
                                       uint32_t *sad_out, uint32_t *satd_out, uint16_t *eob_out,
                                       uint64_t *dist_out, void *qcoeffs_out, void *rec_out,
                                       void *stream) {
  R1_REQUIRE(ctx && org && params && eob_out && dist_out);
  R1_REQUIRE(tx_size >= 0 && tx_size < 19);
  R1_REQUIRE(params->bit_depth == org->bit_depth);
  R1_REQUIRE(dist_kind == R1_DIST_WSSE || dist_kind == R1_DIST_CDEF);
  R1_REQUIRE(xdec >= 0 && xdec <= 1 && ydec >= 0 && ydec <= 1);
  R1_REQUIRE(dist_kind != R1_DIST_CDEF || (xdec == 0 && ydec == 0));
  R1_REQUIRE(!scales || scale_stride > 0);
  RdoQuantArgs qa = {};
  qa.qp = r1q::make_qparams(*params, tx_size, org->bytes_per_px == 1 ? 2 : 4);
  for (int k = 0; k < 3; k++) qa.scan[k] = ctx->scan_dev + ctx->scan_off[tx_size][k];
  qa.tx_size = tx_size;
  qa.q_bin = params->qindex / 32;
  qa.eob = eob_out;
  qa.qcoeffs = qcoeffs_out;
  qa.dist_kind = dist_kind;
  qa.inv_shift = r1itx::kInvShift[tx_size];
  qa.scales = scales;
  qa.scale_stride = scale_stride;
  qa.xdec = xdec;
  qa.ydec = ydec;
  qa.pix_dist = (unsigned long long *)dist_out;
  qa.rec = rec_out;
  return rdo_dispatch(ctx, org, ref, w, h, tx_size, cands, n, sad_out, satd_out, nullptr, nullptr,
                      &qa, stream);
}

extern "C" int r1_rdo_pred_cand_batch(r1_ctx *ctx, const R1Plane *org, const void *pred, int w, int h,
                                      int tx_size, const R1RdoCand *cands, int n,
                                      const R1QuantParams *params, int dist_kind,
                                      const uint32_t *scales, int scale_stride, int xdec, int ydec,
                                      uint32_t *sad_out, uint32_t *satd_out, uint16_t *eob_out,
                                      uint64_t *dist_out, void *qcoeffs_out, void *rec_out,
                                      void *stream) {
  R1_REQUIRE(ctx && org && pred && params && eob_out && dist_out);
  R1_REQUIRE(tx_size >= 0 && tx_size < 19);
  R1_REQUIRE(params->bit_depth == org->bit_depth);
  R1_REQUIRE(dist_kind == 0 || dist_kind == R1_DIST_WSSE || dist_kind == R1_DIST_CDEF);
  R1_REQUIRE(xdec >= 0 && xdec <= 1 && ydec >= 0 && ydec <= 1);
  R1_REQUIRE(dist_kind != R1_DIST_CDEF || (xdec == 0 && ydec == 0));
  R1_REQUIRE(!scales || scale_stride > 0);
  R1_REQUIRE(dist_kind != 0 || !rec_out);
  RdoQuantArgs qa = {};
  qa.qp = r1q::make_qparams(*params, tx_size, org->bytes_per_px == 1 ? 2 : 4);
  for (int k = 0; k < 3; k++) qa.scan[k] = ctx->scan_dev + ctx->scan_off[tx_size][k];
  qa.tx_size = tx_size;
  qa.q_bin = params->qindex / 32;
  qa.eob = eob_out;
  qa.qcoeffs = qcoeffs_out;
  qa.pred_in = pred;
  if (dist_kind == 0) {
    qa.tx_dist = (unsigned long long *)dist_out;   // transform-domain distortion (QM 1)
  } else {
    qa.dist_kind = dist_kind;
    qa.inv_shift = r1itx::kInvShift[tx_size];
    qa.scales = scales;
    qa.scale_stride = scale_stride;
    qa.xdec = xdec;
    qa.ydec = ydec;
    qa.pix_dist = (unsigned long long *)dist_out;
    qa.rec = rec_out;
  }
  return rdo_dispatch(ctx, org, nullptr, w, h, tx_size, cands, n, sad_out, satd_out, nullptr, nullptr,
                      &qa, stream);
}
// av1_tx_used[get_tx_set(tx_size, is_inter, use_reduced_set)] (src/context/transform_unit.rs:37-44,
// 123-148) as a bit mask over TxType, optionally cut down to RAV1E_TX_TYPES (src/transform/mod.rs:28-44):
// the types the loop of rdo_tx_type_decision (src/rdo.rs:1732-1736) does not skip.
extern "C" uint32_t r1_tx_type_mask(int tx_size, int is_inter, int use_reduced_set, int rav1e_types_only) {
  if (tx_size < 0 || tx_size >= 19) return 0;
  const int wl = r1tx::kTxWLog2[tx_size], hl = r1tx::kTxHLog2[tx_size];
  const int up = wl > hl ? wl : hl, dn = wl < hl ? wl : hl;   // sqr_up / sqr as log2 of the side
  // TxSet rows of av1_tx_used
  constexpr uint32_t DCTONLY = 0x0001, INTER_3 = 0x0201, INTRA_2 = 0x020F, INTRA_1 = 0x0E0F, INTER_2 = 0x0FFF,
                     INTER_1 = 0xFFFF;
  uint32_t m;
  if (up > 5) m = DCTONLY;
  else if (is_inter) m = (use_reduced_set || up == 5) ? INTER_3 : (dn == 4 ? INTER_2 : INTER_1);
  else m = up == 5 ? DCTONLY : ((use_reduced_set || dn == 4) ? INTRA_2 : INTRA_1);
  return rav1e_types_only ? (m & 0x0E0Fu) : m;
}

extern "C" uint32_t r1_tx_type_mask(int tx_size, int is_inter, int use_reduced_set, int rav1e_types_only);
extern "C" int r1_rdo_txsearch_batch(r1_ctx *ctx, const R1Plane *org, const R1Plane *ref, const void *pred,
                                     int w, int h, int tx_size, const R1RdoCand *cands, int n,
                                     uint32_t tx_type_mask, const R1QuantParams *params, int dist_kind,
                                     const uint32_t *scales, int scale_stride, int xdec, int ydec,
                                     uint32_t *sad_out, uint32_t *satd_out, uint16_t *eob_out,
                                     uint64_t *dist_out, uint64_t *est_rate_out, void *qcoeffs_out,
                                     void *rec_out, void *stream) {
  R1_REQUIRE(ctx && org && params && eob_out && dist_out);
  R1_REQUIRE((ref != nullptr) != (pred != nullptr));
  R1_REQUIRE(tx_size >= 0 && tx_size < 19);
  R1_REQUIRE(params->bit_depth == org->bit_depth);
  R1_REQUIRE(dist_kind == 0 || dist_kind == R1_DIST_WSSE || dist_kind == R1_DIST_CDEF);
  R1_REQUIRE(xdec >= 0 && xdec <= 1 && ydec >= 0 && ydec <= 1);
  R1_REQUIRE(dist_kind != R1_DIST_CDEF || (xdec == 0 && ydec == 0));
  R1_REQUIRE(!scales || scale_stride > 0);
  R1_REQUIRE(dist_kind != 0 || !rec_out);
  R1_REQUIRE(dist_kind == 0 || !est_rate_out);
  // WHT (16) has no scan order; the mask is over the 16 TxTypes of the tx sets
  R1_REQUIRE(tx_type_mask != 0 && tx_type_mask <= 0xFFFFu);
  const int up = r1tx::kTxWLog2[tx_size] > r1tx::kTxHLog2[tx_size] ? r1tx::kTxWLog2[tx_size] : r1tx::kTxHLog2[tx_size];
  const bool side64 = up > 5, side32 = up == 5;
  // a 64-point side codes DCT_DCT only (TX_SET_DCTONLY)
  R1_REQUIRE(!side64 || tx_type_mask == 1u);
  // every type of the mask must exist for the size: the inter sets are the largest (av1_tx_used; a 32-point side has
  // DCT_DCT and IDTX only -- the reference's 1-D tables have no other kernel there and would panic)
  R1_REQUIRE((tx_type_mask & ~r1_tx_type_mask(tx_size, 1, 0, 0)) == 0);
  RdoQuantArgs qa = {};
  qa.qp = r1q::make_qparams(*params, tx_size, org->bytes_per_px == 1 ? 2 : 4);
  for (int k = 0; k < 3; k++) qa.scan[k] = ctx->scan_dev + ctx->scan_off[tx_size][k];
  qa.tx_size = tx_size;
  qa.q_bin = params->qindex / 32;
  qa.eob = eob_out;
  qa.qcoeffs = qcoeffs_out;
  qa.pred_in = pred;
  qa.tx_mask = tx_type_mask;
  qa.nt = __builtin_popcount(tx_type_mask);
  if (dist_kind == 0) {
    qa.tx_dist = (unsigned long long *)dist_out;
    qa.est_rate = (unsigned long long *)est_rate_out;
  } else {
    qa.dist_kind = dist_kind;
    qa.inv_shift = r1itx::kInvShift[tx_size];
    qa.scales = scales;
    qa.scale_stride = scale_stride;
    qa.xdec = xdec;
    qa.ydec = ydec;
    qa.pix_dist = (unsigned long long *)dist_out;
    qa.rec = rec_out;
  }
  if (!side64 && !side32)
    return rdo_dispatch(ctx, org, ref, w, h, tx_size, cands, n, sad_out, satd_out, nullptr, nullptr, &qa, stream, true);
  // 32- and 64-point sides: one plain launch per type (at most two), the type forced, results into its slot
  int slot = 0;
  for (uint32_t m = tx_type_mask; m != 0; m &= m - 1, slot++) {
    qa.tx_mask = m & (0u - m);
    qa.slot = slot;
    const int rc = rdo_dispatch(ctx, org, ref, w, h, tx_size, cands, n, slot == 0 ? sad_out : nullptr,
                                slot == 0 ? satd_out : nullptr, nullptr, nullptr, &qa, stream, false);
    if (rc != R1_OK) return rc;
  }
  return R1_OK;
}
#endif   // R1_RDO_DISPATCH_TU
