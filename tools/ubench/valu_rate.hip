// valu_rate.hip -- micro-benchmark: issue rate of the VALU ops the fused RDO
// kernel is made of (wave64, 8 waves per SIMD, 8 independent accumulators per
// lane, inline asm so that the compiler cannot fold anything).
// Reports cycles per wave64 instruction per SIMD at the nominal 2.4 GHz.
// build: hipcc -O3 --offload-arch=gfx950 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define REP 2048
#define OP2(name, txt)                                                         \
  struct name {                                                                \
    static __device__ __forceinline__ void op(uint32_t &a, uint32_t b) {       \
      asm volatile(txt : "+v"(a) : "v"(b));                                    \
    }                                                                          \
  };
OP2(AddU32, "v_add_u32 %0, %0, %1")
OP2(SubU32, "v_sub_u32 %0, %0, %1")
OP2(Ashr, "v_ashrrev_i32 %0, 1, %0")
OP2(Lshr, "v_lshrrev_b32 %0, 3, %0")
OP2(Xor, "v_xor_b32 %0, %0, %1")
OP2(MaxI32, "v_max_i32 %0, %0, %1")
OP2(Mul24, "v_mul_i32_i24 %0, %0, %1")
OP2(Mad24, "v_mad_i32_i24 %0, %0, %1, %1")
OP2(MulLo, "v_mul_lo_u32 %0, %0, %1")
OP2(Add3, "v_add3_u32 %0, %0, %1, %1")
OP2(Med3, "v_med3_i32 %0, %0, %1, %1")
OP2(AlignByte, "v_alignbyte_b32 %0, %0, %1, %1")
OP2(Perm, "v_perm_b32 %0, %0, %1, %1")
OP2(Dot4, "v_dot4_i32_i8 %0, %0, %1, %0")
OP2(Dot2, "v_dot2_i32_i16 %0, %0, %1, %0")
OP2(PkAddI16, "v_pk_add_i16 %0, %0, %1")
OP2(PkMaxI16, "v_pk_max_i16 %0, %0, %1")
OP2(FmaF32, "v_fma_f32 %0, %0, %1, %1")
OP2(PkFmaF32x, "v_add_f32 %0, %0, %1")
OP2(SadU8, "v_sad_u8 %0, %0, %1, %0")
OP2(MovDpp, "v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
OP2(Cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
// round 2: candidates for cheaper sequences
OP2(CvtPkI16, "v_cvt_pk_i16_i32 %0, %0, %1")
OP2(SadU32, "v_sad_u32 %0, %0, %1, %0")
OP2(SatPkU8, "v_sat_pk_u8_i16 %0, %0")
OP2(AddE64, "v_add_u32_e64 %0, %0, %1")
OP2(LshlAdd, "v_lshl_add_u32 %0, %0, 1, %1")
OP2(BfeI32, "v_bfe_i32 %0, %0, 2, 16")
OP2(MulHi24, "v_mul_hi_i32_i24 %0, %0, %1")
OP2(PkMadI16, "v_pk_mad_i16 %0, %0, %1, %1")
OP2(PkSubI16, "v_pk_sub_i16 %0, %0, %1")
OP2(AndOr, "v_and_or_b32 %0, %0, %1, %1")
OP2(Bfi, "v_bfi_b32 %0, %0, %1, %1")
OP2(PackF16, "v_pack_b32_f16 %0, %0, %1")
OP2(Mul24Imm, "v_mul_i32_i24 %0, 0x2d41, %0")
OP2(Mad24Sgpr, "v_mad_i32_i24 %0, %0, s0, %1")
OP2(MinI32, "v_min_i32 %0, %0, %1")
OP2(AshrImm, "v_ashrrev_i32 %0, 13, %0")
OP2(MovDppRhm, "v_mov_b32_dpp %0, %1 row_half_mirror row_mask:0xf bank_mask:0xf")
OP2(AddDpp, "v_add_u32_dpp %0, %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
OP2(SubI16Sdwa, "v_sub_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1")

template <typename OP>
__global__ __launch_bounds__(256) void k(uint32_t *out, uint32_t seed) {
  uint32_t a[8];
  for (int i = 0; i < 8; i++) a[i] = seed + threadIdx.x * (i + 1);
  for (int r = 0; r < REP; r++) {
#pragma unroll
    for (int i = 0; i < 8; i++) OP::op(a[i], seed);
  }
  uint32_t s = 0;
  for (int i = 0; i < 8; i++) s += a[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <typename OP>
void run(const char *name) {
  uint32_t *d;
  (void)hipMalloc(&d, 256 * 8 * 256 * sizeof(uint32_t));
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int grid = 256 * 8;   // 8 workgroups of 4 waves per CU -> 8 waves per SIMD
  hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(256), 0, 0, d, 3u);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(256), 0, 0, d, 3u);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double wave_instr_per_simd = (double)grid * 4 * REP * 8 / (256.0 * 4);
  printf("%-16s %8.3f ms  %.2f cycles per wave64 instruction per SIMD (2.4 GHz nominal)\n", name,
         ms, ms * 1e-3 * 2.4e9 / wave_instr_per_simd);
  (void)hipFree(d);
}

// ---- LDS read rates: the H pass of put_8tap reads 8 neighbouring bytes per lane ----
template <int MODE>
__global__ __launch_bounds__(256) void k_lds(uint32_t *out, int off) {
  __shared__ __attribute__((aligned(16))) uint8_t sm[16384];
  for (int i = threadIdx.x; i < 4096; i += 256) ((uint32_t *)sm)[i] = i * 2654435761u;
  __syncthreads();
  typedef uint64_t __attribute__((aligned(1))) u64u;
  typedef uint64_t __attribute__((aligned(8))) u64a;
  struct u128 { uint32_t x, y, z, w; };
  typedef u128 __attribute__((aligned(2))) u128u;
  uint32_t acc = 0;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  // 8 candidates x 8 columns per wave: rows of 16 bytes, as the 8x8 window rows are
  int base = wv * 4096 + (lane >> 3) * 240 + (lane & 7) + off;
  for (int r = 0; r < REP / 8; r++) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const int a = base + ((r * 8 + i) % 15) * 16;
      if (MODE == 0) {           // three aligned dwords (what the kernel did)
        const uint32_t *p = (const uint32_t *)(sm + (a & ~3));
        acc += p[0] ^ p[1] ^ p[2];
      } else if (MODE == 1) {    // one unaligned 8-byte read
        const uint64_t v = *(const u64u *)(sm + a);
        acc += (uint32_t)v ^ (uint32_t)(v >> 32);
      } else if (MODE == 2) {    // one aligned 8-byte read (reference point)
        const uint64_t v = *(const u64a *)(sm + (a & ~7));
        acc += (uint32_t)v ^ (uint32_t)(v >> 32);
      } else {                   // one 2-byte-aligned 16-byte read (16-bit pixels)
        const u128 v = *(const u128u *)(sm + (a & ~1));
        acc += v.x ^ v.y ^ v.z ^ v.w;
      }
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}
template <int MODE>
void run_lds(const char *name) {
  uint32_t *d;
  (void)hipMalloc(&d, 256 * 8 * 256 * sizeof(uint32_t));
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int grid = 256 * 8;
  hipLaunchKernelGGL(k_lds<MODE>, dim3(grid), dim3(256), 0, 0, d, 3);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k_lds<MODE>, dim3(grid), dim3(256), 0, 0, d, 3);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double rows_per_cu = (double)grid * 4 * REP / 256.0;
  printf("%-40s %8.3f ms  %.2f cycles per wave-row-read per CU (2.4 GHz nominal)\n", name, ms,
         ms * 1e-3 * 2.4e9 / rows_per_cu);
  (void)hipFree(d);
}
void lds_rates() {
  run_lds<0>("LDS 3 x ds_read_b32 aligned (+xor)");
  run_lds<1>("LDS 1 x ds_read_b64 unaligned");
  run_lds<2>("LDS 1 x ds_read_b64 aligned");
  run_lds<3>("LDS 1 x ds_read_b128 2-byte aligned");
}

int main() {
  run<AddU32>("v_add_u32"); run<SubU32>("v_sub_u32"); run<Ashr>("v_ashrrev_i32");
  run<Lshr>("v_lshrrev_b32"); run<Xor>("v_xor_b32"); run<MaxI32>("v_max_i32");
  run<Mul24>("v_mul_i32_i24"); run<Mad24>("v_mad_i32_i24"); run<MulLo>("v_mul_lo_u32");
  run<Add3>("v_add3_u32"); run<Med3>("v_med3_i32"); run<AlignByte>("v_alignbyte_b32");
  run<Perm>("v_perm_b32"); run<Dot4>("v_dot4_i32_i8"); run<Dot2>("v_dot2_i32_i16");
  run<PkAddI16>("v_pk_add_i16"); run<PkMaxI16>("v_pk_max_i16"); run<SadU8>("v_sad_u8");
  run<FmaF32>("v_fma_f32"); run<PkFmaF32x>("v_add_f32"); run<MovDpp>("v_mov_b32_dpp");
  run<Cndmask>("v_cndmask_b32");
  run<CvtPkI16>("v_cvt_pk_i16_i32"); run<SadU32>("v_sad_u32"); run<SatPkU8>("v_sat_pk_u8_i16");
  run<AddE64>("v_add_u32_e64"); run<LshlAdd>("v_lshl_add_u32"); run<BfeI32>("v_bfe_i32");
  run<MulHi24>("v_mul_hi_i32_i24"); run<PkMadI16>("v_pk_mad_i16"); run<PkSubI16>("v_pk_sub_i16");
  run<AndOr>("v_and_or_b32"); run<Bfi>("v_bfi_b32"); run<PackF16>("v_pack_b32_f16");
  run<Mul24Imm>("v_mul_i32_i24 imm"); run<Mad24Sgpr>("v_mad_i32_i24 sgpr"); run<MinI32>("v_min_i32");
  run<AshrImm>("v_ashrrev 13"); run<MovDppRhm>("v_mov_dpp rhm"); run<AddDpp>("v_add_u32_dpp");
  run<SubI16Sdwa>("v_sub_u32_sdwa");
  lds_rates();
  return 0;
}
