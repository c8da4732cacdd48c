// valu_rate2.hip -- round-3 micro-benchmarks behind the headline-kernel rewrite:
//  (1) issue rate of the candidate replacement ops (SDWA forms, VOP2 dot-accumulate forms,
//      conversions, packed shifts, 16-bit SADs), same harness as valu_rate.hip;
//  (2) mixes: does a "fast" op next to a "slow" op cost the sum of the two?
//  (3) cross-lane moves through the LDS pipe (ds_swizzle / ds_bpermute) beside VALU work:
//      do they take VALU issue slots away?
//  (4) unaligned LDS reads (ds_read_b64 at byte granularity): correctness and rate.
// build: hipcc -O3 --offload-arch=gfx950 valu_rate2.hip -o valu_rate2
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define REP 2048
#define OP2(name, txt)                                                         \
  struct name {                                                                \
    static __device__ __forceinline__ void op(uint32_t &a, uint32_t b) {       \
      asm volatile(txt : "+v"(a) : "v"(b));                                    \
    }                                                                          \
  };
OP2(AddU32, "v_add_u32 %0, %0, %1")
OP2(Mad24, "v_mad_i32_i24 %0, %0, %1, %1")
OP2(AndB32, "v_and_b32 %0, %0, %1")
OP2(OrB32, "v_or_b32 %0, %0, %1")
OP2(Lshl, "v_lshlrev_b32 %0, 3, %0")
OP2(MovB32, "v_mov_b32 %0, %1")
OP2(Dot4c, "v_dot4c_i32_i8 %0, %1, %1")
OP2(Dot2c, "v_dot2c_i32_i16 %0, %1, %1")
OP2(Dot4Vop3, "v_dot4_i32_i8 %0, %1, %1, %0")
OP2(SubSdwaB3, "v_sub_u32_sdwa %0, %0, sext(%0) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3")
OP2(AshrSdwaW1, "v_ashrrev_i32_sdwa %0, %1, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD")
OP2(AddSdwaW0, "v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1")
OP2(CvtF32I32, "v_cvt_f32_i32 %0, %0")
OP2(CvtI32F32, "v_cvt_i32_f32 %0, %0")
OP2(MulHi24, "v_mul_hi_i32_i24 %0, %0, %1")
OP2(PkMulLo, "v_pk_mul_lo_u16 %0, %0, %1")
OP2(PkAshr, "v_pk_ashrrev_i16 %0, 1, %0")
OP2(PkLshl, "v_pk_lshlrev_b16 %0, 1, %0")
OP2(SadU16, "v_sad_u16 %0, %0, %1, %0")
OP2(SadU8, "v_sad_u8 %0, %0, %1, %0")
OP2(MsadU8, "v_msad_u8 %0, %0, %1, %0")
OP2(BfeU32, "v_bfe_u32 %0, %0, 2, 16")
OP2(MinU32, "v_min_u32 %0, %0, %1")
OP2(MaxU32, "v_max_u32 %0, %0, %1")
OP2(MinI16, "v_min_i16 %0, %0, %1")
OP2(SubRev, "v_subrev_u32 %0, %0, %1")
OP2(AddCo, "v_add_co_u32 %0, vcc, %0, %1")
OP2(CmpLt, "v_cmp_lt_i32 vcc, %0, %1")
OP2(Add3, "v_add3_u32 %0, %0, %1, %1")
OP2(AddLshl, "v_add_lshl_u32 %0, %0, %1, 4")
OP2(LshlOr, "v_lshl_or_b32 %0, %0, 16, %1")
OP2(CvtPkU8, "v_cvt_pk_u8_f32 %0, %0, 1, %1")
OP2(SatPkU8, "v_sat_pk_u8_i16 %0, %0")
OP2(AlignBit, "v_alignbit_b32 %0, %0, %1, 3")
OP2(AlignByteImm, "v_alignbyte_b32 %0, %0, %1, 1")
OP2(Med3U16, "v_med3_i16 %0, %0, %1, %1")
OP2(PkMin, "v_pk_min_i16 %0, %0, %1")
OP2(Mul24, "v_mul_i32_i24 %0, %0, %1")
OP2(MulU24, "v_mul_u32_u24 %0, %0, %1")
OP2(Fmac, "v_fmac_f32 %0, %1, %1")

OP2(XorDpp, "v_xor_b32_dpp %0, %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
OP2(SubDpp, "v_sub_u32_dpp %0, %1, %0 row_half_mirror row_mask:0xf bank_mask:0xf")
OP2(Permlane32, "v_permlane32_swap_b32 %0, %1")
OP2(Permlane16, "v_permlane16_swap_b32 %0, %1")
// two-op mixes on independent registers (a: slow op, b via the second asm operand: fast op)
struct MixMadAdd {
  static __device__ __forceinline__ void op(uint32_t &a, uint32_t b) {
    asm volatile("v_mad_i32_i24 %0, %0, %1, %1\n\tv_add_u32 %0, %0, %1" : "+v"(a) : "v"(b));
  }
};
struct MixMadAshr {
  static __device__ __forceinline__ void op(uint32_t &a, uint32_t b) {
    asm volatile("v_mad_i32_i24 %0, %0, %1, %1\n\tv_ashrrev_i32 %0, 1, %0" : "+v"(a) : "v"(b));
  }
};
struct MixAddAdd {
  static __device__ __forceinline__ void op(uint32_t &a, uint32_t b) {
    asm volatile("v_add_u32 %0, %0, %1\n\tv_xor_b32 %0, %0, %1" : "+v"(a) : "v"(b));
  }
};
struct MixDot4Align {
  static __device__ __forceinline__ void op(uint32_t &a, uint32_t b) {
    asm volatile("v_alignbyte_b32 %0, %0, %1, %1\n\tv_dot4c_i32_i8 %0, %1, %1" : "+v"(a) : "v"(b));
  }
};

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

static uint32_t *g_d;
template <typename OP>
void run(const char *name, int ops_per_call = 1) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int grid = 256 * 8;   // 8 workgroups of 4 waves per CU -> 8 waves per SIMD
  hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(256), 0, 0, g_d, 3u);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(256), 0, 0, g_d, 3u);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double calls_per_simd = (double)grid * 4 * REP * 8 / (256.0 * 4);
  printf("%-28s %8.3f ms  %.2f cycles per call (%d VALU op%s) per SIMD at 2.4 GHz nominal\n", name, ms,
         ms * 1e-3 * 2.4e9 / calls_per_simd, ops_per_call, ops_per_call > 1 ? "s" : "");
}

// ---- LDS-pipe cross-lane moves beside VALU work ----
// MODE 0: 4 v_mad per iteration only; 1: + 1 v_mov_dpp; 2: + 1 ds_swizzle (xor 1);
// 3: + 1 ds_bpermute; 4: 4 v_mad + 2 ds_swizzle
template <int MODE>
__global__ __launch_bounds__(256) void k_xl(uint32_t *out, uint32_t seed) {
  uint32_t a[4], x = seed * threadIdx.x, idx = ((threadIdx.x ^ 1) & 63) * 4;
  for (int i = 0; i < 4; i++) a[i] = seed + threadIdx.x * (i + 1);
  for (int r = 0; r < REP; r++) {
#pragma unroll
    for (int u = 0; u < 4; u++) {
#pragma unroll
      for (int i = 0; i < 4; i++) asm volatile("v_mad_i32_i24 %0, %0, %1, %1" : "+v"(a[i]) : "v"(seed));
      if (MODE == 1) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(x));
      if (MODE == 2 || MODE == 4) asm volatile("ds_swizzle_b32 %0, %0 offset:0x041F" : "+v"(x));   // xor 1
      if (MODE == 4) asm volatile("ds_swizzle_b32 %0, %0 offset:0x081F" : "+v"(x));               // xor 2
      if (MODE == 3) asm volatile("ds_bpermute_b32 %0, %1, %0" : "+v"(x) : "v"(idx));
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    a[0] ^= x & 1;
  }
  out[blockIdx.x * 256 + threadIdx.x] = a[0] + a[1] + a[2] + a[3] + x;
}
template <int MODE>
void run_xl(const char *name) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int grid = 256 * 8;
  hipLaunchKernelGGL(k_xl<MODE>, dim3(grid), dim3(256), 0, 0, g_d, 3u);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k_xl<MODE>, dim3(grid), dim3(256), 0, 0, g_d, 3u);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double groups_per_simd = (double)grid * 4 * REP * 4 / (256.0 * 4);
  printf("%-44s %8.3f ms  %.2f cycles per group of 4 v_mad (+ the move) per SIMD\n", name, ms,
         ms * 1e-3 * 2.4e9 / groups_per_simd);
}

// ---- unaligned LDS reads: correctness and rate ----
__global__ void k_unal(uint32_t *out, int *bad) {
  __shared__ __attribute__((aligned(16))) uint8_t sm[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) sm[i] = (uint8_t)(i * 7 + 3);
  __syncthreads();
  for (int off = 0; off < 24; off++) {
    const uint32_t a = 40 * threadIdx.x + off;
    uint64_t v64; uint32_t v32; uint32_t q0, q1, q2, q3;
    asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v64) : "v"(a) : "memory");
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v32) : "v"(a) : "memory");
    uint64_t w64 = 0; uint32_t w32 = 0;
    for (int b = 7; b >= 0; b--) w64 = (w64 << 8) | sm[a + b];
    for (int b = 3; b >= 0; b--) w32 = (w32 << 8) | sm[a + b];
    if (v64 != w64) atomicAdd(&bad[0], 1);
    if (v32 != w32) atomicAdd(&bad[1], 1);
    struct U4 { uint32_t x, y, z, w; } v128;
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v128) : "v"(a) : "memory");
    q0 = v128.x; q1 = v128.y; q2 = v128.z; q3 = v128.w;
    uint32_t e[4];
    for (int j = 0; j < 4; j++) { e[j] = 0; for (int b = 3; b >= 0; b--) e[j] = (e[j] << 8) | sm[a + 4 * j + b]; }
    if (q0 != e[0] || q1 != e[1] || q2 != e[2] || q3 != e[3]) atomicAdd(&bad[2], 1);
    // unaligned 8-byte store
    asm volatile("ds_write_b64 %0, %1 offset:2048\n\ts_waitcnt lgkmcnt(0)" :: "v"(a & 1023), "v"(w64) : "memory");
  }
  out[threadIdx.x] = sm[threadIdx.x];
}

// H-pass read patterns of the 8x8 kernel: 8 candidates x 8 columns per wave, window rows of 16 bytes
template <int MODE>
__global__ __launch_bounds__(256) void k_lds(uint32_t *out, int off) {
  __shared__ __attribute__((aligned(16))) uint8_t sm[16384];
  for (int i = threadIdx.x; i < 4096; i += 256) ((uint32_t *)sm)[i] = i * 2654435761u;
  __syncthreads();
  uint32_t acc = 0;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint32_t base = wv * 4096 + (lane >> 3) * 240 + (lane & 7) + off;
  const uint32_t sh = base & 3;
  for (int r = 0; r < REP / 8; r++) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const uint32_t a = base + ((r * 8 + i) % 15) * 16;
      if (MODE == 0) {           // three aligned dwords + two v_alignbyte (what the kernel does)
        const uint32_t *p = (const uint32_t *)(sm + (a & ~3u));
        const uint32_t d0 = p[0], d1 = p[1], d2 = p[2];
        acc += __builtin_amdgcn_alignbyte(d1, d0, sh) ^ __builtin_amdgcn_alignbyte(d2, d1, sh);
      } else if (MODE == 1) {    // one unaligned 8-byte read
        uint64_t v;
        asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(a));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        acc += (uint32_t)v ^ (uint32_t)(v >> 32);
      } else {                   // aligned 8-byte read
        uint64_t v;
        asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(a & ~7u));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        acc += (uint32_t)v ^ (uint32_t)(v >> 32);
      }
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}
template <int MODE>
void run_lds(const char *name, int off) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int grid = 256 * 8;
  hipLaunchKernelGGL(k_lds<MODE>, dim3(grid), dim3(256), 0, 0, g_d, off);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k_lds<MODE>, dim3(grid), dim3(256), 0, 0, g_d, off);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double rows_per_simd = (double)grid * 4 * REP / (256.0 * 4);
  printf("%-44s off=%d %8.3f ms  %.2f cycles per wave-row-read per SIMD\n", name, off, ms,
         ms * 1e-3 * 2.4e9 / rows_per_simd);
}

int main() {
  (void)hipMalloc(&g_d, 256 * 8 * 256 * sizeof(uint32_t));
  run<AddU32>("v_add_u32"); run<Mad24>("v_mad_i32_i24");
  run<AndB32>("v_and_b32"); run<OrB32>("v_or_b32"); run<Lshl>("v_lshlrev_b32"); run<MovB32>("v_mov_b32");
  run<Dot4c>("v_dot4c_i32_i8 (VOP2)"); run<Dot2c>("v_dot2c_i32_i16 (VOP2)"); run<Dot4Vop3>("v_dot4_i32_i8 (VOP3P)");
  run<SubSdwaB3>("v_sub_u32_sdwa sext BYTE_3"); run<AshrSdwaW1>("v_ashrrev_i32_sdwa ->WORD_1");
  run<AddSdwaW0>("v_add_u32_sdwa WORD sel");
  run<CvtF32I32>("v_cvt_f32_i32"); run<CvtI32F32>("v_cvt_i32_f32"); run<MulHi24>("v_mul_hi_i32_i24");
  run<PkMulLo>("v_pk_mul_lo_u16"); run<PkAshr>("v_pk_ashrrev_i16"); run<PkLshl>("v_pk_lshlrev_b16");
  run<SadU16>("v_sad_u16"); run<SadU8>("v_sad_u8"); run<MsadU8>("v_msad_u8"); run<BfeU32>("v_bfe_u32");
  run<MinU32>("v_min_u32"); run<MaxU32>("v_max_u32"); run<MinI16>("v_min_i16"); run<SubRev>("v_subrev_u32");
  run<AddCo>("v_add_co_u32"); run<CmpLt>("v_cmp_lt_i32"); run<Add3>("v_add3_u32"); run<AddLshl>("v_add_lshl_u32");
  run<LshlOr>("v_lshl_or_b32"); run<CvtPkU8>("v_cvt_pk_u8_f32"); run<SatPkU8>("v_sat_pk_u8_i16");
  run<AlignBit>("v_alignbit_b32 imm"); run<AlignByteImm>("v_alignbyte_b32 imm"); run<Med3U16>("v_med3_i16");
  run<PkMin>("v_pk_min_i16"); run<Mul24>("v_mul_i32_i24"); run<MulU24>("v_mul_u32_u24");
  run<Fmac>("v_fmac_f32");
  run<XorDpp>("v_xor_b32_dpp"); run<SubDpp>("v_sub_u32_dpp rhm");
  run<Permlane32>("v_permlane32_swap"); run<Permlane16>("v_permlane16_swap");
  run<MixMadAdd>("mix v_mad24 + v_add", 2); run<MixMadAshr>("mix v_mad24 + v_ashr", 2);
  run<MixAddAdd>("mix v_add + v_xor", 2); run<MixDot4Align>("mix v_alignbyte + v_dot4c", 2);
  run_xl<0>("4 v_mad only"); run_xl<1>("4 v_mad + v_mov_dpp"); run_xl<2>("4 v_mad + ds_swizzle");
  run_xl<3>("4 v_mad + ds_bpermute"); run_xl<4>("4 v_mad + 2 ds_swizzle");
  int *bad; (void)hipMalloc(&bad, 16); (void)hipMemset(bad, 0, 16);
  hipLaunchKernelGGL(k_unal, dim3(1), dim3(64), 0, 0, g_d, bad);
  int hb[4] = {0, 0, 0, 0};
  hipError_t err = hipDeviceSynchronize();
  (void)hipMemcpy(hb, bad, 16, hipMemcpyDeviceToHost);
  printf("unaligned LDS reads (24 byte offsets x 64 lanes): ds_read_b64 %d wrong, ds_read_b32 %d wrong, "
         "ds_read_b128 %d wrong (%s)\n", hb[0], hb[1], hb[2], hipGetErrorString(err));
  for (int off = 0; off < 4; off++) {
    run_lds<0>("LDS 3 x b32 aligned + 2 alignbyte", off);
    run_lds<1>("LDS 1 x ds_read_b64 unaligned", off);
    run_lds<2>("LDS 1 x ds_read_b64 aligned", off);
  }
  return 0;
}
