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

int main() {
  run<AddU32>("v_add_u32"); run<SubU32>("v_sub_u32"); run<Ashr>("v_ashrrev_i32");
  run<Lshr>("v_lshrrev_b32"); run<Xor>("v_xor_b32"); run<MaxI32>("v_max_i32");
  run<Mul24>("v_mul_i32_i24"); run<Mad24>("v_mad_i32_i24"); run<MulLo>("v_mul_lo_u32");
  run<Add3>("v_add3_u32"); run<Med3>("v_med3_i32"); run<AlignByte>("v_alignbyte_b32");
  run<Perm>("v_perm_b32"); run<Dot4>("v_dot4_i32_i8"); run<Dot2>("v_dot2_i32_i16");
  run<PkAddI16>("v_pk_add_i16"); run<PkMaxI16>("v_pk_max_i16"); run<SadU8>("v_sad_u8");
  run<FmaF32>("v_fma_f32"); run<PkFmaF32x>("v_add_f32"); run<MovDpp>("v_mov_b32_dpp");
  run<Cndmask>("v_cndmask_b32");
  return 0;
}
