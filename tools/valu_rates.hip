// Issue cost (cycles per wave-instruction) of the VALU instructions the attention softmax is made of, measured with
// s_memtime around an unrolled run of independent instructions, for 1, 2 and 4 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O2 -o tools/valu_rates tools/valu_rates.hip && tools/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define REP8(x) x x x x x x x x
#define BODY(INSTR)                                                                                   \
  {                                                                                                   \
    unsigned long long t0, t1;                                                                        \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory"); \
    for (int it = 0; it < 64; ++it) {                                                                 \
      asm volatile(REP8(INSTR) REP8(INSTR) REP8(INSTR) REP8(INSTR)                                    \
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)   \
                   : "v"(b0), "v"(b1));                                                               \
    }                                                                                                 \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");                      \
    /* INSTR holds 4 instructions: 64 x 32 x 4 per wave.  Oldest-first arbitration: time first start .. last end */ \
    if ((threadIdx.x & 63) == 0) {                                                                    \
      atomicMin((unsigned long long*)(out + 2), t0);                                                  \
      atomicMax((unsigned long long*)(out + 4), t1);                                                  \
    }                                                                                                 \
  }

#define KERNEL(NAME, INSTR)                                                       \
  __global__ void NAME(float* out, float seed) {                                  \
    float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7; \
    float b0 = seed * 0.5f, b1 = seed * 0.25f;                                    \
    BODY(INSTR)                                                                   \
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.f) out[1] = a0;           \
  }

// 4 independent destination registers in rotation (a0..a3), sources b0/b1 or a4..a7
KERNEL(k_add, "v_add_f32 %0, %8, %4\n\tv_add_f32 %1, %8, %5\n\tv_add_f32 %2, %9, %6\n\tv_add_f32 %3, %9, %7\n\t")
KERNEL(k_fma, "v_fma_f32 %0, %8, %4, %9\n\tv_fma_f32 %1, %8, %5, %9\n\tv_fma_f32 %2, %9, %6, %8\n\tv_fma_f32 %3, %9, %7, %8\n\t")
KERNEL(k_exp, "v_exp_f32 %0, %4\n\tv_exp_f32 %1, %5\n\tv_exp_f32 %2, %6\n\tv_exp_f32 %3, %7\n\t")
KERNEL(k_exp16, "v_exp_f16 %0, %4\n\tv_exp_f16 %1, %5\n\tv_exp_f16 %2, %6\n\tv_exp_f16 %3, %7\n\t")
KERNEL(k_cvtpk, "v_cvt_pk_f16_f32 %0, %4, %5\n\tv_cvt_pk_f16_f32 %1, %5, %6\n\tv_cvt_pk_f16_f32 %2, %6, %7\n\tv_cvt_pk_f16_f32 %3, %7, %4\n\t")
KERNEL(k_cvtpkbf, "v_cvt_pk_bf16_f32 %0, %4, %5\n\tv_cvt_pk_bf16_f32 %1, %5, %6\n\tv_cvt_pk_bf16_f32 %2, %6, %7\n\tv_cvt_pk_bf16_f32 %3, %7, %4\n\t")
KERNEL(k_dot2c, "v_dot2c_f32_f16 %0, %8, %4\n\tv_dot2c_f32_f16 %1, %8, %5\n\tv_dot2c_f32_f16 %2, %9, %6\n\tv_dot2c_f32_f16 %3, %9, %7\n\t")
KERNEL(k_pkaddf16, "v_pk_add_f16 %0, %8, %4\n\tv_pk_add_f16 %1, %8, %5\n\tv_pk_add_f16 %2, %9, %6\n\tv_pk_add_f16 %3, %9, %7\n\t")
KERNEL(k_pkfmaf16, "v_pk_fma_f16 %0, %8, %4, %9\n\tv_pk_fma_f16 %1, %8, %5, %9\n\tv_pk_fma_f16 %2, %9, %6, %8\n\tv_pk_fma_f16 %3, %9, %7, %8\n\t")
KERNEL(k_max, "v_max_f32 %0, %8, %4\n\tv_max_f32 %1, %8, %5\n\tv_max_f32 %2, %9, %6\n\tv_max_f32 %3, %9, %7\n\t")
KERNEL(k_ldexp, "v_ldexp_f32 %0, %4, %8\n\tv_ldexp_f32 %1, %5, %8\n\tv_ldexp_f32 %2, %6, %9\n\tv_ldexp_f32 %3, %7, %9\n\t")
KERNEL(k_rndne, "v_rndne_f32 %0, %4\n\tv_rndne_f32 %1, %5\n\tv_rndne_f32 %2, %6\n\tv_rndne_f32 %3, %7\n\t")
KERNEL(k_mov, "v_mov_b32 %0, %4\n\tv_mov_b32 %1, %5\n\tv_mov_b32 %2, %6\n\tv_mov_b32 %3, %7\n\t")
KERNEL(k_perm, "v_perm_b32 %0, %4, %5, %8\n\tv_perm_b32 %1, %5, %6, %8\n\tv_perm_b32 %2, %6, %7, %9\n\tv_perm_b32 %3, %7, %4, %9\n\t")
KERNEL(k_cvtf16, "v_cvt_f16_f32 %0, %4\n\tv_cvt_f16_f32 %1, %5\n\tv_cvt_f16_f32 %2, %6\n\tv_cvt_f16_f32 %3, %7\n\t")
KERNEL(k_lshl_or, "v_lshl_or_b32 %0, %4, 16, %5\n\tv_lshl_or_b32 %1, %5, 16, %6\n\tv_lshl_or_b32 %2, %6, 16, %7\n\tv_lshl_or_b32 %3, %7, 16, %4\n\t")

// packed fp32 ops need 64-bit register pairs
__global__ void k_pkadd32(float* out, float seed) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 a0 = {seed, seed}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, b0 = a0 * 0.5f, b1 = a0 * 0.25f;
  unsigned long long t0, t1;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
  for (int it = 0; it < 64; ++it) {
#define PK "v_pk_add_f32 %0, %4, %0\n\tv_pk_add_f32 %1, %4, %1\n\tv_pk_add_f32 %2, %5, %2\n\tv_pk_add_f32 %3, %5, %3\n\t"
    asm volatile(REP8(PK) REP8(PK) REP8(PK) REP8(PK) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1));
#undef PK
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
  if ((threadIdx.x & 63) == 0) {
    atomicMin((unsigned long long*)(out + 2), t0);
    atomicMax((unsigned long long*)(out + 4), t1);
  }
  if (a0[0] + a1[0] + a2[1] + a3[1] == 12345.f) out[1] = a0[0];
}
__global__ void k_pkfma32(float* out, float seed) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 a0 = {seed, seed}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, b0 = a0 * 0.5f, b1 = a0 * 0.25f;
  unsigned long long t0, t1;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
  for (int it = 0; it < 64; ++it) {
#define PK "v_pk_fma_f32 %0, %4, %0, %5\n\tv_pk_fma_f32 %1, %4, %1, %5\n\tv_pk_fma_f32 %2, %5, %2, %4\n\tv_pk_fma_f32 %3, %5, %3, %4\n\t"
    asm volatile(REP8(PK) REP8(PK) REP8(PK) REP8(PK) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1));
#undef PK
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
  if ((threadIdx.x & 63) == 0) {
    atomicMin((unsigned long long*)(out + 2), t0);
    atomicMax((unsigned long long*)(out + 4), t1);
  }
  if (a0[0] + a1[0] + a2[1] + a3[1] == 12345.f) out[1] = a0[0];
}

int main() {
  float* d;
  (void)hipMalloc(&d, 32);   // [2..3] = first start (u64), [4..5] = last end (u64)
  struct K { const char* name; void (*fn)(float*, float); };
  K ks[] = {{"v_add_f32", k_add}, {"v_fma_f32", k_fma}, {"v_max_f32", k_max}, {"v_mov_b32", k_mov}, {"v_exp_f32", k_exp},
            {"v_exp_f16", k_exp16}, {"v_cvt_pk_f16_f32", k_cvtpk}, {"v_cvt_pk_bf16_f32", k_cvtpkbf}, {"v_cvt_f16_f32", k_cvtf16},
            {"v_dot2c_f32_f16", k_dot2c}, {"v_pk_add_f16", k_pkaddf16}, {"v_pk_fma_f16", k_pkfmaf16}, {"v_ldexp_f32", k_ldexp},
            {"v_rndne_f32", k_rndne}, {"v_perm_b32", k_perm}, {"v_lshl_or_b32", k_lshl_or}, {"v_pk_add_f32", k_pkadd32},
            {"v_pk_fma_f32", k_pkfma32}};
  printf("%-20s %10s %10s %10s   (cycles per wave-instruction, first start .. last end of all waves; 256/512/1024 threads = 1/2/4 waves per SIMD)\n", "instruction", "1 w/SIMD", "2 w/SIMD", "4 w/SIMD");
  for (auto& k : ks) {
    float r[3];
    int ti = 0;
    for (int threads : {256, 512, 1024}) {
      hipLaunchKernelGGL(k.fn, dim3(1), dim3(threads), 0, 0, d, 1.5f);   // warm-up
      unsigned long long init[4] = {0, ~0ull, 0, 0};
      (void)hipMemcpy(d, init, 32, hipMemcpyHostToDevice);
      hipLaunchKernelGGL(k.fn, dim3(1), dim3(threads), 0, 0, d, 1.5f);
      unsigned long long h[4];
      (void)hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
      r[ti++] = (float)(h[2] - h[1]) / (64.f * 128.f);
    }
    printf("%-20s %10.2f %10.2f %10.2f\n", k.name, r[0], r[1], r[2]);
  }
  return 0;
}
