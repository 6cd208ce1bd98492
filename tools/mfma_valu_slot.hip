// How much VALU issue fits in the shadow of an MFMA on gfx950?  One wave per SIMD (256 threads) or two (512): a loop of
// slots [v_mfma_f32_32x32x16_f16 ; N independent VALU instructions], 4 accumulator tuples in rotation (no dependent
// MFMAs closer than 4 apart).  Prints s_memtime cycles per slot: full overlap = max(32, ~4 + cost(N)), none = 32 + cost(N).
//   hipcc --offload-arch=gfx950 -O2 -o tools/mfma_valu_slot tools/mfma_valu_slot.hip && tools/mfma_valu_slot
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

#define V1(OP) OP " %4, %8, %4\n\t"
#define V2(OP) V1(OP) OP " %5, %8, %5\n\t"
#define V4(OP) V2(OP) OP " %6, %9, %6\n\t" OP " %7, %9, %7\n\t"
#define V6(OP) V4(OP) OP " %4, %9, %4\n\t" OP " %5, %9, %5\n\t"
#define V8(OP) V4(OP) V4(OP)
#define E2 "v_exp_f32 %4, %4\n\tv_exp_f32 %5, %5\n\t"
#define E4 E2 "v_exp_f32 %6, %6\n\tv_exp_f32 %7, %7\n\t"
#define SLOT(M, V) "v_mfma_f32_32x32x16_f16 " M ", %10, %11, " M "\n\t" V
#define FOUR(V) SLOT("%0", V) SLOT("%1", V) SLOT("%2", V) SLOT("%3", V)

#define KERNEL(NAME, V)                                                                                      \
  __global__ void NAME(float* out, float seed) {                                                             \
    f16v c0, c1, c2, c3;                                                                                     \
    for (int e = 0; e < 16; ++e) { c0[e] = seed; c1[e] = seed; c2[e] = seed; c3[e] = seed; }                 \
    float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, b0 = 0.5f * seed, b1 = 0.25f * seed;        \
    h8 fa, fb;                                                                                               \
    for (int j = 0; j < 8; ++j) { fa[j] = (_Float16)(seed + j + threadIdx.x % 7); fb[j] = (_Float16)(seed - j); } \
    unsigned long long t0, t1;                                                                               \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory"); \
    for (int it = 0; it < 256; ++it) {                                                                       \
      asm volatile(FOUR(V) FOUR(V)                                                                           \
                   : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)           \
                   : "v"(b0), "v"(b1), "v"(fa), "v"(fb));                                                    \
    }                                                                                                        \
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");      \
    /* issue arbitration is oldest-first: wave 0 alone would never see its SIMD partner; take first start .. last end */ \
    if ((threadIdx.x & 63) == 0) {                                                                           \
      atomicMin((unsigned long long*)(out + 2), t0);                                                         \
      atomicMax((unsigned long long*)(out + 4), t1);                                                         \
    }                                                                                                        \
    if (c0[0] + c1[1] + c2[2] + c3[3] + a0 + a1 + a2 + a3 == 12345.f) out[1] = a0;                            \
  }

#define FOUR_D1(V) SLOT("%0", V) SLOT("%0", V) SLOT("%0", V) SLOT("%0", V)
#define FOUR_D2(V) SLOT("%0", V) SLOT("%1", V) SLOT("%0", V) SLOT("%1", V)
#define KERNEL_D(NAME, FOURX, V)                                                                             \
  __global__ void NAME(float* out, float seed) {                                                             \
    f16v c0, c1, c2, c3;                                                                                     \
    for (int e = 0; e < 16; ++e) { c0[e] = seed; c1[e] = seed; c2[e] = seed; c3[e] = seed; }                 \
    float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, b0 = 0.5f * seed, b1 = 0.25f * seed;        \
    h8 fa, fb;                                                                                               \
    for (int j = 0; j < 8; ++j) { fa[j] = (_Float16)(0.01f * (j + threadIdx.x % 7)); fb[j] = (_Float16)(0.01f * j); } \
    unsigned long long t0, t1;                                                                               \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory"); \
    for (int it = 0; it < 256; ++it) {                                                                       \
      asm volatile(FOURX(V) FOURX(V)                                                                         \
                   : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)           \
                   : "v"(b0), "v"(b1), "v"(fa), "v"(fb));                                                    \
    }                                                                                                        \
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");      \
    if ((threadIdx.x & 63) == 0) {                                                                           \
      atomicMin((unsigned long long*)(out + 2), t0);                                                         \
      atomicMax((unsigned long long*)(out + 4), t1);                                                         \
    }                                                                                                        \
    if (c0[0] + c1[1] + c2[2] + c3[3] + a0 + a1 + a2 + a3 == 12345.f) out[1] = a0;                            \
  }
KERNEL_D(k_dep1, FOUR_D1, "")
KERNEL_D(k_dep2, FOUR_D2, "")
KERNEL_D(k_dep1v, FOUR_D1, E2 "v_add_f32 %6, %8, %6\n\tv_add_f32 %7, %9, %7\n\t")
KERNEL_D(k_dep2v, FOUR_D2, E2 "v_add_f32 %6, %8, %6\n\tv_add_f32 %7, %9, %7\n\t")
KERNEL(k_mixi, E2 "v_cvt_pk_f16_f32 %6, %6, %7\n\tv_dot2c_f32_f16 %7, %8, %9\n\t")
KERNEL(k_cvt, E2 "v_cvt_pk_f16_f32 %6, %6, %7\n\t")
KERNEL(k_dot, E2 "v_dot2c_f32_f16 %7, %8, %9\n\t")
KERNEL(k_full_add, E2 "v_cvt_pk_f16_f32 %6, %6, %7\n\tv_add_f32 %6, %8, %6\n\tv_add_f32 %7, %9, %7\n\t")
KERNEL(k_exp2add3, E2 "v_add_f32 %6, %8, %6\n\tv_add_f32 %7, %9, %7\n\tv_add_f32 %6, %9, %6\n\t")
KERNEL(k_pkf16, E2 "v_cvt_pk_f16_f32 %6, %6, %7\n\tv_pk_add_f16 %7, %8, %7\n\t")
KERNEL(k_m0, "")
KERNEL(k_add2, V2("v_add_f32"))
KERNEL(k_add4, V4("v_add_f32"))
KERNEL(k_add6, V6("v_add_f32"))
KERNEL(k_add8, V8("v_add_f32"))
KERNEL(k_exp2, E2)
KERNEL(k_exp4, E4)
KERNEL(k_exp2add2, E2 "v_add_f32 %6, %8, %6\n\tv_add_f32 %7, %9, %7\n\t")
KERNEL(k_mix, E2 "v_cvt_pk_f16_f32 %6, %6, %7\n\tv_dot2c_f32_f16 %7, %8, %6\n\t")

__global__ void k_pk32(float* out, float seed) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  f16v c0, c1, c2, c3;
  for (int e = 0; e < 16; ++e) { c0[e] = seed; c1[e] = seed; c2[e] = seed; c3[e] = seed; }
  float a0 = seed, a1 = seed + 1;
  f2 p0 = {seed, seed}, p1 = {seed, seed}, q0 = {0.5f * seed, seed};
  h8 fa, fb;
  for (int j = 0; j < 8; ++j) { fa[j] = (_Float16)(0.01f * (j + threadIdx.x % 7)); fb[j] = (_Float16)(0.01f * j); }
  unsigned long long t0, t1;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
  for (int it = 0; it < 256; ++it) {
#define PSLOT(M, P) "v_mfma_f32_32x32x16_f16 " M ", %9, %10, " M "\n\tv_exp_f32 %4, %4\n\tv_exp_f32 %5, %5\n\tv_pk_add_f32 " P ", %8, " P "\n\t"
#define PFOUR PSLOT("%0", "%6") PSLOT("%1", "%7") PSLOT("%2", "%6") PSLOT("%3", "%7")
    asm volatile(PFOUR PFOUR
                 : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(a0), "+v"(a1), "+v"(p0), "+v"(p1)
                 : "v"(q0), "v"(fa), "v"(fb));
  }
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
  if ((threadIdx.x & 63) == 0) {
    atomicMin((unsigned long long*)(out + 2), t0);
    atomicMax((unsigned long long*)(out + 4), t1);
  }
  if (c0[0] + c1[1] + c2[2] + c3[3] + a0 + a1 + p0[0] + p1[1] == 12345.f) out[1] = a0;
}

int main() {
  float* d;   // [0..1] scratch, [2..3] = min start (u64), [4..5] = max end (u64)
  (void)hipMalloc(&d, 32);
  struct K { const char* name; void (*fn)(float*, float); };
  K ks[] = {{"mfma only", k_m0}, {"mfma + 2 v_add", k_add2}, {"mfma + 4 v_add", k_add4}, {"mfma + 6 v_add", k_add6},
            {"mfma + 8 v_add", k_add8}, {"mfma + 2 v_exp", k_exp2}, {"mfma + 4 v_exp", k_exp4},
            {"mfma + 2 v_exp + 2 v_add", k_exp2add2}, {"mfma + 2 v_exp + cvt_pk + dot2c", k_mix},
            {"mfma + 2 v_exp + cvt_pk + dot2c indep", k_mixi}, {"mfma + 2 v_exp + cvt_pk", k_cvt}, {"mfma + 2 v_exp + dot2c", k_dot},
            {"mfma + 2 v_exp + cvt_pk + 2 v_add", k_full_add}, {"mfma + 2 v_exp + 3 v_add", k_exp2add3},
            {"mfma + 2 v_exp + cvt_pk + v_pk_add_f16", k_pkf16},
            {"mfma + 2 v_exp + v_pk_add_f32", k_pk32},
            {"dependent mfma, distance 1", k_dep1}, {"dependent mfma, distance 2", k_dep2},
            {"dep. distance 1 + 2 exp + 2 add", k_dep1v}, {"dep. distance 2 + 2 exp + 2 add", k_dep2v}};
  printf("%-40s %12s %12s   (cycles per slot: first start .. last end of all waves / slots per wave)\n", "slot", "1 wave/SIMD", "2 waves/SIMD");
  for (auto& k : ks) {
    float r[2];
    int ti = 0;
    for (int threads : {256, 512}) {
      hipLaunchKernelGGL(k.fn, dim3(1), dim3(threads), 0, 0, d, 1.5f);   // warm-up
      unsigned long long init[4] = {0, ~0ull, 0, 0};
      (void)hipMemcpy(d, init, 32, hipMemcpyHostToDevice);
      hipLaunchKernelGGL(k.fn, dim3(1), dim3(threads), 0, 0, d, 1.5f);
      unsigned long long h[4];
      (void)hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
      r[ti++] = (float)(h[2] - h[1]) / (256.f * 8.f);
    }
    printf("%-40s %12.1f %12.1f\n", k.name, r[0], r[1]);
  }
  return 0;
}
