// How much VALU work (softmax-like: v_exp_f32 + FMAs) hides under MFMA work on one SIMD, and in
// which arrangement: the question behind the attention kernel's "VALU busy 48 % + MFMA busy 28 %".
// Register-only loops, 256 threads per workgroup (1 wave per SIMD) with 1 or 2 workgroups per CU.
//   mode 0: MFMA block only          (32 x 32x32x16 per iteration)
//   mode 1: VALU block only          (64 v_exp_f32 + 128 v_fma_f32 per iteration)
//   mode 2: MFMA block then VALU block in the same wave (what a straightforward kernel does)
//   mode 3: the two blocks interleaved instruction by instruction in one wave
// Build: hipcc --offload-arch=gfx950 -O3 tools/valu_mfma_overlap.hip -o tools/valu_mfma_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define MFMA(i) acc[(i) & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(i) & 1], b[((i) >> 1) & 1], acc[(i) & 3], 0, 0, 0);
#define MFMA_A(i) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[(i) & 3]) : "v"(a[(i) & 1]), "v"(b[((i) >> 1) & 1]));
#define VALU2(j)                                               \
  v[(j) & 15] = __builtin_amdgcn_exp2f(v[(j) & 15]);            \
  w[(j) & 15] = __builtin_fmaf(w[(j) & 15], 0.999f, v[(j) & 15]); \
  w[((j) + 5) & 15] = __builtin_fmaf(w[((j) + 5) & 15], 1.001f, -0.5f);

template <int MODE>
__global__ __launch_bounds__(MODE >= 5 ? 512 : 256, 2) void k(const _Float16* in, float* out, int iters) {
  f16x8 a[2], b[2];
  for (int i = 0; i < 2; ++i) a[i] = *(const f16x8*)(in + ((threadIdx.x * 8 + i) % 4096) * 8);
  for (int i = 0; i < 2; ++i) b[i] = *(const f16x8*)(in + ((threadIdx.x * 4 + i + 77) % 4096) * 8);
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0;
  float v[16], w[16];
  for (int i = 0; i < 16; ++i) { v[i] = -(float)in[(threadIdx.x + i) % 4096]; w[i] = 0.25f * i; }
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0 || MODE == 2) {
#pragma unroll
      for (int i = 0; i < 32; ++i) MFMA(i)
    }
    if (MODE == 2) __builtin_amdgcn_sched_barrier(0);
    if (MODE == 1 || MODE == 2) {
#pragma unroll
      for (int j = 0; j < 64; ++j) { VALU2(j) }
    }
    if (MODE == 4) {   // as mode 2 with the accumulators in AGPRs
#pragma unroll
      for (int i = 0; i < 32; ++i) MFMA_A(i)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 64; ++j) { VALU2(j) }
    }
    if (MODE == 5 || MODE == 6) {   // barrier-synchronised anti-phase: waves 0-3 MFMA while waves 4-7 VALU, then swap
      const bool first = (threadIdx.x >> 8) == 0;
      if (first) {
#pragma unroll
        for (int i = 0; i < 32; ++i) { if (MODE == 5) { MFMA(i) } else { MFMA_A(i) } }
      } else {
#pragma unroll
        for (int j = 0; j < 64; ++j) { VALU2(j) }
      }
      __builtin_amdgcn_s_barrier();
      if (!first) {
#pragma unroll
        for (int i = 0; i < 32; ++i) { if (MODE == 5) { MFMA(i) } else { MFMA_A(i) } }
      } else {
#pragma unroll
        for (int j = 0; j < 64; ++j) { VALU2(j) }
      }
      __builtin_amdgcn_s_barrier();
    }
    if (MODE == 3) {
#pragma unroll
      for (int i = 0; i < 32; ++i) {
        MFMA(i)
        __builtin_amdgcn_sched_barrier(0);
        VALU2(2 * i) VALU2(2 * i + 1)
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = v[i] * 0.5f - 1.0f;   // keep exp inputs bounded
  }
  float s = 0;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][9];
  for (int i = 0; i < 16; ++i) s += v[i] + w[i];
  out[(blockIdx.x * 512 + threadIdx.x) % (1024 * 256)] = s;
}

int main() {
  const int n = 4096 * 8;
  _Float16* h = (_Float16*)malloc(n * 2);
  srand(1);
  for (int i = 0; i < n; ++i) h[i] = (_Float16)((rand() / (float)RAND_MAX) * 2 - 1);
  _Float16* d; float* o;
  hipMalloc(&d, n * 2); hipMalloc(&o, 1024 * 256 * 4);
  hipMemcpy(d, h, n * 2, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 2000;
  const char* names[7] = {"MFMA only", "VALU only", "MFMA block then VALU block", "interleaved in one wave",
                          "block then block, AGPR accumulators", "anti-phase by barrier (8-wave WG)", "anti-phase by barrier, AGPR acc"};
  for (int wg_per_cu = 1; wg_per_cu <= 2; ++wg_per_cu)
    for (int mode = 0; mode < 7; ++mode) {
      if (mode >= 5 && wg_per_cu == 2) continue;   // the 8-wave workgroup already puts 2 waves on every SIMD
      float best = 1e9;
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        const dim3 g(256 * wg_per_cu), b(mode >= 5 ? 512 : 256);
        if (mode == 0) hipLaunchKernelGGL(k<0>, g, b, 0, 0, d, o, iters);
        if (mode == 1) hipLaunchKernelGGL(k<1>, g, b, 0, 0, d, o, iters);
        if (mode == 2) hipLaunchKernelGGL(k<2>, g, b, 0, 0, d, o, iters);
        if (mode == 3) hipLaunchKernelGGL(k<3>, g, b, 0, 0, d, o, iters);
        if (mode == 4) hipLaunchKernelGGL(k<4>, g, b, 0, 0, d, o, iters);
        if (mode == 5) hipLaunchKernelGGL(k<5>, g, b, 0, 0, d, o, iters);
        if (mode == 6) hipLaunchKernelGGL(k<6>, g, b, 0, 0, d, o, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
      }
      // per iteration per wave: 32 MFMAs (32 x 32 cycles nominal = 1024), 64 exp + 128 fma
      printf("%d wave(s)/SIMD  %-30s %8.3f ms  -> %7.0f ns per iteration (per SIMD: %d iterations in flight)\n", wg_per_cu,
             names[mode], best, best * 1e6 / iters, wg_per_cu);
    }
  return 0;
}
