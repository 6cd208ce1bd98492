// Practical MFMA ceiling on this chip: register-only MFMA loops on random data
// (no LDS, no global traffic), 8 waves per CU, both f16 shapes.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(512, 2) void k16(const _Float16* in, float* out, int iters) {
  f16x8 a[8], b[4];
  for (int i = 0; i < 8; ++i) a[i] = *(const f16x8*)(in + ((threadIdx.x * 8 + i) % 4096) * 8);
  for (int i = 0; i < 4; ++i) b[i] = *(const f16x8*)(in + ((threadIdx.x * 4 + i + 77) % 4096) * 8);
  f32x4 acc[8][4];
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[j], a[i], acc[i][j], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][3];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}
__global__ __launch_bounds__(512, 2) void k32(const _Float16* in, float* out, int iters) {
  f16x8 a[4], b[2];
  for (int i = 0; i < 4; ++i) a[i] = *(const f16x8*)(in + ((threadIdx.x * 8 + i) % 4096) * 8);
  for (int i = 0; i < 2; ++i) b[i] = *(const f16x8*)(in + ((threadIdx.x * 4 + i + 77) % 4096) * 8);
  f32x16 acc[4][2];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], acc[i][j], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) s += acc[i][j][0] + acc[i][j][7];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}
int main() {
  const int n = 4096 * 8;
  _Float16* h = (_Float16*)malloc(n * 2);
  srand(1);
  for (int i = 0; i < n; ++i) h[i] = (_Float16)((rand() / (float)RAND_MAX) * 2 - 1);
  _Float16* d; float* o;
  hipMalloc(&d, n * 2); hipMalloc(&o, 256 * 512 * 4 * 4);
  hipMemcpy(d, h, n * 2, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 4000;
  for (int which = 0; which < 2; ++which) {
    for (int rep = 0; rep < 4; ++rep) {
      hipEventRecord(e0);
      if (which == 0) hipLaunchKernelGGL(k16, dim3(256), dim3(512), 0, 0, d, o, iters);
      else hipLaunchKernelGGL(k32, dim3(256), dim3(512), 0, 0, d, o, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      // flops: 16x16x32: 64 mfma * 2*16*16*32 per iter per wave; 32x32x16: 32 mfma * 2*32*32*16
      double fl = (double)256 * 8 * iters * (which == 0 ? 64.0 * 16384 : 32.0 * 32768);
      printf("%s rep %d: %.3f ms -> %.0f TFLOP/s\n", which == 0 ? "16x16x32" : "32x32x16", rep, ms, fl / ms / 1e9);
    }
  }
  return 0;
}
