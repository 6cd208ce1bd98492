// How long does workgroup turnover take?  5632 workgroups of 512 threads with 128 KiB of LDS
// (one per CU at a time) that do almost nothing, vs the same with 64 KiB / 0 KiB.
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int LDS>
__global__ __launch_bounds__(512, 2) void k(float* out, int n) {
  __shared__ float smem[LDS / 4 + 1];
  asm volatile("v_mov_b32 v230, 0" ::: "v230");   // force a ~232-VGPR allocation like the GEMM kernel
  smem[threadIdx.x] = (float)blockIdx.x;
  __syncthreads();
  if (n == 12345) out[blockIdx.x] = smem[(threadIdx.x + 1) & 511];
}
template <int LDS>
void run(const char* name, float* o) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<LDS>, dim3(5632), dim3(512), 0, 0, o, 0);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%s: %.3f ms for 5632 workgroups = %.2f us per round of 256\n", name, ms, ms * 1e3 / 22.0);
  }
}
int main() {
  float* o; (void)hipMalloc(&o, 5632 * 4);
  run<131072>("128 KiB LDS", o);
  run<65536>("64 KiB LDS", o);
  run<2048>("2 KiB LDS", o);
  return 0;
}
