// Relative error of v_exp_f32 (__builtin_amdgcn_exp2f) per unit interval of its argument, against exp2 in double.
// Build: hipcc --offload-arch=gfx950 -O2 tools/vexp_probe.hip -o tools/vexp_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
__global__ void k(const float* x, float* y, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) y[i] = __builtin_amdgcn_exp2f(x[i]);
}
int main() {
  const int per = 4096, lo = -24, hi = 12, n = per * (hi - lo);
  float* hx = new float[n]; float* hy = new float[n];
  for (int u = lo; u < hi; ++u)
    for (int j = 0; j < per; ++j) hx[(u - lo) * per + j] = (float)u + (j + 0.37f) / per;
  float *dx, *dy;
  hipMalloc(&dx, n * 4); hipMalloc(&dy, n * 4);
  hipMemcpy(dx, hx, n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3((n + 255) / 256), dim3(256), 0, 0, dx, dy, n);
  hipMemcpy(hy, dy, n * 4, hipMemcpyDeviceToHost);
  for (int u = lo; u < hi; ++u) {
    double worst = 0, rms = 0;
    for (int j = 0; j < per; ++j) {
      const int i = (u - lo) * per + j;
      const double r = exp2((double)hx[i]), e = fabs((double)hy[i] - r) / r;
      worst = fmax(worst, e); rms += e * e;
    }
    printf("x in [%3d, %3d): max rel err %.3e (%.2f ulp), rms %.3e\n", u, u + 1, worst, worst / 5.96e-8, sqrt(rms / per));
  }
  return 0;
}
