// Do v_cvt_f16_f32 and v_cvt_pk_f16_f32 (gfx950) round alike?  hipcc --offload-arch=gfx950 -O3 -o tools/cvt_pk_probe tools/cvt_pk_probe.hip
// Every f32 bit pattern of a coarse sweep plus the neighbourhood of every f16 value (half-way cases, f16 denormals, overflow),
// with MODE.FP16_OVFL clear and set (the split kernels set it: common.h fp8_saturate_mode).  Prints the classes that differ.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
#include <string.h>
#include <vector>
typedef _Float16 f16;
typedef f16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__global__ void k(const float* in, unsigned short* a, unsigned short* b, int n, int ovfl) {
  if (ovfl) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1");
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float v = in[i];
  unsigned s, p;
  asm volatile("v_cvt_f16_f32_e32 %0, %1" : "=v"(s) : "v"(v));
  asm volatile("v_cvt_pk_f16_f32 %0, %1, %1" : "=v"(p) : "v"(v));
  a[i] = (unsigned short)s;
  b[i] = (unsigned short)p;
}
int main() {
  std::vector<float> h;
  for (unsigned e = 0; e < 65536; ++e) {            // around every f16 value: the value, +-1..3 ulp of f32, and the half-way points
    f16 x; unsigned short bits = (unsigned short)e; memcpy(&x, &bits, 2);
    float f = (float)x;
    if (!isfinite(f)) continue;
    unsigned u; memcpy(&u, &f, 4);
    for (int d = -3; d <= 3; ++d) { unsigned w = u + d; float g; memcpy(&g, &w, 4); h.push_back(g); }
    f16 y; unsigned short b2 = (unsigned short)(e + 1); memcpy(&y, &b2, 2);
    float fy = (float)y;
    if (isfinite(fy)) { float mid = 0.5f * (f + fy); unsigned m; memcpy(&m, &mid, 4);
      for (int d = -2; d <= 2; ++d) { unsigned w = m + d; float g; memcpy(&g, &w, 4); h.push_back(g); } }
  }
  for (unsigned u = 0; u < 0xFFFFFFFFu - 65537u; u += 65537u) { float g; memcpy(&g, &u, 4); h.push_back(g); }
  for (float t : {1e-8f, 3e-8f, 5.9e-8f, 6e-8f, 1e-7f, 1e-6f, 1e-5f, 6.0e-5f, 6.1e-5f, 6.2e-5f, 65504.f, 65519.f, 65520.f, 70000.f, 1e6f}) { h.push_back(t); h.push_back(-t); }
  const int n = (int)h.size();
  float* din; unsigned short *da, *db;
  hipMalloc(&din, n * 4); hipMalloc(&da, n * 2); hipMalloc(&db, n * 2);
  hipMemcpy(din, h.data(), n * 4, hipMemcpyHostToDevice);
  std::vector<unsigned short> a(n), b(n);
  for (int ovfl = 0; ovfl < 2; ++ovfl) {
    k<<<(n + 255) / 256, 256>>>(din, da, db, n, ovfl);
    hipMemcpy(a.data(), da, n * 2, hipMemcpyDeviceToHost);
    hipMemcpy(b.data(), db, n * 2, hipMemcpyDeviceToHost);
    long diff = 0, diff_den = 0, diff_ovf = 0, diff_other = 0; int shown = 0;
    for (int i = 0; i < n; ++i) {
      if (a[i] == b[i]) continue;
      ++diff;
      const float av = fabsf(h[i]);
      if (av < 6.104e-5f) ++diff_den; else if (av > 65504.f || !isfinite(h[i])) ++diff_ovf; else ++diff_other;
      if (shown < 12) { printf("  FP16_OVFL=%d  v = %.9g (0x%08x): v_cvt_f16_f32 -> 0x%04x, v_cvt_pk_f16_f32 -> 0x%04x\n", ovfl, h[i], *(unsigned*)&h[i], a[i], b[i]); ++shown; }
    }
    printf("FP16_OVFL=%d: %d inputs, %ld differ (|v| below the smallest normal f16: %ld, beyond 65504 / non-finite: %ld, normal range: %ld)\n",
           ovfl, n, diff, diff_den, diff_ovf, diff_other);
  }
  return 0;
}
