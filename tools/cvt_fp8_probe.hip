// Probe of the scaled fp8 conversions of gfx950 (v_cvt_scalef32_pk_fp8_f32 / _f16), before using them in the split8
// producers instead of {v_mul, v_med3 (clamp to +-448), v_cvt_pk_fp8_f32}: what the scale operand does, the rounding,
// and what happens beyond +-448 (v_cvt_pk_fp8_f32 returns NaN there, tools/mfma_f8_probe.hip).
// Build: hipcc --offload-arch=gfx950 -O2 tools/cvt_fp8_probe.hip -o tools/cvt_fp8_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
typedef short v2s __attribute__((ext_vector_type(2)));
typedef _Float16 v2h __attribute__((ext_vector_type(2)));
static float e4m3_decode(uint8_t b) {
  const int s = b >> 7, e = (b >> 3) & 15, m = b & 7;
  float v;
  if (e == 0) v = ldexpf((float)m, -9);
  else if (e == 15 && m == 7) v = NAN;
  else v = ldexpf(1.0f + m / 8.0f, e - 7);
  return s ? -v : v;
}
__global__ void k(const float* in, float scale, uint32_t* out32, uint32_t* out16, uint32_t* outplain, int ovfl) {
  const int i = threadIdx.x;
  // MODE.FP16_OVFL (bit 23): "an overflowed fp16 result is clamped" -- does it make the fp8 conversions saturate too?
  if (ovfl) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1");
  const float a = in[2 * i], b = in[2 * i + 1];
  v2s z = {0, 0};
  v2s r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(z, a, b, scale, false);
  out32[i] = (uint16_t)r[0];
  v2h h = {(_Float16)a, (_Float16)b};
  v2s r2 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(z, h, scale, false);
  out16[i] = (uint16_t)r2[0];
  outplain[i] = (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(a / scale, b / scale, 0, false) & 0xFFFF;
}
int main() {
  const float vals[32] = {0.f, 1.f, -3.3f, 0.001f, 0.0019f, 447.f, 448.f, 449.f, 464.f, 480.f, 500.f, 1000.f, -1e6f, 0.0625f, 0.0146f, 17.5f,
                          1.0625f, 1.1875f, 9.5f, 10.5f, 3e-4f, -2e-3f, 60000.f, -470.f, 0.0009765625f, 0.00146484375f, 208.f, 216.f, 232.f, 1e-5f, 65000.f, 7.f};
  float* din; uint32_t *d32, *d16, *dpl;
  hipMalloc(&din, sizeof(vals)); hipMalloc(&d32, 64); hipMalloc(&d16, 64); hipMalloc(&dpl, 64);
  hipMemcpy(din, vals, sizeof(vals), hipMemcpyHostToDevice);
  for (int pass = 0; pass < 5; ++pass) {
    const float scale = (pass == 0 || pass == 3) ? 1.f : (pass == 1 || pass == 4) ? 0.0009765625f : 4.f;
    hipLaunchKernelGGL(k, dim3(1), dim3(16), 0, 0, din, scale, d32, d16, dpl, pass >= 3 ? 1 : 0);
    uint32_t h32[16], h16[16], hpl[16];
    hipMemcpy(h32, d32, 64, hipMemcpyDeviceToHost); hipMemcpy(h16, d16, 64, hipMemcpyDeviceToHost); hipMemcpy(hpl, dpl, 64, hipMemcpyDeviceToHost);
    printf("scale operand %g%s:\n", scale, pass >= 3 ? ", MODE.FP16_OVFL = 1" : "");
    for (int i = 0; i < 32; ++i) {
      const uint8_t a = (h32[i / 2] >> (8 * (i & 1))) & 0xFF, b = (h16[i / 2] >> (8 * (i & 1))) & 0xFF, c = (hpl[i / 2] >> (8 * (i & 1))) & 0xFF;
      printf("  %-14g scalef32_f32 -> 0x%02x = %-10g  scalef32_f16 -> 0x%02x = %-10g  cvt_pk_fp8_f32(v/scale) -> 0x%02x = %g\n", vals[i], a,
             e4m3_decode(a), b, e4m3_decode(b), c, e4m3_decode(c));
    }
  }
  return 0;
}
