// Probe of v_mfma_scale_f32_16x16x128_f8f6f4 (gfx950) with e4m3 operands, before building a kernel on it:
//   1. operand lane map hypothesis: lane l holds row (l & 15), K bytes [32*(l>>4), 32*(l>>4)+32) of A and of B^T;
//      C/D as the other 16x16 shapes (col = l & 15, row = 4*(l>>4) + j)
//   2. block-scale semantics: the scale byte of a lane (e8m0, 127 = 1.0) multiplies that lane's 32-element K block;
//      opsel picks the byte of the scale register
//   3. f32 -> e4m3 conversion: __builtin_amdgcn_cvt_pk_fp8_f32 rounding and overflow behaviour
//   4. issue rate against v_mfma_f32_16x16x32_f16 (one wave, register operands)
// Build: hipcc --offload-arch=gfx950 -O2 tools/mfma_f8_probe.hip -o tools/mfma_f8_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));

static float e4m3_decode(uint8_t b) {   // OCP e4m3fn: bias 7, no inf, 0x7F/0xFF = NaN
  const int s = b >> 7, e = (b >> 3) & 15, m = b & 7;
  float v;
  if (e == 0) v = ldexpf((float)m, -9);
  else if (e == 15 && m == 7) v = NAN;
  else v = ldexpf(1.0f + m / 8.0f, e - 7);
  return s ? -v : v;
}

__global__ void mfma_kernel(const uint8_t* A, const uint8_t* Bt, float* C, const int* sa, const int* sb, int opsel) {
  const int lane = threadIdx.x, row = lane & 15, g = lane >> 4;
  v8i a, b;
  const int* pa = (const int*)(A + row * 128 + 32 * g);
  const int* pb = (const int*)(Bt + row * 128 + 32 * g);
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = pa[i]; b[i] = pb[i]; }
  v4f c = {0.f, 0.f, 0.f, 0.f};
  if (opsel == 0) c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, sa[lane], 0, sb[lane]);
  else c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 1, sa[lane], 1, sb[lane]);
#pragma unroll
  for (int j = 0; j < 4; ++j) C[(4 * g + j) * 16 + row] = c[j];   // C[row i][col n]: A row i = 4g+j?  see host check
}

__global__ void cvt_kernel(const float* in, uint32_t* out, int n) {
  const int i = threadIdx.x;
  if (2 * i + 1 < n + 1) {
    const float a = in[2 * i], b = in[2 * i + 1];
    out[i] = (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
  }
}

__global__ void rate_kernel(unsigned long long* out, int iters) {
  v8i a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {8, 7, 6, 5, 4, 3, 2, 1};
  v8h ha, hb;
  for (int i = 0; i < 8; ++i) { ha[i] = (_Float16)(i + threadIdx.x); hb[i] = (_Float16)(i - 3); }
  v4f c[8];
  for (int i = 0; i < 8; ++i) c[i] = (v4f){0.f, 0.f, 0.f, 0.f};
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c[i], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, c[i], 0, 0, 0);
  }
  unsigned long long t2 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += c[i][0] + c[i][3];
  if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = t2 - t1; out[2] = (unsigned long long)s; }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

int main() {
  uint8_t hA[16 * 128], hB[16 * 128];   // A[i][k], Bt[n][k]
  srand(7);
  for (int i = 0; i < 16 * 128; ++i) {
    uint8_t a = rand() & 0xFF, b = rand() & 0xFF;
    if ((a & 0x7F) == 0x7F) a ^= 1;   // no NaN
    if ((b & 0x7F) == 0x7F) b ^= 1;
    // keep magnitudes small so that the fp32 accumulation is exact enough: exponent field <= 9
    if (((a >> 3) & 15) > 9) a &= ~0x40;
    if (((b >> 3) & 15) > 9) b &= ~0x40;
    hA[i] = a; hB[i] = b;
  }
  uint8_t *dA, *dB; float* dC; int *dsa, *dsb;
  CK(hipMalloc(&dA, sizeof(hA))); CK(hipMalloc(&dB, sizeof(hB))); CK(hipMalloc(&dC, 256 * 4));
  CK(hipMalloc(&dsa, 64 * 4)); CK(hipMalloc(&dsb, 64 * 4));
  CK(hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice)); CK(hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice));
  int fails = 0;
  for (int test = 0; test < 4; ++test) {
    int sa[64], sb[64];
    double blockscale_a[4] = {1, 1, 1, 1}, blockscale_b[4] = {1, 1, 1, 1};
    for (int l = 0; l < 64; ++l) { sa[l] = 0x7F7F7F7F; sb[l] = 0x7F7F7F7F; }
    int opsel = 0;
    if (test == 1) {   // A's K block 1 scaled by 2^3
      for (int l = 16; l < 32; ++l) sa[l] = 0x7F7F7F00 | 130;
      blockscale_a[1] = 8;
    } else if (test == 2) {   // B's K block 2 scaled by 2^-2, A's block 3 by 2^5
      for (int l = 32; l < 48; ++l) sb[l] = 0x7F7F7F00 | 125;
      for (int l = 48; l < 64; ++l) sa[l] = 0x7F7F7F00 | 132;
      blockscale_b[2] = 0.25; blockscale_a[3] = 32;
    } else if (test == 3) {   // opsel = 1: byte 1 of the scale register
      opsel = 1;
      for (int l = 0; l < 16; ++l) sa[l] = 0x7F7F0000 | (129 << 8) | 0x11;   // byte 1 = 129 (x4), byte 0 = garbage
      for (int l = 16; l < 64; ++l) sa[l] = 0x7F7F7F11;
      for (int l = 0; l < 64; ++l) sb[l] = 0x7F7F7F33;
      blockscale_a[0] = 4;
    }
    CK(hipMemcpy(dsa, sa, sizeof(sa), hipMemcpyHostToDevice)); CK(hipMemcpy(dsb, sb, sizeof(sb), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(mfma_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dC, dsa, dsb, opsel);
    float hC[256];
    CK(hipMemcpy(hC, dC, sizeof(hC), hipMemcpyDeviceToHost));
    // the kernel wrote c[j] of lane (row r, group g) to C[(4g+j)*16 + r]: if D follows the standard map
    // (col = lane & 15, row = 4*(lane>>4) + j) then hC[i*16 + n] = sum_k A[i][k] * Bt[n][k]
    double maxerr = 0, maxref = 0;
    double maxerr_T = 0;   // the transposed interpretation, for the record
    for (int i = 0; i < 16; ++i)
      for (int n = 0; n < 16; ++n) {
        double ref = 0;
        for (int k = 0; k < 128; ++k)
          ref += (double)e4m3_decode(hA[i * 128 + k]) * e4m3_decode(hB[n * 128 + k]) * blockscale_a[k / 32] * blockscale_b[k / 32];
        maxref = fmax(maxref, fabs(ref));
        maxerr = fmax(maxerr, fabs(hC[i * 16 + n] - ref));
        maxerr_T = fmax(maxerr_T, fabs(hC[n * 16 + i] - ref));
      }
    printf("test %d: max |C - ref| = %.3e (transposed reading %.3e), max |ref| = %.3e -> %s\n", test, maxerr, maxerr_T,
           maxref, maxerr <= 1e-5 * maxref ? "OK" : "MISMATCH");
    if (!(maxerr <= 1e-5 * maxref)) ++fails;
  }
  // conversions
  const float vals[16] = {0.0f, 1.0f, -3.3f, 0.001f, 0.0019f, 447.0f, 448.0f, 449.0f, 464.0f, 500.0f, 1000.0f, -1e6f, 0.0625f,
                          0.0146f, 17.5f, 1.0625f};
  float* dv; uint32_t* dout;
  CK(hipMalloc(&dv, sizeof(vals))); CK(hipMalloc(&dout, 8 * 4));
  CK(hipMemcpy(dv, vals, sizeof(vals), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(cvt_kernel, dim3(1), dim3(8), 0, 0, dv, dout, 16);
  uint32_t hout[8];
  CK(hipMemcpy(hout, dout, sizeof(hout), hipMemcpyDeviceToHost));
  for (int i = 0; i < 16; ++i) {
    const uint8_t b = (hout[i / 2] >> (8 * (i & 1))) & 0xFF;
    printf("cvt_pk_fp8_f32(%g) = 0x%02x = %g\n", vals[i], b, e4m3_decode(b));
  }
  // rate
  unsigned long long* dt;
  CK(hipMalloc(&dt, 3 * 8));
  hipLaunchKernelGGL(rate_kernel, dim3(1), dim3(64), 0, 0, dt, 2000);
  unsigned long long ht[3];
  CK(hipMemcpy(ht, dt, sizeof(ht), hipMemcpyDeviceToHost));
  printf("one wave, 8 accumulators: %.1f cycles per 16x16x128 e4m3 MFMA, %.1f per 16x16x32 f16 MFMA (s_memtime ticks)\n",
         ht[0] / 16000.0, ht[1] / 16000.0);
  printf("%s\n", fails ? "PROBE FAILED" : "probe ok");
  return fails ? 1 : 0;
}
