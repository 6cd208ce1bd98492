// Probe of v_mfma_scale_f32_32x32x64_f8f6f4 (gfx950) with e4m3 operands, before using it for the correction products of
// the split attention (head dim 64 = ONE instruction per 32 keys x 32 queries):
//   operand map   H1: lane (row = l & 31, g = l >> 5) holds K bytes [16g, 16g+16) then [32+16g, 32+16g+16)
//                 H2: K bytes [32g, 32g+32)
//   block scales  lane row + 32 b supplies the e8m0 scale of K block [32b, 32b+32) of its row (b = 0, 1)
//   C/D           as the other 32x32 shapes: lane (col = l & 31, h = l >> 5), element e = row (e & 3) + 8 (e >> 2) + 4 h
//   rate          against v_mfma_f32_32x32x16_f16 (one wave, register operands)
// Build: hipcc --offload-arch=gfx950 -O2 tools/mfma_f8_32_probe.hip -o tools/mfma_f8_32_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));

static float e4m3_decode(uint8_t b) {
  const int s = b >> 7, e = (b >> 3) & 15, m = b & 7;
  float v = e == 0 ? ldexpf((float)m, -9) : ldexpf(1.0f + m / 8.0f, e - 7);
  return s ? -v : v;
}

__global__ void mfma_kernel(const int* A, const int* B, float* C, const int* sa, const int* sb) {
  const int lane = threadIdx.x;
  v8i a, b;
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = A[lane * 8 + i]; b[i] = B[lane * 8 + i]; }
  v16f c;
#pragma unroll
  for (int i = 0; i < 16; ++i) c[i] = 0.f;
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, sa[lane], 0, sb[lane]);
#pragma unroll
  for (int e = 0; e < 16; ++e) C[((e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)) * 32 + (lane & 31)] = c[e];
}

__global__ void rate_kernel(unsigned long long* out, int iters) {
  v8i a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {8, 7, 6, 5, 4, 3, 2, 1};
  v8h ha, hb;
  for (int i = 0; i < 8; ++i) { ha[i] = (_Float16)(i + threadIdx.x); hb[i] = (_Float16)(i - 3); }
  v16f c[4];
  for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) c[i][e] = 0.f;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) c[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c[i], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) c[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, c[i], 0, 0, 0);
  }
  unsigned long long t2 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < 4; ++i) s += c[i][0] + c[i][7];
  if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = t2 - t1; out[2] = (unsigned long long)s; }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

static int kmap(int hyp, int g, int j) { return hyp == 1 ? (j < 16 ? 16 * g + j : 32 + 16 * g + (j - 16)) : 32 * g + j; }

int main() {
  static uint8_t regA[64][32], regB[64][32];
  srand(5);
  for (int l = 0; l < 64; ++l)
    for (int j = 0; j < 32; ++j) {
      uint8_t a = rand() & 0xFF, b = rand() & 0xFF;
      if (((a >> 3) & 15) > 9) a &= ~0x40;   // moderate magnitudes, no NaN
      if (((b >> 3) & 15) > 9) b &= ~0x40;
      regA[l][j] = a; regB[l][j] = b;
    }
  int *dA, *dB, *dsa, *dsb; float* dC;
  CK(hipMalloc(&dA, sizeof(regA))); CK(hipMalloc(&dB, sizeof(regB))); CK(hipMalloc(&dC, 1024 * 4));
  CK(hipMalloc(&dsa, 64 * 4)); CK(hipMalloc(&dsb, 64 * 4));
  CK(hipMemcpy(dA, regA, sizeof(regA), hipMemcpyHostToDevice)); CK(hipMemcpy(dB, regB, sizeof(regB), hipMemcpyHostToDevice));
  int fails = 0;
  for (int test = 0; test < 2; ++test) {
    int sa[64], sb[64];
    for (int l = 0; l < 64; ++l) {
      sa[l] = test == 0 ? 127 : 0x11223300 | (121 + (l * 7) % 11);
      sb[l] = test == 0 ? 127 : 0x44556600 | (124 + (l * 5) % 7);
    }
    CK(hipMemcpy(dsa, sa, sizeof(sa), hipMemcpyHostToDevice)); CK(hipMemcpy(dsb, sb, sizeof(sb), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(mfma_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dC, dsa, dsb);
    float hC[1024];
    CK(hipMemcpy(hC, dC, sizeof(hC), hipMemcpyDeviceToHost));
    bool any = false;
    for (int hyp = 1; hyp <= 2; ++hyp) {
      static double Am[32][64], Bm[32][64];
      for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 32; ++j) {
          const int k = kmap(hyp, l >> 5, j), row = l & 31;
          Am[row][k] = e4m3_decode(regA[l][j]) * ldexp(1.0, (sa[row + 32 * (k / 32)] & 255) - 127);
          Bm[row][k] = e4m3_decode(regB[l][j]) * ldexp(1.0, (sb[row + 32 * (k / 32)] & 255) - 127);
        }
      double maxerr = 0, maxref = 0;
      for (int i = 0; i < 32; ++i)
        for (int n = 0; n < 32; ++n) {
          double ref = 0;
          for (int k = 0; k < 64; ++k) ref += Am[i][k] * Bm[n][k];
          maxref = fmax(maxref, fabs(ref));
          maxerr = fmax(maxerr, fabs(hC[i * 32 + n] - ref));
        }
      const bool ok = maxerr <= 1e-5 * maxref;
      any |= ok;
      printf("test %d (%s scales) K map H%d: max |C - ref| = %.3e of %.3e -> %s\n", test, test ? "per-lane" : "unit", hyp, maxerr,
             maxref, ok ? "MATCH" : "no");
    }
    if (!any) ++fails;
  }
  unsigned long long* dt;
  CK(hipMalloc(&dt, 3 * 8));
  hipLaunchKernelGGL(rate_kernel, dim3(1), dim3(64), 0, 0, dt, 2000);
  unsigned long long ht[3];
  CK(hipMemcpy(ht, dt, sizeof(ht), hipMemcpyDeviceToHost));
  printf("one wave, 4 accumulators (s_memtime ticks per MFMA): 32x32x64 e4m3 %.2f, 32x32x16 f16 %.2f\n", ht[0] / 8000.0, ht[1] / 8000.0);
  printf("%s\n", fails ? "PROBE FAILED" : "probe ok");
  return fails ? 1 : 0;
}
