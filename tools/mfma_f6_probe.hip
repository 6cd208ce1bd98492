// Probe of v_mfma_scale_f32_16x16x128_f8f6f4 with fp6 (e2m3) operands on gfx950, before building a kernel on it:
//   1. operand lane map: which K elements the 6 data registers of lane (row = l & 15, g = l >> 4) hold
//        H1: K [32g, 32g+32), element j in bits [6j, 6j+6) of the 192-bit tuple
//        H2: the fp8 map (K [16g,16g+16) then [64+16g, 64+16g+16))
//   2. block scales: lane row + 16 b supplies the e8m0 scale of K block [32b, 32b+32) (as for fp8)
//   3. the packing conversion v_cvt_scalef32_2xpk16_fp6_f32 (32 floats of one lane -> 6 registers): element order,
//      rounding, what the scale operand does
//   4. issue rate: one wave, and the whole chip under the K-loop mixes of the split GEMM
//        mix8: 4 x 16x16x32 f16 + 2 x 16x16x128 e4m3     (today's fp16x2 GEMM)
//        mix6: 4 x 16x16x32 f16 + 2 x 16x16x128 e2m3
// Build: hipcc --offload-arch=gfx950 -O2 tools/mfma_f6_probe.hip -o tools/mfma_f6_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef int v6i __attribute__((ext_vector_type(6)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));

static float e2m3_decode(int c) {
  const int s = (c >> 5) & 1, e = (c >> 3) & 3, m = c & 7;
  const float v = e == 0 ? m / 8.0f : ldexpf(1.0f + m / 8.0f, e - 1);
  return s ? -v : v;
}

__global__ void mfma6_kernel(const int* A, const int* B, float* C, const int* sa, const int* sb) {
  const int lane = threadIdx.x;
  v8i a = {0, 0, 0, 0, 0, 0, 0, 0}, b = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < 6; ++i) { a[i] = A[lane * 6 + i]; b[i] = B[lane * 6 + i]; }
  v4f c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 2, 2, 0, sa[lane], 0, sb[lane]);   // cbsz 2, blgp 2: e2m3
#pragma unroll
  for (int j = 0; j < 4; ++j) C[(4 * (lane >> 4) + j) * 16 + (lane & 15)] = c[j];
}

// the same product from inline asm with 6-register operands (what the GEMM kernel would issue)
__global__ void mfma6_asm_kernel(const int* A, const int* B, float* C, const int* sa, const int* sb) {
  const int lane = threadIdx.x;
  v6i a, b;
#pragma unroll
  for (int i = 0; i < 6; ++i) { a[i] = A[lane * 6 + i]; b[i] = B[lane * 6 + i]; }
  v4f c = {0.f, 0.f, 0.f, 0.f};
  const int s0 = sa[lane], s1 = sb[lane];
  asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0] cbsz:2 blgp:2\n s_nop 7\n s_nop 7"
               : "+v"(c) : "v"(a), "v"(b), "v"(s0), "v"(s1));
#pragma unroll
  for (int j = 0; j < 4; ++j) C[(4 * (lane >> 4) + j) * 16 + (lane & 15)] = c[j];
}

__global__ void cvt6_kernel(const float* in, int* out, float scale) {
  v16f x, y;
#pragma unroll
  for (int i = 0; i < 16; ++i) { x[i] = in[threadIdx.x * 32 + i]; y[i] = in[threadIdx.x * 32 + 16 + i]; }
  const v6i r = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(x, y, scale);
#pragma unroll
  for (int i = 0; i < 6; ++i) out[threadIdx.x * 6 + i] = r[i];
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
    for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c[i], 2, 2, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
  }
  unsigned long long t2 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, c[i], 0, 0, 0);
  }
  unsigned long long t3 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += c[i][0] + c[i][3];
  if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = t2 - t1; out[2] = t3 - t2; out[3] = (unsigned long long)s; }
}

// Whole chip: every wave issues the K-loop mix of one K = 128 step of the split GEMM on 8 accumulators (register
// operands with varied bit patterns so that the data path toggles).  MIX 0: f16 only (4 per accumulator), 8: + 2 e4m3,
// 6: + 2 e2m3.
template <int MIX>
__global__ __launch_bounds__(256) void mix_kernel(float* out, int iters, unsigned seed) {
  unsigned r = seed + threadIdx.x * 2654435761u + blockIdx.x * 40503u;
  v8i a, b;
  v8h ha, hb;
  for (int i = 0; i < 8; ++i) {
    r = r * 1664525u + 1013904223u; a[i] = (int)(r & 0x77777777u);   // e4m3 / e2m3 codes without the top exponent bits
    r = r * 1664525u + 1013904223u; b[i] = (int)(r & 0x77777777u);
    ha[i] = (_Float16)((int)(r >> 20 & 255) - 128) * (_Float16)0.01f;
    hb[i] = (_Float16)((int)(r >> 10 & 255) - 128) * (_Float16)0.01f;
  }
  v4f c[8];
  for (int i = 0; i < 8; ++i) c[i] = (v4f){0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 4; ++rep)
#pragma unroll
      for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, c[i], 0, 0, 0);
    if (MIX == 8) {
#pragma unroll
      for (int rep = 0; rep < 2; ++rep)
#pragma unroll
        for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c[i], 0, 0, 0, 0x70707070, 0, 0x70707070);
    } else if (MIX == 6) {
#pragma unroll
      for (int rep = 0; rep < 2; ++rep)
#pragma unroll
        for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c[i], 2, 2, 0, 0x70707070, 0, 0x70707070);
    }
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
  if (s == 1.2345f) out[0] = s;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

static int kmap(int hyp, int g, int j) { return hyp == 1 ? 32 * g + j : (j < 16 ? 16 * g + j : 64 + 16 * g + (j - 16)); }

template <int MIX> static float time_mix(float* dout, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(mix_kernel<MIX>, dim3(256 * 8), dim3(256), 0, 0, dout, iters / 8, 1u);   // warm
  hipEventRecord(e0, 0);
  for (int rep = 0; rep < 8; ++rep) hipLaunchKernelGGL(mix_kernel<MIX>, dim3(256 * 8), dim3(256), 0, 0, dout, iters, 7u + rep);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / 8;
}

int main() {
  // ---- 1, 2: layout and scales
  static int hA[64 * 6], hB[64 * 6];
  static uint8_t codeA[64][32], codeB[64][32];
  srand(11);
  memset(hA, 0, sizeof(hA)); memset(hB, 0, sizeof(hB));
  for (int l = 0; l < 64; ++l)
    for (int j = 0; j < 32; ++j) {
      codeA[l][j] = rand() & 63; codeB[l][j] = rand() & 63;
      const int bit = 6 * j;
      for (int t = 0; t < 6; ++t) {
        if (codeA[l][j] >> t & 1) hA[l * 6 + (bit + t) / 32] |= 1u << ((bit + t) & 31);
        if (codeB[l][j] >> t & 1) hB[l * 6 + (bit + t) / 32] |= 1u << ((bit + t) & 31);
      }
    }
  int *dA, *dB, *dsa, *dsb; float* dC;
  CK(hipMalloc(&dA, sizeof(hA))); CK(hipMalloc(&dB, sizeof(hB))); CK(hipMalloc(&dC, 256 * 4));
  CK(hipMalloc(&dsa, 64 * 4)); CK(hipMalloc(&dsb, 64 * 4));
  CK(hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice)); CK(hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice));
  int fails = 0;
  for (int test = 0; test < 4; ++test) {
    int sa[64], sb[64];
    for (int l = 0; l < 64; ++l) {
      sa[l] = test == 0 ? 127 : 0x11223300 | (120 + (l * 7) % 13);
      sb[l] = test == 0 ? 127 : 0x44556600 | (125 + (l * 5) % 7);
    }
    CK(hipMemcpy(dsa, sa, sizeof(sa), hipMemcpyHostToDevice)); CK(hipMemcpy(dsb, sb, sizeof(sb), hipMemcpyHostToDevice));
    if (test < 2) hipLaunchKernelGGL(mfma6_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dC, dsa, dsb);
    else hipLaunchKernelGGL(mfma6_asm_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dC, dsa, dsb);
    float hC[256];
    CK(hipMemcpy(hC, dC, sizeof(hC), hipMemcpyDeviceToHost));
    for (int hyp = 1; hyp <= 2; ++hyp) {
      double Am[16][128], Bm[16][128];
      for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 32; ++j) {
          const int k = kmap(hyp, l >> 4, j);
          const double sca = ldexp(1.0, (sa[(l & 15) + 16 * (k / 32)] & 255) - 127), scb = ldexp(1.0, (sb[(l & 15) + 16 * (k / 32)] & 255) - 127);
          Am[l & 15][k] = e2m3_decode(codeA[l][j]) * sca;
          Bm[l & 15][k] = e2m3_decode(codeB[l][j]) * scb;
        }
      double maxerr = 0, maxref = 0;
      for (int i = 0; i < 16; ++i)
        for (int n = 0; n < 16; ++n) {
          double ref = 0;
          for (int k = 0; k < 128; ++k) ref += Am[i][k] * Bm[n][k];
          maxref = fmax(maxref, fabs(ref));
          maxerr = fmax(maxerr, fabs(hC[i * 16 + n] - ref));
        }
      const bool ok = maxerr <= 1e-5 * maxref;
      printf("test %d (%s, %s scales) K map H%d: max |C - ref| = %.3e of %.3e -> %s\n", test, test < 2 ? "builtin" : "asm v6",
             (test & 1) ? "per-lane" : (test == 0 ? "unit" : "unit"), hyp, maxerr, maxref, ok ? "MATCH" : "no");
      if (hyp == 1 && !ok) ++fails;
    }
  }
  // ---- 3: conversion
  {
    float hin[64 * 32];
    for (int l = 0; l < 64; ++l)
      for (int j = 0; j < 32; ++j) hin[l * 32 + j] = l == 0 ? (j - 16) * 0.25f : l == 1 ? (j + 1) * 0.0625f * (j & 1 ? -1 : 1) : (float)((rand() % 2001) - 1000) / 130.0f;
    float* din; int* dout;
    CK(hipMalloc(&din, sizeof(hin))); CK(hipMalloc(&dout, 64 * 6 * 4));
    CK(hipMemcpy(din, hin, sizeof(hin), hipMemcpyHostToDevice));
    for (int pass = 0; pass < 2; ++pass) {
      const float scale = pass == 0 ? 1.0f : 4.0f;
      hipLaunchKernelGGL(cvt6_kernel, dim3(1), dim3(64), 0, 0, din, dout, scale);
      int hout[64 * 6];
      CK(hipMemcpy(hout, dout, sizeof(hout), hipMemcpyDeviceToHost));
      double worst = 0; int clamps = 0;
      for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 32; ++j) {
          int code = 0;
          for (int t = 0; t < 6; ++t) code |= ((hout[l * 6 + (6 * j + t) / 32] >> ((6 * j + t) & 31)) & 1) << t;
          const float got = e2m3_decode(code) * scale, want = hin[l * 32 + j];
          if (l < 2 && pass == 0 && j < 12) printf("  cvt(%g) -> code 0x%02x = %g\n", want, code, got);
          if (fabsf(want) > 7.5f * scale) { ++clamps; continue; }
          const double step = fabsf(want) < 2 * scale ? 0.125 * scale : fabsf(want) < 4 * scale ? 0.25 * scale : 0.5 * scale;
          worst = fmax(worst, fabs(got - want) / step);
        }
      printf("cvt_scalef32_2xpk16_fp6_f32 with scale %g: element j in bits [6j,6j+6), value = e2m3 * scale: worst error %.3f steps "
             "(<= 0.5 = round to nearest), %d clamped\n", scale, worst, clamps);
      if (worst > 0.5001) ++fails;
    }
  }
  // ---- 4: rates
  unsigned long long* dt;
  CK(hipMalloc(&dt, 4 * 8));
  hipLaunchKernelGGL(rate_kernel, dim3(1), dim3(64), 0, 0, dt, 2000);
  unsigned long long ht[4];
  CK(hipMemcpy(ht, dt, sizeof(ht), hipMemcpyDeviceToHost));
  printf("one wave, 8 accumulators (s_memtime ticks per MFMA): 16x16x128 e4m3 %.2f, 16x16x128 e2m3 %.2f, 16x16x32 f16 %.2f\n",
         ht[0] / 16000.0, ht[1] / 16000.0, ht[2] / 16000.0);
  float* dout2;
  CK(hipMalloc(&dout2, 64));
  const int iters = 20000;
  for (int round = 0; round < 2; ++round) {
    const float t0 = time_mix<0>(dout2, iters), t8 = time_mix<8>(dout2, iters), t6 = time_mix<6>(dout2, iters);
    printf("whole chip (2048 WG x 4 waves, %d iterations of the K=128 step on 8 accumulators): f16 only %.3f ms, + 2 e4m3 %.3f ms "
           "(x%.2f), + 2 e2m3 %.3f ms (x%.2f); e2m3 mix / e4m3 mix = %.3f\n", iters, t0, t8, t8 / t0, t6, t6 / t0, t6 / t8);
  }
  printf("%s\n", fails ? "PROBE FAILED" : "probe ok");
  return fails ? 1 : 0;
}
