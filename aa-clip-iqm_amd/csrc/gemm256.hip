// 256x256x64 NT GEMM for the big tower products (M = B*1370 rows), gfx950.
//
// Why a second kernel: the 128x128 tile moves 32 KB into LDS per 2.1 MFLOP, which
// caps it near the L2->LDS rate of a CU; a 256x256 tile halves the bytes per flop.
//
// Structure (one workgroup = 8 waves = 2 (M) x 4 (N), 128x64 of C per wave, one
// workgroup per CU, 128 KiB LDS = 2 stages x (A 32 KiB + W 32 KiB)):
//   * every operand tile is cut into two half-operands of 16 KiB (the rows the
//     waves need for output quadrant a-sub / b-sub 0 and 1);
//   * a K tile is consumed in four phases, one 64x32 output quadrant per wave and
//     phase: (A0,B0) (A0,B1) (A1,B1) (A1,B0); each phase reads only the fragments
//     it is missing from LDS and issues the DMA of ONE half-operand of the next K
//     tile, in the order the next tile consumes them;
//   * waits are counted: s_waitcnt vmcnt(4) before the barrier of a phase leaves
//     the two youngest half-operands (4 DMA instructions per wave) in flight, so
//     every half-operand has about three phases of latency cover; never vmcnt(0)
//     in the steady state;
//   * MFMA operand roles are swapped (W rows are the A operand): the accumulator
//     then holds 4 consecutive output columns per register group, so 16-bit
//     results are packed, staged through LDS and written as full 128-byte rows,
//     fp32 results are written as 16-byte vectors.
#include "common.h"
#include "kernels.h"

namespace aaclip {

// GELU for 16-bit outputs: erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7,
// far below the 16-bit output rounding); the fp32 parity path keeps erff.
AACLIP_DEV float gelu_fast(float x) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  float poly = fmaf(1.061405429f, t, -1.453152027f);
  poly = fmaf(poly, t, 1.421413741f);
  poly = fmaf(poly, t, -0.284496736f);
  poly = fmaf(poly, t, 0.254829592f);
  poly *= t;
  const float e = __expf(-z * z);
  const float erf_abs = 1.0f - poly * e;          // erf(|x|/sqrt2)
  const float half_x = 0.5f * x;
  return fmaf(fabsf(half_x), erf_abs, half_x);     // 0.5x(1 + sign(x) erf|.|) = 0.5x + 0.5|x| erf|.|
}

template <typename T, int EPI>
__global__ __launch_bounds__(512, 2) void gemm16_256_kernel(GemmParams p, int PN, int patches_n, int total_patches) {
  typedef typename Elem<T>::vec8 vec8;
  typedef typename Elem<T>::vec4 vec4;
  __shared__ __attribute__((aligned(16))) char smem[131072];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int wr = wave >> 2, wc = wave & 3;

  // ---- XCD-aware tile order: each XCD walks 8 x PN patches of tiles ----
  const int tiles_m = (p.M + 255) >> 8;
  int tm, tn;
  {
    const int P = 8 * PN;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int gp = (j / P) * 8 + xcd, local = j % P;
    if (gp >= total_patches) return;
    const int pm = gp / patches_n, pn = gp - pm * patches_n;
    tm = pm * 8 + local / PN;
    tn = pn * PN + local % PN;
    if (tm >= tiles_m) return;
  }

  // ---- DMA sources: per half-operand, two 1-KiB wave-instructions per wave ----
  // LDS row group g (8 rows) of a 256-row tile; half-operand `sub` of the M side
  // holds groups {0..7,16..23}+8*sub, of the N side groups {0..3,8..11,16..19,24..27}+4*sub.
  const T* srcA[2][2];
  const T* srcW[2][2];
  int dstA[2][2], dstW[2][2];
#pragma unroll
  for (int sub = 0; sub < 2; ++sub)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int idx = wave * 2 + j;
      const int ga = (idx & 7) + (idx >> 3) * 16 + sub * 8;
      const int gw = (idx & 3) + (idx >> 2) * 8 + sub * 4;
      int row, chunk;
      tile_src(ga * 64 + lane, row, chunk);
      int ar = tm * 256 + row;
      ar = ar < p.M ? ar : p.M - 1;
      srcA[sub][j] = (const T*)p.A + (long)ar * p.lda + chunk * 8;
      dstA[sub][j] = ga * 1024;
      tile_src(gw * 64 + lane, row, chunk);
      srcW[sub][j] = (const T*)p.W + (long)(tn * 256 + row) * p.K + chunk * 8;
      dstW[sub][j] = 32768 + gw * 1024;
    }
  // ---- fragment read offsets ----
  int offM[4][4], offN[2][4];   // [tile][ks]
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) offM[mi][ks] = tile_off(wr * 128 + mi * 32 + r, 2 * ks + h);
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) offN[ni][ks] = 32768 + tile_off(wc * 64 + ni * 32 + r, 2 * ks + h);
  }

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int nk = p.K >> 6;
#define ISSUE_A(sub, st, kt)                                              \
  {                                                                       \
    glds16(srcA[sub][0] + (kt) * 64, smem + (st) * 65536 + dstA[sub][0]); \
    glds16(srcA[sub][1] + (kt) * 64, smem + (st) * 65536 + dstA[sub][1]); \
  }
#define ISSUE_W(sub, st, kt)                                              \
  {                                                                       \
    glds16(srcW[sub][0] + (kt) * 64, smem + (st) * 65536 + dstW[sub][0]); \
    glds16(srcW[sub][1] + (kt) * 64, smem + (st) * 65536 + dstW[sub][1]); \
  }
#define WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define QUADRANT(a, b)                                                       \
  {                                                                          \
    __builtin_amdgcn_s_setprio(1);                                           \
    _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                       \
      acc[2 * a][b] = Elem<T>::mma32(fn[ks], fm[0][ks], acc[2 * a][b]);         \
      acc[2 * a + 1][b] = Elem<T>::mma32(fn[ks], fm[1][ks], acc[2 * a + 1][b]); \
    }                                                                        \
    __builtin_amdgcn_s_setprio(0);                                           \
  }

  // prologue: the four half-operands of tile 0, in consumption order
  ISSUE_A(0, 0, 0);
  ISSUE_W(0, 0, 0);
  ISSUE_W(1, 0, 0);
  ISSUE_A(1, 0, 0);

  vec8 fm[2][4];  // M-side fragments of the current a-sub: [mi within sub][ks]
  vec8 fn[4];     // N-side fragments of the current b-sub: [ks]
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1, nxt = cur ^ 1;
    const bool more = kt + 1 < nk;
    const char* sb = smem + cur * 65536;
    // ---- phase 0: quadrant (A0, B0); needs A0(kt), B0(kt); younger in flight: B1, A1
    WAIT_VM(4);
    __builtin_amdgcn_s_barrier();
    if (more) ISSUE_A(0, nxt, kt + 1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      fn[ks] = *(const vec8*)(sb + offN[0][ks]);
      fm[0][ks] = *(const vec8*)(sb + offM[0][ks]);
      fm[1][ks] = *(const vec8*)(sb + offM[1][ks]);
    }
    QUADRANT(0, 0);
    // ---- phase 1: quadrant (A0, B1); needs B1(kt); younger: A1(kt) [+ A0(kt+1)]
    if (more) WAIT_VM(4); else WAIT_VM(2);
    __builtin_amdgcn_s_barrier();
    if (more) ISSUE_W(0, nxt, kt + 1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) fn[ks] = *(const vec8*)(sb + offN[1][ks]);
    QUADRANT(0, 1);
    // ---- phase 2: quadrant (A1, B1); needs A1(kt); younger: [A0(kt+1), B0(kt+1)]
    if (more) WAIT_VM(4); else WAIT_VM(0);
    __builtin_amdgcn_s_barrier();
    if (more) ISSUE_W(1, nxt, kt + 1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      fm[0][ks] = *(const vec8*)(sb + offM[2][ks]);
      fm[1][ks] = *(const vec8*)(sb + offM[3][ks]);
    }
    QUADRANT(1, 1);
    // ---- phase 3: quadrant (A1, B0); B0(kt) landed before phase 0
    if (more) ISSUE_A(1, nxt, kt + 1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) fn[ks] = *(const vec8*)(sb + offN[0][ks]);
    QUADRANT(1, 0);
  }
#undef ISSUE_A
#undef ISSUE_W
#undef QUADRANT

  // ---- epilogue.  acc[mi][ni][e]: column m = mi*32 + r (lane), row n = ni*32 + (e&3) + 8(e>>2) + 4h
  const int m_base = tm * 256 + wr * 128, n_base = tn * 256 + wc * 64;
  if (EPI == EPI_BIAS || EPI == EPI_BIAS_GELU) {
    __syncthreads();  // every wave is done reading the operand tiles
    char* st = smem + wave * 16384;  // this wave's 128 x 64 tile of T, rows of 128 B, chunk ^= (m & 7)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n0 = n_base + ni * 32 + 8 * g + 4 * h;
        const f32x4 bv = *(const f32x4*)(p.bias + n0);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
          vec4 o;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float v = acc[mi][ni][4 * g + j] + bv[j];
            if (EPI == EPI_BIAS_GELU) v = gelu_fast(v);
            else if (n0 + j < p.scale_cols) v *= p.scale;
            o[j] = from_float<T>(v);
          }
          const int m = mi * 32 + r;
          *(vec4*)(st + m * 128 + (((ni * 4 + g) ^ (m & 7)) << 4) + 8 * h) = o;
        }
      }
    // read back 16 B per lane: 8 lanes cover one 128-byte row
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      const int m = it * 8 + (lane >> 3), c = lane & 7;
      const u32x4 v = *(const u32x4*)(st + m * 128 + ((c ^ (m & 7)) << 4));
      const int row = m_base + m;
      if (row < p.M) *(u32x4*)((T*)p.out + (long)row * p.ldc + n_base + c * 8) = v;
    }
  } else {
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      const int row = m_base + mi * 32 + r;
      if (row < p.M) {
        long orow = row;
        const float* posr = nullptr;
        if (EPI == EPI_PATCH) {
          const int b = row / p.P, pi = row - b * p.P;
          orow = (long)b * p.L + 1 + pi;
          posr = p.pos + (long)(1 + pi) * p.N;
        }
        float* op = (float*)p.out + orow * p.ldc;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int n0 = n_base + ni * 32 + 8 * g + 4 * h;
            f32x4 v;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = acc[mi][ni][4 * g + j];
            if (EPI == EPI_BIAS_RESID) {
              const f32x4 bv = *(const f32x4*)(p.bias + n0);
              const f32x4 x = *(const f32x4*)(op + n0);
              v = x + (v + bv);
            } else if (EPI == EPI_ACT_F32) {
              if (p.bias) v = v + *(const f32x4*)(p.bias + n0);
              if (p.act == 1) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = leaky(v[j]);
              }
            } else if (EPI == EPI_PATCH) {
              v = v + *(const f32x4*)(posr + n0);
            }
            *(f32x4*)(op + n0) = v;
          }
      }
    }
  }
}

template <typename T>
static void launch256_t(int epi, const GemmParams& p, hipStream_t s) {
  const int tiles_n = p.N / 256, tiles_m = (p.M + 255) / 256;
  const int PN = (tiles_n % 4 == 0) ? 4 : (tiles_n % 3 == 0) ? 3 : (tiles_n % 2 == 0) ? 2 : 1;
  const int patches_n = tiles_n / PN, patches_m = (tiles_m + 7) / 8;
  const int total = patches_n * patches_m;
  const int grid = ((total + 7) / 8) * 8 * 8 * PN;
  dim3 g(grid), b(512);
  switch (epi) {
    case EPI_BIAS: hipLaunchKernelGGL((gemm16_256_kernel<T, EPI_BIAS>), g, b, 0, s, p, PN, patches_n, total); break;
    case EPI_BIAS_GELU: hipLaunchKernelGGL((gemm16_256_kernel<T, EPI_BIAS_GELU>), g, b, 0, s, p, PN, patches_n, total); break;
    case EPI_BIAS_RESID: hipLaunchKernelGGL((gemm16_256_kernel<T, EPI_BIAS_RESID>), g, b, 0, s, p, PN, patches_n, total); break;
    case EPI_ACT_F32: hipLaunchKernelGGL((gemm16_256_kernel<T, EPI_ACT_F32>), g, b, 0, s, p, PN, patches_n, total); break;
    case EPI_PATCH: hipLaunchKernelGGL((gemm16_256_kernel<T, EPI_PATCH>), g, b, 0, s, p, PN, patches_n, total); break;
  }
}

bool gemm256_applicable(int dtype, const GemmParams& p) {
  return dtype != AACLIP_F32 && p.N % 256 == 0 && p.K % 64 == 0 && p.ldc % 8 == 0 && (p.scale_cols % 4) == 0;
}

void launch_gemm256(int dtype, int epi, const GemmParams& p, hipStream_t s) {
  if (dtype == AACLIP_F16) launch256_t<f16>(epi, p, s);
  else launch256_t<bf16>(epi, p, s);
}

}  // namespace aaclip
