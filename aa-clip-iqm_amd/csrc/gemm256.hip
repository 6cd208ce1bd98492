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

#ifndef AACLIP_MEASURE
#error "gemm256.hip holds A/B variants and timing ablations: it is part of the measurement library only (make measure)"
#endif

namespace aaclip {

template <typename T, int EPI>
AACLIP_DEV void epilogue256(const GemmParams& p, f32x16 (&acc)[4][2], char* smem, int tm, int tn, int wave, int lane) {
  typedef typename Elem<T>::vec4 vec4;
  const int r = lane & 31, h = lane >> 5;
  const int wr = wave >> 2, wc = wave & 3;
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
              const f32x4 x = *(const f32x4*)((p.resid ? p.resid + orow * p.ldc : (const float*)op) + n0);
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

template <typename T, int EPI>
__global__ __launch_bounds__(512, 2) void gemm16_256_kernel(GemmParams p, int PN, int patches_n, int total_patches) {
  typedef typename Elem<T>::vec8 vec8;
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

  epilogue256<T, EPI>(p, acc, smem, tm, tn, wave, lane);
}

// ---------------------------------------------------------------------------
// Variant with the DMA instructions of a phase issued BETWEEN its MFMAs (an LDS-DMA
// issue costs the wave ~60 cycles among MFMAs against 100-185 in a load burst,
// MI355X_MICROARCH.md cycle constants) instead of in front of them.
// ABL (timing-only ablations, wrong results): 1 = no DMA waits, 2 = no DMA issue in the K loop.
template <typename T, int EPI, int ABL>
__global__ __launch_bounds__(512, 2) void gemm16_256s_kernel(GemmParams p, int PN, int patches_n, int total_patches) {
  typedef typename Elem<T>::vec8 vec8;
  __shared__ __attribute__((aligned(16))) char smem[131072];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int wr = wave >> 2, wc = wave & 3;
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
  int srcA[2][2], srcW[2][2], dstA[2][2], dstW[2][2];
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
      srcA[sub][j] = (ar - tm * 256) * (int)p.lda + chunk * 8;
      dstA[sub][j] = ga * 1024;
      tile_src(gw * 64 + lane, row, chunk);
      srcW[sub][j] = row * p.K + chunk * 8;
      dstW[sub][j] = 32768 + gw * 1024;
    }
  const T* baseA = (const T*)p.A + (long)tm * 256 * p.lda;
  const T* baseW = (const T*)p.W + (long)tn * 256 * p.K;
  int offM[4], offN[4];   // [ks]; row tiles differ by the constant 4096 bytes (32 rows)
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    offM[ks] = tile_off(wr * 128 + r, 2 * ks + h);
    offN[ks] = 32768 + tile_off(wc * 64 + r, 2 * ks + h);
  }

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int nk = p.K >> 6;
#define SISSUE_A(sub, st, kt)                                                       \
  {                                                                                 \
    glds16(baseA + srcA[sub][0] + (kt) * 64, smem + (st) * 65536 + dstA[sub][0]);   \
    glds16(baseA + srcA[sub][1] + (kt) * 64, smem + (st) * 65536 + dstA[sub][1]);   \
  }
#define SISSUE_W(sub, st, kt)                                                       \
  {                                                                                 \
    glds16(baseW + srcW[sub][0] + (kt) * 64, smem + (st) * 65536 + dstW[sub][0]);   \
    glds16(baseW + srcW[sub][1] + (kt) * 64, smem + (st) * 65536 + dstW[sub][1]);   \
  }
#define SWAIT(n) { if (ABL != 1) WAIT_VM(n); }
#define BAR __builtin_amdgcn_s_barrier();
#define LGKM0 asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#define LD_N(sb, bsub) \
  _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) fn[ks] = *(const vec8*)((sb) + offN[ks] + (bsub) * 4096);
#define LD_M(sb, asub)                                                         \
  _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                           \
    fm[0][ks] = *(const vec8*)((sb) + offM[ks] + (2 * (asub)) * 4096);         \
    fm[1][ks] = *(const vec8*)((sb) + offM[ks] + (2 * (asub) + 1) * 4096);     \
  }
#define SQUAD(a, b)                                                           \
  {                                                                           \
    __builtin_amdgcn_s_setprio(1);                                            \
    _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                        \
      acc[2 * a][b] = Elem<T>::mma32(fn[ks], fm[0][ks], acc[2 * a][b]);         \
      acc[2 * a + 1][b] = Elem<T>::mma32(fn[ks], fm[1][ks], acc[2 * a + 1][b]); \
    }                                                                         \
    __builtin_amdgcn_s_setprio(0);                                            \
  }

#define G1(base, src, dst, st, kt) glds16(base + src + (kt) * 64, smem + (st) * 65536 + dst);
#define PINB __builtin_amdgcn_sched_barrier(0);
// quadrant with the two DMA instructions of this phase issued between the MFMAs
#define SQUADI(a, b, I0, I1)                                                  \
  {                                                                           \
    __builtin_amdgcn_s_setprio(1);                                            \
    acc[2 * a][b] = Elem<T>::mma32(fn[0], fm[0][0], acc[2 * a][b]);             \
    acc[2 * a + 1][b] = Elem<T>::mma32(fn[0], fm[1][0], acc[2 * a + 1][b]);     \
    PINB I0 PINB                                                              \
    acc[2 * a][b] = Elem<T>::mma32(fn[1], fm[0][1], acc[2 * a][b]);             \
    acc[2 * a + 1][b] = Elem<T>::mma32(fn[1], fm[1][1], acc[2 * a + 1][b]);     \
    acc[2 * a][b] = Elem<T>::mma32(fn[2], fm[0][2], acc[2 * a][b]);             \
    acc[2 * a + 1][b] = Elem<T>::mma32(fn[2], fm[1][2], acc[2 * a + 1][b]);     \
    PINB I1 PINB                                                              \
    acc[2 * a][b] = Elem<T>::mma32(fn[3], fm[0][3], acc[2 * a][b]);             \
    acc[2 * a + 1][b] = Elem<T>::mma32(fn[3], fm[1][3], acc[2 * a + 1][b]);     \
    __builtin_amdgcn_s_setprio(0);                                            \
  }

  // prologue: the four half-operands of tile 0 in consumption order
  SISSUE_A(0, 0, 0);
  SISSUE_W(0, 0, 0);
  SISSUE_W(1, 0, 0);
  SISSUE_A(1, 0, 0);

  vec8 fm[2][4], fn[4];
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1, nxt = cur ^ 1;
    const bool more = kt + 1 < nk;
    const bool dma = more && ABL != 2;
    const char* sb = smem + cur * 65536;
    // ---- P0: needs A0,B0(kt); younger in flight: B1, A1
    SWAIT(4)
    BAR
    LD_N(sb, 0)
    LD_M(sb, 0)
    SQUADI(0, 0, if (dma) G1(baseA, srcA[0][0], dstA[0][0], nxt, kt + 1), if (dma) G1(baseA, srcA[0][1], dstA[0][1], nxt, kt + 1))
    // ---- P1: needs B1(kt); younger: A1(kt) [+ A0(kt+1)]
    if (more) SWAIT(4) else SWAIT(2)
    BAR
    LD_N(sb, 1)
    SQUADI(0, 1, if (dma) G1(baseW, srcW[0][0], dstW[0][0], nxt, kt + 1), if (dma) G1(baseW, srcW[0][1], dstW[0][1], nxt, kt + 1))
    // ---- P2: needs A1(kt); younger: [A0(kt+1), B0(kt+1)]
    if (more) SWAIT(4) else SWAIT(0)
    BAR
    LD_M(sb, 1)
    SQUADI(1, 1, if (dma) G1(baseW, srcW[1][0], dstW[1][0], nxt, kt + 1), if (dma) G1(baseW, srcW[1][1], dstW[1][1], nxt, kt + 1))
    // ---- P3: B0(kt) again (landed before P0)
    LD_N(sb, 0)
    SQUADI(1, 0, if (dma) G1(baseA, srcA[1][0], dstA[1][0], nxt, kt + 1), if (dma) G1(baseA, srcA[1][1], dstA[1][1], nxt, kt + 1))
  }
#undef G1
#undef PINB
#undef SQUADI
#undef SISSUE_A
#undef SISSUE_W
#undef SWAIT
#undef BAR
#undef LGKM0
#undef LD_N
#undef LD_M
#undef SQUAD
  epilogue256<T, EPI>(p, acc, smem, tm, tn, wave, lane);
}

template <typename T>
static void launch256_t(int epi, const GemmParams& p, hipStream_t s, int pipelined) {
  const int tiles_n = p.N / 256, tiles_m = (p.M + 255) / 256;
  const int PN = (tiles_n % 4 == 0) ? 4 : (tiles_n % 3 == 0) ? 3 : (tiles_n % 2 == 0) ? 2 : 1;
  const int patches_n = tiles_n / PN, patches_m = (tiles_m + 7) / 8;
  const int total = patches_n * patches_m;
  const int grid = ((total + 7) / 8) * 8 * 8 * PN;
  dim3 g(grid), b(512);
  if (pipelined == 2 || pipelined == 3) {
    // timing ablations (wrong results) exist for the fp32-output epilogue only: no silent substitute for the others,
    // and never an fp32-storing kernel on a 16-bit output buffer
    if (epi != EPI_ACT_F32) { set_launch_error("gemm: ablation kernels exist for the fp32-output epilogue only"); return; }
    if (pipelined == 2) hipLaunchKernelGGL((gemm16_256s_kernel<T, EPI_ACT_F32, 1>), g, b, 0, s, p, PN, patches_n, total);
    else hipLaunchKernelGGL((gemm16_256s_kernel<T, EPI_ACT_F32, 2>), g, b, 0, s, p, PN, patches_n, total);
    return;
  }
  if (pipelined) {
    switch (epi) {
      case EPI_BIAS: hipLaunchKernelGGL((gemm16_256s_kernel<T, EPI_BIAS, 0>), g, b, 0, s, p, PN, patches_n, total); break;
      case EPI_BIAS_GELU: hipLaunchKernelGGL((gemm16_256s_kernel<T, EPI_BIAS_GELU, 0>), g, b, 0, s, p, PN, patches_n, total); break;
      case EPI_BIAS_RESID: hipLaunchKernelGGL((gemm16_256s_kernel<T, EPI_BIAS_RESID, 0>), g, b, 0, s, p, PN, patches_n, total); break;
      case EPI_ACT_F32: hipLaunchKernelGGL((gemm16_256s_kernel<T, EPI_ACT_F32, 0>), g, b, 0, s, p, PN, patches_n, total); break;
      case EPI_PATCH: hipLaunchKernelGGL((gemm16_256s_kernel<T, EPI_PATCH, 0>), g, b, 0, s, p, PN, patches_n, total); break;
    }
    return;
  }
  switch (epi) {
    case EPI_BIAS: hipLaunchKernelGGL((gemm16_256_kernel<T, EPI_BIAS>), g, b, 0, s, p, PN, patches_n, total); break;
    case EPI_BIAS_GELU: hipLaunchKernelGGL((gemm16_256_kernel<T, EPI_BIAS_GELU>), g, b, 0, s, p, PN, patches_n, total); break;
    case EPI_BIAS_RESID: hipLaunchKernelGGL((gemm16_256_kernel<T, EPI_BIAS_RESID>), g, b, 0, s, p, PN, patches_n, total); break;
    case EPI_ACT_F32: hipLaunchKernelGGL((gemm16_256_kernel<T, EPI_ACT_F32>), g, b, 0, s, p, PN, patches_n, total); break;
    case EPI_PATCH: hipLaunchKernelGGL((gemm16_256_kernel<T, EPI_PATCH>), g, b, 0, s, p, PN, patches_n, total); break;
  }
}

void launch_gemm256(int dtype, int epi, const GemmParams& p, hipStream_t s, int pipelined) {
  if (dtype == AACLIP_F16) launch256_t<f16>(epi, p, s, pipelined);
  else launch256_t<bf16>(epi, p, s, pipelined);
}

}  // namespace aaclip
