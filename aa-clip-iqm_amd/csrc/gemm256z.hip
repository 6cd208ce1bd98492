// Persistent 256x256x64 NT GEMM on v_mfma_f32_16x16x32 (gfx950): gemm16_256x_kernel's K pipeline
// (gemm256t.hip) run CONTINUOUSLY over the tiles of one workgroup.
//
// Why: with one workgroup per tile, every tile pays the launch of a workgroup, ~4 us of cold
// prologue (first operand tiles from HBM), and the drain of its epilogue stores before the CU can
// take the next tile; for the K = 1024 GEMMs of the tower that is a third of a tile's life
// ("per-tile fixed cost").  Here one workgroup per CU walks its tiles
// (virtual block ids blockIdx.x, +gridDim.x, ... in the same XCD-aware order), and
//   * the DMA pipeline never drains: in the last two K tiles of a tile the "next K tile" slots of
//     the schedule fetch K tile 0 (and the B0 half of K tile 1) of the NEXT output tile;
//   * the epilogue stages through a wave-private 4 KiB of LDS (the 32 KiB the two operand stages
//     leave free), so it needs no barrier and does not touch the stages being filled;
//   * epilogue stores are younger than the prefetched tiles in the in-order vmcnt queue, so the
//     first K tile after an epilogue waits with vmcnt(4 + STORES): the stores drain behind three
//     phases of MFMA work instead of in front of a new workgroup.
// The two wave groups leave their one-segment stagger for the epilogue (both run it at once) and
// re-enter it afterwards; barrier counts per tile are equal for all waves.
#include "common.h"
#include "kernels.h"
#include "mma16.h"

#ifndef AACLIP_MEASURE
#error "gemm256z.hip (persistent-tile experiment, stamp builds) is part of the measurement library only (make measure)"
#endif

namespace aaclip {

// vm operations (global stores) the epilogue issues per wave for a full tile
#ifdef Z_NOSTORE
template <int EPI> struct EpiStores { static constexpr int n = 0; };
#define Z_ST(cond) ((cond) && p.act == 77)
#else
template <int EPI> struct EpiStores { static constexpr int n = (EPI == EPI_BIAS || EPI == EPI_BIAS_GELU) ? 16 : 32; };
#define Z_ST(cond) (cond)
#endif

#ifdef Z_STAMP
// diagnosis build (-DZ_STAMP): per-workgroup cycle sums of the tile segments, wave 0 only
__device__ unsigned long long g_zstamp[8 * 512];
AACLIP_DEV unsigned long long zclock() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
#define ZT(k) { const unsigned long long tnow = zclock(); zsum[k] += tnow - zlast; zlast = tnow; }
#else
#define ZT(k)
#endif

template <int N> AACLIP_DEV void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// acc[mi][ni][j]: m = mi*16 + (lane&15), n = ni*16 + 4*(lane>>4) + j   (mi 0..7, ni 0..3)
// stg: this wave's private 4 KiB.  No barrier inside.
template <typename T, int EPI>
AACLIP_DEV void epilogue256z(const GemmParams& p, f32x4 (&acc)[8][4], char* stg, int tm, int tn, int wave, int lane) {
  typedef typename Elem<T>::vec4 vec4;
  const int c16 = lane & 15, q4 = lane >> 4;
  const int wr = wave >> 2, wc = wave & 3;
  const int m_base = tm * 256 + wr * 128, n_base = tn * 256 + wc * 64;
  if (EPI == EPI_BIAS || EPI == EPI_BIAS_GELU) {
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {   // 32 rows x 128 B per pass; 16-B chunk ^= (row & 7)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        const int nl = ni * 16 + 4 * q4;
        const f32x4 bvn = *(const f32x4*)(p.bias + n_base + nl);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int mi = 2 * pass + h;
          vec4 o;
          if (EPI == EPI_BIAS_GELU) {
            f32x2 g0 = {acc[mi][ni][0] + bvn[0], acc[mi][ni][1] + bvn[1]};
            f32x2 g1 = {acc[mi][ni][2] + bvn[2], acc[mi][ni][3] + bvn[3]};
            g0 = gelu_fast2(g0);
            g1 = gelu_fast2(g1);
            o[0] = from_float<T>(g0[0]); o[1] = from_float<T>(g0[1]);
            o[2] = from_float<T>(g1[0]); o[3] = from_float<T>(g1[1]);
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              float v = acc[mi][ni][j] + bvn[j];
              if (n_base + nl + j < p.scale_cols) v *= p.scale;
              o[j] = from_float<T>(v);
            }
          }
          const int m = h * 16 + c16;
          *(vec4*)(stg + m * 128 + ((((nl >> 3)) ^ (m & 7)) << 4) + (nl & 4) * 2) = o;
        }
      }
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int m = it * 8 + (lane >> 3), c = lane & 7;
        const u32x4 v = *(const u32x4*)(stg + m * 128 + ((c ^ (m & 7)) << 4));
        const int row = m_base + pass * 32 + m;
        if (Z_ST(row < p.M)) *(u32x4*)((T*)p.out + (long)row * p.ldc + n_base + c * 8) = v;
      }
    }
  } else {
    // fp32 outputs: 16 rows x 256 B per pass, 16-B chunk ^= row; read back as 4 rows x 256 B per instruction
    const int cc = lane & 15, rr = lane >> 4;
    const int n0 = n_base + cc * 4;
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (EPI == EPI_BIAS_RESID || (EPI == EPI_ACT_F32 && p.bias)) bv = *(const f32x4*)(p.bias + n0);
#pragma unroll
    for (int pass = 0; pass < 8; ++pass) {
      f32x4 extra[4];
      long orow[4];
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int row = m_base + pass * 16 + it * 4 + rr;
        orow[it] = row;
        if (EPI == EPI_PATCH) {
          const int rc = row < p.M ? row : p.M - 1;
          const int b = rc / p.P, pi = rc - b * p.P;
          orow[it] = (long)b * p.L + 1 + pi;
          extra[it] = *(const f32x4*)(p.pos + (long)(1 + pi) * p.N + n0);
        } else if (EPI == EPI_BIAS_RESID) {
          const long rc = row < p.M ? row : p.M - 1;
          extra[it] = *(const f32x4*)((p.resid ? p.resid : (const float*)p.out) + rc * p.ldc + n0);
        }
      }
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) *(f32x4*)(stg + c16 * 256 + (((ni * 4 + q4) ^ c16) << 4)) = acc[pass][ni];
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int m = it * 4 + rr;
        f32x4 v = *(const f32x4*)(stg + m * 256 + ((cc ^ m) << 4));
        if (EPI == EPI_BIAS_RESID) {
          v = extra[it] + (v + bv);
        } else if (EPI == EPI_ACT_F32) {
          v = v + bv;
          if (p.act == 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = leaky(v[j]);
          }
        } else if (EPI == EPI_PATCH) {
          v = v + extra[it];
        }
        if (Z_ST(m_base + pass * 16 + m < p.M)) *(f32x4*)((float*)p.out + orow[it] * p.ldc + n0) = v;
      }
    }
  }
}

// virtual block id -> tile, the mapping of gemm16_256x_kernel (XCD = id & 7 gets whole patches)
struct TileMap {
  int PM, PN, patches_n, total_patches, tiles_m;
  AACLIP_DEV bool decode(int vb, int& tm, int& tn) const {
    const int P = PM * PN;
    const int xcd = vb & 7, j = vb >> 3;
    const int gp = (j / P) * 8 + xcd, local = j % P;
    if (gp >= total_patches) return false;
    const int pm = gp / patches_n, pn = gp - pm * patches_n;
    tm = pm * PM + local / PN;
    tn = pn * PN + local % PN;
    return tm < tiles_m;
  }
};

template <typename T, int EPI>
__global__ __launch_bounds__(512, 2) void gemm16_256z_kernel(GemmParams p, int PN, int patches_n, int total_patches,
                                                             int PM, int nvirtual) {
  typedef typename Elem<T>::vec8 vec8;
  constexpr int S = EpiStores<EPI>::n;
  __shared__ __attribute__((aligned(16))) char smem[131072 + 32768];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c16 = lane & 15, q4 = lane >> 4;
  const int wr = wave >> 2, wc = wave & 3;
  TileMap map{PM, PN, patches_n, total_patches, (p.M + 255) >> 8};
  const int G = gridDim.x;

  int vb = blockIdx.x, tm = 0, tn = 0;
  while (vb < nvirtual && !map.decode(vb, tm, tn)) vb += G;
  if (vb >= nvirtual) return;   // uniform for the workgroup: no barrier has been executed yet

  // lane-dependent parts of the DMA source offsets (bytes); the A side depends on the tile row
  // only through the clamp of rows past M
  int srcW[2][2], dstA[2][2], dstW[2][2];
#pragma unroll
  for (int sub = 0; sub < 2; ++sub)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int idx = wave * 2 + j;
      const int ga = (idx & 7) + (idx >> 3) * 16 + sub * 8;
      const int gw = (idx & 3) + (idx >> 2) * 8 + sub * 4;
      int row, chunk;
      dstA[sub][j] = ga * 1024;
      tile_src_id(gw * 64 + lane, row, chunk);
      srcW[sub][j] = (row * p.K + chunk * 8) * 2;
      dstW[sub][j] = 32768 + gw * 1024;
    }
  int offM[2][2], offN[2][2];   // [ks][tile parity]
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int par = 0; par < 2; ++par) {
      offM[ks][par] = tile_off_id(wr * 128 + par * 16 + c16, 4 * ks + q4);
      offN[ks][par] = 32768 + tile_off_id(wc * 64 + par * 16 + c16, 4 * ks + q4);
    }
  char* stg = smem + 131072 + wave * 4096;
  const int nk = p.K >> 6;   // even and >= 4 (checked by the launcher)
  const int ldaB = (int)p.lda * 2;

#define ZSRCA(dst, tmv)                                                          \
  _Pragma("unroll") for (int sub = 0; sub < 2; ++sub) _Pragma("unroll") for (int j = 0; j < 2; ++j) { \
    const int idx = wave * 2 + j;                                                \
    const int ga = (idx & 7) + (idx >> 3) * 16 + sub * 8;                        \
    int row, chunk;                                                              \
    tile_src_id(ga * 64 + lane, row, chunk);                                     \
    int ar = (tmv) * 256 + row;                                                  \
    ar = ar < p.M ? ar : p.M - 1;                                                \
    dst[sub][j] = (ar - (tmv) * 256) * ldaB + chunk * 16;                        \
  }
#define ZRSRC(ptr) __builtin_amdgcn_make_buffer_rsrc((void*)(ptr), 0, 0x7FFFFFF0, 0x00020000)
#define DMA(rs, src, dst, st, kt) \
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(smem + (st) * 65536 + dst), 16, src, (kt) * 128, 0, 0);
#define GA(rs, sa, sub, st, kt) { DMA(rs, sa[sub][0], dstA[sub][0], st, kt) DMA(rs, sa[sub][1], dstA[sub][1], st, kt) }
#define GW(rs, sub, st, kt) { DMA(rs, srcW[sub][0], dstW[sub][0], st, kt) DMA(rs, srcW[sub][1], dstW[sub][1], st, kt) }
#define LGKM0 asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#define BAR __builtin_amdgcn_s_barrier();
#define PINB __builtin_amdgcn_sched_barrier(0);
#define LD_M(sb, a)                                                                         \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int t = 0; t < 4; ++t) \
      fm[t][ks] = *(const vec8*)((sb) + offM[ks][t & 1] + ((a) * 2 + (t >> 1)) * 4096);
#define LD_N(FN, sb, b)                                                                     \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int t = 0; t < 2; ++t) \
      FN[t][ks] = *(const vec8*)((sb) + offN[ks][t] + (b) * 4096);
#define MM(FN, a, b, t, u, ks) acc[4 * (a) + (t)][2 * (b) + (u)] = Mma16<T>::mma(FN[u][ks], fm[t][ks], acc[4 * (a) + (t)][2 * (b) + (u)]);
#define QUADX(FN, a, b)                                                                      \
  {                                                                                          \
    LGKM0                                                                                    \
    PINB                                                                                     \
    __builtin_amdgcn_s_setprio(1);                                                           \
    MM(FN, a, b, 0, 0, 0) MM(FN, a, b, 0, 1, 0) MM(FN, a, b, 1, 0, 0) MM(FN, a, b, 1, 1, 0)  \
    MM(FN, a, b, 2, 0, 0) MM(FN, a, b, 2, 1, 0) MM(FN, a, b, 3, 0, 0) MM(FN, a, b, 3, 1, 0)  \
    MM(FN, a, b, 0, 0, 1) MM(FN, a, b, 0, 1, 1) MM(FN, a, b, 1, 0, 1) MM(FN, a, b, 1, 1, 1)  \
    MM(FN, a, b, 2, 0, 1) MM(FN, a, b, 2, 1, 1) MM(FN, a, b, 3, 0, 1) MM(FN, a, b, 3, 1, 1)  \
    __builtin_amdgcn_s_setprio(0);                                                           \
    PINB                                                                                     \
  }
// One K tile living in stage CUR.  FB0 holds its B0 half on entry; FB1 receives its B1 half and then the
// B0 half of the following K tile.  m1: a following K tile exists (RS*1 / SA1 / K1 = its buffers and K
// offset); m2: one after that exists (RSW2 / K2).  XS = vm operations younger than this tile's DMAs that
// may stay outstanding during phases 0-2 (the previous tile's epilogue stores).
#define KTILE(CUR, FB0, FB1, XS, m1, RSA1, SA1, RSW1, K1, m2, RSW2, K2, RDN)                       \
  {                                                                                           \
    const char* sb = smem + (CUR) * 65536;                                                    \
    /* P0: confirm B1(kt); read A0(kt); issue A0(kt+1) */                                     \
    if (m1) wait_vm<4 + XS>(); else wait_vm<2 + XS>();                                        \
    LD_M(sb, 0)                                                                               \
    if (m1) GA(RSA1, SA1, 0, (CUR) ^ 1, K1)                                                   \
    BAR                                                                                       \
    QUADX(FB0, 0, 0)                                                                          \
    BAR                                                                                       \
    /* P1: confirm A1(kt); read B1(kt); issue B1(kt+1) */                                     \
    if (m1) wait_vm<4 + XS>(); else wait_vm<0>();                                             \
    LD_N(FB1, sb, 1)                                                                          \
    if (m1) GW(RSW1, 1, (CUR) ^ 1, K1)                                                        \
    BAR                                                                                       \
    QUADX(FB1, 0, 1)                                                                          \
    BAR                                                                                       \
    /* P2: confirm B0(kt+1); read A1(kt); issue A1(kt+1) */                                   \
    if (m1) wait_vm<4 + XS>();                                                                \
    LD_M(sb, 1)                                                                               \
    if (m1) GA(RSA1, SA1, 1, (CUR) ^ 1, K1)                                                   \
    BAR                                                                                       \
    QUADX(FB1, 1, 1)                                                                          \
    BAR                                                                                       \
    /* P3: confirm A0(kt+1); read B0(kt+1) into the set B1 vacated; issue B0(kt+2) */         \
    if (m1) wait_vm<4>();                                                                     \
    if ((m1) && (RDN)) LD_N(FB1, smem + ((CUR) ^ 1) * 65536, 0)                                 \
    if (m2) GW(RSW2, 0, CUR, K2)                                                              \
    BAR                                                                                       \
    QUADX(FB0, 1, 0)                                                                          \
    BAR                                                                                       \
  }

  int srcA[2][2], srcAn[2][2];
  ZSRCA(srcA, tm)
  __amdgpu_buffer_rsrc_t rsA = ZRSRC((const T*)p.A + (long)tm * 256 * p.lda);
  __amdgpu_buffer_rsrc_t rsW = ZRSRC((const T*)p.W + (long)tn * 256 * p.K);

  // cold start: K tile 0 and the B0 half of K tile 1, fully drained (so that the XS waits of the first
  // K tile, which assume S younger operations, cannot pass early)
  GW(rsW, 0, 0, 0) GA(rsA, srcA, 0, 0, 0) GW(rsW, 1, 0, 0) GA(rsA, srcA, 1, 0, 0)
  GW(rsW, 0, 1, 1)
  vec8 fm[4][2], fnX[2][2], fnY[2][2];
  wait_vm<0>();
  BAR
  LD_N(fnX, smem, 0)
  if (wr == 1) BAR   // waves 4-7 now run one segment behind waves 0-3

  f32x4 acc[8][4];
#ifdef Z_STAMP
  unsigned long long zsum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, zlast = zclock();
#endif
  for (;;) {
    // next tile of this workgroup (uniform scalar work)
    int nvb = vb + G, ntm = 0, ntn = 0;
    while (nvb < nvirtual && !map.decode(nvb, ntm, ntn)) nvb += G;
    const bool hn = nvb < nvirtual;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;

    ZT(0)   // tile bookkeeping + accumulator clear
    // first pair: the previous epilogue's stores may still be in flight behind this tile's DMAs
    KTILE(0, fnX, fnY, S, true, rsA, srcA, rsW, 1, true, rsW, 2, true)
    KTILE(1, fnY, fnX, 0, true, rsA, srcA, rsW, 2, true, rsW, 3, true)
    ZT(1)
    for (int kt = 2; kt < nk - 2; kt += 2) {
      KTILE(0, fnX, fnY, 0, true, rsA, srcA, rsW, kt + 1, true, rsW, kt + 2, true)
      KTILE(1, fnY, fnX, 0, true, rsA, srcA, rsW, kt + 2, true, rsW, kt + 3, true)
    }
    ZT(2)
    // last pair: the "following K tile" slots fetch the next output tile
    __amdgpu_buffer_rsrc_t rsAn = rsA, rsWn = rsW;
    if (hn) {
      ZSRCA(srcAn, ntm)
      rsAn = ZRSRC((const T*)p.A + (long)ntm * 256 * p.lda);
      rsWn = ZRSRC((const T*)p.W + (long)ntn * 256 * p.K);
    }
    KTILE(0, fnX, fnY, 0, true, rsA, srcA, rsW, nk - 1, hn, rsWn, 0, true)
    KTILE(1, fnY, fnX, 0, hn, rsAn, srcAn, rsWn, 0, hn, rsWn, 1, false)
    ZT(3)
    if (wr == 0) BAR   // both groups aligned for the epilogue
    ZT(4)
    epilogue256z<T, EPI>(p, acc, stg, tm, tn, wave, lane);
    ZT(5)
#ifdef Z_STAMP
    zsum[7] += 1;
#endif
    if (!hn) break;
    if (tm * 256 + 256 > p.M) wait_vm<0>();   // partial tile: fewer stores than S were issued
    LD_N(fnX, smem, 0)   // B0 of the next tile's K tile 0 (landed and confirmed by every wave in the last P3)
    if (wr == 1) BAR   // re-enter the stagger
    ZT(6)
    vb = nvb; tm = ntm; tn = ntn;
    rsA = rsAn; rsW = rsWn;
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
      for (int j = 0; j < 2; ++j) srcA[sub][j] = srcAn[sub][j];
  }
#ifdef Z_STAMP
  if (wave == 0 && lane == 0 && blockIdx.x < 512)
    for (int k = 0; k < 8; ++k) g_zstamp[8 * blockIdx.x + k] = zsum[k];
#endif
#undef ZSRCA
#undef ZRSRC
#undef DMA
#undef GA
#undef GW
#undef LGKM0
#undef BAR
#undef PINB
#undef LD_M
#undef LD_N
#undef MM
#undef QUADX
#undef KTILE
}

template <typename T>
static bool launch_z(int epi, const GemmParams& p, hipStream_t s) {
  const int nk = p.K >> 6;
  if ((nk & 1) || nk < 4) return false;
  static int n_cu = 0;
  if (!n_cu) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return false;
    n_cu = prop.multiProcessorCount > 0 ? (prop.multiProcessorCount / 8) * 8 : 256;
    if (n_cu < 8) n_cu = 8;
  }
  const int tiles_n = p.N / 256, tiles_m = (p.M + 255) / 256;
  const int PN = (tiles_n % 4 == 0) ? 4 : (tiles_n % 3 == 0) ? 3 : (tiles_n % 2 == 0) ? 2 : 1;
  const int PM = 8;
  const int patches_n = tiles_n / PN, patches_m = (tiles_m + PM - 1) / PM;
  const int total = patches_n * patches_m;
  const int nvirtual = ((total + 7) / 8) * 8 * PM * PN;
  const int grid = nvirtual < n_cu ? nvirtual : n_cu;   // both are multiples of 8: XCD = id & 7 is kept
  dim3 g(grid), b(512);
  switch (epi) {
    case EPI_BIAS: hipLaunchKernelGGL((gemm16_256z_kernel<T, EPI_BIAS>), g, b, 0, s, p, PN, patches_n, total, PM, nvirtual); break;
    case EPI_BIAS_GELU: hipLaunchKernelGGL((gemm16_256z_kernel<T, EPI_BIAS_GELU>), g, b, 0, s, p, PN, patches_n, total, PM, nvirtual); break;
    case EPI_BIAS_RESID: hipLaunchKernelGGL((gemm16_256z_kernel<T, EPI_BIAS_RESID>), g, b, 0, s, p, PN, patches_n, total, PM, nvirtual); break;
    case EPI_ACT_F32: hipLaunchKernelGGL((gemm16_256z_kernel<T, EPI_ACT_F32>), g, b, 0, s, p, PN, patches_n, total, PM, nvirtual); break;
    case EPI_PATCH: hipLaunchKernelGGL((gemm16_256z_kernel<T, EPI_PATCH>), g, b, 0, s, p, PN, patches_n, total, PM, nvirtual); break;
    default: return false;
  }
  return true;
}

void read_gemm_zstamps(double* out8) {
  for (int k = 0; k < 8; ++k) out8[k] = 0;
#ifdef Z_STAMP
  static unsigned long long host[8 * 512];
  (void)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_zstamp), sizeof(host));
  double tiles = 0;
  for (int w = 0; w < 512; ++w) {
    tiles += (double)host[8 * w + 7];
    for (int k = 0; k < 7; ++k) out8[k] += (double)host[8 * w + k];
  }
  for (int k = 0; k < 7; ++k) out8[k] = tiles > 0 ? out8[k] / tiles : 0;   // cycles per tile
  out8[7] = tiles;
#endif
}

bool launch_gemm256z(int dtype, int epi, const GemmParams& p, hipStream_t s) {
  return dtype == AACLIP_F16 ? launch_z<f16>(epi, p, s) : launch_z<bf16>(epi, p, s);
}

}  // namespace aaclip
