// NT GEMM with fused epilogues for the AA-CLIP towers on gfx950.
//
//   C[M,N] = A[M,K] . W[N,K]^T      (W is the nn.Linear weight as stored: [out,in])
//
// 16-bit operands (f16 / bf16), fp32 accumulate on v_mfma_f32_32x32x16:
//   128x128x64 block tile, 4 waves (2x2), 64x64 per wave, two LDS stages filled
//   by the 16-byte global->LDS DMA with the XOR-swizzled tile image of common.h,
//   one barrier per K tile, XCD-aware tile order.
// fp32 operands: exact-fp32 v_mfma_f32_32x32x2_f32 kernel (parity path).
//
// Epilogues (what the reference does right after each Linear):
//   EPI_BIAS       out_T = acc + bias, columns < scale_cols scaled (q * hd^-1/2,
//                  nn.MultiheadAttention, reference model/transformer.py:200,237)
//   EPI_BIAS_GELU  out_T = gelu_erf(acc + bias)          (mlp.c_fc, transformer.py:211-219)
//   EPI_BIAS_RESID x_f32 += acc + bias                    (out_proj / c_proj + residual, :256-257)
//   EPI_ACT_F32    out_f32 = act(acc [+ bias])            (adapters, seg/det proj: adapter_modules.py:6-26)
//   EPI_PATCH      x_f32[b*L+1+p] = acc + pos[1+p]        (conv1 as GEMM + positional, adapter.py:139-153)
#include "common.h"
#include "kernels.h"
#include "mma16.h"

namespace aaclip {

// 16-bit outputs.  SPLIT (AACLIP_F16X2, common.h): S8 = false -> split16 row (hi plane, lo plane N columns further: the
// attention kernel's input); S8 = true -> split8 row (hi plane, then the two e4m3 planes: the next product's A operand)
template <typename TOut, bool SPLIT, bool S8>
AACLIP_DEV void store16(const GemmParams& p, int row, int col, float v) {
  if constexpr (SPLIT) {
    f16* o = (f16*)p.out + (long)row * p.ldc;
    f16 hi, lo;
    split16(v, hi, lo);
    o[col] = hi;
    if (!S8) {
      o[p.N + col] = lo;
    } else {
      uint8_t* o8 = (uint8_t*)(o + p.N);
      o8[col] = (uint8_t)pack_e4m3x4<SPLIT8_ACT_LO_EXP>(v - (float)hi, 0.f, 0.f, 0.f);
      if (!p.out_no_hi8) o8[p.N + col] = (uint8_t)pack_e4m3x4<SPLIT8_ACT_HI_EXP>(v, 0.f, 0.f, 0.f);
    }
  } else {
    ((TOut*)p.out)[(long)row * p.ldc + col] = from_float<TOut>(v);
  }
}

template <int EPI, typename TOut, bool SPLIT = false>
AACLIP_DEV void epi_store(const GemmParams& p, int row, int col, float v) {
  if (EPI == EPI_BIAS) {
    v += p.bias[col];
    if (col < p.scale_cols) v *= p.scale;
    store16<TOut, SPLIT, false>(p, row, col, v);
  } else if (EPI == EPI_BIAS_GELU) {
    // 16-bit outputs: the polynomial erf of common.h (error far below the output rounding), same in every 16-bit kernel so that
    // results do not depend on which tile size a batch selects (split fp16 included: the polynomial's 1.2e-5 is 40x
    // below one fp16 rounding); fp32 keeps erff
    v = sizeof(TOut) == 4 ? gelu_erf(v + p.bias[col]) : gelu_fast(v + p.bias[col]);
    store16<TOut, SPLIT, true>(p, row, col, v);
  } else if (EPI == EPI_BIAS_RESID) {
    float* x = (float*)p.out + (long)row * p.ldc + col;
    const float r = p.resid ? p.resid[(long)row * p.ldc + col] : *x;
    *x = r + (v + p.bias[col]);
  } else if (EPI == EPI_ACT_F32) {
    if (p.bias) v += p.bias[col];
    if (p.act == 1) v = leaky(v);
    else if (p.act == 2) v = fmaxf(v, 0.f);
    ((float*)p.out)[(long)row * p.ldc + col] = v;
  } else if (EPI == EPI_PATCH) {
    int b = row / p.P, pi = row - b * p.P;
    long orow = (long)b * p.L + 1 + pi;
    ((float*)p.out)[orow * p.ldc + col] = v + p.pos[(long)(1 + pi) * p.N + col];
  }
}

// ------------------------------------------------------------------ 16-bit
template <typename T, int EPI>
__global__ __launch_bounds__(256, 2) void gemm16_kernel(GemmParams p) {
  typedef typename Elem<T>::vec8 vec8;
  __shared__ __attribute__((aligned(16))) char smem[65536];  // 2 stages x (A 16K + W 16K)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int wr = wave >> 1, wc = wave & 1;
  const int tiles_n = p.N >> 7;
  const int tiles_m = (p.M + 127) >> 7;
  const int t = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  const int tm = t / tiles_n, tn = t - tm * tiles_n;

  // --- DMA source pointers: 4 wave-instructions of 1 KiB for A and 4 for W per stage
  const T* asrc[4];
  const T* wsrc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    int row, chunk;
    tile_src((wave * 4 + j) * 64 + lane, row, chunk);
    int ar = tm * 128 + row;
    ar = ar < p.M ? ar : p.M - 1;
    asrc[j] = (const T*)p.A + (long)ar * p.lda + chunk * 8;
    wsrc[j] = (const T*)p.W + (long)(tn * 128 + row) * p.K + chunk * 8;
  }
  // --- fragment read offsets inside a tile
  int aoff[2][4], boff[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      aoff[i][ks] = tile_off(wr * 64 + i * 32 + r, 2 * ks + h);
      boff[i][ks] = 16384 + tile_off(wc * 64 + i * 32 + r, 2 * ks + h);
    }

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int nk = p.K >> 6;
  auto stage = [&](int s, int kt) {
    char* base = smem + s * 32768 + wave * 4096;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      glds16(asrc[j] + kt * 64, base + j * 1024);
      glds16(wsrc[j] + kt * 64, base + 16384 + j * 1024);
    }
  };

  stage(0, 0);
  wait_vm0();
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
    const char* sb = smem + cur * 32768;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      vec8 a0 = *(const vec8*)(sb + aoff[0][ks]);
      vec8 a1 = *(const vec8*)(sb + aoff[1][ks]);
      vec8 b0 = *(const vec8*)(sb + boff[0][ks]);
      vec8 b1 = *(const vec8*)(sb + boff[1][ks]);
      acc[0][0] = Elem<T>::mma32(a0, b0, acc[0][0]);
      acc[0][1] = Elem<T>::mma32(a0, b1, acc[0][1]);
      acc[1][0] = Elem<T>::mma32(a1, b0, acc[1][0]);
      acc[1][1] = Elem<T>::mma32(a1, b1, acc[1][1]);
    }
    wait_vm0();
    __syncthreads();
  }

  // --- epilogue: lane holds column r of each 32x32 sub-tile, rows (e&3)+8(e>>2)+4h
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = tm * 128 + wr * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
      if (row < p.M) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int col = tn * 128 + wc * 64 + j * 32 + r;
          epi_store<EPI, T>(p, row, col, acc[i][j][e]);
        }
      }
    }
}

// ------------------------------------------------------------------ fp32
// Exact fp32: v_mfma_f32_32x32x2_f32 is bit-for-bit a k-ordered fmaf chain.
// T = 128: 128 x 128 tile, a wave owns 64 x 64 (2 x 2 MFMA tiles).  T = 64: 64 x 64 tile, a wave owns 32 x 32 -- for small
// grids (the IQM branch's [128..1024, 768] products under fp32 / fp16x2: 6-48 tiles of 128 on 256 CUs, each tile a serial
// chain of K/16 x 2048 MFMA cycles): four times the workgroups and a quarter of the chain per workgroup, 85 -> ~25 us.
// Every output element sums its k terms in the same order in both forms: results are bit-identical.
template <int EPI, int T = 128>
__global__ __launch_bounds__(256) void gemm32_kernel(GemmParams p) {
  constexpr int LD = T + 4, NJ = T / 64, NI = T / 64;   // NJ float4 per thread and operand; NI x NI MFMA tiles per wave
  __shared__ float As[16 * LD];
  __shared__ float Ws[16 * LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int wr = wave >> 1, wc = wave & 1;
  const int tiles_n = p.N / T;
  const int tiles_m = (p.M + T - 1) / T;
  const int t = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  const int tm = t / tiles_n, tn = t - tm * tiles_n;

  const float* ap[NJ];
  const float* wp[NJ];
  int srow[NJ], sk[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    int f = tid + 256 * j;
    srow[j] = f >> 2;
    sk[j] = (f & 3) * 4;
    int ar = tm * T + srow[j];
    ar = ar < p.M ? ar : p.M - 1;
    ap[j] = (const float*)p.A + (long)ar * p.lda + sk[j];
    wp[j] = (const float*)p.W + (long)(tn * T + srow[j]) * p.K + sk[j];
  }
  f32x16 acc[NI][NI];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int nk = p.K >> 4;
  f32x4 ra[NJ], rw[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    ra[j] = *(const f32x4*)(ap[j]);
    rw[j] = *(const f32x4*)(wp[j]);
  }
  for (int kt = 0; kt < nk; ++kt) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        As[(sk[j] + e) * LD + srow[j]] = ra[j][e];
        Ws[(sk[j] + e) * LD + srow[j]] = rw[j][e];
      }
    __syncthreads();
    if (kt + 1 < nk) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        ra[j] = *(const f32x4*)(ap[j] + (kt + 1) * 16);
        rw[j] = *(const f32x4*)(wp[j] + (kt + 1) * 16);
      }
    }
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const int k = 2 * ks + h;
      float a[NI], b[NI];
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        a[i] = As[k * LD + wr * (T / 2) + 32 * i + r];
        b[i] = Ws[k * LD + wc * (T / 2) + 32 * i + r];
      }
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  }
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = tm * T + wr * (T / 2) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
      if (row < p.M) {
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          const int col = tn * T + wc * (T / 2) + j * 32 + r;
          epi_store<EPI, float>(p, row, col, acc[i][j][e]);
        }
      }
    }
}

template <typename T>
static void launch16(int epi, const GemmParams& p, dim3 g, hipStream_t s) {
  switch (epi) {
    case EPI_BIAS: hipLaunchKernelGGL((gemm16_kernel<T, EPI_BIAS>), g, dim3(256), 0, s, p); break;
    case EPI_BIAS_GELU: hipLaunchKernelGGL((gemm16_kernel<T, EPI_BIAS_GELU>), g, dim3(256), 0, s, p); break;
    case EPI_BIAS_RESID: hipLaunchKernelGGL((gemm16_kernel<T, EPI_BIAS_RESID>), g, dim3(256), 0, s, p); break;
    case EPI_ACT_F32: hipLaunchKernelGGL((gemm16_kernel<T, EPI_ACT_F32>), g, dim3(256), 0, s, p); break;
    case EPI_PATCH: hipLaunchKernelGGL((gemm16_kernel<T, EPI_PATCH>), g, dim3(256), 0, s, p); break;
  }
}

// ------------------------------------------------------------------ split fp16 (AACLIP_F16X2), small M
// The 128x128 counterpart of the 256-tile split kernel for small batches (text tower, heads, unit tests): split8
// operands (common.h), per pair of K tiles two fp16 tiles on v_mfma_f32_16x16x32_f16 and two (one) e4m3 correction
// tiles on the block-scaled 16x16x128 MFMA.  4 waves (2 x 2), 64 x 64 of C per wave as 4 x 4 tiles of 16 x 16; two LDS
// stages of (A 16 KiB + W 16 KiB) filled by the 16-byte DMA, one barrier per tile.  A rows are the MFMA's A side, so a
// lane holds output column n = lane & 15 and rows m = 4 (lane >> 4) + j of each 16 x 16 tile.
template <int EPI, int NP>
__global__ __launch_bounds__(256, 2) void gemm16s_kernel(GemmParams p) {
  __shared__ __attribute__((aligned(16))) char smem[65536];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c16 = lane & 15, q4 = lane >> 4;
  const int wr = wave >> 1, wc = wave & 1;
  const int tiles_n = p.N >> 7;
  const int tiles_m = (p.M + 127) >> 7;
  const int t = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  const int tm = t / tiles_n, tn = t - tm * tiles_n;
  const long ldw = NP == 4 ? 2L * p.K : p.K + (p.K >> 1);   // W row stride in halves (4K / 3K bytes)

  const char* asrc[4];
  const char* wsrc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    int row, chunk;
    tile_src_s((wave * 4 + j) * 64 + lane, row, chunk);
    int ar = tm * 128 + row;
    ar = ar < p.M ? ar : p.M - 1;
    asrc[j] = (const char*)p.A + ((long)ar * p.lda) * 2 + chunk * 16;
    wsrc[j] = (const char*)p.W + ((long)(tn * 128 + row) * ldw) * 2 + chunk * 16;
  }
  int aoff[4][2], boff[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      aoff[i][ks] = tile_off_s(wr * 64 + i * 16 + c16, 4 * ks + q4);
      boff[i][ks] = 16384 + tile_off_s(wc * 64 + i * 16 + c16, 4 * ks + q4);
    }
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nk = vtile_count<NP>(p.K);
  auto stage = [&](int s, int kt) {
    char* base = smem + s * 32768 + wave * 4096;
    int kind;
    const int off = vtile_off<NP>(kt, p.K, kind);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      glds16(asrc[j] + off, base + j * 1024);
      glds16(wsrc[j] + off, base + 16384 + j * 1024);
    }
  };
  stage(0, 0);
  wait_vm0();
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
    const char* sb = smem + cur * 32768;
    int kind;
    (void)vtile_off<NP>(kt, p.K, kind);
    f16x8 a[4][2], b[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        a[i][ks] = *(const f16x8*)(sb + aoff[i][ks]);
        b[i][ks] = *(const f16x8*)(sb + boff[i][ks]);
      }
    if (kind == 0) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i][ks], b[j][ks], acc[i][j], 0, 0, 0);
    } else {
      const int sa = vtile_scale_act(kind), sw = vtile_scale_w(kind);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = mma_e4m3(a[i][0], a[i][1], b[j][0], b[j][1], acc[i][j], sa, sw);
    }
    wait_vm0();
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int row = tm * 128 + wr * 64 + i * 16 + 4 * q4 + e;
      if (row < p.M) {
#pragma unroll
        for (int j = 0; j < 4; ++j) epi_store<EPI, f16, true>(p, row, tn * 128 + wc * 64 + j * 16 + c16, acc[i][j][e]);
      }
    }
}

template <int NP>
static void launch16s(int epi, const GemmParams& p, dim3 g, hipStream_t s) {
  switch (epi) {
    case EPI_BIAS: hipLaunchKernelGGL((gemm16s_kernel<EPI_BIAS, NP>), g, dim3(256), 0, s, p); break;
    case EPI_BIAS_GELU: hipLaunchKernelGGL((gemm16s_kernel<EPI_BIAS_GELU, NP>), g, dim3(256), 0, s, p); break;
    case EPI_BIAS_RESID: hipLaunchKernelGGL((gemm16s_kernel<EPI_BIAS_RESID, NP>), g, dim3(256), 0, s, p); break;
    case EPI_ACT_F32: hipLaunchKernelGGL((gemm16s_kernel<EPI_ACT_F32, NP>), g, dim3(256), 0, s, p); break;
    case EPI_PATCH: hipLaunchKernelGGL((gemm16s_kernel<EPI_PATCH, NP>), g, dim3(256), 0, s, p); break;
  }
}

const char* gemm_check(int dtype, int epi, const GemmParams& p) {
  if (p.M <= 0 || p.N <= 0 || p.K <= 0) return "gemm: empty problem";
  if (p.N % 128) return "gemm: N must be a multiple of 128";
  if (p.K % (dtype == AACLIP_F32 ? 16 : 64)) return "gemm: K must be a multiple of 64 (16 for f32)";
  if (p.lda % 8 || p.ldc % 2) return "gemm: lda must be a multiple of 8, ldc of 2";
  if (dtype == AACLIP_F16X2) {
    if (p.K % 128) return "gemm: split fp16 products take K tiles in pairs: K must be a multiple of 128";
    if (p.lda < 2L * p.K) return "gemm: split8 rows of A hold 4 bytes per element: lda (in halves) must be >= 2K";
    if ((epi == EPI_BIAS || epi == EPI_BIAS_GELU) && p.ldc < 2L * p.N)
      return "gemm: split output rows hold 4 bytes per element: ldc (in halves) must be >= 2N";
    if ((long)p.K >= (1L << 20)) return "gemm: K too large for the split kernels";
  }
  if (epi < 0 || epi > EPI_PATCH) return "gemm: unknown epilogue";
  if ((epi == EPI_BIAS || epi == EPI_BIAS_GELU || epi == EPI_BIAS_RESID) && !p.bias) return "gemm: bias required";
  if (epi == EPI_PATCH && (!p.pos || p.P <= 0 || p.L <= p.P)) return "gemm: patch epilogue needs pos, P, L";
  if (!p.A || !p.W || !p.out) return "gemm: null pointer";
  return nullptr;
}

static int g_tail_peel = 0;  // measured: not a win (the 128-tile kernel is too slow for the peeled rows)
void set_tail_peel(int v) { g_tail_peel = v; }
// 0 automatic, 1 force the 128-tile kernel, 80 / 81 / 82 force the 8-wave 256 x 256 kernel (one tile per workgroup) / the
// 4-wave 256 x 128 half-tile kernel / the walking 8-wave kernel (all bit-identical); more in the measurement library only
static int g_gemm_variant = 0;
// launch_gemm256t kernel id: 14 = 8 waves, 256 x 256 tile; 15 = 4 waves, 256 x 128 half tile, two workgroups per CU.
// Measured (profiles/r04_gemm_half_tile_ab.txt): the half tile's K loop is 6-12 % slower (12 instead of 8 DMA
// pieces per 64 MFMAs on a loop that is bound by exactly that); the overlap of one workgroup's epilogue with the other's K
// loop wins it back only on isolated epilogue-heavy 16-bit products (out_proj +9...12 %, seg / det projections +10 %) and
// not inside the tower, where the same products carry the LayerNorm-folding epilogue (tower: 917 vs 923 images/s with
// them on the half tile, 878 with every product on it); split operands lose 2-8 % on every shape.  So the automatic
// choice is the 8-wave kernel everywhere and the half tile stays selectable (variant 81).
// 16 = the 8-wave kernel WALKING its tiles (one workgroup per CU; the next tile's first K tile is fetched under the
// epilogue of the current one; split operands only, launch_gemm256t falls back to 14 for the others): bit-identical,
// tower +0.5 % (profiles/r04_gemm_walk_ab.txt).  Automatic for split operands; 80 forces the one-tile-per-workgroup form.
static int big_kernel_id(int dtype, const GemmParams& p) {
  (void)p;
  if (g_gemm_variant == 81) return 15;
  if (g_gemm_variant == 80) return 14;
  return (dtype == AACLIP_F16X2 || g_gemm_variant == 82) ? 16 : 14;
}

static thread_local const char* g_launch_err = nullptr;
void set_launch_error(const char* msg) { if (!g_launch_err) g_launch_err = msg; }
const char* take_launch_error() { const char* m = g_launch_err; g_launch_err = nullptr; return m; }

// split fp16: K tiles are taken in pairs, and the 256-tile kernel walks an even number of virtual tiles
// (4 per pair, or 3 when the weight is exact in fp16: then K must be a multiple of 256)
static bool split256_applicable(const GemmParams& p) {
  return p.N % 256 == 0 && p.K % (p.w_exact16 ? 256 : 128) == 0 && p.ldc % 8 == 0 && (p.scale_cols % 4) == 0 &&
         (long)256 * p.lda < (1L << 29) && (long)256 * p.K < (1L << 28);
}

bool gemm256_applicable(int dtype, const GemmParams& p) {
  if (dtype == AACLIP_F16X2) return split256_applicable(p);
  return dtype != AACLIP_F32 && p.N % 256 == 0 && p.K % 64 == 0 && p.ldc % 8 == 0 && (p.scale_cols % 4) == 0 &&
         (long)256 * p.lda < (1L << 30) && (long)256 * p.K < (1L << 30);
}

bool set_gemm_variant(int v) {
#ifdef AACLIP_MEASURE
  // 2..5 = 256-tile kernels on 32x32x16 MFMAs (gemm256.hip), 6..60 = the 16x16x32 family incl. timing ablations and
  // the stamp build (gemm256t.hip), 70 = persistent tiles (gemm256z.hip)
  const bool ok = v >= 0 && (v <= 60 || v == 70 || v == 80 || v == 81 || v == 82);
#else
  const bool ok = v == 0 || v == 1 || v == 80 || v == 81 || v == 82;
#endif
  if (ok) g_gemm_variant = v;
  return ok;
}

static void launch_gemm_big(int dtype, int epi, const GemmParams& p, hipStream_t s) {
#ifdef AACLIP_MEASURE
  if (g_gemm_variant == 70) {   // persistent tiles (gemm256z.hip); falls through when K/64 is odd or < 4
    if (launch_gemm256z(dtype, epi, p, s)) return;
    launch_gemm256t(dtype, epi, p, s, 14);
    return;
  }
  if (g_gemm_variant >= 6 && g_gemm_variant <= 60) {
    // 16x16x32 MFMA shape; 6 plain, 7 overlapped LDS reads, 8/9/10 staggered with 0/1/2 DMA issues in the load
    // segment, 11..17 timing ablations / stamps of 10 (fp32-out epilogue only), 18 staggered + in-cluster reads,
    // 19 = 10 with buffer_load ... lds, 20 = the default
    launch_gemm256t(dtype, epi, p, s, g_gemm_variant - 6);
    return;
  }
  if (g_gemm_variant >= 2 && g_gemm_variant < 80) {
    // 32x32x16 kernels: 2 DMA at the phase start, 3 DMA between the MFMAs, 4/5 timing ablations
    launch_gemm256(dtype, epi, p, s, g_gemm_variant == 2 ? 0 : g_gemm_variant - 2);
    return;
  }
#endif
  launch_gemm256t(dtype, epi, p, s, big_kernel_id(dtype, p));
}

// True when launch_gemm will run one of the 16x16x32 256-tile kernels (gemm256t.hip) on the whole problem:
// those are the kernels whose epilogue implements the LayerNorm-folding options of GemmParams.
bool gemm_routes_to_256t(int dtype, const GemmParams& p) {
  if (dtype == AACLIP_F32 || dtype == AACLIP_F16X2 || !gemm256_applicable(dtype, p) || p.M < 4096) return false;
  if (!(g_gemm_variant == 0 || g_gemm_variant >= 80 || (g_gemm_variant >= 6 && g_gemm_variant <= 60))) return false;   // 2..70: measurement library
  if (g_tail_peel) return false;
  return true;
}

// split fp16: true when launch_gemm will run the 256-tile kernel -- the one whose EPI_BIAS epilogue can write the
// attention kernel's e4m3 records (GemmParams::out_qk8)
bool gemm_split_routes_to_256t(const GemmParams& p) { return g_gemm_variant != 1 && split256_applicable(p) && p.M >= 4096; }

void launch_gemm(int dtype, int epi, const GemmParams& p, hipStream_t s) {
  if (dtype == AACLIP_F16X2) {   // split fp16: the default 256-tile kernel from M = 4096 rows, else the 128-tile kernel
    if (gemm_split_routes_to_256t(p)) {
      launch_gemm256t(dtype, epi, p, s, big_kernel_id(dtype, p));
      return;
    }
    if (p.out_qk8) { set_launch_error("gemm: out_qk8 needs the 256-tile kernel"); return; }
    dim3 g(((p.M + 127) / 128) * (p.N / 128));
    if (p.w_exact16) launch16s<3>(epi, p, g, s);
    else launch16s<4>(epi, p, g, s);
    return;
  }
  // variants: 0 automatic (256-tile kernels from M = 4096 rows), 1 the 128-tile kernel, >= 2 (measurement library) 256-tile
  // kernels at any M
  if (g_gemm_variant != 1 && gemm256_applicable(dtype, p) && ((g_gemm_variant >= 2 && g_gemm_variant < 80) || p.M >= 4096)) {
    // Tail peeling: 256x256 tiles run one per CU in rounds of 256.  When the last round would be
    // less than 60 % full, the rows of that partial round go to the 128-tile kernel instead (two
    // workgroups per CU, finer granularity); both kernels produce bit-identical results.
    const int tiles_n = p.N / 256, tiles_m = (p.M + 255) / 256;
    const long tiles = (long)tiles_m * tiles_n;
    const long rem = tiles % 256;
    if (g_tail_peel && epi != EPI_PATCH && tiles > 256 && rem > 0 && rem < 154) {
      const int m_full = (int)((tiles - rem) / tiles_n);   // whole M tiles covered by full rounds
      const long rows_full = (long)m_full * 256;
      if (rows_full > 0 && rows_full < p.M) {
        GemmParams a = p, b = p;
        a.M = (int)rows_full;
        const size_t es = 2, os = (epi == EPI_BIAS || epi == EPI_BIAS_GELU) ? 2 : 4;
        b.A = (const char*)p.A + rows_full * p.lda * es;
        b.out = (char*)p.out + rows_full * p.ldc * os;
        b.M = p.M - (int)rows_full;
        launch_gemm_big(dtype, epi, a, s);
        const int t128 = ((b.M + 127) / 128) * (b.N / 128);
        if (dtype == AACLIP_F16) launch16<f16>(epi, b, dim3(t128), s);
        else launch16<bf16>(epi, b, dim3(t128), s);
        return;
      }
    }
    launch_gemm_big(dtype, epi, p, s);
    return;
  }
  const int tiles = ((p.M + 127) / 128) * (p.N / 128);
  dim3 g(tiles);
  if (dtype == AACLIP_F16) {
    launch16<f16>(epi, p, g, s);
  } else if (dtype == AACLIP_BF16) {
    launch16<bf16>(epi, p, g, s);
  } else {
    if (tiles < 128) {   // fewer 128-tiles than half the CUs: 64 x 64 tiles (same sums, four times the workgroups)
      dim3 g64(((p.M + 63) / 64) * (p.N / 64));
      switch (epi) {
        case EPI_BIAS: hipLaunchKernelGGL((gemm32_kernel<EPI_BIAS, 64>), g64, dim3(256), 0, s, p); break;
        case EPI_BIAS_GELU: hipLaunchKernelGGL((gemm32_kernel<EPI_BIAS_GELU, 64>), g64, dim3(256), 0, s, p); break;
        case EPI_BIAS_RESID: hipLaunchKernelGGL((gemm32_kernel<EPI_BIAS_RESID, 64>), g64, dim3(256), 0, s, p); break;
        case EPI_ACT_F32: hipLaunchKernelGGL((gemm32_kernel<EPI_ACT_F32, 64>), g64, dim3(256), 0, s, p); break;
        case EPI_PATCH: hipLaunchKernelGGL((gemm32_kernel<EPI_PATCH, 64>), g64, dim3(256), 0, s, p); break;
      }
      return;
    }
    switch (epi) {
      case EPI_BIAS: hipLaunchKernelGGL((gemm32_kernel<EPI_BIAS>), g, dim3(256), 0, s, p); break;
      case EPI_BIAS_GELU: hipLaunchKernelGGL((gemm32_kernel<EPI_BIAS_GELU>), g, dim3(256), 0, s, p); break;
      case EPI_BIAS_RESID: hipLaunchKernelGGL((gemm32_kernel<EPI_BIAS_RESID>), g, dim3(256), 0, s, p); break;
      case EPI_ACT_F32: hipLaunchKernelGGL((gemm32_kernel<EPI_ACT_F32>), g, dim3(256), 0, s, p); break;
      case EPI_PATCH: hipLaunchKernelGGL((gemm32_kernel<EPI_PATCH>), g, dim3(256), 0, s, p); break;
    }
  }
}

}  // namespace aaclip
