// 256x256x64 NT GEMM on v_mfma_f32_16x16x32 (gfx950).
//
// Same workgroup geometry, phase schedule and DMA placement as gemm16_256s_kernel
// (gemm256.hip): 8 waves = 2 (M) x 4 (N), 128x64 of C per wave, four output
// quadrants per K tile, the two DMA instructions of a phase issued between its
// MFMAs, counted vmcnt waits.  The differences:
//   * MFMA shape 16x16x32 instead of 32x32x16.  Cycles per FLOP are equal, but in
//     an MFMA-dense loop on random data the chip holds a higher clock on this
//     shape (MI355X_MICROARCH.md, DVFS give-back item 7), so it is faster by wall.
//   * LDS tile swizzle slot ^= (row_pair & 15), which is bank-conflict-free for
//     the 16x16x32 operand read (lane -> row l&15, chunk 4*ks + (l>>4)) as well
//     as for the 32x32x16 one (tools/lds_bank_model.py).
// Operand roles are swapped as in the other kernel (W rows = MFMA A operand), so a
// lane holds output column m = lane&15 and rows n = 4*(lane>>4) + j of each 16x16 tile.
#include <stdio.h>
#include <stdlib.h>
#include "common.h"
#include "kernels.h"
#include "mma16.h"

namespace aaclip {

// Output tiles are written once and read by a later kernel, and the residual is read once: non-temporal
// accesses keep them from displacing the operand tiles in L2 (measured: c_fc +6 %, out_proj +12 %).
#define ST_OUT(ptr, v) __builtin_nontemporal_store(v, ptr)   // the nt bit is what helps; sc0/sc1 made no difference
#define LD_RESID(ptr) __builtin_nontemporal_load(ptr)
template <typename P> AACLIP_DEV P* uniform_ptr(P* ptr) {
  const unsigned long long a = (unsigned long long)ptr;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  return (P*)(((unsigned long long)hi << 32) | lo);
}
#if defined(AACLIP_MEASURE) && defined(X_WALK_STAMP)   // tools/walk_stamps.py: the compact epilogue in segments (wave 0 of a workgroup)
__device__ unsigned long long g_estamp[8];
AACLIP_DEV unsigned long long estamp() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
  return t;
}
#define ES(...) __VA_ARGS__
#else
#define ES(...)
#endif
// sum over the 16 lanes of a DPP row (rotations by 8, 4, 2, 1), result in every lane
AACLIP_DEV float row16_sum(float x) {
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x128, 0xF, 0xF, false));
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x124, 0xF, 0xF, false));
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x122, 0xF, 0xF, false));
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x121, 0xF, 0xF, false));
  return x;
}
// acc[mi][ni][j]: m = mi*16 + (lane&15), n = ni*16 + 4*(lane>>4) + j   (mi 0..7, ni 0..3)
// NWC = waves along N: 4 for the 256 x 256 tile of 8 waves, 2 for the 256 x 128 half tile of 4 waves (gemm16_256h_kernel)
// COMPACT: the staging area of a wave is 8 KiB instead of 16 (split operands only): the walking kernel
// (gemm16_256x_kernel<..., WALK>) stages in ONE operand stage while the next tile's first pieces land in the other
template <typename T, int EPI, bool SPLIT = false, bool QK8 = false, int NWC = 4, bool COMPACT = false>
AACLIP_DEV void epilogue256t(const GemmParams& p, f32x4 (&acc)[8][4], char* smem, int tm, int tn, int wave, int lane,
                             const f32x2* ab_pre = nullptr) {
  static_assert(!COMPACT || SPLIT, "the compact staging geometry exists for the split-operand epilogues");
  constexpr int STG = COMPACT ? 8192 : 16384;      // staging bytes per wave
  constexpr int SROWS = COMPACT ? 31 : 63;         // row mask of a 16-bit staging pass
  constexpr int SPL = COMPACT ? 4096 : 8192;       // offset of the e4m3 planes (or the lo tile) behind the hi tile
  typedef typename Elem<T>::vec4 vec4;
  // Everything the epilogue addresses with is derived from `lane` below this point: the empty asm keeps hipcc from
  // computing it before the K loop and carrying it through (measured: the folding code alone cost the residual
  // GEMMs 15 % that way, with the K loop unchanged in source).
  asm volatile("" : "+v"(lane));
  const int c16 = lane & 15, q4 = lane >> 4;
  const int wr = NWC == 4 ? wave >> 2 : wave >> 1, wc = wave & (NWC - 1);
  const int m_base = tm * 256 + wr * 128, n_base = tn * (NWC * 64) + wc * 64;
  if (EPI == EPI_BIAS || EPI == EPI_BIAS_GELU) {
    __syncthreads();  // every wave is done reading the operand tiles
    char* st = smem + wave * STG;  // this wave's 128 x 64 tile of T: rows of 128 B, 16-B chunk ^= (m & 7)
    const bool fold = p.row_ab != nullptr;   // LayerNorm folded into this product: acc -> a_m * acc + b_m * s_n
    f32x2 ab[8];
    if (fold) {
#pragma unroll
      for (int mi = 0; mi < 8; ++mi) {
        if (ab_pre) {
          ab[mi] = ab_pre[mi];   // fetched before the K loop by the caller (their latency is exposed here otherwise)
        } else {
          int row = m_base + mi * 16 + c16;
          row = row < p.M ? row : p.M - 1;
          ab[mi] = *(const f32x2*)(p.row_ab + 2L * row);
        }
      }
    }
    // per-column vectors of this lane's four column groups
    f32x4 bv[4], sv[4];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      bv[ni] = *(const f32x4*)(p.bias + n_base + ni * 16 + 4 * q4);
      sv[ni] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (fold) sv[ni] = *(const f32x4*)(p.col_s + n_base + ni * 16 + 4 * q4);
    }
    // four passes of 32 rows: convert (GELU) + stage + store one pass, then the next, so that the stores of a
    // pass drain while the VALU works on the following one (all CUs reach this point together and the stores
    // are bandwidth-bound: computing everything first and storing afterwards serialises the two)
    ES(unsigned long long es_cv = 0, es_st = 0;)
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
      ES(const unsigned long long es0 = estamp();)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        const int nl = ni * 16 + 4 * q4;   // local column of this lane's 4 values
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          const int mi = 2 * pass + hh;
          if (fold) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[mi][ni][j] = fmaf(ab[mi][0], acc[mi][ni][j], ab[mi][1] * sv[ni][j]);
          }
          vec4 o, o2;
          float vv[4];
          if (EPI == EPI_BIAS_GELU) {
            f32x2 g0 = {acc[mi][ni][0] + bv[ni][0], acc[mi][ni][1] + bv[ni][1]};
            f32x2 g1 = {acc[mi][ni][2] + bv[ni][2], acc[mi][ni][3] + bv[ni][3]};
            g0 = gelu_fast2(g0);
            g1 = gelu_fast2(g1);
            vv[0] = g0[0]; vv[1] = g0[1]; vv[2] = g1[0]; vv[3] = g1[1];
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              float v = acc[mi][ni][j] + bv[ni][j];
              if (n_base + nl + j < p.scale_cols) v *= p.scale;
              vv[j] = v;
            }
          }
          const int m = mi * 16 + c16;
          if constexpr (SPLIT && EPI == EPI_BIAS) {
            if constexpr (QK8) {
              // attention records for the e4m3 correction form (attention.hip, QK8): hi tile as below; the e4m3 planes
              // of this wave's 64 columns (ONE head of q or k) share a 128-byte staging row [lo8 | hi8], which is also
              // the record's layout in memory.  v columns (>= out_qk8) carry no correction.
              char* sp = st + (m & SROWS) * 128;
              if (n_base < p.out_qk8) {   // wave-uniform: this wave's 64 columns are one head of q or of k
                uint32_t l8, h8;
                split8x4_sat(vv, o, l8, h8);   // (the split kernels run fp8_saturate_mode() at entry)
                *(uint32_t*)(sp + SPL + (((nl >> 4) ^ (m & 7)) << 4) + (nl & 12)) = l8;
                *(uint32_t*)(sp + SPL + (((4 + (nl >> 4)) ^ (m & 7)) << 4) + (nl & 12)) = h8;
              } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = (f16)vv[j];
              }
              *(vec4*)(sp + ((((nl >> 3)) ^ (m & 7)) << 4) + (nl & 4) * 2) = o;
            } else {
            // split16 rows (attention inputs): the hi and lo tiles of a pass are staged side by side (passes 2, 3
            // reuse the LDS of passes 0, 1: a wave's LDS accesses execute in order) and stored as two row segments
            // N columns apart
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              f16 hi, lo;
              split16(vv[j], hi, lo);
              o[j] = hi;
              o2[j] = lo;
            }
            char* sp = st + (m & SROWS) * 128 + ((((nl >> 3)) ^ (m & 7)) << 4) + (nl & 4) * 2;
            *(vec4*)sp = o;
            *(vec4*)(sp + SPL) = o2;
            }
          } else if constexpr (SPLIT) {
            // split8 rows (the next product's A operand): hi tile as above; the e4m3 planes of a row share one
            // 128-byte staging row, [lo8: 64 bytes][hi8: 64 bytes], 16-byte chunks swizzled like the hi tile
            uint32_t l8, h8;
            split8x4_sat(vv, o, l8, h8);   // (the split kernels run fp8_saturate_mode() at entry)
            char* sp = st + (m & SROWS) * 128;
            *(vec4*)(sp + ((((nl >> 3)) ^ (m & 7)) << 4) + (nl & 4) * 2) = o;
            *(uint32_t*)(sp + SPL + (((nl >> 4) ^ (m & 7)) << 4) + (nl & 12)) = l8;
            *(uint32_t*)(sp + SPL + (((4 + (nl >> 4)) ^ (m & 7)) << 4) + (nl & 12)) = h8;
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = from_float<T>(vv[j]);
            *(vec4*)(st + m * 128 + ((((nl >> 3)) ^ (m & 7)) << 4) + (nl & 4) * 2) = o;
          }
        }
      }
      ES(const unsigned long long es1 = estamp(); es_cv += es1 - es0;)
      // Read the pass back (all LDS reads first: one wait instead of one per store) and store it.  Addresses: one 64-bit
      // base per lane for the whole tile (row m_base + lane / 8, this lane's chunk), plus a wave-uniform row offset per
      // store; the per-lane row test only in tiles that reach past M (a wave-uniform branch).  Before: ~25 instructions
      // of 64-bit multiplies, compares and exec-mask branches per stored row group, and every ds_read waited for alone.
      {
        const int c = lane & 7, r8 = lane >> 3;
        u32x4 vh[4], vl[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int m = (4 * pass + k) * 8 + r8;
          if constexpr (SPLIT) {
            const char* sp = st + (m & SROWS) * 128 + ((c ^ (m & 7)) << 4);
            vh[k] = *(const u32x4*)sp;
            vl[k] = *(const u32x4*)(sp + SPL);
          } else {
            vh[k] = *(const u32x4*)(st + m * 128 + ((c ^ (m & 7)) << 4));
          }
        }
        char* const hb = (char*)((T*)p.out + (long)(m_base + r8) * p.ldc + n_base + c * 8);   // hi / 16-bit chunk of row r8
        // second store of a split row: lo tile (split16), e4m3 record (QK8) or e4m3 planes (split8)
        char* lb = hb;
        bool second = false;
        if constexpr (SPLIT) {
          char* const rb = (char*)((T*)p.out + (long)(m_base + r8) * p.ldc);
          if constexpr (EPI == EPI_BIAS) {
            if constexpr (QK8) {
              lb = rb + 2 * p.N + (n_base >> 6) * 128 + c * 16;
              second = n_base < p.out_qk8;
            } else {
              lb = rb + ((long)p.N + n_base + c * 8) * sizeof(T);
              second = true;
            }
          } else {
            lb = rb + 2 * p.N + (c >> 2) * p.N + n_base + (c & 3) * 16;
            second = c < 4 || !p.out_no_hi8;
          }
        }
        const long rstride = (long)p.ldc * sizeof(T) * 8;   // bytes between consecutive row groups (wave-uniform)
        const bool inside = m_base + 128 <= p.M;            // wave-uniform: every row of this wave's band exists
        if (inside) {   // (two copies of the four stores: the common one has no per-lane row test and no exec-mask branches)
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int it = 4 * pass + k;
            ST_OUT((u32x4*)(hb + it * rstride), vh[k]);
            if constexpr (SPLIT) {
              if (second) ST_OUT((u32x4*)(lb + it * rstride), vl[k]);
            }
          }
        } else {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int it = 4 * pass + k;
            if (m_base + it * 8 + r8 < p.M) {
              ST_OUT((u32x4*)(hb + it * rstride), vh[k]);
              if constexpr (SPLIT) {
                if (second) ST_OUT((u32x4*)(lb + it * rstride), vl[k]);
              }
            }
          }
        }
      }
      ES(es_st += estamp() - es1;)
    }
    ES(if (COMPACT && wave == 0 && lane == 0) { atomicAdd(&g_estamp[0], es_cv); atomicAdd(&g_estamp[1], es_st); atomicAdd(&g_estamp[2], 1ull); })
  } else {
    // fp32 outputs.  In the accumulator layout the 16 lanes of a quarter-wave hold 16 different rows, i.e. one
    // global instruction would touch 64 cache lines for 1 KiB; staged through LDS (64 rows x 256 B per
    // half, 16-B chunk ^= row & 15, conflict-free both ways) one instruction covers 4 rows x 256 B = 8 lines.
    __syncthreads();  // every wave is done reading the operand tiles
    char* st = smem + wave * STG;
    constexpr int PARTS = COMPACT ? 4 : 2;         // row groups staged one after the other (64 or 32 rows each)
    constexpr int PIT = 32 / PARTS, PMI = 8 / PARTS;
    const int cc = lane & 15, rr = lane >> 4;      // read-back: chunk (4 columns) and row-in-group of this lane
    const int n0 = n_base + cc * 4;
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (EPI == EPI_BIAS_RESID || (EPI == EPI_ACT_F32 && p.bias)) bv = *(const f32x4*)(p.bias + n0);
#pragma unroll
    for (int half = 0; half < PARTS; ++half) {
      ES(const unsigned long long ef0 = estamp();)
      f32x4 extra[PIT];
      long orow[PIT];
#pragma unroll
      for (int it = 0; it < PIT; ++it) {
        const int row = m_base + half * (4 * PIT) + it * 4 + rr;
        orow[it] = row;
        if (EPI == EPI_PATCH) {
          const int rc = row < p.M ? row : p.M - 1;
          const int b = rc / p.P, pi = rc - b * p.P;
          orow[it] = (long)b * p.L + 1 + pi;
          extra[it] = *(const f32x4*)(p.pos + (long)(1 + pi) * p.N + n0);
        } else if (EPI == EPI_BIAS_RESID) {
          const long rc = row < p.M ? row : p.M - 1;
          extra[it] = LD_RESID((const f32x4*)((p.resid ? p.resid : (const float*)p.out) + rc * p.ldc + n0));
        }
      }
#pragma unroll
      for (int mi = 0; mi < PMI; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
          const int m = mi * 16 + c16;
          *(f32x4*)(st + m * 256 + (((ni * 4 + q4) ^ (m & 15)) << 4)) = acc[half * PMI + mi][ni];
        }
      ES(const unsigned long long ef1 = estamp(); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); const unsigned long long ef2 = estamp();)
#pragma unroll
      for (int it = 0; it < PIT; ++it) {
        const int m = it * 4 + rr;
        f32x4 v = *(const f32x4*)(st + m * 256 + ((cc ^ (m & 15)) << 4));
        if (EPI == EPI_BIAS_RESID) {
          v = extra[it] + (v + bv);
        } else if (EPI == EPI_ACT_F32) {
          v = v + bv;
          if (p.act == 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = leaky(v[j]);
          } else if (p.act == 2) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
          }
        } else if (EPI == EPI_PATCH) {
          v = v + extra[it];
        }
        const bool live = m_base + half * (4 * PIT) + m < p.M;
        if (live) ST_OUT((f32x4*)((float*)p.out + orow[it] * p.ldc + n0), v);
        if (EPI == EPI_BIAS_RESID && p.out16) {
          // LayerNorm folding: the next product reads the new residual rows in 16 bits (of THOSE rounded
          // values the statistics are taken, so that mean and variance match what the MFMAs will see)
          vec4 c16;
          float ps = 0.f, pq = 0.f;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            c16[j] = from_float<T>(v[j]);
            const float r = to_float<T>(c16[j]);
            ps += r;
            pq = fmaf(r, r, pq);
          }
          if (live) *(vec4*)((T*)p.out16 + orow[it] * (long)p.N + n0) = c16;
          ps = row16_sum(ps);   // the 16 lanes of a row hold this wave's 64 columns of it
          pq = row16_sum(pq);
          if (live && cc == 0) {
            const f32x2 st2 = {ps, pq};
            *(f32x2*)(p.stats_out + (orow[it] * (p.N >> 6) + (tn * NWC + wc)) * 2) = st2;
          }
        }
      }
      ES(if (COMPACT && wave == 0 && lane == 0) { atomicAdd(&g_estamp[3], ef1 - ef0); atomicAdd(&g_estamp[4], ef2 - ef1); atomicAdd(&g_estamp[5], estamp() - ef2); atomicAdd(&g_estamp[6], 1ull); })
    }
  }
}
#undef ES

#ifdef AACLIP_MEASURE   // lock-step predecessors of the staggered kernels, kept for A/B runs (measurement library)
template <typename T, int EPI>
__global__ __launch_bounds__(512, 2) void gemm16_256t_kernel(GemmParams p, int PN, int patches_n, int total_patches) {
  typedef typename Elem<T>::vec8 vec8;
  __shared__ __attribute__((aligned(16))) char smem[131072];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c16 = lane & 15, q4 = lane >> 4;
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
  // DMA: half-operand `sub` of the M side = LDS row groups {0..7,16..23}+8*sub (rows of the
  // waves' a-sub), of the N side = groups {0..3,8..11,16..19,24..27}+4*sub; 2 x 1 KiB per wave.
  int srcA[2][2], srcW[2][2], dstA[2][2], dstW[2][2];
#pragma unroll
  for (int sub = 0; sub < 2; ++sub)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int idx = wave * 2 + j;
      const int ga = (idx & 7) + (idx >> 3) * 16 + sub * 8;
      const int gw = (idx & 3) + (idx >> 2) * 8 + sub * 4;
      int row, chunk;
      tile_src_id(ga * 64 + lane, row, chunk);
      int ar = tm * 256 + row;
      ar = ar < p.M ? ar : p.M - 1;
      srcA[sub][j] = (ar - tm * 256) * (int)p.lda + chunk * 8;
      dstA[sub][j] = ga * 1024;
      tile_src_id(gw * 64 + lane, row, chunk);
      srcW[sub][j] = row * p.K + chunk * 8;
      dstW[sub][j] = 32768 + gw * 1024;
    }
  const T* baseA = (const T*)p.A + (long)tm * 256 * p.lda;
  const T* baseW = (const T*)p.W + (long)tn * 256 * p.K;
  // fragment reads: row = base + 16*t + (lane&15), chunk = 4*ks + (lane>>4).  Tiles 32 rows
  // apart differ by 4096 bytes; the odd 16-row tile flips the 128-byte half, so it has its own base.
  int offM[2][2], offN[2][2];   // [ks][tile parity]
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int par = 0; par < 2; ++par) {
      offM[ks][par] = tile_off_id(wr * 128 + par * 16 + c16, 4 * ks + q4);
      offN[ks][par] = 32768 + tile_off_id(wc * 64 + par * 16 + c16, 4 * ks + q4);
    }

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;

  const int nk = p.K >> 6;
#define G1(base, src, dst, st, kt) glds16(base + src + (kt) * 64, smem + (st) * 65536 + dst);
#define GA(sub, st, kt) { G1(baseA, srcA[sub][0], dstA[sub][0], st, kt) G1(baseA, srcA[sub][1], dstA[sub][1], st, kt) }
#define GW(sub, st, kt) { G1(baseW, srcW[sub][0], dstW[sub][0], st, kt) G1(baseW, srcW[sub][1], dstW[sub][1], st, kt) }
#define WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define BAR __builtin_amdgcn_s_barrier();
#define PINB __builtin_amdgcn_sched_barrier(0);
// M-side fragments of a-sub `a`: 16-row tiles 4a..4a+3 -> fm[t][ks]; N-side of b-sub `b`: tiles 2b, 2b+1 -> fn[t][ks]
#define LD_M(sb, a)                                                                         \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int t = 0; t < 4; ++t) \
      fm[t][ks] = *(const vec8*)((sb) + offM[ks][t & 1] + ((a) * 2 + (t >> 1)) * 4096);
#define LD_N(sb, b)                                                                         \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int t = 0; t < 2; ++t) \
      fn[t][ks] = *(const vec8*)((sb) + offN[ks][t] + (b) * 4096);
#define MM(a, b, t, u, ks) acc[4 * (a) + (t)][2 * (b) + (u)] = Mma16<T>::mma(fn[u][ks], fm[t][ks], acc[4 * (a) + (t)][2 * (b) + (u)]);
// 16 MFMAs of one quadrant, the phase's two DMA instructions after the 4th and the 10th
#define QUAD(a, b, I0, I1)                                                   \
  {                                                                          \
    __builtin_amdgcn_s_setprio(1);                                           \
    MM(a, b, 0, 0, 0) MM(a, b, 0, 1, 0) MM(a, b, 1, 0, 0) MM(a, b, 1, 1, 0)  \
    PINB I0 PINB                                                             \
    MM(a, b, 2, 0, 0) MM(a, b, 2, 1, 0) MM(a, b, 3, 0, 0) MM(a, b, 3, 1, 0)  \
    MM(a, b, 0, 0, 1) MM(a, b, 0, 1, 1)                                      \
    PINB I1 PINB                                                             \
    MM(a, b, 1, 0, 1) MM(a, b, 1, 1, 1) MM(a, b, 2, 0, 1) MM(a, b, 2, 1, 1)  \
    MM(a, b, 3, 0, 1) MM(a, b, 3, 1, 1)                                      \
    __builtin_amdgcn_s_setprio(0);                                           \
  }

  GA(0, 0, 0) GW(0, 0, 0) GW(1, 0, 0) GA(1, 0, 0)   // prologue: tile 0 in consumption order

  vec8 fm[4][2], fn[2][2];
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1, nxt = cur ^ 1;
    const bool more = kt + 1 < nk;
    const char* sb = smem + cur * 65536;
    // P0 (A0,B0): needs A0,B0(kt); younger in flight: B1, A1
    WAIT_VM(4);
    BAR
    LD_N(sb, 0)
    LD_M(sb, 0)
    QUAD(0, 0, if (more) G1(baseA, srcA[0][0], dstA[0][0], nxt, kt + 1), if (more) G1(baseA, srcA[0][1], dstA[0][1], nxt, kt + 1))
    // P1 (A0,B1): needs B1(kt); younger: A1(kt) [+ A0(kt+1)]
    if (more) WAIT_VM(4); else WAIT_VM(2);
    BAR
    LD_N(sb, 1)
    QUAD(0, 1, if (more) G1(baseW, srcW[0][0], dstW[0][0], nxt, kt + 1), if (more) G1(baseW, srcW[0][1], dstW[0][1], nxt, kt + 1))
    // P2 (A1,B1): needs A1(kt); younger: [A0(kt+1), B0(kt+1)]
    if (more) WAIT_VM(4); else WAIT_VM(0);
    BAR
    LD_M(sb, 1)
    QUAD(1, 1, if (more) G1(baseW, srcW[1][0], dstW[1][0], nxt, kt + 1), if (more) G1(baseW, srcW[1][1], dstW[1][1], nxt, kt + 1))
    // P3 (A1,B0): B0(kt) landed before P0
    LD_N(sb, 0)
    QUAD(1, 0, if (more) G1(baseA, srcA[1][0], dstA[1][0], nxt, kt + 1), if (more) G1(baseA, srcA[1][1], dstA[1][1], nxt, kt + 1))
  }
#undef G1
#undef GA
#undef GW
#undef WAIT_VM
#undef BAR
#undef PINB
#undef LD_M
#undef LD_N
#undef MM
#undef QUAD
  epilogue256t<T, EPI>(p, acc, smem, tm, tn, wave, lane);
}

// ---------------------------------------------------------------------------
// Overlapped variant: the LDS fragment reads of the NEXT quadrant are issued
// inside the current MFMA cluster, each one right after the last MFMA that
// consumes the register it overwrites (one register set per operand side).  The
// 224 KiB of LDS reads per K tile (896 LDS cycles per CU) then run under the
// matrix pipe instead of in front of it.  Consequence for the DMA ring: a
// half-operand must be confirmed landed one phase before the phase that
// multiplies it, so the issue schedule is shifted one phase earlier:
//   issue  A0(t+1)@(t-1,P3)  B0(t+1)@(t,P0)  B1(t+1)@(t,P1)  A1(t+1)@(t,P2)
//   read   A0,B0(t+1) during (t,P3)   B1(t+1) during (t+1,P0)   A1(t+1) during (t+1,P1)
// All steady-state waits are vmcnt(4): two half-operands stay in flight.
template <typename T, int EPI>
__global__ __launch_bounds__(512, 2) void gemm16_256u_kernel(GemmParams p, int PN, int patches_n, int total_patches) {
  typedef typename Elem<T>::vec8 vec8;
  __shared__ __attribute__((aligned(16))) char smem[131072];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c16 = lane & 15, q4 = lane >> 4;
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
      tile_src_id(ga * 64 + lane, row, chunk);
      int ar = tm * 256 + row;
      ar = ar < p.M ? ar : p.M - 1;
      srcA[sub][j] = (ar - tm * 256) * (int)p.lda + chunk * 8;
      dstA[sub][j] = ga * 1024;
      tile_src_id(gw * 64 + lane, row, chunk);
      srcW[sub][j] = row * p.K + chunk * 8;
      dstW[sub][j] = 32768 + gw * 1024;
    }
  const T* baseA = (const T*)p.A + (long)tm * 256 * p.lda;
  const T* baseW = (const T*)p.W + (long)tn * 256 * p.K;
  int offM[2][2], offN[2][2];   // [ks][tile parity]
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int par = 0; par < 2; ++par) {
      offM[ks][par] = tile_off_id(wr * 128 + par * 16 + c16, 4 * ks + q4);
      offN[ks][par] = 32768 + tile_off_id(wc * 64 + par * 16 + c16, 4 * ks + q4);
    }

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;

  const int nk = p.K >> 6;
#define G1(base, src, dst, st, kt) glds16(base + src + (kt) * 64, smem + (st) * 65536 + dst);
#define GA(sub, st, kt) { G1(baseA, srcA[sub][0], dstA[sub][0], st, kt) G1(baseA, srcA[sub][1], dstA[sub][1], st, kt) }
#define GW(sub, st, kt) { G1(baseW, srcW[sub][0], dstW[sub][0], st, kt) G1(baseW, srcW[sub][1], dstW[sub][1], st, kt) }
#define WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define BAR __builtin_amdgcn_s_barrier();
#define PINB __builtin_amdgcn_sched_barrier(0);
#define RM(sb, a, t, ks) fm[t][ks] = *(const vec8*)((sb) + offM[ks][(t) & 1] + ((a) * 2 + ((t) >> 1)) * 4096);
#define RN(sb, b, u, ks) fn[u][ks] = *(const vec8*)((sb) + offN[ks][u] + (b) * 4096);
#define MM(a, b, t, u, ks) acc[4 * (a) + (t)][2 * (b) + (u)] = Mma16<T>::mma(fn[u][ks], fm[t][ks], acc[4 * (a) + (t)][2 * (b) + (u)]);
// one pair of MFMAs (both n tiles of m tile t, k-step ks), then whatever rides behind it
#define SLOT(a, b, ks, t, X) MM(a, b, t, 0, ks) MM(a, b, t, 1, ks) PINB X PINB
// cluster that keeps fm and replaces fn by b-sub nb of stage sn (reads after the last use of each k-step)
#define CL_NEWN(a, b, sn, nb, D0, D1)                                                    \
  {                                                                                      \
    __builtin_amdgcn_s_setprio(1);                                                       \
    SLOT(a, b, 0, 0, ) SLOT(a, b, 0, 1, D0) SLOT(a, b, 0, 2, )                           \
    SLOT(a, b, 0, 3, RN(sn, nb, 0, 0) RN(sn, nb, 1, 0))                                  \
    SLOT(a, b, 1, 0, ) SLOT(a, b, 1, 1, D1) SLOT(a, b, 1, 2, )                           \
    SLOT(a, b, 1, 3, RN(sn, nb, 0, 1) RN(sn, nb, 1, 1))                                  \
    __builtin_amdgcn_s_setprio(0);                                                       \
  }
// cluster that keeps fn and replaces fm by a-sub na of stage sn (each fragment right after its two MFMAs)
#define CL_NEWM(a, b, sn, na, D0, D1)                                                    \
  {                                                                                      \
    __builtin_amdgcn_s_setprio(1);                                                       \
    SLOT(a, b, 0, 0, RM(sn, na, 0, 0)) SLOT(a, b, 0, 1, RM(sn, na, 1, 0) D0)             \
    SLOT(a, b, 0, 2, RM(sn, na, 2, 0)) SLOT(a, b, 0, 3, RM(sn, na, 3, 0))                \
    SLOT(a, b, 1, 0, RM(sn, na, 0, 1)) SLOT(a, b, 1, 1, RM(sn, na, 1, 1) D1)             \
    SLOT(a, b, 1, 2, RM(sn, na, 2, 1)) SLOT(a, b, 1, 3, RM(sn, na, 3, 1))                \
    __builtin_amdgcn_s_setprio(0);                                                       \
  }
// cluster that replaces both (transition to the next K tile)
#define CL_NEWMN(a, b, sn, na, nb, D0, D1)                                               \
  {                                                                                      \
    __builtin_amdgcn_s_setprio(1);                                                       \
    SLOT(a, b, 0, 0, RM(sn, na, 0, 0)) SLOT(a, b, 0, 1, RM(sn, na, 1, 0) D0)             \
    SLOT(a, b, 0, 2, RM(sn, na, 2, 0))                                                   \
    SLOT(a, b, 0, 3, RM(sn, na, 3, 0) RN(sn, nb, 0, 0) RN(sn, nb, 1, 0))                 \
    SLOT(a, b, 1, 0, RM(sn, na, 0, 1)) SLOT(a, b, 1, 1, RM(sn, na, 1, 1) D1)             \
    SLOT(a, b, 1, 2, RM(sn, na, 2, 1))                                                   \
    SLOT(a, b, 1, 3, RM(sn, na, 3, 1) RN(sn, nb, 0, 1) RN(sn, nb, 1, 1))                 \
    __builtin_amdgcn_s_setprio(0);                                                       \
  }
#define CL_LAST(a, b)                                                                    \
  {                                                                                      \
    SLOT(a, b, 0, 0, ) SLOT(a, b, 0, 1, ) SLOT(a, b, 0, 2, ) SLOT(a, b, 0, 3, )          \
    SLOT(a, b, 1, 0, ) SLOT(a, b, 1, 1, ) SLOT(a, b, 1, 2, ) SLOT(a, b, 1, 3, )          \
  }

  // prologue: tile 0 in consumption order plus A0 of tile 1; first fragments
  GA(0, 0, 0) GW(0, 0, 0) GW(1, 0, 0) GA(1, 0, 0)
  if (nk > 1) GA(0, 1, 1)
  vec8 fm[4][2], fn[2][2];
  if (nk > 1) WAIT_VM(6); else WAIT_VM(4);
  BAR
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
    for (int t = 0; t < 4; ++t) RM(smem, 0, t, ks)
#pragma unroll
    for (int u = 0; u < 2; ++u) RN(smem, 0, u, ks)
  }
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1, nxt = cur ^ 1;
    const bool more1 = kt + 1 < nk, more2 = kt + 2 < nk;
    const char* sb = smem + cur * 65536;
    const char* sn = smem + nxt * 65536;
    // P0 (A0,B0): fn <- B1(kt).  confirm B1(kt); younger: A1(kt) [, A0(kt+1)]
    if (more1) WAIT_VM(4); else WAIT_VM(2);
    BAR
    CL_NEWN(0, 0, sb, 1, if (more1) G1(baseW, srcW[0][0], dstW[0][0], nxt, kt + 1), if (more1) G1(baseW, srcW[0][1], dstW[0][1], nxt, kt + 1))
    // P1 (A0,B1): fm <- A1(kt).  confirm A1(kt); younger: [A0(kt+1), B0(kt+1)]
    if (more1) WAIT_VM(4); else WAIT_VM(0);
    BAR
    CL_NEWM(0, 1, sb, 1, if (more1) G1(baseW, srcW[1][0], dstW[1][0], nxt, kt + 1), if (more1) G1(baseW, srcW[1][1], dstW[1][1], nxt, kt + 1))
    // P2 (A1,B1): fn <- B0(kt) again
    CL_NEWN(1, 1, sb, 0, if (more1) G1(baseA, srcA[1][0], dstA[1][0], nxt, kt + 1), if (more1) G1(baseA, srcA[1][1], dstA[1][1], nxt, kt + 1))
    // P3 (A1,B0): fm,fn <- A0,B0(kt+1).  confirm them; younger: B1(kt+1), A1(kt+1)
    if (more1) {
      WAIT_VM(4);
      BAR
      CL_NEWMN(1, 0, sn, 0, 0, if (more2) G1(baseA, srcA[0][0], dstA[0][0], cur, kt + 2), if (more2) G1(baseA, srcA[0][1], dstA[0][1], cur, kt + 2))
    } else {
      CL_LAST(1, 0)
    }
  }
#undef G1
#undef GA
#undef GW
#undef WAIT_VM
#undef BAR
#undef PINB
#undef RM
#undef RN
#undef MM
#undef SLOT
#undef CL_NEWN
#undef CL_NEWM
#undef CL_NEWMN
#undef CL_LAST
  epilogue256t<T, EPI>(p, acc, smem, tm, tn, wave, lane);
}

#endif  // AACLIP_MEASURE

// ---------------------------------------------------------------------------
// Staggered variant.  Every phase is split into a LOAD segment (counted DMA wait,
// LDS fragment reads, optionally one DMA issue) and a COMPUTE segment (16 MFMAs with
// the remaining DMA issue between them), each closed by s_barrier, and waves 4-7 run
// ONE BARRIER BEHIND waves 0-3.  Waves w and w+4 share a SIMD, so each SIMD always
// has one wave in its MFMA cluster while the partner reads LDS, waits and syncs: the
// barrier wait of one group is the compute time of the other
// (MI355X_MICROARCH.md "Two waves per SIMD"; guide section 5, 8-phase template).
// The LDS latency of a LOAD segment is waited for AFTER its barrier.  Because the
// groups are one segment apart, a half-operand is confirmed landed (own vmcnt, then
// a barrier) one phase before the phase that reads it; DMA issue schedule as in the
// overlapped kernel above.  GL = number of the phase's two DMA instructions issued in
// the LOAD segment (the rest go between the MFMAs).
// Diagnostic stamps (ABL == 7 build only): per wave, cycles summed over the K loop for
// [0] load segment + its barrier wait, [1] compute segment, [2] barrier wait after compute.
#ifdef AACLIP_MEASURE
__device__ unsigned long long g_stamp[6 * 16384];
#else
__device__ unsigned long long g_stamp[6];   // never written: the stamp build (ABL == 7) is not instantiated in the product library
#endif
AACLIP_DEV unsigned long long stamp() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
  return t;
}
// ABL (timing-only ablations, wrong results): 1 no DMA waits, 2 no DMA issue in the K loop,
// 3 no LDS fragment reads in the K loop, 4 no barriers in the K loop, 5 = 2+3, 6 = 2+3+4.
// BUF: issue the DMA as buffer_load ... lds (wave-uniform base in the descriptor, 32-bit per-lane
// byte offset, K-tile advance in the scalar offset: no VALU address arithmetic per DMA).
template <typename T, int EPI, int GL, int ABL = 0, bool BUF = false>
__global__ __launch_bounds__(512, 2) void gemm16_256v_kernel(GemmParams p, int PN, int patches_n, int total_patches) {
  typedef typename Elem<T>::vec8 vec8;
  __shared__ __attribute__((aligned(16))) char smem[131072];

  const unsigned long long t_entry = (ABL == 7) ? stamp() : 0ull;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c16 = lane & 15, q4 = lane >> 4;
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
      tile_src_id(ga * 64 + lane, row, chunk);
      int ar = tm * 256 + row;
      ar = ar < p.M ? ar : p.M - 1;
      srcA[sub][j] = (ar - tm * 256) * (int)p.lda + chunk * 8;
      dstA[sub][j] = ga * 1024;
      tile_src_id(gw * 64 + lane, row, chunk);
      srcW[sub][j] = row * p.K + chunk * 8;
      dstW[sub][j] = 32768 + gw * 1024;
    }
  const T* baseA = (const T*)p.A + (long)tm * 256 * p.lda;
  const T* baseW = (const T*)p.W + (long)tn * 256 * p.K;
  int offM[2][2], offN[2][2];   // [ks][tile parity]
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int par = 0; par < 2; ++par) {
      offM[ks][par] = tile_off_id(wr * 128 + par * 16 + c16, 4 * ks + q4);
      offN[ks][par] = 32768 + tile_off_id(wc * 64 + par * 16 + c16, 4 * ks + q4);
    }

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;

  const int nk = p.K >> 6;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)baseA, 0, 0x7FFFFFF0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)baseW, 0, 0x7FFFFFF0, 0x00020000);
#define RS_baseA rsA
#define RS_baseW rsW
#define G1(base, src, dst, st, kt)                                                                       \
  {                                                                                                      \
    if (BUF)                                                                                             \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(RS_##base, (lds_void*)(smem + (st) * 65536 + dst), 16,    \
                                               (src) * 2, (kt) * 128, 0, 0);                             \
    else                                                                                                 \
      glds16(base + src + (kt) * 64, smem + (st) * 65536 + dst);                                         \
  }
#define GA(sub, st, kt) { G1(baseA, srcA[sub][0], dstA[sub][0], st, kt) G1(baseA, srcA[sub][1], dstA[sub][1], st, kt) }
#define GW(sub, st, kt) { G1(baseW, srcW[sub][0], dstW[sub][0], st, kt) G1(baseW, srcW[sub][1], dstW[sub][1], st, kt) }
#define WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define LGKM0 asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#define BAR __builtin_amdgcn_s_barrier();
#define PINB __builtin_amdgcn_sched_barrier(0);
#define LD_M(sb, a)                                                                         \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int t = 0; t < 4; ++t) \
      fm[t][ks] = *(const vec8*)((sb) + offM[ks][t & 1] + ((a) * 2 + (t >> 1)) * 4096);
#define LD_N(sb, b)                                                                         \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int t = 0; t < 2; ++t) \
      fn[t][ks] = *(const vec8*)((sb) + offN[ks][t] + (b) * 4096);
#define MM(a, b, t, u, ks) acc[4 * (a) + (t)][2 * (b) + (u)] = Mma16<T>::mma(fn[u][ks], fm[t][ks], acc[4 * (a) + (t)][2 * (b) + (u)]);
// COMPUTE segment: 16 MFMAs, DMA instructions I0 / I1 after the 4th / 10th
#define QUADV(a, b, I0, I1)                                                  \
  {                                                                          \
    LGKM0                                                                    \
    PINB                                                                     \
    __builtin_amdgcn_s_setprio(1);                                           \
    MM(a, b, 0, 0, 0) MM(a, b, 0, 1, 0) MM(a, b, 1, 0, 0) MM(a, b, 1, 1, 0)  \
    PINB I0 PINB                                                             \
    MM(a, b, 2, 0, 0) MM(a, b, 2, 1, 0) MM(a, b, 3, 0, 0) MM(a, b, 3, 1, 0)  \
    MM(a, b, 0, 0, 1) MM(a, b, 0, 1, 1)                                      \
    PINB I1 PINB                                                             \
    MM(a, b, 1, 0, 1) MM(a, b, 1, 1, 1) MM(a, b, 2, 0, 1) MM(a, b, 2, 1, 1)  \
    MM(a, b, 3, 0, 1) MM(a, b, 3, 1, 1)                                      \
    __builtin_amdgcn_s_setprio(0);                                           \
    PINB                                                                     \
  }
// the phase's two DMA instructions: the first GL of them in the LOAD segment
#define DL0(X) if (GL >= 1) { X }
#define DL1(X) if (GL >= 2) { X }
#define DC0(X) if (GL < 1) { X }
#define DC1(X) if (GL < 2) { X }

  // prologue: tile 0 in consumption order plus A0 of tile 1; A0,B0(0) confirmed by everyone
  unsigned long long t_setup = 0;
  if (ABL == 7) t_setup = stamp();
  GA(0, 0, 0) GW(0, 0, 0) GW(1, 0, 0) GA(1, 0, 0)
  if (nk > 1) GA(0, 1, 1)
  if (nk > 1) WAIT_VM(6); else WAIT_VM(4);
  BAR
  if (wr == 1) BAR   // waves 4-7 now run one segment behind waves 0-3

  vec8 fm[4][2], fn[2][2];
  constexpr bool NO_WAIT = ABL == 1, NO_DMA = ABL == 2 || ABL == 5 || ABL == 6, NO_LDS = ABL == 3 || ABL == 5 || ABL == 6,
                 NO_BAR = ABL == 4 || ABL == 6;
  constexpr bool STAMP = ABL == 7;
  unsigned long long tl = 0, tc = 0, tb = 0, t_prev = 0, t_a = 0, t_b = 0, tw = 0, tr = 0, tg = 0;
  if (STAMP) { t_prev = stamp(); tw = t_prev - t_entry; tg = t_setup - t_entry; }
  if (NO_LDS) { LD_M(smem, 0) LD_N(smem, 0) }
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1, nxt = cur ^ 1;
    const bool more1 = kt + 1 < nk && !NO_DMA, more2 = kt + 2 < nk && !NO_DMA;
    const char* sb = smem + cur * 65536;
    // ---- P0: load A0,B0(kt) fragments; confirm B1(kt) (younger: A1(kt) [, A0(kt+1)]); DMA B0(kt+1)
    unsigned long long u0 = 0, u1 = 0, u2 = 0, u3 = 0;
    if (STAMP) { PINB u0 = stamp(); PINB }
    if (!NO_WAIT) { if (more1) WAIT_VM(4); else WAIT_VM(2); }
    if (STAMP) { PINB u1 = stamp(); PINB }
    if (!NO_LDS) { LD_N(sb, 0) }
    if (!NO_LDS) { LD_M(sb, 0) }
    if (STAMP) { PINB u2 = stamp(); PINB }
    DL0(if (more1) G1(baseW, srcW[0][0], dstW[0][0], nxt, kt + 1)) DL1(if (more1) G1(baseW, srcW[0][1], dstW[0][1], nxt, kt + 1))
    if (STAMP) { PINB u3 = stamp(); PINB tw += u1 - u0; tr += u2 - u1; tg += u3 - u2; }
    if (!NO_BAR) BAR
    if (STAMP) { PINB t_a = stamp(); PINB tl += t_a - t_prev; }
    QUADV(0, 0, DC0(if (more1) G1(baseW, srcW[0][0], dstW[0][0], nxt, kt + 1)), DC1(if (more1) G1(baseW, srcW[0][1], dstW[0][1], nxt, kt + 1)))
    if (STAMP) { PINB t_b = stamp(); PINB tc += t_b - t_a; }
    if (!NO_BAR) BAR
    if (STAMP) { PINB t_prev = stamp(); PINB tb += t_prev - t_b; }
    // ---- P1: load B1(kt); confirm A1(kt) (younger: [A0(kt+1), B0(kt+1)]); DMA B1(kt+1)
    if (!NO_WAIT) { if (more1) WAIT_VM(4); else WAIT_VM(0); }
    if (!NO_LDS) { LD_N(sb, 1) }
    DL0(if (more1) G1(baseW, srcW[1][0], dstW[1][0], nxt, kt + 1)) DL1(if (more1) G1(baseW, srcW[1][1], dstW[1][1], nxt, kt + 1))
    if (!NO_BAR) BAR
    if (STAMP) { PINB t_a = stamp(); PINB tl += t_a - t_prev; }
    QUADV(0, 1, DC0(if (more1) G1(baseW, srcW[1][0], dstW[1][0], nxt, kt + 1)), DC1(if (more1) G1(baseW, srcW[1][1], dstW[1][1], nxt, kt + 1)))
    if (STAMP) { PINB t_b = stamp(); PINB tc += t_b - t_a; }
    if (!NO_BAR) BAR
    if (STAMP) { PINB t_prev = stamp(); PINB tb += t_prev - t_b; }
    // ---- P2: load A1(kt); DMA A1(kt+1)
    if (!NO_LDS) { LD_M(sb, 1) }
    DL0(if (more1) G1(baseA, srcA[1][0], dstA[1][0], nxt, kt + 1)) DL1(if (more1) G1(baseA, srcA[1][1], dstA[1][1], nxt, kt + 1))
    if (!NO_BAR) BAR
    if (STAMP) { PINB t_a = stamp(); PINB tl += t_a - t_prev; }
    QUADV(1, 1, DC0(if (more1) G1(baseA, srcA[1][0], dstA[1][0], nxt, kt + 1)), DC1(if (more1) G1(baseA, srcA[1][1], dstA[1][1], nxt, kt + 1)))
    if (STAMP) { PINB t_b = stamp(); PINB tc += t_b - t_a; }
    if (!NO_BAR) BAR
    if (STAMP) { PINB t_prev = stamp(); PINB tb += t_prev - t_b; }
    // ---- P3: re-load B0(kt); confirm A0,B0(kt+1) (younger: B1(kt+1), A1(kt+1)); DMA A0(kt+2)
    if (!NO_WAIT) { if (more1) WAIT_VM(4); }
    if (!NO_LDS) { LD_N(sb, 0) }
    DL0(if (more2) G1(baseA, srcA[0][0], dstA[0][0], cur, kt + 2)) DL1(if (more2) G1(baseA, srcA[0][1], dstA[0][1], cur, kt + 2))
    if (!NO_BAR) BAR
    if (STAMP) { PINB t_a = stamp(); PINB tl += t_a - t_prev; }
    QUADV(1, 0, DC0(if (more2) G1(baseA, srcA[0][0], dstA[0][0], cur, kt + 2)), DC1(if (more2) G1(baseA, srcA[0][1], dstA[0][1], cur, kt + 2)))
    if (STAMP) { PINB t_b = stamp(); PINB tc += t_b - t_a; }
    if (!NO_BAR) BAR
    if (STAMP) { PINB t_prev = stamp(); PINB tb += t_prev - t_b; }
  }
  if (wr == 0) BAR   // balance the barrier count of the two groups
  unsigned long long t_loop_end = 0;
  if (ABL == 7) {   // stamp build: the epilogue sits between its stamps (every other build runs it once, below)
    t_loop_end = stamp();
    epilogue256t<T, EPI>(p, acc, smem, tm, tn, wave, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    tr = stamp() - t_loop_end;
  }
  if (ABL == 7 && lane == 0) {
    const int w = (blockIdx.x * 8 + wave) & 16383;
    g_stamp[6 * w + 0] = tl; g_stamp[6 * w + 1] = tc; g_stamp[6 * w + 2] = tb;
    g_stamp[6 * w + 3] = tw; g_stamp[6 * w + 4] = tr; g_stamp[6 * w + 5] = tg;
  }
#undef G1
#undef RS_baseA
#undef RS_baseW
#undef GA
#undef GW
#undef WAIT_VM
#undef LGKM0
#undef BAR
#undef PINB
#undef LD_M
#undef LD_N
#undef MM
#undef QUADV
#undef DL0
#undef DL1
#undef DC0
#undef DC1
  if (ABL == 8) {   // timing ablation: no epilogue stores (one dummy store keeps the accumulators live)
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) sum += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (sum == 123.456f) ((float*)p.out)[0] = sum;
  } else if (ABL != 7) {
    epilogue256t<T, EPI>(p, acc, smem, tm, tn, wave, lane);
  }
}

// ---------------------------------------------------------------------------
#ifdef AACLIP_MEASURE
// Staggered + overlapped: the LOAD segment only waits for DMA and issues the two
// DMA instructions of the phase; the LDS fragment reads of the next quadrant ride
// inside the MFMA cluster (in-place, as in gemm16_256u_kernel).  With waves 4-7 one
// barrier behind, a half-operand read during compute segment C(p) must have been
// confirmed in L(p-1) by both groups:
//   issue    A0(t+1)@L(t-1,P3)  B0(t+1)@L(t,P0)  B1(t+1)@L(t,P1)  A1(t+1)@L(t,P2)
//   confirm  A1(t)@L(t,P0)      A0,B0(t+1)@L(t,P2)               B1(t+1)@L(t,P3)
//   read     B1(t) in C(t,P0)   A1(t) in C(t,P1)   B0(t) in C(t,P2)   A0,B0(t+1) in C(t,P3)
// every wait is vmcnt(2): one younger half-operand stays in flight behind it.
template <typename T, int EPI>
__global__ __launch_bounds__(512, 2) void gemm16_256w_kernel(GemmParams p, int PN, int patches_n, int total_patches) {
  typedef typename Elem<T>::vec8 vec8;
  __shared__ __attribute__((aligned(16))) char smem[131072];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c16 = lane & 15, q4 = lane >> 4;
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
      tile_src_id(ga * 64 + lane, row, chunk);
      int ar = tm * 256 + row;
      ar = ar < p.M ? ar : p.M - 1;
      srcA[sub][j] = (ar - tm * 256) * (int)p.lda + chunk * 8;
      dstA[sub][j] = ga * 1024;
      tile_src_id(gw * 64 + lane, row, chunk);
      srcW[sub][j] = row * p.K + chunk * 8;
      dstW[sub][j] = 32768 + gw * 1024;
    }
  const T* baseA = (const T*)p.A + (long)tm * 256 * p.lda;
  const T* baseW = (const T*)p.W + (long)tn * 256 * p.K;
  int offM[2][2], offN[2][2];   // [ks][tile parity]
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int par = 0; par < 2; ++par) {
      offM[ks][par] = tile_off_id(wr * 128 + par * 16 + c16, 4 * ks + q4);
      offN[ks][par] = 32768 + tile_off_id(wc * 64 + par * 16 + c16, 4 * ks + q4);
    }

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;

  const int nk = p.K >> 6;
#define G1(base, src, dst, st, kt) glds16(base + src + (kt) * 64, smem + (st) * 65536 + dst);
#define GA(sub, st, kt) { G1(baseA, srcA[sub][0], dstA[sub][0], st, kt) G1(baseA, srcA[sub][1], dstA[sub][1], st, kt) }
#define GW(sub, st, kt) { G1(baseW, srcW[sub][0], dstW[sub][0], st, kt) G1(baseW, srcW[sub][1], dstW[sub][1], st, kt) }
#define WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define BAR __builtin_amdgcn_s_barrier();
#define PINB __builtin_amdgcn_sched_barrier(0);
#define RM(sb, a, t, ks) fm[t][ks] = *(const vec8*)((sb) + offM[ks][(t) & 1] + ((a) * 2 + ((t) >> 1)) * 4096);
#define RN(sb, b, u, ks) fn[u][ks] = *(const vec8*)((sb) + offN[ks][u] + (b) * 4096);
#define MM(a, b, t, u, ks) acc[4 * (a) + (t)][2 * (b) + (u)] = Mma16<T>::mma(fn[u][ks], fm[t][ks], acc[4 * (a) + (t)][2 * (b) + (u)]);
#define SLOT(a, b, ks, t, X) MM(a, b, t, 0, ks) MM(a, b, t, 1, ks) PINB X PINB
#define CW_NEWN(a, b, sn, nb)                                                            \
  {                                                                                      \
    __builtin_amdgcn_s_setprio(1);                                                       \
    SLOT(a, b, 0, 0, ) SLOT(a, b, 0, 1, ) SLOT(a, b, 0, 2, )                             \
    SLOT(a, b, 0, 3, RN(sn, nb, 0, 0) RN(sn, nb, 1, 0))                                  \
    SLOT(a, b, 1, 0, ) SLOT(a, b, 1, 1, ) SLOT(a, b, 1, 2, )                             \
    SLOT(a, b, 1, 3, RN(sn, nb, 0, 1) RN(sn, nb, 1, 1))                                  \
    __builtin_amdgcn_s_setprio(0);                                                       \
  }
#define CW_NEWM(a, b, sn, na)                                                            \
  {                                                                                      \
    __builtin_amdgcn_s_setprio(1);                                                       \
    SLOT(a, b, 0, 0, RM(sn, na, 0, 0)) SLOT(a, b, 0, 1, RM(sn, na, 1, 0))                \
    SLOT(a, b, 0, 2, RM(sn, na, 2, 0)) SLOT(a, b, 0, 3, RM(sn, na, 3, 0))                \
    SLOT(a, b, 1, 0, RM(sn, na, 0, 1)) SLOT(a, b, 1, 1, RM(sn, na, 1, 1))                \
    SLOT(a, b, 1, 2, RM(sn, na, 2, 1)) SLOT(a, b, 1, 3, RM(sn, na, 3, 1))                \
    __builtin_amdgcn_s_setprio(0);                                                       \
  }
#define CW_NEWMN(a, b, sn, na, nb)                                                       \
  {                                                                                      \
    __builtin_amdgcn_s_setprio(1);                                                       \
    SLOT(a, b, 0, 0, RM(sn, na, 0, 0)) SLOT(a, b, 0, 1, RM(sn, na, 1, 0))                \
    SLOT(a, b, 0, 2, RM(sn, na, 2, 0))                                                   \
    SLOT(a, b, 0, 3, RM(sn, na, 3, 0) RN(sn, nb, 0, 0) RN(sn, nb, 1, 0))                 \
    SLOT(a, b, 1, 0, RM(sn, na, 0, 1)) SLOT(a, b, 1, 1, RM(sn, na, 1, 1))                \
    SLOT(a, b, 1, 2, RM(sn, na, 2, 1))                                                   \
    SLOT(a, b, 1, 3, RM(sn, na, 3, 1) RN(sn, nb, 0, 1) RN(sn, nb, 1, 1))                 \
    __builtin_amdgcn_s_setprio(0);                                                       \
  }
#define CW_LAST(a, b)                                                                    \
  {                                                                                      \
    SLOT(a, b, 0, 0, ) SLOT(a, b, 0, 1, ) SLOT(a, b, 0, 2, ) SLOT(a, b, 0, 3, )          \
    SLOT(a, b, 1, 0, ) SLOT(a, b, 1, 1, ) SLOT(a, b, 1, 2, ) SLOT(a, b, 1, 3, )          \
  }

  // prologue: tile 0 in consumption order plus A0 of tile 1; A0,B0,B1(0) confirmed by everyone
  GA(0, 0, 0) GW(0, 0, 0) GW(1, 0, 0) GA(1, 0, 0)
  if (nk > 1) GA(0, 1, 1)
  vec8 fm[4][2], fn[2][2];
  if (nk > 1) WAIT_VM(4); else WAIT_VM(2);
  BAR
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
    for (int t = 0; t < 4; ++t) RM(smem, 0, t, ks)
#pragma unroll
    for (int u = 0; u < 2; ++u) RN(smem, 0, u, ks)
  }
  if (wr == 1) BAR   // waves 4-7 now run one segment behind waves 0-3

  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1, nxt = cur ^ 1;
    const bool more1 = kt + 1 < nk, more2 = kt + 2 < nk;
    const char* sb = smem + cur * 65536;
    const char* sn = smem + nxt * 65536;
    // ---- P0: confirm A1(kt) (younger: A0(kt+1)); issue B0(kt+1); compute (A0,B0) while reading B1(kt)
    if (more1) WAIT_VM(2); else WAIT_VM(0);
    if (more1) GW(0, nxt, kt + 1)
    BAR
    CW_NEWN(0, 0, sb, 1)
    BAR
    // ---- P1: issue B1(kt+1); compute (A0,B1) while reading A1(kt)
    if (more1) GW(1, nxt, kt + 1)
    BAR
    CW_NEWM(0, 1, sb, 1)
    BAR
    // ---- P2: confirm A0,B0(kt+1) (younger: B1(kt+1)); issue A1(kt+1); compute (A1,B1) while re-reading B0(kt)
    if (more1) { WAIT_VM(2); GA(1, nxt, kt + 1) }
    BAR
    CW_NEWN(1, 1, sb, 0)
    BAR
    // ---- P3: confirm B1(kt+1) (younger: A1(kt+1)); issue A0(kt+2); compute (A1,B0) while reading A0,B0(kt+1)
    if (more1) WAIT_VM(2);
    if (more2) GA(0, cur, kt + 2)
    BAR
    if (more1) CW_NEWMN(1, 0, sn, 0, 0) else CW_LAST(1, 0)
    BAR
  }
  if (wr == 0) BAR   // balance the barrier count of the two groups
#undef G1
#undef GA
#undef GW
#undef WAIT_VM
#undef BAR
#undef PINB
#undef RM
#undef RN
#undef MM
#undef SLOT
#undef CW_NEWN
#undef CW_NEWM
#undef CW_NEWMN
#undef CW_LAST
  epilogue256t<T, EPI>(p, acc, smem, tm, tn, wave, lane);
}

// ---------------------------------------------------------------------------
#endif  // AACLIP_MEASURE

// Staggered kernel with two N-side fragment sets (the default).  Tried on top of it and dropped
// (no gain): a persistent workgroup walking the tile list (static order and per-XCD
// atomic queues with stealing), start-time de-phasing of the first round, smaller XCD patches.  B0 stays in registers for the
// whole K tile (no re-read in phase 3) and the next tile's B0 is read one phase
// early into the set that B1 has just vacated, so a LOAD segment reads at most 8
// fragments (12 before) and a K tile 24 (28 before).  DMA by buffer_load ... lds,
// both in the LOAD segment.  Schedule (reads in L(p) must be confirmed in L(p-1)
// by both wave groups):
//   issue    A0(t+1)@L(t,P0)  B1(t+1)@L(t,P1)  A1(t+1)@L(t,P2)  B0(t+2)@L(t,P3)
//   confirm  B1(t)@L(t,P0)    A1(t)@L(t,P1)    B0(t+1)@L(t,P2)  A0(t+1)@L(t,P3)
//   read     A0(t)@L(t,P0)    B1(t)@L(t,P1)    A1(t)@L(t,P2)    B0(t+1)@L(t,P3)
// NP = 0: plain 16-bit operands; NP = 4 / 3: split8 operands (common.h): per pair of K tiles two fp16 tiles and two
// (one: weight exact in fp16) e4m3 correction tiles, all through the same phases; a phase of an e4m3 tile issues 8
// 16x16x128 block-scaled MFMAs (32 cycles each) where an fp16 tile issues 16 16x16x32 ones (16 cycles each)
// The two 16-byte fragments (k-steps) of one 16-row tile.  Plain operands: two independent 4-register values, as the
// kernel always had them.  Split8 operands: ONE 8-register value, because the block-scaled e4m3 MFMA takes both
// fragments as a single 32-byte operand -- kept separate, hipcc copies them into fresh 8-register tuples for every MFMA
// and spills (61-201 registers); the fp16 MFMAs of the same kernel read the halves as sub-registers.
template <typename T, bool WIDE> struct FragPair {
  typename Elem<T>::vec8 h[2];
  AACLIP_DEV typename Elem<T>::vec8 get(int ks) const { return h[ks]; }
  AACLIP_DEV void set(int ks, typename Elem<T>::vec8 v) { h[ks] = v; }
};
template <> struct FragPair<f16, true> {
  i32x8 v;
  AACLIP_DEV f16x8 get(int ks) const {
    const i32x4 x = {v[4 * ks], v[4 * ks + 1], v[4 * ks + 2], v[4 * ks + 3]};
    return __builtin_bit_cast(f16x8, x);
  }
  AACLIP_DEV void set(int ks, f16x8 f) {
    const i32x4 x = __builtin_bit_cast(i32x4, f);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[4 * ks + e] = x[e];
  }
};
// The block-scaled e4m3 MFMA of a correction tile (KIND 1: Al8 . Wh8, KIND 2: Ah8 . Wl8), from inline asm with the
// accumulator tied in place: through the builtin hipcc gives many of these MFMAs a destination tuple different from
// their C operand (copies back, spills).  No result of a correction tile is read before the next s_barrier, so no wait
// states are needed here.  Both scale bytes come from ONE register (the kernel is at its register limit): `sc` holds
// [byte 0: T1 act, byte 1: T1 weight, byte 2: T2 weight, byte 3: T2 act]; op_sel / op_sel_hi pick the byte per operand.
template <int KIND>
AACLIP_DEV f32x4 mma_e4m3k(const FragPair<f16, true>& w, const FragPair<f16, true>& a, f32x4 c, int sc) {
  if (KIND == 1)
    asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel:[1,0,0] op_sel_hi:[0,0,0]"
                 : "+v"(c) : "v"(w.v), "v"(a.v), "v"(sc));
  else
    asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel:[0,1,0] op_sel_hi:[1,1,0]"
                 : "+v"(c) : "v"(w.v), "v"(a.v), "v"(sc));
  return c;
}
template <int KIND, typename T> AACLIP_DEV f32x4 mma_e4m3k(const FragPair<T, false>&, const FragPair<T, false>&, f32x4 c, int) { return c; }

// X_PRIO_MODE (build-time switch): 0 = s_setprio 1 around every MFMA cluster (rounds 1-3), 1 = static priority for waves
// 4-7 and no per-segment flips (MI355X_MICROARCH.md, "Two waves per SIMD", item 4), 2 = no priorities at all.  Measured in
// round 4 on one box, three builds side by side: split products -1.1 ... -1.7 % with 1 or 2, tower 516 -> 520.5 images/s
// with either, plain fp16 within noise.  Default 2: the per-segment flips were costing what the guide says they cost.
#ifndef X_PRIO_MODE
#define X_PRIO_MODE 2
#endif
// X_LOAD_ORDER (build-time experiment switch) of a LOAD segment: 0 = counted wait, fragment reads, DMA issue; 1 = DMA issue
// before the reads; 2 = the counted wait LAST (two more instructions outstanding: the same piece confirmed a segment later)
#ifndef X_LOAD_ORDER
#define X_LOAD_ORDER 0
#endif
// WALK (round 4, split operands only): ONE workgroup per CU walks the tiles b, b + gridDim.x, ... of the same virtual grid
// (same XCD: gridDim.x is a multiple of 8), and issues the first K tile of its NEXT tile into the free operand stage before
// the epilogue of the current one, which then stages its rows in the other stage alone (epilogue256t<..., COMPACT>): the
// prologue of a tile (argument loads, first DMA round trip, 32-64 KiB of LDS fill: ~5 us with the matrix pipe idle) runs
// under the previous tile's stores.  `stagger` then carries the size of the virtual grid.
template <typename T, int EPI, int NP = 0, bool QK8 = false, bool WALK = false>   // QK8: GemmParams::out_qk8 (EPI_BIAS, split operands only)
__global__ __launch_bounds__(512, 2) void gemm16_256x_kernel(GemmParams p, int PN, int patches_n, int total_patches, int PM, int stagger) {
  static_assert(!WALK || NP != 0, "the walking form exists for split operands (tile-independent per-lane DMA offsets)");
  typedef typename Elem<T>::vec8 vec8;
  __shared__ __attribute__((aligned(16))) char smem[131072];

  if (NP != 0) fp8_saturate_mode();   // the split8 epilogues convert with split8x4_sat
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c16 = lane & 15, q4 = lane >> 4;
  const int wr = wave >> 2, wc = wave & 3;
  const int tiles_m = (p.M + 255) >> 8;
  auto map_tile = [&](int b, int& tm_, int& tn_) -> bool {
    const int P = PM * PN;
    const int xcd = b & 7, j = b >> 3;
    const int gp = (j / P) * 8 + xcd, local = j % P;
    if (gp >= total_patches) return false;
    const int pm = gp / patches_n, pn = gp - pm * patches_n;
    tm_ = pm * PM + local / PN;
    tn_ = pn * PN + local % PN;
    return tm_ < tiles_m;
  };
  int tm = 0, tn = 0;
  int vb = blockIdx.x;
  if constexpr (WALK) {
    while (vb < stagger && !map_tile(vb, tm, tn)) vb += gridDim.x;
    if (vb >= stagger) return;   // whole workgroup, before any barrier
  } else {
    if (!map_tile(vb, tm, tn)) return;
  }
  if (!WALK && stagger > 1 && blockIdx.x < 256) {
    // De-phase the first round: workgroup slot k of an XCD starts k/stagger of a tile later, and the
    // dispatcher keeps the CUs de-phased afterwards.  Otherwise every CU reaches its epilogue at the
    // same moment and the residual read-modify-write (67 + 67 MB per round) is an HBM burst with no
    // MFMA running beside it.
    const int slot = (blockIdx.x >> 3) % stagger;
    const long long target = (long long)slot * (p.K >> 6) * 2800 / stagger;
    const long long t0 = __builtin_amdgcn_s_memtime();
    while ((long long)__builtin_amdgcn_s_memtime() - t0 < target) __builtin_amdgcn_s_sleep(64);
  }
  const int ldw = NP == 4 ? 2 * p.K : (NP == 3 ? p.K + (p.K >> 1) : p.K);   // W row stride in halves (4K / 3K / 2K bytes)
  int srcA[2][2], srcW[2][2], dstA[2][2], dstW[2][2];   // byte offsets
#pragma unroll
  for (int sub = 0; sub < 2; ++sub)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int idx = wave * 2 + j;
      const int ga = (idx & 7) + (idx >> 3) * 16 + sub * 8;
      const int gw = (idx & 3) + (idx >> 2) * 8 + sub * 4;
      int row, chunk;
      tile_src_id(ga * 64 + lane, row, chunk);
      int ar = tm * 256 + row;
      // plain operands: rows past M re-read row M - 1.  Split operands: no clamp -- the buffer descriptor ends with the
      // matrix and reads past it return zero (see rsA), so the second half's offsets are the first half's plus a
      // wave-uniform constant and need no registers (this kernel has none to spare: a spilled address is reloaded
      // with scratch_load + s_waitcnt vmcnt(0) in front of a DMA, which drains the whole DMA pipeline once per tile)
      if (NP == 0) ar = ar < p.M ? ar : p.M - 1;
      srcA[sub][j] = ((ar - tm * 256) * (int)p.lda + chunk * 8) * 2;
      dstA[sub][j] = ga * 1024;
      tile_src_id(gw * 64 + lane, row, chunk);
      srcW[sub][j] = (row * ldw + chunk * 8) * 2;
      dstW[sub][j] = 32768 + gw * 1024;
    }
  // the descriptors must be PROVABLY wave-uniform, or hipcc wraps every buffer_load ... lds of the K loop in a
  // waterfall loop (v_readfirstlane x4, compare, s_and_saveexec): it lost the proof when the folding epilogue was
  // added and the residual GEMMs ran 15 % slower with an unchanged K loop in source
  auto desc_a = [&](int tm_) {
    const T* baseA = uniform_ptr((const T*)p.A + (long)tm_ * 256 * p.lda);
    // split operands: num_records = the bytes from this tile's first row to the end of A (out-of-range rows read as zero)
    const long bytesA = ((long)p.M - (long)tm_ * 256) * p.lda * 2;
    const int recA = NP == 0 ? 0x7FFFFFF0 : (int)(bytesA < 0x7FFFFFF0L ? bytesA : 0x7FFFFFF0L);
    return __builtin_amdgcn_make_buffer_rsrc((void*)baseA, 0, __builtin_amdgcn_readfirstlane(recA), 0x00020000);
  };
  auto desc_w = [&](int tn_) {
    const T* baseW = uniform_ptr((const T*)p.W + (long)tn_ * 256 * ldw);
    return __builtin_amdgcn_make_buffer_rsrc((void*)baseW, 0, 0x7FFFFFF0, 0x00020000);
  };
  __amdgpu_buffer_rsrc_t rsA = desc_a(tm), rsW = desc_w(tn);
  const int subA = 64 * (int)p.lda * 2, subW = 32 * ldw * 2;   // bytes from a DMA piece of half 0 to the same piece of half 1
  int offM[2][2], offN[2][2];   // [ks][tile parity]
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int par = 0; par < 2; ++par) {
      offM[ks][par] = tile_off_id(wr * 128 + par * 16 + c16, 4 * ks + q4);
      offN[ks][par] = 32768 + tile_off_id(wc * 64 + par * 16 + c16, 4 * ks + q4);
    }

  f32x4 acc[8][4];   // zeroed at the head of every tile (below)
  // LayerNorm folding: this lane's (rstd, -mean*rstd) pairs, requested now so that they are there at the epilogue
  f32x2 ab_pre[8];
  const bool fold_pre = NP == 0 && (EPI == EPI_BIAS || EPI == EPI_BIAS_GELU) && p.row_ab != nullptr;
  if (fold_pre) {
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) {
      int row = tm * 256 + wr * 128 + mi * 16 + c16;
      row = row < p.M ? row : p.M - 1;
      ab_pre[mi] = *(const f32x2*)(p.row_ab + 2L * row);
    }
  }

  const int nk = vtile_count<NP>(p.K);   // (virtual) K tiles
  // e8m0 scale bytes of the correction tiles (stored operand = value * 2^EXP): T1 act | T1 weight | T2 weight | T2 act
  int sc_pack = (127 - SPLIT8_ACT_LO_EXP) | ((127 - SPLIT8_W_HI_EXP) << 8) | ((127 - SPLIT8_W_LO_EXP) << 16) |
                ((127 - SPLIT8_ACT_HI_EXP) << 24);
  (void)sc_pack;
#define DMA(rs, src, dst, st, so) \
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(smem + (st) * 65536 + dst), 16, src, so, 0, 0);
#define GAx(rs, sub, st, kt) { int kd; const int so = vtile_off<NP>(kt, p.K, kd) + (NP != 0 ? (sub) * subA : 0); \
    DMA(rs, srcA[NP != 0 ? 0 : (sub)][0], dstA[sub][0], st, so) DMA(rs, srcA[NP != 0 ? 0 : (sub)][1], dstA[sub][1], st, so) }
#define GWx(rs, sub, st, kt) { int kd; const int so = vtile_off<NP>(kt, p.K, kd) + (NP != 0 ? (sub) * subW : 0); \
    DMA(rs, srcW[NP != 0 ? 0 : (sub)][0], dstW[sub][0], st, so) DMA(rs, srcW[NP != 0 ? 0 : (sub)][1], dstW[sub][1], st, so) }
#define GA(sub, st, kt) GAx(rsA, sub, st, kt)
#define GW(sub, st, kt) GWx(rsW, sub, st, kt)
#define WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define LGKM0 asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#define BAR __builtin_amdgcn_s_barrier();
#define PINB __builtin_amdgcn_sched_barrier(0);
// Split kernels (no register to spare, see srcA): the offset of the odd 16-row tile is recomputed from the even one
// at every use -- 16 rows further = 8 row pairs = 2048 bytes, and the swizzle's bit 3 flips: (off ^ 128) + 2048.  The
// xor goes through an opaque asm so that hipcc does not hoist it back into a loop-invariant register.
#define ODD_OFF(off) ({ int o_; asm volatile("v_xor_b32 %0, 0x80, %1" : "=v"(o_) : "v"(off)); o_ + 2048; })
#define LD_M(sb, a)                                                                         \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int t = 0; t < 4; ++t) \
      fm[t].set(ks, *(const vec8*)((sb) + ((NP != 0 && (t & 1)) ? ODD_OFF(offM[ks][0]) : offM[ks][t & 1]) + ((a) * 2 + (t >> 1)) * 4096));
#define LD_N(FN, sb, b)                                                                     \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int t = 0; t < 2; ++t) \
      FN[t].set(ks, *(const vec8*)((sb) + ((NP != 0 && t == 1) ? ODD_OFF(offN[ks][0]) : offN[ks][t]) + (b) * 4096));
#define MM(FN, a, b, t, u, ks) acc[4 * (a) + (t)][2 * (b) + (u)] = Mma16<T>::mma(FN[u].get(ks), fm[t].get(ks), acc[4 * (a) + (t)][2 * (b) + (u)]);
// e4m3 tile: the two 16-byte fragments of a row are one 32-byte operand (common.h, mma_e4m3); W rows are the MFMA's A side
#define MM8(FN, a, b, t, u, KIND) acc[4 * (a) + (t)][2 * (b) + (u)] = mma_e4m3k<(KIND) == 1 ? 1 : 2>(FN[u], fm[t], acc[4 * (a) + (t)][2 * (b) + (u)], sc_pack);
// KIND is a literal (0 fp16 tile, 1 / 2 e4m3 correction tiles): the K loop is unrolled over one period of the tile
// kinds, so no branch surrounds the MFMAs -- with a run-time branch the accumulators of the two arms meet in phi
// nodes hipcc does not coalesce (results in fresh registers + 472 v_mov + 200 spilled registers, 4x slower)
#define QUADX(FN, a, b, KIND)                                                                \
  {                                                                                          \
    LGKM0                                                                                    \
    PINB                                                                                     \
    if (X_PRIO_MODE == 0) __builtin_amdgcn_s_setprio(1);                                     \
    if ((KIND) != 0) {                                                                       \
      MM8(FN, a, b, 0, 0, KIND) MM8(FN, a, b, 0, 1, KIND) MM8(FN, a, b, 1, 0, KIND) MM8(FN, a, b, 1, 1, KIND) \
      MM8(FN, a, b, 2, 0, KIND) MM8(FN, a, b, 2, 1, KIND) MM8(FN, a, b, 3, 0, KIND) MM8(FN, a, b, 3, 1, KIND) \
    } else {                                                                                 \
    MM(FN, a, b, 0, 0, 0) MM(FN, a, b, 0, 1, 0) MM(FN, a, b, 1, 0, 0) MM(FN, a, b, 1, 1, 0)  \
    MM(FN, a, b, 2, 0, 0) MM(FN, a, b, 2, 1, 0) MM(FN, a, b, 3, 0, 0) MM(FN, a, b, 3, 1, 0)  \
    MM(FN, a, b, 0, 0, 1) MM(FN, a, b, 0, 1, 1) MM(FN, a, b, 1, 0, 1) MM(FN, a, b, 1, 1, 1)  \
    MM(FN, a, b, 2, 0, 1) MM(FN, a, b, 2, 1, 1) MM(FN, a, b, 3, 0, 1) MM(FN, a, b, 3, 1, 1)  \
    }                                                                                        \
    if (X_PRIO_MODE == 0) __builtin_amdgcn_s_setprio(0);                                     \
    PINB                                                                                     \
  }
// one K tile: FB0 holds B0(kt) on entry; FB1 receives B1(kt) and then B0(kt+1)
#define KTILE(kt, FB0, FB1, KIND)                                                             \
  {                                                                                           \
    const int cur = (kt) & 1, nxt = cur ^ 1;                                                  \
    const bool more1 = (kt) + 1 < nk, more2 = (kt) + 2 < nk;                                  \
    const char* sb = smem + cur * 65536;                                                      \
    /* P0: confirm B1(kt); read A0(kt); issue A0(kt+1) */                                     \
    if (X_LOAD_ORDER != 2) { if (more1) WAIT_VM(4); else WAIT_VM(2); }                        \
    if (X_LOAD_ORDER == 1 && more1) GA(0, nxt, (kt) + 1)                                      \
    LD_M(sb, 0)                                                                               \
    if (X_LOAD_ORDER != 1 && more1) GA(0, nxt, (kt) + 1)                                      \
    if (X_LOAD_ORDER == 2) { if (more1) WAIT_VM(6); else WAIT_VM(2); }                        \
    BAR                                                                                       \
    QUADX(FB0, 0, 0, KIND)                                                                          \
    BAR                                                                                       \
    /* P1: confirm A1(kt); read B1(kt); issue B1(kt+1) */                                     \
    if (X_LOAD_ORDER != 2) { if (more1) WAIT_VM(4); else WAIT_VM(0); }                        \
    if (X_LOAD_ORDER == 1 && more1) GW(1, nxt, (kt) + 1)                                      \
    LD_N(FB1, sb, 1)                                                                          \
    if (X_LOAD_ORDER != 1 && more1) GW(1, nxt, (kt) + 1)                                      \
    if (X_LOAD_ORDER == 2) { if (more1) WAIT_VM(6); else WAIT_VM(0); }                        \
    BAR                                                                                       \
    QUADX(FB1, 0, 1, KIND)                                                                          \
    BAR                                                                                       \
    /* P2: confirm B0(kt+1); read A1(kt); issue A1(kt+1) */                                   \
    if (X_LOAD_ORDER != 2 && more1) WAIT_VM(4);                                               \
    if (X_LOAD_ORDER == 1 && more1) GA(1, nxt, (kt) + 1)                                      \
    LD_M(sb, 1)                                                                               \
    if (X_LOAD_ORDER != 1 && more1) GA(1, nxt, (kt) + 1)                                      \
    if (X_LOAD_ORDER == 2 && more1) WAIT_VM(6);                                               \
    BAR                                                                                       \
    QUADX(FB1, 1, 1, KIND)                                                                          \
    BAR                                                                                       \
    /* P3: confirm A0(kt+1); read B0(kt+1) into the set B1 vacated; issue B0(kt+2) */         \
    if (X_LOAD_ORDER != 2 && more1) WAIT_VM(4);                                               \
    if (X_LOAD_ORDER == 1 && more2) GW(0, cur, (kt) + 2)                                      \
    if (more1) LD_N(FB1, smem + nxt * 65536, 0)                                               \
    if (X_LOAD_ORDER != 1 && more2) GW(0, cur, (kt) + 2)                                      \
    if (X_LOAD_ORDER == 2 && more1) { if (more2) WAIT_VM(6); else WAIT_VM(4); }               \
    BAR                                                                                       \
    QUADX(FB0, 1, 0, KIND)                                                                          \
    BAR                                                                                       \
  }

  // prologue: B0, A0, B1, A1 of tile 0 and B0 of tile 1; B0(0) in registers, A0(0) confirmed
  GW(0, 0, 0) GA(0, 0, 0) GW(1, 0, 0) GA(1, 0, 0)
  if (nk > 1) GW(0, 1, 1)
  bool first_tile = true;
#if defined(AACLIP_MEASURE) && defined(X_WALK_STAMP)   // tools/walk_stamps.py: where a walked tile's time goes, per wave
  unsigned long long ws_wait = 0, ws_loop = 0, ws_pre = 0, ws_epi = 0, ws_tiles = 0, ws_t0 = stamp(), ws_first = 0;
#define WS(x) x
#else
#define WS(x)
#endif
#pragma unroll 1
  for (;;) {   // (one trip unless WALK)
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;
  FragPair<T, NP != 0> fm[4], fnX[2], fnY[2];
  WS(const unsigned long long ws_a = stamp();)
  if (!WALK || first_tile) {
    if (nk > 1) WAIT_VM(6); else WAIT_VM(4);
  } else {
    // a later tile of a walk: its first K tile was issued before the previous epilogue's stores, B0 of its second K tile
    // after them.  Loads return in order among loads: once at most two operations are outstanding, every load older than
    // the two youngest has landed (an outstanding older load would make three).  Stores still in flight only make the
    // counted waits of the K loop stricter.
    if (nk > 1) WAIT_VM(2); else WAIT_VM(0);
  }
  BAR
  WS(const unsigned long long ws_b = stamp(); if (first_tile) ws_first = ws_b - ws_t0; else ws_wait += ws_b - ws_a;)
  LD_N(fnX, smem, 0)
#ifndef X_NO_STAGGER
  if (wr == 1) BAR   // waves 4-7 now run one segment behind waves 0-3
#endif
  if (X_PRIO_MODE == 1 && wr == 1) __builtin_amdgcn_s_setprio(1);   // experiment: static priority for the younger half, no flips
  if constexpr (NP == 0) {
    for (int kt = 0; kt < nk; kt += 2) {   // nk is even (checked by the launcher)
      KTILE(kt, fnX, fnY, 0)
      KTILE(kt + 1, fnY, fnX, 0)
    }
  } else if constexpr (NP == 4) {          // per pair of K tiles: fp16, fp16, Al8.Wh8, Ah8.Wl8
    for (int kt = 0; kt < nk; kt += 4) {
      KTILE(kt, fnX, fnY, 0)
      KTILE(kt + 1, fnY, fnX, 0)
      KTILE(kt + 2, fnX, fnY, 1)
      KTILE(kt + 3, fnY, fnX, 2)
    }
  } else {                                 // weight exact in fp16: fp16, fp16, Al8.Wh8; two periods per trip
    for (int kt = 0; kt < nk; kt += 6) {   // (K is a multiple of 256: checked by the launcher)
      KTILE(kt, fnX, fnY, 0)
      KTILE(kt + 1, fnY, fnX, 0)
      KTILE(kt + 2, fnX, fnY, 1)
      KTILE(kt + 3, fnY, fnX, 0)
      KTILE(kt + 4, fnX, fnY, 0)
      KTILE(kt + 5, fnY, fnX, 1)
    }
  }
#ifndef X_NO_STAGGER
  if (wr == 0) BAR   // balance the barrier count of the two groups
#endif
  if constexpr (!WALK) {
    epilogue256t<T, EPI, NP != 0, QK8>(p, acc, smem, tm, tn, wave, lane, fold_pre ? ab_pre : nullptr);
    break;
  } else {
    // every wave is past its last MFMA cluster, and stage 0 was last read one K tile ago (nk is even: the last K tile lives
    // in stage 1): the next tile's first K tile goes there now, under this tile's epilogue, which stages in stage 1 alone
    int tm2 = 0, tn2 = 0, vb2 = vb + (int)gridDim.x;
    while (vb2 < stagger && !map_tile(vb2, tm2, tn2)) vb2 += gridDim.x;
    const bool more_tiles = vb2 < stagger;
    WS(const unsigned long long ws_c = stamp(); ws_loop += ws_c - ws_b;)
    const __amdgpu_buffer_rsrc_t rsA2 = desc_a(more_tiles ? tm2 : tm), rsW2 = desc_w(more_tiles ? tn2 : tn);
    if (more_tiles) { GWx(rsW2, 0, 0, 0) GAx(rsA2, 0, 0, 0) GWx(rsW2, 1, 0, 0) GAx(rsA2, 1, 0, 0) }
    WS(const unsigned long long ws_d = stamp(); ws_pre += ws_d - ws_c;)
    epilogue256t<T, EPI, true, QK8, 4, true>(p, acc, smem + 65536, tm, tn, wave, lane, nullptr);
    WS(ws_epi += stamp() - ws_d; ws_tiles += 1;)
#if defined(AACLIP_MEASURE) && defined(X_WALK_STAMP)
    if (!more_tiles) {
      if (lane == 0) {
        const int w = (int)blockIdx.x * 8 + wave;
        if (w < 16384) {
          g_stamp[6 * w + 0] = ws_wait; g_stamp[6 * w + 1] = ws_loop; g_stamp[6 * w + 2] = ws_pre;
          g_stamp[6 * w + 3] = ws_epi; g_stamp[6 * w + 4] = ws_tiles; g_stamp[6 * w + 5] = ws_first;
        }
      }
    }
#endif
    if (!more_tiles) break;
    __syncthreads();   // every wave is done with its staging rows: stage 1 may take B0 of the next tile's second K tile
    if (nk > 1) GWx(rsW2, 0, 1, 1)
    tm = tm2;
    tn = tn2;
    vb = vb2;
    rsA = rsA2;
    rsW = rsW2;
    first_tile = false;
  }
  }
#undef DMA
#undef GA
#undef GW
#undef GAx
#undef GWx
#undef WS
#undef WAIT_VM
#undef LGKM0
#undef BAR
#undef PINB
#undef LD_M
#undef ODD_OFF
#undef LD_N
#undef MM
#undef MM8
#undef QUADX
#undef KTILE
}

// ------------------------------------------------------------------------------------------------------------------
// Half-tile kernel (round 4): 256 x 128 tile, 4 waves (2 M x 2 N, 128 x 64 of C per wave as above), TWO workgroups per
// CU (80 KiB of LDS each), so that one workgroup's prologue / epilogue runs beside the other's K loop -- the 8-wave
// kernel above owns its CU and spends 29 % of a tile's life outside the K loop with the matrix pipe idle
// (profiles/r04_*).  tools/kloop_skeleton.hip measured the economy first: the K loop of any
// kernel fed by buffer_load ... lds is bound by the PIECES per MFMA (a 1 KiB piece costs its CU ~46 cycles of DMA
// throughput), a lone workgroup of 4 waves already draws that rate, and 12 pieces per 64 MFMAs (this tile) sustain
// 69 % of the bare MFMA rate against 72-76 % for 8 (the 256 x 256 tile) -- a few percent of K-loop rate for the overlap.
//   * One wave per SIMD per workgroup: a wave hides its own LDS latency.  Fragments are refilled IN PLACE behind the last
//     MFMA that reads them (A side: fm[t] right after the four MFMAs of row tile t) or into the N-side set that is dead
//     (as in the kernel above), so the register budget is the 8-wave kernel's.
//   * LDS: A double-buffered (2 x 32 KiB), W single-buffered (16 KiB; weights come from L2, 3 phases of flight are
//     enough).  Chunks: A0 / A1 = the row halves a phase reads (16 KiB, 4 pieces per wave), B0 / B1 (8 KiB, 2 pieces).
//   * Per K tile (phases = output quadrants (A0,B0) (A0,B1) (A1,B1) (A1,B0), 16 MFMAs each):
//       issue    P0: B0(t+1)   P1: B1(t+1)             P2: A0(t+2)   P3: A1(t+2)
//       read     P0: B1(t)     P1: A1(t) (in place)    P2: --        P3: B0(t+1), A0(t+1) (in place)
//       confirm (counted vmcnt at the phase end, then s_barrier) the chunk(s) the NEXT phase reads; P1 needs neither.
//     DMA instructions of one wave complete in order, issue order = A0 A1 B0 B1 of tile 0, A0 A1 of tile 1, then the
//     table: the counts below follow from it.
// Same products, same K order per accumulator as the 8-wave kernel: results are bit-identical to it.
template <typename T, int EPI, int NP = 0, bool QK8 = false>
__global__ __launch_bounds__(256, 2) void gemm16_256h_kernel(GemmParams p, int PN, int patches_n, int total_patches, int PM, int stagger) {
  typedef typename Elem<T>::vec8 vec8;
  __shared__ __attribute__((aligned(16))) char smem[81920];   // [A stage 0: 32 KiB][A stage 1: 32 KiB][W: 16 KiB]

  if (NP != 0) fp8_saturate_mode();   // the split8 epilogues convert with split8x4_sat
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c16 = lane & 15, q4 = lane >> 4;
  const int wr = wave >> 1, wc = wave & 1;
  const int tiles_m = (p.M + 255) >> 8;
  int tm, tn;
  {
    const int P = PM * PN;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int gp = (j / P) * 8 + xcd, local = j % P;
    if (gp >= total_patches) return;
    const int pm = gp / patches_n, pn = gp - pm * patches_n;
    tm = pm * PM + local / PN;
    tn = pn * PN + local % PN;
    if (tm >= tiles_m) return;
  }
  if (stagger > 0 && blockIdx.x < 512 && (blockIdx.x & 256)) {
    // The two workgroups of a CU start together and, left alone, stay in phase: both in their K loops, both in their
    // epilogues -- nothing overlaps.  The second workgroup of every CU (the dispatcher fills all CUs once before it
    // doubles up: ids 256..511 of the first 512) therefore starts `stagger` cycles per K tile late, about half a tile's
    // life; the offset then persists (a workgroup that ends late is replaced late).
    const long long target = (long long)stagger * vtile_count<NP>(p.K);
    const long long t0 = __builtin_amdgcn_s_memtime();
    while ((long long)__builtin_amdgcn_s_memtime() - t0 < target) __builtin_amdgcn_s_sleep(32);
  }
  const int ldw = NP == 4 ? 2 * p.K : (NP == 3 ? p.K + (p.K >> 1) : p.K);   // W row stride in halves
  // DMA pieces (1 KiB = 8 rows of the tile image): wave w fetches pieces w + 4m, which all share ONE lane pattern
  // (tile_src_id depends on the piece index mod 4 only), so a piece is (this lane's offset) + (scalar: 32m rows)
  int srcA, srcW;
  {
    int row, chunk;
    tile_src_id(wave * 64 + lane, row, chunk);
    srcA = (row * (int)p.lda + chunk * 8) * 2;
    srcW = (row * ldw + chunk * 8) * 2;
  }
  const T* baseA = uniform_ptr((const T*)p.A + (long)tm * 256 * p.lda);
  const T* baseW = uniform_ptr((const T*)p.W + (long)tn * 128 * ldw);
  // rows past M are outside the descriptor and read as zero (no clamp: the scalar row offsets need none)
  const long bytesA = ((long)p.M - (long)tm * 256) * p.lda * 2;
  const int recA = (int)(bytesA < 0x7FFFFFF0L ? bytesA : 0x7FFFFFF0L);
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)baseA, 0, __builtin_amdgcn_readfirstlane(recA), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)baseW, 0, 0x7FFFFFF0, 0x00020000);
  const int rA32 = __builtin_amdgcn_readfirstlane(32 * (int)p.lda * 2), rW32 = __builtin_amdgcn_readfirstlane(32 * ldw * 2);
  int offM[2][2], offN[2][2];   // [ks][tile parity]
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int par = 0; par < 2; ++par) {
      offM[ks][par] = tile_off_id(wr * 128 + par * 16 + c16, 4 * ks + q4);
      offN[ks][par] = 65536 + tile_off_id(wc * 64 + par * 16 + c16, 4 * ks + q4);
    }

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;
  f32x2 ab_pre[8];
  const bool fold_pre = NP == 0 && (EPI == EPI_BIAS || EPI == EPI_BIAS_GELU) && p.row_ab != nullptr;
  if (fold_pre) {
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) {
      int row = tm * 256 + wr * 128 + mi * 16 + c16;
      row = row < p.M ? row : p.M - 1;
      ab_pre[mi] = *(const f32x2*)(p.row_ab + 2L * row);
    }
  }

  const int nk = vtile_count<NP>(p.K);   // (virtual) K tiles
  int sc_pack = (127 - SPLIT8_ACT_LO_EXP) | ((127 - SPLIT8_W_HI_EXP) << 8) | ((127 - SPLIT8_W_LO_EXP) << 16) |
                ((127 - SPLIT8_ACT_HI_EXP) << 24);
  (void)sc_pack;
#define DMA(rs, src, dst, so) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(smem + (dst)), 16, src, so, 0, 0);
// piece j (0..3) of A chunk `sub` of tile kt into stage st: m = {0, 1, 4, 5}[j] + 2 sub; piece j (0..1) of W chunk `sub`: m = 2j + sub
// (offsets go through plain int locals: a template-dependent expression as a direct argument makes the builtin call
// type-dependent, and the HOST pass then drops the kernel's stub without a diagnostic)
#define PA(sub, j, st, kt) { int kd; const int m_ = ((j) & 1) + 4 * ((j) >> 1) + 2 * (sub); \
    const int so_ = vtile_off<NP>(kt, p.K, kd) + m_ * rA32; const int ds_ = (st) * 32768 + (wave + 4 * m_) * 1024; DMA(rsA, srcA, ds_, so_) }
#define PW(sub, j, kt) { int kd; const int m_ = 2 * (j) + (sub); \
    const int so_ = vtile_off<NP>(kt, p.K, kd) + m_ * rW32; const int ds_ = 65536 + (wave + 4 * m_) * 1024; DMA(rsW, srcW, ds_, so_) }
#define WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define WAIT_LGKM(n) asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory")
#define BAR __builtin_amdgcn_s_barrier();
#define PINB __builtin_amdgcn_sched_barrier(0);
#define ODD_OFF(off) ({ int o_; asm volatile("v_xor_b32 %0, 0x80, %1" : "=v"(o_) : "v"(off)); o_ + 2048; })
// row tile t (0..3) of A half a from the A stage at sb; N-side tiles of W half b
#define LD_M1(sb, a, t)                                                                     \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                           \
      fm[t].set(ks, *(const vec8*)((sb) + ((NP != 0 && ((t) & 1)) ? ODD_OFF(offM[ks][0]) : offM[ks][(t) & 1]) + ((a) * 2 + ((t) >> 1)) * 4096));
#define LD_N(FN, b)                                                                         \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int t = 0; t < 2; ++t) \
      FN[t].set(ks, *(const vec8*)(smem + ((NP != 0 && t == 1) ? ODD_OFF(offN[ks][0]) : offN[ks][t]) + (b) * 4096));
#define MM(FN, a, b, t, u, ks) acc[4 * (a) + (t)][2 * (b) + (u)] = Mma16<T>::mma(FN[u].get(ks), fm[t].get(ks), acc[4 * (a) + (t)][2 * (b) + (u)]);
#define MM8(FN, a, b, t, u, KIND) acc[4 * (a) + (t)][2 * (b) + (u)] = mma_e4m3k<(KIND) == 1 ? 1 : 2>(FN[u], fm[t], acc[4 * (a) + (t)][2 * (b) + (u)], sc_pack);
// the MFMAs of row tile t of a quadrant (KIND literal: 0 fp16 tile, 1 / 2 e4m3 correction tiles)
#define ROWT(FN, a, b, t, KIND)                                                              \
  if ((KIND) != 0) { MM8(FN, a, b, t, 0, KIND) MM8(FN, a, b, t, 1, KIND) }                   \
  else { MM(FN, a, b, t, 0, 0) MM(FN, a, b, t, 1, 0) MM(FN, a, b, t, 0, 1) MM(FN, a, b, t, 1, 1) }
// one K tile: FB0 holds B0(kt) on entry and fm holds A0(kt); FB1 receives B1(kt) and then B0(kt+1)
#define KTILE(kt, FB0, FB1, KIND)                                                             \
  {                                                                                           \
    const int cur = (kt) & 1, nxt = cur ^ 1;                                                  \
    const bool more1 = (kt) + 1 < nk, more2 = (kt) + 2 < nk;                                  \
    const char* sa = smem + cur * 32768;                                                      \
    const char* sn = smem + nxt * 32768;                                                      \
    /* P0 (A0,B0): read B1(kt); issue B0(kt+1); confirm A1(kt) */                             \
    LD_N(FB1, 1)                                                                              \
    PINB                                                                                      \
    ROWT(FB0, 0, 0, 0, KIND) PINB                                                             \
    if (more1) PW(0, 0, (kt) + 1)                                                             \
    ROWT(FB0, 0, 0, 1, KIND) PINB                                                             \
    if (more1) PW(0, 1, (kt) + 1)                                                             \
    ROWT(FB0, 0, 0, 2, KIND) PINB                                                             \
    ROWT(FB0, 0, 0, 3, KIND) PINB                                                             \
    WAIT_LGKM(0);                                                                             \
    if (more1) WAIT_VM(14); else WAIT_VM(4);                                                  \
    BAR                                                                                       \
    /* P1 (A0,B1): A1(kt) replaces A0(kt) tile by tile; issue B1(kt+1) */                      \
    ROWT(FB1, 0, 1, 0, KIND) LD_M1(sa, 1, 0) PINB                                             \
    if (more1) PW(1, 0, (kt) + 1)                                                             \
    ROWT(FB1, 0, 1, 1, KIND) LD_M1(sa, 1, 1) PINB                                             \
    if (more1) PW(1, 1, (kt) + 1)                                                             \
    ROWT(FB1, 0, 1, 2, KIND) LD_M1(sa, 1, 2) PINB                                             \
    ROWT(FB1, 0, 1, 3, KIND) LD_M1(sa, 1, 3) PINB                                             \
    /* P2 (A1,B1): issue A0(kt+2); confirm B0(kt+1), A0(kt+1) */                              \
    if (more2) PA(0, 0, cur, (kt) + 2)                                                        \
    ROWT(FB1, 1, 1, 0, KIND) PINB                                                             \
    if (more2) PA(0, 1, cur, (kt) + 2)                                                        \
    ROWT(FB1, 1, 1, 1, KIND) PINB                                                             \
    if (more2) PA(0, 2, cur, (kt) + 2)                                                        \
    ROWT(FB1, 1, 1, 2, KIND) PINB                                                             \
    if (more2) PA(0, 3, cur, (kt) + 2)                                                        \
    ROWT(FB1, 1, 1, 3, KIND) PINB                                                             \
    if (more1) { if (more2) WAIT_VM(6); else WAIT_VM(2); }                                    \
    BAR                                                                                       \
    /* P3 (A1,B0): read B0(kt+1) into the set B1 vacated; A0(kt+1) replaces A1(kt); issue A1(kt+2); confirm B1(kt+1) */ \
    if (more1) LD_N(FB1, 0)                                                                   \
    PINB                                                                                      \
    if (more2) PA(1, 0, cur, (kt) + 2)                                                        \
    ROWT(FB0, 1, 0, 0, KIND) if (more1) LD_M1(sn, 0, 0) PINB                                  \
    if (more2) PA(1, 1, cur, (kt) + 2)                                                        \
    ROWT(FB0, 1, 0, 1, KIND) if (more1) LD_M1(sn, 0, 1) PINB                                  \
    if (more2) PA(1, 2, cur, (kt) + 2)                                                        \
    ROWT(FB0, 1, 0, 2, KIND) if (more1) LD_M1(sn, 0, 2) PINB                                  \
    if (more2) PA(1, 3, cur, (kt) + 2)                                                        \
    ROWT(FB0, 1, 0, 3, KIND) if (more1) LD_M1(sn, 0, 3) PINB                                  \
    if (more1) { WAIT_LGKM(8); if (more2) WAIT_VM(8); else WAIT_VM(0); }                      \
    BAR                                                                                       \
  }

  // prologue: all of tile 0 and the A chunks of tile 1, in the order the counted waits assume
  PA(0, 0, 0, 0) PA(0, 1, 0, 0) PA(0, 2, 0, 0) PA(0, 3, 0, 0)
  PA(1, 0, 0, 0) PA(1, 1, 0, 0) PA(1, 2, 0, 0) PA(1, 3, 0, 0)
  PW(0, 0, 0) PW(0, 1, 0) PW(1, 0, 0) PW(1, 1, 0)
  if (nk > 1) {
    PA(0, 0, 1, 1) PA(0, 1, 1, 1) PA(0, 2, 1, 1) PA(0, 3, 1, 1)
    PA(1, 0, 1, 1) PA(1, 1, 1, 1) PA(1, 2, 1, 1) PA(1, 3, 1, 1)
  }
  FragPair<T, NP != 0> fm[4], fnX[2], fnY[2];
  if (nk > 1) WAIT_VM(8); else WAIT_VM(0);      // A0, A1, B0, B1 of tile 0 have landed (this wave's pieces)
  BAR
  LD_N(fnX, 0)
  LD_M1(smem, 0, 0) LD_M1(smem, 0, 1) LD_M1(smem, 0, 2) LD_M1(smem, 0, 3)
  WAIT_LGKM(0);
  BAR                                           // every wave holds B0(0): the slot may take B0(1)
  if constexpr (NP == 0) {
    for (int kt = 0; kt < nk; kt += 2) {   // nk is even (checked by the launcher)
      KTILE(kt, fnX, fnY, 0)
      KTILE(kt + 1, fnY, fnX, 0)
    }
  } else if constexpr (NP == 4) {          // per pair of K tiles: fp16, fp16, Al8.Wh8, Ah8.Wl8
    for (int kt = 0; kt < nk; kt += 4) {
      KTILE(kt, fnX, fnY, 0)
      KTILE(kt + 1, fnY, fnX, 0)
      KTILE(kt + 2, fnX, fnY, 1)
      KTILE(kt + 3, fnY, fnX, 2)
    }
  } else {                                 // weight exact in fp16: fp16, fp16, Al8.Wh8
    // One period (3 tiles) per trip: the N-side sets have swapped roles after an odd number of tiles, so B0 of the next
    // tile is moved back into fnX (16 v_mov per 3 tiles); the A stage parity is then a run-time value.  (Two periods per
    // trip with literal roles, as in the 8-wave kernel, spilled 9-10 registers here.)
#pragma unroll 1
    for (int kt = 0; kt < nk; kt += 3) {
      KTILE(kt, fnX, fnY, 0)
      KTILE(kt + 1, fnY, fnX, 0)
      KTILE(kt + 2, fnX, fnY, 1)
      fnX[0] = fnY[0];
      fnX[1] = fnY[1];
    }
  }
#undef DMA
#undef PA
#undef PW
#undef WAIT_VM
#undef WAIT_LGKM
#undef BAR
#undef PINB
#undef ODD_OFF
#undef LD_M1
#undef LD_N
#undef MM
#undef MM8
#undef ROWT
#undef KTILE
  epilogue256t<T, EPI, NP != 0, QK8, 2>(p, acc, smem, tm, tn, wave, lane, fold_pre ? ab_pre : nullptr);
}

// launcher of the half-tile kernel: 8 x PN patches of 256 x 128 tiles per XCD (64 workgroups = the 2 per CU an XCD holds)
template <typename T, int NP>
static void launch_h(int epi, const GemmParams& p, hipStream_t s) {
  const int tiles_n = p.N / 128, tiles_m = (p.M + 255) / 256;
  int PN = 1;
  for (int c : {8, 6, 4, 3, 2}) if (tiles_n % c == 0) { PN = c; break; }
  const int patches_n = tiles_n / PN, PMx = 8;
  const int pm_x = (tiles_m + PMx - 1) / PMx;
  const int total_x = patches_n * pm_x;
  dim3 gx(((total_x + 7) / 8) * 8 * PMx * PN), b(256);
  static const int stg = getenv("AACLIP_H_STAGGER") ? atoi(getenv("AACLIP_H_STAGGER")) : 1400;   // cycles per K tile
  switch (epi) {
    case EPI_BIAS:
      if constexpr (NP != 0) {
        if (p.out_qk8) { hipLaunchKernelGGL((gemm16_256h_kernel<T, EPI_BIAS, NP, true>), gx, b, 0, s, p, PN, patches_n, total_x, PMx, stg); break; }
      }
      hipLaunchKernelGGL((gemm16_256h_kernel<T, EPI_BIAS, NP>), gx, b, 0, s, p, PN, patches_n, total_x, PMx, stg);
      break;
    case EPI_BIAS_GELU: hipLaunchKernelGGL((gemm16_256h_kernel<T, EPI_BIAS_GELU, NP>), gx, b, 0, s, p, PN, patches_n, total_x, PMx, stg); break;
    case EPI_BIAS_RESID: hipLaunchKernelGGL((gemm16_256h_kernel<T, EPI_BIAS_RESID, NP>), gx, b, 0, s, p, PN, patches_n, total_x, PMx, stg); break;
    case EPI_ACT_F32: hipLaunchKernelGGL((gemm16_256h_kernel<T, EPI_ACT_F32, NP>), gx, b, 0, s, p, PN, patches_n, total_x, PMx, stg); break;
    case EPI_PATCH: hipLaunchKernelGGL((gemm16_256h_kernel<T, EPI_PATCH, NP>), gx, b, 0, s, p, PN, patches_n, total_x, PMx, stg); break;
    default: set_launch_error("gemm: no 256-tile kernel for this epilogue");
  }
  if (p.out_qk8 && (epi != EPI_BIAS || NP == 0)) set_launch_error("gemm: out_qk8 goes with the bias epilogue of the split kernels only");
}

// split fp16 (AACLIP_F16X2): the default kernel on split8 operands, 4 (3: W exact in fp16) virtual tiles per K-tile pair
// XCD patch shape of the 8-wave kernels: PM x PN tiles of one patch run together on one XCD (32 CUs).  8 x 4 by default;
// AACLIP_GEMM_PATCH="pm,pn" overrides it for the traffic experiment of profiles/r04_cfc_patch_shapes.txt (a patch
// column re-reads the A rows once per XCD, a patch row the W rows).  Read once per process; pn must divide N / 256.
static void patch_shape(int tiles_n, int& PMx, int& PN) {
  PN = (tiles_n % 4 == 0) ? 4 : (tiles_n % 3 == 0) ? 3 : (tiles_n % 2 == 0) ? 2 : 1;
  PMx = 8;
  static const char* env = getenv("AACLIP_GEMM_PATCH");
  if (env) {
    int pm = 0, pn = 0;
    if (sscanf(env, "%d,%d", &pm, &pn) == 2 && pm >= 1 && pn >= 1 && tiles_n % pn == 0) { PMx = pm; PN = pn; }
  }
}

// WALK: one workgroup per CU walks the tiles of the same virtual grid (gemm16_256x_kernel<..., WALK>)
static int walk_grid() {
  static int g = 0;
  if (g == 0) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 8)
      cus = 256;
    g = cus / 8 * 8;   // a multiple of the 8 XCDs: workgroup w and its later tiles w + g, w + 2g, ... stay on XCD w % 8
  }
  return g;
}

template <int NP, bool WALK = false>
static void launch_split(int epi, const GemmParams& p, hipStream_t s) {
  const int tiles_n = p.N / 256, tiles_m = (p.M + 255) / 256;
  int PN, PMx;
  patch_shape(tiles_n, PMx, PN);
  const int patches_n = tiles_n / PN;
  const int pm_x = (tiles_m + PMx - 1) / PMx;
  const int total_x = patches_n * pm_x;
  const int vgrid = ((total_x + 7) / 8) * 8 * PMx * PN;
  dim3 gx(WALK ? (vgrid < walk_grid() ? vgrid : walk_grid()) : vgrid), b(512);
  const int last = WALK ? vgrid : 1;   // WALK: the size of the virtual grid; otherwise the start stagger (off)
  switch (epi) {
    case EPI_BIAS:
      if (p.out_qk8) hipLaunchKernelGGL((gemm16_256x_kernel<f16, EPI_BIAS, NP, true, WALK>), gx, b, 0, s, p, PN, patches_n, total_x, PMx, last);
      else hipLaunchKernelGGL((gemm16_256x_kernel<f16, EPI_BIAS, NP, false, WALK>), gx, b, 0, s, p, PN, patches_n, total_x, PMx, last);
      break;
    case EPI_BIAS_GELU: hipLaunchKernelGGL((gemm16_256x_kernel<f16, EPI_BIAS_GELU, NP, false, WALK>), gx, b, 0, s, p, PN, patches_n, total_x, PMx, last); break;
    case EPI_BIAS_RESID: hipLaunchKernelGGL((gemm16_256x_kernel<f16, EPI_BIAS_RESID, NP, false, WALK>), gx, b, 0, s, p, PN, patches_n, total_x, PMx, last); break;
    case EPI_ACT_F32: hipLaunchKernelGGL((gemm16_256x_kernel<f16, EPI_ACT_F32, NP, false, WALK>), gx, b, 0, s, p, PN, patches_n, total_x, PMx, last); break;
    case EPI_PATCH: hipLaunchKernelGGL((gemm16_256x_kernel<f16, EPI_PATCH, NP, false, WALK>), gx, b, 0, s, p, PN, patches_n, total_x, PMx, last); break;
    default: set_launch_error("gemm: no 256-tile kernel for this epilogue");
  }
  if (p.out_qk8 && epi != EPI_BIAS) set_launch_error("gemm: out_qk8 goes with the bias epilogue only");
}

template <typename T>
static void launch_t(int epi, const GemmParams& p, hipStream_t s, int overlapped) {
  const int tiles_n = p.N / 256, tiles_m = (p.M + 255) / 256;
  const int PN = (tiles_n % 4 == 0) ? 4 : (tiles_n % 3 == 0) ? 3 : (tiles_n % 2 == 0) ? 2 : 1;
  const int patches_n = tiles_n / PN, patches_m = (tiles_m + 7) / 8;
  const int total = patches_n * patches_m;
  const int grid = ((total + 7) / 8) * 8 * 8 * PN;
  dim3 g(grid), b(512);
  if (epi < EPI_BIAS || epi > EPI_PATCH) { set_launch_error("gemm: no 256-tile kernel for this epilogue"); return; }
  if (overlapped >= 14 && ((p.K >> 6) & 1)) overlapped = 13;   // the two-set kernels need an even K-tile count
  if (overlapped == 14 || (overlapped >= 30 && overlapped < 40)) {   // staggered, two N-side fragment sets
    // 14: 8-row patches, no start stagger.  30 + 8*b + k (measurement runs): balanced patches if b, k start slots
    int PMx = 8, stg = 1;
    if (overlapped >= 30) {
      const int code = overlapped - 30;
      stg = (code & 7) + 1;
      if (code >> 3) {   // smaller patches until every XCD gets >= 16 of them (even load across the 8 XCDs)
        while (PMx > 1 && ((tiles_m + PMx - 1) / PMx) * patches_n < 128) PMx >>= 1;
      }
    }
    const int pm_x = (tiles_m + PMx - 1) / PMx;
    const int total_x = patches_n * pm_x;
    dim3 gx(((total_x + 7) / 8) * 8 * PMx * PN);
    switch (epi) {
      case EPI_BIAS: hipLaunchKernelGGL((gemm16_256x_kernel<T, EPI_BIAS>), gx, b, 0, s, p, PN, patches_n, total_x, PMx, stg); break;
      case EPI_BIAS_GELU: hipLaunchKernelGGL((gemm16_256x_kernel<T, EPI_BIAS_GELU>), gx, b, 0, s, p, PN, patches_n, total_x, PMx, stg); break;
      case EPI_BIAS_RESID: hipLaunchKernelGGL((gemm16_256x_kernel<T, EPI_BIAS_RESID>), gx, b, 0, s, p, PN, patches_n, total_x, PMx, stg); break;
      case EPI_ACT_F32: hipLaunchKernelGGL((gemm16_256x_kernel<T, EPI_ACT_F32>), gx, b, 0, s, p, PN, patches_n, total_x, PMx, stg); break;
      case EPI_PATCH: hipLaunchKernelGGL((gemm16_256x_kernel<T, EPI_PATCH>), gx, b, 0, s, p, PN, patches_n, total_x, PMx, stg); break;
    }
    return;
  }
  if (overlapped == 13) {   // staggered, both DMA issues in the load segment, buffer_load ... lds
    switch (epi) {
      case EPI_BIAS: hipLaunchKernelGGL((gemm16_256v_kernel<T, EPI_BIAS, 2, 0, true>), g, b, 0, s, p, PN, patches_n, total); break;
      case EPI_BIAS_GELU: hipLaunchKernelGGL((gemm16_256v_kernel<T, EPI_BIAS_GELU, 2, 0, true>), g, b, 0, s, p, PN, patches_n, total); break;
      case EPI_BIAS_RESID: hipLaunchKernelGGL((gemm16_256v_kernel<T, EPI_BIAS_RESID, 2, 0, true>), g, b, 0, s, p, PN, patches_n, total); break;
      case EPI_ACT_F32: hipLaunchKernelGGL((gemm16_256v_kernel<T, EPI_ACT_F32, 2, 0, true>), g, b, 0, s, p, PN, patches_n, total); break;
      case EPI_PATCH: hipLaunchKernelGGL((gemm16_256v_kernel<T, EPI_PATCH, 2, 0, true>), g, b, 0, s, p, PN, patches_n, total); break;
    }
    return;
  }
#ifdef AACLIP_MEASURE
  if (overlapped == 12) {   // staggered + overlapped LDS reads
    switch (epi) {
      case EPI_BIAS: hipLaunchKernelGGL((gemm16_256w_kernel<T, EPI_BIAS>), g, b, 0, s, p, PN, patches_n, total); break;
      case EPI_BIAS_GELU: hipLaunchKernelGGL((gemm16_256w_kernel<T, EPI_BIAS_GELU>), g, b, 0, s, p, PN, patches_n, total); break;
      case EPI_BIAS_RESID: hipLaunchKernelGGL((gemm16_256w_kernel<T, EPI_BIAS_RESID>), g, b, 0, s, p, PN, patches_n, total); break;
      case EPI_ACT_F32: hipLaunchKernelGGL((gemm16_256w_kernel<T, EPI_ACT_F32>), g, b, 0, s, p, PN, patches_n, total); break;
      case EPI_PATCH: hipLaunchKernelGGL((gemm16_256w_kernel<T, EPI_PATCH>), g, b, 0, s, p, PN, patches_n, total); break;
    }
    return;
  }
  if (overlapped == 40 || (overlapped >= 5 && overlapped <= 11)) {
    // Timing ablations and the stamp build of the staggered kernel (GL = 2).  They exist for the fp32-output epilogue
    // only (their stores, where they store at all, are 4 bytes per element): any other epilogue has no such kernel and
    // is an error -- launching one on a 16-bit output buffer overruns it by a factor of two.
    if (epi != EPI_ACT_F32) { set_launch_error("gemm: ablation/stamp kernels exist for the fp32-output epilogue only"); return; }
    switch (overlapped) {
      case 5: hipLaunchKernelGGL((gemm16_256v_kernel<T, EPI_ACT_F32, 2, 1>), g, b, 0, s, p, PN, patches_n, total); break;
      case 6: hipLaunchKernelGGL((gemm16_256v_kernel<T, EPI_ACT_F32, 2, 2>), g, b, 0, s, p, PN, patches_n, total); break;
      case 7: hipLaunchKernelGGL((gemm16_256v_kernel<T, EPI_ACT_F32, 2, 3>), g, b, 0, s, p, PN, patches_n, total); break;
      case 8: hipLaunchKernelGGL((gemm16_256v_kernel<T, EPI_ACT_F32, 2, 4>), g, b, 0, s, p, PN, patches_n, total); break;
      case 9: hipLaunchKernelGGL((gemm16_256v_kernel<T, EPI_ACT_F32, 2, 5>), g, b, 0, s, p, PN, patches_n, total); break;
      case 10: hipLaunchKernelGGL((gemm16_256v_kernel<T, EPI_ACT_F32, 2, 6>), g, b, 0, s, p, PN, patches_n, total); break;
      case 11: hipLaunchKernelGGL((gemm16_256v_kernel<T, EPI_ACT_F32, 2, 7>), g, b, 0, s, p, PN, patches_n, total); break;
      default: hipLaunchKernelGGL((gemm16_256v_kernel<T, EPI_ACT_F32, 2, 8>), g, b, 0, s, p, PN, patches_n, total); break;
    }
    return;
  }
  if (overlapped >= 2 && overlapped <= 4) {   // staggered kernel, GL = overlapped - 2 DMA instructions in the load segment
#define LV(E) { if (overlapped == 2) hipLaunchKernelGGL((gemm16_256v_kernel<T, E, 0>), g, b, 0, s, p, PN, patches_n, total); \
               else if (overlapped == 3) hipLaunchKernelGGL((gemm16_256v_kernel<T, E, 1>), g, b, 0, s, p, PN, patches_n, total); \
               else hipLaunchKernelGGL((gemm16_256v_kernel<T, E, 2>), g, b, 0, s, p, PN, patches_n, total); }
    switch (epi) {
      case EPI_BIAS: LV(EPI_BIAS) break;
      case EPI_BIAS_GELU: LV(EPI_BIAS_GELU) break;
      case EPI_BIAS_RESID: LV(EPI_BIAS_RESID) break;
      case EPI_ACT_F32: LV(EPI_ACT_F32) break;
      case EPI_PATCH: LV(EPI_PATCH) break;
    }
#undef LV
    return;
  }
  if (overlapped == 1) {
    switch (epi) {
      case EPI_BIAS: hipLaunchKernelGGL((gemm16_256u_kernel<T, EPI_BIAS>), g, b, 0, s, p, PN, patches_n, total); break;
      case EPI_BIAS_GELU: hipLaunchKernelGGL((gemm16_256u_kernel<T, EPI_BIAS_GELU>), g, b, 0, s, p, PN, patches_n, total); break;
      case EPI_BIAS_RESID: hipLaunchKernelGGL((gemm16_256u_kernel<T, EPI_BIAS_RESID>), g, b, 0, s, p, PN, patches_n, total); break;
      case EPI_ACT_F32: hipLaunchKernelGGL((gemm16_256u_kernel<T, EPI_ACT_F32>), g, b, 0, s, p, PN, patches_n, total); break;
      case EPI_PATCH: hipLaunchKernelGGL((gemm16_256u_kernel<T, EPI_PATCH>), g, b, 0, s, p, PN, patches_n, total); break;
    }
    return;
  }
  if (overlapped == 0) {
    switch (epi) {
      case EPI_BIAS: hipLaunchKernelGGL((gemm16_256t_kernel<T, EPI_BIAS>), g, b, 0, s, p, PN, patches_n, total); break;
      case EPI_BIAS_GELU: hipLaunchKernelGGL((gemm16_256t_kernel<T, EPI_BIAS_GELU>), g, b, 0, s, p, PN, patches_n, total); break;
      case EPI_BIAS_RESID: hipLaunchKernelGGL((gemm16_256t_kernel<T, EPI_BIAS_RESID>), g, b, 0, s, p, PN, patches_n, total); break;
      case EPI_ACT_F32: hipLaunchKernelGGL((gemm16_256t_kernel<T, EPI_ACT_F32>), g, b, 0, s, p, PN, patches_n, total); break;
      case EPI_PATCH: hipLaunchKernelGGL((gemm16_256t_kernel<T, EPI_PATCH>), g, b, 0, s, p, PN, patches_n, total); break;
    }
    return;
  }
#endif  // AACLIP_MEASURE
  set_launch_error("gemm: unknown 256-tile kernel id");
}

#ifdef AACLIP_MEASURE
#ifdef X_WALK_STAMP
void read_gemm_estamps(double* out8) {   // sums since the last read, then reset
  unsigned long long host[8], zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  (void)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_estamp), sizeof(host));
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_estamp), zero, sizeof(zero));
  for (int j = 0; j < 8; ++j) out8[j] = (double)host[j];
}
#else
void read_gemm_estamps(double* out8) { for (int j = 0; j < 8; ++j) out8[j] = 0; }
#endif
void read_gemm_stamps(double* out6, int nwaves) {
  static unsigned long long host[6 * 16384];
  (void)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_stamp), sizeof(host));
  if (nwaves > 16384) nwaves = 16384;
  double sum[6] = {0, 0, 0, 0, 0, 0};
  int n = 0;
  for (int i = 0; i < nwaves; ++i) {
    if (host[6 * i + 1] == 0) continue;
    for (int j = 0; j < 6; ++j) sum[j] += (double)host[6 * i + j];
    ++n;
  }
  for (int j = 0; j < 6; ++j) out6[j] = n ? sum[j] / n : 0;
}

#endif  // AACLIP_MEASURE

void launch_gemm256t(int dtype, int epi, const GemmParams& p, hipStream_t s, int overlapped) {
  if (overlapped == 15 && dtype != AACLIP_F16X2 && ((p.K >> 6) & 1) != 0) overlapped = 13;   // odd K-tile count: the one-set 8-wave kernel
  if (overlapped == 15) {   // the half-tile kernel (256 x 128, two workgroups per CU)
    if (dtype == AACLIP_F16X2) {
      if (p.w_exact16) launch_h<f16, 3>(epi, p, s);
      else launch_h<f16, 4>(epi, p, s);
    } else if (dtype == AACLIP_F16) launch_h<f16, 0>(epi, p, s);
    else launch_h<bf16, 0>(epi, p, s);
    return;
  }
  if (dtype == AACLIP_F16X2) {
    if (overlapped == 16) {   // the walking form of the 8-wave kernel (split operands only)
      if (p.w_exact16) launch_split<3, true>(epi, p, s);
      else launch_split<4, true>(epi, p, s);
    } else if (p.w_exact16) launch_split<3>(epi, p, s);
    else launch_split<4>(epi, p, s);
  } else if (dtype == AACLIP_F16) launch_t<f16>(epi, p, s, overlapped == 16 ? 14 : overlapped);
  else launch_t<bf16>(epi, p, s, overlapped == 16 ? 14 : overlapped);
}

}  // namespace aaclip
