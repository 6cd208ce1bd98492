// Kernels of the IQM side branch (reference model/iqm.py + the glue of model/adapter.py:186-269 and the maps of
// test_last.py:102-147).  The branch is 2 QUERIES per image attending to 4 x 1369 projected patch
// rows and to the anchors: its matrix products (patch-row projections, key/value projections: ~45 GFLOP per image)
// run on the library's MFMA GEMM; what is left is tiny and HBM- or latency-bound, one wavefront-level kernel each:
//   small_attention      softmax(q k^T / sqrt(d)) v for <= 4 queries per (image, head) over Lk keys
//   residual_layernorm   LayerNorm(a + b)                 (IQM_SelfOutput / IQM_Output, iqm.py:143-154,219-230)
//   combine3             wa a + wb b + wc c               (the fixed 0.4 / 0.3 / 0.3 fusion of iqm.py:311-315)
//   linear_smallk        y = x W^T + b for in_features <= 4 (the anchors arrive as [B, 768, 2]: "768 tokens of width
//                        2", adapter.py:229-246)
//   drop_cls_rows        [B, L, E] -> rows 1.. of every image written at a row offset of [B, NLP, E] (torch.cat of
//                        the projected tap levels, adapter.py:206-211)
//   iqm_scores / iqm_upsample   sigmoid(cos(f, q_abnormal) - cos(f, q_normal)) per patch, bilinear
//                        (align_corners=False) upsampling, level sum and the 0.6 / 0.4 fusion with the text map
#include "common.h"
#include "kernels.h"
#include "mma16.h"

namespace aaclip {

template <typename T> AACLIP_DEV float ldf(const T* p) { return (float)*p; }

// ---- small_attention: grid (H, B), 256 threads.  q [B, nq, H*hd] fp32 (nq <= 4); k, v [B*Lk, H*hd] of T;
// out [B, nq, H*hd] fp32.  Pass 1: one key per thread and step, scores to LDS (Lk <= 8192), block max / sum.
// Pass 2: thread = (4-column chunk, key slice), partial sums reduced through LDS.
constexpr int SA_MAXQ = 4, SA_MAXK = 8192, SA_MAXHD = 128;
template <typename T>
__global__ __launch_bounds__(256) void small_attention_kernel(const float* __restrict__ q, const T* __restrict__ k,
                                                              const T* __restrict__ v, float* __restrict__ out, int nq,
                                                              int Lk, int H, int hd, float scale) {
  extern __shared__ __attribute__((aligned(16))) float sm[];   // [nq][Lk] scores, then reduction scratch
  __shared__ float qs[SA_MAXQ][SA_MAXHD];
  __shared__ float red[SA_MAXQ][256];
  __shared__ float stat[SA_MAXQ][2];
  const int h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int D = H * hd;
  for (int i = tid; i < nq * hd; i += 256) qs[i / hd][i % hd] = q[((long)b * nq + i / hd) * D + h * hd + i % hd] * scale;
  __syncthreads();
  float mx[SA_MAXQ];
#pragma unroll
  for (int a = 0; a < SA_MAXQ; ++a) mx[a] = -INFINITY;
  for (int j = tid; j < Lk; j += 256) {
    const T* kr = k + ((long)b * Lk + j) * D + h * hd;
    float s[SA_MAXQ] = {0.f, 0.f, 0.f, 0.f};
    if (sizeof(T) == 2 && (hd & 7) == 0 && (D & 7) == 0) {
      // 16-byte loads (the row slice of a head starts on a 16-byte boundary when hd and D are multiples of 8); the
      // products are summed in the same element order as the 4-wide loop below
      typedef T t8 __attribute__((ext_vector_type(8)));
      for (int d = 0; d < hd; d += 8) {
        const t8 kv = *(const t8*)(kr + d);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          const float k0 = (float)kv[4 * half], k1 = (float)kv[4 * half + 1], k2 = (float)kv[4 * half + 2],
                      k3 = (float)kv[4 * half + 3];
          const int dd = d + 4 * half;
#pragma unroll
          for (int a = 0; a < SA_MAXQ; ++a)
            if (a < nq)
              s[a] = fmaf(k3, qs[a][dd + 3], fmaf(k2, qs[a][dd + 2], fmaf(k1, qs[a][dd + 1], fmaf(k0, qs[a][dd], s[a]))));
        }
      }
    } else
    for (int d = 0; d < hd; d += 4) {
      const float k0 = ldf(kr + d), k1 = ldf(kr + d + 1), k2 = ldf(kr + d + 2), k3 = ldf(kr + d + 3);
#pragma unroll
      for (int a = 0; a < SA_MAXQ; ++a)
        if (a < nq) s[a] = fmaf(k3, qs[a][d + 3], fmaf(k2, qs[a][d + 2], fmaf(k1, qs[a][d + 1], fmaf(k0, qs[a][d], s[a]))));
    }
#pragma unroll
    for (int a = 0; a < SA_MAXQ; ++a)
      if (a < nq) { sm[a * Lk + j] = s[a]; mx[a] = fmaxf(mx[a], s[a]); }
  }
#pragma unroll
  for (int a = 0; a < SA_MAXQ; ++a) red[a][tid] = mx[a];
  __syncthreads();
  if (tid < nq) {
    float m = -INFINITY;
    for (int i = 0; i < 256; ++i) m = fmaxf(m, red[tid][i]);
    stat[tid][0] = m;
  }
  __syncthreads();
  float sum[SA_MAXQ] = {0.f, 0.f, 0.f, 0.f};
  for (int j = tid; j < Lk; j += 256)
#pragma unroll
    for (int a = 0; a < SA_MAXQ; ++a)
      if (a < nq) { const float p = expf(sm[a * Lk + j] - stat[a][0]); sm[a * Lk + j] = p; sum[a] += p; }
#pragma unroll
  for (int a = 0; a < SA_MAXQ; ++a) red[a][tid] = sum[a];
  __syncthreads();
  if (tid < nq) {
    float t = 0.f;
    for (int i = 0; i < 256; ++i) t += red[tid][i];   // fixed order: deterministic
    stat[tid][1] = 1.0f / t;
  }
  __syncthreads();
  // pass 2
  const int nch = hd / 4;                 // 4-column chunks (hd % 4 == 0)
  const int slices = 256 / nch;           // key slices
  const int c = tid % nch, sl = tid / nch;
  float acc[SA_MAXQ][4];
#pragma unroll
  for (int a = 0; a < SA_MAXQ; ++a) acc[a][0] = acc[a][1] = acc[a][2] = acc[a][3] = 0.f;
  if (sl < slices) {
    typedef T t4 __attribute__((ext_vector_type(4)));
    const bool vec = (hd & 3) == 0 && (D & 3) == 0;
    auto loadv = [&](int j, float (&o4)[4]) {
      const T* vr = v + ((long)b * Lk + j) * D + h * hd + c * 4;
      if (vec) {                              // one 8- / 16-byte load
        const t4 vv = *(const t4*)vr;
        o4[0] = (float)vv[0]; o4[1] = (float)vv[1]; o4[2] = (float)vv[2]; o4[3] = (float)vv[3];
      } else {
        o4[0] = ldf(vr); o4[1] = ldf(vr + 1); o4[2] = ldf(vr + 2); o4[3] = ldf(vr + 3);
      }
    };
    auto fma4 = [&](int j, const float (&v4)[4]) {
#pragma unroll
      for (int a = 0; a < SA_MAXQ; ++a)
        if (a < nq) {
          const float p = sm[a * Lk + j];
          acc[a][0] = fmaf(p, v4[0], acc[a][0]); acc[a][1] = fmaf(p, v4[1], acc[a][1]);
          acc[a][2] = fmaf(p, v4[2], acc[a][2]); acc[a][3] = fmaf(p, v4[3], acc[a][3]);
        }
    };
    // four keys of the slice in flight (the loop is latency-bound: one small load per key); accumulated in key order
    int j = sl;
    for (; j + 3 * slices < Lk; j += 4 * slices) {
      float va[4], vb[4], vc[4], vd[4];
      loadv(j, va); loadv(j + slices, vb); loadv(j + 2 * slices, vc); loadv(j + 3 * slices, vd);
      fma4(j, va); fma4(j + slices, vb); fma4(j + 2 * slices, vc); fma4(j + 3 * slices, vd);
    }
    for (; j < Lk; j += slices) {
      float va[4];
      loadv(j, va);
      fma4(j, va);
    }
  }
  __syncthreads();                        // everyone is done reading the probabilities: reuse sm as [slices][nq][hd]
  if (sl < slices)
#pragma unroll
    for (int a = 0; a < SA_MAXQ; ++a)
      if (a < nq)
#pragma unroll
        for (int e = 0; e < 4; ++e) sm[(sl * nq + a) * hd + c * 4 + e] = acc[a][e];
  __syncthreads();
  for (int i = tid; i < nq * hd; i += 256) {
    const int a = i / hd, d = i % hd;
    float t = 0.f;
    for (int s2 = 0; s2 < slices; ++s2) t += sm[(s2 * nq + a) * hd + d];
    out[((long)b * nq + a) * D + h * hd + d] = t * stat[a][1];
  }
}

// ---- cross_rows: attention of a handful of EFFECTIVE queries over raw (unprojected) rows.
// IQM's cross-attention projects 4 x 1369 patch rows per image through W_k and W_v (2 x 5476 x 768 x 768 MACs per
// layer and image) to be consumed by two queries.  The algebra avoids both products:
//   scores_j = q_h . (W_k[h] x_j + b_k[h]) / sqrt(d) = (W_k[h]^T q_h / sqrt(d)) . x_j + const   (const: softmax-invariant)
//   ctx_h    = sum_j p_j (W_v[h] x_j + b_v[h])       = W_v[h] (sum_j p_j x_j) + b_v[h]
// so per (image, query, head) one effective query qt = W_k[h]^T q_h / sqrt(d) in the rows' own space (a [R, Dk] GEMM on
// the 2-row query side), this kernel for ebar = softmax(qt X^T) X over the raw rows X [Lk, Dk], and one more small GEMM
// for W_v[h] ebar + b_v[h].  R = queries x heads rows of qt per image (16 for IQM), all attending to the same X.
//   partial: grid (slices, B), R / 4 waves; a wave owns 4 rows of qt and walks the slice's keys: the key row is read
//            once per wave (coalesced, 4 elements per lane and 256-column chunk), 4 dot products by wave reduction,
//            online softmax per query row (running maximum, sum, accumulator [4][Dk / 64] per lane)
//   combine: merges the slices' (m, l, acc) in a fixed order (deterministic)
constexpr int CR_SLICES = 16;
// Sum over the 64 lanes on the VALU's DPP paths (rotations inside the rows of 16, then row_bcast:15 / row_bcast:31),
// result broadcast from lane 63 through an SGPR: 6 adds + 1 v_readlane instead of 6 LDS-crossbar shuffles.  The order
// of the additions is fixed, so results do not depend on anything but the inputs.
AACLIP_DEV float wave_sum_dpp(float x) {
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x128, 0xF, 0xF, false));
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x124, 0xF, 0xF, false));
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x122, 0xF, 0xF, false));
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x121, 0xF, 0xF, false));
  // every lane of a row of 16 now holds its row's sum; add row 0 into row 1 and row 2 into row 3, then rows 0+1 into 3
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x142, 0xA, 0xF, false));
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x143, 0xC, 0xF, false));
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 63));
}
template <typename T, int NCH>
__global__ __launch_bounds__(256) void cross_rows_partial_kernel(const float* __restrict__ qt, const T* __restrict__ x,
                                                                 float* __restrict__ part, int R, int Lk, int slices) {
  constexpr int Dk = NCH * 256;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sl = blockIdx.x, b = blockIdx.y;
  const int r0 = wave * 4;
  if (r0 >= R) return;
  const int per = (Lk + slices - 1) / slices;
  const int j0 = sl * per, j1 = min(Lk, j0 + per);
  f32x4 q[4][NCH], acc[4][NCH];
  float m[4], l[4];
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    m[a] = -INFINITY;
    l[a] = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      q[a][c] = *(const f32x4*)(qt + ((long)b * R + r0 + a) * Dk + (c * 64 + lane) * 4);
      acc[a][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  }
  // keys in groups of KB: all loads of a group are issued before any arithmetic (a wave walks its keys one after the
  // other; with one key in flight the loop ran at the memory latency: 0.8 ms per call for 5476 keys)
  constexpr int KB = 4;
  auto load_key = [&](int j, f32x4 (&xv)[NCH]) {
    const int jj = j < j1 ? j : j1 - 1;          // the tail group re-reads the last key (its result is skipped)
    const T* xr = x + ((long)b * Lk + jj) * Dk;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      if (sizeof(T) == 4) {
        xv[c] = *(const f32x4*)((const float*)xr + (c * 64 + lane) * 4);
      } else {
        typedef T t4 __attribute__((ext_vector_type(4)));
        const t4 v = *(const t4*)(xr + (c * 64 + lane) * 4);
        xv[c] = (f32x4){(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
      }
    }
  };
  for (int j = j0; j < j1; j += KB) {
    f32x4 xk[KB][NCH];
#pragma unroll
    for (int u = 0; u < KB; ++u) load_key(j + u, xk[u]);
#pragma unroll
    for (int u = 0; u < KB; ++u) {
      if (j + u >= j1) break;
      const f32x4 (&xv)[NCH] = xk[u];
      float s[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        float d = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c)
          d = fmaf(xv[c][3], q[a][c][3], fmaf(xv[c][2], q[a][c][2], fmaf(xv[c][1], q[a][c][1], fmaf(xv[c][0], q[a][c][0], d))));
        s[a] = wave_sum_dpp(d) * 1.4426950408889634f;   // log2 units: one v_exp_f32 per probability
      }
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        if (s[a] > m[a]) {            // wave-uniform: every lane holds the same score
          const float f = __builtin_amdgcn_exp2f(m[a] - s[a]);   // 2^-inf = 0 on the first key
          l[a] *= f;
#pragma unroll
          for (int c = 0; c < NCH; ++c) acc[a][c] = acc[a][c] * f;
          m[a] = s[a];
        }
        const float pj = __builtin_amdgcn_exp2f(s[a] - m[a]);
        l[a] += pj;
#pragma unroll
        for (int c = 0; c < NCH; ++c) acc[a][c] = acc[a][c] + xv[c] * pj;
      }
    }
  }
  // partial record of (b, slice, row): [Dk accumulator][m][l]
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    float* pr = part + (((long)b * slices + sl) * R + r0 + a) * (Dk + 2);
#pragma unroll
    for (int c = 0; c < NCH; ++c) *(f32x4*)(pr + (c * 64 + lane) * 4) = acc[a][c];
    if (lane == 0) { pr[Dk] = m[a]; pr[Dk + 1] = l[a]; }
  }
}
__global__ __launch_bounds__(256) void cross_rows_combine_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                                 int R, int Dk, int slices) {
  const int r = blockIdx.x, b = blockIdx.y;
  float m = -INFINITY;
  for (int s = 0; s < slices; ++s) m = fmaxf(m, part[(((long)b * slices + s) * R + r) * (Dk + 2) + Dk]);
  float l = 0.f;
  for (int s = 0; s < slices; ++s) {
    const float* pr = part + (((long)b * slices + s) * R + r) * (Dk + 2);
    if (pr[Dk + 1] > 0.f) l += pr[Dk + 1] * exp2f(pr[Dk] - m);   // the partial maxima are in log2 units
  }
  const float inv = 1.0f / l;
  for (int d = threadIdx.x; d < Dk; d += 256) {
    float a = 0.f;
    for (int s = 0; s < slices; ++s) {
      const float* pr = part + (((long)b * slices + s) * R + r) * (Dk + 2);
      if (pr[Dk + 1] > 0.f) a += pr[d] * exp2f(pr[Dk] - m);
    }
    out[((long)b * R + r) * Dk + d] = a * inv;
  }
}
const char* cross_rows_check(int R, int Lk, int Dk) {
  if (R < 4 || R > 16 || (R & 3)) return "cross_rows: 4, 8, 12 or 16 effective queries per image";
  if (Lk < 1) return "cross_rows: no keys";
  if (Dk != 256 && Dk != 512 && Dk != 768 && Dk != 1024) return "cross_rows: row width must be 256, 512, 768 or 1024";
  return nullptr;
}
int cross_rows_slices(int Lk) { return Lk >= 64 * CR_SLICES ? CR_SLICES : (Lk >= 64 ? Lk / 64 : 1); }
size_t cross_rows_ws_bytes(int B, int R, int Lk, int Dk) { return (size_t)B * cross_rows_slices(Lk) * R * (Dk + 2) * 4; }
template <typename T>
static void cross_rows_t(const float* qt, const T* x, float* out, float* part, int B, int R, int Lk, int Dk, hipStream_t s) {
  const int slices = cross_rows_slices(Lk);
  dim3 g(slices, B), blk(64 * (R / 4));
  switch (Dk / 256) {
    case 1: hipLaunchKernelGGL((cross_rows_partial_kernel<T, 1>), g, blk, 0, s, qt, x, part, R, Lk, slices); break;
    case 2: hipLaunchKernelGGL((cross_rows_partial_kernel<T, 2>), g, blk, 0, s, qt, x, part, R, Lk, slices); break;
    case 3: hipLaunchKernelGGL((cross_rows_partial_kernel<T, 3>), g, blk, 0, s, qt, x, part, R, Lk, slices); break;
    case 4: hipLaunchKernelGGL((cross_rows_partial_kernel<T, 4>), g, blk, 0, s, qt, x, part, R, Lk, slices); break;
  }
  hipLaunchKernelGGL(cross_rows_combine_kernel, dim3(R, B), dim3(256), 0, s, part, out, R, Dk, slices);
}
void launch_cross_rows(int x_dtype, const float* qt, const void* x, float* out, void* ws, int B, int R, int Lk, int Dk,
                       hipStream_t s) {
  if (x_dtype == AACLIP_F32) cross_rows_t<float>(qt, (const float*)x, out, (float*)ws, B, R, Lk, Dk, s);
  else if (x_dtype == AACLIP_F16) cross_rows_t<f16>(qt, (const f16*)x, out, (float*)ws, B, R, Lk, Dk, s);
  else cross_rows_t<bf16>(qt, (const bf16*)x, out, (float*)ws, B, R, Lk, Dk, s);
}

// ---- cross_rows over SEGMENTS of 16-bit rows, on the matrix cores (aaclip_cross_rows_levels).
// The visual cross-attention of the IQM layers reads the LayerNorm'ed rows of the four tap levels directly: level k's
// projections (query_adapters[k], visual_feature_proj, the key / value Linear) are moved onto the query side and
// behind the probability-weighted row sums, so that per (image, effective query r) the work is
//     scores_j = qt[b, r, seg] . x_seg[b, j]     over every key row j of every segment (ONE softmax over all of them)
//     out[b, r, seg] = sum_{j in seg} p_j x_seg[b, j]
// 16 effective queries x DK = 1024 columns per key row: 64 KFLOP per row, 23 GFLOP per call at B = 64 -- the VALU kernel
// above needs 0.6 ms for that; as MFMAs it is ~10 us of pipe time and the kernel runs at the rate HBM delivers the rows.
//   workgroup (256 threads) = (image, segment, slice of the segment's keys); 32-key tiles, double-buffered in LDS by
//   buffer_load ... lds issued from asm (hipcc would order every LDS read behind a DMA it knows about); LDS image =
//   plain rows of DK x 2 bytes with the 16-byte chunk index XORed by f(row) = 2 (row & 7) ^ (row >> 2 & 1): the row
//   reads of the score MFMAs (16 rows, one chunk) and the transposed reads of the P.V MFMAs (4 rows x 2 chunks per 16
//   lanes) are both conflict-free per 16-lane group.
//   scores  D[key][q] = X[key][.] . Q[q][.]   16x16x32, A = key rows by ds_read_b128, B = the query fragments (fp16 hi
//           and lo: two MFMAs per chunk, so the effective queries keep ~22 bits); wave w sums over columns
//           [w DK/4, (w+1) DK/4), the four partial tiles meet in LDS and every wave adds them in the same order
//   softmax online, per query = per lane column, identical in every wave (log2 units, fp32)
//   P.V     D[col][q] = X^T[col][key] . P[key][q]: A by ds_read_b64_tr_b16 (keys 4g..4g+3 of both 16-key blocks on lane
//           group g -- exactly the keys whose probabilities that lane group's score registers hold), B = P as it
//           stands, converted to 16 bits; wave w owns output columns [w DK/4, (w+1) DK/4) in 64 / 48 registers.
// Partial record per (image, segment, slice, query): [DK sums][m][l] like the kernel above; the combine kernel applies
// ONE maximum and ONE normaliser over all slices of all segments.
struct CrossSegs { const void* x[4]; };
// chunk swizzle of the LDS image.  ds_read_b128 is serviced in four groups of 16 lanes -- {0-3, 12-15, 20-27}, {4-11,
// 16-19, 28-31} and the same + 32 (MI355X_MICROARCH.md, LDS): rows 0-3 and 12-15 of lane group g together with rows 4-11
// of lane group g + 1, i.e. neighbouring chunks -- so bit 0 of the swizzle must tell row r from row r ^ 8 AFTER the lane
// group's chunk bit (row bit 2 ^ row bit 3 inside a group) is folded in: bit 0 = row bit 2.  The transposed reads (8 rows
// x 2 chunks x 2 halves per 32 lanes) only need (row & 7) in bits 1-3.  tools/lds_bank_model.py-style check: both 1-way.
AACLIP_DEV int cr_swz(int row) { return ((row & 7) << 1) ^ ((row >> 2) & 1); }

template <typename T, int DK>
__global__ __launch_bounds__(256) void cross_rows_mfma_kernel(CrossSegs segs, const float* __restrict__ qt,
                                                              float* __restrict__ part, int R, int rows_per_image,
                                                              int row0, int Lk, int ldx, int nseg, int slices) {
  typedef typename Elem<T>::vec8 vec8;
  constexpr int RB = DK * 2, TILEB = 32 * RB, QW = DK / 4, NCQ = QW / 32, NT = QW / 16, NI = TILEB / 4096;
  __shared__ __attribute__((aligned(16))) char smem[2 * TILEB + 8192];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c16 = lane & 15, g = lane >> 4;
  const int seg = __builtin_amdgcn_readfirstlane(blockIdx.x / slices), sl = blockIdx.x - seg * slices, b = blockIdx.y;
  const int per = (((Lk + slices - 1) / slices) + 31) & ~31;
  const int j0 = sl * per, j1 = min(Lk, j0 + per);
  const int dbase = wave * QW;
  float* prec = part + ((((long)b * nseg + seg) * slices + sl) * 16) * (DK + 2);
  if (j0 >= j1) {   // an empty slice (workgroup-uniform): a record that the combine kernel skips
    if (threadIdx.x < 16) { prec[(long)threadIdx.x * (DK + 2) + DK] = -INFINITY; prec[(long)threadIdx.x * (DK + 2) + DK + 1] = 0.f; }
    return;
  }
  // effective queries of this (image, segment): fp32 -> hi + lo fragments of this wave's column quarter (log2 units)
  vec8 qh[NCQ], ql[NCQ];
  {
    const float* qr = qt + ((long)b * R + (c16 < R ? c16 : 0)) * ((long)nseg * DK) + (long)seg * DK + dbase + 8 * g;
#pragma unroll
    for (int c = 0; c < NCQ; ++c) {
      const f32x4 v0 = *(const f32x4*)(qr + 32 * c), v1 = *(const f32x4*)(qr + 32 * c + 4);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float v = (e < 4 ? v0[e] : v1[e - 4]) * 1.4426950408889634f;
        if (c16 >= R) v = 0.f;
        const T h = (T)v;
        qh[c][e] = h;
        ql[c][e] = (T)(v - (float)h);
      }
    }
  }
  // DMA: descriptor over this segment's rows of image b, per-lane offsets of the wave's NI slots of a tile
  const unsigned long long ubase = (unsigned long long)((const T*)segs.x[seg] + ((long)b * rows_per_image + row0) * ldx);
  u32x4 rs;
  rs[0] = __builtin_amdgcn_readfirstlane((unsigned)ubase);
  rs[1] = __builtin_amdgcn_readfirstlane((unsigned)(ubase >> 32)) & 0xFFFFu;
  rs[2] = 0x7FFFFFF0u;
  rs[3] = 0x00020000u;
  const int ldb = ldx * 2;
  int dvo[NI], drow[NI];
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int o = (wave * NI + j) * 1024 + lane * 16;
    const int row = o / RB, ch = (o - row * RB) >> 4;
    drow[j] = row;
    dvo[j] = row * ldb + ((ch ^ cr_swz(row)) << 4);
  }
  const unsigned lds0 = (unsigned)(size_t)(lds_void*)smem + wave * (NI * 1024);
  auto dma16 = [&](unsigned lds_addr, int voff_b, int soff_b) {
    unsigned keep;
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_addr), "v"(voff_b), "s"(rs), "s"(soff_b) : "memory");
  };
  auto stage = [&](int st, int jt) {
    const unsigned dst = lds0 + st * TILEB;
    const int so = __builtin_amdgcn_readfirstlane(jt * ldb);
    if (jt + 32 <= j1) {
#pragma unroll
      for (int j = 0; j < NI; ++j) dma16(dst + j * 1024, dvo[j], so);
    } else {   // last tile: rows past the slice re-read its last row (finite data; their scores are masked)
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        int over = jt + drow[j] - (j1 - 1);
        over = over > 0 ? over : 0;
        dma16(dst + j * 1024, dvo[j] - over * ldb, so);
      }
    }
  };
  // LDS read offsets: score A operand (key row c16 of block kb, this wave's chunk c, lane group's 16 bytes) and the
  // transposed P.V operand (lane 4q+p of a group: key row 4g+q, columns 4p..4p+3 of the 16-column tile)
  const int sw_s[2] = {cr_swz(c16), cr_swz(16 + c16)};
  const int trow = 4 * g + (c16 >> 2), sw_t[2] = {cr_swz(trow), cr_swz(16 + trow)};
  const int tcol = (c16 & 3) >> 1, tb8 = (c16 & 1) * 8;

  f32x4 acc[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float m = -INFINITY, l = 0.f;
  f32x4* red = (f32x4*)(smem + 2 * TILEB);
  const int ntiles = (j1 - j0 + 31) >> 5;
  stage(0, j0);
  for (int t = 0; t < ntiles; ++t) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                       // tile t has landed for every wave; buffer (t+1)&1 is free
    if (t + 1 < ntiles) stage((t + 1) & 1, j0 + (t + 1) * 32);
    const char* sb = smem + (t & 1) * TILEB;
    f32x4 sp[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      sp[kb] = (f32x4){0.f, 0.f, 0.f, 0.f};
      const char* rowp = sb + (kb * 16 + c16) * RB;
#pragma unroll
      for (int c = 0; c < NCQ; ++c) {
        const int ch = ((dbase + 32 * c) >> 3) + g;
        const vec8 a = *(const vec8*)(rowp + ((ch ^ sw_s[kb]) << 4));
        sp[kb] = Mma16<T>::mma(a, qh[c], sp[kb]);
        sp[kb] = Mma16<T>::mma(a, ql[c], sp[kb]);
      }
      red[(wave * 2 + kb) * 64 + lane] = sp[kb];
    }
    __syncthreads();
    float s[2][4];
    const int jt = j0 + t * 32;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      f32x4 v = red[kb * 64 + lane];
#pragma unroll
      for (int w = 1; w < 4; ++w) v = v + red[(w * 2 + kb) * 64 + lane];
#pragma unroll
      for (int e = 0; e < 4; ++e) s[kb][e] = (jt + kb * 16 + 4 * g + e < j1) ? v[e] : -INFINITY;
    }
    float tm = fmaxf(fmaxf(fmaxf(s[0][0], s[0][1]), fmaxf(s[0][2], s[0][3])), fmaxf(fmaxf(s[1][0], s[1][1]), fmaxf(s[1][2], s[1][3])));
    tm = fmaxf(tm, __shfl_xor(tm, 16, 64));
    tm = fmaxf(tm, __shfl_xor(tm, 32, 64));
    const float mn = fmaxf(m, tm);                       // finite: every tile has at least one key of the slice
    const float f = __builtin_amdgcn_exp2f(m - mn);      // 2^-inf = 0 on the first tile
    float ps = 0.f;
    vec8 pb;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float pj = __builtin_amdgcn_exp2f(s[kb][e] - mn);
        ps += pj;
        pb[4 * kb + e] = (T)pj;
      }
    ps += __shfl_xor(ps, 16, 64);
    ps += __shfl_xor(ps, 32, 64);
    l = l * f + ps;
    m = mn;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int cb = ((dbase + 16 * nt) >> 3) + tcol;
      const i16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          (__attribute__((address_space(3))) i16x4*)(sb + trow * RB + ((cb ^ sw_t[0]) << 4) + tb8));
      const i16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          (__attribute__((address_space(3))) i16x4*)(sb + (16 + trow) * RB + ((cb ^ sw_t[1]) << 4) + tb8));
      typedef short i16x8 __attribute__((ext_vector_type(8)));
      const i16x8 a8 = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      acc[nt] = acc[nt] * f;
      acc[nt] = Mma16<T>::mma(__builtin_bit_cast(vec8, a8), pb, acc[nt]);
    }
  }
  if (c16 < R) {
    float* pr = prec + (long)c16 * (DK + 2);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) *(f32x4*)(pr + dbase + 16 * nt + 4 * g) = acc[nt];
    if (wave == 0 && g == 0) { pr[DK] = m; pr[DK + 1] = l; }
  }
}
// one maximum and one normaliser over all slices of all segments; out [B*R, nseg*Dk]
__global__ __launch_bounds__(256) void cross_rows_levels_combine_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                                        int R, int Dk, int nseg, int slices) {
  const int r = blockIdx.x, b = blockIdx.y;
  const int n = nseg * slices;
  const float* base = part + ((long)b * n * 16 + r) * (Dk + 2);
  const long rec = 16L * (Dk + 2);
  float m = -INFINITY;
  for (int s = 0; s < n; ++s) m = fmaxf(m, base[s * rec + Dk]);
  float l = 0.f;
  for (int s = 0; s < n; ++s)
    if (base[s * rec + Dk + 1] > 0.f) l += base[s * rec + Dk + 1] * exp2f(base[s * rec + Dk] - m);
  const float inv = 1.0f / l;
  for (int i = threadIdx.x; i < nseg * Dk; i += 256) {
    const int sg = i / Dk, d = i - sg * Dk;
    float a = 0.f;
    for (int s = sg * slices; s < (sg + 1) * slices; ++s)
      if (base[s * rec + Dk + 1] > 0.f) a += base[s * rec + d] * exp2f(base[s * rec + Dk] - m);
    out[((long)b * R + r) * ((long)nseg * Dk) + i] = a * inv;
  }
}
int cross_rows_levels_slices(int B, int nseg, int Lk) {
  int s = (512 + B * nseg - 1) / (B * nseg);          // >= 2 workgroups per CU's worth of slices at small batches
  const int most = (Lk + 63) / 64;                    // >= 64 keys per slice
  s = s < most ? s : most;
  s = s < 1 ? 1 : (s > 32 ? 32 : s);
  return s;
}
const char* cross_rows_levels_check(int x_dtype, int R, int nseg, int Lk, int Dk, long ldx) {
  if (x_dtype != AACLIP_F16 && x_dtype != AACLIP_BF16) return "cross_rows_levels: rows must be fp16 or bf16";
  if (R < 1 || R > 16) return "cross_rows_levels: 1..16 effective queries per image";
  if (nseg < 1 || nseg > 4) return "cross_rows_levels: 1..4 segments";
  if (Lk < 1) return "cross_rows_levels: no keys";
  if (Dk != 768 && Dk != 1024) return "cross_rows_levels: row width must be 768 or 1024";
  if (ldx < Dk || (ldx & 7)) return "cross_rows_levels: row stride must be >= the width and a multiple of 8 elements";
  return nullptr;
}
size_t cross_rows_levels_ws_bytes(int B, int nseg, int Lk, int Dk) {
  return (size_t)B * nseg * cross_rows_levels_slices(B, nseg, Lk) * 16 * (Dk + 2) * 4;
}
void launch_cross_rows_levels(int x_dtype, const float* qt, const void* const* x, int nseg, float* out, void* ws, int B,
                              int R, int rows_per_image, int row0, int Lk, int Dk, long ldx, hipStream_t s) {
  const int slices = cross_rows_levels_slices(B, nseg, Lk);
  CrossSegs segs;
  for (int i = 0; i < 4; ++i) segs.x[i] = x[i < nseg ? i : 0];
  dim3 g(nseg * slices, B), blk(256);
  float* part = (float*)ws;
#define CRL(T, DKV) hipLaunchKernelGGL((cross_rows_mfma_kernel<T, DKV>), g, blk, 0, s, segs, qt, part, R, rows_per_image, row0, Lk, (int)ldx, nseg, slices)
  if (x_dtype == AACLIP_F16) { if (Dk == 1024) CRL(f16, 1024); else CRL(f16, 768); }
  else { if (Dk == 1024) CRL(bf16, 1024); else CRL(bf16, 768); }
#undef CRL
  hipLaunchKernelGGL(cross_rows_levels_combine_kernel, dim3(R, B), dim3(256), 0, s, part, out, R, Dk, nseg, slices);
}

// head_expand: q [rows, D] fp32 -> qm [rows * H, D] of T, row (r, h) = q[r] * scale inside head h's column slice, zero
// outside -- the A operand of the effective-query product qt = qm . W_k (all heads in one GEMM).
template <typename T>
__global__ __launch_bounds__(256) void head_expand_kernel(const float* __restrict__ q, T* __restrict__ qm, int H, int D,
                                                          float scale) {
  const long r = blockIdx.x / H;
  const int h = blockIdx.x % H, hd = D / H;
  for (int d = threadIdx.x; d < D; d += 256)
    qm[(long)blockIdx.x * D + d] = from_float<T>((d / hd == h) ? q[r * D + d] * scale : 0.f);
}
void launch_head_expand(int dtype, const float* q, void* qm, long rows, int H, int D, float scale, hipStream_t s) {
  dim3 g((unsigned)(rows * H));
  if (dtype == AACLIP_F32) hipLaunchKernelGGL(head_expand_kernel<float>, g, dim3(256), 0, s, q, (float*)qm, H, D, scale);
  else if (dtype == AACLIP_F16) hipLaunchKernelGGL(head_expand_kernel<f16>, g, dim3(256), 0, s, q, (f16*)qm, H, D, scale);
  else hipLaunchKernelGGL(head_expand_kernel<bf16>, g, dim3(256), 0, s, q, (bf16*)qm, H, D, scale);
}
// head_diag: full [rows * H, D] fp32 -> ctx [rows, D]: ctx[r, h*hd + d] = full[(r, h), h*hd + d]
__global__ __launch_bounds__(256) void head_diag_kernel(const float* __restrict__ full, float* __restrict__ ctx, int H, int D) {
  const long r = blockIdx.x;
  const int hd = D / H;
  for (int d = threadIdx.x; d < D; d += 256) ctx[r * D + d] = full[(r * H + d / hd) * D + d];
}
void launch_head_diag(const float* full, float* ctx, long rows, int H, int D, hipStream_t s) {
  hipLaunchKernelGGL(head_diag_kernel, dim3((unsigned)rows), dim3(256), 0, s, full, ctx, H, D);
}

const char* small_attention_check(int nq, int Lk, int H, int hd) {
  if (nq < 1 || nq > SA_MAXQ) return "small_attention: 1..4 queries per image";
  if (Lk < 1 || Lk > SA_MAXK) return "small_attention: 1..8192 keys";
  if (H < 1 || hd < 4 || hd > SA_MAXHD || hd % 4) return "small_attention: head size must be a multiple of 4, <= 128";
  if ((256 / (hd / 4)) < 1) return "small_attention: head size too large";
  return nullptr;
}

void launch_small_attention(int kv_dtype, const float* q, const void* k, const void* v, float* out, int B, int nq, int Lk,
                            int H, int hd, float scale, hipStream_t s) {
  const int slices = 256 / (hd / 4);
  size_t a = (size_t)nq * Lk, b2 = (size_t)slices * nq * hd;
  const size_t shm = (a > b2 ? a : b2) * sizeof(float);
  dim3 g(H, B);
  if (kv_dtype == AACLIP_F16)
    hipLaunchKernelGGL(small_attention_kernel<f16>, g, dim3(256), shm, s, q, (const f16*)k, (const f16*)v, out, nq, Lk, H, hd, scale);
  else if (kv_dtype == AACLIP_BF16)
    hipLaunchKernelGGL(small_attention_kernel<bf16>, g, dim3(256), shm, s, q, (const bf16*)k, (const bf16*)v, out, nq, Lk, H, hd, scale);
  else
    hipLaunchKernelGGL(small_attention_kernel<float>, g, dim3(256), shm, s, q, (const float*)k, (const float*)v, out, nq, Lk, H, hd, scale);
}

// ---- residual_layernorm: out = LayerNorm(a + b) over D (multiple of 64, <= 4096), one wave per row, fp32
__global__ __launch_bounds__(256) void residual_layernorm_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                                 const float* __restrict__ w, const float* __restrict__ bias,
                                                                 float* __restrict__ out, long rows, int D, float eps) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* pa = a + row * D;
  const float* pb = b ? b + row * D : nullptr;
  float sum = 0.f;
  for (int i = lane; i < D; i += 64) sum += pa[i] + (pb ? pb[i] : 0.f);
  const float mean = wave_sum(sum) / (float)D;
  float var = 0.f;
  for (int i = lane; i < D; i += 64) { const float x = pa[i] + (pb ? pb[i] : 0.f) - mean; var = fmaf(x, x, var); }
  const float rstd = rsqrtf(wave_sum(var) / (float)D + eps);
  for (int i = lane; i < D; i += 64) out[row * D + i] = (pa[i] + (pb ? pb[i] : 0.f) - mean) * rstd * w[i] + bias[i];
}
void launch_residual_layernorm(const float* a, const float* b, const float* w, const float* bias, float* out, long rows,
                               int D, float eps, hipStream_t s) {
  hipLaunchKernelGGL(residual_layernorm_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, a, b, w, bias, out, rows, D, eps);
}

// ---- combine3
__global__ __launch_bounds__(256) void combine3_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                       const float* __restrict__ c, float wa, float wb, float wc,
                                                       float* __restrict__ out, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
    out[i] = wa * a[i] + wb * (b ? b[i] : 0.f) + wc * (c ? c[i] : 0.f);
}
void launch_combine3(const float* a, const float* b, const float* c, float wa, float wb, float wc, float* out, long n,
                     hipStream_t s) {
  long blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(combine3_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a, b, c, wa, wb, wc, out, n);
}

// ---- linear_smallk: y[r, n] = sum_{k < K} x[r, k] W[n, k] + bias[n], K <= 4; x fp32 [R, K]; y [R, N] of T
template <typename T>
__global__ __launch_bounds__(256) void linear_smallk_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                            const float* __restrict__ bias, T* __restrict__ y, long R,
                                                            int N, int K) {
  const long total = R * N;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long r = i / N;
    const int n = (int)(i - r * N);
    float acc = bias ? bias[n] : 0.f;
    for (int kk = 0; kk < K; ++kk) acc = fmaf(x[r * K + kk], W[n * K + kk], acc);
    y[i] = (T)acc;
  }
}
void launch_linear_smallk(int out_dtype, const float* x, const float* W, const float* bias, void* y, long R, int N, int K,
                          hipStream_t s) {
  long blocks = (R * N + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  dim3 g((unsigned)blocks);
  if (out_dtype == AACLIP_F16) hipLaunchKernelGGL(linear_smallk_kernel<f16>, g, dim3(256), 0, s, x, W, bias, (f16*)y, R, N, K);
  else if (out_dtype == AACLIP_BF16) hipLaunchKernelGGL(linear_smallk_kernel<bf16>, g, dim3(256), 0, s, x, W, bias, (bf16*)y, R, N, K);
  else hipLaunchKernelGGL(linear_smallk_kernel<float>, g, dim3(256), 0, s, x, W, bias, (float*)y, R, N, K);
}

// ---- drop_cls_rows: src [B, L, E] of T -> dst rows [b][row_off + t - 1], t = 1..L-1, of [B, rows_per_image, E]
template <typename T>
__global__ __launch_bounds__(256) void drop_cls_rows_kernel(const T* __restrict__ src, T* __restrict__ dst, int B, int L,
                                                            int E, int rows_per_image, int row_off) {
  const int e8 = E / 8;                        // 16-byte units for 16-bit T, 32-byte for fp32 (two loads)
  const long total = (long)B * (L - 1) * e8;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long r = i / e8;
    const int c = (int)(i - r * e8) * 8;
    const long b = r / (L - 1), t = r - b * (L - 1);
    const T* s = src + ((b * L) + 1 + t) * E + c;
    T* d = dst + ((b * rows_per_image) + row_off + t) * E + c;
#pragma unroll
    for (int j = 0; j < 8; ++j) d[j] = s[j];
  }
}
void launch_drop_cls_rows(int dtype, const void* src, void* dst, int B, int L, int E, int rows_per_image, int row_off,
                          hipStream_t s) {
  long blocks = ((long)B * (L - 1) * (E / 8) + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  dim3 g((unsigned)blocks);
  if (dtype == AACLIP_F16) hipLaunchKernelGGL(drop_cls_rows_kernel<f16>, g, dim3(256), 0, s, (const f16*)src, (f16*)dst, B, L, E, rows_per_image, row_off);
  else if (dtype == AACLIP_BF16) hipLaunchKernelGGL(drop_cls_rows_kernel<bf16>, g, dim3(256), 0, s, (const bf16*)src, (bf16*)dst, B, L, E, rows_per_image, row_off);
  else hipLaunchKernelGGL(drop_cls_rows_kernel<float>, g, dim3(256), 0, s, (const float*)src, (float*)dst, B, L, E, rows_per_image, row_off);
}

// ---- IQM maps.  iqm_scores: one wave per patch row: sigmoid(cos(f, q1) - cos(f, q0)) with F.cosine_similarity's
// eps (1e-8 on the product of norms).  seg [B, P, E] fp32, queries [B, 2, E] fp32 -> grid [B, P]
template <int NCH>
__global__ __launch_bounds__(256) void iqm_scores_kernel(const float* __restrict__ seg, const float* __restrict__ qv,
                                                         float* __restrict__ out, int B, int P) {
  constexpr int E = NCH * 256;
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (long)B * P) return;
  const long b = row / P;
  const float* f = seg + row * E;
  const float* q0 = qv + b * 2 * E;
  const float* q1 = q0 + E;
  float ff = 0.f, d0 = 0.f, d1 = 0.f, n0 = 0.f, n1 = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int e0 = (c * 64 + lane) * 4;
    const f32x4 fv = *(const f32x4*)(f + e0), a = *(const f32x4*)(q0 + e0), bb = *(const f32x4*)(q1 + e0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      ff = fmaf(fv[j], fv[j], ff);
      d0 = fmaf(fv[j], a[j], d0); n0 = fmaf(a[j], a[j], n0);
      d1 = fmaf(fv[j], bb[j], d1); n1 = fmaf(bb[j], bb[j], n1);
    }
  }
  ff = wave_sum(ff); d0 = wave_sum(d0); d1 = wave_sum(d1); n0 = wave_sum(n0); n1 = wave_sum(n1);
  if (lane == 0) {
    // torch: x.y / max(|x| |y|, eps) computed as x.y / sqrt(clamp(|x|^2 |y|^2, eps^2))
    const float c0 = d0 / sqrtf(fmaxf(ff * n0, 1e-16f)), c1 = d1 / sqrtf(fmaxf(ff * n1, 1e-16f));
    out[row] = 1.0f / (1.0f + expf(-(c1 - c0)));
  }
}
void launch_iqm_scores(const float* seg, const float* q, float* grid, int B, int P, int E, hipStream_t s) {
  dim3 g((unsigned)(((long)B * P + 3) / 4));
  switch (E / 256) {
    case 1: hipLaunchKernelGGL(iqm_scores_kernel<1>, g, dim3(256), 0, s, seg, q, grid, B, P); break;
    case 2: hipLaunchKernelGGL(iqm_scores_kernel<2>, g, dim3(256), 0, s, seg, q, grid, B, P); break;
    case 3: hipLaunchKernelGGL(iqm_scores_kernel<3>, g, dim3(256), 0, s, seg, q, grid, B, P); break;
    case 4: hipLaunchKernelGGL(iqm_scores_kernel<4>, g, dim3(256), 0, s, seg, q, grid, B, P); break;
  }
}

// grids [NL][B, g, g] -> out [B, S, S] = w_base * base + w_iqm * sum_l bilinear_{align_corners=False}(grid_l)
constexpr int IQ_MAXG = 40;
__global__ __launch_bounds__(256) void iqm_upsample_kernel(const float* __restrict__ grids, const float* __restrict__ base,
                                                           float* __restrict__ out, int B, int g, int S, int NL,
                                                           float w_base, float w_iqm, int rows_per_band) {
  __shared__ float m[4][IQ_MAXG * IQ_MAXG];
  const int b = blockIdx.y, tid = threadIdx.x;
  for (int l = 0; l < NL; ++l)
    for (int i = tid; i < g * g; i += 256) m[l][i] = grids[((long)l * B + b) * g * g + i];
  __syncthreads();
  const float scale = (float)g / (float)S;
  const int y_begin = blockIdx.x * rows_per_band;
  int y_end = y_begin + rows_per_band;
  if (y_end > S) y_end = S;
  for (long i = (long)y_begin * S + tid; i < (long)y_end * S; i += 256) {
    const int y = i / S, x = i - (long)y * S;
    float sy = scale * ((float)y + 0.5f) - 0.5f, sx = scale * ((float)x + 0.5f) - 0.5f;
    sy = sy < 0.f ? 0.f : sy;
    sx = sx < 0.f ? 0.f : sx;
    const int y0 = (int)sy, x0 = (int)sx;
    const int y1 = y0 + (y0 < g - 1 ? 1 : 0), x1 = x0 + (x0 < g - 1 ? 1 : 0);
    const float ly1 = sy - y0, ly0 = 1.0f - ly1, lx1 = sx - x0, lx0 = 1.0f - lx1;
    float acc = 0.f;
    for (int l = 0; l < NL; ++l) {
      const float* p = m[l];
      const float v = ly0 * (lx0 * p[y0 * g + x0] + lx1 * p[y0 * g + x1]) + ly1 * (lx0 * p[y1 * g + x0] + lx1 * p[y1 * g + x1]);
      acc = (l == 0) ? v : acc + v;
    }
    const long o = (long)b * S * S + i;
    out[o] = base ? w_base * base[o] + w_iqm * acc : w_iqm * acc;
  }
}
void launch_iqm_upsample(const float* grids, const float* base, float* out, int B, int g, int S, int NL, float w_base,
                         float w_iqm, hipStream_t s) {
  const int bands = 14;
  const int rpb = (S + bands - 1) / bands;
  hipLaunchKernelGGL(iqm_upsample_kernel, dim3(bands, B), dim3(256), 0, s, grids, base, out, B, g, S, NL, w_base, w_iqm, rpb);
}

}  // namespace aaclip
