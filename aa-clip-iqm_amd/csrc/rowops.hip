// Row-wise (HBM-bound) kernels of the AA-CLIP path: one 64-lane wave owns one
// token row, keeps it in registers (16-byte coalesced loads), and reduces with
// cross-lane shuffles.  Row widths are 256 / 768 / 1024 (NCH = D/256 float4
// chunks per lane).
#include "common.h"
#include "kernels.h"

namespace aaclip {

const char* row_width_check(int D) {
  if (D == 256 || D == 512 || D == 768 || D == 1024) return nullptr;
  return "row width must be 256, 512, 768 or 1024";
}

template <int NCH>
AACLIP_DEV void load_row(const float* p, int lane, f32x4 (&v)[NCH]) {
#pragma unroll
  for (int c = 0; c < NCH; ++c) v[c] = *(const f32x4*)(p + (c * 64 + lane) * 4);
}

template <typename T>
AACLIP_DEV void store4(T* p, f32x4 v);
template <>
AACLIP_DEV void store4<float>(float* p, f32x4 v) { *(f32x4*)p = v; }
template <>
AACLIP_DEV void store4<f16>(f16* p, f32x4 v) {
  f16x4 o = {(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
  *(f16x4*)p = o;
}
template <>
AACLIP_DEV void store4<bf16>(bf16* p, f32x4 v) {
  bf16x4 o = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
  *(bf16x4*)p = o;
}

// LayerNorm over the last dim, reference model/transformer.py:37-43 (eps 1e-5),
// two-pass statistics in registers; out may alias x when T == float.
// split8 row (AACLIP_F16X2, common.h): 4 values at column `col` of a row of logical width `width` starting at `row`:
// hi plane (fp16), then the lo8 and hi8 planes (e4m3, one byte per element)
// hi8 = false: the hi8 plane is not written (its only reader is the Ah8 . Wl8 correction tile, which a product whose
// weight is exact in fp16 skips)
AACLIP_DEV void store4_split(f16* row, int col, f32x4 v, int width, bool hi8 = true) {
  const float vv[4] = {v[0], v[1], v[2], v[3]};
  f16x4 hi;
  uint32_t l8, h8;
  split8x4(vv, hi, l8, h8);
  *(f16x4*)(row + col) = hi;
  uint8_t* p8 = (uint8_t*)(row + width);
  *(uint32_t*)(p8 + col) = l8;
  if (hi8) *(uint32_t*)(p8 + width + col) = h8;
}

template <typename T, int NCH, bool SPLIT = false>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* x, const float* __restrict__ w,
                                                        const float* __restrict__ b, T* out, long rows, float eps,
                                                        bool hi8 = true) {
  constexpr int D = NCH * 256;
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  f32x4 v[NCH];
  load_row<NCH>(x + row * D, lane, v);
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) s += (v[c][0] + v[c][1]) + (v[c][2] + v[c][3]);
  const float mean = wave_sum(s) * (1.0f / D);
  float q = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float d = v[c][e] - mean;
      q = fmaf(d, d, q);
    }
  const float rstd = rsqrtf(wave_sum(q) * (1.0f / D) + eps);
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int col = (c * 64 + lane) * 4;
    f32x4 g = *(const f32x4*)(w + col), bb = *(const f32x4*)(b + col), y;
#pragma unroll
    for (int e = 0; e < 4; ++e) y[e] = (v[c][e] - mean) * rstd * g[e] + bb[e];
    if constexpr (SPLIT) store4_split(out + row * 2 * D, col, y, D, hi8);
    else store4<T>(out + row * D + col, y);
  }
}

template <typename T>
static void ln_dispatch(const float* x, const float* w, const float* b, T* out, long rows, int D, float eps,
                        hipStream_t s) {
  dim3 g((unsigned)((rows + 3) / 4));
  switch (D / 256) {
    case 1: hipLaunchKernelGGL((layernorm_kernel<T, 1>), g, dim3(256), 0, s, x, w, b, out, rows, eps); break;
    case 2: hipLaunchKernelGGL((layernorm_kernel<T, 2>), g, dim3(256), 0, s, x, w, b, out, rows, eps); break;
    case 3: hipLaunchKernelGGL((layernorm_kernel<T, 3>), g, dim3(256), 0, s, x, w, b, out, rows, eps); break;
    case 4: hipLaunchKernelGGL((layernorm_kernel<T, 4>), g, dim3(256), 0, s, x, w, b, out, rows, eps); break;
  }
}

static void ln_dispatch_split(const float* x, const float* w, const float* b, f16* out, long rows, int D, float eps,
                              hipStream_t s, bool hi8) {
  dim3 g((unsigned)((rows + 3) / 4));
  switch (D / 256) {
    case 1: hipLaunchKernelGGL((layernorm_kernel<f16, 1, true>), g, dim3(256), 0, s, x, w, b, out, rows, eps, hi8); break;
    case 2: hipLaunchKernelGGL((layernorm_kernel<f16, 2, true>), g, dim3(256), 0, s, x, w, b, out, rows, eps, hi8); break;
    case 3: hipLaunchKernelGGL((layernorm_kernel<f16, 3, true>), g, dim3(256), 0, s, x, w, b, out, rows, eps, hi8); break;
    case 4: hipLaunchKernelGGL((layernorm_kernel<f16, 4, true>), g, dim3(256), 0, s, x, w, b, out, rows, eps, hi8); break;
  }
}

void launch_layernorm(int out_dtype, const float* x, const float* w, const float* b, void* out, long rows, int D,
                      float eps, hipStream_t s, bool hi8) {
  if (out_dtype == AACLIP_F16X2) ln_dispatch_split(x, w, b, (f16*)out, rows, D, eps, s, hi8);
  else if (out_dtype == AACLIP_F32) ln_dispatch<float>(x, w, b, (float*)out, rows, D, eps, s);
  else if (out_dtype == AACLIP_F16) ln_dispatch<f16>(x, w, b, (f16*)out, rows, D, eps, s);
  else ln_dispatch<bf16>(x, w, b, (bf16*)out, rows, D, eps, s);
}

// Residual adapter mix, reference model/adapter.py:165-170 (and :290-295):
//   a <- a * |x| / |a| per token (no eps), x <- w*a + (1-w)*x, in place on x.
// With out16 / rowab (LayerNorm folding, capi.hip): also writes the new rows in the compute dtype and the
// (rstd, -mean*rstd) pair of those rounded values, which the next block's QKV product folds ln_1 with.
template <int NCH, typename T>
__global__ __launch_bounds__(256) void adapter_mix_kernel(float* x, const float* __restrict__ a, long rows,
                                                          float weight, T* out16, float* rowab, float eps) {
  constexpr int D = NCH * 256;
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  f32x4 xv[NCH], av[NCH];
  load_row<NCH>(x + row * D, lane, xv);
  load_row<NCH>(a + row * D, lane, av);
  float sx = 0.f, sa = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      sx = fmaf(xv[c][e], xv[c][e], sx);
      sa = fmaf(av[c][e], av[c][e], sa);
    }
  const float nx = sqrtf(wave_sum(sx)), na = sqrtf(wave_sum(sa));
  float ps = 0.f, pq = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    f32x4 y;
#pragma unroll
    for (int e = 0; e < 4; ++e) y[e] = weight * (av[c][e] * nx / na) + (1.0f - weight) * xv[c][e];
    *(f32x4*)(x + row * D + (c * 64 + lane) * 4) = y;
    if (out16) {
      store4<T>(out16 + row * D + (c * 64 + lane) * 4, y);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float r = to_float<T>(from_float<T>(y[e]));
        ps += r;
        pq = fmaf(r, r, pq);
      }
    }
  }
  if (out16) {
    ps = wave_sum(ps);
    pq = wave_sum(pq);
    const float mean = ps * (1.0f / D);
    float var = pq * (1.0f / D) - mean * mean;
    var = var > 0.f ? var : 0.f;
    const float rstd = rsqrtf(var + eps);
    if (lane == 0) {
      const f32x2 o = {rstd, -mean * rstd};
      *(f32x2*)(rowab + 2 * row) = o;
    }
  }
}

template <typename T>
static void launch_adapter_mix_t(float* x, const float* a, long rows, int D, float weight, T* out16, float* rowab,
                                 hipStream_t s) {
  dim3 g((unsigned)((rows + 3) / 4));
  switch (D / 256) {
    case 1: hipLaunchKernelGGL((adapter_mix_kernel<1, T>), g, dim3(256), 0, s, x, a, rows, weight, out16, rowab, 1e-5f); break;
    case 2: hipLaunchKernelGGL((adapter_mix_kernel<2, T>), g, dim3(256), 0, s, x, a, rows, weight, out16, rowab, 1e-5f); break;
    case 3: hipLaunchKernelGGL((adapter_mix_kernel<3, T>), g, dim3(256), 0, s, x, a, rows, weight, out16, rowab, 1e-5f); break;
    case 4: hipLaunchKernelGGL((adapter_mix_kernel<4, T>), g, dim3(256), 0, s, x, a, rows, weight, out16, rowab, 1e-5f); break;
  }
}
void launch_adapter_mix(float* x, const float* a, long rows, int D, float weight, hipStream_t s) {
  launch_adapter_mix_t<f16>(x, a, rows, D, weight, nullptr, nullptr, s);
}
void launch_adapter_mix_fold(int dtype, float* x, const float* a, long rows, int D, float weight, void* out16,
                             float* rowab, hipStream_t s) {
  if (dtype == AACLIP_BF16) launch_adapter_mix_t<bf16>(x, a, rows, D, weight, (bf16*)out16, rowab, s);
  else launch_adapter_mix_t<f16>(x, a, rows, D, weight, (f16*)out16, rowab, s);
}

// Unfold non-overlapping ps x ps patches of an NCHW fp32 image into GEMM rows:
// cols[(b*g*g + py*g + px)][c*ps*ps + ky*ps + kx], zero padded to Kpad
// (conv1 of reference model/transformer.py:359-365 as a GEMM).
template <typename T, bool SPLIT = false>
__global__ __launch_bounds__(256) void im2col_kernel(const float* __restrict__ img, T* __restrict__ cols, int B, int C,
                                                     int H, int W, int ps, int Kpad) {
  const int g = H / ps, gw = W / ps;
  const long row = blockIdx.x;  // one patch per block
  const int b = row / (g * gw), pi = row % (g * gw), py = pi / gw, px = pi % gw;
  const int K = C * ps * ps;
  for (int k = threadIdx.x; k < Kpad; k += blockDim.x) {
    float v = 0.f;
    if (k < K) {
      int c = k / (ps * ps), rem = k % (ps * ps), ky = rem / ps, kx = rem % ps;
      v = img[(((long)b * C + c) * H + py * ps + ky) * W + px * ps + kx];
    }
    if constexpr (SPLIT) {   // split8 row: hi plane, lo8 plane, hi8 plane
      f16* r = cols + row * 2 * Kpad;
      const f16 hi = (f16)v;
      r[k] = hi;
      uint8_t* p8 = (uint8_t*)(r + Kpad);
      p8[k] = (uint8_t)pack_e4m3x4<SPLIT8_ACT_LO_EXP>(v - (float)hi, 0.f, 0.f, 0.f);
      p8[Kpad + k] = (uint8_t)pack_e4m3x4<SPLIT8_ACT_HI_EXP>(v, 0.f, 0.f, 0.f);
    } else {
      cols[row * Kpad + k] = from_float<T>(v);
    }
  }
}

void launch_im2col(int dtype, const float* img, void* cols, int B, int C, int H, int W, int ps, int Kpad,
                   hipStream_t s) {
  dim3 g((unsigned)((long)B * (H / ps) * (W / ps)));
  if (dtype == AACLIP_F16X2)
    hipLaunchKernelGGL((im2col_kernel<f16, true>), g, dim3(256), 0, s, img, (f16*)cols, B, C, H, W, ps, Kpad);
  else if (dtype == AACLIP_F32)
    hipLaunchKernelGGL(im2col_kernel<float>, g, dim3(256), 0, s, img, (float*)cols, B, C, H, W, ps, Kpad);
  else if (dtype == AACLIP_F16)
    hipLaunchKernelGGL(im2col_kernel<f16>, g, dim3(256), 0, s, img, (f16*)cols, B, C, H, W, ps, Kpad);
  else
    hipLaunchKernelGGL(im2col_kernel<bf16>, g, dim3(256), 0, s, img, (bf16*)cols, B, C, H, W, ps, Kpad);
}

// x[b*L + 0] = class_embedding + positional_embedding[0]  (reference model/adapter.py:143-153)
__global__ void cls_rows_kernel(float* x, const float* __restrict__ cls, const float* __restrict__ pos, int L, int D) {
  const int b = blockIdx.x;
  for (int d = threadIdx.x; d < D; d += blockDim.x) x[(long)b * L * D + d] = cls[d] + pos[d];
}
void launch_cls_rows(float* x, const float* cls, const float* pos, int B, int L, int D, hipStream_t s) {
  hipLaunchKernelGGL(cls_rows_kernel, dim3(B), dim3(256), 0, s, x, cls, pos, L, D);
}

// x[i*T + t] = token_embedding[tokens[i,t]] + positional_embedding[t]  (reference model/adapter.py:277-281)
__global__ void embed_text_kernel(const int32_t* __restrict__ tokens, const float* __restrict__ table,
                                  const float* __restrict__ pos, float* x, int T, int D, int vocab) {
  const long row = blockIdx.x;
  int id = tokens[row];
  id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
  const int t = row % T;
  for (int d = threadIdx.x * 4; d < D; d += blockDim.x * 4) {
    f32x4 e = *(const f32x4*)(table + (long)id * D + d), p = *(const f32x4*)(pos + (long)t * D + d);
    *(f32x4*)(x + row * D + d) = e + p;
  }
}
void launch_embed_text(const int32_t* tokens, const float* table, const float* pos, float* x, int n, int T, int D,
                       int vocab, hipStream_t s) {
  hipLaunchKernelGGL(embed_text_kernel, dim3(n * T), dim3(64), 0, s, tokens, table, pos, x, T, D, vocab);
}

// Row gather: mode 0 = row at argmax(tokens[i,:]) (first maximum, = EOT:
// reference model/adapter.py:299), mode 1 = row 0 of each sequence (CLS).
template <typename T>
__global__ void gather_rows_kernel(const T* __restrict__ src, T* __restrict__ dst, const int32_t* __restrict__ tokens,
                                   int Tn, int D, int mode) {
  const int i = blockIdx.x;
  __shared__ int pick;
  if (threadIdx.x == 0) {
    int best = 0;
    if (mode == 0) {
      int bv = tokens[(long)i * Tn];
      for (int t = 1; t < Tn; ++t) {
        int v = tokens[(long)i * Tn + t];
        if (v > bv) { bv = v; best = t; }
      }
    }
    pick = best;
  }
  __syncthreads();
  const T* s = src + ((long)i * Tn + pick) * D;
  for (int d = threadIdx.x; d < D; d += blockDim.x) dst[(long)i * D + d] = s[d];
}
void launch_gather_rows(int dtype, const void* src, void* dst, const int32_t* tokens, int n, int T, int D, int mode,
                        hipStream_t s) {
  if (dtype == AACLIP_F16X2)   // a split row is 2D halves = D words
    hipLaunchKernelGGL(gather_rows_kernel<float>, dim3(n), dim3(256), 0, s, (const float*)src, (float*)dst, tokens, T,
                       D, mode);
  else if (dtype == AACLIP_F32)
    hipLaunchKernelGGL(gather_rows_kernel<float>, dim3(n), dim3(256), 0, s, (const float*)src, (float*)dst, tokens, T,
                       D, mode);
  else
    hipLaunchKernelGGL(gather_rows_kernel<uint16_t>, dim3(n), dim3(256), 0, s, (const uint16_t*)src, (uint16_t*)dst,
                       tokens, T, D, mode);
}

// F.normalize(dim=-1) (eps 1e-12) of rows [B*L, E], dropping the first `skip`
// rows of every image (CLS): dst [B, L-skip, E]   (reference model/adapter.py:182)
template <int NCH>
__global__ __launch_bounds__(256) void normalize_rows_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                             int B, int L, int skip) {
  constexpr int E = NCH * 256;
  const int lane = threadIdx.x & 63;
  const long orow = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int P = L - skip;
  if (orow >= (long)B * P) return;
  const long b = orow / P, pi = orow - b * P;
  const float* sp = src + (b * L + skip + pi) * E;
  f32x4 v[NCH];
  load_row<NCH>(sp, lane, v);
  float q = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int e = 0; e < 4; ++e) q = fmaf(v[c][e], v[c][e], q);
  const float n = fmaxf(sqrtf(wave_sum(q)), 1e-12f);
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    f32x4 y;
#pragma unroll
    for (int e = 0; e < 4; ++e) y[e] = v[c][e] / n;
    *(f32x4*)(dst + orow * E + (c * 64 + lane) * 4) = y;
  }
}
void launch_normalize_rows(const float* src, float* dst, int B, int L, int skip, int E, hipStream_t s) {
  dim3 g((unsigned)(((long)B * (L - skip) + 3) / 4));
  switch (E / 256) {
    case 1: hipLaunchKernelGGL(normalize_rows_kernel<1>, g, dim3(256), 0, s, src, dst, B, L, skip); break;
    case 2: hipLaunchKernelGGL(normalize_rows_kernel<2>, g, dim3(256), 0, s, src, dst, B, L, skip); break;
    case 3: hipLaunchKernelGGL(normalize_rows_kernel<3>, g, dim3(256), 0, s, src, dst, B, L, skip); break;
    case 4: hipLaunchKernelGGL(normalize_rows_kernel<4>, g, dim3(256), 0, s, src, dst, B, L, skip); break;
  }
}

// det token (reference model/adapter.py:183-184): mean over the patch rows of the L2-normalised det projections.
// One pass over src: a workgroup takes one image and one slice of its rows, a wave one row at a time (row in
// registers: norm by wave reduction, accumulate v / |v|); the four waves are added in LDS and the slice sum is
// written to part[b][slice][E]; a second tiny kernel adds the slices in a fixed order (deterministic).
template <int NCH>
__global__ __launch_bounds__(256) void det_partial_kernel(const float* __restrict__ src, float* __restrict__ part, int L,
                                                          int skip, int rows_per_slice) {
  constexpr int E = NCH * 256;
  __shared__ float red[4][E];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.y, sl = blockIdx.x;
  const int t0 = skip + sl * rows_per_slice;
  const int t1 = min(L, t0 + rows_per_slice);
  f32x4 acc[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int t = t0 + wave; t < t1; t += 4) {
    f32x4 v[NCH];
    load_row<NCH>(src + ((long)b * L + t) * E, lane, v);
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int e = 0; e < 4; ++e) q = fmaf(v[c][e], v[c][e], q);
    const float inv = 1.0f / fmaxf(sqrtf(wave_sum(q)), 1e-12f);   // F.normalize eps
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[c][e] = fmaf(v[c][e], inv, acc[c][e]);
  }
#pragma unroll
  for (int c = 0; c < NCH; ++c) *(f32x4*)&red[wave][(c * 64 + lane) * 4] = acc[c];
  __syncthreads();
  for (int i = threadIdx.x; i < E; i += 256)
    part[((long)b * gridDim.x + sl) * E + i] = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
}
__global__ void det_finish_kernel(const float* __restrict__ part, float* __restrict__ dst, int slices, int E, float inv_n) {
  const int b = blockIdx.x;
  for (int i = threadIdx.x; i < E; i += blockDim.x) {
    float a = 0.f;
    for (int s = 0; s < slices; ++s) a += part[((long)b * slices + s) * E + i];
    dst[(long)b * E + i] = a * inv_n;
  }
}
// scratch holds B * slices * E floats.  The slice count must NOT depend on B: the order in which an image's patches
// are added has to be the same in every batch, or an image's det token changes in the last bit with the batch it
// happens to share (found by the B = 128 case of tests/test_gpu_configs.py: 2048 / B slices gave 16 instead of 32).
void launch_det_mean(const float* src, float* scratch, size_t scratch_floats, float* dst, int B, int L, int skip, int E,
                     hipStream_t s) {
  const int n = L - skip;
  int slices = 32;
  if (slices > n) slices = n;
  const size_t fit = scratch_floats / ((size_t)B * E);
  if ((size_t)slices > fit) slices = fit < 1 ? 1 : (int)fit;
  const int rps = (n + slices - 1) / slices;
  slices = (n + rps - 1) / rps;
  dim3 g(slices, B);
  switch (E / 256) {
    case 1: hipLaunchKernelGGL(det_partial_kernel<1>, g, dim3(256), 0, s, src, scratch, L, skip, rps); break;
    case 2: hipLaunchKernelGGL(det_partial_kernel<2>, g, dim3(256), 0, s, src, scratch, L, skip, rps); break;
    case 3: hipLaunchKernelGGL(det_partial_kernel<3>, g, dim3(256), 0, s, src, scratch, L, skip, rps); break;
    case 4: hipLaunchKernelGGL(det_partial_kernel<4>, g, dim3(256), 0, s, src, scratch, L, skip, rps); break;
  }
  hipLaunchKernelGGL(det_finish_kernel, dim3(B), dim3(256), 0, s, scratch, dst, slices, E, 1.0f / (float)n);
}

}  // namespace aaclip

namespace aaclip {
// fp32 -> compute dtype copy (adapter GEMM input is the raw residual stream)
template <typename T>
__global__ __launch_bounds__(256) void cast_rows_kernel(const float* __restrict__ src, T* __restrict__ dst, long n4) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256)
    store4<T>(dst + i * 4, *(const f32x4*)(src + i * 4));
}
// fp32 rows [rows, D] -> split8 rows [rows, 2D halves]
__global__ __launch_bounds__(256) void split_rows_kernel(const float* __restrict__ src, f16* __restrict__ dst, long n4,
                                                         int d4, bool hi8) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const long r = i / d4;
    const int c = (int)(i - r * d4) * 4;
    store4_split(dst + r * 8 * d4, c, *(const f32x4*)(src + i * 4), 4 * d4, hi8);
  }
}
void launch_split_rows(const float* src, void* dst, long rows, int D, hipStream_t s, bool hi8) {
  const long n4 = rows * D / 4;
  long blocks = (n4 + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(split_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, s, src, (f16*)dst, n4, D / 4, hi8);
}
void launch_cast_rows(int dtype, const float* src, void* dst, long n, hipStream_t s) {
  const long n4 = n / 4;
  long blocks = (n4 + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  if (dtype == AACLIP_F16)
    hipLaunchKernelGGL(cast_rows_kernel<f16>, dim3((unsigned)blocks), dim3(256), 0, s, src, (f16*)dst, n4);
  else if (dtype == AACLIP_BF16)
    hipLaunchKernelGGL(cast_rows_kernel<bf16>, dim3((unsigned)blocks), dim3(256), 0, s, src, (bf16*)dst, n4);
  else
    hipLaunchKernelGGL(cast_rows_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, s, src, (float*)dst, n4);
}
}  // namespace aaclip

namespace aaclip {
// ---- "surgery" (V-V) attention over the batch axis, reference model/transformer.py:102-152 as it
// runs after DAPM_replace (the module unpacks the LND stream as [B,N,C]).
// The attention kernels walk rows batch*L + pos; here the "sequence" is the image index b and the
// "batch" is the token position l, so the value projection v [B*L, D] is regrouped into a packed
// q|k|v buffer with rows l*B + b: q = v * scale (the score scale the QKV epilogue would have folded
// into q), k = v, v = v.  4 elements per thread.
template <typename T>
__global__ __launch_bounds__(256) void vv_spread_kernel(const T* __restrict__ v, T* __restrict__ qkv, int B, int L,
                                                        int D, float scale) {
  const long n4 = (long)B * L * D / 4;
  const int d4 = D / 4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const long r = i / d4;
    const int c = (int)(i % d4) * 4;
    const int b = (int)(r / L), l = (int)(r % L);
    const T* src = v + r * D + c;
    T* dst = qkv + ((long)l * B + b) * 3 * D + c;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const T val = src[j];
      dst[j] = from_float<T>(to_float<T>(val) * scale);
      dst[D + j] = val;
      dst[2 * D + j] = val;
    }
  }
}
// split fp16: v rows [B*L][hi D | lo D] -> qkv rows l*B + b of [q k v hi | q k v lo]; q = v * scale, split again
__global__ __launch_bounds__(256) void vv_spread_split_kernel(const f16* __restrict__ v, f16* __restrict__ qkv, int B,
                                                              int L, int D, float scale) {
  const long n4 = (long)B * L * D / 4;
  const int d4 = D / 4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const long r = i / d4;
    const int c = (int)(i % d4) * 4;
    const int b = (int)(r / L), l = (int)(r % L);
    const f16* src = v + r * 2 * D + c;
    f16* dst = qkv + ((long)l * B + b) * 6 * D + c;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f16 hi = src[j], lo = src[D + j];
      f16 qh, ql;
      split16(((float)hi + (float)lo) * scale, qh, ql);
      dst[j] = qh;
      dst[3 * D + j] = ql;
      dst[D + j] = hi;
      dst[4 * D + j] = lo;
      dst[2 * D + j] = hi;
      dst[5 * D + j] = lo;
    }
  }
}
// ctx rows l*B + b -> rows b*L + l
template <typename T>
__global__ __launch_bounds__(256) void vv_regroup_kernel(const T* __restrict__ src, T* __restrict__ dst, int B, int L,
                                                         int D) {
  const long n4 = (long)B * L * D / 4;
  const int d4 = D / 4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const long r = i / d4;
    const int c = (int)(i % d4) * 4;
    const int b = (int)(r / L), l = (int)(r % L);
    const T* s4 = src + ((long)l * B + b) * D + c;
    T* d4p = dst + r * D + c;
#pragma unroll
    for (int j = 0; j < 4; ++j) d4p[j] = s4[j];
  }
}
static unsigned vv_blocks(long n4) {
  long blocks = (n4 + 255) / 256;
  return (unsigned)(blocks > 16384 ? 16384 : blocks);
}
void launch_vv_spread(int dtype, const void* v, void* qkv, int B, int L, int D, float scale, hipStream_t s) {
  const unsigned g = vv_blocks((long)B * L * D / 4);
  if (dtype == AACLIP_F16X2)
    hipLaunchKernelGGL(vv_spread_split_kernel, dim3(g), dim3(256), 0, s, (const f16*)v, (f16*)qkv, B, L, D, scale);
  else if (dtype == AACLIP_F16)
    hipLaunchKernelGGL(vv_spread_kernel<f16>, dim3(g), dim3(256), 0, s, (const f16*)v, (f16*)qkv, B, L, D, scale);
  else if (dtype == AACLIP_BF16)
    hipLaunchKernelGGL(vv_spread_kernel<bf16>, dim3(g), dim3(256), 0, s, (const bf16*)v, (bf16*)qkv, B, L, D, scale);
  else
    hipLaunchKernelGGL(vv_spread_kernel<float>, dim3(g), dim3(256), 0, s, (const float*)v, (float*)qkv, B, L, D, scale);
}
void launch_vv_regroup(int dtype, const void* src, void* dst, int B, int L, int D, hipStream_t s) {
  if (dtype == AACLIP_F16X2) {   // split rows: 2D halves each
    hipLaunchKernelGGL(vv_regroup_kernel<f16>, dim3(vv_blocks((long)B * L * D / 2)), dim3(256), 0, s, (const f16*)src,
                       (f16*)dst, B, L, 2 * D);
    return;
  }
  const unsigned g = vv_blocks((long)B * L * D / 4);
  if (dtype == AACLIP_F16)
    hipLaunchKernelGGL(vv_regroup_kernel<f16>, dim3(g), dim3(256), 0, s, (const f16*)src, (f16*)dst, B, L, D);
  else if (dtype == AACLIP_BF16)
    hipLaunchKernelGGL(vv_regroup_kernel<bf16>, dim3(g), dim3(256), 0, s, (const bf16*)src, (bf16*)dst, B, L, D);
  else
    hipLaunchKernelGGL(vv_regroup_kernel<float>, dim3(g), dim3(256), 0, s, (const float*)src, (float*)dst, B, L, D);
}
}  // namespace aaclip

namespace aaclip {
// LayerNorm folding: per-row partial (sum, sum of squares) of the 16-bit residual rows, one pair per 64-column
// slice in a fixed order (deterministic), -> (a, b) = (rstd, -mean * rstd) with the biased variance and eps
// of nn.LayerNorm (reference model/transformer.py:37-43).
// 16 lanes per row (one per 64-column slice: coalesced 128-byte lines instead of one strided 8-byte load per lane and
// slice), summed over the lane row by DPP rotations -- a fixed order, so results do not depend on the batch.
AACLIP_DEV float row16_total(float x) {
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x128, 0xF, 0xF, false));
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x124, 0xF, 0xF, false));
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x122, 0xF, 0xF, false));
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x121, 0xF, 0xF, false));
  return x;
}
__global__ __launch_bounds__(256) void ln_stats_finalize_kernel(const float* __restrict__ partials, float* __restrict__ ab,
                                                                long rows, int slots, float inv_d, float eps) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  const long row = t >> 4;
  const int slot = (int)(t & 15);
  f32x2 v = {0.f, 0.f};
  if (row < rows && slot < slots) v = *(const f32x2*)(partials + (row * slots + slot) * 2);
  const float s = row16_total(v[0]), q = row16_total(v[1]);
  if (row >= rows || slot != 0) return;
  const float mean = s * inv_d;
  float var = q * inv_d - mean * mean;
  var = var > 0.f ? var : 0.f;
  const float rstd = rsqrtf(var + eps);
  const f32x2 o = {rstd, -mean * rstd};
  *(f32x2*)(ab + 2 * row) = o;
}
void launch_ln_stats_finalize(const float* partials, float* ab, long rows, int slots, int D, float eps, hipStream_t s) {
  // slots = D / 64 <= 16 (row widths up to 1024, checked by row_width_check)
  hipLaunchKernelGGL(ln_stats_finalize_kernel, dim3((unsigned)((rows * 16 + 255) / 256)), dim3(256), 0, s, partials, ab, rows,
                     slots, 1.0f / D, eps);
}
}  // namespace aaclip
