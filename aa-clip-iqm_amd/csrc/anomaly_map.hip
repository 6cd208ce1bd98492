// Patch x text-anchor similarity map, reference forward_utils.py:196-216 and the
// level fusion of test_last.py:95-100,149.  HBM-bound: the 518x518 output is
// written exactly once for all tap levels.
//   patch_scores : s = 100 * f . t  per patch (one wave per patch row);
//                  test mode keeps m = (s1 + 1 - s0) / 2, train mode keeps both channels
//   blur_upsample: per level Gaussian blur (kornia gaussian_blur2d semantics:
//                  normalised taps exp(-x^2/2s^2), reflect border, separable) in LDS,
//                  bilinear align_corners=True to S x S, summed over levels in level order
//   upsample_softmax2: train-mode branch (bilinear on both channels, softmax over C=2)
#include "common.h"
#include "kernels.h"

namespace aaclip {

template <int NCH>
__global__ __launch_bounds__(256) void patch_scores_kernel(const float* __restrict__ seg,
                                                           const float* __restrict__ anchors, long anchor_bstride,
                                                           float* __restrict__ out, int B, int P, int mode) {
  constexpr int E = NCH * 256;
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (long)B * P) return;
  const long b = row / P, pi = row - b * P;
  const float* f = seg + row * E;
  const float* t = anchors + b * anchor_bstride;  // [E, 2]
  float d0 = 0.f, d1 = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int e0 = (c * 64 + lane) * 4;
    f32x4 fv = *(const f32x4*)(f + e0);
    f32x4 ta = *(const f32x4*)(t + 2 * e0), tb = *(const f32x4*)(t + 2 * e0 + 4);
    d0 = fmaf(fv[0], ta[0], d0); d1 = fmaf(fv[0], ta[1], d1);
    d0 = fmaf(fv[1], ta[2], d0); d1 = fmaf(fv[1], ta[3], d1);
    d0 = fmaf(fv[2], tb[0], d0); d1 = fmaf(fv[2], tb[1], d1);
    d0 = fmaf(fv[3], tb[2], d0); d1 = fmaf(fv[3], tb[3], d1);
  }
  d0 = 100.0f * wave_sum(d0);
  d1 = 100.0f * wave_sum(d1);
  if (lane == 0) {
    if (mode == 0) {
      out[row] = (d1 + 1.0f - d0) / 2.0f;
    } else {
      out[(b * 2 + 0) * P + pi] = d0;
      out[(b * 2 + 1) * P + pi] = d1;
    }
  }
}

void launch_patch_scores(const float* seg, const float* anchors, long anchor_bstride, float* pre, int B, int P, int E,
                         int mode, hipStream_t s) {
  dim3 g((unsigned)(((long)B * P + 3) / 4));
  switch (E / 256) {
    case 1: hipLaunchKernelGGL(patch_scores_kernel<1>, g, dim3(256), 0, s, seg, anchors, anchor_bstride, pre, B, P, mode); break;
    case 2: hipLaunchKernelGGL(patch_scores_kernel<2>, g, dim3(256), 0, s, seg, anchors, anchor_bstride, pre, B, P, mode); break;
    case 3: hipLaunchKernelGGL(patch_scores_kernel<3>, g, dim3(256), 0, s, seg, anchors, anchor_bstride, pre, B, P, mode); break;
    case 4: hipLaunchKernelGGL(patch_scores_kernel<4>, g, dim3(256), 0, s, seg, anchors, anchor_bstride, pre, B, P, mode); break;
  }
}

AACLIP_DEV int reflect(int i, int n) {  // torch 'reflect' padding index (no edge repeat)
  if (i < 0) i = -i;
  if (i >= n) i = 2 * (n - 1) - i;
  return i;
}

// pre: [NL][B, g, g] (level stride B*g*g)  ->  out [B, S, S] = sum_l up(blur(pre_l))
// grid (bands, B); each block re-blurs its image (g*g*2*ksize FMAs: negligible)
// and writes rows [band*rows_per_band, ...).
constexpr int MAXG = 40;  // 37 at 518/14
__global__ __launch_bounds__(256) void blur_upsample_kernel(const float* __restrict__ pre, float* __restrict__ out,
                                                            int B, int g, int S, int NL, int ksize, float sigma,
                                                            int rows_per_band) {
  __shared__ float taps[16];
  __shared__ float src[MAXG * MAXG];
  __shared__ float tmp[MAXG * MAXG];
  __shared__ float blur[4][MAXG * MAXG];
  const int b = blockIdx.y, tid = threadIdx.x;
  const int r = ksize / 2;
  if (tid == 0) {
    float sum = 0.f;
    for (int i = 0; i < ksize; ++i) {
      float x = (float)(i - r) + ((ksize & 1) ? 0.f : 0.5f);
      float v = expf(-(x * x) / (2.0f * sigma * sigma));
      taps[i] = v;
      sum += v;
    }
    for (int i = 0; i < ksize; ++i) taps[i] /= sum;
  }
  for (int l = 0; l < NL; ++l) {
    __syncthreads();
    for (int i = tid; i < g * g; i += 256) src[i] = pre[((long)l * B + b) * g * g + i];
    __syncthreads();
    if (ksize > 1) {
      for (int i = tid; i < g * g; i += 256) {
        int y = i / g, x = i - y * g;
        float a = 0.f;
        for (int k = 0; k < ksize; ++k) a = fmaf(taps[k], src[y * g + reflect(x + k - r, g)], a);
        tmp[i] = a;
      }
      __syncthreads();
      for (int i = tid; i < g * g; i += 256) {
        int y = i / g, x = i - y * g;
        float a = 0.f;
        for (int k = 0; k < ksize; ++k) a = fmaf(taps[k], tmp[reflect(y + k - r, g) * g + x], a);
        blur[l][i] = a;
      }
    } else {
      for (int i = tid; i < g * g; i += 256) blur[l][i] = src[i];
    }
  }
  __syncthreads();
  const float scale = S > 1 ? (float)(g - 1) / (float)(S - 1) : 0.f;
  const int y_begin = blockIdx.x * rows_per_band;
  int y_end = y_begin + rows_per_band;
  if (y_end > S) y_end = S;
  for (long i = (long)y_begin * S + tid; i < (long)y_end * S; i += 256) {
    const int y = i / S, x = i - (long)y * S;
    const float sy = scale * y, sx = scale * x;
    const int y0 = (int)sy, x0 = (int)sx;
    const int y1 = y0 + (y0 < g - 1 ? 1 : 0), x1 = x0 + (x0 < g - 1 ? 1 : 0);
    const float ly1 = sy - y0, ly0 = 1.0f - ly1, lx1 = sx - x0, lx0 = 1.0f - lx1;
    float acc = 0.f;
    for (int l = 0; l < NL; ++l) {
      const float* m = blur[l];
      float v = ly0 * (lx0 * m[y0 * g + x0] + lx1 * m[y0 * g + x1]) + ly1 * (lx0 * m[y1 * g + x0] + lx1 * m[y1 * g + x1]);
      acc = (l == 0) ? v : acc + v;
    }
    out[(long)b * S * S + i] = acc;
  }
}

void launch_blur_upsample(const float* pre, float* out, int B, int g, int S, int NL, int ksize, float sigma,
                          hipStream_t s) {
  const int bands = 14;
  const int rpb = (S + bands - 1) / bands;
  hipLaunchKernelGGL(blur_upsample_kernel, dim3(bands, B), dim3(256), 0, s, pre, out, B, g, S, NL, ksize, sigma, rpb);
}

// scores [B, 2, g, g] -> out [B, 2, S, S]: bilinear (align_corners=True) then softmax over the channel pair
__global__ __launch_bounds__(256) void upsample_softmax2_kernel(const float* __restrict__ sc, float* __restrict__ out,
                                                                int g, int S, int rows_per_band) {
  __shared__ float m0[MAXG * MAXG];
  __shared__ float m1[MAXG * MAXG];
  const int b = blockIdx.y, tid = threadIdx.x;
  for (int i = tid; i < g * g; i += 256) {
    m0[i] = sc[((long)b * 2 + 0) * g * g + i];
    m1[i] = sc[((long)b * 2 + 1) * g * g + i];
  }
  __syncthreads();
  const float scale = S > 1 ? (float)(g - 1) / (float)(S - 1) : 0.f;
  const int y_begin = blockIdx.x * rows_per_band;
  int y_end = y_begin + rows_per_band;
  if (y_end > S) y_end = S;
  for (long i = (long)y_begin * S + tid; i < (long)y_end * S; i += 256) {
    const int y = i / S, x = i - (long)y * S;
    const float sy = scale * y, sx = scale * x;
    const int y0 = (int)sy, x0 = (int)sx;
    const int y1 = y0 + (y0 < g - 1 ? 1 : 0), x1 = x0 + (x0 < g - 1 ? 1 : 0);
    const float ly1 = sy - y0, ly0 = 1.0f - ly1, lx1 = sx - x0, lx0 = 1.0f - lx1;
    float a = ly0 * (lx0 * m0[y0 * g + x0] + lx1 * m0[y0 * g + x1]) + ly1 * (lx0 * m0[y1 * g + x0] + lx1 * m0[y1 * g + x1]);
    float c = ly0 * (lx0 * m1[y0 * g + x0] + lx1 * m1[y0 * g + x1]) + ly1 * (lx0 * m1[y1 * g + x0] + lx1 * m1[y1 * g + x1]);
    const float mx = fmaxf(a, c);
    const float ea = expf(a - mx), ec = expf(c - mx);
    const float inv = 1.0f / (ea + ec);
    out[((long)b * 2 + 0) * S * S + i] = ea * inv;
    out[((long)b * 2 + 1) * S * S + i] = ec * inv;
  }
}

void launch_upsample_softmax2(const float* scores, float* out, int B, int g, int S, hipStream_t s) {
  const int bands = 14;
  const int rpb = (S + bands - 1) / bands;
  hipLaunchKernelGGL(upsample_softmax2_kernel, dim3(bands, B), dim3(256), 0, s, scores, out, g, S, rpb);
}

}  // namespace aaclip
