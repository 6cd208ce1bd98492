// Image pre-processing in front of the patch embed: 8-bit bicubic resize to S x S, ToTensor,
// Normalize -- the reference's dataset transform (reference dataset/__init__.py:150-161), whose
// resize is Pillow's 8-bit two-pass resampler.  Integer / byte work: one kernel reads
// every source pixel once (plus the tile halo), keeps the horizontally resampled rows of a tile in
// LDS as uint8 (Pillow rounds to uint8 between its two passes, so the intermediate IS 8-bit) and
// writes the normalised fp32 planes once.  The tile's source rectangle is first copied to LDS with
// 16-byte loads (byte-granular gathers straight from global memory were TA-bound: 27 one-byte
// loads per intermediate pixel).  Bit-exact with Pillow: same 22-bit fixed-point weights,
// same int32 accumulation, same rounding and clipping; the u8 -> fp32 normalisation is a
// 3 x 256-entry table built by the host with the very fp32 operations ToTensor/Normalize perform.
#include "common.h"
#include "kernels.h"
#include <math.h>

namespace aaclip {

static constexpr int PRECISION_BITS = 32 - 8 - 2;
static constexpr int PP_TX = 64;  // output columns per workgroup

// --------------------------------------------------------------------------- host: weight tables
static double bicubic_weight(double x) {
  const double a = -0.5;
  if (x < 0.0) x = -x;
  if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
  if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
  return 0.0;
}

int resample_ksize(int in_size, int out_size) {
  if (in_size == out_size) return 1;
  double scale = (double)in_size / out_size;
  double filterscale = scale < 1.0 ? 1.0 : scale;
  return (int)ceil(2.0 * filterscale) * 2 + 1;
}

// bounds[2*i] = first source index, bounds[2*i+1] = tap count; coefs[i*ksize + k] fixed point.
// Equal sizes give the identity table (Pillow skips that pass; 1<<22 weights reproduce the pixel).
void resample_table(int in_size, int out_size, int32_t* bounds, int32_t* coefs) {
  const int ksize = resample_ksize(in_size, out_size);
  if (in_size == out_size) {
    for (int i = 0; i < out_size; ++i) {
      bounds[2 * i] = i;
      bounds[2 * i + 1] = 1;
      coefs[i] = 1 << PRECISION_BITS;
    }
    return;
  }
  const double scale = (double)in_size / out_size;
  const double filterscale = scale < 1.0 ? 1.0 : scale;
  const double support = 2.0 * filterscale;
  const double ss = 1.0 / filterscale;
  double* w = new double[ksize];
  for (int i = 0; i < out_size; ++i) {
    const double center = (i + 0.5) * scale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    const int n = xmax - xmin;
    double ww = 0.0;
    for (int x = 0; x < n; ++x) {
      w[x] = bicubic_weight((x + xmin - center + 0.5) * ss);
      ww += w[x];
    }
    for (int x = 0; x < ksize; ++x) {
      double v = 0.0;
      if (x < n) v = ww != 0.0 ? w[x] / ww : w[x];
      coefs[(size_t)i * ksize + x] =
          v < 0 ? (int32_t)(-0.5 + v * (1 << PRECISION_BITS)) : (int32_t)(0.5 + v * (1 << PRECISION_BITS));
    }
    bounds[2 * i] = xmin;
    bounds[2 * i + 1] = n;
  }
  delete[] w;
}

// Upper bound of the source rows one tile of `ty` output rows touches.
int preprocess_tile_rows(int in_size, int out_size, int ty) {
  const int k = resample_ksize(in_size, out_size);
  const double scale = (double)in_size / out_size;
  return (int)ceil((ty - 1) * scale) + 1 + k;
}

// LDS bytes per staged source row: the 64 output columns of a tile touch at most this many source
// pixels; + 16 for the 16-byte alignment of the staged segment, rounded to 16.
int preprocess_row_pitch(int in_w, int out_size) {
  const int span = preprocess_tile_rows(in_w, out_size, PP_TX);
  return ((span * 3 + 16 + 15) / 16) * 16;
}

// ------------------------------------------------------------------------------------- device
AACLIP_DEV int clip8(int acc) {
  int v = acc >> PRECISION_BITS;
  return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// grid (ceil(S/64), ceil(S/TY), B), 256 threads.  Dynamic LDS: [lds_rows][pitch] source bytes (16-byte
// aligned segments copied with 16-byte loads), then [3][lds_rows][64] uint8 planes of the horizontal pass.
template <bool STAGE>
__global__ __launch_bounds__(256) void preprocess_kernel(const uint8_t* __restrict__ src, int Hs, int Ws, int S,
                                                         const int32_t* __restrict__ hb, const int32_t* __restrict__ hk,
                                                         int kx, const int32_t* __restrict__ vb,
                                                         const int32_t* __restrict__ vk, int ky, int TY, int lds_rows,
                                                         int pitch, long total_bytes, const float* __restrict__ lut,
                                                         float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  __shared__ float s_lut[3 * 256];
  uint8_t* stage = lds;
  uint8_t* tmp = lds + (STAGE ? (size_t)lds_rows * pitch : 0);   // !STAGE: no source copy, pass 1 reads global memory
  const int x0 = blockIdx.x * PP_TX, y0 = blockIdx.y * TY, b = blockIdx.z;
  const int ny = min(TY, S - y0), nx = min(PP_TX, S - x0);
  for (int i = threadIdx.x; i < 3 * 256; i += 256) s_lut[i] = lut[i];
  const int r0 = vb[2 * y0];
  const int ylast = y0 + ny - 1;
  int R = vb[2 * ylast] + vb[2 * ylast + 1] - r0;
  if (R > lds_rows) R = lds_rows;  // cannot happen when the host bound holds; never index past the tile
  const int c0 = hb[2 * x0];
  const int xlast = x0 + nx - 1;
  const int c1 = hb[2 * xlast] + hb[2 * xlast + 1];
  const long img0 = (long)b * Hs * Ws * 3;

  // pass 0: source rows r0 .. r0+R-1, byte range [c0*3, c1*3) of each, into LDS; a row's segment starts at
  // its 16-byte aligned global address, so the first `shift` bytes of the LDS row are padding
  const int chunks = pitch >> 4;
  for (int i = threadIdx.x; STAGE && i < R * chunks; i += 256) {
    const int r = i / chunks, ch = i - r * chunks;
    const long a0 = img0 + ((long)(r0 + r) * Ws + c0) * 3;
    const long a = (a0 & ~15L) + ch * 16;
    const long aend = img0 + ((long)(r0 + r) * Ws + c1) * 3;
    if (a >= aend) continue;
    uint8_t* d = stage + r * pitch + ch * 16;
    if (a + 16 <= total_bytes) {
      *(u32x4*)d = *(const u32x4*)(src + a);
    } else {
      for (int j = 0; j < 16; ++j) d[j] = a + j < total_bytes ? src[a + j] : 0;
    }
  }
  __syncthreads();

  // pass 1: horizontal, from LDS (measured: this flat one-element-per-thread form beats a lane = column /
  // wave = row arrangement with the weights held in registers; the kernel is bound by instruction issue
  // -- one 8-bit x 22-bit multiply-add per instruction -- not by HBM)
  for (int i = threadIdx.x; i < R * PP_TX; i += 256) {
    const int r = i / PP_TX, xl = i % PP_TX;
    if (xl >= nx) continue;
    const int x = x0 + xl;
    const int first = hb[2 * x], n = hb[2 * x + 1];
    const int32_t* k = hk + (size_t)x * kx;
    const int shift = (int)((img0 + ((long)(r0 + r) * Ws + c0) * 3) & 15);
    const uint8_t* p = STAGE ? stage + r * pitch + shift + (first - c0) * 3
                             : src + img0 + ((long)(r0 + r) * Ws + first) * 3;
    int a0 = 1 << (PRECISION_BITS - 1), a1 = a0, a2 = a0;
    for (int t = 0; t < n; ++t) {
      const int w = k[t];
      a0 += (int)p[3 * t] * w;
      a1 += (int)p[3 * t + 1] * w;
      a2 += (int)p[3 * t + 2] * w;
    }
    tmp[(0 * lds_rows + r) * PP_TX + xl] = (uint8_t)clip8(a0);
    tmp[(1 * lds_rows + r) * PP_TX + xl] = (uint8_t)clip8(a1);
    tmp[(2 * lds_rows + r) * PP_TX + xl] = (uint8_t)clip8(a2);
  }
  __syncthreads();

  // pass 2: vertical + normalisation table, one output element per (channel, row, column)
  const size_t plane = (size_t)S * S;
  float* ob = out + (size_t)b * 3 * plane;
  for (int i = threadIdx.x; i < 3 * ny * PP_TX; i += 256) {
    const int xl = i % PP_TX, yc = i / PP_TX;
    const int yl = yc % ny, c = yc / ny;
    if (xl >= nx) continue;
    const int y = y0 + yl;
    const int first = vb[2 * y] - r0, n = vb[2 * y + 1];
    const int32_t* k = vk + (size_t)y * ky;
    int acc = 1 << (PRECISION_BITS - 1);
    for (int t = 0; t < n; ++t) {
      const int r = first + t;
      if (r < R) acc += (int)tmp[(c * lds_rows + r) * PP_TX + xl] * k[t];
    }
    ob[c * plane + (size_t)y * S + x0 + xl] = s_lut[c * 256 + clip8(acc)];
  }
}

void launch_preprocess(const uint8_t* src, int B, int Hs, int Ws, int S, const int32_t* hb, const int32_t* hk, int kx,
                       const int32_t* vb, const int32_t* vk, int ky, int TY, int lds_rows, int pitch, const float* lut,
                       float* out, hipStream_t s) {
  dim3 grid((S + PP_TX - 1) / PP_TX, (S + TY - 1) / TY, B);
  if (pitch > 0) {
    const size_t lds = (size_t)lds_rows * pitch + (size_t)3 * lds_rows * PP_TX;
    hipLaunchKernelGGL(preprocess_kernel<true>, grid, dim3(256), lds, s, src, Hs, Ws, S, hb, hk, kx, vb, vk, ky, TY,
                       lds_rows, pitch, (long)B * Hs * Ws * 3, lut, out);
  } else {   // pitch 0: the source rectangle of a tile does not fit in LDS (very large frames); gather from global memory
    const size_t lds = (size_t)3 * lds_rows * PP_TX;
    hipLaunchKernelGGL(preprocess_kernel<false>, grid, dim3(256), lds, s, src, Hs, Ws, S, hb, hk, kx, vb, vk, ky, TY,
                       lds_rows, 0, (long)B * Hs * Ws * 3, lut, out);
  }
}

}  // namespace aaclip
