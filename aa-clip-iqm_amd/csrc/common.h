// Shared device helpers for the AA-CLIP gfx950 kernels.
// Written for CDNA4 only: 64-lane waves, MFMA 32x32x16 (f16/bf16), 32x32x2 (f32),
// 160 KiB LDS, global->LDS DMA.  No portability layer on purpose.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef _Float16 f16;
typedef __bf16 bf16;

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short i16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

#define AACLIP_DEV __device__ __forceinline__

// ---------------------------------------------------------------- dtype traits
template <typename T> struct Elem;
template <> struct Elem<f16> {
  typedef f16x8 vec8;
  typedef f16x4 vec4;
  static AACLIP_DEV f32x16 mma32(vec8 a, vec8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
};
template <> struct Elem<bf16> {
  typedef bf16x8 vec8;
  typedef bf16x4 vec4;
  static AACLIP_DEV f32x16 mma32(vec8 a, vec8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};

template <typename T> AACLIP_DEV T from_float(float v) { return (T)v; }
template <typename T> AACLIP_DEV float to_float(T v) { return (float)v; }

// ------------------------------------------------------------ split fp16 (AACLIP_F16X2)
// A value v is carried as two fp16 numbers hi = fp16(v), lo = fp16(v - hi): 21-22 significant bits for the values
// this path sees (|v| well inside fp16's normal range; lo goes subnormal below |v| ~ 2^-3 and then still resolves
// 2^-24 absolute).  A split row of width C is stored as C hi values followed by C lo values (row stride >= 2C
// halves), so a product over K becomes a product over "virtual" K tiles: for K tile j the kernels accumulate
// Ah_j.Wh_j + Al_j.Wh_j + Ah_j.Wl_j into the same fp32 accumulator (the Al.Wl term, 2^-22 relative, is dropped).
AACLIP_DEV void split16(float v, f16& hi, f16& lo) {
  hi = (f16)v;
  lo = (f16)(v - (float)hi);
}
// virtual K tile t of a split product -> byte offsets (from the hi plane's start) of its A and W tiles;
// K2 = bytes from a row's hi plane to its lo plane (2*K for a [.., 2K] split row).  NP = 3: both operands split;
// NP = 2: W has no lo plane (its values are exact in fp16), products Ah.W + Al.W only.
template <int NP> AACLIP_DEV int split_off_a(int t, int K2) {   // NP = 0: plain operands, tile t at t * 128 bytes
  if (NP == 0) return t * 128;
  if (NP == 3) {
    const int j = (int)(((unsigned)t * 43691u) >> 17);
    return j * 128 + (t - 3 * j == 1 ? K2 : 0);
  }
  return (t >> 1) * 128 + ((t & 1) ? K2 : 0);
}
template <int NP> AACLIP_DEV int split_off_w(int t, int K2) {
  if (NP == 0) return t * 128;
  if (NP == 3) {
    const int j = (int)(((unsigned)t * 43691u) >> 17);
    return j * 128 + (t - 3 * j == 2 ? K2 : 0);
  }
  return (t >> 1) * 128;
}
template <int NP> AACLIP_DEV void split_tile_off(int t, int K2, int& offA, int& offW) {
  if (NP == 3) {
    const int j = (int)(((unsigned)t * 43691u) >> 17);   // t / 3 for t < 98304
    const int r = t - 3 * j;
    offA = j * 128 + (r == 1 ? K2 : 0);
    offW = j * 128 + (r == 2 ? K2 : 0);
  } else {
    offA = (t >> 1) * 128 + ((t & 1) ? K2 : 0);
    offW = (t >> 1) * 128;
  }
}

// ------------------------------------------------------------ wave reductions
AACLIP_DEV float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
AACLIP_DEV float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ------------------------------------------------------- LDS tile addressing
// A staged tile is R rows of 128 bytes (64 sixteen-bit elements).  Two rows form
// one 256-byte LDS bank row of sixteen 16-byte slots.  The slot of (row, chunk)
// is XOR-swizzled so that
//   * the ds_read_b128 fragment read of an MFMA operand (16 lanes of a lane
//     group = 16 different rows, same chunk) touches 16 different slots, and
//   * the ds_read_b64_tr_b16 transposed read (4 rows x 4 chunks per 32-lane
//     half) touches 32 different 8-byte banks.
// swz() swaps bits 0 and 2 of the row-pair index: bit 0 of the pair index must
// move the slot by 4 for the transposed read, while any bijection works for the
// row read.  (Checked by tools/lds_bank_model.py against the guide's bank rules.)
AACLIP_DEV int swz16(int rp) { return (rp & 0xA) | ((rp & 1) << 2) | ((rp >> 2) & 1); }

// byte offset inside a tile of element-chunk (row, chunk), chunk = 16-byte unit 0..7
AACLIP_DEV int tile_off(int row, int chunk) {
  int rp = row >> 1;
  int slot = (((row & 1) << 3) | chunk) ^ swz16(rp & 15);
  return rp * 256 + slot * 16;
}

// Inverse map used on the SOURCE side of the LDS DMA (the DMA writes LDS
// linearly: slot p = wave-instruction base + lane): which (row, chunk) must the
// lane that lands in linear slot p fetch.
AACLIP_DEV void tile_src(int p, int& row, int& chunk) {
  int rp = p >> 4;
  int s = (p & 15) ^ swz16(rp & 15);
  row = rp * 2 + (s >> 3);
  chunk = s & 7;
}

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

// 16-byte global -> LDS DMA.  lds_base must be wave-uniform; lane i lands at
// lds_base + 16*i.
AACLIP_DEV void glds16(const void* gsrc, void* lds_base) {
  __builtin_amdgcn_global_load_lds((gbl_void*)gsrc, (lds_void*)lds_base, 16, 0, 0);
}

AACLIP_DEV void wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// XCD-aware bijective remap of a 1-D workgroup id (guide T1): workgroups that
// share an XCD (id % 8 equal) get a contiguous chunk of the tile list so that
// neighbouring tiles hit the same L2.
AACLIP_DEV int xcd_remap(int id, int n) {
  int q = n >> 3, r = n & 7, x = id & 7, j = id >> 3;
  int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
  return base + j;
}

// exact-erf GELU (nn.GELU default), reference model/model.py:84
AACLIP_DEV float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
// GELU for 16-bit outputs, on the packed-f32 FMA pipe only (v_pk_fma_f32 / v_pk_mul_f32, no transcendental):
// erf(z) = z * P(z^2) with z = x/sqrt2 clamped to +-3.2 (erf(3.2) = 1 - 6e-6) and P a degree-10 minimax-style
// fit in s = z^2 * (2/3.2^2) - 1.  |gelu error| <= 1.2e-5 for |x| < 8 and <= 1.6e-6 relative for x > 0.1, far
// below the 16-bit output rounding; the fp32 parity path keeps erff.  In the GEMM epilogues no MFMA runs beside
// it, so instruction count is what matters: 18 packed instructions + 2 clamps per pair of values (the A&S form
// with v_rcp/v_exp it replaces cost ~1.7x as many issue cycles).  Every 16-bit kernel uses this one routine so
// results do not depend on which kernel a batch size selects.
typedef float f32x2 __attribute__((ext_vector_type(2)));
AACLIP_DEV f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
AACLIP_DEV f32x2 splat2(float v) { return (f32x2){v, v}; }
AACLIP_DEV f32x2 gelu_fast2(f32x2 x) {
  f32x2 z = x * 0.70710678118654752440f;
  z[0] = __builtin_amdgcn_fmed3f(z[0], -3.2f, 3.2f);
  z[1] = __builtin_amdgcn_fmed3f(z[1], -3.2f, 3.2f);
  const f32x2 s = fma2(z * z, splat2(0.1953125f), splat2(-1.0f));
  f32x2 p = fma2(splat2(2.982273671e-03f), s, splat2(-7.046153473e-03f));
  p = fma2(p, s, splat2(7.957076705e-03f));
  p = fma2(p, s, splat2(-1.521942819e-02f));
  p = fma2(p, s, splat2(3.318292224e-02f));
  p = fma2(p, s, splat2(-5.471928813e-02f));
  p = fma2(p, s, splat2(8.062700147e-02f));
  p = fma2(p, s, splat2(-1.136467381e-01f));
  p = fma2(p, s, splat2(1.543549678e-01f));
  p = fma2(p, s, splat2(-2.173077339e-01f));
  p = fma2(p, s, splat2(4.413341836e-01f));
  const f32x2 hx = x * 0.5f;
  return fma2(hx, z * p, hx);   // 0.5x (1 + erf(x/sqrt2))
}
AACLIP_DEV float gelu_fast(float x) { return gelu_fast2((f32x2){x, x})[0]; }

AACLIP_DEV float leaky(float x) { return x >= 0.f ? x : 0.01f * x; }
