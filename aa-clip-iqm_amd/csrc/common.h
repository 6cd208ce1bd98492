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
// A value v is carried as hi = fp16(v) plus a correction for v - hi, so that a product sees ~15-21 bits of each operand
// instead of 11.  Two row formats (logical width C, 4 bytes per element, row stride >= 2C halves):
//   split16 row  [hi: C x fp16][lo: C x fp16]                    lo = fp16(v - hi).  Attention inputs (q | k | v): the
//                attention kernel runs q.k^T as Kh.Qh + Kl.Qh + Kh.Ql and p.v as Vh.P on the fp16 MFMAs (v's lo half
//                is not used: attention.hip says why).
//   split8 row   [hi: C x fp16][lo8: C x e4m3][hi8: C x e4m3]    lo8 = e4m3((v - hi) * 2^10), hi8 = e4m3(v).  GEMM
//                A operands.  Weights likewise: [Wh: K x fp16][Wh8 = e4m3(W * 2^6)][Wl8 = e4m3((W - Wh) * 2^17)]
//                (the last plane is absent when the weight is exact in fp16).  A product over K is accumulated as
//                Ah.Wh (fp16 MFMA) + Al8.Wh8 + Ah8.Wl8 (block-scaled e4m3 MFMA, 16x16x128, twice the fp16 rate): the
//                correction terms are 2^-11 of the main term, so 4 significant bits in them leave ~2^-15 relative
//                error per operand (measured on random data: 28x below plain fp16), at 2 instead of 3 MFMA time units.
// The scales are fixed powers of two (e4m3 spans 2^-9 ... 448): activations |v| in 0.016 ... 448 and weights |w| in
// 2.4e-4 ... 7 keep >= 4 significant bits in their correction operands; beyond, a correction operand saturates or goes
// subnormal, which only degrades THAT element's correction towards plain fp16 -- never the main term.
constexpr int SPLIT8_ACT_LO_EXP = 10, SPLIT8_ACT_HI_EXP = 0, SPLIT8_W_HI_EXP = 6, SPLIT8_W_LO_EXP = 17;
AACLIP_DEV void split16(float v, f16& hi, f16& lo) {
  hi = (f16)v;
  lo = (f16)(v - (float)hi);
}
// four values -> four e4m3 bytes (value * 2^EXP, clamped to +-448: the conversion itself returns NaN on overflow)
template <int EXP> AACLIP_DEV uint32_t pack_e4m3x4(float a, float b, float c, float d) {
  constexpr float S = (float)(1 << EXP);
  a = __builtin_amdgcn_fmed3f(a * S, -448.f, 448.f);
  b = __builtin_amdgcn_fmed3f(b * S, -448.f, 448.f);
  c = __builtin_amdgcn_fmed3f(c * S, -448.f, 448.f);
  d = __builtin_amdgcn_fmed3f(d * S, -448.f, 448.f);
  int r = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
  r = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, r, true);
  return (uint32_t)r;
}
// the three planes of four split8 values
AACLIP_DEV void split8x4(const float (&v)[4], f16x4& hi, uint32_t& lo8, uint32_t& hi8) {
  float r[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    hi[j] = (f16)v[j];
    r[j] = v[j] - (float)hi[j];
  }
  lo8 = pack_e4m3x4<SPLIT8_ACT_LO_EXP>(r[0], r[1], r[2], r[3]);
  hi8 = pack_e4m3x4<SPLIT8_ACT_HI_EXP>(v[0], v[1], v[2], v[3]);
}
// The same from three instructions less per value: with MODE.FP16_OVFL set the fp8 conversions SATURATE at +-448 instead
// of returning NaN, and v_cvt_scalef32_pk_fp8_f32 divides by a power-of-two scale on the way (tools/cvt_fp8_probe.hip:
// bit-identical to multiply + clamp + convert on every probe value).  A kernel that calls split8x4_sat must have run
// fp8_saturate_mode() first (the mode is per wave and lasts until the wave ends; it also turns an overflowing f32 ->
// f16 conversion into +-65504 instead of inf, which no caller here distinguishes).
AACLIP_DEV void fp8_saturate_mode() { asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1"); }
AACLIP_DEV void split8x4_sat(const float (&v)[4], f16x4& hi, uint32_t& lo8, uint32_t& hi8) {
  typedef short i16x2 __attribute__((ext_vector_type(2)));
  typedef float f32x4_ __attribute__((ext_vector_type(4)));
  // as VECTOR conversions: hipcc then packs with v_cvt_pk_f16_f32 and unpacks the odd halves with v_cvt_f32_f16_sdwa
  // (14 instructions per four values; the element-wise form gave 4 + 4 scalar conversions AND the two packs: 18)
  const f32x4_ vv = {v[0], v[1], v[2], v[3]};
  hi = __builtin_convertvector(vv, f16x4);
  const f32x4_ r = vv - __builtin_convertvector(hi, f32x4_);
  constexpr float S_LO = 1.0f / (float)(1 << SPLIT8_ACT_LO_EXP);
  i16x2 a = {0, 0};
  a = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(a, r[0], r[1], S_LO, false);
  a = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(a, r[2], r[3], S_LO, true);
  lo8 = __builtin_bit_cast(uint32_t, a);
  static_assert(SPLIT8_ACT_HI_EXP == 0, "hi8 is the plain conversion");
  int b = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], 0, false);
  b = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], b, true);
  hi8 = (uint32_t)b;
}
// Virtual K tiles of a split8 product.  K tiles (64 wide) are taken in pairs (2i, 2i+1):
//   NP = 4: [fp16 tile 2i] [fp16 tile 2i+1] [e4m3 tile T1: Al8 . Wh8 over both] [e4m3 tile T2: Ah8 . Wl8 over both]
//   NP = 3: the same without T2 (weight exact in fp16)
// Every tile is rows of 128 bytes in LDS; an e4m3 tile's 128 bytes are 128 K values.  Byte offsets from the row start
// (K = logical width): fp16 tile j at 128*j, T1 pair i at 2K + 128*i, T2 pair i at 3K + 128*i, the same on both sides.
// kind: 0 fp16, 1 T1, 2 T2.  NP = 0: plain operands, tile t at 128*t.
template <int NP> AACLIP_DEV int vtile_off(int t, int K, int& kind) {
  if (NP == 0) { kind = 0; return t * 128; }
  int i, r;
  if (NP == 4) { i = t >> 2; r = t & 3; }
  else { i = (int)(((unsigned)t * 43691u) >> 17); r = t - 3 * i; }   // t / 3 for t < 98304
  kind = r < 2 ? 0 : r - 1;
  return r < 2 ? (2 * i + r) * 128 : r * K + 128 * i;
}
template <int NP> AACLIP_DEV int vtile_count(int K) { return NP == 0 ? (K >> 6) : (NP == 4 ? (K >> 5) : 3 * (K >> 7)); }
// e8m0 scale bytes of the block-scaled MFMA for an e4m3 tile of kind 1 / 2 (stored = value * 2^EXP, so scale 2^-EXP)
AACLIP_DEV int vtile_scale_act(int kind) { return 127 - (kind == 1 ? SPLIT8_ACT_LO_EXP : SPLIT8_ACT_HI_EXP); }
AACLIP_DEV int vtile_scale_w(int kind) { return 127 - (kind == 1 ? SPLIT8_W_HI_EXP : SPLIT8_W_LO_EXP); }
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
// 16x16x128 e4m3 MFMA on two 16-byte fragments per side: lane group g = lane >> 4 holds K bytes [16g, 16g+16) in the
// first fragment and [64 + 16g, 64 + 16g + 16) in the second (tools/mfma_f8_probe.hip), i.e. exactly the 16-byte chunks
// g and 4 + g of a 128-byte tile row that the fp16 16x16x32 kernels read for their two k-steps.
AACLIP_DEV f32x4 mma_e4m3(f16x8 a0, f16x8 a1, f16x8 b0, f16x8 b1, f32x4 c, int scale_a, int scale_b) {
  const i32x4 x0 = __builtin_bit_cast(i32x4, a0), x1 = __builtin_bit_cast(i32x4, a1);
  const i32x4 y0 = __builtin_bit_cast(i32x4, b0), y1 = __builtin_bit_cast(i32x4, b1);
  const i32x8 a = __builtin_shufflevector(x0, x1, 0, 1, 2, 3, 4, 5, 6, 7);
  const i32x8 b = __builtin_shufflevector(y0, y1, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, scale_a, 0, scale_b);
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
