// v_mfma_f32_16x16x32 wrapper and the LDS tile addressing shared by the 256x256 GEMM kernels
// (gemm256t.hip, gemm256z.hip).
#pragma once
#include "common.h"

namespace aaclip {

template <typename T> struct Mma16;
template <> struct Mma16<f16> {
  static AACLIP_DEV f32x4 mma(f16x8 a, f16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};
template <> struct Mma16<bf16> {
  static AACLIP_DEV f32x4 mma(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};

AACLIP_DEV int tile_off_id(int row, int chunk) {
  const int rp = row >> 1;
  return rp * 256 + (((((row & 1) << 3) | chunk) ^ (rp & 15)) << 4);
}
AACLIP_DEV void tile_src_id(int p, int& row, int& chunk) {
  const int rp = p >> 4;
  const int s = (p & 15) ^ (rp & 15);
  row = rp * 2 + (s >> 3);
  chunk = s & 7;
}

// the same addressing for kernels outside the 256-tile family (gemm.hip: small-M split kernel)
AACLIP_DEV int tile_off_s(int row, int chunk) { return tile_off_id(row, chunk); }
AACLIP_DEV void tile_src_s(int p, int& row, int& chunk) { tile_src_id(p, row, chunk); }

}  // namespace aaclip
