// Fused multi-head self-attention for head_dim 64 on gfx950 (never materialises
// the L x L score matrix).  Computes what nn.MultiheadAttention's math path
// does for the reference (model/transformer.py:200,237): softmax(q k^T + mask) v
// per head; q arrives already scaled by head_dim^-1/2 (QKV GEMM epilogue).
//
// 16-bit kernel: one workgroup = 4 waves = 128 query rows of one (batch, head);
// key/value tiles of 64 rows are DMA'd into a 2-stage swizzled LDS ring.
//   S^T = K . Q^T  (v_mfma 32x32x16): the query sits on the LANE, keys in the
//          accumulator registers, so max / sum / rescale are lane-local + one
//          cross-half exchange.
//   O^T = V^T . P^T: the S^T accumulator tile is reused as the B operand without
//          leaving registers (guide section 3, "accumulator tile as the next MFMA's
//          operand"); V^T fragments come from ds_read_b64_tr_b16 on the row-major
//          V tile.
// L need not be a multiple of anything: out-of-range keys are masked to -inf,
// out-of-range query rows are computed on clamped data and not stored.
// fp32 kernel: plain VALU flash attention, one query per lane (parity path).
#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace aaclip {

// LDS images of the 64-row x 128-byte K and V tiles (16-byte chunk index XOR-swizzled):
//   K tile: chunk ^= (row>>1)&7   -> the MFMA operand ds_read_b128 (16 rows per lane group)
//                                    is conflict-free; rows 32 apart differ by a constant
//   V tile: chunk ^= ((row>>1)&1)<<2 -> the ds_read_b64_tr_b16 (4 rows x 4 chunks per
//                                    32-lane half) is conflict-free; the 16 reads of a tile
//                                    are two per-lane bases + immediates
// (checked with tools/lds_bank_model.py).  The DMA writes LDS linearly, so the inverse
// map goes on the source address.
AACLIP_DEV int xk(int row) { return (row >> 1) & 7; }
AACLIP_DEV int xv(int row) { return ((row >> 1) & 1) << 2; }

// LOG2Q = true: q arrives multiplied by head_dim^-1/2 * log2(e) (aaclip_block folds the
// factor into the QKV GEMM epilogue, one rounding), so scores are already in log2 units and
// the running maximum is subtracted INSIDE the S^T MFMA chain by starting its accumulator at
// -max: the softmax numerator is one v_exp_f32 per element.  LOG2Q = false: q carries
// head_dim^-1/2 only (the plain aaclip_attention contract) and the factor is applied here.
template <typename T, bool LOG2Q>
__global__ __launch_bounds__(256, 3) void attn16_kernel(const T* __restrict__ qkv, T* __restrict__ ctx, int L, int H,
                                                        int causal) {
  typedef typename Elem<T>::vec8 vec8;
  typedef typename Elem<T>::vec4 vec4;
  typedef short i16x8 __attribute__((ext_vector_type(8)));
  __shared__ __attribute__((aligned(16))) char smem[32768];  // 2 stages x (K 8K + V 8K)
  constexpr float LOG2E = 1.4426950408889634f;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int qt = blockIdx.x, head = blockIdx.y, b = blockIdx.z;
  const int D = H * 64;
  const long ld = 3L * D;
  const T* base = qkv + (long)b * L * ld + head * 64;
  const int q0 = qt * 128 + wave * 32;
  const int qi = q0 + r;
  const int qrow = qi < L ? qi : L - 1;

  vec8 qf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    qf[ks] = *(const vec8*)(base + (long)qrow * ld + 16 * ks + 8 * h);
  }

  // DMA: this wave fills rows [16*wave, 16*wave+16) of the K and of the V tile (2 x 1 KiB each)
  const T* ksrc[2];
  const T* vsrc[2];
  int drow[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int pslot = (wave * 2 + j) * 64 + lane;  // linear 16-byte slot
    const int row = pslot >> 3, sl = pslot & 7;
    drow[j] = row;
    ksrc[j] = base + (long)row * ld + D + (sl ^ xk(row)) * 8;
    vsrc[j] = base + (long)row * ld + 2 * D + (sl ^ xv(row)) * 8;
  }
  // K operand reads: row = sub*32 + r, chunk = 2*ks + h  ->  4 per-lane bases, sub adds 4096
  int koff[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) koff[ks] = r * 128 + (((2 * ks + h) ^ xk(r)) << 4);
  // V transposed reads: lane group g = lane>>4, i = lane&15 = 4*qq + pp; block rows key0+qq,
  // key0 = sub*32 + 16*s2 + 8*t + 4*(g>>1); columns db*32 + (g&1)*16 + 4*pp
  int voff[2];
  {
    const int g = lane >> 4, i = lane & 15, qq = i >> 2, pp = i & 3;
    const int row = 4 * (g >> 1) + qq;
#pragma unroll
    for (int db = 0; db < 2; ++db) {
      const int chunk = db * 4 + (g & 1) * 2 + (pp >> 1);
      voff[db] = 8192 + row * 128 + ((chunk ^ xv(row)) << 4) + (pp & 1) * 8;
    }
  }

  int last_q = qt * 128 + 127;
  if (last_q > L - 1) last_q = L - 1;
  const int nkt = causal ? (last_q / 64 + 1) : ((L + 63) / 64);

  auto stage = [&](int st, int kt) {
    char* dst = smem + st * 16384 + wave * 2048;
    const long step = (long)kt * 64 * ld;
    if (kt * 64 + 64 <= L) {   // whole tile in range: no clamping
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        glds16(ksrc[j] + step, dst + j * 1024);
        glds16(vsrc[j] + step, dst + 8192 + j * 1024);
      }
    } else {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        int over = kt * 64 + drow[j] - (L - 1);
        over = over > 0 ? over : 0;   // rows past the end re-read row L-1 (masked later)
        glds16(ksrc[j] + step - (long)over * ld, dst + j * 1024);
        glds16(vsrc[j] + step - (long)over * ld, dst + 8192 + j * 1024);
      }
    }
  };

  f32x16 o[2];
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int e = 0; e < 16; ++e) o[db][e] = 0.f;
  float m2 = 0.f, l = 0.f;   // running row max in log2 units (valid after the first tile), row sum

  auto tile = [&](const char* sb, int kt) {
    // S'^T = K . Q^T - m2 : the accumulator starts at -m2 (the query is on the lane)
    f32x16 s[2];
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
      for (int e = 0; e < 16; ++e) s[sub][e] = LOG2Q ? -m2 : 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        vec8 a = *(const vec8*)(sb + koff[ks] + sub * 4096);
        s[sub] = Elem<T>::mma32(a, qf[ks], s[sub]);
      }
      if (!LOG2Q) {
#pragma unroll
        for (int e = 0; e < 16; ++e) s[sub][e] = fmaf(s[sub][e], LOG2E, -m2);
      }
    }
    const int k0 = kt * 64;
    const bool need_mask = (k0 + 64 > L) || (causal && (k0 + 63 > q0));
    if (need_mask) {
#pragma unroll
      for (int sub = 0; sub < 2; ++sub)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          int key = k0 + sub * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
          bool dead = (key >= L) || (causal && key > qi);
          s[sub][e] = dead ? -INFINITY : s[sub][e];
        }
    }
    float mt = s[0][0];
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
      for (int e = 0; e < 16; ++e) mt = fmaxf(mt, s[sub][e]);
    mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
    // mt is the tile maximum relative to the running maximum.  Only when some row of this
    // wave raises its maximum (always in the first tile) are scores, O and l re-based.
    const bool first = kt == 0;
    if (first || __any(mt > 0.f)) {
      const float delta = first ? mt : fmaxf(mt, 0.f);
      // tile 0: l and o are still zero and must stay so -- exp2(-mt) overflows to +inf when every score of the
      // tile is below about -128 (log2 units), and 0 * inf would make the whole row NaN
      const float alpha = first ? 1.f : __builtin_amdgcn_exp2f(-delta);
      m2 += delta;
      l *= alpha;
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[db][e] *= alpha;
#pragma unroll
      for (int sub = 0; sub < 2; ++sub)
#pragma unroll
        for (int e = 0; e < 16; ++e) s[sub][e] -= delta;
    }
    float rs = 0.f;
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        float pv = __builtin_amdgcn_exp2f(s[sub][e]);
        s[sub][e] = pv;
        rs += pv;
      }
    rs += __shfl_xor(rs, 32, 64);
    l += rs;

#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        vec8 pf;
#pragma unroll
        for (int j = 0; j < 8; ++j) pf[j] = from_float<T>(s[sub][8 * s2 + j]);
#pragma unroll
        for (int db = 0; db < 2; ++db) {
          const char* vp = sb + voff[db] + (sub * 32 + 16 * s2) * 128;
          i16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)(vp));
          i16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)(vp + 8 * 128));
          i16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
          o[db] = Elem<T>::mma32(__builtin_bit_cast(vec8, both), pf, o[db]);
        }
      }
  };

  stage(0, 0);
  wait_vm0();
  __syncthreads();
  for (int kt = 0; kt < nkt; kt += 2) {   // unrolled by two so the stage offsets are immediates
    if (kt + 1 < nkt) stage(1, kt + 1);
    tile(smem, kt);
    wait_vm0();
    __syncthreads();
    if (kt + 1 >= nkt) break;
    if (kt + 2 < nkt) stage(0, kt + 2);
    tile(smem + 16384, kt + 1);
    wait_vm0();
    __syncthreads();
  }

  if (qi < L) {
    const float inv = 1.0f / l;
    T* dst = ctx + ((long)b * L + qi) * D + head * 64;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int gi = 0; gi < 4; ++gi) {
        vec4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = from_float<T>(o[db][4 * gi + j] * inv);
        *(vec4*)(dst + db * 32 + 8 * gi + 4 * h) = v;
      }
  }
}

// ---------------------------------------------------------------------------
// Split-fp16 variant (AACLIP_F16X2, common.h): q, k, v arrive as hi + lo fp16 pairs (split16 rows [hi 3D | lo 3D]), the
// context leaves as split8 rows ([hi D fp16 | lo8 D | hi8 D]: the A operand of the out_proj product).  Same structure as
// attn16_kernel; per 64-key tile a wave issues
//   S^T = Kh.Qh^T + Kl.Qh^T + Kh.Ql^T   (24 MFMAs: scores carry ~21 bits of q and k)
//   O^T += Vh^T.P^T                      (8 MFMAs: P and v in fp16)
// v's lo half is NOT used: its rounding errors are independent per key and average out under the softmax weights,
// while an error of q or k moves a whole row / column of scores.  Measured on the full-size B = 4 golden record (worst
// ratio to the 1e-3 + 1e-2 |ref| bound over taps and maps): all five products 0.14, without Vl.P 0.28, without Kl.Qh
// 0.82, without Kh.Ql 0.67 -- so Vl.P (20 % of the MFMAs, a quarter of the LDS bytes) is dropped and the other two stay.
// Stage image: [Kh 8K][Vh 8K][Kl 8K], two stages = 48 KiB: three workgroups per CU.
// VL = true keeps the fifth product (stage image + [Vl 8K], two workgroups per CU): short sequences (the text tower's
// L = 77, the V-V path over the batch axis) have too few keys per row for the averaging argument, and cost nothing.
// QK8 = true (the block path at L >= 512): the two correction products run on the block-scaled e4m3 MFMA like the
// GEMMs' (v_mfma_scale_f32_32x32x64_f8f6f4: head dim 64 = ONE instruction per 32 keys x 32 queries, 64 pipe cycles
// where the four fp16 MFMAs of a correction product take 128).  Rows then are
//   [q k v hi: 3D fp16][q: per head [lo8 64 B | hi8 64 B]][k: the same]        10 D bytes, stride 5D halves
// (lo8 = e4m3((x - hi) 2^10), hi8 = e4m3(x), written by the QKV epilogue, GemmParams::out_qk8): the 128-byte e4m3 record
// of a (row, head) has the geometry of a 64-column fp16 row, so the K8 tile is staged by the same DMA pattern as Kl was
// and tools/mfma_f8_32_probe.hip's operand map (lane (row, h) holds K bytes [16h, 16h+16) and [32+16h, 32+16h+16)) makes
// the fragments of the fp16 row read -- chunks h, 2+h | 4+h, 6+h -- exactly the Kl8 and Kh8 operands; likewise the four
// 16-byte loads that fetched ql now fetch Ql8 | Qh8.  Per 64-key tile: 768 pipe cycles instead of 1024.
template <bool LOG2Q, bool VL, bool QK8 = false>
__global__ __launch_bounds__(256, VL ? 2 : 3) void attn16s_kernel(const f16* __restrict__ qkv, f16* __restrict__ ctx, int L, int H,
                                                         int causal, int nqt, int total, int per_xcd, bool hi8) {
  static_assert(!(QK8 && VL), "the e4m3 correction form exists for the long-row kernel only");
  fp8_saturate_mode();   // the context rows are converted with split8x4_sat
  typedef f16x8 vec8;
  typedef f16x4 vec4;
  typedef short i16x8 __attribute__((ext_vector_type(8)));
  constexpr int STAGE = VL ? 32768 : 24576;
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];   // 2 stages x (Kh 8K + Vh 8K + Kl 8K [+ Vl 8K])
  constexpr float LOG2E = 1.4426950408889634f;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  // XCD-aware numbering (see attn16x2_kernel): the query tiles of one (image, head) stream the same K/V rows; numbered
  // consecutively on ONE XCD they share its L2 (measured before: 4.7 GB fetched per launch for 1.1 GB of q/k/v)
  const int lin = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
  if (lin >= total || (int)(blockIdx.x >> 3) >= per_xcd) return;   // whole workgroup, before any barrier
  const int qt = lin % nqt, bh = lin / nqt;
  const int head = bh % H, b = bh / H;
  const int D = H * 64;
  const long ld = QK8 ? 5L * D : 6L * D;      // split row: [q k v hi | q k v lo], or [q k v hi | q8 | k8] (see above)
  const int LO = 3 * D;
  const f16* base = qkv + (long)b * L * ld + head * 64;
  const int q0 = qt * 128 + wave * 32;
  const int qi = q0 + r;
  const int qrow = qi < L ? qi : L - 1;

  vec8 qh[4], ql[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    qh[ks] = *(const vec8*)(base + (long)qrow * ld + 16 * ks + 8 * h);
    ql[ks] = *(const vec8*)(base + (long)qrow * ld + LO + 16 * ks + 8 * h);
  }

  const f16* ksrc[2];
  const f16* vsrc[2];
  int drow[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int pslot = (wave * 2 + j) * 64 + lane;
    const int row = pslot >> 3, sl = pslot & 7;
    drow[j] = row;
    ksrc[j] = base + (long)row * ld + D + (sl ^ xk(row)) * 8;
    vsrc[j] = base + (long)row * ld + 2 * D + (sl ^ xv(row)) * 8;
  }
  int koff[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) koff[ks] = r * 128 + (((2 * ks + h) ^ xk(r)) << 4);
  int voff[2];
  {
    const int g = lane >> 4, i = lane & 15, qq = i >> 2, pp = i & 3;
    const int row = 4 * (g >> 1) + qq;
#pragma unroll
    for (int db = 0; db < 2; ++db) {
      const int chunk = db * 4 + (g & 1) * 2 + (pp >> 1);
      voff[db] = 8192 + row * 128 + ((chunk ^ xv(row)) << 4) + (pp & 1) * 8;
    }
  }

  int last_q = qt * 128 + 127;
  if (last_q > L - 1) last_q = L - 1;
  const int nkt = causal ? (last_q / 64 + 1) : ((L + 63) / 64);

  // K / V pieces as buffer_load ... lds (round 4): per-lane 32-bit offsets fixed for the whole kernel, the tile advance in
  // the SCALAR offset -- no 64-bit VALU address arithmetic per piece (~26 VALU instructions per tile before).  The base
  // goes through unsigned halves: readfirstlane returns int, and OR-ing a negative low half into the 64-bit value
  // sign-extends it (a descriptor base of 0xffff... whenever bit 31 of the address is set: the intermittent fault of the
  // first version of this change).  Rows past the end of the image re-read row L-1 (their keys are masked): per-lane
  // arithmetic only in the one tile that has them.
  int kvo[2], vvo[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    kvo[j] = (int)((ksrc[j] - base) * 2);
    vvo[j] = (int)((vsrc[j] - base) * 2);
  }
  __amdgpu_buffer_rsrc_t rs;
  {
    const unsigned long long ub = (unsigned long long)base;
    const unsigned blo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)ub);
    const unsigned bhi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(ub >> 32));
    const f16* ubase = (const f16*)(((unsigned long long)bhi << 32) | (unsigned long long)blo);
    const int nrec = __builtin_amdgcn_readfirstlane((int)(((long)L * ld - head * 64) * 2));   // this image, from `base` on
    rs = __builtin_amdgcn_make_buffer_rsrc((void*)ubase, 0, nrec, 0x00020000);
  }
  const int ldb = __builtin_amdgcn_readfirstlane((int)(64 * ld * 2));
  auto stage = [&](int st, int kt) {
    char* dst = smem + st * STAGE + wave * 2048;
    const int so = kt * ldb;
    const bool tail = kt * 64 + 63 > L - 1;   // wave-uniform: only the last tile of a row of keys has rows past the end
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      int ko = kvo[j], vo = vvo[j];
      if (tail) {
        int over = kt * 64 + drow[j] - (L - 1);
        over = over > 0 ? over : 0;
        ko -= over * (int)ld * 2;
        vo -= over * (int)ld * 2;
      }
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(dst + j * 1024), 16, ko, so, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(dst + 8192 + j * 1024), 16, vo, so, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(dst + 16384 + j * 1024), 16, ko, so + LO * 2, 0, 0);
      if (VL) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(dst + 24576 + j * 1024), 16, vo, so + LO * 2, 0, 0);
    }
  };

  f32x16 o[2];
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int e = 0; e < 16; ++e) o[db][e] = 0.f;
  float m2 = 0.f, l = 0.f;

  auto tile = [&](const char* sb, int kt) {
    f32x16 s[2];
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
      for (int e = 0; e < 16; ++e) s[sub][e] = 0.f;   // (an inline constant as the first MFMA's C operand: no v_mov)
      if constexpr (QK8) {
        typedef int i32x4 __attribute__((ext_vector_type(4)));
        typedef int i32x8 __attribute__((ext_vector_type(8)));
        i32x4 k8[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) k8[ks] = *(const i32x4*)(sb + 16384 + koff[ks] + sub * 4096);
        const i32x8 kl8 = __builtin_shufflevector(k8[0], k8[1], 0, 1, 2, 3, 4, 5, 6, 7);
        const i32x8 kh8 = __builtin_shufflevector(k8[2], k8[3], 0, 1, 2, 3, 4, 5, 6, 7);
        const i32x8 ql8 = __builtin_shufflevector(__builtin_bit_cast(i32x4, ql[0]), __builtin_bit_cast(i32x4, ql[1]), 0, 1, 2, 3, 4, 5, 6, 7);
        const i32x8 qh8 = __builtin_shufflevector(__builtin_bit_cast(i32x4, ql[2]), __builtin_bit_cast(i32x4, ql[3]), 0, 1, 2, 3, 4, 5, 6, 7);
        constexpr int S_LO = (127 - SPLIT8_ACT_LO_EXP) * 0x01010101, S_ONE = 127 * 0x01010101;   // e8m0 block scales
        // the fp16 product first: its chain starts from the inline constant 0 (the scaled MFMA wants C in registers)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const vec8 ah = *(const vec8*)(sb + koff[ks] + sub * 4096);
          s[sub] = Elem<f16>::mma32(ah, qh[ks], s[sub]);
        }
        s[sub] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(kl8, qh8, s[sub], 0, 0, 0, S_LO, 0, S_ONE);
        s[sub] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(kh8, ql8, s[sub], 0, 0, 0, S_ONE, 0, S_LO);
      } else {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const vec8 ah = *(const vec8*)(sb + koff[ks] + sub * 4096);
        const vec8 al = *(const vec8*)(sb + 16384 + koff[ks] + sub * 4096);
        s[sub] = Elem<f16>::mma32(al, qh[ks], s[sub]);
        s[sub] = Elem<f16>::mma32(ah, ql[ks], s[sub]);
        s[sub] = Elem<f16>::mma32(ah, qh[ks], s[sub]);
      }
      }
      if (!LOG2Q) {
#pragma unroll
        for (int e = 0; e < 16; ++e) s[sub][e] *= LOG2E;
      }
    }
    // Round 4: the scores stay RAW (relative to zero) until the exponent.  Before, every chain started from -m (32 v_mov per
    // tile to re-initialise the accumulators) and a row whose maximum grew paid 32 more v_sub to re-base them; now the
    // reference is subtracted once, where the exponent needs it (32 v_sub per tile, always), and a new maximum only
    // rescales o and l.  Exact re-basing as before: the largest probability of a row is 2^0 when it is formed.
    const int k0 = kt * 64;
    const bool need_mask = (k0 + 64 > L) || (causal && (k0 + 63 > q0));
    if (need_mask) {
#pragma unroll
      for (int sub = 0; sub < 2; ++sub)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          int key = k0 + sub * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
          bool dead = (key >= L) || (causal && key > qi);
          s[sub][e] = dead ? -INFINITY : s[sub][e];
        }
    }
    float mt = s[0][0];
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
      for (int e = 0; e < 16; ++e) mt = fmaxf(mt, s[sub][e]);
    {   // the other 32-lane half's maximum: v_permlane32_swap (VALU) instead of ds_bpermute (an LDS round trip on the
        // critical path); inline asm for the reason given at attn16x2_kernel's xhalf_max
      float b2 = mt;
      asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(mt), "+v"(b2));
      mt = fmaxf(mt, b2);
    }
    const bool first = kt == 0;
    if (first || __any(mt > m2)) {
      const float mn = first ? mt : fmaxf(mt, m2);
      const float alpha = first ? 1.f : __builtin_amdgcn_exp2f(m2 - mn);   // tile 0: l and o are still zero
      m2 = mn;
      l *= alpha;
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[db][e] *= alpha;
    }
    float rs = 0.f;
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        float pv = __builtin_amdgcn_exp2f(s[sub][e] - m2);
        s[sub][e] = pv;
        rs += pv;
      }
    {
      float b2 = rs;
      asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(rs), "+v"(b2));
      rs += b2;
    }
    l += rs;

#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        vec8 pf;
#pragma unroll
        for (int j = 0; j < 8; ++j) pf[j] = (f16)s[sub][8 * s2 + j];
#pragma unroll
        for (int db = 0; db < 2; ++db) {
          const char* vp = sb + voff[db] + (sub * 32 + 16 * s2) * 128;
#pragma unroll
          for (int part = 0; part < (VL ? 2 : 1); ++part) {   // Vh, then (VL) Vl, 16 KiB further
            i16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)(vp + part * 16384));
            i16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)(vp + part * 16384 + 8 * 128));
            i16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            // accumulator tied in place: through the builtin hipcc gave these MFMAs destination tuples different from o
            // and copied all 32 registers of o back once per tile (16 v_mov_b64; tools' ISA histogram, round 4).  The
            // s_nop covers the VALU-write (pf: v_cvt_pk) -> MFMA-read distance the compiler can no longer see.
            asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(o[db]) : "v"(both), "v"(pf));
          }
        }
      }
  };

  stage(0, 0);
  wait_vm0();
  __syncthreads();
  // ONE tile per loop trip, the stage chosen at run time: with the two-tile body hipcc kept o in different register
  // tuples in the two halves and copied all 32 registers across once per tile
#pragma unroll 1
  for (int kt = 0; kt < nkt; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nkt) stage(cur ^ 1, kt + 1);
    tile(smem + cur * STAGE, kt);
    wait_vm0();
    __syncthreads();
  }

  asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");   // the last P.V MFMAs (inline asm) have written o before VALU reads it
  // Epilogue (round 4): a lane holds 4-value groups of ONE query row, so storing straight from the accumulator layout is
  // 24 instructions per wave of 4 or 8 scattered bytes per lane (16 bytes per row and instruction); skipping a quarter of
  // them (the hi8 plane) was worth 6.5 % of the kernel, which priced the whole tail at ~20 %.  The rows go through the
  // (now idle) LDS instead: per wave 32 rows x [hi 128 B | lo8 64 B | hi8 64 B] at a 272-byte stride, read back as 16-byte
  // chunks so that one store instruction covers 4 rows with whole 128- / 64-byte segments (8 instructions per wave).
  {
    const float inv = 1.0f / l;
    char* st = smem + wave * (32 * 272);
    char* srow = st + r * 272;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int gi = 0; gi < 4; ++gi) {
        const float vv[4] = {o[db][4 * gi] * inv, o[db][4 * gi + 1] * inv, o[db][4 * gi + 2] * inv,
                             o[db][4 * gi + 3] * inv};
        vec4 vh;
        uint32_t l8, h8;
        split8x4_sat(vv, vh, l8, h8);
        const int col = db * 32 + 8 * gi + 4 * h;
        *(vec4*)(srow + col * 2) = vh;
        *(uint32_t*)(srow + 128 + col) = l8;
        *(uint32_t*)(srow + 192 + col) = h8;
      }
    char* cbase = (char*)ctx + ((long)b * L + q0) * 4 * D;
    const int chunk = lane & 15;
    // byte offset of this lane's chunk inside its row: hi plane | lo8 plane (2D) | hi8 plane (3D)
    const int coff = chunk < 8 ? head * 128 + chunk * 16
                               : (chunk < 12 ? 2 * D + head * 64 + (chunk - 8) * 16 : 3 * D + head * 64 + (chunk - 12) * 16);
    const bool live_chunk = hi8 || chunk < 12;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int row = it * 4 + (lane >> 4);
      const u32x4 v = *(const u32x4*)(st + row * 272 + chunk * 16);
      if (live_chunk && q0 + row < L) *(u32x4*)(cbase + (long)row * 4 * D + coff) = v;
    }
  }
}

// ---------------------------------------------------------------------------
// Long-sequence variant (the visual tower, L = 1370): one wave owns TWO 32-row
// query blocks (64 rows), a workgroup 256 rows, so every K/V fragment read from
// LDS and every DMA'd byte feeds twice the MFMA work, and the two blocks give
// the scheduler two independent softmax chains.  K/V tiles go through a 4-stage
// ring (three tiles = 48 KiB in flight per workgroup, two workgroups per CU)
// with counted vmcnt waits.
//
// At head dim 64 the kernel is bound by VALU ISSUE, not by the matrix pipe: per
// 64-key tile a wave issues 32 MFMAs (1024 pipe cycles) and needs 64 v_exp_f32
// (8 issue cycles each).  Round-1 counters (profiles/r02a_attn_pmc.json): 11.7
// VALU instructions per MFMA, VALU issue busy 58 %, MFMA pipe busy 34 %.  Most of
// those instructions were not softmax arithmetic, so this version removes them:
//   * key masking (out-of-range / causal) was if-converted by hipcc into 64
//     v_cndmask + ~77 v_cmp per tile on EVERY tile: masking is now a real
//     wave-uniform branch, taken only by the tiles that need it (the last tile
//     of a row of keys, the diagonal tiles when causal);
//   * the S^T accumulators were re-initialised to -max with 64 v_mov per tile:
//     the chains now start from persistent 16-register tuples holding -max
//     (MFMA reads C from one tuple and writes D to another), rewritten only when
//     a row maximum grows;
//   * K/V DMA addresses cost 64-bit VALU arithmetic per instruction: the DMA is a
//     buffer_load ... lds with a per-lane 32-bit offset fixed for the whole
//     kernel and the tile advance in the SCALAR offset (no VALU per DMA);
//   * the cross-half exchanges of row maxima and row sums are v_permlane32_swap
//     (VALU, inline asm: the clang builtin is broken) instead of ds_bpermute_b32
//     (an LDS round trip on the critical path);
//   * the ring is walked with a run-time stage index (6 v_add per tile) instead of
//     a 4x unrolled body.
// Workgroups are numbered so that the query blocks of one (image, head) -- which
// stream the same K/V rows -- run on the same XCD at the same time and share its
// L2 (before: 2.3 GB fetched per launch for 0.54 GB of q/k/v).
#ifdef AACLIP_MEASURE
__device__ unsigned long long g_attn_passes[4];   // measurement library only: tile passes by kind (read_attn_passes)
// -DATTN_STAMP builds of the measurement library: s_memtime cycles summed per segment of the tile loop, waves 0 and 1
// of each workgroup: [0] wait + barrier, [1] DMA issue, [2] B1, [3] B2, [4] check 0 (+ rare paths), [5] B3, [6] check 1,
// [7] B4 + sums, [8] tiles
__device__ unsigned long long g_attn_stamp[16];   // [9..12]: the DMA instructions of a stage, one by one
#endif
#if defined(AACLIP_MEASURE) && defined(ATTN_STAMP)
#define STAMP_DECL unsigned long long st_acc[13] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_t = 0;
#define STAMP_START { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_t) :: "memory"); }
#define STAMP(i) { unsigned long long st_u; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_u) :: "memory"); st_acc[i] += st_u - st_t; st_t = st_u; }
#else
#define STAMP_DECL
#define STAMP_START
#define STAMP(i)
#endif
template <typename T, bool LOG2Q, int NW, int PD>
__global__ __launch_bounds__(NW * 64, 2) void attn16x2_kernel(const T* __restrict__ qkv, T* __restrict__ ctx, int L, int H,
                                                          int causal, int nqt, int total, int per_xcd) {
  typedef typename Elem<T>::vec8 vec8;
  typedef typename Elem<T>::vec4 vec4;
  typedef short i16x8 __attribute__((ext_vector_type(8)));
  __shared__ __attribute__((aligned(16))) char smem[65536];  // 4 stages x (K 8K + V 8K)
  constexpr float LOG2E = 1.4426950408889634f;
  constexpr float P_LIMIT = 32768.f;   // bound of a lane's 16-key partial row sum on the fast path (f16 max 65504)

  // XCD-aware numbering: workgroup ids are dealt round-robin over the 8 XCDs, so ids with equal id % 8 share an L2;
  // each XCD walks a contiguous range of (image, head, query block) with the query block fastest
  const int lin = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
  if (lin >= total || (int)(blockIdx.x >> 3) >= per_xcd) return;   // whole workgroup, before any barrier
  const int qt = lin % nqt, bh = lin / nqt;
  const int head = bh % H, b = bh / H;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int D = H * 64;
  const long ld = 3L * D;
  const T* base = qkv + (long)b * L * ld + head * 64;
  const int q0 = qt * (NW * 64) + wave * 64;
  const bool active = q0 < L;   // wave-uniform: idle waves only feed the ring and the barriers

  vec8 qf[2][4];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    int qrow = q0 + qb * 32 + r;
    qrow = qrow < L ? qrow : L - 1;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[qb][ks] = *(const vec8*)(base + (long)qrow * ld + 16 * ks + 8 * h);
  }
  // plain aaclip_attention contract (LOG2Q = false): q carries head_dim^-1/2 only and the scores are moved to log2
  // units in fp32 AFTER the MFMA chain (POSTSCALE: 64 v_fma per tile).  Multiplying the 16-bit q by log2(e) instead
  // would round it a second time: |d score| = 2^-11 |score| in f16, i.e. 2-4 % on a probability at |score| = 40 --
  // tools/stress_attention.py catches that.  The block path (LOG2Q) folds the factor into the QKV epilogue BEFORE the
  // one rounding to 16 bits and pays nothing here.
  constexpr bool POSTSCALE = !LOG2Q;

  // K/V DMA: buffer_load ... lds issued from an asm statement (dma16).  Descriptor base = this (image, head)'s q
  // column block, built from readfirstlane'd words (provably wave-uniform), per-lane byte offsets fixed for the whole
  // kernel, tile advance in the scalar offset.  Why asm: hipcc orders every LDS read behind ALL LDS-DMA it knows to
  // be pending (s_waitcnt vmcnt(0) in front of the V reads of every tile: the ring was drained once per tile);
  // DMA it does not see is ordered by the counted waits + barrier of the ring below, as intended.
  const unsigned long long ubase = (unsigned long long)base;
  u32x4 rs;
  rs[0] = __builtin_amdgcn_readfirstlane((unsigned)ubase);
  rs[1] = __builtin_amdgcn_readfirstlane((unsigned)(ubase >> 32)) & 0xFFFFu;   // stride 0
  rs[2] = 0x7FFFFFF0u;                                                          // num_records: no clamping wanted
  rs[3] = 0x00020000u;
  const int ldb = (int)(ld * sizeof(T));   // bytes per token row (host checks L * ldb < 2^31)
  constexpr int NJ = 8 / NW;   // DMA instructions per wave, tile and operand: 512 slots of 16 B per 64 x 128 B tile
  int kvo[NJ], vvo[NJ], drow[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int pslot = (wave * NJ + j) * 64 + lane;
    const int row = pslot >> 3, sl = pslot & 7;
    drow[j] = row;
    kvo[j] = row * ldb + (D + (sl ^ xk(row)) * 8) * (int)sizeof(T);
    vvo[j] = row * ldb + (2 * D + (sl ^ xv(row)) * 8) * (int)sizeof(T);
  }
  int koff[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) koff[ks] = r * 128 + (((2 * ks + h) ^ xk(r)) << 4);
  int voff[2];
  {
    const int g = lane >> 4, i = lane & 15, qq = i >> 2, pp = i & 3;
    const int row = 4 * (g >> 1) + qq;
#pragma unroll
    for (int db = 0; db < 2; ++db) {
      const int chunk = db * 4 + (g & 1) * 2 + (pp >> 1);
      voff[db] = 8192 + row * 128 + ((chunk ^ xv(row)) << 4) + (pp & 1) * 8;
    }
  }

  int last_q = qt * (NW * 64) + NW * 64 - 1;
  if (last_q > L - 1) last_q = L - 1;
  const int nkt = causal ? (last_q / 64 + 1) : ((L + 63) / 64);

  STAMP_DECL
  const unsigned lds0 = (unsigned)(size_t)(lds_void*)smem + wave * (NJ * 1024);   // LDS byte address of this wave's DMA slots
  auto dma16 = [&](unsigned lds_addr, int voff_b, int soff_b) {
    unsigned keep;   // M0 (the DMA's LDS base) is compiler-reserved: save, set, use and restore it in ONE statement
    // (s_nop 4 first: a descriptor / scalar offset fresh from v_readfirstlane needs 5 wait states before VMEM reads it)
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_addr), "v"(voff_b), "s"(rs), "s"(soff_b) : "memory");
  };
  auto stage = [&](int st, int kt) {
    const unsigned dst = lds0 + st * 16384;
    const int so = __builtin_amdgcn_readfirstlane(kt * 64 * ldb);
    if (kt * 64 + 64 <= L) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        dma16(dst + j * 1024, kvo[j], so);
        STAMP(9 + 2 * j)
        dma16(dst + 8192 + j * 1024, vvo[j], so);
        STAMP(10 + 2 * j)
      }
    } else {   // last tile of the key axis: rows beyond L-1 re-read row L-1 (finite data; their scores are masked)
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        int over = kt * 64 + drow[j] - (L - 1);
        over = over > 0 ? over : 0;
        dma16(dst + j * 1024, kvo[j] - over * ldb, so);
        dma16(dst + 8192 + j * 1024, vvo[j] - over * ldb, so);
      }
    }
  };

  f32x16 o[2][2];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb)
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int e = 0; e < 16; ++e) o[qb][db][e] = 0.f;
  float m2[2] = {0.f, 0.f}, l[2] = {0.f, 0.f};
  // start values of the S^T accumulator chains: every element = -(the row's reference point m, log2 units).
  // The empty asm makes the tuples opaque, so they stay 16 live registers instead of 16 v_mov per chain per tile.
  f32x16 cinit[2];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
#pragma unroll
    for (int e = 0; e < 16; ++e) cinit[qb][e] = 0.f;
    asm volatile("" : "+v"(cinit[qb]));
  }

  // Exchange between the two 32-lane halves by v_permlane32_swap (swaps lanes 32..63 of its first operand with lanes
  // 0..31 of its second): fed the same value twice it leaves {lower half's value in every lane, upper half's value in
  // every lane}.  Inline asm because clang's __builtin_amdgcn_permlane32_swap (ROCm 7.2) returns element 0 of the
  // intrinsic's result pair for BOTH vector elements (extractvalue ..., 0 twice in the IR): max / sum over ONE half,
  // silently.  s_nop 1: a VALU-written VGPR needs two wait states before a permlane reads it.
  auto xhalf_max = [](float a) {   // max over the two 32-lane halves, in every lane
    float b = a;
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return fmaxf(a, b);
  };
  auto xhalf_sum = [](float a) {
    float b = a;
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a + b;
  };

  // ---- tile pieces.  A tile = two 32-key sub-tiles; s[qb][sub] holds S'^T = K . Q^T - m (log2 units), later 2^that.
  // 8 MFMAs: the chains of sub-tile `sub` for both query blocks, started from the -m tuples
  auto chain = [&](const char* sb, int sub, f32x16 (&s)[2][2]) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const vec8 kf = *(const vec8*)(sb + koff[ks] + sub * 4096);
      s[0][sub] = Elem<T>::mma32(kf, qf[0][ks], ks == 0 ? cinit[0] : s[0][sub]);
      s[1][sub] = Elem<T>::mma32(kf, qf[1][ks], ks == 0 ? cinit[1] : s[1][sub]);
    }
  };
  // dead keys (beyond L, or above the diagonal when causal) -> -inf.  Key index of element e: k0 + 32 sub + c_e + 4 h
  // with c_e = (e & 3) + 8 (e >> 2); dead <=> c_e >= thr for one per-lane threshold, so the compares take c_e as an
  // inline constant.  Only called under the wave-uniform need_mask branch.
  auto mask_sub = [&](int kt, int sub, f32x16 (&s)[2][2]) {
    asm volatile("" ::: "memory");   // keeps hipcc from if-converting the caller's branch into selects on every tile
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      const int kb = kt * 64 + sub * 32 + 4 * h;
      int thr = L - kb;
      if (causal) thr = min(thr, q0 + qb * 32 + r + 1 - kb);
      asm volatile("" : "+v"(thr));   // defined HERE: hipcc otherwise hoists the compares onto the path of every tile
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int ce = (e & 3) + 8 * (e >> 2);
        s[qb][sub][e] = (ce >= thr) ? -INFINITY : s[qb][sub][e];
      }
    }
  };
  // p = 2^s in place for one sub-tile; rs[qb] = this lane's sum over its 16 keys of the sub-tile
  auto exp_sub = [&](int sub, f32x16 (&s)[2][2], float (&rs)[2]) {
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      float ra = 0.f, rb = 0.f;
#pragma unroll
      for (int e = 0; e < 16; e += 2) {
        float xa = s[qb][sub][e], xb = s[qb][sub][e + 1];
        if (POSTSCALE) {
          xa = fmaf(xa, LOG2E, -m2[qb]);
          xb = fmaf(xb, LOG2E, -m2[qb]);
        }
        const float pa = __builtin_amdgcn_exp2f(xa), pb = __builtin_amdgcn_exp2f(xb);
        s[qb][sub][e] = pa;
        s[qb][sub][e + 1] = pb;
        ra += pa;
        rb += pb;
      }
      rs[qb] = ra + rb;
    }
  };
  // Exact treatment of sub-tile `sub` (rare, out of line): scores recomputed from the LDS tile, true row maxima, the
  // reference point m advanced, everything that is still expressed against the old m re-based: O, l, the pending
  // partial sums of sub-tile 0 (`pend`, when sub == 1), the scores of sub-tile 1 that are already computed but not
  // yet exponentiated (when sub == 0).  Then the sub-tile is exponentiated again.
  auto rebase = [&](const char* sb, int kt, int sub, bool need_mask, f32x16 (&s)[2][2], float (&rs)[2], float* pend) {
    asm volatile("" ::: "memory");
    chain(sb, sub, s);
    if (need_mask) mask_sub(kt, sub, s);
    const bool first = kt == 0 && sub == 0;
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      float a = s[qb][sub][0];
#pragma unroll
      for (int e = 1; e < 16; ++e) a = fmaxf(a, s[qb][sub][e]);
      if (POSTSCALE) a = fmaf(a, LOG2E, -m2[qb]);
      const float mt = xhalf_max(a);
      // very first sub-tile: m = its maximum (may be below 0); later m only ever grows.  A row whose every key so
      // far is masked (-inf) keeps m where it is.
      float delta = first ? mt : fmaxf(mt, 0.f);
      delta = delta == -INFINITY ? 0.f : delta;
      // very first sub-tile: l and O are still zero and must stay so (exp2(-mt) is +inf when every score of the
      // sub-tile is below about -128 in log2 units, and 0 * inf = NaN for the whole row)
      const float alpha = first ? 1.f : __builtin_amdgcn_exp2f(-delta);
      m2[qb] += delta;
      l[qb] *= alpha;
      if (pend) pend[qb] *= alpha;
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[qb][db][e] *= alpha;
      if (!POSTSCALE) {
#pragma unroll
        for (int e = 0; e < 16; ++e) s[qb][sub][e] -= delta;
        if (sub == 0) {
#pragma unroll
          for (int e = 0; e < 16; ++e) s[qb][1][e] -= delta;
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) cinit[qb][e] = -m2[qb];
        asm volatile("" : "+v"(cinit[qb]));
      }
    }
    exp_sub(sub, s, rs);
  };
  auto vread = [&](const char* sb, int sub, int s2, int db) {   // V^T fragment: 32 d x 16 keys, transposed LDS read
    const char* vp = sb + voff[db] + (sub * 32 + 16 * s2) * 128;
    i16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)(vp));
    i16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)(vp + 8 * 128));
    i16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(vec8, both);
  };
  auto kread = [&](const char* sb, int sub, int ks) { return *(const vec8*)(sb + koff[ks] + sub * 4096); };
  auto pcvt = [&](const f32x16& p, int s2) {   // 8 probabilities of one (query block, 16-key step) -> MFMA B operand
    vec8 pf;
#pragma unroll
    for (int j = 0; j < 8; ++j) pf[j] = from_float<T>(p[8 * s2 + j]);
    return pf;
  };
  // 2^s in place for elements [4c, 4c+4) of sub-tile `sub`, both query blocks (8 v_exp), sums into acc[qb][0..1]
  auto exp_chunk = [&](int sub, int c, f32x16 (&s)[2][2], float (&acc)[2][2]) {
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
      for (int e = 4 * c; e < 4 * c + 4; ++e) {
        float x = s[qb][sub][e];
        if (POSTSCALE) x = fmaf(x, LOG2E, -m2[qb]);
        const float pe = __builtin_amdgcn_exp2f(x);
        s[qb][sub][e] = pe;
        acc[qb][e & 1] += pe;
      }
  };
#define SB0 __builtin_amdgcn_sched_barrier(0)

  // One key tile, software-pipelined at sub-tile granularity INSIDE the wave: on this chip VALU work hides in the
  // shadow of the SAME wave's MFMAs, hardly in its SIMD partner's, so the MFMAs of one step are
  // issued between the exponentials of the step before it, in chunks of 2 MFMAs + 8 v_exp + 8 adds pinned by
  // sched_barrier (hipcc's own order clusters the MFMAs; sched_group_barrier patterns were not honoured):
  //   B1  S(sub 0) chains                                               8 MFMA
  //   B2  S(sub 1) chains  ||  2^S(sub 0), its partial row sums         4 x (2 MFMA, 8 v_exp, 8 add)
  //   B3  O += V0 . P0     ||  2^S(sub 1), partial sums, P conversions  4 x (2 MFMA, 8 v_exp, 8 add, 4 cvt)
  //   B4  O += V1 . P1     ||  P conversions, l += sums                 4 x (2 MFMA, 4 cvt)
  // The reference point m of a row (cinit = -m) is NOT advanced tile by tile: softmax does not care which m numerator
  // and denominator share, so the fast path exponentiates against the m the row already has and never computes a row
  // maximum.  The stored probabilities must stay convertible to 16 bits, which the row sums -- needed anyway --
  // police: every p >= 0, so a lane's partial sum <= P_LIMIT bounds each of its p.  A sub-tile that breaks the bound,
  // or produces a non-finite sum (a score more than 2^127 above m), is redone exactly BEFORE its probabilities are
  // converted or multiplied into O (rebase).  The very first sub-tile of a row always takes the exact path (m starts
  // at 0, not at a maximum).  With the synthetic tower 0.9 % of the sub-tiles are redone.
  auto tile = [&](const char* sb, int kt, bool need_mask) {
    f32x16 s[2][2];
    float rs0[2], rs1[2];
    // ---- B1
    {
      vec8 kf = kread(sb, 0, 0);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const vec8 kn = ks < 3 ? kread(sb, 0, ks + 1) : kread(sb, 1, 0);
        s[0][0] = Elem<T>::mma32(kf, qf[0][ks], ks == 0 ? cinit[0] : s[0][0]);
        s[1][0] = Elem<T>::mma32(kf, qf[1][ks], ks == 0 ? cinit[1] : s[1][0]);
        kf = kn;
      }
      if (need_mask) mask_sub(kt, 0, s);
      SB0;
      STAMP(2)
      // ---- B2
      float acc[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        vec8 kn = kf;
        if (ks < 3) kn = kread(sb, 1, ks + 1);
        s[0][1] = Elem<T>::mma32(kf, qf[0][ks], ks == 0 ? cinit[0] : s[0][1]);
        s[1][1] = Elem<T>::mma32(kf, qf[1][ks], ks == 0 ? cinit[1] : s[1][1]);
        exp_chunk(0, ks, s, acc);
        kf = kn;
        SB0;
      }
      rs0[0] = acc[0][0] + acc[0][1];
      rs0[1] = acc[1][0] + acc[1][1];
    }
    // (the ballot is evaluated on every tile, tile 0 included: written as `kt == 0 || ...` hipcc sees that tile 0 does
    //  not need the exponentials above and sinks them out of B2, away from the MFMAs they are meant to hide behind)
    STAMP(3)
    const bool bad0 = __any(!(fmaxf(rs0[0], rs0[1]) <= P_LIMIT));
    if ((kt == 0) | bad0) rebase(sb, kt, 0, need_mask, s, rs0, nullptr);
    if (need_mask) mask_sub(kt, 1, s);
    SB0;
    STAMP(4)
    // ---- B3   (chunk g: 16-key step s2 = g >> 1 of sub-tile 0, query block qb = g & 1)
    vec8 v1a[2];   // first V^T fragments of sub-tile 1, prefetched in B3 for B4
    {
      vec8 va[2] = {vread(sb, 0, 0, 0), vread(sb, 0, 0, 1)};
      vec8 vb[2];
      vec8 pf = pcvt(s[0][0], 0);
      float acc[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
      SB0;
      vb[0] = vread(sb, 0, 1, 0);                                // g = 0
      vb[1] = vread(sb, 0, 1, 1);
      o[0][0] = Elem<T>::mma32(va[0], pf, o[0][0]);
      o[0][1] = Elem<T>::mma32(va[1], pf, o[0][1]);
      pf = pcvt(s[1][0], 0);
      exp_chunk(1, 0, s, acc);
      SB0;
      o[1][0] = Elem<T>::mma32(va[0], pf, o[1][0]);              // g = 1
      o[1][1] = Elem<T>::mma32(va[1], pf, o[1][1]);
      pf = pcvt(s[0][0], 1);
      exp_chunk(1, 1, s, acc);
      SB0;
      v1a[0] = vread(sb, 1, 0, 0);                               // g = 2
      v1a[1] = vread(sb, 1, 0, 1);
      o[0][0] = Elem<T>::mma32(vb[0], pf, o[0][0]);
      o[0][1] = Elem<T>::mma32(vb[1], pf, o[0][1]);
      pf = pcvt(s[1][0], 1);
      exp_chunk(1, 2, s, acc);
      SB0;
      o[1][0] = Elem<T>::mma32(vb[0], pf, o[1][0]);              // g = 3
      o[1][1] = Elem<T>::mma32(vb[1], pf, o[1][1]);
      exp_chunk(1, 3, s, acc);
      SB0;
      rs1[0] = acc[0][0] + acc[0][1];
      rs1[1] = acc[1][0] + acc[1][1];
    }
    STAMP(5)
    const bool bad1 = __any(!(fmaxf(rs1[0], rs1[1]) <= P_LIMIT));
    if (bad1) rebase(sb, kt, 1, need_mask, s, rs1, rs0);
    SB0;
    STAMP(6)
    // ---- B4
    {
      vec8 v1b[2];
      vec8 pf = pcvt(s[0][1], 0);
      v1b[0] = vread(sb, 1, 1, 0);                               // g = 0
      v1b[1] = vread(sb, 1, 1, 1);
      o[0][0] = Elem<T>::mma32(v1a[0], pf, o[0][0]);
      o[0][1] = Elem<T>::mma32(v1a[1], pf, o[0][1]);
      pf = pcvt(s[1][1], 0);
      SB0;
      o[1][0] = Elem<T>::mma32(v1a[0], pf, o[1][0]);             // g = 1
      o[1][1] = Elem<T>::mma32(v1a[1], pf, o[1][1]);
      pf = pcvt(s[0][1], 1);
      SB0;
      o[0][0] = Elem<T>::mma32(v1b[0], pf, o[0][0]);             // g = 2
      o[0][1] = Elem<T>::mma32(v1b[1], pf, o[0][1]);
      pf = pcvt(s[1][1], 1);
      SB0;
      o[1][0] = Elem<T>::mma32(v1b[0], pf, o[1][0]);             // g = 3
      o[1][1] = Elem<T>::mma32(v1b[1], pf, o[1][1]);
    }
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) l[qb] += xhalf_sum(rs0[qb] + rs1[qb]);
    STAMP(7)
  };
#undef SB0

  // ring prologue: PD tiles in flight (PD = prefetch distance in tiles, 1..3; 4 LDS stages)
  stage(0, 0);
  if (PD > 1 && nkt > 1) stage(1, 1);
  if (PD > 2 && nkt > 2) stage(2, 2);
  // The query fragments were loaded by ordinary global loads hipcc counts; it does not count the DMA above.  Touch
  // them here so that its wait for them lands HERE (a vmcnt(0) that also covers the prologue DMA) and not inside
  // the loop, where a vmcnt(small) computed without the DMA in mind would drain the ring every tile.
#pragma unroll
  for (int qb = 0; qb < 2; ++qb)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) asm volatile("" : "+v"(qf[qb][ks]));
  STAMP_START
#pragma unroll 1
  for (int t = 0; t < nkt; ++t) {
    // tiles already staged behind tile t stay in flight: 2 * NJ DMA instructions per wave each
    int younger = nkt - 1 - t;
    younger = younger < PD - 1 ? younger : PD - 1;
#if defined(AACLIP_MEASURE) && defined(ATTN_NODMA)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
    if (younger >= 2) { if (NJ == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
    else if (younger == 1) { if (NJ == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    __builtin_amdgcn_s_barrier();
    STAMP(0)
#ifdef ATTN_DMA_PRIO
    __builtin_amdgcn_s_setprio(3);
#endif
#if defined(AACLIP_MEASURE) && defined(ATTN_NODMA)
    if (t + PD < nkt && t < 1) stage((t + PD) & 3, t + PD);   // timing ablation (WRONG RESULTS): the ring is filled once
#else
    if (t + PD < nkt) stage((t + PD) & 3, t + PD);
#endif
#ifdef ATTN_DMA_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    STAMP(1)
    if (active) {
      const char* sb = smem + (t & 3) * 16384;
      const int k0 = t * 64;
      const bool need_mask = (k0 + 64 > L) || (causal && (k0 + 63 > q0));   // wave-uniform
      tile(sb, t, need_mask);
    }
  }

#if defined(AACLIP_MEASURE) && defined(ATTN_STAMP)
  if (active && (wave & 3) < 2 && lane == 0) {
    for (int i = 0; i < 8; ++i) atomicAdd(&g_attn_stamp[i], st_acc[i]);
    for (int i = 9; i < 13; ++i) atomicAdd(&g_attn_stamp[i], st_acc[i]);
    atomicAdd(&g_attn_stamp[8], (unsigned long long)nkt);
  }
#endif
  if (active) {
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      const int qi = q0 + qb * 32 + r;
      if (qi < L) {
        const float inv = 1.0f / l[qb];
        T* dst = ctx + ((long)b * L + qi) * D + head * 64;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
          for (int gi = 0; gi < 4; ++gi) {
            vec4 v;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = from_float<T>(o[qb][db][4 * gi + j] * inv);
            *(vec4*)(dst + db * 32 + 8 * gi + 4 * h) = v;
          }
      }
    }
  }
}

#ifdef AACLIP_MEASURE
// ---------------------------------------------------------------------------
// Unit-pipelined long-sequence kernel -- an EXPERIMENT, measurement library only (variant 6; tools/attn_ab.py).
//
// Stamps of attn16x2_kernel say that inside a wave the MFMA time and the VALU time of a block ADD UP (B2 = 8 MFMAs +
// 32 v_exp + 32 adds takes 657 cycles = 256 + 259 + 130): a wave issues in order, two adjacent MFMAs serialise on the
// pipe and only the last MFMA of a chunk has VALU work in its shadow; the pure-MFMA blocks B1 / B4 have none.
// tools/mfma_valu_slot.hip prices the alternative: [1 MFMA ; <= 24 cycles of VALU] repeats at the MFMA rate (32.3
// cycles), 2 v_exp + v_cvt_pk + 2 v_add at 37 -- but v_dot2c_f32_f16 or v_pk_add_f32 in the slot WAIT for the MFMA (49).
//
// Here the work is cut into UNITS = (32-key sub-tile, 32-row query block): 4 S MFMAs (A), 16 exponentials + 8 packed
// conversions + 16 adds (E), 4 P.V MFMAs (P).  Unit u's E runs beside A(u+1) and P(u-1): every step is 8 MFMAs, each
// followed by 2 v_exp + 1 v_cvt_pk + 2 v_add, never two MFMAs back to back.  Consecutive units alternate between the
// wave's two query blocks, so
//   * a re-base of unit u (its query block's reference point moves) never touches scores that are already in flight:
//     A(u+1) belongs to the OTHER query block, A(u+2) is issued after the check;
//   * the K fragments of a sub-tile serve A of both query blocks, likewise the V^T fragments: all LDS reads of a
//     sub-tile sit in ONE step and are consumed a step later (no LDS latency on any MFMA's path);
//   * only two score tuples and two packed-P tuples are live, the row sums stay per lane until the end.
// The ring's wait + barrier for tile t+1 sits between the two sub-tiles of tile t (the last step of tile t multiplies
// K(t+1)); the four DMA pieces a wave owes a tile are issued one per step.
//
// RESULT: stamped cycles per tile per wave drop from 5.9 k to 4.9 k, and the wall time at B = 64
// is IDENTICAL to attn16x2_kernel's to three digits (0.643 vs 0.645 ms, also 0.50 vs 0.50 ms on all-zero operands):
// with every CU busy the chip is power-limited, a kernel that stalls less simply runs at a lower clock.  With few
// workgroups (B = 1: one per CU, full clock) this kernel is SLOWER (38 vs 32 us): more instructions on the path of a
// lone wave.  Correct on every attention test; not used by the product library.
template <typename T> struct Pair16;   // two 16-bit values in one register
template <> struct Pair16<f16> { typedef _Float16 type __attribute__((ext_vector_type(2))); };
template <> struct Pair16<bf16> { typedef __bf16 type __attribute__((ext_vector_type(2))); };
template <typename T, bool LOG2Q, int NW>
AACLIP_DEV void attn16u_body(char* smem, const T* __restrict__ qkv, T* __restrict__ ctx, int L, int H, int causal,
                             int b, int head, int qbase) {
  typedef typename Elem<T>::vec8 vec8;
  typedef typename Elem<T>::vec4 vec4;
  typedef typename Pair16<T>::type vec2;
  typedef short i16x8 __attribute__((ext_vector_type(8)));
  constexpr float LOG2E = 1.4426950408889634f;
  constexpr float P_LIMIT = 32768.f;
  constexpr bool POSTSCALE = !LOG2Q;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int D = H * 64;
  const long ld = 3L * D;
  const T* base = qkv + (long)b * L * ld + head * 64;
  const int q0 = qbase + wave * 64;
  const bool active = q0 < L;   // wave-uniform: idle waves only feed the ring and the barriers

  vec8 qf[2][4];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    int qrow = q0 + qb * 32 + r;
    qrow = qrow < L ? qrow : L - 1;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[qb][ks] = *(const vec8*)(base + (long)qrow * ld + 16 * ks + 8 * h);
  }

  // K/V DMA from inline asm (see attn16x2_body)
  const unsigned long long ubase = (unsigned long long)base;
  u32x4 rs;
  rs[0] = __builtin_amdgcn_readfirstlane((unsigned)ubase);
  rs[1] = __builtin_amdgcn_readfirstlane((unsigned)(ubase >> 32)) & 0xFFFFu;
  rs[2] = 0x7FFFFFF0u;
  rs[3] = 0x00020000u;
  const int ldb = (int)(ld * sizeof(T));
  constexpr int NJ = 8 / NW;
  int kvo[NJ], vvo[NJ], drow[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int pslot = (wave * NJ + j) * 64 + lane;
    const int row = pslot >> 3, sl = pslot & 7;
    drow[j] = row;
    kvo[j] = row * ldb + (D + (sl ^ xk(row)) * 8) * (int)sizeof(T);
    vvo[j] = row * ldb + (2 * D + (sl ^ xv(row)) * 8) * (int)sizeof(T);
  }
  int koff[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) koff[ks] = r * 128 + (((2 * ks + h) ^ xk(r)) << 4);
  int voff[2];
  {
    const int g = lane >> 4, i = lane & 15, qq = i >> 2, pp = i & 3;
    const int row = 4 * (g >> 1) + qq;
#pragma unroll
    for (int db = 0; db < 2; ++db) {
      const int chunk = db * 4 + (g & 1) * 2 + (pp >> 1);
      voff[db] = 8192 + row * 128 + ((chunk ^ xv(row)) << 4) + (pp & 1) * 8;
    }
  }
  int last_q = qbase + NW * 64 - 1;
  if (last_q > L - 1) last_q = L - 1;
  const int nkt = causal ? (last_q / 64 + 1) : ((L + 63) / 64);

  STAMP_DECL   // -DATTN_STAMP: [0] wait + barrier, [1] DMA issue, [2] step a, [3] check a, [4] step b, [5] check b
  const unsigned lds0 = (unsigned)(size_t)(lds_void*)smem + wave * (NJ * 1024);
  auto dma16 = [&](unsigned lds_addr, int voff_b, int soff_b) {
    unsigned keep;
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_addr), "v"(voff_b), "s"(rs), "s"(soff_b) : "memory");
  };
  auto stage = [&](int st, int kt) {
    const unsigned dst = lds0 + st * 16384;
    const int so = __builtin_amdgcn_readfirstlane(kt * 64 * ldb);
    if (kt * 64 + 64 <= L) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        dma16(dst + j * 1024, kvo[j], so);
        dma16(dst + 8192 + j * 1024, vvo[j], so);
      }
    } else {   // last tile of the key axis: rows beyond L-1 re-read row L-1 (finite data; their scores are masked)
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        int over = kt * 64 + drow[j] - (L - 1);
        over = over > 0 ? over : 0;
        dma16(dst + j * 1024, kvo[j] - over * ldb, so);
        dma16(dst + 8192 + j * 1024, vvo[j] - over * ldb, so);
      }
    }
  };

  // One DMA instruction ("piece": 1 KiB of K or V) of tile kt; which = 2 j + (0: K, 1: V).  The four pieces a wave owes
  // a tile are issued one per STEP, spread over the four steps after the ring barrier that frees the tile's stage:
  // issued back to back behind the barrier -- by all 8 waves of the CU at once -- they queued for 300 cycles EACH
  // (stamps: 1264 cycles per tile per wave), a quarter of the tile time with the wave's MFMAs waiting behind them.
  static_assert(NW == 4, "four DMA pieces per wave and tile, one per step");
  auto piece = [&](int kt, int which) {
    if (kt >= nkt) return;   // wave-uniform
    const int j = which >> 1;
    const unsigned dst = lds0 + (kt & 3) * 16384 + (which & 1) * 8192 + j * 1024;
    const int so = __builtin_amdgcn_readfirstlane(kt * 64 * ldb);
    int off = (which & 1) ? vvo[j] : kvo[j];
    if (kt * 64 + 64 > L) {   // last tile of the key axis: rows beyond L-1 re-read row L-1
      int over = kt * 64 + drow[j] - (L - 1);
      over = over > 0 ? over : 0;
      off -= over * ldb;
    }
    dma16(dst, off, so);
  };

  f32x16 o[2][2];
  float m2[2] = {0.f, 0.f}, ll[2] = {0.f, 0.f};   // ll: this LANE's part of the row sum (its 16 keys of every sub-tile)
  f32x16 cinit[2];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int e = 0; e < 16; ++e) o[qb][db][e] = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) cinit[qb][e] = 0.f;
    asm volatile("" : "+v"(cinit[qb]));
  }
  auto xhalf_max = [](float a) {
    float bb = a;
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(bb));
    return fmaxf(a, bb);
  };
  auto xhalf_sum = [](float a) {
    float bb = a;
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(bb));
    return a + bb;
  };
  auto kread = [&](const char* sb, int sub, int ks) { return *(const vec8*)(sb + koff[ks] + sub * 4096); };
  auto vread = [&](const char* sb, int sub, int s2, int db) {   // V^T fragment: 32 d x 16 keys, transposed LDS read
    const char* vp = sb + voff[db] + (sub * 32 + 16 * s2) * 128;
    i16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)(vp));
    i16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)(vp + 8 * 128));
    i16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(vec8, both);
  };
  // dead keys of one unit -> -inf (see mask_sub above); only called under the wave-uniform need_mask branch
  auto mask_unit = [&](int kt, int sub, int qb, f32x16& s) {
    asm volatile("" ::: "memory");
    const int kb = kt * 64 + sub * 32 + 4 * h;
    int thr = L - kb;
    if (causal) thr = min(thr, q0 + qb * 32 + r + 1 - kb);
    asm volatile("" : "+v"(thr));
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int ce = (e & 3) + 8 * (e >> 2);
      s[e] = (ce >= thr) ? -INFINITY : s[e];
    }
  };
  // (timing ablations of the measurement library, WRONG RESULTS: -DATTNU_NOEXP, -DATTNU_NOPV, -DATTNU_NOBAR; -DATTNU_ONEWG = one workgroup per CU)
#if defined(AACLIP_MEASURE) && defined(ATTNU_NOEXP)
#define ATTNU_EXP(x) ((x) * 0.001f)
#else
#define ATTNU_EXP(x) __builtin_amdgcn_exp2f(x)
#endif
#if defined(AACLIP_MEASURE) && defined(ATTNU_NOPV)
#define ATTNU_PV(acc, a, b) acc
#else
#define ATTNU_PV(acc, a, b) Elem<T>::mma32(a, b, acc)
#endif
  // E of a unit in 8 slots, skewed by one slot per dependent instruction so that nothing in a slot waits for a result
  // of the SAME slot (a v_cvt_pk right behind the two v_exp it packs stalls the wave for the transcendental latency;
  // stamped: 87 cycles per slot instead of ~32):
  //   slot i:  2^s of elements 2i, 2i+1  |  pack the pair of slot i-1 into P, add its two values to the row sums
  // EXP_TAIL finishes slot 7 after the last MFMA of the step.
#define EXP_E(S, QB, EV, i)                                       \
  {                                                               \
    float x0 = S[2 * (i)], x1 = S[2 * (i) + 1];                   \
    if (POSTSCALE) {                                              \
      x0 = fmaf(x0, LOG2E, -m2[QB]);                              \
      x1 = fmaf(x1, LOG2E, -m2[QB]);                              \
    }                                                             \
    EV[i][0] = ATTNU_EXP(x0);                                     \
    EV[i][1] = ATTNU_EXP(x1);                                     \
    asm volatile("" : "+v"(EV[i][0]), "+v"(EV[i][1]));   /* computed HERE: LLVM sinks pure ops to their first use */ \
  }
#define EXP_C(EV, PR, PKV, i)                                     \
  {                                                               \
    PR[i] = (vec2){from_float<T>(EV[i][0]), from_float<T>(EV[i][1])}; \
    asm volatile("" : "+v"(PR[i]));                               \
    PKV[(i) >> 2][2 * ((i) & 3)] = PR[i][0];                      \
    PKV[(i) >> 2][2 * ((i) & 3) + 1] = PR[i][1];                  \
  }
  // (row sums: plain fp32 adds of the unrounded values.  v_dot2c_f32_f16 on the packed pair -- or v_pk_add_f32 -- halves
  //  the instruction count but WAITS for the MFMA in flight: tools/mfma_valu_slot.hip, 49 cycles per slot against 37.)
#define EXP_D(EV, ACC, i) { ACC[0] += EV[i][0]; ACC[1] += EV[i][1]; }
#define EXP_SLOT(S, QB, PKV, ACC, i)                              \
  {                                                               \
    EXP_E(S, QB, ev, i)                                           \
    if ((i) >= 1) EXP_C(ev, prv, PKV, ((i) >= 1 ? (i) - 1 : 0))   \
    if ((i) >= 1) EXP_D(ev, ACC, ((i) >= 1 ? (i) - 1 : 0))        \
  }
#define EXP_TAIL(PKV, ACC) { EXP_C(ev, prv, PKV, 7) EXP_D(ev, ACC, 7) }
  // Exact treatment of one unit (rare, out of line): scores recomputed from the LDS tile, true row maximum, the query
  // block's reference point advanced, O and the row sum re-based, the unit exponentiated again.
  auto rebase_unit = [&](const char* sb, int kt, int sub, int qb, bool need_mask, f32x16& s, vec8 (&pk)[2], float& rsum) {
    asm volatile("" ::: "memory");
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) s = Elem<T>::mma32(kread(sb, sub, ks), qf[qb][ks], ks == 0 ? cinit[qb] : s);
    if (need_mask) mask_unit(kt, sub, qb, s);
    float a = s[0];
#pragma unroll
    for (int e = 1; e < 16; ++e) a = fmaxf(a, s[e]);
    if (POSTSCALE) a = fmaf(a, LOG2E, -m2[qb]);
    const float mt = xhalf_max(a);
    float delta = (kt == 0 && sub == 0) ? mt : fmaxf(mt, 0.f);   // very first unit: m = its maximum; later m only grows
    delta = delta == -INFINITY ? 0.f : delta;                     // every key so far masked: m stays
    const float alpha = (kt == 0 && sub == 0) ? 1.f : __builtin_amdgcn_exp2f(-delta);   // l, O still zero: no 0 * inf
    m2[qb] += delta;
    ll[qb] *= alpha;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int e = 0; e < 16; ++e) o[qb][db][e] *= alpha;
    if (!POSTSCALE) {
#pragma unroll
      for (int e = 0; e < 16; ++e) s[e] -= delta;
#pragma unroll
      for (int e = 0; e < 16; ++e) cinit[qb][e] = -m2[qb];
      asm volatile("" : "+v"(cinit[qb]));
    }
    float acc[2] = {0.f, 0.f};
    float ev[8][2];
    vec2 prv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) EXP_SLOT(s, qb, pk, acc, i)
    EXP_TAIL(pk, acc)
    rsum = acc[0] + acc[1];
  };
#define SB0 __builtin_amdgcn_sched_barrier(0)

  // pipeline state between steps
  f32x16 s0, s1;        // scores of the unit being exponentiated / of the unit after it
  vec8 kf[4];           // K fragments A(next unit) multiplies
  vec8 vf[2][2];        // V^T fragments [16-key step][d block] P(previous unit) multiplies
  vec8 pq0[2], pq1[2];  // packed P of the last unit of query block 0 / 1
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      pq1[i][j] = from_float<T>(0.f);
      pq0[i][j] = from_float<T>(0.f);
      vf[i][0][j] = from_float<T>(0.f);
      vf[i][1][j] = from_float<T>(0.f);
    }
  }

  // One sub-tile = two steps.  Entry: s0 = S(sigma, block 0), kf = K(sigma), vf = V(sigma - 1), pq1 = P(sigma - 1, block 1).
  //   step a:  E(sigma, 0)  ||  P(sigma-1, 1) = O1 += vf . pq1,  A(sigma, 1) = kf . Q1 -> s1;   vf <- V(sigma), kf <- K(sigma+1)
  //   step b:  E(sigma, 1)  ||  P(sigma, 0)   = O0 += vf . pq0,  A(sigma+1, 0) = kf . Q0 -> s0
  auto pair = [&](const char* sb, const char* sbn, int kt, int sub, int subn, bool need_mask, int dma_kt, int dma_first) {
    float rsum;
    STAMP_START
    // ---- step a
    {
      if (need_mask) mask_unit(kt, sub, 0, s0);
      float acc[2] = {0.f, 0.f};
      float ev[8][2];
      vec2 prv[8];
      SB0;
      o[1][0] = ATTNU_PV(o[1][0], vf[0][0], pq1[0]);
      vf[0][0] = vread(sb, sub, 0, 0);
      EXP_SLOT(s0, 0, pq0, acc, 0)
      SB0;
      s1 = Elem<T>::mma32(kf[0], qf[1][0], cinit[1]);
      kf[0] = kread(sbn, subn, 0);
      EXP_SLOT(s0, 0, pq0, acc, 1)
      SB0;
      o[1][1] = ATTNU_PV(o[1][1], vf[0][1], pq1[0]);
      vf[0][1] = vread(sb, sub, 0, 1);
      EXP_SLOT(s0, 0, pq0, acc, 2)
      SB0;
      s1 = Elem<T>::mma32(kf[1], qf[1][1], s1);
      kf[1] = kread(sbn, subn, 1);
      EXP_SLOT(s0, 0, pq0, acc, 3)
      SB0;
      piece(dma_kt, dma_first);
      SB0;
      o[1][0] = ATTNU_PV(o[1][0], vf[1][0], pq1[1]);
      vf[1][0] = vread(sb, sub, 1, 0);
      EXP_SLOT(s0, 0, pq0, acc, 4)
      SB0;
      s1 = Elem<T>::mma32(kf[2], qf[1][2], s1);
      kf[2] = kread(sbn, subn, 2);
      EXP_SLOT(s0, 0, pq0, acc, 5)
      SB0;
      o[1][1] = ATTNU_PV(o[1][1], vf[1][1], pq1[1]);
      vf[1][1] = vread(sb, sub, 1, 1);
      EXP_SLOT(s0, 0, pq0, acc, 6)
      SB0;
      s1 = Elem<T>::mma32(kf[3], qf[1][3], s1);
      kf[3] = kread(sbn, subn, 3);
      EXP_SLOT(s0, 0, pq0, acc, 7)
      SB0;
      EXP_TAIL(pq0, acc)
      rsum = acc[0] + acc[1];
    }
    STAMP(2)
    // (evaluated on every unit, the first included: see the note in attn16x2_body)
    const bool bad0 = __any(!(rsum <= P_LIMIT));
    if (((kt | sub) == 0) | bad0) rebase_unit(sb, kt, sub, 0, need_mask, s0, pq0, rsum);
    ll[0] += rsum;
    SB0;
    STAMP(3)
    // ---- step b
    {
      if (need_mask) mask_unit(kt, sub, 1, s1);
      float acc[2] = {0.f, 0.f};
      float ev[8][2];
      vec2 prv[8];
      SB0;
      o[0][0] = ATTNU_PV(o[0][0], vf[0][0], pq0[0]);
      EXP_SLOT(s1, 1, pq1, acc, 0)
      SB0;
      s0 = Elem<T>::mma32(kf[0], qf[0][0], cinit[0]);
      EXP_SLOT(s1, 1, pq1, acc, 1)
      SB0;
      o[0][1] = ATTNU_PV(o[0][1], vf[0][1], pq0[0]);
      EXP_SLOT(s1, 1, pq1, acc, 2)
      SB0;
      s0 = Elem<T>::mma32(kf[1], qf[0][1], s0);
      EXP_SLOT(s1, 1, pq1, acc, 3)
      SB0;
      piece(dma_kt, dma_first + 1);
      SB0;
      o[0][0] = ATTNU_PV(o[0][0], vf[1][0], pq0[1]);
      EXP_SLOT(s1, 1, pq1, acc, 4)
      SB0;
      s0 = Elem<T>::mma32(kf[2], qf[0][2], s0);
      EXP_SLOT(s1, 1, pq1, acc, 5)
      SB0;
      o[0][1] = ATTNU_PV(o[0][1], vf[1][1], pq0[1]);
      EXP_SLOT(s1, 1, pq1, acc, 6)
      SB0;
      s0 = Elem<T>::mma32(kf[3], qf[0][3], s0);
      EXP_SLOT(s1, 1, pq1, acc, 7)
      SB0;
      EXP_TAIL(pq1, acc)
      rsum = acc[0] + acc[1];
    }
    STAMP(4)
    const bool bad1 = __any(!(rsum <= P_LIMIT));
    if (((kt | sub) == 0) | bad1) rebase_unit(sb, kt, sub, 1, need_mask, s1, pq1, rsum);
    ll[1] += rsum;
    SB0;
    STAMP(5)
  };

  // ring prologue: tiles 0, 1 and half of tile 2 in flight, tile 0 confirmed.  Afterwards sub-tile (t, 0) issues pieces
  // 2, 3 of tile t+2 and sub-tile (t, 1) -- behind the barrier that frees the stage -- pieces 0, 1 of tile t+3.
  stage(0, 0);
  if (nkt > 1) stage(1, 1);
  piece(2, 0);
  piece(2, 1);
#pragma unroll
  for (int qb = 0; qb < 2; ++qb)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) asm volatile("" : "+v"(qf[qb][ks]));   // hipcc's wait for the q loads lands here
  if (nkt > 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if (nkt > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (active) {   // A(0, block 0), unpipelined
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      kf[ks] = kread(smem, 0, ks);
      s0 = Elem<T>::mma32(kf[ks], qf[0][ks], ks == 0 ? cinit[0] : s0);
    }
  }
#pragma unroll 1
  for (int t = 0; t < nkt; ++t) {
    const char* sb = smem + (t & 3) * 16384;
    const char* sbn = smem + ((t + 1) & 3) * 16384;
    const int k0 = t * 64;
    const bool need_mask = (k0 + 64 > L) || (causal && (k0 + 63 > q0));   // wave-uniform
    if (active) pair(sb, sb, t, 0, 1, need_mask, t + 2, 2);
    else { piece(t + 2, 2); piece(t + 2, 3); }
    STAMP_START
    if (t + 1 < nkt) {
      // tile t+1 must be complete before the second sub-tile (its last step multiplies K(t+1)); tile t+2 may stay in
      // flight; every wave is past tile t-1 here, so its stage takes tile t+3
      if (t + 2 < nkt) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#if !(defined(AACLIP_MEASURE) && defined(ATTNU_NOBAR))
      __builtin_amdgcn_s_barrier();
#endif
      STAMP(0)
      STAMP(1)
    }
    if (active) pair(sb, sbn, t, 1, 0, need_mask, t + 3, 0);
    else { piece(t + 3, 0); piece(t + 3, 1); }   // (after the last tile A(t+1, .) multiplies stale LDS: never used)
  }
#undef SB0
#undef EXP_SLOT
#undef EXP_TAIL
#undef EXP_E
#undef EXP_C
#undef EXP_D
#undef ATTNU_EXP
#undef ATTNU_PV
#if defined(AACLIP_MEASURE) && defined(ATTN_STAMP)
  if (active && (wave & 3) < 2 && lane == 0) {
    for (int i = 0; i < 6; ++i) atomicAdd(&g_attn_stamp[i], st_acc[i]);
    atomicAdd(&g_attn_stamp[8], (unsigned long long)nkt);
  }
#endif
  if (active) {
    // P of the very last unit
    o[1][0] = Elem<T>::mma32(vf[0][0], pq1[0], o[1][0]);
    o[1][1] = Elem<T>::mma32(vf[0][1], pq1[0], o[1][1]);
    o[1][0] = Elem<T>::mma32(vf[1][0], pq1[1], o[1][0]);
    o[1][1] = Elem<T>::mma32(vf[1][1], pq1[1], o[1][1]);
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      const float lsum = xhalf_sum(ll[qb]);
      const int qi = q0 + qb * 32 + r;
      if (qi < L) {
        const float inv = 1.0f / lsum;
        T* dst = ctx + ((long)b * L + qi) * D + head * 64;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
          for (int gi = 0; gi < 4; ++gi) {
            vec4 v;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = from_float<T>(o[qb][db][4 * gi + j] * inv);
            *(vec4*)(dst + db * 32 + 8 * gi + 4 * h) = v;
          }
      }
    }
  }
}

template <typename T, bool LOG2Q, int NW>
__global__ __launch_bounds__(NW * 64, 2) void attn16u_kernel(const T* __restrict__ qkv, T* __restrict__ ctx, int L, int H,
                                                         int causal, int nqt, int total, int per_xcd) {
#if defined(AACLIP_MEASURE) && defined(ATTNU_ONEWG)   // timing experiment: one workgroup per CU (one wave per SIMD)
  __shared__ __attribute__((aligned(16))) char smem[98304];
#else
  __shared__ __attribute__((aligned(16))) char smem[65536];  // 4 stages x (K 8K + V 8K)
#endif
  const int lin = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);   // XCD-aware numbering, see attn16x2_kernel
  if (lin >= total || (int)(blockIdx.x >> 3) >= per_xcd) return;
  const int qt = lin % nqt, bh = lin / nqt;
  attn16u_body<T, LOG2Q, NW>(smem, qkv, ctx, L, H, causal, bh / H, bh % H, qt * (NW * 64));
}

#endif  // AACLIP_MEASURE (unit-pipelined experiment)

#ifdef AACLIP_MEASURE   // software-pipelined variant: measured slower, kept for A/B runs (measurement library)
// ---------------------------------------------------------------------------
// Software-pipelined variant (32 query rows per wave, 4-stage K/V ring): the
// S^T MFMA chain of key tile j+1 is issued BEFORE the exponentials of tile j and
// the compiler is told (sched_group_barrier) to interleave one MFMA with a slice
// of the softmax VALU work, so the matrix pipe and the VALU run concurrently
// inside one wave instead of in turns (guide T15/T19).  The running maximum used
// for tile j+1's accumulator start is the one AFTER tile j's re-base, so the
// arithmetic is identical to the kernels above.
template <typename T, bool LOG2Q>
__global__ __launch_bounds__(256, 2) void attn16p_kernel(const T* __restrict__ qkv, T* __restrict__ ctx, int L, int H,
                                                         int causal) {
  typedef typename Elem<T>::vec8 vec8;
  typedef typename Elem<T>::vec4 vec4;
  typedef short i16x8 __attribute__((ext_vector_type(8)));
  __shared__ __attribute__((aligned(16))) char smem[65536];  // 4 stages x (K 8K + V 8K)
  constexpr float LOG2E = 1.4426950408889634f;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int qt = blockIdx.x, head = blockIdx.y, b = blockIdx.z;
  const int D = H * 64;
  const long ld = 3L * D;
  const T* base = qkv + (long)b * L * ld + head * 64;
  const int q0 = qt * 128 + wave * 32;
  const int qi = q0 + r;
  const int qrow = qi < L ? qi : L - 1;

  vec8 qf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const vec8*)(base + (long)qrow * ld + 16 * ks + 8 * h);

  const T* ksrc[2];
  const T* vsrc[2];
  int drow[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int pslot = (wave * 2 + j) * 64 + lane;
    const int row = pslot >> 3, sl = pslot & 7;
    drow[j] = row;
    ksrc[j] = base + (long)row * ld + D + (sl ^ xk(row)) * 8;
    vsrc[j] = base + (long)row * ld + 2 * D + (sl ^ xv(row)) * 8;
  }
  int koff[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) koff[ks] = r * 128 + (((2 * ks + h) ^ xk(r)) << 4);
  int voff[2];
  {
    const int g = lane >> 4, i = lane & 15, qq = i >> 2, pp = i & 3;
    const int row = 4 * (g >> 1) + qq;
#pragma unroll
    for (int db = 0; db < 2; ++db) {
      const int chunk = db * 4 + (g & 1) * 2 + (pp >> 1);
      voff[db] = 8192 + row * 128 + ((chunk ^ xv(row)) << 4) + (pp & 1) * 8;
    }
  }
  int last_q = qt * 128 + 127;
  if (last_q > L - 1) last_q = L - 1;
  const int nkt = causal ? (last_q / 64 + 1) : ((L + 63) / 64);

  auto stage = [&](int st, int kt) {
    char* dst = smem + st * 16384 + wave * 2048;
    const long step = (long)kt * 64 * ld;
    if (kt * 64 + 64 <= L) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        glds16(ksrc[j] + step, dst + j * 1024);
        glds16(vsrc[j] + step, dst + 8192 + j * 1024);
      }
    } else {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        int over = kt * 64 + drow[j] - (L - 1);
        over = over > 0 ? over : 0;
        glds16(ksrc[j] + step - (long)over * ld, dst + j * 1024);
        glds16(vsrc[j] + step - (long)over * ld, dst + 8192 + j * 1024);
      }
    }
  };

  f32x16 o[2];
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int e = 0; e < 16; ++e) o[db][e] = 0.f;
  float m2 = 0.f, l = 0.f;

  // mask (tail / causal) + scaling for the non-log2 contract + tile maximum
  auto finish_scores = [&](f32x16 (&s)[2], int kt) -> float {
    if (!LOG2Q) {
#pragma unroll
      for (int sub = 0; sub < 2; ++sub)
#pragma unroll
        for (int e = 0; e < 16; ++e) s[sub][e] = fmaf(s[sub][e], LOG2E, -m2);
    }
    const int k0 = kt * 64;
    const bool need_mask = (k0 + 64 > L) || (causal && (k0 + 63 > q0));
    if (need_mask) {
#pragma unroll
      for (int sub = 0; sub < 2; ++sub)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          int key = k0 + sub * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
          bool dead = (key >= L) || (causal && key > qi);
          s[sub][e] = dead ? -INFINITY : s[sub][e];
        }
    }
    float a = s[0][0], c = s[1][0];
#pragma unroll
    for (int e = 1; e < 16; ++e) {
      a = fmaxf(a, s[0][e]);
      c = fmaxf(c, s[1][e]);
    }
    a = fmaxf(a, c);
    return fmaxf(a, __shfl_xor(a, 32, 64));
  };

// S'^T(next) = K . Q^T - m2 : 8 LDS reads, then 8 MFMAs interleaved with the exponentials of the current tile
#define QK_EXP(SB, SNEXT, SCUR, WITH_QK)                                                      \
  {                                                                                           \
    vec8 kf[2][4];                                                                            \
    if (WITH_QK) {                                                                            \
      _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                                      \
        kf[0][ks] = *(const vec8*)((SB) + koff[ks]);                                          \
        kf[1][ks] = *(const vec8*)((SB) + koff[ks] + 4096);                                   \
      }                                                                                       \
      _Pragma("unroll") for (int sub = 0; sub < 2; ++sub)                                     \
        _Pragma("unroll") for (int e = 0; e < 16; ++e) SNEXT[sub][e] = LOG2Q ? -m2 : 0.f;     \
    }                                                                                         \
    float ra = 0.f, rb = 0.f;                                                                 \
    _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                                        \
      if (WITH_QK) {                                                                          \
        SNEXT[0] = Elem<T>::mma32(kf[0][ks], qf[ks], SNEXT[0]);                               \
        SNEXT[1] = Elem<T>::mma32(kf[1][ks], qf[ks], SNEXT[1]);                               \
      }                                                                                       \
      _Pragma("unroll") for (int e = 4 * ks; e < 4 * ks + 4; ++e) {                           \
        float pa = __builtin_amdgcn_exp2f(SCUR[0][e]);                                        \
        float pb = __builtin_amdgcn_exp2f(SCUR[1][e]);                                        \
        SCUR[0][e] = pa; SCUR[1][e] = pb;                                                     \
        ra += pa; rb += pb;                                                                   \
      }                                                                                       \
      __builtin_amdgcn_sched_barrier(0);   /* keep 2 MFMAs + 8 exponentials per slice */      \
    }                                                                                         \
    ra += rb;                                                                                 \
    l += ra + __shfl_xor(ra, 32, 64);                                                         \
  }

  auto pv = [&](const char* sb, f32x16 (&s)[2]) {
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        vec8 pf;
#pragma unroll
        for (int j = 0; j < 8; ++j) pf[j] = from_float<T>(s[sub][8 * s2 + j]);
#pragma unroll
        for (int db = 0; db < 2; ++db) {
          const char* vp = sb + voff[db] + (sub * 32 + 16 * s2) * 128;
          i16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)(vp));
          i16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)(vp + 8 * 128));
          i16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
          o[db] = Elem<T>::mma32(__builtin_bit_cast(vec8, both), pf, o[db]);
        }
      }
  };

  auto rebase = [&](f32x16 (&s)[2], float mt, bool first) {
    if (first || __any(mt > 0.f)) {
      const float delta = first ? mt : fmaxf(mt, 0.f);
      const float alpha = first ? 1.f : __builtin_amdgcn_exp2f(-delta);   // l, O still zero: no 0 * inf
      m2 += delta;
      l *= alpha;
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[db][e] *= alpha;
#pragma unroll
      for (int sub = 0; sub < 2; ++sub)
#pragma unroll
        for (int e = 0; e < 16; ++e) s[sub][e] -= delta;
    }
  };

  // ring prologue: up to three tiles in flight; tile 0 visible
  stage(0, 0);
  if (nkt > 1) stage(1, 1);
  if (nkt > 2) stage(2, 2);
  if (nkt > 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if (nkt > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  f32x16 sA[2], sB[2];
  {  // S(0)
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
      for (int e = 0; e < 16; ++e) sA[sub][e] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) sA[sub] = Elem<T>::mma32(*(const vec8*)(smem + koff[ks] + sub * 4096), qf[ks], sA[sub]);
    }
  }
  float mt = finish_scores(sA, 0);

// one key tile that has a successor: SCUR holds S'(t); S'(t+1) is computed into SNEXT while SCUR is exponentiated
#define PIPE_STEP(i, SCUR, SNEXT)                                                  \
  {                                                                                \
    const int t = kt + (i);                                                        \
    if (t + 1 >= nkt) break;                                                       \
    /* make tile t+1 visible, free the stage of tile t-1 */                        \
    if (t + 2 < nkt) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");              \
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                          \
    __builtin_amdgcn_s_barrier();                                                  \
    if (t + 3 < nkt) stage(((i) + 3) & 3, t + 3);                                  \
    rebase(SCUR, mt, t == 0);                                                      \
    QK_EXP(smem + (((i) + 1) & 3) * 16384, SNEXT, SCUR, true)                      \
    pv(smem + ((i) & 3) * 16384, SCUR);                                            \
    mt = finish_scores(SNEXT, t + 1);                                              \
  }
  for (int kt = 0; kt < nkt; kt += 4) {
    PIPE_STEP(0, sA, sB)
    PIPE_STEP(1, sB, sA)
    PIPE_STEP(2, sA, sB)
    PIPE_STEP(3, sB, sA)
  }
  {  // last tile (no successor): its scores sit in sA for an even tile index, sB for an odd one
    const int t = nkt - 1;
    const char* sb = smem + (t & 3) * 16384;
    if (t & 1) {
      rebase(sB, mt, t == 0);
      QK_EXP(smem, sA, sB, false)
      pv(sb, sB);
    } else {
      rebase(sA, mt, t == 0);
      QK_EXP(smem, sB, sA, false)
      pv(sb, sA);
    }
  }
#undef PIPE_STEP
#undef QK_EXP

  if (qi < L) {
    const float inv = 1.0f / l;
    T* dst = ctx + ((long)b * L + qi) * D + head * 64;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int gi = 0; gi < 4; ++gi) {
        vec4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = from_float<T>(o[db][4 * gi + j] * inv);
        *(vec4*)(dst + db * 32 + 8 * gi + 4 * h) = v;
      }
  }
}

#endif  // AACLIP_MEASURE

// ------------------------------------------------------------------ fp32 path
__global__ __launch_bounds__(256) void attn32_kernel(const float* __restrict__ qkv, float* __restrict__ ctx, int L,
                                                     int H, int causal) {
  __shared__ __attribute__((aligned(16))) float Ks[32 * 64];
  __shared__ __attribute__((aligned(16))) float Vs[32 * 64];
  const int tid = threadIdx.x;
  const int head = blockIdx.y, b = blockIdx.z;
  const int D = H * 64;
  const long ld = 3L * D;
  const float* base = qkv + (long)b * L * ld + head * 64;
  const int qi = blockIdx.x * 256 + tid;
  const int qrow = qi < L ? qi : L - 1;
  float q[64], o[64];
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    f32x4 v = *(const f32x4*)(base + (long)qrow * ld + 4 * c);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      q[4 * c + e] = v[e];
      o[4 * c + e] = 0.f;
    }
  }
  float m = -INFINITY, l = 0.f;
  int last_q = blockIdx.x * 256 + 255;
  if (last_q > L - 1) last_q = L - 1;
  const int nkt = causal ? (last_q / 32 + 1) : ((L + 31) / 32);
  for (int kt = 0; kt < nkt; ++kt) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      int idx = tid + 256 * j;
      int key = idx >> 4, c4 = (idx & 15) * 4;
      int kg = kt * 32 + key;
      kg = kg < L ? kg : L - 1;
      *(f32x4*)(Ks + key * 64 + c4) = *(const f32x4*)(base + (long)kg * ld + D + c4);
      *(f32x4*)(Vs + key * 64 + c4) = *(const f32x4*)(base + (long)kg * ld + 2 * D + c4);
    }
    __syncthreads();
    float s[32];
    float mt = -INFINITY;
#pragma unroll
    for (int key = 0; key < 32; ++key) {
      float a = 0.f;
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        f32x4 kv = *(const f32x4*)(Ks + key * 64 + 4 * c);
        a = fmaf(q[4 * c + 0], kv[0], a);
        a = fmaf(q[4 * c + 1], kv[1], a);
        a = fmaf(q[4 * c + 2], kv[2], a);
        a = fmaf(q[4 * c + 3], kv[3], a);
      }
      int kg = kt * 32 + key;
      bool dead = (kg >= L) || (causal && kg > qi);
      a = dead ? -INFINITY : a;
      s[key] = a;
      mt = fmaxf(mt, a);
    }
    const float mn = fmaxf(m, mt);
    const float alpha = expf(m - mn);
    float rs = 0.f;
#pragma unroll
    for (int key = 0; key < 32; ++key) {
      s[key] = expf(s[key] - mn);
      rs += s[key];
    }
    l = l * alpha + rs;
    m = mn;
#pragma unroll
    for (int d = 0; d < 64; ++d) o[d] *= alpha;
#pragma unroll
    for (int key = 0; key < 32; ++key) {
      const float pk = s[key];
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        f32x4 vv = *(const f32x4*)(Vs + key * 64 + 4 * c);
        o[4 * c + 0] = fmaf(pk, vv[0], o[4 * c + 0]);
        o[4 * c + 1] = fmaf(pk, vv[1], o[4 * c + 1]);
        o[4 * c + 2] = fmaf(pk, vv[2], o[4 * c + 2]);
        o[4 * c + 3] = fmaf(pk, vv[3], o[4 * c + 3]);
      }
    }
  }
  if (qi < L) {
    const float inv = 1.0f / l;
    float* dst = ctx + ((long)b * L + qi) * D + head * 64;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = o[4 * c + e] * inv;
      *(f32x4*)(dst + 4 * c) = v;
    }
  }
}

// fp32 attention on the fp32 MATRIX pipe: v_mfma_f32_32x32x2_f32 is an exact fmaf chain (guide section 3) at the
// fp32 vector rate, so the parity path loses nothing by using it, and unlike the VALU kernel above it runs near that
// rate: the VALU kernel was 48 % of an exact-fp32 step (432 of 900 ms at B = 64; 27 TFLOP/s).
// One wave = 32 queries, a workgroup = 4 waves = 128 queries; 32-key tiles, double-buffered in LDS.
//   S^T[key, q] = K . Q^T : 32 MFMAs (K = 2 each); A operand = K[key = lane%32][d = 2i + lane/32] from a tile stored
//                 de-interleaved by parity ([parity][key][36]: the pad makes the 8 ds_read_b128 per lane conflict-free),
//                 B operand = Q (registers, x log2(e))
//   online softmax on the accumulator: the query is on the lane, its 32 keys are split over the two lane halves
//   O^T[d, q] += V^T . P^T : 2 x 16 MFMAs; MFMA i takes P element i of each lane as its B operand as it stands (the key
//                 of element i is (i&3) + 8(i>>2) + 4(lane/32): a sum over keys does not care about their order), and
//                 the A operand reads that key's row of V.
__global__ __launch_bounds__(256) void attn32m_kernel(const float* __restrict__ qkv, float* __restrict__ ctx, int L, int H,
                                                      int causal) {
  __shared__ __attribute__((aligned(16))) float Ks[2][2][32][36];
  __shared__ __attribute__((aligned(16))) float Vs[2][32][64];
  constexpr float LOG2E = 1.4426950408889634f;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int head = blockIdx.y, b = blockIdx.z;
  const int D = H * 64;
  const long ld = 3L * D;
  const float* base = qkv + (long)b * L * ld + head * 64;
  const int q0 = blockIdx.x * 128 + wave * 32;
  const int qi = q0 + r;
  const int qrow = qi < L ? qi : L - 1;
  float qreg[32];
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    const f32x4 v = *(const f32x4*)(base + (long)qrow * ld + 4 * c);   // d = 4c .. 4c+3; this lane keeps parity h
    qreg[2 * c] = (h ? v[1] : v[0]) * LOG2E;
    qreg[2 * c + 1] = (h ? v[3] : v[2]) * LOG2E;
  }
  f32x16 o[2];
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int e = 0; e < 16; ++e) o[db][e] = 0.f;
  float m = -1e30f, l = 0.f;   // finite start: a fully masked tile must not produce inf - inf
  int last_q = blockIdx.x * 128 + 127;
  if (last_q > L - 1) last_q = L - 1;
  const int nkt = causal ? (last_q / 32 + 1) : ((L + 31) / 32);

  f32x4 kreg[2], vreg[2];
  auto load_tile = [&](int kt) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int idx = tid + 256 * j;
      const int key = idx >> 4, c4 = (idx & 15) * 4;
      int kg = kt * 32 + key;
      kg = kg < L ? kg : L - 1;
      kreg[j] = *(const f32x4*)(base + (long)kg * ld + D + c4);
      vreg[j] = *(const f32x4*)(base + (long)kg * ld + 2 * D + c4);
    }
  };
  auto store_tile = [&](int st) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int idx = tid + 256 * j;
      const int key = idx >> 4, c4 = (idx & 15) * 4;
      *(f32x2*)&Ks[st][0][key][c4 >> 1] = (f32x2){kreg[j][0], kreg[j][2]};
      *(f32x2*)&Ks[st][1][key][c4 >> 1] = (f32x2){kreg[j][1], kreg[j][3]};
      *(f32x4*)&Vs[st][key][c4] = vreg[j];
    }
  };
  auto xswap = [](float& a, float& b2) { asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b2)); };

  load_tile(0);
  store_tile(0);
  __syncthreads();
  for (int kt = 0; kt < nkt; ++kt) {
    const int st = kt & 1;
    if (kt + 1 < nkt) load_tile(kt + 1);
    if (q0 < L) {
      f32x16 s;
#pragma unroll
      for (int e = 0; e < 16; ++e) s[e] = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const f32x4 kf = *(const f32x4*)&Ks[st][h][r][4 * j];
#pragma unroll
        for (int e = 0; e < 4; ++e) s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[e], qreg[4 * j + e], s, 0, 0, 0);
      }
      const int k0 = kt * 32;
      if ((k0 + 32 > L) || (causal && (k0 + 31 > q0))) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int key = k0 + (e & 3) + 8 * (e >> 2) + 4 * h;
          const bool dead = (key >= L) || (causal && key > qi);
          s[e] = dead ? -INFINITY : s[e];
        }
      }
      float mt = s[0];
#pragma unroll
      for (int e = 1; e < 16; ++e) mt = fmaxf(mt, s[e]);
      {
        float a = mt, c = mt;
        xswap(a, c);
        mt = fmaxf(a, c);
      }
      const float mn = fmaxf(m, mt);
      const float alpha = __builtin_amdgcn_exp2f(m - mn);
      float rs = 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        s[e] = __builtin_amdgcn_exp2f(s[e] - mn);
        rs += s[e];
      }
      {
        float a = rs, c = rs;
        xswap(a, c);
        rs = a + c;
      }
      l = l * alpha + rs;
      m = mn;
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[db][e] *= alpha;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int key = (i & 3) + 8 * (i >> 2) + 4 * h;
        o[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(Vs[st][key][r], s[i], o[0], 0, 0, 0);
        o[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(Vs[st][key][32 + r], s[i], o[1], 0, 0, 0);
      }
    }
    if (kt + 1 < nkt) store_tile(st ^ 1);   // the other stage: everyone left it at the barrier below, one tile ago
    __syncthreads();
  }
  if (qi < L) {
    const float inv = 1.0f / l;
    float* dst = ctx + ((long)b * L + qi) * D + head * 64;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = o[db][4 * g + j] * inv;
        *(f32x4*)(dst + db * 32 + 8 * g + 4 * h) = v;
      }
  }
}

#ifdef AACLIP_MEASURE
void read_attn_passes(unsigned long long* out4, int reset) {
  (void)hipMemcpyFromSymbol(out4, HIP_SYMBOL(g_attn_passes), 4 * sizeof(unsigned long long));
  if (reset) {
    unsigned long long z[4] = {0, 0, 0, 0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_attn_passes), z, sizeof(z));
  }
}
void read_attn_stamps(unsigned long long* out9, int reset) {
  (void)hipMemcpyFromSymbol(out9, HIP_SYMBOL(g_attn_stamp), 13 * sizeof(unsigned long long));
  if (reset) {
    unsigned long long z[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_attn_stamp), z, sizeof(z));
  }
}
#endif
#ifndef ATTN_PD
#define ATTN_PD 2   // prefetch distance of the long-sequence kernel's K/V ring, in tiles (1, 2, 3 measured alike;
                    // beyond ~8 DMA instructions in flight per wave their ISSUE starts to block: tools/ldsdma_probe.hip)
#endif
static int g_attn_variant = 0;  // 1 = always the 2-stage 128-row kernel (A/B measurements)
bool set_attn_variant(int v) {
#ifdef AACLIP_MEASURE
  const bool ok = (v >= 0 && v <= 3) || v == 6;   // 6 = unit-pipelined experiment (attn16u_kernel)   // 2 = software-pipelined 128-query kernel, 3 = long-sequence kernel with 8-wave workgroups
#else
  const bool ok = v == 0 || v == 1;
#endif
  if (ok) g_attn_variant = v;
  return ok;
}

bool attention_qk8_applicable(int L, int causal) { return L >= 512 && !causal; }
void launch_attention(int dtype, const void* qkv, void* ctx, int B, int L, int H, int causal, int log2q,
                      hipStream_t s, bool hi8, bool qk8) {
  if (dtype == AACLIP_F16X2) {   // split fp16 rows in, split fp16 rows out
    const int nqt = (L + 127) / 128;
    const long totl = (long)nqt * H * B;
    const int per_xcd = (int)((totl + 7) / 8), tot = (int)totl;
    dim3 g((unsigned)(per_xcd * 8));
#define ATTN_LAUNCH_S(LQ, VLO) hipLaunchKernelGGL((attn16s_kernel<LQ, VLO>), g, dim3(256), 0, s, (const f16*)qkv, (f16*)ctx, L, H, causal, nqt, tot, per_xcd, hi8)
    if (qk8) {        // block path, long rows: [q k v hi | q8 | k8] records, correction products on the e4m3 MFMA
      hipLaunchKernelGGL((attn16s_kernel<true, false, true>), g, dim3(256), 0, s, (const f16*)qkv, (f16*)ctx, L, H, causal, nqt, tot, per_xcd, hi8);
    } else if (L >= 512) {   // long rows: v's lo half is not read (see attn16s_kernel)
      if (log2q) ATTN_LAUNCH_S(true, false); else ATTN_LAUNCH_S(false, false);
    } else {
      if (log2q) ATTN_LAUNCH_S(true, true); else ATTN_LAUNCH_S(false, true);
    }
#undef ATTN_LAUNCH_S
  } else if (dtype == AACLIP_F32 && L >= 64 && g_attn_variant != 1) {   // fp32 MFMA kernel (32 queries per wave)
    dim3 g((L + 127) / 128, H, B);
    hipLaunchKernelGGL(attn32m_kernel, g, dim3(256), 0, s, (const float*)qkv, (float*)ctx, L, H, causal);
  } else if (dtype == AACLIP_F32) {                               // short sequences: one query per lane on the VALU
    dim3 g((L + 255) / 256, H, B);
    hipLaunchKernelGGL(attn32_kernel, g, dim3(256), 0, s, (const float*)qkv, (float*)ctx, L, H, causal);
#ifdef AACLIP_MEASURE
  } else if (L >= 512 && g_attn_variant == 2) {   // software-pipelined kernel
    dim3 g((L + 127) / 128, H, B);
    if (dtype == AACLIP_F16) {
      if (log2q) hipLaunchKernelGGL((attn16p_kernel<f16, true>), g, dim3(256), 0, s, (const f16*)qkv, (f16*)ctx, L, H, causal);
      else hipLaunchKernelGGL((attn16p_kernel<f16, false>), g, dim3(256), 0, s, (const f16*)qkv, (f16*)ctx, L, H, causal);
    } else {
      if (log2q) hipLaunchKernelGGL((attn16p_kernel<bf16, true>), g, dim3(256), 0, s, (const bf16*)qkv, (bf16*)ctx, L, H, causal);
      else hipLaunchKernelGGL((attn16p_kernel<bf16, false>), g, dim3(256), 0, s, (const bf16*)qkv, (bf16*)ctx, L, H, causal);
    }
#endif
#ifdef AACLIP_MEASURE
  } else if (L >= 512 && g_attn_variant == 6) {   // unit-pipelined experiment
    const int nqt = (L + 255) / 256;
    const long total = (long)nqt * H * B;
    const int per_xcd = (int)((total + 7) / 8);
    dim3 g((unsigned)(per_xcd * 8)), blk(256);
    const int tot = (int)total;
#define ATTN_LAUNCH_U(TT, LQ) hipLaunchKernelGGL((attn16u_kernel<TT, LQ, 4>), g, blk, 0, s, (const TT*)qkv, (TT*)ctx, L, H, causal, nqt, tot, per_xcd)
    if (dtype == AACLIP_F16) { if (log2q) ATTN_LAUNCH_U(f16, true); else ATTN_LAUNCH_U(f16, false); }
    else { if (log2q) ATTN_LAUNCH_U(bf16, true); else ATTN_LAUNCH_U(bf16, false); }
#undef ATTN_LAUNCH_U
#endif
  } else if (L >= 512 && g_attn_variant != 1) {
    // 256 queries (4 waves) per workgroup, two workgroups per CU.  512 queries in one 8-wave workgroup halve the K/V
    // bytes a CU streams, and measured the same (0.699 vs 0.703 ms): kept as variant 3 of the measurement library.
#ifdef AACLIP_MEASURE
    const int nw = g_attn_variant == 3 ? 8 : 4;
#else
    const int nw = 4;
#endif
    const int nqt = (L + nw * 64 - 1) / (nw * 64);
    const long total = (long)nqt * H * B;
    const int per_xcd = (int)((total + 7) / 8);
    dim3 g((unsigned)(per_xcd * 8)), blk(nw * 64);
    const int tot = (int)total;
#define ATTN_LAUNCH(TT, LQ, NWV) hipLaunchKernelGGL((attn16x2_kernel<TT, LQ, NWV, ATTN_PD>), g, blk, 0, s, (const TT*)qkv, (TT*)ctx, L, H, causal, nqt, tot, per_xcd)
#ifdef AACLIP_MEASURE
    if (nw == 8) {
      if (dtype == AACLIP_F16) { if (log2q) ATTN_LAUNCH(f16, true, 8); else ATTN_LAUNCH(f16, false, 8); }
      else { if (log2q) ATTN_LAUNCH(bf16, true, 8); else ATTN_LAUNCH(bf16, false, 8); }
    } else
#endif
    {
      if (dtype == AACLIP_F16) { if (log2q) ATTN_LAUNCH(f16, true, 4); else ATTN_LAUNCH(f16, false, 4); }
      else { if (log2q) ATTN_LAUNCH(bf16, true, 4); else ATTN_LAUNCH(bf16, false, 4); }
    }
#undef ATTN_LAUNCH
  } else {
    dim3 g((L + 127) / 128, H, B);
    if (dtype == AACLIP_F16) {
      if (log2q) hipLaunchKernelGGL((attn16_kernel<f16, true>), g, dim3(256), 0, s, (const f16*)qkv, (f16*)ctx, L, H, causal);
      else hipLaunchKernelGGL((attn16_kernel<f16, false>), g, dim3(256), 0, s, (const f16*)qkv, (f16*)ctx, L, H, causal);
    } else {
      if (log2q) hipLaunchKernelGGL((attn16_kernel<bf16, true>), g, dim3(256), 0, s, (const bf16*)qkv, (bf16*)ctx, L, H, causal);
      else hipLaunchKernelGGL((attn16_kernel<bf16, false>), g, dim3(256), 0, s, (const bf16*)qkv, (bf16*)ctx, L, H, causal);
    }
  }
}

}  // namespace aaclip
