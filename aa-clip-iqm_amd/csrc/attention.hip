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
      const float alpha = __builtin_amdgcn_exp2f(-delta);
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
// Long-sequence variant (the visual tower, L = 1370): one wave owns TWO 32-row
// query blocks (64 rows), a workgroup 256 rows, so every K/V fragment read from
// LDS and every DMA'd byte feeds twice the MFMA work, and the two blocks give
// the scheduler two independent softmax chains.  K/V tiles go through a 4-stage
// ring (three tiles = 48 KiB in flight per workgroup, two workgroups per CU)
// with counted vmcnt waits: the 2-stage kernel above was bound by DMA latency x
// bytes in flight, not by MFMA or VALU.
template <typename T, bool LOG2Q>
__global__ __launch_bounds__(256, 2) void attn16x2_kernel(const T* __restrict__ qkv, T* __restrict__ ctx, int L, int H,
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
  const int q0 = qt * 256 + wave * 64;
  const bool active = q0 < L;   // wave-uniform: idle waves only feed the ring and the barriers

  vec8 qf[2][4];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    int qrow = q0 + qb * 32 + r;
    qrow = qrow < L ? qrow : L - 1;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[qb][ks] = *(const vec8*)(base + (long)qrow * ld + 16 * ks + 8 * h);
  }

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

  int last_q = qt * 256 + 255;
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

  f32x16 o[2][2];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb)
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int e = 0; e < 16; ++e) o[qb][db][e] = 0.f;
  float m2[2] = {0.f, 0.f}, l[2] = {0.f, 0.f};

  auto tile = [&](const char* sb, int kt) {
    f32x16 s[2][2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
      for (int sub = 0; sub < 2; ++sub)
#pragma unroll
        for (int e = 0; e < 16; ++e) s[qb][sub][e] = LOG2Q ? -m2[qb] : 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      vec8 k0f = *(const vec8*)(sb + koff[ks]);
      vec8 k1f = *(const vec8*)(sb + koff[ks] + 4096);
      s[0][0] = Elem<T>::mma32(k0f, qf[0][ks], s[0][0]);
      s[1][0] = Elem<T>::mma32(k0f, qf[1][ks], s[1][0]);
      s[0][1] = Elem<T>::mma32(k1f, qf[0][ks], s[0][1]);
      s[1][1] = Elem<T>::mma32(k1f, qf[1][ks], s[1][1]);
    }
    if (!LOG2Q) {
#pragma unroll
      for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
          for (int e = 0; e < 16; ++e) s[qb][sub][e] = fmaf(s[qb][sub][e], LOG2E, -m2[qb]);
    }
    const int k0 = kt * 64;
    const bool need_mask = (k0 + 64 > L) || (causal && (k0 + 63 > q0));
    if (need_mask) {
#pragma unroll
      for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            int key = k0 + sub * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
            bool dead = (key >= L) || (causal && key > q0 + qb * 32 + r);
            s[qb][sub][e] = dead ? -INFINITY : s[qb][sub][e];
          }
    }
    float mt[2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      float a = s[qb][0][0], c = s[qb][1][0];
#pragma unroll
      for (int e = 1; e < 16; ++e) {
        a = fmaxf(a, s[qb][0][e]);
        c = fmaxf(c, s[qb][1][e]);
      }
      a = fmaxf(a, c);
      mt[qb] = fmaxf(a, __shfl_xor(a, 32, 64));
    }
    const bool first = kt == 0;
    if (first || __any(fmaxf(mt[0], mt[1]) > 0.f)) {
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) {
        const float delta = first ? mt[qb] : fmaxf(mt[qb], 0.f);
        const float alpha = __builtin_amdgcn_exp2f(-delta);
        m2[qb] += delta;
        l[qb] *= alpha;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
          for (int e = 0; e < 16; ++e) o[qb][db][e] *= alpha;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
          for (int e = 0; e < 16; ++e) s[qb][sub][e] -= delta;
      }
    }
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      float ra = 0.f, rb = 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        float pa = __builtin_amdgcn_exp2f(s[qb][0][e]);
        float pb = __builtin_amdgcn_exp2f(s[qb][1][e]);
        s[qb][0][e] = pa;
        s[qb][1][e] = pb;
        ra += pa;
        rb += pb;
      }
      ra += rb;
      l[qb] += ra + __shfl_xor(ra, 32, 64);
    }
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        vec8 vf[2];
#pragma unroll
        for (int db = 0; db < 2; ++db) {
          const char* vp = sb + voff[db] + (sub * 32 + 16 * s2) * 128;
          i16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)(vp));
          i16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)(vp + 8 * 128));
          i16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
          vf[db] = __builtin_bit_cast(vec8, both);
        }
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
          vec8 pf;
#pragma unroll
          for (int j = 0; j < 8; ++j) pf[j] = from_float<T>(s[qb][sub][8 * s2 + j]);
          o[qb][0] = Elem<T>::mma32(vf[0], pf, o[qb][0]);
          o[qb][1] = Elem<T>::mma32(vf[1], pf, o[qb][1]);
        }
      }
  };

  // ring prologue: up to three tiles in flight
  stage(0, 0);
  if (nkt > 1) stage(1, 1);
  if (nkt > 2) stage(2, 2);
#define RING_STEP(i)                                                      \
  {                                                                       \
    const int t = kt + (i);                                               \
    if (t >= nkt) break;                                                  \
    const int younger = nkt - 1 - t;                                      \
    if (younger >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");    \
    else if (younger == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); \
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 \
    __builtin_amdgcn_s_barrier();                                         \
    if (t + 3 < nkt) stage(((i) + 3) & 3, t + 3);                         \
    if (active) tile(smem + (i) * 16384, t);                              \
  }
  for (int kt = 0; kt < nkt; kt += 4) {
    RING_STEP(0)
    RING_STEP(1)
    RING_STEP(2)
    RING_STEP(3)
  }
#undef RING_STEP

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
      const float alpha = __builtin_amdgcn_exp2f(-delta);
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

static int g_attn_variant = 0;  // 1 = always the 2-stage 128-row kernel (A/B measurements)
bool set_attn_variant(int v) {
#ifdef AACLIP_MEASURE
  const bool ok = v >= 0 && v <= 2;
#else
  const bool ok = v == 0 || v == 1;
#endif
  if (ok) g_attn_variant = v;
  return ok;
}

void launch_attention(int dtype, const void* qkv, void* ctx, int B, int L, int H, int causal, int log2q,
                      hipStream_t s) {
  if (dtype == AACLIP_F32) {
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
  } else if (L >= 512 && g_attn_variant != 1) {
    dim3 g((L + 255) / 256, H, B);
    if (dtype == AACLIP_F16) {
      if (log2q) hipLaunchKernelGGL((attn16x2_kernel<f16, true>), g, dim3(256), 0, s, (const f16*)qkv, (f16*)ctx, L, H, causal);
      else hipLaunchKernelGGL((attn16x2_kernel<f16, false>), g, dim3(256), 0, s, (const f16*)qkv, (f16*)ctx, L, H, causal);
    } else {
      if (log2q) hipLaunchKernelGGL((attn16x2_kernel<bf16, true>), g, dim3(256), 0, s, (const bf16*)qkv, (bf16*)ctx, L, H, causal);
      else hipLaunchKernelGGL((attn16x2_kernel<bf16, false>), g, dim3(256), 0, s, (const bf16*)qkv, (bf16*)ctx, L, H, causal);
    }
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
