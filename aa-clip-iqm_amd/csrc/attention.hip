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

template <typename T>
__global__ __launch_bounds__(256) void attn16_kernel(const T* __restrict__ qkv, T* __restrict__ ctx, int L, int H,
                                                     int causal) {
  typedef typename Elem<T>::vec8 vec8;
  typedef typename Elem<T>::vec4 vec4;
  __shared__ __attribute__((aligned(16))) char smem[32768];  // 2 stages x (K 8K + V 8K)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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

  // DMA source (row, chunk) of the two wave-instructions this wave issues per tile
  int srow[2], scol[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    int row, chunk;
    tile_src((wave * 2 + j) * 64 + lane, row, chunk);
    srow[j] = row;
    scol[j] = chunk * 8;
  }
  int koff[2][4];
#pragma unroll
  for (int sub = 0; sub < 2; ++sub)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) koff[sub][ks] = tile_off(sub * 32 + r, 2 * ks + h);
  // transposed-read addresses: lane group g = lane>>4, i = lane&15 = 4*qq + pp
  int voff[2][2][2][2];
  {
    const int g = lane >> 4, i = lane & 15, qq = i >> 2, pp = i & 3;
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int db = 0; db < 2; ++db) {
            int key = sub * 32 + 16 * s2 + 8 * t + 4 * (g >> 1) + qq;
            int col = db * 32 + (g & 1) * 16 + 4 * pp;
            voff[sub][s2][t][db] = 8192 + tile_off(key, col >> 3) + (col & 7) * 2;
          }
  }

  int last_q = qt * 128 + 127;
  if (last_q > L - 1) last_q = L - 1;
  const int nkt = causal ? (last_q / 64 + 1) : ((L + 63) / 64);

  auto stage = [&](int s, int kt) {
    char* dst = smem + s * 16384 + wave * 2048;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      int key = kt * 64 + srow[j];
      key = key < L ? key : L - 1;
      const T* src = base + (long)key * ld + scol[j];
      glds16(src + D, dst + j * 1024);
      glds16(src + 2 * D, dst + 8192 + j * 1024);
    }
  };

  f32x16 o[2];
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int e = 0; e < 16; ++e) o[db][e] = 0.f;
  float m = -INFINITY, l = 0.f;

  stage(0, 0);
  wait_vm0();
  __syncthreads();
  for (int kt = 0; kt < nkt; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nkt) stage(cur ^ 1, kt + 1);
    const char* sb = smem + cur * 16384;

    f32x16 s[2];
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
      for (int e = 0; e < 16; ++e) s[sub][e] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        vec8 a = *(const vec8*)(sb + koff[sub][ks]);
        s[sub] = Elem<T>::mma32(a, qf[ks], s[sub]);
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
    const float mn = fmaxf(m, mt);
    const float alpha = __expf(m - mn);
    float rs = 0.f;
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        float pv = __expf(s[sub][e] - mn);
        s[sub][e] = pv;
        rs += pv;
      }
    rs += __shfl_xor(rs, 32, 64);
    l = l * alpha + rs;
    m = mn;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int e = 0; e < 16; ++e) o[db][e] *= alpha;

#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        vec8 pf;
#pragma unroll
        for (int j = 0; j < 8; ++j) pf[j] = from_float<T>(s[sub][8 * s2 + j]);
#pragma unroll
        for (int db = 0; db < 2; ++db) {
          i16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) i16x4*)(sb + voff[sub][s2][0][db]));
          i16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) i16x4*)(sb + voff[sub][s2][1][db]));
          typedef short i16x8 __attribute__((ext_vector_type(8)));
          i16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
          vec8 vf = __builtin_bit_cast(vec8, both);
          o[db] = Elem<T>::mma32(vf, pf, o[db]);
        }
      }
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

void launch_attention(int dtype, const void* qkv, void* ctx, int B, int L, int H, int causal, hipStream_t s) {
  if (dtype == AACLIP_F32) {
    dim3 g((L + 255) / 256, H, B);
    hipLaunchKernelGGL(attn32_kernel, g, dim3(256), 0, s, (const float*)qkv, (float*)ctx, L, H, causal);
  } else {
    dim3 g((L + 127) / 128, H, B);
    if (dtype == AACLIP_F16)
      hipLaunchKernelGGL(attn16_kernel<f16>, g, dim3(256), 0, s, (const f16*)qkv, (f16*)ctx, L, H, causal);
    else
      hipLaunchKernelGGL(attn16_kernel<bf16>, g, dim3(256), 0, s, (const bf16*)qkv, (bf16*)ctx, L, H, causal);
  }
}

}  // namespace aaclip
