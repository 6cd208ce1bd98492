// LDS-DMA issue/landing probe (gfx950): what does a wave pay to ISSUE buffer_load ... lds, and does the DMA touch lgkmcnt?
//   hipcc --offload-arch=gfx950 -O3 -o tools/ldsdma_probe tools/ldsdma_probe.hip && tools/ldsdma_probe
// Per workgroup of `nw` waves (one or two workgroups per CU), every wave issues `n` DMA instructions of 1 KiB each
// (8 rows x 128 B, rows `stride` bytes apart, like a K/V tile of the attention kernel), then
//   t_issue  = s_memtime around the n issues (no waits inside)
//   t_lgkm   = a second stamp right after (its own cost; the first already waited lgkmcnt(0): if LDS-DMA held lgkmcnt,
//              t_issue would contain the landing time and t_land would be ~0)
//   t_land   = extra cycles until s_waitcnt vmcnt(0)
// repeated `iters` times over a buffer larger than the L2 (HBM) or small (L2 hits).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned long long clk() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
template <int N, int WORK>
__global__ __launch_bounds__(512) void probe(const char* src, size_t bytes, int stride, int iters, unsigned long long* out) {
  __shared__ __attribute__((aligned(16))) char smem[65536];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned long long ub = (unsigned long long)src;
  u32x4 rs;
  rs[0] = __builtin_amdgcn_readfirstlane((unsigned)ub);
  rs[1] = __builtin_amdgcn_readfirstlane((unsigned)(ub >> 32)) & 0xFFFFu;
  rs[2] = 0x7FFFFFF0u;
  rs[3] = 0x00020000u;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)smem + wave * (N * 1024 > 8192 ? 8192 : N * 1024);
  const int voff = (lane >> 3) * stride + (lane & 7) * 16;
  unsigned long long ti = 0, tl = 0, tv = 0;
  size_t pos = ((size_t)blockIdx.x * (blockDim.x >> 6) + wave) * (size_t)N * 8 * stride;
  for (int it = 0; it < iters; ++it) {
    pos %= (bytes - (size_t)N * 8 * stride - 4096);
    const int so = __builtin_amdgcn_readfirstlane((int)(pos & 0x7FFFFFF0));
    const unsigned long long t0 = clk();
#pragma unroll
    for (int i = 0; i < N; ++i) {
      unsigned keep;
      asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "s"(lds0 + (i & 7) * 1024), "v"(voff + i * 8 * stride), "s"(rs), "s"(so) : "memory");
    }
    const unsigned long long t1 = clk();   // s_memtime + lgkmcnt(0): if LDS-DMA held lgkmcnt this would include the landing
    const unsigned long long t2 = clk();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t3 = clk();
    ti += t1 - t0; tl += t2 - t1; tv += t3 - t2;
    if (WORK) {   // what the attention kernel does between two issues: LDS fragment reads + MFMAs (+ VALU), ~2-3k cycles
      typedef _Float16 h8 __attribute__((ext_vector_type(8)));
      typedef float f16v __attribute__((ext_vector_type(16)));
      f16v acc = {0};
      for (int k = 0; k < WORK; ++k) {
        const h8 a = *(const h8*)(smem + ((lane * 16 + k * 1024) & 0xFFF0));
        const h8 b = *(const h8*)(smem + ((lane * 16 + k * 1024 + 8192) & 0xFFF0));
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
        acc[0] = __builtin_amdgcn_exp2f(acc[1]) + acc[2];
      }
      if (acc[3] == 12345.f) out[5] = 1;
    }
    pos += (size_t)gridDim.x * (blockDim.x >> 6) * N * 8 * stride;
    __syncthreads();
  }
  if (lane == 0) {
    atomicAdd(&out[0], ti); atomicAdd(&out[1], tl); atomicAdd(&out[2], tv); atomicAdd(&out[3], (unsigned long long)iters);
  }
  if (smem[threadIdx.x] == 77 && out[3] == 1) out[4] = 1;   // keep the LDS alive
}
template <int N, int WORK = 0> void run(const char* buf, size_t bytes, int stride, int nw, int wgs, const char* what, unsigned long long* dout) {
  hipMemset(dout, 0, 64);
  hipLaunchKernelGGL((probe<N, WORK>), dim3(wgs), dim3(nw * 64), 0, 0, buf, bytes, stride, 200, dout);
  hipDeviceSynchronize();
  unsigned long long h[4];
  hipMemcpy(h, dout, 32, hipMemcpyDeviceToHost);
  printf("%-34s work %2d N=%2d DMA/wave, %d waves/WG, %4d WGs: issue %6.0f cyc (%4.0f each), +lgkmcnt(0) %5.0f, +vmcnt(0) %6.0f  per round per wave\n", what, N, nw, wgs,
         (double)h[0] / h[3], (double)h[0] / h[3] / N, (double)h[1] / h[3], (double)h[2] / h[3]);
}
int main() {
  const size_t big = 2048ull << 20, small = 16ull << 20;
  char* buf; unsigned long long* dout;
  hipMalloc(&buf, big); hipMalloc(&dout, 64);
  hipMemset(buf, 1, big);
  for (int pass = 0; pass < 2; ++pass) {
    const size_t bytes = pass ? small : big;
    const char* what = pass ? "16 MiB buffer (L2 / MALL hits)" : "2 GiB buffer (HBM)";
    run<1>(buf, bytes, 6144, 4, 512, what, dout);
    run<4>(buf, bytes, 6144, 4, 512, what, dout);
    run<12>(buf, bytes, 6144, 4, 512, what, dout);
    run<4>(buf, bytes, 6144, 8, 256, what, dout);
    run<12>(buf, bytes, 6144, 8, 256, what, dout);
    run<4>(buf, bytes, 128, 4, 512, pass ? "16 MiB, contiguous rows" : "2 GiB, contiguous rows", dout);
    run<4, 32>(buf, bytes, 6144, 4, 512, what, dout);
    run<4, 64>(buf, bytes, 6144, 4, 512, what, dout);
    run<2, 64>(buf, bytes, 6144, 8, 256, what, dout);
  }
  return 0;
}
