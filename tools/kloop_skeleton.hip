// K-loop skeletons of candidate GEMM tile economies (gfx950), measured chip-wide on random data under the power cap:
//   hipcc --offload-arch=gfx950 -O3 -o tools/kloop_skeleton tools/kloop_skeleton.hip && tools/kloop_skeleton
// VERDICT round 3, item 1 asks for a K loop with fewer LDS reads and DMA issues per MFMA (4 waves x 128x128 with the
// accumulators in AGPRs, ...).  Before a kernel is built on such an economy, this probe runs its SKELETON: per wave and
// K tile (64 wide) the exact instruction mix of the design -- MT x NT x 2 v_mfma_f32_16x16x32_f16 on an MT x NT grid of
// 16x16 accumulator tiles, (MT + NT) x 2 conflict-free ds_read_b128 fragment reads from a swizzled LDS image, NDMA
// buffer_load ... lds pieces of 1 KiB that stream real A panels (256 rows, HBM, re-read by 4 workgroups) and a shared W
// panel (L2), NBAR workgroup barriers -- software-pipelined inside the wave (fragments double-buffered by k-step, reads and
// DMA issues placed between the MFMAs and pinned there), with no epilogue and no real product.  What it reports is the
// MFMA rate each economy can sustain at best, as TFLOP/s and as a fraction of the same loop with the MFMAs alone.
//   design A "x kernel":     8 waves (2 per SIMD), wave tile 128x64:  64 MFMA, 24 reads,  8 DMA per wave and K tile
//   design B "quad AGPR":    4 waves (1 per SIMD), wave tile 128x128: 128 MFMA, 32 reads, 16 DMA, accumulators in AGPRs
//   design C "half tiles":   2 workgroups of 4 waves per CU, 256x128 tiles, wave tile 128x64: 64 MFMA, 24 reads, 12 DMA
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int tile_off_id(int row, int chunk) {
  const int rp = row >> 1;
  return rp * 256 + (((((row & 1) << 3) | chunk) ^ (rp & 15)) << 4);
}
template <bool AG> __device__ __forceinline__ void mfma(f32x4& c, f16x8 a, f16x8 b) {
  if (AG) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
  else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
#define PIN __builtin_amdgcn_sched_barrier(0);

// NW waves per workgroup; MT x NT accumulator tiles per wave; per K tile: NA + NWD DMA pieces (A / W), NBAR barriers;
// RD = 0 drops the fragment reads (fragments stay what the first tile loaded); AG = accumulators in AGPRs; LDSB bytes of LDS
template <int NW, int MT, int NT, int NA, int NWD, int NBAR, int RD, bool AG, int LDSB, int WPE, int FEED>
__global__ __launch_bounds__(NW * 64, WPE) void skel(const char* A, const char* W, int ld, int ktiles, int tiles, int npanels, float* out) {
  __shared__ __attribute__((aligned(16))) char smem[LDSB];
  constexpr int STAGE = LDSB / 2;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c16 = lane & 15, q4 = lane >> 4;
  // fragment read offsets: A image rows [0, 128) of the stage (this wave's 128 rows), W image behind it
  int offA[2], offW[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    offA[ks] = tile_off_id(c16, 4 * ks + q4);
    offW[ks] = (STAGE / 2) + tile_off_id(c16, 4 * ks + q4);
  }
  // fill the LDS once with something that is not zero (the DMA keeps refilling it with panel data)
  for (int i = threadIdx.x; i < LDSB / 4; i += NW * 64) ((unsigned*)smem)[i] = 0x3C003800u + (i * 2654435761u >> 20);
  __syncthreads();
  constexpr int ND_ = NA + NWD;
  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  f16x8 fa[2][MT], fb[2][NT];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
    for (int t = 0; t < MT; ++t) fa[ks][t] = *(const f16x8*)(smem + ((offA[ks] + t * 2048) % STAGE));
#pragma unroll
    for (int t = 0; t < NT; ++t) fb[ks][t] = *(const f16x8*)(smem + ((offW[ks] + t * 2048) % STAGE));
  }
  const int voff = (lane >> 3) * ld + (lane & 7) * 16;     // a DMA piece: 8 rows x 128 bytes
  const unsigned long long ua = (unsigned long long)A, uw = (unsigned long long)W;
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)uw, 0, 0x7FFFFFF0, 0x00020000);
  u32x4 stg[ND_ ? ND_ : 1];
#pragma unroll
  for (int i = 0; i < (ND_ ? ND_ : 1); ++i) stg[i] = (u32x4){0u, 0u, 0u, 0u};
  constexpr int NMM = MT * NT;                 // MFMAs per k-step
  constexpr int NRD = RD ? (MT + NT) : 0;      // reads per k-step (the other k-step's fragments)
  constexpr int ND = NA + NWD;                 // DMA pieces per K tile; half per k-step
  for (int tile = 0; tile < tiles; ++tile) {
    // an A panel is shared by 4 workgroups of ONE XCD (workgroup b runs on XCD b % 8), like the 4 N tiles of an 8x4 patch
    const long panel = ((long)((blockIdx.x >> 5) * 8 + (blockIdx.x & 7)) + (long)tile * (gridDim.x >> 2)) % npanels;
    const char* ap = (const char*)(ua + panel * 256L * ld);
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)ap, 0, 0x7FFFFFF0, 0x00020000);
    for (int kt = 0; kt < ktiles; ++kt) {
      const int st = kt & 1;
      const int kso = __builtin_amdgcn_readfirstlane((kt * 128) % ld);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        // MFMAs of k-step ks on fragment set ks; meanwhile set ks is NOT touched: the reads below refill set ks ^ 1 ...
        // (they are consumed in the next k-step: a wave hides its own LDS latency under its own MFMAs)
#pragma unroll
        for (int i = 0; i < NMM; ++i) {
          const int t = i / NT, u = i % NT;
          mfma<AG>(acc[t][u], fb[ks][u], fa[ks][t]);
          // one read per NMM / NRD MFMAs, one DMA piece per NMM / (ND / 2) MFMAs, pinned where they are written
          if (NRD && (i % (NMM / NRD)) == (NMM / NRD) - 1) {
            const int r = i / (NMM / NRD);
            const char* sb = smem + (st ^ (ks & 1 ? 1 : 0)) * STAGE;
            if (r < MT) fa[ks ^ 1][r] = *(const f16x8*)(sb + ((offA[ks ^ 1] + r * 2048) % (STAGE / 2)));
            else fb[ks ^ 1][r - MT] = *(const f16x8*)(sb + (STAGE / 2) + ((offW[ks ^ 1] - STAGE / 2 + (r - MT) * 2048) % (STAGE / 2)));
          }
          if (ND && (i % (NMM / (ND / 2))) == (NMM / (ND / 2)) / 2) {
            const int d = ks * (ND / 2) + i / (NMM / (ND / 2));      // piece index within the K tile
            const int slot = ((wave * ND + d) * 1024) % STAGE;
            lds_void* dst = (lds_void*)(smem + (st ^ 1) * STAGE + slot);
            if (FEED == 0) {
              if (d < NA) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, dst, 16, voff, kso + ((wave * NA + d) * 8 % 256) * ld, 0, 0);
              else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, dst, 16, voff, kso + ((wave * NWD + d - NA) * 8 % 256) * ld, 0, 0);
            } else {
              // FEED 1: the same piece into REGISTERS (buffer_load_dwordx4), discarded one K tile later; FEED 2: ... and
              // written to the LDS by ds_write_b128 one K tile later (the classic global-read / local-write feed)
              if (FEED == 2) *(u32x4*)(smem + (st ^ 1) * STAGE + ((slot + lane * 16) % STAGE)) = stg[d];
              else asm volatile("; keep %0" ::"v"(stg[d]));
              if (d < NA) stg[d] = __builtin_amdgcn_raw_buffer_load_b128(rsA, voff, kso + ((wave * NA + d) * 8 % 256) * ld, 0);
              else stg[d] = __builtin_amdgcn_raw_buffer_load_b128(rsW, voff, kso + ((wave * NWD + d - NA) * 8 % 256) * ld, 0);
            }
          }
          PIN
        }
        if (NBAR >= 2 || (NBAR == 1 && ks == 1)) {
          if (ND && FEED == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(ND) : "memory");   // the previous K tile's pieces have landed
          __builtin_amdgcn_s_barrier();
        }
        if (NBAR == 4) {   // two more per K tile: mid k-step barriers are modelled as back-to-back ones here
          __builtin_amdgcn_s_barrier();
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  if (s == 12345.678f) out[threadIdx.x] = s;
}

static char* dA; static char* dW; static float* dOut;
static int LD = 2048;                // row stride in bytes: K = 1024 halves per row (argv[1] overrides: a padded stride)
static const int NPANELS = 2048;     // 2048 x 256 rows x 2 KiB = 1 GiB of A
static double bare_tf[2] = {0, 0};

template <int NW, int MT, int NT, int NA, int NWD, int NBAR, int RD, bool AG, int LDSB, int WPE, int FEED = 0>
void run(const char* what, int wgs_per_cu, int bare_slot = -1, int npanels = NPANELS) {
  const int grid = 256 * wgs_per_cu * 4, ktiles = 16, tiles = 8;
  auto k = skel<NW, MT, NT, NA, NWD, NBAR, RD, AG, LDSB, WPE, FEED>;
  hipFuncAttributes fa;
  hipFuncGetAttributes(&fa, (const void*)k);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k, dim3(grid), dim3(NW * 64), 0, 0, dA, dW, LD, ktiles, tiles, npanels, dOut);
  hipDeviceSynchronize();
  const int reps = 5;
  hipEventRecord(e0);
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k, dim3(grid), dim3(NW * 64), 0, 0, dA, dW, LD, ktiles, tiles, npanels, dOut);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  const double flop = (double)grid * NW * tiles * ktiles * 2.0 * MT * NT * (2.0 * 16 * 16 * 32);
  const double tf = flop / (ms * 1e-3) / 1e12;
  if (bare_slot >= 0) bare_tf[bare_slot] = tf;
  const double ref = bare_tf[NW == 4 && MT * NT == 64 ? 1 : 0];
  printf("%-58s feed %d, %d waves x %dx%d tiles, %2d+%2d DMA, %d bar, %s, %3d VGPR %3d AGPR, LDS %6d: %8.3f ms  %7.1f TFLOP/s  %5.1f %% of bare\n",
         what, FEED, NW, MT * 16, NT * 16, NA, NWD, NBAR, RD ? "reads" : "no reads", fa.numRegs, 0, (int)fa.sharedSizeBytes, ms, tf,
         ref > 0 ? 100.0 * tf / ref : 100.0);
  fflush(stdout);
}

int main(int argc, char** argv) {
  if (argc > 1) LD = atoi(argv[1]);
  printf("row stride %d bytes\n", LD);
  const size_t abytes = (size_t)NPANELS * 256 * LD;
  hipMalloc(&dA, abytes + (1 << 20));
  hipMalloc(&dW, (size_t)256 * LD + (1 << 20));
  hipMalloc(&dOut, 4096);
  {   // random fp16 values in (-2, 2): realistic bit toggling (the chip's clock depends on it)
    std::vector<unsigned short> h((abytes + (1 << 20)) / 2);
    unsigned s = 12345u;
    for (size_t i = 0; i < h.size(); ++i) { s = s * 1664525u + 1013904223u; h[i] = (unsigned short)(((s >> 16) & 0x8000u) | (0x3400u + ((s >> 8) & 0x0FFFu))); }
    hipMemcpy(dA, h.data(), abytes + (1 << 20), hipMemcpyHostToDevice);
    hipMemcpy(dW, h.data(), (size_t)256 * LD + (1 << 20), hipMemcpyHostToDevice);
  }
  //          NW MT NT NA NWD NBAR RD AG     LDS    WPE
  run<8, 8, 4, 0, 0, 0, 0, false, 131072, 2>("bare MFMA, 8 waves (2 / SIMD), 128x64 per wave", 1, 0);
  run<4, 8, 8, 0, 0, 0, 0, true, 131072, 1>("bare MFMA, 4 waves (1 / SIMD), 128x128 per wave, AGPR acc", 1);
  run<4, 8, 4, 0, 0, 0, 0, false, 81920, 2>("bare MFMA, 2 WGs x 4 waves, 128x64 per wave", 2, 1);
  printf("-- design A economy (x kernel's mix, but pipelined inside the wave, no stagger)\n");
  run<8, 8, 4, 0, 0, 0, 1, false, 131072, 2>("A: reads only", 1);
  run<8, 8, 4, 4, 4, 0, 1, false, 131072, 2>("A: reads + DMA", 1);
  run<8, 8, 4, 4, 4, 2, 1, false, 131072, 2>("A: reads + DMA + 2 barriers", 1);
  run<8, 8, 4, 4, 4, 4, 1, false, 131072, 2>("A: reads + DMA + 4 barriers", 1);
  printf("-- design B: 4 waves, 128x128 per wave, accumulators in AGPRs\n");
  run<4, 8, 8, 0, 0, 0, 1, true, 131072, 1>("B: reads only", 1);
  run<4, 8, 8, 8, 8, 0, 0, true, 131072, 1>("B: DMA only", 1);
  run<4, 8, 8, 8, 8, 0, 1, true, 131072, 1>("B: reads + DMA", 1);
  run<4, 8, 8, 8, 8, 2, 1, true, 131072, 1>("B: reads + DMA + 2 barriers", 1);
  run<4, 8, 8, 8, 8, 4, 1, true, 131072, 1>("B: reads + DMA + 4 barriers", 1);
  printf("-- design C: two workgroups of 4 waves per CU, 256x128 tiles\n");
  run<4, 8, 4, 0, 0, 0, 1, false, 81920, 2>("C: reads only", 2);
  run<4, 8, 4, 8, 4, 0, 1, false, 81920, 2>("C: reads + 12 DMA", 2);
  run<4, 8, 4, 8, 4, 2, 1, false, 81920, 2>("C: reads + 12 DMA + 2 barriers", 2);
  run<4, 8, 4, 8, 4, 4, 1, false, 81920, 2>("C: reads + 12 DMA + 4 barriers", 2);
  run<4, 8, 4, 4, 4, 2, 1, false, 81920, 2>("C': reads + 8 DMA + 2 barriers (A's DMA count)", 2);
  run<4, 8, 4, 8, 4, 2, 1, false, 81920, 2>("C alone: ONE workgroup per CU (its partner in an epilogue)", 1);
  printf("-- where the pieces come from: 8 A panels = 4 MiB in all (L2 / MALL hits only) instead of 1 GiB streamed from HBM\n");
  run<8, 8, 4, 4, 4, 2, 1, false, 131072, 2>("A: reads + DMA + 2 barriers, hot panels", 1, -1, 8);
  run<4, 8, 8, 8, 8, 2, 1, true, 131072, 1>("B: reads + DMA + 2 barriers, hot panels", 1, -1, 8);
  run<4, 8, 4, 8, 4, 2, 1, false, 81920, 2>("C: reads + 12 DMA + 2 barriers, hot panels", 2, -1, 8);
  printf("-- the feed itself: the same pieces as LDS-DMA (0), into registers and dropped (1), registers + ds_write_b128 (2)\n");
  run<4, 8, 8, 8, 8, 2, 1, true, 131072, 1, 0>("B: LDS-DMA", 1);
  run<4, 8, 8, 8, 8, 2, 1, true, 131072, 1, 1>("B: buffer_load_dwordx4 to registers, dropped", 1);
  run<4, 8, 8, 8, 8, 2, 1, true, 131072, 1, 2>("B: buffer_load_dwordx4 + ds_write_b128", 1);
  run<4, 8, 8, 8, 8, 2, 0, true, 131072, 1, 1>("B: registers, dropped, no fragment reads", 1);
  return 0;
}
