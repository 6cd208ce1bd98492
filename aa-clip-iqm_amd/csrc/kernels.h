// Internal launch interface between the C-ABI (capi.hip) and the kernel files.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/aaclip.h"

namespace aaclip {

enum { EPI_BIAS = 0, EPI_BIAS_GELU = 1, EPI_BIAS_RESID = 2, EPI_ACT_F32 = 3, EPI_PATCH = 4 };

struct GemmParams {
  const void* A;  // [M, lda] compute dtype
  long lda;
  const void* W;  // [N, K] compute dtype
  int M, N, K;
  const float* bias;  // [N] fp32 or null
  void* out;          // compute dtype or fp32, by epilogue
  long ldc;
  int scale_cols;  // EPI_BIAS: columns < scale_cols are multiplied by scale
  float scale;
  int act;           // EPI_ACT_F32: 0 none, 1 LeakyReLU(0.01)
  const float* pos;  // EPI_PATCH: positional embedding [L, N]
  int P, L;          // EPI_PATCH: patches per image, tokens per image
  const float* resid;      // EPI_BIAS_RESID: where the residual is READ ([M, ldc] fp32); null = in place, from `out`
  // LayerNorm folding (16x16x32 256-tile kernels only; all null = off), see capi.hip aaclip_block:
  void* out16;             // EPI_BIAS_RESID: also write the new residual rows in the compute dtype, [M, N]
  float* stats_out;        // EPI_BIAS_RESID: per row and 64-column slice (sum, sum of squares) of the new rows, [M][N/64][2]
  const float* row_ab;     // EPI_BIAS / EPI_BIAS_GELU: per row (a, b); value = a*acc + b*col_s[n] + bias[n]
  const float* col_s;      //   [N] row sums of the (gamma-scaled) weight
  // AACLIP_F16X2 (split fp16, common.h): A is [M, >= 2K] (hi | lo per row, lda = row stride), W is [N, 2K] (hi | lo)
  // or, with w_exact16, [N, K] (the weight is exact in fp16: the Ah.Wl product is skipped); 16-bit outputs are split
  // rows too ([M, >= 2N], ldc = row stride, lo plane N columns after the hi plane).
  int w_exact16;
  int out_qk8;                // EPI_BIAS, split fp16, 256-tile kernel only: > 0 = the output is the attention kernel's
                              // [hi: N fp16][per 64 columns below out_qk8: lo8 64 B | hi8 64 B] record (ldc >= (N + out_qk8 / 2) halves... see gemm256t.hip)
  int out_no_hi8;             // EPI_BIAS_GELU, split fp16: the consumer's weight is exact in fp16 -- skip the hi8 plane
};

const char* gemm_check(int dtype, int epi, const GemmParams& p);
void launch_gemm(int dtype, int epi, const GemmParams& p, hipStream_t s);
bool gemm256_applicable(int dtype, const GemmParams& p);
// 256-tile kernels on 16x16x32 MFMAs.  kernel_id: 14 = the default two-fragment-set kernel (falls back to 13 when
// K/64 is odd), 13 = its one-set predecessor; every other id exists only in the measurement library.  A (kernel_id,
// epilogue) pair without a kernel launches NOTHING and records a launch error (launch_error()).
void launch_gemm256t(int dtype, int epi, const GemmParams& p, hipStream_t s, int kernel_id);
// Kernel selection for A/B runs.  The product library knows GEMM variants 0 (automatic) and 1 (128-tile kernel) and
// attention variants 0 and 1 (128-query kernel); both return false for anything else and leave the selection alone.
bool set_gemm_variant(int v);
bool set_attn_variant(int v);
void set_tail_peel(int v);
// Sticky per-thread launch error: set by a launcher that was asked for a kernel it does not have; capi's finish()
// reports and clears it, so the entry point returns rc < 0 instead of running something else.
void set_launch_error(const char* msg);
const char* take_launch_error();
#ifdef AACLIP_MEASURE
// Measurement library only (libaaclip_hip_measure.so, `make measure`): A/B variants, timing ablations that compute
// WRONG results, and s_memtime stamp builds.  None of this is compiled into libaaclip_hip.so.
void launch_gemm256(int dtype, int epi, const GemmParams& p, hipStream_t s, int pipelined);
void read_gemm_zstamps(double* out8);   // -DZ_STAMP builds only, zeros otherwise
bool launch_gemm256z(int dtype, int epi, const GemmParams& p, hipStream_t s);   // persistent tiles; false = not applicable
void read_gemm_stamps(double* out3, int nwaves);
void read_gemm_estamps(double* out8);   // -DX_WALK_STAMP builds of the measurement library only, zeros otherwise
void read_attn_passes(unsigned long long* out4, int reset);
void read_attn_stamps(unsigned long long* out9, int reset);   // -DATTN_STAMP builds: cycles per tile-loop segment   // attn16x2 tile passes: [0] tile 0, [1] fast, [2] exact redo
#endif

// fused softmax(q k^T) v over packed qkv [B*L, 3*H*64] (q pre-scaled) -> ctx [B*L, H*64]
// log2q != 0: q is pre-multiplied by log2(e) as well (16-bit kernels only)
void launch_attention(int dtype, const void* qkv, void* ctx, int B, int L, int H, int causal, int log2q,
                      hipStream_t s, bool hi8 = true, bool qk8 = false);
// qk8 (split fp16, log2q, L >= 512 only): qkv rows are [q k v hi: 3D fp16][q8][k8] records of 10 D bytes (attention.hip,
// GemmParams::out_qk8) instead of split16 rows
bool attention_qk8_applicable(int L, int causal);

// row ops (rowops.hip); D in {256, 768, 1024}
const char* row_width_check(int D);
// hi8 = false (split fp16 only): do not write the hi8 plane of the split8 rows -- for a consumer whose weight is exact
// in fp16 and therefore never reads it (common.h); the same flag exists on every producer of split8 rows
void launch_layernorm(int out_dtype, const float* x, const float* w, const float* b, void* out, long rows, int D,
                      float eps, hipStream_t s, bool hi8 = true);
void launch_adapter_mix(float* x, const float* a, long rows, int D, float weight, hipStream_t s);
void launch_adapter_mix_fold(int dtype, float* x, const float* a, long rows, int D, float weight, void* out16,
                             float* rowab, hipStream_t s);   // also emits the 16-bit rows and (rstd, -mean*rstd)
void launch_im2col(int dtype, const float* img, void* cols, int B, int C, int H, int W, int ps, int Kpad,
                   hipStream_t s);
void launch_cls_rows(float* x, const float* cls, const float* pos, int B, int L, int D, hipStream_t s);
void launch_embed_text(const int32_t* tokens, const float* table, const float* pos, float* x, int n, int T, int D,
                       int vocab, hipStream_t s);
void launch_gather_rows(int dtype, const void* src, void* dst, const int32_t* tokens, int n, int T, int D, int mode,
                        hipStream_t s);
void launch_normalize_rows(const float* src, float* dst, int B, int L, int skip, int E, hipStream_t s);
void launch_det_mean(const float* src, float* scratch, size_t scratch_floats, float* dst, int B, int L, int skip, int E,
                     hipStream_t s);

// anomaly map (anomaly_map.hip)
// mode 0: test-mode map m = (s1 + 1 - s0)/2 -> pre [B, P]; mode 1: raw scores -> [B, 2, P]
void launch_patch_scores(const float* seg, const float* anchors, long anchor_bstride, float* pre, int B, int P, int E,
                         int mode, hipStream_t s);
// pre [NL][B, g, g] -> out [B, S, S] = sum over levels of upsample(blur(pre_l)); NL <= 4, g <= 40
void launch_blur_upsample(const float* pre, float* out, int B, int g, int S, int NL, int ksize, float sigma,
                          hipStream_t s);
void launch_upsample_softmax2(const float* scores, float* out, int B, int g, int S, hipStream_t s);
void launch_cast_rows(int dtype, const float* src, void* dst, long n, hipStream_t s);
void launch_split_rows(const float* src, void* dst, long rows, int D, hipStream_t s, bool hi8 = true);   // fp32 [rows, D] -> split fp16 [rows, 2D]
// LayerNorm folding: [M][slots][2] partial (sum, sumsq) -> [M][2] (rstd, -mean*rstd)
void launch_ln_stats_finalize(const float* partials, float* ab, long rows, int slots, int D, float eps, hipStream_t s);
bool gemm_routes_to_256t(int dtype, const GemmParams& p);   // launch_gemm will run a kernel with the folding epilogue
bool gemm_split_routes_to_256t(const GemmParams& p);        // split fp16: ... the 256-tile kernel (out_qk8 epilogue)
// V-V "surgery" attention over the batch axis: regroup v [B*L,D] -> packed q|k|v rows l*B+b and back
void launch_vv_spread(int dtype, const void* v, void* qkv, int B, int L, int D, float scale, hipStream_t s);
void launch_vv_regroup(int dtype, const void* src, void* dst, int B, int L, int D, hipStream_t s);

// ---- iqm.hip : the IQM side branch's small kernels (include/aaclip.h, "IQM side branch")
const char* small_attention_check(int nq, int Lk, int H, int hd);
void launch_small_attention(int kv_dtype, const float* q, const void* k, const void* v, float* out, int B, int nq, int Lk,
                            int H, int hd, float scale, hipStream_t s);
const char* cross_rows_check(int R, int Lk, int Dk);
size_t cross_rows_ws_bytes(int B, int R, int Lk, int Dk);
void launch_cross_rows(int x_dtype, const float* qt, const void* x, float* out, void* ws, int B, int R, int Lk, int Dk,
                       hipStream_t s);
const char* cross_rows_levels_check(int x_dtype, int R, int nseg, int Lk, int Dk, long ldx);
size_t cross_rows_levels_ws_bytes(int B, int nseg, int Lk, int Dk);
void launch_cross_rows_levels(int x_dtype, const float* qt, const void* const* x, int nseg, float* out, void* ws, int B,
                              int R, int rows_per_image, int row0, int Lk, int Dk, long ldx, hipStream_t s);
void launch_head_expand(int dtype, const float* q, void* qm, long rows, int H, int D, float scale, hipStream_t s);
void launch_head_diag(const float* full, float* ctx, long rows, int H, int D, hipStream_t s);
void launch_residual_layernorm(const float* a, const float* b, const float* w, const float* bias, float* out, long rows,
                               int D, float eps, hipStream_t s);
void launch_combine3(const float* a, const float* b, const float* c, float wa, float wb, float wc, float* out, long n,
                     hipStream_t s);
void launch_linear_smallk(int out_dtype, const float* x, const float* W, const float* bias, void* y, long R, int N, int K,
                          hipStream_t s);
void launch_drop_cls_rows(int dtype, const void* src, void* dst, int B, int L, int E, int rows_per_image, int row_off,
                          hipStream_t s);
void launch_iqm_scores(const float* seg, const float* q, float* grid, int B, int P, int E, hipStream_t s);
void launch_iqm_upsample(const float* grids, const float* base, float* out, int B, int g, int S, int NL, float w_base,
                         float w_iqm, hipStream_t s);

// ---- preprocess.hip : 8-bit bicubic resize + ToTensor + Normalize (Pillow-exact)
int resample_ksize(int in_size, int out_size);
void resample_table(int in_size, int out_size, int32_t* bounds, int32_t* coefs);   // host buffers
int preprocess_tile_rows(int in_size, int out_size, int ty);
int preprocess_row_pitch(int in_w, int out_size);
void launch_preprocess(const uint8_t* src, int B, int Hs, int Ws, int S, const int32_t* hb, const int32_t* hk, int kx,
                       const int32_t* vb, const int32_t* vk, int ky, int TY, int lds_rows, int pitch, const float* lut,
                       float* out, hipStream_t s);

}  // namespace aaclip
