/* aaclip.h -- C ABI of libaaclip_hip.so: the MI355X (gfx950) AA-CLIP inference hot path.
 *
 * The reference (liu20050510/AA-CLIP-IQM) has no FFI of its own: its callers reach
 * into nn.Module attributes.  This header is therefore the boundary a binding for
 * the reference's hot path would use; every entry point names the reference code it
 * replaces (file:line under the reference root).  INTEGRATION.md shows the ctypes
 * stub a maintainer would add on the reference side.
 *
 * Conventions
 *  - All pointers are DEVICE pointers into memory owned by the caller (torch tensors
 *    on the Python side); the library allocates nothing it returns.  Scratch is a
 *    caller-provided workspace (aaclip_workspace_bytes).
 *  - `stream` is a hipStream_t passed as void*; every call only enqueues work on it.
 *  - `dtype` selects the arithmetic type of the matrix products:
 *      AACLIP_F32  exact fp32 MFMA (v_mfma_f32_32x32x2_f32), parity path
 *      AACLIP_F16 / AACLIP_BF16  16-bit operands, fp32 accumulate (v_mfma_f32_32x32x16)
 *      AACLIP_F16X2  split fp16: every matrix-product operand is carried as hi = fp16(v) plus a correction for v - hi
 *                    and a product is accumulated into one fp32 accumulator as main term + correction terms -- the mode
 *                    that meets 1e-3 abs + 1e-2 rel against the fp32 reference on taps and anomaly maps at MFMA speed.
 *                    GEMMs: A.W^T = Ah.Wh^T (fp16 MFMA) + Al8.Wh8^T + Ah8.Wl8^T (block-scaled e4m3 MFMA,
 *                    v_mfma_scale_f32_16x16x128_f8f6f4, twice the fp16 rate; the corrections are 2^-11 of the main
 *                    term, so 4 significant bits in them leave ~2^-15 per operand).  Attention: q.k^T = Kh.Qh + Kl.Qh
 *                    + Kh.Ql on the fp16 MFMAs, p.v with p and v in fp16.  Row formats, 4 bytes per element, row
 *                    strides counted in halves (>= 2C for a logical width C):
 *                      split8  [hi: C x fp16][lo8: C x e4m3 = (v - hi) * 2^10][hi8: C x e4m3 = v]
 *                              the A operand of aaclip_gemm, the context rows aaclip_attention writes, the output of
 *                              AACLIP_EPI_BIAS_GELU and of aaclip_layernorm(out_dtype = AACLIP_F16X2);
 *                              weights [out, in] likewise: [Wh: in x fp16][e4m3(W * 2^6)][e4m3((W - Wh) * 2^17)]
 *                      split16 [hi: C x fp16][lo: C x fp16 = v - hi]
 *                              the packed q|k|v rows aaclip_attention reads (C = 3 * H * 64) = the output of
 *                              AACLIP_EPI_BIAS.  (Inside aaclip_block / aaclip_blocks*, for rows of >= 512 tokens, q and k
 *                              travel from the QKV product to the attention kernel as fp16 + a 128-byte
 *                              [lo8 | hi8] e4m3 record per head instead, and the two correction products of q.k^T run
 *                              on v_mfma_scale_f32_32x32x64_f8f6f4 -- a workspace-internal format, not part of this
 *                              interface.)
 *                    e4m3 = OCP e4m3fn; the scales are fixed (activations 0.016 ... 448 and weights 2.4e-4 ... 7 keep
 *                    >= 4 significant bits in their correction operands; beyond, that element's correction degrades
 *                    towards plain fp16, never the main term).  A weight whose values are exact in fp16 (OpenAI's
 *                    CLIP checkpoints are stored in fp16) and whose in_features is a multiple of 256 may be passed
 *                    WITHOUT its last plane ([out, 3 * in] bytes) where an `exact16` flag exists
 *                    (aaclip_block_weights): the Ah8.Wl8^T product is then skipped.  K must be a multiple of 128.
 *    Weights of matrix products are passed already converted to `dtype`, row-major
 *    [out_features, in_features] exactly like nn.Linear.weight.  LayerNorm
 *    parameters, biases, embeddings, the residual stream and all outputs are fp32.
 *  - Return value: 0 on success, negative on error; aaclip_last_error() returns a
 *    thread-local message.  No exceptions cross the ABI.
 */
#ifndef AACLIP_H
#define AACLIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AACLIP_ABI_VERSION 5   /* 5: + aaclip_tap_head_keep_rows, aaclip_cross_rows_levels (additions only) */

enum { AACLIP_F32 = 0, AACLIP_F16 = 1, AACLIP_BF16 = 2, AACLIP_F16X2 = 3 };
enum { AACLIP_ACT_NONE = 0, AACLIP_ACT_LEAKY = 1, AACLIP_ACT_RELU = 2 };
/* generic GEMM epilogues (aaclip_gemm) */
enum { AACLIP_EPI_BIAS = 0, AACLIP_EPI_BIAS_GELU = 1, AACLIP_EPI_BIAS_RESID = 2, AACLIP_EPI_ACT_F32 = 3 };

int aaclip_version(void);            /* AACLIP_ABI_VERSION the library was built from; bind only if it matches */
const char* aaclip_last_error(void);
/* 0 for libaaclip_hip.so.  1 for libaaclip_hip_measure.so, the separate build (`make measure`) that adds A/B kernel
 * variants, timing ablations that compute WRONG results and s_memtime stamp kernels: never bind a product to it. */
int aaclip_is_measurement_build(void);

/* Scratch needed by any call below for `rows` token rows of width D, MLP width F,
 * embed width E (pass the largest you will use). */
size_t aaclip_workspace_bytes(int dtype, long rows, int D, int F, int E);

/* Weights of one ResidualAttentionBlock (reference model/transformer.py:183-258)
 * plus the optional residual adapter applied after it (model/adapter.py:163-170). */
typedef struct aaclip_block_weights {
  /* ABI version >= 3: sizeof(aaclip_block_weights) as the CALLER compiled it.  Every entry point that takes this
   * struct rejects (rc < 0, nothing launched) an element whose struct_bytes differs from the library's own sizeof, so
   * a binding generated from an older header (13 or 19 pointer fields, no size field: its first word is the ln1_w
   * pointer; or version 3's 20 fields without `exact16`) is refused instead of being read past its end. */
  size_t struct_bytes;
  const float* ln1_w;   /* [D] */
  const float* ln1_b;
  const void* qkv_w;    /* attn.in_proj_weight [3D, D], dtype */
  const float* qkv_b;   /* attn.in_proj_bias [3D] */
  const void* out_w;    /* attn.out_proj.weight [D, D], dtype */
  const float* out_b;
  const float* ln2_w;
  const float* ln2_b;
  const void* fc_w;     /* mlp.c_fc.weight [F, D], dtype */
  const float* fc_b;
  const void* proj_w;   /* mlp.c_proj.weight [D, F], dtype */
  const float* proj_b;
  const void* adapter_w; /* SimpleAdapter fc.0.weight [D, D], dtype; NULL = no adapter */
  /* Optional (all three or none; ABI version >= 2): ln_2 folded into c_fc.  When present and the block runs
   * on the large-batch 16-bit kernels, the ln_2 pass disappears: out_proj's epilogue also emits the new
   * residual rows in `dtype` with per-row sums, and c_fc computes
   *   rstd_m * (x16 . fc_w_fold^T)_mn - rstd_m * mean_m * fc_fold_s[n] + fc_fold_b[n]
   * which equals c_fc(ln_2(x)) up to rounding.  Otherwise the fields are ignored. */
  const void* fc_w_fold;  /* [F, D] dtype: c_fc.weight * ln_2.weight[None, :] */
  const float* fc_fold_s; /* [F]: row sums of fc_w_fold (as stored, in fp32) */
  const float* fc_fold_b; /* [F]: c_fc.bias + c_fc.weight @ ln_2.bias */
  /* The same for ln_1 and the QKV product; used by aaclip_blocks for every block after the first of a call
   * (the previous block's last epilogue, or its adapter mix, emits the 16-bit rows and their statistics). */
  const void* qkv_w_fold;  /* [3D, D] dtype: in_proj_weight * ln_1.weight[None, :] */
  const float* qkv_fold_s; /* [3D] */
  const float* qkv_fold_b; /* [3D]: in_proj_bias + in_proj_weight @ ln_1.bias */
  /* ABI version >= 4, AACLIP_F16X2 only (ignored otherwise): bit mask of the matrix weights above that are passed in
   * the 3-plane form (fp16 plane + e4m3 plane, 3 bytes per element) because every value is exact in fp16; the others
   * carry all three planes (4 bytes per element). */
  unsigned exact16;
} aaclip_block_weights;
enum { AACLIP_EXACT16_QKV = 1, AACLIP_EXACT16_OUT = 2, AACLIP_EXACT16_FC = 4, AACLIP_EXACT16_PROJ = 8,
       AACLIP_EXACT16_ADAPTER = 16 };

/* Patch embedding + class token + positional embedding + ln_pre.
 * Replaces reference model/adapter.py:139-156 (== model/transformer.py:507-526).
 * img [B,3,H,W] fp32 NCHW; conv_w = conv1.weight reshaped [D, 3*ps*ps] and zero
 * padded to [D, Kpad], Kpad = round_up(3*ps*ps, 64), dtype; pos [L, D];
 * x (out) [B*L, D] fp32 with L = (H/ps)*(W/ps)+1. */
int aaclip_patch_embed(const float* img, const void* conv_w, const float* cls, const float* pos,
                       const float* ln_pre_w, const float* ln_pre_b, float* x, int B, int H, int W, int ps, int D,
                       int dtype, void* ws, size_t ws_bytes, void* stream);

/* One pre-LN residual attention block, in place on the fp32 residual stream
 * x [B*L, D]:  x += out_proj(MHA(ln_1 x));  x += c_proj(gelu_erf(c_fc(ln_2 x)));
 * then, if w->adapter_w: a = LeakyReLU(x Wa^T); x = mix*a*|x|/|a| + (1-mix)*x.
 * Replaces ResidualAttentionBlock.forward (reference model/transformer.py:239-258,
 * nn.MultiheadAttention at :200,237) and the adapter lines model/adapter.py:162-170
 * (visual, AACLIP_ATTN_FULL) / :285-295 (text, AACLIP_ATTN_CAUSAL: mask of model/transformer.py:629-635).
 * AACLIP_ATTN_VV_BATCH: the block after VisionTransformer.DAPM_replace (reference
 * model/transformer.py:406-425): its attention is the "surgery" module (:102-152) with
 * q = k = v = value projection, which -- fed the LND stream -- attends over the BATCH axis per
 * token position.  Reproduced as the reference runs it (outputs depend on the batch); needs F >= 4*D.
 * H heads of 64; D = 64*H. */
enum { AACLIP_ATTN_FULL = 0, AACLIP_ATTN_CAUSAL = 1, AACLIP_ATTN_VV_BATCH = 2 };
int aaclip_block(float* x, const aaclip_block_weights* w, float mix, int B, int L, int D, int H, int F, int attn_mode,
                 int dtype, void* ws, size_t ws_bytes, void* stream);

/* n_blocks consecutive blocks (w[0..n_blocks-1], same mix / shape / attn_mode) in one call, in place on x.
 * Identical to n_blocks aaclip_block calls, except that nothing else can touch x in between, which lets the
 * library fold ln_1 of blocks 1..n-1 into their QKV products (see aaclip_block_weights).  Callers split the
 * tower at the layers whose output they read (reference model/adapter.py:171-172 taps). */
int aaclip_blocks(float* x, const aaclip_block_weights* w, int n_blocks, float mix, int B, int L, int D, int H, int F,
                  int attn_mode, int dtype, void* ws, size_t ws_bytes, void* stream);

/* As aaclip_blocks, but the stream is READ from x_in, which stays untouched, and continued in x (the first residual
 * update reads x_in and writes x; x_in == x is aaclip_blocks).  This is how a caller keeps the tower's state at a
 * tap layer -- CLIP.encode_image(image, out_layers), reference model/transformer.py:295-317 -- without copying it:
 * the tap tensor is the buffer the previous run wrote, the next run writes a fresh one. */
int aaclip_blocks_to(const float* x_in, float* x, const aaclip_block_weights* w, int n_blocks, float mix, int B, int L,
                     int D, int H, int F, int attn_mode, int dtype, void* ws, size_t ws_bytes, void* stream);

/* A whole tower with taps in ONE call: x_out[i] is the buffer that holds the stream AFTER block i; block i reads
 * the stream from x_in (i = 0) or from x_out[i-1] and writes x_out[i] (in place when they are the same buffer; a buffer
 * that is not written again keeps the tower's state at that layer).  Equivalent to one aaclip_blocks_to call per run
 * of blocks between taps, except that the library sees the whole sequence, so ln_1 of the first block after a tap is
 * folded into its QKV product like every other (those calls run it as a LayerNorm pass: nothing tells them that the
 * workspace still holds the previous call's rows).  Reference: model/transformer.py:295-317 (out_layers),
 * model/adapter.py:160-172 (levels). */
int aaclip_blocks_taps(const float* x_in, float* const* x_out, const aaclip_block_weights* w, int n_blocks, float mix,
                       int B, int L, int D, int H, int F, int attn_mode, int dtype, void* ws, size_t ws_bytes,
                       void* stream);

/* Tap head: ln_post -> seg_proj (Linear no bias [+LeakyReLU]) -> F.normalize, CLS row
 * dropped.  Replaces reference model/adapter.py:171-182.  x [B*L, D] fp32 (tap of the
 * residual stream); proj_w [E, D] dtype; seg_out [B, L-1, E] fp32 unit rows.
 * If det_w != NULL also writes det_out [B, E] = mean over patches of
 * F.normalize(det_proj(ln_post x)) (model/adapter.py:183-184). */
int aaclip_tap_head(const float* x, const float* ln_post_w, const float* ln_post_b, const void* proj_w, int act,
                    float* seg_out, const void* det_w, float* det_out, int B, int L, int D, int E, int dtype, void* ws,
                    size_t ws_bytes, void* stream);

/* The same, and the LayerNorm'ed rows ln_post(x) [B*L, D] are left in ln_rows_out in the layout of `dtype` (fp32 /
 * fp16 / bf16 rows of D elements; AACLIP_F16X2: split8 rows of 4 D bytes whose first 2 D bytes are the fp16 values):
 * the IQM branch reads exactly these rows again (reference model/adapter.py:205-208 applies ln_post to every tap a
 * second time), so they are computed once. */
int aaclip_tap_head_keep_rows(const float* x, const float* ln_post_w, const float* ln_post_b, const void* proj_w, int act,
                              float* seg_out, const void* det_w, float* det_out, void* ln_rows_out, int B, int L, int D,
                              int E, int dtype, void* ws, size_t ws_bytes, void* stream);

/* Detection head alone (reference model/adapter.py:183-184). */
int aaclip_det_head(const float* x, const float* ln_post_w, const float* ln_post_b, const void* det_w, int act,
                    float* det_out, int B, int L, int D, int E, int dtype, void* ws, size_t ws_bytes, void* stream);

/* Fused test-mode anomaly map over NL tap levels: per level s = 100 f.t,
 * m = (s1 + 1 - s0)/2, Gaussian blur (ksize, sigma; reflect), bilinear
 * align_corners=True to S x S; out [B,S,S] = sum over levels.  Replaces
 * calculate_similarity_map(test=True) (reference forward_utils.py:196-213) applied per
 * level and the level sum of test_last.py:95-100,149.  seg[l] [B, g*g, E] unit rows;
 * anchors [E,2] (anchor_bstride 0) or [B,E,2] (anchor_bstride E*2).  NL <= 4, g <= 40.
 * ksize 1 skips the blur. */
int aaclip_anomaly_map(const float* const* seg, int NL, const float* anchors, long anchor_bstride, float* out, int B,
                       int g, int E, int S, int ksize, float sigma, void* ws, size_t ws_bytes, void* stream);

/* Train-mode similarity map: bilinear on both channels then softmax over C=2,
 * out [B,2,S,S].  Replaces calculate_similarity_map(test=False)
 * (reference forward_utils.py:199-203,211-215). */
int aaclip_similarity_map_train(const float* seg, const float* anchors, long anchor_bstride, float* out, int B, int g,
                                int E, int S, void* ws, size_t ws_bytes, void* stream);

/* Image pre-processing in front of the patch embed: Pillow's 8-bit BICUBIC resize to S x S,
 * ToTensor (v/255) and Normalize((v - mean)/std), bit-exact.  Replaces the reference's
 * dataset transform (reference dataset/__init__.py:150-161 and :62-71: transforms.Resize(
 * (S,S), Image.BICUBIC) -> ToTensor -> Normalize), which runs in the DataLoader workers on CPU.
 *   aaclip_resample_ksize / aaclip_resample_table are HOST functions: they fill host buffers
 *   bounds[2*out] (first source index, tap count) and coefs[out*ksize] (22-bit fixed point) with
 *   Pillow's weights for one axis; the caller uploads them once per (source size, S).
 *   aaclip_preprocess: src uint8 [B,Hs,Ws,3] (HWC, device), tables for the horizontal (Ws -> S)
 *   and vertical (Hs -> S) pass (device), lut fp32 [3,256] = normalised value of every byte per
 *   channel (device), out fp32 [B,3,S,S]. */
int aaclip_resample_ksize(int in_size, int out_size);
int aaclip_resample_table(int in_size, int out_size, int32_t* bounds, int32_t* coefs);
int aaclip_preprocess(const uint8_t* src, int B, int Hs, int Ws, int S, const int32_t* hbounds, const int32_t* hcoefs,
                      const int32_t* vbounds, const int32_t* vcoefs, const float* lut, float* out, void* stream);

/* Text embedding: x[i*T+t] = token_embedding[tokens[i,t]] + positional_embedding[t].
 * Replaces reference model/adapter.py:277-281 (model/model.py:192-194). */
int aaclip_text_embed(const int32_t* tokens, const float* table, const float* pos, float* x, int n, int T, int D,
                      int vocab, void* stream);

/* Row head: LayerNorm, pick one row per sequence, project.
 *   mode 0: row at argmax(tokens[i,:]) (EOT) -- ln_final + text_adapter[-1] / text_projection,
 *           reference model/adapter.py:297-299 and model/model.py:198-200
 *   mode 1: row 0 (CLS) -- ln_post + visual.proj, reference model/transformer.py:542-546
 * x [n*T, D] fp32; proj_w [E, D] dtype (pass text_projection / visual.proj transposed);
 * out [n, E] fp32. */
int aaclip_row_head(const float* x, const int32_t* tokens, const float* ln_w, const float* ln_b, const void* proj_w,
                    int act, float* out, int n, int T, int D, int E, int mode, int dtype, void* ws, size_t ws_bytes,
                    void* stream);

/* In-stream timing of the kernels inside aaclip_block for the benchmark's roofline leg:
 * begin(tag_mask, capacity) arms hipEvent pairs around kernels whose tag bit is set
 * (0 layernorm, 1 qkv gemm, 2 attention, 3 out_proj gemm, 4 c_fc gemm, 5 c_proj gemm,
 * 6 adapter); end() synchronises the recorded events and returns how many (ms[i], tags[i])
 * pairs it wrote.  Not part of the reference's surface; measurement only. */
int aaclip_profile_begin(unsigned tag_mask, int capacity);
int aaclip_profile_end(float* ms, int* tags, int max_n);

/* Kernel selection for A/B measurements (tools/bench_gemm.py, tools/bench_attn.py); 0 = automatic (default).
 * bits 0..7   GEMM: 1 = always the 128x128-tile kernel; 80 / 81 / 82 = where the automatic choice is a 256-row-tile
 *             kernel, always the 8-wave 256x256 one with one tile per workgroup / the 4-wave 256x128 half-tile one /
 *             the 8-wave one walking its tiles (split operands; one workgroup per CU) -- bit-identical results; 0
 *             picks the fastest.  Measurement library only: 2..5 = 256-tile kernels on
 *             32x32x16 MFMAs; 6..60 = the 16x16x32 family (20 = the default kernel, others: lock-step /
 *             in-cluster-read variants, timing ablations and the stamp build); 70 = persistent tiles (gemm256z.hip)
 * bits 8..15  attention: 1 = always the 128-query kernel; measurement library only: 2 = software-pipelined kernel
 * bit 16      peel the partial last round of 256-tile GEMMs to the 128-tile kernel (off by default: -1.6 %)
 * bit 17      turn the LayerNorm folding of aaclip_block(s) off
 * Every selectable kernel of libaaclip_hip.so computes the same function.  A value that names a variant the loaded
 * library does not contain is rejected (rc < 0) and leaves the selection unchanged; inside the measurement library a
 * (variant, epilogue) pair without a kernel makes the affected call return rc < 0 instead of running a substitute. */
int aaclip_set_gemm_variant(int v);

/* Diagnostics of stamp builds (tools/gemm_stamps.py, tools/gemm_zstamps.py): nwaves >= 0 -> 6 averaged s_memtime
 * segment sums of GEMM variant 17; nwaves < 0 -> 8 values of the persistent kernel built with -DZ_STAMP (zeros in
 * a normal build).  `out` is HOST memory.  rc < 0 in libaaclip_hip.so (measurement library only). */
int aaclip_debug_gemm_stamps(double* out, int nwaves);

/* Building blocks, exported for unit parity tests and for callers that fuse differently. */
int aaclip_layernorm(const float* x, const float* w, const float* b, void* out, int out_dtype, long rows, int D,
                     float eps, void* stream);
int aaclip_gemm(int dtype, int epi, const void* A, long lda, const void* W, const float* bias, void* out, long ldc,
                int M, int N, int K, int act, int scale_cols, float scale, void* stream);
int aaclip_attention(int dtype, const void* qkv, void* ctx, int B, int L, int H, int causal, void* stream);
/* The same with q already in log2 units: q = (x Wq^T + b) * head_dim^-1/2 * log2(e), rounded to `dtype` once -- what
 * the QKV epilogue of aaclip_block produces (16-bit dtypes only).  This is the kernel variant the block path runs
 * (one v_exp_f32 per score, no per-score multiply); exported so that it can be tested and timed on its own. */
int aaclip_attention_log2q(int dtype, const void* qkv, void* ctx, int B, int L, int H, int causal, void* stream);
int aaclip_adapter_mix(float* x, const float* a, long rows, int D, float weight, void* stream);

/* ---- IQM side branch (reference model/iqm.py, the glue of model/adapter.py:186-269 and the IQM maps of
 * test_last.py:102-147).  Its matrix products -- class_query_mlp, query_adapters,
 * visual_feature_proj, every query / key / value / dense / intermediate Linear of IQMLayer -- go through aaclip_gemm
 * (AACLIP_EPI_BIAS for 16-bit key / value / patch operands, AACLIP_EPI_BIAS_GELU for intermediate_query,
 * AACLIP_EPI_ACT_F32 with a bias for the 2-row query side).  The entry points below are what is left. */

/* Cross-attention of IQM without the key / value projections of the patch rows (reference model/iqm.py:108-139 with
 * encoder_hidden_states = the 4 x 1369 projected patch rows; the reference projects every row through W_k and W_v):
 *   scores_j = q_h . (W_k[h] x_j + b_k[h]) / sqrt(d) = (W_k[h]^T q_h / sqrt(d)) . x_j + const  (softmax-invariant)
 *   ctx_h    = sum_j p_j (W_v[h] x_j + b_v[h])       = W_v[h] (sum_j p_j x_j) + b_v[h]
 * aaclip_head_expand builds the A operand of the effective-query product (q [rows, D] fp32 -> [rows * H, D] dtype, row
 * (r, h) = q[r] * scale on head h's columns, zero elsewhere; qt = that . W_k through aaclip_gemm);
 * aaclip_cross_rows computes out[b, r] = softmax_j(qt[b, r] . x[b, j]) . x[b] for R = queries x heads effective
 * queries per image over the raw rows x [B * Lk, Dk] (x_dtype), fp32 accumulation, qt / out [B, R, Dk] fp32;
 * aaclip_head_diag picks the head-diagonal blocks of the [rows * H, D] product W_v . ebar + b_v -> ctx [rows, D]. */
size_t aaclip_cross_rows_workspace_bytes(int B, int R, int Lk, int Dk);
int aaclip_cross_rows(int x_dtype, const float* qt, const void* x, float* out, int B, int R, int Lk, int Dk, void* ws,
                      size_t ws_bytes, void* stream);
/* The same over up to 4 SEGMENTS of 16-bit rows that share one softmax, on the matrix cores: segment s has its own row
 * buffer x[s] (image b's keys are rows b * rows_per_image + row0 + j, j < Lk, of ldx elements each, the first Dk of
 * which are read) and its own effective queries: qt, out [B, R, nseg, Dk] fp32,
 *   p = softmax over all (s, j) of qt[b, r, s] . x[s][b, j];   out[b, r, s] = sum_j p_(s, j) x[s][b, j].
 * This is the visual cross-attention of IQM on the LayerNorm'ed tap rows themselves (aaclip_tap_head_keep_rows; row0 = 1
 * skips the CLS row): with query_adapters[s] (reference model/adapter.py:205-208, a Linear without bias when relu is
 * off), torch.cat over the levels (:210-211), visual_feature_proj (:213-221) and the key / value Linear all linear in
 * the rows, each moves to the query side (qt[., s] = W_qa[s]^T P^T W_k[h]^T q_h / sqrt(d)) and behind the weighted
 * sums (sum_s W_qa[s] out[., s], then P, W_v) as [R, .] products through aaclip_gemm -- no per-row projection is left.
 * x_dtype AACLIP_F16 or AACLIP_BF16 (for the fp16 halves of split8 rows pass AACLIP_F16 and ldx = 2 Dk), R <= 16,
 * Dk 768 or 1024, ldx a multiple of 8. */
size_t aaclip_cross_rows_levels_workspace_bytes(int B, int nseg, int Lk, int Dk);
int aaclip_cross_rows_levels(int x_dtype, const float* qt, const void* const* x, int nseg, float* out, int B, int R,
                             int rows_per_image, int row0, int Lk, int Dk, long ldx, void* ws, size_t ws_bytes,
                             void* stream);
int aaclip_head_expand(int dtype, const float* q, void* qm, long rows, int H, int D, float scale, void* stream);
int aaclip_head_diag(const float* full, float* ctx, long rows, int H, int D, void* stream);
/* softmax(q k^T * scale) v per (image, head) for nq <= 4 queries over Lk <= 8192 keys: the core of
 * IQM_MultiHeadAttention.forward (reference model/iqm.py:108-139; masks are all-zero on this path, dropout is the
 * identity in eval).  q, out [B, nq, H*hd] fp32; k, v [B*Lk, H*hd] in kv_dtype; hd a multiple of 4, <= 128. */
int aaclip_small_attention(int kv_dtype, const float* q, const void* k, const void* v, float* out, int B, int nq, int Lk,
                           int H, int hd, float scale, void* stream);
/* out = LayerNorm(a + b) over the last dimension D (b may be NULL): IQM_SelfOutput / IQM_Output (reference
 * model/iqm.py:143-154,219-230, eps 1e-12), IQM.layernorm (:617) and iqm_layer_norm (model/adapter.py:265, eps 1e-5). */
int aaclip_residual_layernorm(const float* a, const float* b, const float* w, const float* bias, float* out, long rows,
                              int D, float eps, void* stream);
/* out = wa a + wb b + wc c (b, c may be NULL): the fixed 0.4 / 0.3 / 0.3 fusion of reference model/iqm.py:311-315 and
 * the query + positional embedding sum of model/adapter.py:199-203. */
int aaclip_combine3(const float* a, const float* b, const float* c, float wa, float wb, float wc, float* out, long n,
                    void* stream);
/* y = x W^T + bias for in_features K <= 4: text_feature_proj, which the reference creates as Linear(2, 768) because
 * the anchors reach the branch as [B, 768, 2] (model/adapter.py:229-246).  x fp32 [R, K], W fp32 [N, K], y [R, N] in
 * out_dtype. */
int aaclip_linear_smallk(int out_dtype, const float* x, const float* W, const float* bias, void* y, long R, int N, int K,
                         void* stream);
/* Rows 1..L-1 of every image of src [B, L, E] (dtype) -> rows row_off.. of every image of dst [B, rows_per_image, E]:
 * the torch.cat over dim 1 of the projected tap levels without their CLS row (model/adapter.py:171,206-211). */
int aaclip_drop_cls_rows(int dtype, const void* src, void* dst, int B, int L, int E, int rows_per_image, int row_off,
                         void* stream);
/* IQM anomaly map, reference test_last.py:102-147: per level sigmoid(cos(f, q_abnormal) - cos(f, q_normal)) on the
 * patch grid, bilinear (align_corners=False) to S x S, summed over NL <= 4 levels; out = w_base * base + w_iqm * that
 * (base = the text anomaly map, 0.6 / 0.4 in test_last.py:67-68,141-147; base NULL = IQM map alone).
 * seg[l] [B, g*g, E] fp32, queries [B, 2, E] fp32 (row 0 normal, row 1 abnormal), ws >= NL*B*g*g floats. */
int aaclip_iqm_map(const float* const* seg, int NL, const float* queries, const float* base, float* out, int B, int g,
                   int E, int S, float w_base, float w_iqm, void* ws, size_t ws_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AACLIP_H */
