// C ABI of libaaclip_hip.so (declared in include/aaclip.h): argument checks,
// workspace carving and kernel sequencing.  No allocation, no synchronisation:
// every entry point only enqueues kernels on the caller's stream.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "kernels.h"

using namespace aaclip;

static thread_local char g_err[512] = "";
static int g_ln_fold = 1;   // aaclip_set_gemm_variant bit 17 turns the LayerNorm folding off (A/B measurements)

static int fail(int code, const char* msg) {
  snprintf(g_err, sizeof(g_err), "%s", msg);
  return code;
}
#define REQUIRE(cond, msg) \
  do {                     \
    if (!(cond)) return fail(-1, msg); \
  } while (0)

static int finish(const char* what) {
  if (const char* le = take_launch_error()) {   // a launcher was asked for a kernel it does not have: nothing ran for it
    snprintf(g_err, sizeof(g_err), "%s: %s", what, le);
    (void)hipGetLastError();
    return -4;
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
    return -2;
  }
  return 0;
}

// ---- optional in-stream timing of tagged kernels (bench.py roofline leg) ----
// Tags: 0 ln, 1 qkv gemm, 2 attention, 3 out_proj gemm, 4 c_fc gemm, 5 c_proj gemm, 6 adapter
struct ProfSlot { hipEvent_t a, b; int tag; };
static ProfSlot* g_prof = nullptr;
static int g_prof_cap = 0, g_prof_n = 0;
static unsigned g_prof_mask = 0;
struct ProfScope {
  int idx;
  hipStream_t s;
  ProfScope(int tag, hipStream_t st) : idx(-1), s(st) {
    if (g_prof_mask & (1u << tag)) {
      if (g_prof_n < g_prof_cap) {
        idx = g_prof_n++;
        g_prof[idx].tag = tag;
        (void)hipEventRecord(g_prof[idx].a, s);
      }
    }
  }
  ~ProfScope() {
    if (idx >= 0) (void)hipEventRecord(g_prof[idx].b, s);
  }
};

// bytes per element of an activation / weight row in the compute dtype (a split fp16 value is a hi and a lo half)
static inline size_t esize(int dtype) { return (dtype == AACLIP_F32 || dtype == AACLIP_F16X2) ? 4 : 2; }
static inline int split_w(int dtype) { return dtype == AACLIP_F16X2 ? 2 : 1; }   // row stride factor of split rows
static inline size_t up256(size_t v) { return (v + 255) & ~(size_t)255; }
// aux region of the workspace: 16-bit copy of the residual rows, per-row partial sums [rows][D/64][2], (a, b) pairs
static inline size_t aux_bytes(size_t es, long rows, int D) {
  return up256((size_t)rows * D * es) + up256((size_t)rows * (D / 64 + 1) * 2 * 4) + up256((size_t)rows * 2 * 4);
}
static inline bool dtype_ok(int d) { return d == AACLIP_F32 || d == AACLIP_F16 || d == AACLIP_BF16 || d == AACLIP_F16X2; }
static inline bool plain_dtype_ok(int d) { return d == AACLIP_F32 || d == AACLIP_F16 || d == AACLIP_BF16; }

extern "C" {

int aaclip_profile_begin(unsigned tag_mask, int capacity) {
  if (capacity > g_prof_cap) {
    ProfSlot* n = (ProfSlot*)realloc(g_prof, sizeof(ProfSlot) * capacity);
    if (!n) return fail(-3, "profile: out of memory");
    g_prof = n;
    for (int i = g_prof_cap; i < capacity; ++i) {
      if (hipEventCreate(&g_prof[i].a) != hipSuccess || hipEventCreate(&g_prof[i].b) != hipSuccess)
        return fail(-2, "profile: hipEventCreate failed");
    }
    g_prof_cap = capacity;
  }
  g_prof_n = 0;
  g_prof_mask = tag_mask;
  return 0;
}

int aaclip_profile_end(float* ms, int* tags, int max_n) {
  g_prof_mask = 0;
  int n = g_prof_n < max_n ? g_prof_n : max_n;
  for (int i = 0; i < n; ++i) {
    if (hipEventSynchronize(g_prof[i].b) != hipSuccess) return fail(-2, "profile: event sync failed");
    float t = 0.f;
    if (hipEventElapsedTime(&t, g_prof[i].a, g_prof[i].b) != hipSuccess) return fail(-2, "profile: elapsed failed");
    ms[i] = t;
    tags[i] = g_prof[i].tag;
  }
  g_prof_n = 0;
  return n;
}

int aaclip_set_gemm_variant(int v) {
  REQUIRE(v >= 0 && (v >> 18) == 0, "set_gemm_variant: unknown bits");
  // Validate everything before changing anything: a rejected call leaves the selection as it was.
  const int gv = v & 0xFF, av = (v >> 8) & 0xFF;
#ifdef AACLIP_MEASURE
  REQUIRE((gv <= 60 || gv == 70 || (gv >= 80 && gv <= 82)) && (av <= 3 || av == 6), "set_gemm_variant: no such kernel variant");
#else
  REQUIRE((gv <= 1 || (gv >= 80 && gv <= 82)) && av <= 1,
          "set_gemm_variant: libaaclip_hip.so only has GEMM variants 0/1/80/81/82 and attention variants 0/1; A/B variants, "
          "timing ablations (wrong results) and stamp builds live in libaaclip_hip_measure.so (make measure)");
#endif
  set_gemm_variant(gv);
  set_attn_variant(av);   // bits 8..15: attention kernel selection
  g_ln_fold = ((v >> 17) & 1) ? 0 : 1;
  set_tail_peel((v >> 16) & 1);  // bit 16: peel the partial last round to the 128-tile kernel (measured: -1.6 %, off by default)
  return 0;
}

int aaclip_debug_gemm_stamps(double* out3, int nwaves) {
#ifdef AACLIP_MEASURE
  REQUIRE(out3, "debug_gemm_stamps: null pointer");
  if (nwaves == -2) {   // walking kernel's compact epilogue (-DX_WALK_STAMP): 8 sums, reset on read
    read_gemm_estamps(out3);
    return 0;
  }
  if (nwaves < 0) {   // persistent kernel: 8 values (cycles per tile of 7 segments, tile count)
    read_gemm_zstamps(out3);
    return 0;
  }
  read_gemm_stamps(out3, nwaves);
  return 0;
#else
  (void)out3; (void)nwaves;
  return fail(-1, "debug_gemm_stamps: stamp kernels are part of libaaclip_hip_measure.so only (make measure)");
#endif
}

#ifdef AACLIP_MEASURE
// measurement library only (not declared in include/aaclip.h): how often the long-sequence attention kernel took each
// of its tile paths since the last reset -- out[0] tile 0, out[1] fast passes, out[2] exact redos
extern "C" int aaclip_measure_attn_passes(unsigned long long* out4, int reset) {
  read_attn_passes(out4, reset);
  return 0;
}
extern "C" int aaclip_measure_attn_stamps(unsigned long long* out9, int reset) {
  read_attn_stamps(out9, reset);
  return 0;
}
#endif

int aaclip_is_measurement_build(void) {
#ifdef AACLIP_MEASURE
  return 1;
#else
  return 0;
#endif
}

int aaclip_version(void) { return AACLIP_ABI_VERSION; }
const char* aaclip_last_error(void) { return g_err; }

// workspace layout: [narrow: rows*max(D,640)*es] [big] [rows floats] [aux: LayerNorm folding] + slack
struct WsLayout { size_t big_off, rowf_off, aux_off, total; };
static WsLayout ws_layout(int dtype, long rows, int D, int F, int E) {
  const size_t es = esize(dtype);
  size_t wide = (size_t)rows * (size_t)(3 * D > F ? 3 * D : F) * es;
  size_t f32d = (size_t)rows * D * 4;
  size_t f32e = (size_t)rows * (E > 0 ? E : 1) * 4;
  size_t big = wide > f32d ? wide : f32d;
  if (f32e > big) big = f32e;
  size_t narrow = (size_t)rows * (D > 640 ? D : 640) * es;
  WsLayout l;
  l.big_off = up256(narrow);
  l.rowf_off = l.big_off + up256(big);
  l.aux_off = l.rowf_off + up256((size_t)rows * 4);
  l.total = l.aux_off + aux_bytes(es, rows, D) + 4096;
  return l;
}

size_t aaclip_workspace_bytes(int dtype, long rows, int D, int F, int E) { return ws_layout(dtype, rows, D, F, E).total; }

int aaclip_layernorm(const float* x, const float* w, const float* b, void* out, int out_dtype, long rows, int D,
                     float eps, void* stream) {
  REQUIRE(x && w && b && out, "layernorm: null pointer");
  REQUIRE(dtype_ok(out_dtype), "layernorm: bad dtype");
  REQUIRE(rows > 0, "layernorm: rows must be positive");
  const char* m = row_width_check(D);
  if (m) return fail(-1, m);
  launch_layernorm(out_dtype, x, w, b, out, rows, D, eps, (hipStream_t)stream);
  return finish("layernorm");
}

int aaclip_gemm(int dtype, int epi, const void* A, long lda, const void* W, const float* bias, void* out, long ldc,
                int M, int N, int K, int act, int scale_cols, float scale, void* stream) {
  REQUIRE(dtype_ok(dtype), "gemm: bad dtype");
  REQUIRE(epi >= AACLIP_EPI_BIAS && epi <= AACLIP_EPI_ACT_F32, "gemm: bad epilogue");
  GemmParams p;
  memset(&p, 0, sizeof(p));
  p.A = A; p.lda = lda; p.W = W; p.M = M; p.N = N; p.K = K; p.bias = bias; p.out = out; p.ldc = ldc;
  p.scale_cols = scale_cols; p.scale = scale; p.act = act;
  const char* m = gemm_check(dtype, epi, p);
  if (m) return fail(-1, m);
  launch_gemm(dtype, epi, p, (hipStream_t)stream);
  return finish("gemm");
}

int aaclip_attention(int dtype, const void* qkv, void* ctx, int B, int L, int H, int causal, void* stream) {
  REQUIRE(dtype_ok(dtype), "attention: bad dtype");
  REQUIRE(qkv && ctx, "attention: null pointer");
  REQUIRE(B > 0 && L > 0 && H > 0, "attention: empty problem");
  REQUIRE(B <= 65535 && H <= 65535, "attention: grid limit");
  REQUIRE((long)L * 3 * 64 * H * 4 < (1L << 31), "attention: L * 3 * 64 * H * 4 must stay below 2^31 (32-bit row offsets)");
  REQUIRE((long)((L + 255) / 256) * H * B < (1L << 30), "attention: too many workgroups");
  launch_attention(dtype, qkv, ctx, B, L, H, causal, 0, (hipStream_t)stream);
  return finish("attention");
}

int aaclip_attention_log2q(int dtype, const void* qkv, void* ctx, int B, int L, int H, int causal, void* stream) {
  REQUIRE(dtype == AACLIP_F16 || dtype == AACLIP_BF16 || dtype == AACLIP_F16X2, "attention_log2q: 16-bit dtypes only");
  REQUIRE(qkv && ctx, "attention: null pointer");
  REQUIRE(B > 0 && L > 0 && H > 0, "attention: empty problem");
  REQUIRE(B <= 65535 && H <= 65535, "attention: grid limit");
  REQUIRE((long)L * 3 * 64 * H * 4 < (1L << 31), "attention: L * 3 * 64 * H * 4 must stay below 2^31 (32-bit row offsets)");
  REQUIRE((long)((L + 255) / 256) * H * B < (1L << 30), "attention: too many workgroups");
  launch_attention(dtype, qkv, ctx, B, L, H, causal, 1, (hipStream_t)stream);
  return finish("attention");
}

int aaclip_adapter_mix(float* x, const float* a, long rows, int D, float weight, void* stream) {
  REQUIRE(x && a && rows > 0, "adapter_mix: bad arguments");
  const char* m = row_width_check(D);
  if (m) return fail(-1, m);
  launch_adapter_mix(x, a, rows, D, weight, (hipStream_t)stream);
  return finish("adapter_mix");
}

int aaclip_small_attention(int kv_dtype, const float* q, const void* k, const void* v, float* out, int B, int nq, int Lk,
                           int H, int hd, float scale, void* stream) {
  REQUIRE(plain_dtype_ok(kv_dtype), "small_attention: bad dtype (fp32, fp16 or bf16)");
  REQUIRE(q && k && v && out, "small_attention: null pointer");
  REQUIRE(B > 0 && B <= 65535 && H <= 65535, "small_attention: bad batch / heads");
  const char* m = small_attention_check(nq, Lk, H, hd);
  if (m) return fail(-1, m);
  launch_small_attention(kv_dtype, q, k, v, out, B, nq, Lk, H, hd, scale, (hipStream_t)stream);
  return finish("small_attention");
}

size_t aaclip_cross_rows_workspace_bytes(int B, int R, int Lk, int Dk) { return cross_rows_ws_bytes(B, R, Lk, Dk); }

int aaclip_cross_rows(int x_dtype, const float* qt, const void* x, float* out, int B, int R, int Lk, int Dk, void* ws,
                      size_t ws_bytes, void* stream) {
  REQUIRE(plain_dtype_ok(x_dtype), "cross_rows: bad dtype (fp32, fp16 or bf16)");
  REQUIRE(qt && x && out && ws, "cross_rows: null pointer");
  REQUIRE(B > 0 && B <= 65535, "cross_rows: bad batch");
  const char* m = cross_rows_check(R, Lk, Dk);
  if (m) return fail(-1, m);
  REQUIRE(ws_bytes >= cross_rows_ws_bytes(B, R, Lk, Dk), "cross_rows: workspace too small");
  launch_cross_rows(x_dtype, qt, x, out, ws, B, R, Lk, Dk, (hipStream_t)stream);
  return finish("cross_rows");
}

size_t aaclip_cross_rows_levels_workspace_bytes(int B, int nseg, int Lk, int Dk) {
  if (B < 1 || nseg < 1 || Lk < 1 || Dk < 1) return 0;
  return cross_rows_levels_ws_bytes(B, nseg, Lk, Dk);
}

int aaclip_cross_rows_levels(int x_dtype, const float* qt, const void* const* x, int nseg, float* out, int B, int R,
                             int rows_per_image, int row0, int Lk, int Dk, long ldx, void* ws, size_t ws_bytes,
                             void* stream) {
  REQUIRE(qt && x && out && ws, "cross_rows_levels: null pointer");
  REQUIRE(B > 0 && B <= 65535, "cross_rows_levels: bad batch");
  const char* m = cross_rows_levels_check(x_dtype, R, nseg, Lk, Dk, ldx);
  if (m) return fail(-1, m);
  REQUIRE(row0 >= 0 && rows_per_image >= row0 + Lk, "cross_rows_levels: the keys [row0, row0 + Lk) must lie inside an image's rows");
  REQUIRE((long)rows_per_image * ldx * 2 < (1L << 31), "cross_rows_levels: an image's rows must span < 2 GiB");
  REQUIRE(ws_bytes >= cross_rows_levels_ws_bytes(B, nseg, Lk, Dk), "cross_rows_levels: workspace too small");
  for (int i = 0; i < nseg; ++i) REQUIRE(x[i] && ((uintptr_t)x[i] & 15) == 0, "cross_rows_levels: segment pointers must be 16-byte aligned");
  launch_cross_rows_levels(x_dtype, qt, x, nseg, out, ws, B, R, rows_per_image, row0, Lk, Dk, ldx, (hipStream_t)stream);
  return finish("cross_rows_levels");
}

int aaclip_head_expand(int dtype, const float* q, void* qm, long rows, int H, int D, float scale, void* stream) {
  REQUIRE(plain_dtype_ok(dtype), "head_expand: bad dtype (fp32, fp16 or bf16)");
  REQUIRE(q && qm && rows > 0 && H > 0 && D > 0 && D % H == 0, "head_expand: bad arguments");
  REQUIRE(rows * H < (1L << 31), "head_expand: too many rows");
  launch_head_expand(dtype, q, qm, rows, H, D, scale, (hipStream_t)stream);
  return finish("head_expand");
}

int aaclip_head_diag(const float* full, float* ctx, long rows, int H, int D, void* stream) {
  REQUIRE(full && ctx && rows > 0 && rows < (1L << 31) && H > 0 && D > 0 && D % H == 0, "head_diag: bad arguments");
  launch_head_diag(full, ctx, rows, H, D, (hipStream_t)stream);
  return finish("head_diag");
}

int aaclip_residual_layernorm(const float* a, const float* b, const float* w, const float* bias, float* out, long rows,
                              int D, float eps, void* stream) {
  REQUIRE(a && w && bias && out, "residual_layernorm: null pointer");
  REQUIRE(rows > 0 && D > 0 && D % 64 == 0 && D <= 4096, "residual_layernorm: D must be a multiple of 64, <= 4096");
  launch_residual_layernorm(a, b, w, bias, out, rows, D, eps, (hipStream_t)stream);
  return finish("residual_layernorm");
}

int aaclip_combine3(const float* a, const float* b, const float* c, float wa, float wb, float wc, float* out, long n,
                    void* stream) {
  REQUIRE(a && out && n > 0, "combine3: bad arguments");
  launch_combine3(a, b, c, wa, wb, wc, out, n, (hipStream_t)stream);
  return finish("combine3");
}

int aaclip_linear_smallk(int out_dtype, const float* x, const float* W, const float* bias, void* y, long R, int N, int K,
                         void* stream) {
  REQUIRE(plain_dtype_ok(out_dtype), "linear_smallk: bad dtype (fp32, fp16 or bf16)");
  REQUIRE(x && W && y, "linear_smallk: null pointer");
  REQUIRE(R > 0 && N > 0 && K >= 1 && K <= 4, "linear_smallk: in_features must be 1..4");
  launch_linear_smallk(out_dtype, x, W, bias, y, R, N, K, (hipStream_t)stream);
  return finish("linear_smallk");
}

int aaclip_drop_cls_rows(int dtype, const void* src, void* dst, int B, int L, int E, int rows_per_image, int row_off,
                         void* stream) {
  REQUIRE(plain_dtype_ok(dtype), "drop_cls_rows: bad dtype (fp32, fp16 or bf16)");
  REQUIRE(src && dst, "drop_cls_rows: null pointer");
  REQUIRE(B > 0 && L > 1 && E > 0 && E % 8 == 0, "drop_cls_rows: E must be a multiple of 8");
  REQUIRE(row_off >= 0 && row_off + (L - 1) <= rows_per_image, "drop_cls_rows: rows do not fit the destination");
  launch_drop_cls_rows(dtype, src, dst, B, L, E, rows_per_image, row_off, (hipStream_t)stream);
  return finish("drop_cls_rows");
}

int aaclip_iqm_map(const float* const* seg, int NL, const float* queries, const float* base, float* out, int B, int g,
                   int E, int S, float w_base, float w_iqm, void* ws, size_t ws_bytes, void* stream) {
  REQUIRE(seg && queries && out && ws, "iqm_map: null pointer");
  REQUIRE(NL >= 1 && NL <= 4, "iqm_map: 1..4 levels");
  REQUIRE(B > 0 && B <= 65535 && g >= 1 && g <= 40 && S >= 1, "iqm_map: bad shape (grid <= 40)");
  const char* m = row_width_check(E);
  if (m) return fail(-1, m);
  const int P = g * g;
  REQUIRE(ws_bytes >= (size_t)NL * B * P * 4, "iqm_map: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  float* grids = (float*)ws;
  for (int l = 0; l < NL; ++l) {
    REQUIRE(seg[l], "iqm_map: null level pointer");
    launch_iqm_scores(seg[l], queries, grids + (size_t)l * B * P, B, P, E, s);
  }
  launch_iqm_upsample(grids, base, out, B, g, S, NL, w_base, w_iqm, s);
  return finish("iqm_map");
}

int aaclip_patch_embed(const float* img, const void* conv_w, const float* cls, const float* pos,
                       const float* ln_pre_w, const float* ln_pre_b, float* x, int B, int H, int W, int ps, int D,
                       int dtype, void* ws, size_t ws_bytes, void* stream) {
  REQUIRE(dtype_ok(dtype), "patch_embed: bad dtype");
  REQUIRE(img && conv_w && cls && pos && ln_pre_w && ln_pre_b && x && ws, "patch_embed: null pointer");
  REQUIRE(B > 0 && ps > 0 && H >= ps && W >= ps, "patch_embed: bad image shape");
  const char* m = row_width_check(D);
  if (m) return fail(-1, m);
  const int g = H / ps, gw = W / ps, P = g * gw, L = P + 1;
  const int K = 3 * ps * ps, Kpad = (K + 63) / 64 * 64;
  REQUIRE(Kpad <= 640, "patch_embed: 3*ps*ps must be <= 640");
  REQUIRE(ws_bytes >= (size_t)B * P * Kpad * esize(dtype), "patch_embed: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  launch_im2col(dtype, img, ws, B, 3, H, W, ps, Kpad, s);
  GemmParams p;
  memset(&p, 0, sizeof(p));
  p.A = ws; p.lda = split_w(dtype) * Kpad; p.W = conv_w; p.M = B * P; p.N = D; p.K = Kpad; p.out = x; p.ldc = D;
  p.pos = pos; p.P = P; p.L = L;
  m = gemm_check(dtype, EPI_PATCH, p);
  if (m) return fail(-1, m);
  launch_gemm(dtype, EPI_PATCH, p, s);
  launch_cls_rows(x, cls, pos, B, L, D, s);
  launch_layernorm(AACLIP_F32, x, ln_pre_w, ln_pre_b, x, (long)B * L, D, 1e-5f, s);
  return finish("patch_embed");
}

// One block.  aux_in: the aux region of the workspace holds the current rows of x in the compute dtype and their
// (rstd, -mean*rstd) pairs, written by the previous block of the same aaclip_blocks call -- then ln_1 is folded
// into the QKV product.  want_out: produce them for the next block (from the c_proj epilogue, or from the adapter
// mix when the block has an adapter).  *aux_out tells the caller whether they were produced.
// x_in != x: the block READS the stream from x_in (left untouched) and continues it in x -- the first residual
// update (out_proj) takes its residual from x_in and writes x.
static int block_impl(const float* x_in, float* x, const aaclip_block_weights* w, float mix, int B, int L, int D, int H,
                      int F, int attn_mode, int dtype, void* ws, size_t ws_bytes, hipStream_t s, bool aux_in,
                      bool want_out, bool* aux_out) {
  *aux_out = false;
  REQUIRE(w->ln1_w && w->ln1_b && w->qkv_w && w->qkv_b && w->out_w && w->out_b && w->ln2_w && w->ln2_b && w->fc_w &&
              w->fc_b && w->proj_w && w->proj_b,
          "block: null weight pointer");
  const long rows = (long)B * L;
  const size_t es = esize(dtype);
  char* narrow = (char*)ws;
  char* big = narrow + up256((size_t)rows * (D > 640 ? D : 640) * es);
  char* aux = (char*)ws + ws_layout(dtype, rows, D, F, 0).aux_off;
  char* x16 = aux;
  float* partials = (float*)(aux + up256((size_t)rows * D * es));
  float* rowab = (float*)((char*)partials + up256((size_t)rows * (D / 64 + 1) * 2 * 4));
  const int M = (int)rows;
  const bool folding = g_ln_fold && (dtype == AACLIP_F16 || dtype == AACLIP_BF16);
  const int sw = split_w(dtype);                    // split fp16 rows are [hi | lo]: twice the row stride
  const unsigned ex = dtype == AACLIP_F16X2 ? w->exact16 : 0u;   // weights whose lo half is all zero

  GemmParams p;
  // ---- x += out_proj(attn(ln_1 x))
  memset(&p, 0, sizeof(p));
  // 16-bit path: fold log2(e) into the q scale (one rounding) so the attention kernel works in log2 units
  const int log2q = dtype != AACLIP_F32;
  const float qscale = log2q ? 0.125f * 1.4426950408889634f : 0.125f;
  const char* ctx = narrow;
  if (attn_mode == AACLIP_ATTN_VV_BATCH) {
    { ProfScope ps(0, s); launch_layernorm(dtype, x_in, w->ln1_w, w->ln1_b, narrow, rows, D, 1e-5f, s, !(ex & AACLIP_EXACT16_QKV)); }
    // only the value third of in_proj is needed; v lives behind the packed q|k|v buffer inside `big`
    char* vbuf = big + (size_t)rows * 3 * D * es;
    p.A = narrow; p.lda = sw * D; p.M = M; p.N = D; p.K = D;
    p.w_exact16 = ex & AACLIP_EXACT16_QKV;
    // rows 2D.. of the packed weight; a split8 weight row holds 4 (3: exact in fp16) bytes per element
    p.W = (const char*)w->qkv_w + (size_t)2 * D * D * (dtype == AACLIP_F16X2 ? (p.w_exact16 ? 3 : 4) : es);
    p.bias = w->qkv_b + 2 * D; p.out = vbuf; p.ldc = sw * D;
    { ProfScope ps(1, s); launch_gemm(dtype, EPI_BIAS, p, s); }
    {
      ProfScope ps(2, s);
      launch_vv_spread(dtype, vbuf, big, B, L, D, qscale, s);
      launch_attention(dtype, big, narrow, /*batches=*/L, /*sequence=*/B, H, 0, log2q, s, !(ex & AACLIP_EXACT16_OUT));
      launch_vv_regroup(dtype, narrow, vbuf, B, L, D, s);
    }
    ctx = vbuf;
  } else {
    p.A = narrow; p.lda = sw * D; p.W = w->qkv_w; p.M = M; p.N = 3 * D; p.K = D; p.bias = w->qkv_b; p.out = big;
    p.ldc = sw * 3 * D; p.scale_cols = D; p.scale = qscale; p.w_exact16 = ex & AACLIP_EXACT16_QKV;
    if (aux_in && folding && w->qkv_w_fold && w->qkv_fold_s && w->qkv_fold_b && gemm_routes_to_256t(dtype, p)) {
      // ln_1 folded into the QKV product (include/aaclip.h, aaclip_block_weights)
      p.A = x16; p.W = w->qkv_w_fold; p.bias = w->qkv_fold_b; p.row_ab = rowab; p.col_s = w->qkv_fold_s;
    } else {
      ProfScope ps(0, s);
      launch_layernorm(dtype, x_in, w->ln1_w, w->ln1_b, narrow, rows, D, 1e-5f, s, !(ex & AACLIP_EXACT16_QKV));
    }
    // split fp16, long rows, 256-tile QKV kernel: q and k leave the epilogue as fp16 + e4m3 records and the attention
    // kernel runs its two correction products on the e4m3 MFMA (attention.hip, QK8)
    bool qk8 = false;
    if (dtype == AACLIP_F16X2 && log2q && attention_qk8_applicable(L, attn_mode == AACLIP_ATTN_CAUSAL)) {
      p.ldc = 5 * D; p.out_qk8 = 2 * D;
      qk8 = gemm_split_routes_to_256t(p);
      if (!qk8) { p.ldc = sw * 3 * D; p.out_qk8 = 0; }
    }
    { ProfScope ps(1, s); launch_gemm(dtype, EPI_BIAS, p, s); }
    { ProfScope ps(2, s); launch_attention(dtype, big, narrow, B, L, H, attn_mode == AACLIP_ATTN_CAUSAL, log2q, s, !(ex & AACLIP_EXACT16_OUT), qk8); }
  }
  memset(&p, 0, sizeof(p));
  p.A = ctx; p.lda = sw * D; p.W = w->out_w; p.M = M; p.N = D; p.K = D; p.bias = w->out_b; p.out = x; p.ldc = D;
  p.w_exact16 = ex & AACLIP_EXACT16_OUT;
  if (x_in != x) p.resid = x_in;
  // ln_2 folded into c_fc: only where both products run on the kernels whose epilogue implements it;
  // everywhere else the ln_2 pass runs as before
  GemmParams fc;
  memset(&fc, 0, sizeof(fc));
  fc.lda = sw * D; fc.M = M; fc.N = F; fc.K = D; fc.out = big; fc.ldc = sw * F;
  fc.A = narrow; fc.W = w->fc_w; fc.bias = w->fc_b; fc.w_exact16 = ex & AACLIP_EXACT16_FC;
  fc.out_no_hi8 = (ex & AACLIP_EXACT16_PROJ) ? 1 : 0;   // c_proj is the only reader of the GELU rows
  const bool fold2 = folding && w->fc_w_fold && w->fc_fold_s && w->fc_fold_b && gemm_routes_to_256t(dtype, p) &&
                     gemm_routes_to_256t(dtype, fc);
  if (fold2) {
    p.out16 = x16;
    p.stats_out = partials;
  }
  { ProfScope ps(3, s); launch_gemm(dtype, EPI_BIAS_RESID, p, s); }
  // ---- x += c_proj(gelu(c_fc(ln_2 x)))
  if (fold2) {
    ProfScope ps(0, s);
    launch_ln_stats_finalize(partials, rowab, rows, D / 64, D, 1e-5f, s);
    fc.A = x16; fc.W = w->fc_w_fold; fc.bias = w->fc_fold_b; fc.row_ab = rowab; fc.col_s = w->fc_fold_s;
  } else {
    ProfScope ps(0, s);
    launch_layernorm(dtype, x, w->ln2_w, w->ln2_b, narrow, rows, D, 1e-5f, s, !(ex & AACLIP_EXACT16_FC));
  }
  { ProfScope ps(4, s); launch_gemm(dtype, EPI_BIAS_GELU, fc, s); }
  memset(&p, 0, sizeof(p));
  p.A = big; p.lda = sw * F; p.W = w->proj_w; p.M = M; p.N = D; p.K = F; p.bias = w->proj_b; p.out = x; p.ldc = D;
  p.w_exact16 = ex & AACLIP_EXACT16_PROJ;
  // the c_proj epilogue can also emit the new rows in 16 bits: input of the adapter product, or (with their
  // row sums) of the next block's folded ln_1
  const bool emit = folding && gemm_routes_to_256t(dtype, p) && (w->adapter_w || want_out);
  if (emit) {
    p.out16 = x16;
    p.stats_out = partials;
  }
  { ProfScope ps(5, s); launch_gemm(dtype, EPI_BIAS_RESID, p, s); }
  if (emit && !w->adapter_w) {
    ProfScope ps(0, s);
    launch_ln_stats_finalize(partials, rowab, rows, D / 64, D, 1e-5f, s);
    *aux_out = true;
  }
  // ---- residual adapter
  if (w->adapter_w) {
    ProfScope ps(6, s);
    const void* a_in = x;
    if (emit) {
      a_in = x16;
    } else if (dtype == AACLIP_F16X2) {
      launch_split_rows(x, narrow, rows, D, s, !(ex & AACLIP_EXACT16_ADAPTER));
      a_in = narrow;
    } else if (dtype != AACLIP_F32) {
      launch_cast_rows(dtype, x, narrow, rows * D, s);
      a_in = narrow;
    }
    memset(&p, 0, sizeof(p));
    p.A = a_in; p.lda = sw * D; p.W = w->adapter_w; p.M = M; p.N = D; p.K = D; p.out = big; p.ldc = D; p.act = 1;
    p.w_exact16 = ex & AACLIP_EXACT16_ADAPTER;
    launch_gemm(dtype, EPI_ACT_F32, p, s);
    if (folding && want_out) {
      launch_adapter_mix_fold(dtype, x, (const float*)big, rows, D, mix, x16, rowab, s);
      *aux_out = true;
    } else {
      launch_adapter_mix(x, (const float*)big, rows, D, mix, s);
    }
  }
  return 0;
}

int aaclip_blocks(float* x, const aaclip_block_weights* w, int n_blocks, float mix, int B, int L, int D, int H, int F,
                  int attn_mode, int dtype, void* ws, size_t ws_bytes, void* stream) {
  return aaclip_blocks_to(x, x, w, n_blocks, mix, B, L, D, H, F, attn_mode, dtype, ws, ws_bytes, stream);
}

// x_out[i] (x_of(i)) = the buffer holding the stream after block i; block i reads the stream from x_in (i == 0) or
// from the buffer block i-1 wrote.  x_out == nullptr: every block writes x_all.
static int blocks_core(const float* x_in, float* const* x_out, float* x_all, const aaclip_block_weights* w, int n_blocks,
                       float mix, int B, int L, int D, int H, int F, int attn_mode, int dtype, void* ws, size_t ws_bytes,
                       void* stream) {
  float* x = x_out ? x_out[0] : x_all;
  REQUIRE(x_in && w, "block: null pointer");
  REQUIRE(n_blocks >= 1, "block: n_blocks must be positive");
  // a caller built against another header revision (no size field, fewer pointer fields) is refused here, before
  // anything reads its struct as if it were ours
  for (int i = 0; i < n_blocks; ++i)
    REQUIRE(w[i].struct_bytes == sizeof(aaclip_block_weights),
            "block: aaclip_block_weights.struct_bytes does not match this library (binding generated from another "
            "include/aaclip.h revision?)");
  REQUIRE(dtype_ok(dtype), "block: bad dtype");
  REQUIRE(attn_mode >= AACLIP_ATTN_FULL && attn_mode <= AACLIP_ATTN_VV_BATCH, "block: attn_mode must be 0, 1 or 2");
  REQUIRE(attn_mode != AACLIP_ATTN_VV_BATCH || F >= 4 * D, "block: V-V attention needs F >= 4*D workspace columns");
  REQUIRE(x && w && ws, "block: null pointer");
  REQUIRE(n_blocks >= 1, "block: n_blocks must be positive");
  REQUIRE(B > 0 && L > 0, "block: empty batch");
  REQUIRE(D == 64 * H, "block: D must equal 64*H (head dim 64)");
  REQUIRE((long)L * 3 * D * 4 < (1L << 31) && (long)B * 3 * D * 4 < (1L << 31),
          "block: sequence too long for the attention kernel's 32-bit row offsets");
  REQUIRE(F % 128 == 0 && F % 64 == 0, "block: F must be a multiple of 128");
  const char* m = row_width_check(D);
  if (m) return fail(-1, m);
  REQUIRE(D % 128 == 0, "block: D must be a multiple of 128");
  const long rows = (long)B * L;
  REQUIRE(rows < (1L << 31) / 4, "block: too many rows");
  REQUIRE(ws_bytes >= aaclip_workspace_bytes(dtype, rows, D, F, 0), "block: workspace too small");
  if (x_out)
    for (int i = 0; i < n_blocks; ++i) REQUIRE(x_out[i], "block: null output buffer");
  bool aux = false;
  const float* src = x_in;
  for (int i = 0; i < n_blocks; ++i) {
    bool produced = false;
    float* dst = x_out ? x_out[i] : x_all;
    int rc = block_impl(src, dst, w + i, mix, B, L, D, H, F, attn_mode, dtype, ws, ws_bytes, (hipStream_t)stream, aux,
                        i + 1 < n_blocks, &produced);
    if (rc) return rc;
    aux = produced;
    src = dst;
  }
  return finish("block");
}

int aaclip_blocks_to(const float* x_in, float* x, const aaclip_block_weights* w, int n_blocks, float mix, int B, int L,
                     int D, int H, int F, int attn_mode, int dtype, void* ws, size_t ws_bytes, void* stream) {
  REQUIRE(x, "block: null pointer");
  return blocks_core(x_in, nullptr, x, w, n_blocks, mix, B, L, D, H, F, attn_mode, dtype, ws, ws_bytes, stream);
}

int aaclip_blocks_taps(const float* x_in, float* const* x_out, const aaclip_block_weights* w, int n_blocks, float mix,
                       int B, int L, int D, int H, int F, int attn_mode, int dtype, void* ws, size_t ws_bytes,
                       void* stream) {
  REQUIRE(x_out && n_blocks >= 1, "block: null output list");
  return blocks_core(x_in, x_out, nullptr, w, n_blocks, mix, B, L, D, H, F, attn_mode, dtype, ws, ws_bytes, stream);
}

int aaclip_block(float* x, const aaclip_block_weights* w, float mix, int B, int L, int D, int H, int F, int attn_mode,
                 int dtype, void* ws, size_t ws_bytes, void* stream) {
  return aaclip_blocks(x, w, 1, mix, B, L, D, H, F, attn_mode, dtype, ws, ws_bytes, stream);
}

static int head_common(const float* x, const float* ln_w, const float* ln_b, int B, int L, int D, int E, int dtype,
                       void* ws, size_t ws_bytes, char** narrow, char** big, char** rowf) {
  REQUIRE(dtype_ok(dtype), "head: bad dtype");
  REQUIRE(x && ln_w && ln_b && ws, "head: null pointer");
  REQUIRE(B > 0 && L > 1, "head: bad shape");
  const char* m = row_width_check(D);
  if (m) return fail(-1, m);
  m = row_width_check(E);
  if (m) return fail(-1, m);
  REQUIRE(E % 128 == 0, "head: E must be a multiple of 128");
  const long rows = (long)B * L;
  REQUIRE(ws_bytes >= aaclip_workspace_bytes(dtype, rows, D, 0, E), "head: workspace too small");
  *narrow = (char*)ws;
  *big = *narrow + up256((size_t)rows * (D > 640 ? D : 640) * esize(dtype));
  size_t wide = (size_t)rows * (size_t)(3 * D) * esize(dtype);
  size_t f32d = (size_t)rows * D * 4, f32e = (size_t)rows * E * 4;
  size_t bigsz = wide > f32d ? wide : f32d;
  if (f32e > bigsz) bigsz = f32e;
  *rowf = *big + up256(bigsz);
  return 0;
}

static int tap_head_impl(const float* x, const float* ln_post_w, const float* ln_post_b, const void* proj_w, int act,
                         float* seg_out, const void* det_w, float* det_out, void* ln_rows_out, int B, int L, int D, int E,
                         int dtype, void* ws, size_t ws_bytes, void* stream) {
  char *narrow, *big, *rowf;
  int rc = head_common(x, ln_post_w, ln_post_b, B, L, D, E, dtype, ws, ws_bytes, &narrow, &big, &rowf);
  if (rc) return rc;
  REQUIRE(proj_w && seg_out, "tap_head: null pointer");
  REQUIRE(!det_w || det_out, "tap_head: det_out missing");
  hipStream_t s = (hipStream_t)stream;
  const long rows = (long)B * L;
  char* ln = ln_rows_out ? (char*)ln_rows_out : narrow;   // the LayerNorm'ed rows, kept for the caller if asked
  launch_layernorm(dtype, x, ln_post_w, ln_post_b, ln, rows, D, 1e-5f, s);
  GemmParams p;
  memset(&p, 0, sizeof(p));
  p.A = ln; p.lda = split_w(dtype) * D; p.W = proj_w; p.M = (int)rows; p.N = E; p.K = D; p.out = big; p.ldc = E; p.act = act;
  launch_gemm(dtype, EPI_ACT_F32, p, s);
  launch_normalize_rows((const float*)big, seg_out, B, L, 1, E, s);
  if (det_w) {
    p.W = det_w;
    launch_gemm(dtype, EPI_ACT_F32, p, s);
    launch_det_mean((const float*)big, (float*)narrow, (size_t)(big - narrow) / 4, det_out, B, L, 1, E, s);   // narrow is free after the GEMM
  }
  return finish("tap_head");
}

int aaclip_tap_head(const float* x, const float* ln_post_w, const float* ln_post_b, const void* proj_w, int act,
                    float* seg_out, const void* det_w, float* det_out, int B, int L, int D, int E, int dtype, void* ws,
                    size_t ws_bytes, void* stream) {
  return tap_head_impl(x, ln_post_w, ln_post_b, proj_w, act, seg_out, det_w, det_out, nullptr, B, L, D, E, dtype, ws, ws_bytes,
                       stream);
}

int aaclip_tap_head_keep_rows(const float* x, const float* ln_post_w, const float* ln_post_b, const void* proj_w, int act,
                              float* seg_out, const void* det_w, float* det_out, void* ln_rows_out, int B, int L, int D,
                              int E, int dtype, void* ws, size_t ws_bytes, void* stream) {
  REQUIRE(ln_rows_out, "tap_head_keep_rows: null pointer");
  return tap_head_impl(x, ln_post_w, ln_post_b, proj_w, act, seg_out, det_w, det_out, ln_rows_out, B, L, D, E, dtype, ws,
                       ws_bytes, stream);
}

int aaclip_det_head(const float* x, const float* ln_post_w, const float* ln_post_b, const void* det_w, int act,
                    float* det_out, int B, int L, int D, int E, int dtype, void* ws, size_t ws_bytes, void* stream) {
  char *narrow, *big, *rowf;
  int rc = head_common(x, ln_post_w, ln_post_b, B, L, D, E, dtype, ws, ws_bytes, &narrow, &big, &rowf);
  if (rc) return rc;
  REQUIRE(det_w && det_out, "det_head: null pointer");
  hipStream_t s = (hipStream_t)stream;
  const long rows = (long)B * L;
  launch_layernorm(dtype, x, ln_post_w, ln_post_b, narrow, rows, D, 1e-5f, s);
  GemmParams p;
  memset(&p, 0, sizeof(p));
  p.A = narrow; p.lda = split_w(dtype) * D; p.W = det_w; p.M = (int)rows; p.N = E; p.K = D; p.out = big; p.ldc = E; p.act = act;
  launch_gemm(dtype, EPI_ACT_F32, p, s);
  launch_det_mean((const float*)big, (float*)narrow, (size_t)(big - narrow) / 4, det_out, B, L, 1, E, s);
  return finish("det_head");
}

int aaclip_anomaly_map(const float* const* seg, int NL, const float* anchors, long anchor_bstride, float* out, int B,
                       int g, int E, int S, int ksize, float sigma, void* ws, size_t ws_bytes, void* stream) {
  REQUIRE(seg && anchors && out && ws, "anomaly_map: null pointer");
  REQUIRE(NL >= 1 && NL <= 4, "anomaly_map: 1..4 levels");
  REQUIRE(B > 0 && B <= 65535 && g >= 1 && g <= 40 && S >= 1, "anomaly_map: bad shape (grid <= 40)");
  REQUIRE(ksize >= 1 && ksize <= 15 && ksize / 2 < g, "anomaly_map: kernel size must be 1..15 and < 2*grid");
  REQUIRE(sigma > 0.f, "anomaly_map: sigma must be positive");
  const char* m = row_width_check(E);
  if (m) return fail(-1, m);
  const int P = g * g;
  REQUIRE(ws_bytes >= (size_t)NL * B * P * 4, "anomaly_map: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  float* pre = (float*)ws;
  for (int l = 0; l < NL; ++l) {
    REQUIRE(seg[l], "anomaly_map: null level pointer");
    launch_patch_scores(seg[l], anchors, anchor_bstride, pre + (size_t)l * B * P, B, P, E, 0, s);
  }
  launch_blur_upsample(pre, out, B, g, S, NL, ksize, sigma, s);
  return finish("anomaly_map");
}

int aaclip_similarity_map_train(const float* seg, const float* anchors, long anchor_bstride, float* out, int B, int g,
                                int E, int S, void* ws, size_t ws_bytes, void* stream) {
  REQUIRE(seg && anchors && out && ws, "similarity_map_train: null pointer");
  REQUIRE(B > 0 && B <= 65535 && g >= 1 && g <= 40 && S >= 1, "similarity_map_train: bad shape (grid <= 40)");
  const char* m = row_width_check(E);
  if (m) return fail(-1, m);
  const int P = g * g;
  REQUIRE(ws_bytes >= (size_t)2 * B * P * 4, "similarity_map_train: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  launch_patch_scores(seg, anchors, anchor_bstride, (float*)ws, B, P, E, 1, s);
  launch_upsample_softmax2((const float*)ws, out, B, g, S, s);
  return finish("similarity_map_train");
}

int aaclip_resample_ksize(int in_size, int out_size) {
  if (in_size < 1 || out_size < 1) return fail(-1, "resample_ksize: sizes must be positive");
  return resample_ksize(in_size, out_size);
}

int aaclip_resample_table(int in_size, int out_size, int32_t* bounds, int32_t* coefs) {
  REQUIRE(bounds && coefs, "resample_table: null pointer");
  REQUIRE(in_size >= 1 && out_size >= 1 && in_size <= (1 << 16) && out_size <= (1 << 16),
          "resample_table: sizes must be 1..65536");
  resample_table(in_size, out_size, bounds, coefs);
  return 0;
}

int aaclip_preprocess(const uint8_t* src, int B, int Hs, int Ws, int S, const int32_t* hbounds, const int32_t* hcoefs,
                      const int32_t* vbounds, const int32_t* vcoefs, const float* lut, float* out, void* stream) {
  REQUIRE(src && hbounds && hcoefs && vbounds && vcoefs && lut && out, "preprocess: null pointer");
  REQUIRE(B > 0 && B <= 65535 && Hs >= 1 && Ws >= 1 && Hs <= (1 << 16) && Ws <= (1 << 16), "preprocess: bad source shape");
  REQUIRE(S >= 1 && S <= 4096, "preprocess: output size must be 1..4096");
  // largest tile height whose source rectangle (rows x pitch bytes) and horizontally resampled rows
  // (rows x 64 columns x 3 planes) fit in 60 KiB of LDS
  int pitch = preprocess_row_pitch(Ws, S);
  int ty = 16, rows = 0;
  for (; ty >= 2; ty >>= 1) {
    rows = preprocess_tile_rows(Hs, S, ty);
    if ((size_t)rows * (pitch + 3 * 64) <= 60 * 1024) break;
  }
  if (ty < 2) {   // very large frames (downscale beyond ~6x): no LDS copy of the source, pass 1 gathers from global memory
    pitch = 0;
    for (ty = 16; ty >= 1; ty >>= 1) {
      rows = preprocess_tile_rows(Hs, S, ty);
      if ((size_t)rows * 3 * 64 <= 60 * 1024) break;
    }
  }
  REQUIRE(ty >= 1, "preprocess: source too tall for one output row to fit in LDS (downscale beyond ~100x)");
  launch_preprocess(src, B, Hs, Ws, S, hbounds, hcoefs, resample_ksize(Ws, S), vbounds, vcoefs, resample_ksize(Hs, S),
                    ty, rows, pitch, lut, out, (hipStream_t)stream);
  return finish("preprocess");
}

int aaclip_text_embed(const int32_t* tokens, const float* table, const float* pos, float* x, int n, int T, int D,
                      int vocab, void* stream) {
  REQUIRE(tokens && table && pos && x, "text_embed: null pointer");
  REQUIRE(n > 0 && T > 0 && D % 4 == 0 && vocab > 0, "text_embed: bad shape");
  launch_embed_text(tokens, table, pos, x, n, T, D, vocab, (hipStream_t)stream);
  return finish("text_embed");
}

int aaclip_row_head(const float* x, const int32_t* tokens, const float* ln_w, const float* ln_b, const void* proj_w,
                    int act, float* out, int n, int T, int D, int E, int mode, int dtype, void* ws, size_t ws_bytes,
                    void* stream) {
  REQUIRE(dtype_ok(dtype), "row_head: bad dtype");
  REQUIRE(x && ln_w && ln_b && proj_w && out && ws, "row_head: null pointer");
  REQUIRE(mode == 1 || tokens, "row_head: tokens required for EOT mode");
  REQUIRE(n > 0 && T > 0, "row_head: bad shape");
  const char* m = row_width_check(D);
  if (m) return fail(-1, m);
  REQUIRE(E % 128 == 0, "row_head: E must be a multiple of 128");
  const size_t es = esize(dtype);
  const long rows = (long)n * T;
  size_t need = up256((size_t)rows * D * es) + up256((size_t)n * D * es);
  REQUIRE(ws_bytes >= need, "row_head: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  char* ln_out = (char*)ws;
  char* picked = ln_out + up256((size_t)rows * D * es);
  launch_layernorm(dtype, x, ln_w, ln_b, ln_out, rows, D, 1e-5f, s);
  launch_gather_rows(dtype, ln_out, picked, tokens, n, T, D, mode, s);
  GemmParams p;
  memset(&p, 0, sizeof(p));
  p.A = picked; p.lda = split_w(dtype) * D; p.W = proj_w; p.M = n; p.N = E; p.K = D; p.out = out; p.ldc = E; p.act = act;
  const char* gm = gemm_check(dtype, EPI_ACT_F32, p);
  if (gm) return fail(-1, gm);
  launch_gemm(dtype, EPI_ACT_F32, p, s);
  return finish("row_head");
}

}  // extern "C"
