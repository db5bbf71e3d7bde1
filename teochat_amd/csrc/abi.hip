// extern "C" surface of libteo_hip.so (declared in include/teo_hip.h).  Argument validation + dispatch only.
#include <stdarg.h>

#include <new>

#include "ops.h"

namespace teo {

static thread_local char g_err[512] = "";

thread_local const char* g_last_kernel = "";
void note_kernel(const char* name) { g_last_kernel = name; }

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int device_cu_count() {
    static int cached[64];                      // 0 = not queried yet (a failed query is retried; a benign race writes the same value)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (dev >= 0 && dev < 64 && cached[dev] > 0) return cached[dev];
    hipDeviceProp_t p;
    const int cus = hipGetDeviceProperties(&p, dev) == hipSuccess ? p.multiProcessorCount : 0;
    if (dev >= 0 && dev < 64 && cus > 0) cached[dev] = cus;
    return cus;
}

int lds_attr_once(const void* kernel, int bytes, unsigned long long* mask, const char* what) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    const unsigned long long bit = 1ull << (dev & 63);
    if (__atomic_load_n(mask, __ATOMIC_RELAXED) & bit) return TEO_OK;
    const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) { set_error("%s: hipFuncSetAttribute(%d bytes of LDS): %s", what, bytes, hipGetErrorString(e)); return TEO_ERR_HIP; }
    __atomic_fetch_or(mask, bit, __ATOMIC_RELAXED);
    return TEO_OK;
}

// ---- performance knobs (tune.h): the block in effect on this thread --------------------------------------------------------------
static const teo_tune g_tune_defaults;                       // immutable: what ships
static thread_local const teo_tune* g_tune_bound = nullptr;  // teo_tune_bind / TuneScope
const teo_tune& tune() { return g_tune_bound ? *g_tune_bound : g_tune_defaults; }
TuneScope::TuneScope(const teo_tune* t) : prev(g_tune_bound), active(t != nullptr) { if (active) g_tune_bound = t; }
TuneScope::~TuneScope() { if (active) g_tune_bound = prev; }

int hip_fail(hipError_t e, const char* what) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return TEO_ERR_HIP;
}

size_t vit_workspace_bytes(const teo_vit_desc* d, int T);
int vit_encode(const teo_vit_desc* d, const void* pixels, int T, void* features, void* ws, size_t ws_bytes, hipStream_t st);
size_t projector_workspace_bytes(const teo_proj_desc* d, int rows);
int projector(const teo_proj_desc* d, const void* x, int rows, void* y, void* ws, size_t ws_bytes, hipStream_t st);
size_t llama_prefill_workspace_bytes(const teo_llama_desc* d, int S);
int llama_prefill(const teo_llama_desc* d, const void* embeds, const int* positions, int S, int past, int last_only,
                  float* logits, void* ws, size_t ws_bytes, hipStream_t st, void* hidden_states, void* attentions);
int llama_prefill_batch(const teo_llama_desc* d, const void* embeds, const int* seq_lens, int nseq, long long cache_stride,
                        int last_only, float* logits, void* ws, size_t ws_bytes, hipStream_t st, void* hidden_states);
size_t llama_decode_workspace_bytes(const teo_llama_desc* d);
int llama_decode_step(const teo_llama_desc* d, const teo_decode_state* s, void* ws, size_t ws_bytes, hipStream_t st);
int llama_decode_step_profile(const teo_llama_desc* d, const teo_decode_state* s, void* ws, size_t ws_bytes, float* ms_out, int* count_out,
                              hipStream_t st);
int llama_decode_batch_step_profile(const teo_llama_desc* d, const teo_decode_batch_state* s, void* ws, size_t ws_bytes, float* ms_out,
                                    int* count_out, hipStream_t st);
int llama_prefill_workspace_status(const teo_llama_desc* d, int S, void* ws, size_t ws_bytes, int* host_flag, hipStream_t st);
int vit_workspace_status(const teo_vit_desc* d, int T, void* ws, size_t ws_bytes, int* host_flag, hipStream_t st);
int llama_decode_begin(const teo_llama_desc* d, const teo_decode_state* s, void* ws, size_t ws_bytes, hipStream_t st);
int decode_graph_create(const teo_llama_desc* d, const teo_decode_state* s, void* ws, size_t ws_bytes, hipStream_t st,
                        teo_graph** out);

size_t llama_decode_batch_workspace_bytes(const teo_llama_desc* d, int batch);
int llama_decode_batch_begin(const teo_llama_desc* d, const teo_decode_batch_state* s, void* ws, size_t ws_bytes, hipStream_t st);
int llama_decode_batch_step(const teo_llama_desc* d, const teo_decode_batch_state* s, void* ws, size_t ws_bytes, hipStream_t st);
int decode_batch_graph_create(const teo_llama_desc* d, const teo_decode_batch_state* s, void* ws, size_t ws_bytes,
                              hipStream_t st, teo_graph** out);

static bool dtype_ok(int dt) { return dt == TEO_F32 || dt == TEO_BF16 || dt == TEO_F16; }

}  // namespace teo

using namespace teo;

#define ST(s) ((hipStream_t)(s))
#define NEED(p, name) TEO_CHECK_ARG((p) != nullptr, "%s: null %s", __func__, name)
#define NEED_DT(dt) TEO_CHECK_ARG(dtype_ok(dt), "%s: bad dtype %d", __func__, dt)
// hipGetLastError is sticky per thread: drop whatever an earlier, unrelated runtime call left behind so that our
// post-launch checks only ever report our own failures.
#define ENTER() (void)hipGetLastError()

extern "C" {

int teo_version(void) { return TEO_ABI_VERSION; }
const char* teo_last_error(void) { return g_err; }
const char* teo_last_kernel(void) { return teo::g_last_kernel; }

teo_tune* teo_tune_create(void) { return new (std::nothrow) teo_tune(); }
int teo_tune_destroy(teo_tune* t) {
    if (t && g_tune_bound == t) g_tune_bound = nullptr;       // other threads that bound it must unbind first (contract in the header)
    delete t;
    return TEO_OK;
}
int teo_tune_set(teo_tune* t, const char* key, int value) {
    TEO_CHECK_ARG(t != nullptr && key != nullptr, "teo_tune_set: null %s", t ? "key" : "block");
    const int v = value;
#define TEO_TUNE_SET(name, def, ok)                                                                       \
    if (!strcmp(key, #name)) {                                                                            \
        TEO_CHECK_ARG((ok), "teo_tune_set: %d is not a value of \"%s\"", value, key);                     \
        t->name = v;                                                                                      \
        return TEO_OK;                                                                                    \
    }
    TEO_TUNE_KEYS(TEO_TUNE_SET)
#undef TEO_TUNE_SET
    set_error("teo_tune_set: unknown key \"%s\"", key);
    return TEO_ERR_ARG;
}
int teo_tune_get(const teo_tune* t, const char* key, int* value) {
    TEO_CHECK_ARG(key != nullptr && value != nullptr, "teo_tune_get: null argument");
    const teo_tune& b = t ? *t : g_tune_defaults;              // NULL: the shipped defaults
#define TEO_TUNE_GET(name, def, ok) if (!strcmp(key, #name)) { *value = b.name; return TEO_OK; }
    TEO_TUNE_KEYS(TEO_TUNE_GET)
#undef TEO_TUNE_GET
    set_error("teo_tune_get: unknown key \"%s\"", key);
    return TEO_ERR_ARG;
}
int teo_tune_reset(teo_tune* t) {
    TEO_CHECK_ARG(t != nullptr, "teo_tune_reset: null block");
    *t = teo_tune();
    return TEO_OK;
}
int teo_tune_bind(const teo_tune* t) { g_tune_bound = t; return TEO_OK; }
const char* teo_tune_keys(void) {
#define TEO_TUNE_NAME(name, def, ok) #name " "
    return TEO_TUNE_KEYS(TEO_TUNE_NAME);
#undef TEO_TUNE_NAME
}
size_t teo_sizeof(const char* struct_name) {
    if (!struct_name) return 0;
    if (!strcmp(struct_name, "teo_vit_desc")) return sizeof(teo_vit_desc);
    if (!strcmp(struct_name, "teo_proj_desc")) return sizeof(teo_proj_desc);
    if (!strcmp(struct_name, "teo_llama_desc")) return sizeof(teo_llama_desc);
    if (!strcmp(struct_name, "teo_decode_state")) return sizeof(teo_decode_state);
    if (!strcmp(struct_name, "teo_decode_batch_state")) return sizeof(teo_decode_batch_state);
    if (!strcmp(struct_name, "teo_attn_args")) return sizeof(teo_attn_args);
    return 0;
}

int teo_gemm_uses_mfma(int M, int N, int K, int dtype, unsigned flags) {
    return gemm_mfma_ok(M, N, K, K, (flags & TEO_GEMM_SWIGLU16) ? N / 2 : N, dtype, flags, nullptr, nullptr, nullptr,
                        nullptr, nullptr) ? 1 : 0;
}

int teo_layernorm(const void* x, const void* w, const void* b, void* y, int rows, int dim, float eps, int dtype,
                  teo_stream_t s) {
    ENTER();
    NEED_DT(dtype); TEO_CHECK_ARG(rows >= 0 && dim > 0, "teo_layernorm: rows %d dim %d", rows, dim);
    if (rows) { NEED(x, "x"); NEED(w, "w"); NEED(b, "b"); NEED(y, "y"); }
    return layernorm(x, w, b, y, rows, dim, eps, dtype, ST(s));
}

int teo_rmsnorm(const void* x, const void* w, void* y, int rows, int dim, float eps, int dtype, teo_stream_t s) {
    ENTER();
    NEED_DT(dtype); TEO_CHECK_ARG(rows >= 0 && dim > 0, "teo_rmsnorm: rows %d dim %d", rows, dim);
    if (rows) { NEED(x, "x"); NEED(w, "w"); NEED(y, "y"); }
    return rmsnorm(x, w, y, rows, dim, eps, dtype, ST(s));
}

int teo_gemm(const void* A, const void* W, const void* bias, const void* res, void* C, int M, int N, int K, int lda,
             int ldc, int act, unsigned flags, int dtype, int out_dtype, teo_stream_t s) {
    ENTER();
    NEED_DT(dtype); NEED_DT(out_dtype);
    TEO_CHECK_ARG(M >= 0 && N >= 0 && K > 0 && lda >= K, "teo_gemm: M %d N %d K %d lda %d", M, N, K, lda);
    TEO_CHECK_ARG(ldc >= ((flags & TEO_GEMM_SWIGLU16) ? N / 2 : N), "teo_gemm: ldc %d too small", ldc);
    TEO_CHECK_ARG(act >= TEO_ACT_NONE && act <= TEO_ACT_QUICK_GELU, "teo_gemm: act %d", act);
    if (M && N) { NEED(A, "A"); NEED(W, "W"); NEED(C, "C"); }
    return gemm(A, W, bias, res, C, M, N, K, lda, ldc, act, flags, dtype, out_dtype, ST(s));
}

int teo_gemm_fp8(const void* A8, const float* a_scale, const void* W8, const float* w_scale, const void* res, void* C, int M, int N, int K,
                 int lda, int ldc, unsigned flags, int out_dtype, teo_stream_t s) {
    ENTER();
    NEED_DT(out_dtype);
    TEO_CHECK_ARG(M >= 0 && N >= 0 && K > 0 && lda >= K, "teo_gemm_fp8: M %d N %d K %d lda %d", M, N, K, lda);
    TEO_CHECK_ARG(ldc >= ((flags & TEO_GEMM_SWIGLU16) ? N / 2 : N), "teo_gemm_fp8: ldc %d too small", ldc);
    if (M && N) { NEED(A8, "A8"); NEED(a_scale, "a_scale"); NEED(W8, "W8"); NEED(w_scale, "w_scale"); NEED(C, "C"); }
    return gemm_fp8(A8, a_scale, W8, w_scale, res, C, M, N, K, lda, ldc, flags, out_dtype, ST(s));
}
int teo_gemm_fp8_ws(const void* A8, const float* a_scale, const void* W8, const float* w_scale, const void* res, void* C, int M, int N,
                    int K, int lda, int ldc, unsigned flags, int out_dtype, void* ws, teo_stream_t s) {
    ENTER();
    NEED_DT(out_dtype);
    TEO_CHECK_ARG(M >= 0 && N >= 0 && K > 0 && lda >= K, "teo_gemm_fp8_ws: M %d N %d K %d lda %d", M, N, K, lda);
    TEO_CHECK_ARG(ldc >= ((flags & TEO_GEMM_SWIGLU16) ? N / 2 : N), "teo_gemm_fp8_ws: ldc %d too small", ldc);
    TEO_CHECK_ARG(ws == nullptr || (reinterpret_cast<uintptr_t>(ws) & 255) == 0, "teo_gemm_fp8_ws: workspace must be 256-byte aligned");
    if (M && N) { NEED(A8, "A8"); NEED(a_scale, "a_scale"); NEED(W8, "W8"); NEED(w_scale, "w_scale"); NEED(C, "C"); }
    return gemm_fp8(A8, a_scale, W8, w_scale, res, C, M, N, K, lda, ldc, flags, out_dtype, ST(s), ws);
}
int teo_quant_rows_fp8(const void* x, const void* norm_w, void* q, float* scale, int rows, int K, int ldx, float eps, teo_stream_t s) {
    ENTER();
    TEO_CHECK_ARG(rows >= 0 && K > 0 && ldx >= K, "teo_quant_rows_fp8: rows %d K %d ldx %d", rows, K, ldx);
    if (rows) { NEED(x, "x"); NEED(q, "q"); NEED(scale, "scale"); }
    return quant_rows_fp8(x, norm_w, q, scale, rows, K, ldx, eps, ST(s));
}

size_t teo_gemm_workspace_bytes(void) { return gemm_sk_workspace_bytes(); }
int teo_gemm_workspace_init(void* ws, teo_stream_t s) {
    ENTER();
    NEED(ws, "workspace");
    return gemm_sk_workspace_init(ws, ST(s));
}
int teo_gemm_workspace_status(const void* ws, int* host_flag, teo_stream_t s) {
    ENTER();
    NEED(ws, "workspace"); NEED(host_flag, "host_flag");
    return gemm_sk_workspace_status(ws, host_flag, ST(s));
}
int teo_gemm_ws(const void* A, const void* W, const void* bias, const void* res, void* C, int M, int N, int K, int lda, int ldc,
                int act, unsigned flags, int dtype, int out_dtype, void* ws, teo_stream_t s) {
    ENTER();
    NEED_DT(dtype); NEED_DT(out_dtype);
    TEO_CHECK_ARG(M >= 0 && N >= 0 && K > 0 && lda >= K, "teo_gemm_ws: M %d N %d K %d lda %d", M, N, K, lda);
    TEO_CHECK_ARG(ldc >= ((flags & TEO_GEMM_SWIGLU16) ? N / 2 : N), "teo_gemm_ws: ldc %d too small", ldc);
    TEO_CHECK_ARG(act >= TEO_ACT_NONE && act <= TEO_ACT_QUICK_GELU, "teo_gemm_ws: act %d", act);
    TEO_CHECK_ARG(ws == nullptr || (reinterpret_cast<uintptr_t>(ws) & 255) == 0, "teo_gemm_ws: workspace must be 256-byte aligned");
    if (M && N) { NEED(A, "A"); NEED(W, "W"); NEED(C, "C"); }
    return gemm(A, W, bias, res, C, M, N, K, lda, ldc, act, flags, dtype, out_dtype, ST(s), ws);
}

int teo_im2col_patches(const void* px, void* cols, int T, int channels, int image, int patch, int ldcols, int dtype,
                       teo_stream_t s) {
    ENTER();
    NEED_DT(dtype); TEO_CHECK_ARG(T >= 0 && patch > 0 && image > 0, "teo_im2col_patches: bad sizes");
    if (T) { NEED(px, "pixels"); NEED(cols, "cols"); }
    return im2col_patches(px, cols, T, channels, image, patch, ldcols, dtype, ST(s));
}

int teo_patch_embed(const void* px, const void* W, void* out, int T, int channels, int image, int patch, int ldw, int dim, int dtype,
                    teo_stream_t s) {
    ENTER();
    NEED_DT(dtype);
    TEO_CHECK_ARG(T >= 0 && patch > 0 && image > 0 && channels > 0 && dim > 0, "teo_patch_embed: bad sizes");
    if (T == 0) return TEO_OK;
    NEED(px, "pixels"); NEED(W, "weight"); NEED(out, "out");
    if (!patch_embed_ok(channels, image, patch, ldw, dim, dtype, px, W, out)) {
        set_error("teo_patch_embed: needs bf16, image %% patch == 0, ldw %% 64 == 0 and >= channels * patch^2, dim %% 4 == 0, 16-byte aligned weight");
        return TEO_ERR_UNSUPPORTED;
    }
    return patch_embed(px, W, out, T, channels, image, patch, ldw, dim, ST(s), dtype == TEO_F16);
}

int teo_vit_embed_ln(const void* patch, const void* cls, const void* pos, const void* w, const void* b, void* out, int T,
                     int n_patches, int dim, float eps, int dtype, teo_stream_t s) {
    ENTER();
    NEED_DT(dtype);
    TEO_CHECK_ARG(T >= 0 && n_patches > 0 && dim > 0 && dim <= 16384, "teo_vit_embed_ln: bad sizes");
    if (T) { NEED(patch, "patch"); NEED(cls, "cls"); NEED(pos, "pos"); NEED(w, "w"); NEED(b, "b"); NEED(out, "out"); }
    return vit_embed_ln(patch, cls, pos, w, b, out, T, n_patches, dim, eps, dtype, ST(s));
}

int teo_attention(const teo_attn_args* a, int dtype, teo_stream_t s) {
    ENTER();
    NEED_DT(dtype); NEED(a, "args");
    TEO_CHECK_ARG(a->batch >= 0 && a->heads > 0 && a->kv_heads > 0 && a->head_dim > 0 && a->q_len >= 0 && a->kv_len >= 0,
                  "teo_attention: bad sizes");
    if (a->batch && a->q_len) { NEED(a->q, "q"); NEED(a->k, "k"); NEED(a->o, "o"); TEO_CHECK_ARG(a->kv_len > 0, "teo_attention: kv_len 0"); }
    return attention(a, dtype, ST(s));
}

int teo_vit_value_transpose(const void* qkv, void* vt, int T, int N, int heads, int head_dim, int ldv, int dtype,
                            teo_stream_t s) {
    ENTER();
    NEED_DT(dtype); TEO_CHECK_ARG(T >= 0 && N > 0 && ldv >= N, "teo_vit_value_transpose: bad sizes");
    if (T) { NEED(qkv, "qkv"); NEED(vt, "vt"); }
    return vit_value_transpose(qkv, vt, T, N, heads, head_dim, ldv, dtype, ST(s));
}

int teo_rope_kv_append(void* qkv, int ld, const int* positions, const float* cs, const float* sn, void* kc, void* vc,
                       void* vtc, int S, int past, int S_max, int heads, int kv_heads, int head_dim, int dtype,
                       teo_stream_t s) {
    ENTER();
    NEED_DT(dtype);
    TEO_CHECK_ARG(S >= 0 && past >= 0 && past + S <= S_max, "teo_rope_kv_append: S %d past %d S_max %d", S, past, S_max);
    if (S) { NEED(qkv, "qkv"); NEED(cs, "cos"); NEED(sn, "sin"); NEED(kc, "k_cache"); NEED(vc, "v_cache"); }
    return rope_kv_append(qkv, ld, positions, cs, sn, kc, vc, vtc, S, past, nullptr, S_max, heads, kv_heads, head_dim,
                          dtype, ST(s));
}

int teo_embed_splice(const int* plan, const void* embed, const void* visual, void* out, int rows, int dim, int dtype,
                     teo_stream_t s) {
    ENTER();
    NEED_DT(dtype); TEO_CHECK_ARG(rows >= 0 && dim > 0, "teo_embed_splice: bad sizes");
    if (rows) { NEED(plan, "plan"); NEED(embed, "embed"); NEED(out, "out"); }
    return embed_splice(plan, embed, visual, out, rows, dim, dtype, ST(s));
}

int teo_drop_cls(const void* in, void* out, int T, int n_tokens, int dim, int dtype, teo_stream_t s) {
    ENTER();
    NEED_DT(dtype); TEO_CHECK_ARG(T >= 0 && n_tokens >= 1 && dim > 0, "teo_drop_cls: bad sizes");
    if (T && n_tokens > 1) { NEED(in, "in"); NEED(out, "out"); }
    return drop_cls(in, out, T, n_tokens, dim, dtype, ST(s));
}

int teo_argmax(const float* logits, long long* tok, int rows, int vocab, teo_stream_t s) {
    ENTER();
    TEO_CHECK_ARG(rows >= 0 && vocab > 0, "teo_argmax: bad sizes");
    if (rows) { NEED(logits, "logits"); NEED(tok, "token"); }
    return argmax(logits, tok, rows, vocab, ST(s));
}

int teo_sample_topk(const float* logits, long long* tok, int vocab, float temperature, int top_k, float top_p,
                    unsigned long long seed, unsigned long long draw, teo_stream_t s) {
    ENTER();
    TEO_CHECK_ARG(vocab > 0 && temperature > 0.f, "teo_sample_topk: vocab %d temperature %g", vocab, temperature);
    NEED(logits, "logits"); NEED(tok, "token");
    { const int rc = sampler_check(vocab, top_k, top_p); if (rc != TEO_OK) return rc; }
    return sample_topk(logits, tok, vocab, temperature, top_k, top_p, seed, draw, ST(s));
}

int teo_gemv(const void* x, const void* W, const void* norm_w, const void* res, void* y, int N, int K, float eps,
             unsigned flags, int dtype, int out_dtype, teo_stream_t s) {
    ENTER();
    NEED_DT(dtype); NEED_DT(out_dtype); TEO_CHECK_ARG(N >= 0 && K > 0, "teo_gemv: N %d K %d", N, K);
    if (N) { NEED(x, "x"); NEED(W, "W"); NEED(y, "y"); }
    return gemv(x, W, norm_w, res, y, N, K, eps, flags, dtype, out_dtype, ST(s));
}

int teo_gemv_w8(const void* x, const void* W8, const float* ws, const void* norm_w, const void* res, void* y, int N, int K,
                float eps, unsigned flags, int out_dtype, teo_stream_t s) {
    ENTER();
    NEED_DT(out_dtype); TEO_CHECK_ARG(N >= 0 && K > 0, "teo_gemv_w8: N %d K %d", N, K);
    if (N) { NEED(x, "x"); NEED(W8, "W8"); NEED(ws, "w_scale"); NEED(y, "y"); }
    return gemv_w(x, W8, ws, 1, norm_w, res, y, N, K, eps, flags, TEO_BF16, out_dtype, ST(s));
}

size_t teo_vit_workspace_bytes(const teo_vit_desc* d, int T) { return d ? vit_workspace_bytes(d, T) : 0; }
int teo_vit_encode(const teo_vit_desc* d, const void* px, int T, void* feat, void* ws, size_t wsb, teo_stream_t s) {
    ENTER();
    NEED(d, "desc"); NEED_DT(d->dtype);
    TuneScope tune_scope(d->tune);
    TEO_CHECK_ARG(T >= 0 && d->hidden % d->heads == 0 && d->image % d->patch == 0, "teo_vit_encode: bad config");
    if (T) { NEED(px, "pixels"); NEED(feat, "features"); NEED(ws, "workspace"); }
    return vit_encode(d, px, T, feat, ws, wsb, ST(s));
}

size_t teo_projector_workspace_bytes(const teo_proj_desc* d, int rows) { return d ? projector_workspace_bytes(d, rows) : 0; }
int teo_projector(const teo_proj_desc* d, const void* x, int rows, void* y, void* ws, size_t wsb, teo_stream_t s) {
    ENTER();
    NEED(d, "desc"); NEED_DT(d->dtype);
    TuneScope tune_scope(d->tune);
    if (rows) { NEED(x, "x"); NEED(y, "y"); }
    return projector(d, x, rows, y, ws, wsb, ST(s));
}

size_t teo_llama_prefill_workspace_bytes(const teo_llama_desc* d, int S) { return d ? llama_prefill_workspace_bytes(d, S) : 0; }
int teo_llama_prefill(const teo_llama_desc* d, const void* emb, const int* pos, int S, int past, int last_only,
                      float* logits, void* ws, size_t wsb, teo_stream_t s, void* hidden_states) {
    ENTER();
    NEED(d, "desc"); NEED_DT(d->dtype);
    TuneScope tune_scope(d->tune);
    TEO_CHECK_ARG(S >= 0 && past >= 0, "teo_llama_prefill: S %d past %d", S, past);
    if (S) { NEED(emb, "embeds"); NEED(logits, "logits"); NEED(ws, "workspace"); }
    return llama_prefill(d, emb, pos, S, past, last_only, logits, ws, wsb, ST(s), hidden_states, nullptr);
}

int teo_llama_prefill_attentions(const teo_llama_desc* d, const void* emb, const int* pos, int S, int past, int last_only,
                                 float* logits, void* ws, size_t wsb, teo_stream_t s, void* hidden_states, void* attentions) {
    ENTER();
    NEED(d, "desc"); NEED_DT(d->dtype); NEED(attentions, "attentions");
    TuneScope tune_scope(d->tune);
    TEO_CHECK_ARG(S >= 0 && past >= 0, "teo_llama_prefill_attentions: S %d past %d", S, past);
    if (S) { NEED(emb, "embeds"); NEED(logits, "logits"); NEED(ws, "workspace"); }
    return llama_prefill(d, emb, pos, S, past, last_only, logits, ws, wsb, ST(s), hidden_states, attentions);
}

int teo_llama_prefill_batch(const teo_llama_desc* d, const void* emb, const int* seq_lens, int nseq, long long cache_stride,
                            int last_only, float* logits, void* ws, size_t wsb, teo_stream_t s, void* hidden_states) {
    ENTER();
    NEED(d, "desc"); NEED_DT(d->dtype); NEED(seq_lens, "seq_lens");
    TuneScope tune_scope(d->tune);
    TEO_CHECK_ARG(nseq >= 0 && (nseq <= 1 || cache_stride > 0), "teo_llama_prefill_batch: nseq %d cache_stride %lld", nseq, cache_stride);
    if (nseq == 0) return TEO_OK;
    NEED(emb, "embeds"); NEED(logits, "logits"); NEED(ws, "workspace");
    return llama_prefill_batch(d, emb, seq_lens, nseq, cache_stride, last_only, logits, ws, wsb, ST(s), hidden_states);
}

size_t teo_llama_decode_workspace_bytes(const teo_llama_desc* d) { return d ? llama_decode_workspace_bytes(d) : 0; }
int teo_llama_decode_step(const teo_llama_desc* d, const teo_decode_state* st, void* ws, size_t wsb, teo_stream_t s) {
    ENTER();
    NEED(d, "desc"); NEED(st, "state"); NEED(ws, "workspace"); NEED_DT(d->dtype);
    TuneScope tune_scope(d->tune);
    NEED(st->d_token, "d_token"); NEED(st->d_pos, "d_pos"); NEED(st->d_out_tokens, "d_out_tokens");
    NEED(st->d_out_count, "d_out_count"); NEED(st->d_logits, "d_logits");
    if (st->do_sample) {
        NEED(st->d_rng, "d_rng"); TEO_CHECK_ARG(st->temperature > 0.f, "teo_llama_decode_step: temperature %g", st->temperature);
        const int rc = sampler_check(d->vocab, st->top_k, st->top_p); if (rc != TEO_OK) return rc;
    }
    return llama_decode_step(d, st, ws, wsb, ST(s));
}

int teo_llama_prefill_workspace_status(const teo_llama_desc* d, int S, void* ws, size_t wsb, int* host_flag, teo_stream_t s) {
    ENTER();
    NEED(d, "desc"); NEED(ws, "workspace"); NEED(host_flag, "host_flag");
    return llama_prefill_workspace_status(d, S, ws, wsb, host_flag, ST(s));
}
int teo_vit_workspace_status(const teo_vit_desc* d, int T, void* ws, size_t wsb, int* host_flag, teo_stream_t s) {
    ENTER();
    NEED(d, "desc"); NEED(ws, "workspace"); NEED(host_flag, "host_flag");
    return vit_workspace_status(d, T, ws, wsb, host_flag, ST(s));
}

int teo_llama_decode_step_profile(const teo_llama_desc* d, const teo_decode_state* st, void* ws, size_t wsb, float* ms_out, int* count_out,
                                  teo_stream_t s) {
    ENTER();
    NEED(d, "desc"); NEED(st, "state"); NEED(ws, "workspace"); NEED_DT(d->dtype);
    TuneScope tune_scope(d->tune);
    NEED(st->d_token, "d_token"); NEED(st->d_pos, "d_pos"); NEED(st->d_out_tokens, "d_out_tokens");
    NEED(st->d_out_count, "d_out_count"); NEED(st->d_logits, "d_logits");
    if (st->do_sample) {
        NEED(st->d_rng, "d_rng"); TEO_CHECK_ARG(st->temperature > 0.f, "teo_llama_decode_step: temperature %g", st->temperature);
        const int rc = sampler_check(d->vocab, st->top_k, st->top_p); if (rc != TEO_OK) return rc;
    }
    NEED(ms_out, "ms_out"); NEED(count_out, "count_out");
    return llama_decode_step_profile(d, st, ws, wsb, ms_out, count_out, ST(s));
}

int teo_llama_decode_begin(const teo_llama_desc* d, const teo_decode_state* st, void* ws, size_t wsb, teo_stream_t s) {
    ENTER();
    NEED(d, "desc"); NEED(st, "state"); NEED(ws, "workspace"); NEED_DT(d->dtype); NEED(st->d_token, "d_token");
    TuneScope tune_scope(d->tune);
    return llama_decode_begin(d, st, ws, wsb, ST(s));
}

int teo_llama_decode_graph_create(const teo_llama_desc* d, const teo_decode_state* st, void* ws, size_t wsb,
                                  teo_stream_t s, teo_graph** out) {
    ENTER();
    NEED(d, "desc"); NEED(st, "state"); NEED(ws, "workspace"); NEED(out, "out"); NEED_DT(d->dtype);
    TuneScope tune_scope(d->tune);
    TEO_CHECK_ARG(s != nullptr, "teo_llama_decode_graph_create: needs a non-default stream to capture on");
    if (st->do_sample) { const int rc = sampler_check(d->vocab, st->top_k, st->top_p); if (rc != TEO_OK) return rc; }
    return decode_graph_create(d, st, ws, wsb, ST(s), out);
}

// ---- batched decode ----
static int check_batch_state(const teo_llama_desc* d, const teo_decode_batch_state* st) {
    NEED(d, "desc"); NEED(st, "state"); NEED_DT(d->dtype);
    TEO_CHECK_ARG(st->batch >= 1 && st->batch <= TEO_MAX_DECODE_BATCH, "decode batch %d outside 1..%d", st->batch, TEO_MAX_DECODE_BATCH);
    NEED(st->d_token, "d_token"); NEED(st->d_pos, "d_pos"); NEED(st->d_out_tokens, "d_out_tokens");
    NEED(st->d_out_count, "d_out_count"); NEED(st->d_stop, "d_stop"); NEED(st->d_logits, "d_logits");
    TEO_CHECK_ARG(st->batch == 1 || st->cache_stride > 0, "decode batch: cache_stride %lld", (long long)st->cache_stride);
    TEO_CHECK_ARG(st->out_stride > 0, "decode batch: out_stride %d", st->out_stride);
    if (st->do_sample) {
        NEED(st->d_rng, "d_rng"); TEO_CHECK_ARG(st->temperature > 0.f, "decode batch: temperature %g", st->temperature);
        const int rc = sampler_check(d->vocab, st->top_k, st->top_p); if (rc != TEO_OK) return rc;
    }
    return TEO_OK;
}

size_t teo_llama_decode_batch_workspace_bytes(const teo_llama_desc* d, int batch) {
    return (d && batch >= 1) ? llama_decode_batch_workspace_bytes(d, batch) : 0;
}

int teo_llama_decode_batch_begin(const teo_llama_desc* d, const teo_decode_batch_state* st, void* ws, size_t wsb, teo_stream_t s) {
    ENTER();
    { const int rc = check_batch_state(d, st); if (rc != TEO_OK) return rc; }
    TuneScope tune_scope(d->tune);
    NEED(ws, "workspace");
    return llama_decode_batch_begin(d, st, ws, wsb, ST(s));
}

int teo_llama_decode_batch_step(const teo_llama_desc* d, const teo_decode_batch_state* st, void* ws, size_t wsb, teo_stream_t s) {
    ENTER();
    { const int rc = check_batch_state(d, st); if (rc != TEO_OK) return rc; }
    TuneScope tune_scope(d->tune);
    NEED(ws, "workspace");
    return llama_decode_batch_step(d, st, ws, wsb, ST(s));
}

int teo_llama_decode_batch_step_profile(const teo_llama_desc* d, const teo_decode_batch_state* st, void* ws, size_t wsb, float* ms_out,
                                        int* count_out, teo_stream_t s) {
    ENTER();
    { const int rc = check_batch_state(d, st); if (rc != TEO_OK) return rc; }
    TuneScope tune_scope(d->tune);
    NEED(ws, "workspace"); NEED(ms_out, "ms_out"); NEED(count_out, "count_out");
    return llama_decode_batch_step_profile(d, st, ws, wsb, ms_out, count_out, ST(s));
}

int teo_llama_decode_batch_graph_create(const teo_llama_desc* d, const teo_decode_batch_state* st, void* ws, size_t wsb,
                                        teo_stream_t s, teo_graph** out) {
    ENTER();
    { const int rc = check_batch_state(d, st); if (rc != TEO_OK) return rc; }
    TuneScope tune_scope(d->tune);
    NEED(ws, "workspace"); NEED(out, "out");
    TEO_CHECK_ARG(s != nullptr, "teo_llama_decode_batch_graph_create: needs a non-default stream to capture on");
    return decode_batch_graph_create(d, st, ws, wsb, ST(s), out);
}

int teo_graph_launch(teo_graph* g, int n_times, teo_stream_t s) {
    ENTER();
    NEED(g, "graph");
    for (int i = 0; i < n_times; ++i) {
        hipError_t e = hipGraphLaunch(g->exec, ST(s));
        if (e != hipSuccess) return hip_fail(e, "hipGraphLaunch");
    }
    return TEO_OK;
}

int teo_graph_destroy(teo_graph* g) {
    if (!g) return TEO_OK;
    if (g->exec) (void)hipGraphExecDestroy(g->exec);
    if (g->graph) (void)hipGraphDestroy(g->graph);
    delete g;
    return TEO_OK;
}

size_t teo_attn_decode_workspace_bytes(int heads, int head_dim, int max_seq, int batch) {
    if (!(heads > 0 && head_dim > 0 && max_seq > 0 && batch > 0)) return 0;
    return attn_decode_ws_bytes(heads, head_dim, max_seq, batch);
}

int teo_attn_decode(const void* q, void* k_cache, void* v_cache, void* vt_cache, const float* rope_cos, const float* rope_sin,
                    void* out, float* partials, const int* d_pos, int max_seq, int heads, int kv_heads, int head_dim, float scale,
                    int dtype, int batch, long long q_stride, long long cache_stride, long long o_stride, teo_stream_t s) {
    ENTER();
    NEED_DT(dtype);
    TEO_CHECK_ARG(batch >= 1 && heads > 0 && kv_heads > 0 && heads % kv_heads == 0 && head_dim > 0 && max_seq > 0,
                  "teo_attn_decode: batch %d heads %d kv_heads %d head_dim %d max_seq %d", batch, heads, kv_heads, head_dim, max_seq);
    NEED(q, "q"); NEED(k_cache, "k_cache"); NEED(v_cache, "v_cache"); NEED(out, "out"); NEED(partials, "partials"); NEED(d_pos, "d_pos");
    TEO_CHECK_ARG((rope_cos == nullptr) == (rope_sin == nullptr), "teo_attn_decode: rope_cos and rope_sin go together");
    AttnBatch bt;
    bt.batch = batch; bt.q_stride = q_stride; bt.cache_stride = cache_stride; bt.o_stride = o_stride;
    return attn_decode(q, k_cache, v_cache, vt_cache, rope_cos, rope_sin, out, partials, d_pos, max_seq, heads, kv_heads, head_dim,
                       scale, dtype, ST(s), bt);
}

int teo_cross_entropy(const float* logits, long long ld, const long long* labels, float* loss_row, float* out, int rows, int vocab,
                      long long ignore_index, teo_stream_t s) {
    ENTER();
    TEO_CHECK_ARG(rows >= 0 && vocab > 0 && ld >= vocab, "teo_cross_entropy: rows %d vocab %d ld %lld", rows, vocab, ld);
    NEED(out, "out");
    if (rows) { NEED(logits, "logits"); NEED(labels, "labels"); NEED(loss_row, "loss_row"); }
    return cross_entropy(logits, ld, labels, loss_row, out, rows, vocab, ignore_index, ST(s));
}

int teo_preprocess_frames(const unsigned char* src, void* out, int T, int H, int W, int S, const float* mean, const float* stdv,
                          int dtype, teo_stream_t s) {
    ENTER();
    NEED_DT(dtype);
    TEO_CHECK_ARG(T >= 0 && H > 0 && W > 0 && S > 0, "teo_preprocess_frames: T %d H %d W %d S %d", T, H, W, S);
    NEED(mean, "mean"); NEED(stdv, "std");
    if (T) { NEED(src, "src"); NEED(out, "out"); }
    return preprocess_frames(src, out, T, H, W, S, mean, stdv, dtype, ST(s));
}

int teo_preprocess_frames_pad(const unsigned char* src, void* out, int T, int H, int W, int S, const float* mean, const float* stdv,
                              const unsigned char* pad_rgb, int dtype, teo_stream_t s) {
    ENTER();
    NEED_DT(dtype);
    TEO_CHECK_ARG(T >= 0 && H > 0 && W > 0 && S > 0, "teo_preprocess_frames_pad: T %d H %d W %d S %d", T, H, W, S);
    NEED(mean, "mean"); NEED(stdv, "std"); NEED(pad_rgb, "pad_rgb");
    if (T) { NEED(src, "src"); NEED(out, "out"); }
    return preprocess_frames(src, out, T, H, W, S, mean, stdv, dtype, ST(s), pad_rgb);
}

int teo_gemm_skinny(const void* x, const void* W, const float* w_scale, int w_fp8, const void* norm_w, float eps, const void* res,
                    void* out, int MB, int N, int K, int ldx, int ldo, unsigned flags, int out_dtype, teo_stream_t s) {
    ENTER();
    TEO_CHECK_ARG(MB >= 0 && N >= 0 && K > 0, "teo_gemm_skinny: MB %d N %d K %d", MB, N, K);
    TEO_CHECK_ARG(out_dtype == TEO_BF16 || out_dtype == TEO_F32 || out_dtype == TEO_F16, "teo_gemm_skinny: out_dtype %d", out_dtype);
    if (MB == 0 || N == 0) return TEO_OK;
    NEED(x, "x"); NEED(W, "W"); NEED(out, "out");
    return skinny_gemm(x, W, w_scale, w_fp8, norm_w, eps, res, out, MB, N, K, ldx, ldo, flags, out_dtype, ST(s));
}

}  // extern "C"
