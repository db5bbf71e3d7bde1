// Composed entry points: the ViT / projector / LLaMA layer loops and the greedy decode step live here, in C++,
// so that one host call enqueues a whole phase on the caller's stream (no per-op Python round trips) and the
// decode step can be captured into a hipGraph and replayed once per token.
#include <algorithm>
#include <vector>

#include "ops.h"

namespace teo {
struct Carver {
    unsigned char* base;
    size_t off = 0, cap;
    Carver(void* p, size_t c) : base((unsigned char*)p), cap(c) {}
    void* take(size_t bytes) {
        void* r = base ? base + off : nullptr;
        off += align_up(bytes);
        return r;
    }
    bool ok() const { return off <= cap; }
};

#define TEO_TRY(expr)                  \
    do {                               \
        int rc__ = (expr);             \
        if (rc__ != TEO_OK) return rc__; \
    } while (0)

// ------------------------------------------------------------------------------------------------
// ViT
// ------------------------------------------------------------------------------------------------
struct VitWs {
    void *cols, *patch, *h, *ln, *qkv, *vt, *attn, *mlp, *sk;
    int ldv;
    size_t total;
};

static VitWs vit_carve(const teo_vit_desc* d, int T, void* ws, size_t cap) {
    const size_t e = esize(d->dtype);
    const int g = d->image / d->patch, NP = g * g, N = NP + 1, D = d->hidden;
    VitWs w;
    w.ldv = (N + 63) / 64 * 64;
    Carver c(ws, cap);
    w.cols = c.take((size_t)T * NP * d->k_pad * e);
    w.patch = c.take((size_t)T * NP * D * e);
    w.h = c.take((size_t)T * N * D * e);
    w.ln = c.take((size_t)T * N * D * e);
    w.qkv = c.take((size_t)T * N * 3 * D * e);
    w.vt = c.take((size_t)T * D * w.ldv * e);
    w.attn = c.take((size_t)T * N * D * e);
    w.mlp = c.take((size_t)T * N * d->inter * e);
    w.sk = c.take(gemm_sk_workspace_bytes());          // stream-K slabs + flags of the MFMA GEMMs
    w.total = c.off;
    return w;
}

size_t vit_workspace_bytes(const teo_vit_desc* d, int T) { return vit_carve(d, T, nullptr, 0).total; }

int vit_encode(const teo_vit_desc* d, const void* pixels, int T, void* features, void* ws, size_t ws_bytes,
               hipStream_t st) {
    if (T == 0) return TEO_OK;
    const VitWs w = vit_carve(d, T, ws, ws_bytes);
    if (w.total > ws_bytes) {
        set_error("teo_vit_encode: workspace %zu < %zu", ws_bytes, w.total);
        return TEO_ERR_WORKSPACE;
    }
    const int dt = d->dtype;
    const int g = d->image / d->patch, NP = g * g, N = NP + 1, D = d->hidden, H = d->heads, hd = D / H;
    const int rows = T * N;
    TEO_TRY(gemm_sk_workspace_init(w.sk, st));
    if (patch_embed_ok(d->channels, d->image, d->patch, d->k_pad, D, dt, pixels, d->patch_w, w.patch)) {
        // patch pixels gathered straight into the MFMA tile's LDS image (patch_embed.hip): no im2col matrix in HBM
        TEO_TRY(patch_embed(pixels, d->patch_w, w.patch, T, d->channels, d->image, d->patch, d->k_pad, D, st, dt == TEO_F16));
    } else {
        TEO_TRY(im2col_patches(pixels, w.cols, T, d->channels, d->image, d->patch, d->k_pad, dt, st));
        TEO_TRY(gemm(w.cols, d->patch_w, nullptr, nullptr, w.patch, T * NP, D, d->k_pad, d->k_pad, D, TEO_ACT_NONE, 0, dt, dt, st, w.sk));
    }
    TEO_TRY(vit_embed_ln(w.patch, d->cls, d->pos, d->pre_ln_w, d->pre_ln_b, w.h, T, NP, D, d->eps, dt, st));
    for (int l = 0; l < d->layers_run; ++l) {
        TEO_TRY(layernorm(w.h, d->ln1_w[l], d->ln1_b[l], w.ln, rows, D, d->eps, dt, st));
        TEO_TRY(gemm(w.ln, d->qkv_w[l], d->qkv_b[l], nullptr, w.qkv, rows, 3 * D, D, D, 3 * D, TEO_ACT_NONE, 0, dt, dt, st, w.sk));
        TEO_TRY(vit_value_transpose(w.qkv, w.vt, T, N, H, hd, w.ldv, dt, st));
        teo_attn_args a;
        memset(&a, 0, sizeof(a));
        const size_t e = esize(dt);
        a.q = w.qkv;
        a.k = (const char*)w.qkv + (size_t)D * e;
        a.v = (const char*)w.qkv + (size_t)2 * D * e;
        a.vt = w.vt;
        a.o = w.attn;
        a.q_bs = a.k_bs = a.v_bs = (long long)N * 3 * D;
        a.q_hs = a.k_hs = a.v_hs = hd;
        a.q_rs = a.k_rs = a.v_rs = 3 * D;
        a.vt_bs = (long long)D * w.ldv;
        a.vt_hs = (long long)hd * w.ldv;
        a.vt_rs = w.ldv;
        a.o_bs = (long long)N * D;
        a.o_rs = D;
        a.batch = T; a.heads = H; a.kv_heads = H; a.head_dim = hd; a.q_len = N; a.kv_len = N;
        a.causal = 0;
        a.scale = 1.0f / sqrtf((float)hd);
        TEO_TRY(attention(&a, dt, st));
        TEO_TRY(gemm(w.attn, d->out_w[l], d->out_b[l], w.h, w.h, rows, D, D, D, D, TEO_ACT_NONE, 0, dt, dt, st, w.sk));
        TEO_TRY(layernorm(w.h, d->ln2_w[l], d->ln2_b[l], w.ln, rows, D, d->eps, dt, st));
        TEO_TRY(gemm(w.ln, d->fc1_w[l], d->fc1_b[l], nullptr, w.mlp, rows, d->inter, D, D, d->inter, d->act, 0, dt, dt, st, w.sk));
        TEO_TRY(gemm(w.mlp, d->fc2_w[l], d->fc2_b[l], w.h, w.h, rows, D, d->inter, d->inter, D, TEO_ACT_NONE, 0, dt, dt, st, w.sk));
    }
    if (d->keep_cls) {                                   // feature_select 'cls_patch': the whole hidden state
        hipError_t he = hipMemcpyAsync(features, w.h, (size_t)rows * D * esize(dt), hipMemcpyDeviceToDevice, st);
        return he == hipSuccess ? TEO_OK : hip_fail(he, "vit_encode copy features");
    }
    return drop_cls(w.h, features, T, N, D, dt, st);
}

// ------------------------------------------------------------------------------------------------
// projector
// ------------------------------------------------------------------------------------------------
size_t projector_workspace_bytes(const teo_proj_desc* d, int rows) {
    return d->depth > 1 ? 2 * align_up((size_t)rows * d->out_dim * esize(d->dtype)) : 0;
}

int projector(const teo_proj_desc* d, const void* x, int rows, void* y, void* ws, size_t ws_bytes, hipStream_t st) {
    if (rows == 0) return TEO_OK;
    TEO_CHECK_ARG(d->depth >= 1 && d->depth <= 4, "teo_projector: depth %d", d->depth);
    if (ws_bytes < projector_workspace_bytes(d, rows)) {
        set_error("teo_projector: workspace too small");
        return TEO_ERR_WORKSPACE;
    }
    const int dt = d->dtype;
    const size_t buf = align_up((size_t)rows * d->out_dim * esize(dt));
    const void* in = x;
    int in_dim = d->in_dim;
    for (int j = 0; j < d->depth; ++j) {
        const bool last = j == d->depth - 1;
        void* out = last ? y : (unsigned char*)ws + (j & 1) * buf;
        TEO_TRY(gemm(in, d->w[j], d->b[j], nullptr, out, rows, d->out_dim, in_dim, in_dim, d->out_dim,
                     last ? TEO_ACT_NONE : TEO_ACT_GELU_ERF, 0, dt, dt, st));
        in = out;
        in_dim = d->out_dim;
    }
    return TEO_OK;
}

// ------------------------------------------------------------------------------------------------
// LLaMA prefill
// ------------------------------------------------------------------------------------------------
struct PrefillWs {
    void *h, *n, *qkv, *attn, *act, *sk, *q8;
    float* qs;
    size_t total;
};

static PrefillWs prefill_carve(const teo_llama_desc* d, int S, void* ws, size_t cap) {
    const size_t e = esize(d->dtype);
    const int QKV = (d->heads + 2 * d->kv_heads) * d->head_dim;
    PrefillWs w;
    Carver c(ws, cap);
    w.h = c.take((size_t)S * d->hidden * e);
    w.n = c.take((size_t)S * d->hidden * e);
    w.qkv = c.take((size_t)S * QKV * e);
    w.attn = c.take((size_t)S * d->heads * d->head_dim * e);
    w.act = c.take((size_t)S * d->inter * e);
    w.sk = c.take(gemm_sk_workspace_bytes());          // stream-K slabs + flags of the MFMA GEMMs
    w.q8 = c.take((size_t)S * (d->inter > d->hidden ? d->inter : d->hidden));   // w8a8 prefill: e4m3 activations of one GEMM ...
    w.qs = (float*)c.take((size_t)S * sizeof(float));                            // ... and their per-token scales
    w.total = c.off;
    return w;
}

// prefill Linear layers on the fp8 MFMA (activations quantised per token, the decode path's e4m3 weights): the descriptor's
// `prefill_fp8` option -- lossy w8a8, selectable per engine, never a default
static bool prefill_uses_fp8(const teo_llama_desc* d) {
    const int Hq = d->heads * d->head_dim;
    return d->prefill_fp8 && d->dtype == TEO_BF16 && d->qkv_w8 && d->o_w8 && d->gateup_w8 && d->down_w8 && d->hidden % 128 == 0 &&
           d->inter % 128 == 0 && Hq % 128 == 0 && d->inter <= 12288 && d->hidden <= 12288;
}
// one decoder layer's four Linear layers in w8a8 form; `attend` runs RoPE / KV append / attention on w.qkv -> w.attn
template <typename F>
static int prefill_layer_fp8(const teo_llama_desc* d, const PrefillWs& w, int l, int S, F attend, hipStream_t st) {
    const int dt = d->dtype;
    const int D = d->hidden, H = d->heads, Hk = d->kv_heads, hd = d->head_dim, Fi = d->inter;
    const int QKV = (H + 2 * Hk) * hd;
    TEO_TRY(quant_rows_fp8(w.h, d->in_norm_w[l], w.q8, w.qs, S, D, D, d->eps, st));
    TEO_TRY(gemm_fp8(w.q8, w.qs, d->qkv_w8[l], d->qkv_s[l], nullptr, w.qkv, S, QKV, D, D, QKV, 0, dt, st));
    TEO_TRY(attend());
    TEO_TRY(quant_rows_fp8(w.attn, nullptr, w.q8, w.qs, S, H * hd, H * hd, d->eps, st));
    TEO_TRY(gemm_fp8(w.q8, w.qs, d->o_w8[l], d->o_s[l], w.h, w.h, S, D, H * hd, H * hd, D, 0, dt, st, w.sk));
    TEO_TRY(quant_rows_fp8(w.h, d->post_norm_w[l], w.q8, w.qs, S, D, D, d->eps, st));
    TEO_TRY(gemm_fp8(w.q8, w.qs, d->gateup_w8[l], d->gateup_s[l], nullptr, w.act, S, 2 * Fi, D, D, Fi, TEO_GEMM_SWIGLU16, dt, st));
    TEO_TRY(quant_rows_fp8(w.act, nullptr, w.q8, w.qs, S, Fi, Fi, d->eps, st));
    return gemm_fp8(w.q8, w.qs, d->down_w8[l], d->down_s[l], w.h, w.h, S, D, Fi, Fi, D, 0, dt, st, w.sk);
}

size_t llama_prefill_workspace_bytes(const teo_llama_desc* d, int S) { return prefill_carve(d, S, nullptr, 0).total; }
// sticky hand-off error word of the GEMM workspace inside a prefill / tower workspace (0 = fine); synchronises the stream
int llama_prefill_workspace_status(const teo_llama_desc* d, int S, void* ws, size_t ws_bytes, int* host_flag, hipStream_t st) {
    const PrefillWs w = prefill_carve(d, S, ws, ws_bytes);
    if (w.total > ws_bytes) { set_error("teo_llama_prefill_workspace_status: workspace too small"); return TEO_ERR_WORKSPACE; }
    return gemm_sk_workspace_status(w.sk, host_flag, st);
}
int vit_workspace_status(const teo_vit_desc* d, int T, void* ws, size_t ws_bytes, int* host_flag, hipStream_t st) {
    const VitWs w = vit_carve(d, T, ws, ws_bytes);
    if (w.total > ws_bytes) { set_error("teo_vit_workspace_status: workspace too small"); return TEO_ERR_WORKSPACE; }
    return gemm_sk_workspace_status(w.sk, host_flag, st);
}

// hidden_states (optional, `output_hidden_states` of the kept forward signature, llava_llama.py:56-69): [layers + 1][S][hidden] in the model
// dtype -- snapshot 0 = the input embeddings, l = the residual stream after layer l - 1, the last one AFTER the final RMSNorm (what
// LlamaModel.forward collects: lm_head(hidden_states[-1]) == logits)
static int snapshot_rows(void* hs, int idx, const void* src, int S, int D, size_t e, hipStream_t st) {
    if (!hs) return TEO_OK;
    const hipError_t he = hipMemcpyAsync((unsigned char*)hs + (size_t)idx * S * D * e, src, (size_t)S * D * e, hipMemcpyDeviceToDevice, st);
    return he == hipSuccess ? TEO_OK : hip_fail(he, "prefill: hidden-state snapshot");
}

// attentions (optional, `output_attentions` of the kept forward signature, llava_llama.py:65,95): [layers][heads][S][past + S] in the model dtype
int llama_prefill(const teo_llama_desc* d, const void* embeds, const int* positions, int S, int past, int last_only,
                  float* logits, void* ws, size_t ws_bytes, hipStream_t st, void* hidden_states, void* attentions) {
    if (S == 0) return TEO_OK;
    TEO_CHECK_ARG(past + S <= d->max_seq, "teo_llama_prefill: past %d + S %d exceeds max_seq %d", past, S, d->max_seq);
    const PrefillWs w = prefill_carve(d, S, ws, ws_bytes);
    if (w.total > ws_bytes) {
        set_error("teo_llama_prefill: workspace %zu < %zu", ws_bytes, w.total);
        return TEO_ERR_WORKSPACE;
    }
    const int dt = d->dtype;
    const size_t e = esize(dt);
    const int D = d->hidden, H = d->heads, Hk = d->kv_heads, hd = d->head_dim, F = d->inter;
    const int QKV = (H + 2 * Hk) * hd;
    hipError_t he = hipMemcpyAsync(w.h, embeds, (size_t)S * D * e, hipMemcpyDeviceToDevice, st);
    if (he != hipSuccess) return hip_fail(he, "prefill copy embeds");
    TEO_TRY(gemm_sk_workspace_init(w.sk, st));
    TEO_TRY(snapshot_rows(hidden_states, 0, w.h, S, D, e, st));
    const bool fp8 = prefill_uses_fp8(d);
    for (int l = 0; l < d->layers; ++l) {
        if (l > 0) TEO_TRY(snapshot_rows(hidden_states, l, w.h, S, D, e, st));         // the residual stream after layer l - 1
        auto attend = [&]() -> int {
        TEO_TRY(rope_kv_append(w.qkv, QKV, positions, d->rope_cos, d->rope_sin, d->k_cache[l], d->v_cache[l],
                               d->vt_cache[l], S, past, nullptr, d->max_seq, H, Hk, hd, dt, st));
        teo_attn_args a;
        memset(&a, 0, sizeof(a));
        a.q = w.qkv; a.k = d->k_cache[l]; a.v = d->v_cache[l]; a.vt = d->vt_cache[l]; a.o = w.attn;
        a.q_bs = 0; a.q_hs = hd; a.q_rs = QKV;
        a.k_bs = 0; a.k_hs = (long long)d->max_seq * hd; a.k_rs = hd;
        a.v_bs = 0; a.v_hs = (long long)d->max_seq * hd; a.v_rs = hd;
        a.vt_bs = 0; a.vt_hs = (long long)hd * d->max_seq; a.vt_rs = d->max_seq;
        a.o_bs = 0; a.o_rs = (long long)H * hd;
        a.batch = 1; a.heads = H; a.kv_heads = Hk; a.head_dim = hd; a.q_len = S; a.kv_len = past + S;
        a.causal = 1;
        a.scale = 1.0f / sqrtf((float)hd);
        if (attentions)      // the maps of this layer, from the operands the attention kernel is about to read (rotated q, cached k)
            TEO_TRY(attention_probs(&a, dt, (unsigned char*)attentions + (size_t)l * H * S * (size_t)(past + S) * e, st));
        return attention(&a, dt, st);
        };
        if (fp8) {
            TEO_TRY(prefill_layer_fp8(d, w, l, S, attend, st));
            continue;
        }
        TEO_TRY(rmsnorm(w.h, d->in_norm_w[l], w.n, S, D, d->eps, dt, st));
        TEO_TRY(gemm(w.n, d->qkv_w[l], nullptr, nullptr, w.qkv, S, QKV, D, D, QKV, TEO_ACT_NONE, 0, dt, dt, st, w.sk));
        TEO_TRY(attend());
        TEO_TRY(gemm(w.attn, d->o_w[l], nullptr, w.h, w.h, S, D, H * hd, H * hd, D, TEO_ACT_NONE, 0, dt, dt, st, w.sk));
        TEO_TRY(rmsnorm(w.h, d->post_norm_w[l], w.n, S, D, d->eps, dt, st));
        TEO_TRY(gemm(w.n, d->gateup_w[l], nullptr, nullptr, w.act, S, 2 * F, D, D, F, TEO_ACT_NONE, TEO_GEMM_SWIGLU16, dt, dt, st, w.sk));
        TEO_TRY(gemm(w.act, d->down_w[l], nullptr, w.h, w.h, S, D, F, F, D, TEO_ACT_NONE, 0, dt, dt, st, w.sk));
    }
    if (last_only) {
        if (hidden_states) TEO_TRY(rmsnorm(w.h, d->final_norm_w, (unsigned char*)hidden_states + (size_t)d->layers * S * D * e, S, D, d->eps, dt, st));
        const void* hl = (const unsigned char*)w.h + (size_t)(S - 1) * D * e;
        return gemv(hl, d->lm_head, d->final_norm_w, nullptr, logits, d->vocab, D, d->eps, 0, dt, TEO_F32, st);
    }
    TEO_TRY(rmsnorm(w.h, d->final_norm_w, w.n, S, D, d->eps, dt, st));
    TEO_TRY(snapshot_rows(hidden_states, d->layers, w.n, S, D, e, st));
    return gemm(w.n, d->lm_head, nullptr, nullptr, logits, S, d->vocab, D, D, d->vocab, TEO_ACT_NONE, 0, dt, TEO_F32, st, w.sk);
}

// Multi-sequence prefill (batched generate): the rows of nseq new conversations are concatenated so the norms and the
// GEMMs run once over sum(S_b) rows (better tile quantisation than nseq separate M = S_b problems); RoPE + KV append and
// the causal attention run per sequence on its row block and its own cache slot (slot b = cache pointer + b*cache_stride
// elements, fresh caches: past = 0).  logits [nseq, vocab]: last position of every sequence.
int llama_prefill_batch(const teo_llama_desc* d, const void* embeds, const int* seq_lens, int nseq, long long cache_stride,
                        int last_only, float* logits, void* ws, size_t ws_bytes, hipStream_t st, void* hidden_states) {
    int total = 0;
    for (int b = 0; b < nseq; ++b) {
        TEO_CHECK_ARG(seq_lens[b] >= 1 && seq_lens[b] <= d->max_seq, "teo_llama_prefill_batch: seq_lens[%d] = %d", b, seq_lens[b]);
        total += seq_lens[b];
    }
    const PrefillWs w = prefill_carve(d, total, ws, ws_bytes);
    if (w.total > ws_bytes) {
        set_error("teo_llama_prefill_batch: workspace %zu < %zu", ws_bytes, w.total);
        return TEO_ERR_WORKSPACE;
    }
    const int dt = d->dtype;
    const size_t e = esize(dt);
    const int D = d->hidden, H = d->heads, Hk = d->kv_heads, hd = d->head_dim, F = d->inter;
    const int QKV = (H + 2 * Hk) * hd;
    const int S = total;
    hipError_t he = hipMemcpyAsync(w.h, embeds, (size_t)S * D * e, hipMemcpyDeviceToDevice, st);
    if (he != hipSuccess) return hip_fail(he, "prefill copy embeds");
    TEO_TRY(gemm_sk_workspace_init(w.sk, st));
    TEO_TRY(snapshot_rows(hidden_states, 0, w.h, S, D, e, st));
    const bool fp8 = prefill_uses_fp8(d);
    for (int l = 0; l < d->layers; ++l) {
        if (l > 0) TEO_TRY(snapshot_rows(hidden_states, l, w.h, S, D, e, st));         // the residual stream after layer l - 1
        auto attend = [&]() -> int {
        int row0 = 0;
        for (int b = 0; b < nseq; ++b) {
            const int Sb = seq_lens[b];
            unsigned char* qkv_b = (unsigned char*)w.qkv + (size_t)row0 * QKV * e;
            unsigned char* kc = (unsigned char*)d->k_cache[l] + (size_t)b * cache_stride * e;
            unsigned char* vc = (unsigned char*)d->v_cache[l] + (size_t)b * cache_stride * e;
            unsigned char* vtc = (unsigned char*)d->vt_cache[l] + (size_t)b * cache_stride * e;
            TEO_TRY(rope_kv_append(qkv_b, QKV, nullptr, d->rope_cos, d->rope_sin, kc, vc, vtc, Sb, 0, nullptr, d->max_seq, H, Hk,
                                   hd, dt, st));
            teo_attn_args a;
            memset(&a, 0, sizeof(a));
            a.q = qkv_b; a.k = kc; a.v = vc; a.vt = vtc; a.o = (unsigned char*)w.attn + (size_t)row0 * H * hd * e;
            a.q_bs = 0; a.q_hs = hd; a.q_rs = QKV;
            a.k_bs = 0; a.k_hs = (long long)d->max_seq * hd; a.k_rs = hd;
            a.v_bs = 0; a.v_hs = (long long)d->max_seq * hd; a.v_rs = hd;
            a.vt_bs = 0; a.vt_hs = (long long)hd * d->max_seq; a.vt_rs = d->max_seq;
            a.o_bs = 0; a.o_rs = (long long)H * hd;
            a.batch = 1; a.heads = H; a.kv_heads = Hk; a.head_dim = hd; a.q_len = Sb; a.kv_len = Sb;
            a.causal = 1;
            a.scale = 1.0f / sqrtf((float)hd);
            TEO_TRY(attention(&a, dt, st));
            row0 += Sb;
        }
        return TEO_OK;
        };
        if (fp8) {
            TEO_TRY(prefill_layer_fp8(d, w, l, S, attend, st));
            continue;
        }
        TEO_TRY(rmsnorm(w.h, d->in_norm_w[l], w.n, S, D, d->eps, dt, st));
        TEO_TRY(gemm(w.n, d->qkv_w[l], nullptr, nullptr, w.qkv, S, QKV, D, D, QKV, TEO_ACT_NONE, 0, dt, dt, st, w.sk));
        TEO_TRY(attend());
        TEO_TRY(gemm(w.attn, d->o_w[l], nullptr, w.h, w.h, S, D, H * hd, H * hd, D, TEO_ACT_NONE, 0, dt, dt, st, w.sk));
        TEO_TRY(rmsnorm(w.h, d->post_norm_w[l], w.n, S, D, d->eps, dt, st));
        TEO_TRY(gemm(w.n, d->gateup_w[l], nullptr, nullptr, w.act, S, 2 * F, D, D, F, TEO_ACT_NONE, TEO_GEMM_SWIGLU16, dt, dt, st, w.sk));
        TEO_TRY(gemm(w.act, d->down_w[l], nullptr, w.h, w.h, S, D, F, F, D, TEO_ACT_NONE, 0, dt, dt, st, w.sk));
    }
    if (!last_only) {                                    // training-shape forward: logits of every row, [sum(seq_lens), vocab]
        TEO_TRY(rmsnorm(w.h, d->final_norm_w, w.n, S, D, d->eps, dt, st));
        TEO_TRY(snapshot_rows(hidden_states, d->layers, w.n, S, D, e, st));
        return gemm(w.n, d->lm_head, nullptr, nullptr, logits, S, d->vocab, D, D, d->vocab, TEO_ACT_NONE, 0, dt, TEO_F32, st, w.sk);
    }
    if (hidden_states) TEO_TRY(rmsnorm(w.h, d->final_norm_w, (unsigned char*)hidden_states + (size_t)d->layers * S * D * e, S, D, d->eps, dt, st));
    int row_end = 0;
    for (int b = 0; b < nseq; ++b) {
        row_end += seq_lens[b];
        const void* hl = (const unsigned char*)w.h + (size_t)(row_end - 1) * D * e;
        TEO_TRY(gemv(hl, d->lm_head, d->final_norm_w, nullptr, logits + (size_t)b * d->vocab, d->vocab, D, d->eps, 0, dt, TEO_F32, st));
    }
    return TEO_OK;
}

// ------------------------------------------------------------------------------------------------
// LLaMA greedy decode step
// ------------------------------------------------------------------------------------------------
struct DecodeWs {
    void *h, *qkv, *attn, *act;
    float* part;
    size_t total;
};

static DecodeWs decode_carve(const teo_llama_desc* d, void* ws, size_t cap) {
    const size_t e = esize(d->dtype);
    const int QKV = (d->heads + 2 * d->kv_heads) * d->head_dim;
    DecodeWs w;
    Carver c(ws, cap);
    w.h = c.take((size_t)d->hidden * e);
    w.qkv = c.take((size_t)QKV * e);
    w.attn = c.take((size_t)d->heads * d->head_dim * e);
    w.act = c.take((size_t)d->inter * e);
    w.part = (float*)c.take(attn_decode_ws_bytes(d->heads, d->head_dim, d->max_seq));
    w.total = c.off;
    return w;
}

size_t llama_decode_workspace_bytes(const teo_llama_desc* d) { return decode_carve(d, nullptr, 0).total; }

int llama_decode_step(const teo_llama_desc* d, const teo_decode_state* s, void* ws, size_t ws_bytes, hipStream_t st) {
    const DecodeWs w = decode_carve(d, ws, ws_bytes);
    if (w.total > ws_bytes) {
        set_error("teo_llama_decode_step: workspace %zu < %zu", ws_bytes, w.total);
        return TEO_ERR_WORKSPACE;
    }
    const int dt = d->dtype;
    const int D = d->hidden, H = d->heads, Hk = d->kv_heads, hd = d->head_dim, F = d->inter;
    const int QKV = (H + 2 * Hk) * hd;
    // w.h holds the embedding of *s->d_token: written by the previous step's tail (or by teo_llama_decode_begin)
    const bool w8 = d->qkv_w8 != nullptr;              // decode streams the fp8 copies when they are present
    for (int l = 0; l < d->layers; ++l) {
        if (d->rope_in_attn) {
            // rmsnorm + QKV projection (plain weight stream); RoPE + KV append ride inside the attention kernel
            // (position read from s->d_pos on the device)
            prof_class(TEO_PROF_QKV);
            TEO_TRY(gemv_w(w.h, w8 ? d->qkv_w8[l] : d->qkv_w[l], w8 ? d->qkv_s[l] : nullptr, w8, d->in_norm_w[l], nullptr,
                           w.qkv, QKV, D, d->eps, 0, dt, dt, st));
            prof_class(TEO_PROF_ATTN);
            TEO_TRY(attn_decode(w.qkv, d->k_cache[l], d->v_cache[l], d->vt_cache[l], d->rope_cos, d->rope_sin, w.attn, w.part,
                                s->d_pos, d->max_seq, H, Hk, hd, 1.0f / sqrtf((float)hd), dt, st));
        } else {
            // rmsnorm -> QKV projection -> RoPE -> KV append in the GEMV epilogue
            prof_class(TEO_PROF_QKV);
            TEO_TRY(gemv_qkv_rope(w.h, w8 ? d->qkv_w8[l] : d->qkv_w[l], w8 ? d->qkv_s[l] : nullptr, w8, d->in_norm_w[l], w.qkv,
                                  d->rope_cos, d->rope_sin, s->d_pos, d->k_cache[l], d->v_cache[l], d->vt_cache[l], d->max_seq,
                                  H, Hk, hd, D, d->eps, dt, st));
            prof_class(TEO_PROF_ATTN);
            TEO_TRY(attn_decode(w.qkv, d->k_cache[l], d->v_cache[l], nullptr, nullptr, nullptr, w.attn, w.part, s->d_pos,
                                d->max_seq, H, Hk, hd, 1.0f / sqrtf((float)hd), dt, st));
        }
        prof_class(TEO_PROF_O);
        TEO_TRY(gemv_w(w.attn, w8 ? d->o_w8[l] : d->o_w[l], w8 ? d->o_s[l] : nullptr, w8, nullptr, w.h, w.h, D, H * hd, d->eps,
                       0, dt, dt, st));
        prof_class(TEO_PROF_GATEUP);
        TEO_TRY(gemv_w(w.h, w8 ? d->gateup_w8[l] : d->gateup_w[l], w8 ? d->gateup_s[l] : nullptr, w8, d->post_norm_w[l], nullptr,
                       w.act, 2 * F, D, d->eps, TEO_GEMM_SWIGLU16, dt, dt, st));
        prof_class(TEO_PROF_DOWN);
        TEO_TRY(gemv_w(w.act, w8 ? d->down_w8[l] : d->down_w[l], w8 ? d->down_s[l] : nullptr, w8, nullptr, w.h, w.h, D, F, d->eps,
                       0, dt, dt, st));
    }
    {
        const bool h8 = d->lm_head8 != nullptr;
        prof_class(TEO_PROF_LM_HEAD);
        TEO_TRY(gemv_w(w.h, h8 ? d->lm_head8 : d->lm_head, h8 ? d->lm_head_s : nullptr, h8, d->final_norm_w, nullptr,
                       s->d_logits, d->vocab, D, d->eps, 0, dt, TEO_F32, st));
    }
    // argmax -> append/advance/stop test -> embedding row of the next token into w.h, one launch
    prof_class(TEO_PROF_TAIL);
    return decode_tail(s->d_logits, s, d->embed, w.h, d->vocab, D, dt, st);
}


// ------------------------------------------------------------------------------------------------
// batched decode: B conversations advance one token per step; every weight matrix is streamed once per step
// ------------------------------------------------------------------------------------------------
struct DecodeBatchWs {
    void *h, *hg, *qkv, *attn, *act;
    float *ssq, *part;
    int nparts;
    size_t total;
};

static DecodeBatchWs decode_batch_carve(const teo_llama_desc* d, int B, void* ws, size_t cap) {
    const size_t e = esize(d->dtype);
    const int QKV = (d->heads + 2 * d->kv_heads) * d->head_dim;
    DecodeBatchWs w;
    Carver c(ws, cap);
    w.nparts = cdiv(d->hidden, 16);                       // one partial per 16-column workgroup of the o / down GEMM
    w.h = c.take((size_t)B * d->hidden * e);
    w.hg = c.take((size_t)B * d->hidden * e);
    w.ssq = (float*)c.take((size_t)B * w.nparts * sizeof(float));
    w.qkv = c.take((size_t)B * QKV * e);
    w.attn = c.take((size_t)B * d->heads * d->head_dim * e);
    w.act = c.take((size_t)B * d->inter * e);
    w.part = (float*)c.take(attn_decode_ws_bytes(d->heads, d->head_dim, d->max_seq, B));
    w.total = c.off;
    return w;
}

size_t llama_decode_batch_workspace_bytes(const teo_llama_desc* d, int batch) {
    return decode_batch_carve(d, batch, nullptr, 0).total;
}

// Can every Linear layer of the step run on the MFMA skinny GEMM?  (bf16 activations, K multiples of the k-step.)
static bool batch_uses_skinny(const teo_llama_desc* d, int B) {
    if (d->dtype != TEO_BF16 && d->dtype != TEO_F16) return false;
    const int w8 = d->qkv_w8 != nullptr, h8 = d->lm_head8 != nullptr;
    if (d->dtype == TEO_F16 && (w8 || h8)) return false;   // fp8 weights go with bfloat16 activations
    const int D = d->hidden, H = d->heads, Hk = d->kv_heads, hd = d->head_dim, F = d->inter;
    const void* al = reinterpret_cast<const void*>(16);   // alignment of the real pointers is checked at launch
    return skinny_gemm_ok(B, (H + 2 * Hk) * hd, D, D, w8, 0, al, al) && skinny_gemm_ok(B, D, H * hd, H * hd, w8, 0, al, al) &&
           skinny_gemm_ok(B, 2 * F, D, D, w8, TEO_GEMM_SWIGLU16, al, al) && skinny_gemm_ok(B, D, F, F, w8, 0, al, al) &&
           skinny_gemm_ok(B, d->vocab, D, D, h8, 0, al, al);
}

// fallback row loop: y[b] = f(norm(x[b])) W^T (+ res[b]) with the single-conversation GEMV (fp32, odd K)
static int batch_linear_rows(int B, const void* x, int ldx, const void* W, const float* wscale, int w8, const void* norm_w,
                             const void* res, void* y, int ldy, int N, int K, float eps, unsigned flags, int dt, int out_dt,
                             hipStream_t st) {
    const size_t e = esize(dt), eo = esize(out_dt);
    for (int b = 0; b < B; ++b)
        TEO_TRY(gemv_w((const char*)x + (size_t)b * ldx * e, W, wscale, w8, norm_w, res ? (const char*)res + (size_t)b * ldy * eo : nullptr,
                       (char*)y + (size_t)b * ldy * eo, N, K, eps, flags, dt, out_dt, st));
    return TEO_OK;
}

static teo_decode_state batch_as_state(const teo_decode_batch_state* s) {
    teo_decode_state t;
    t.d_token = s->d_token; t.d_pos = s->d_pos; t.d_out_tokens = s->d_out_tokens; t.d_out_count = s->d_out_count;
    t.d_stop = s->d_stop; t.d_stop_ids = s->d_stop_ids; t.n_stop_ids = s->n_stop_ids; t.d_logits = s->d_logits;
    t.do_sample = s->do_sample; t.top_k = s->top_k; t.temperature = s->temperature; t.d_rng = s->d_rng; t.top_p = s->top_p;
    return t;
}

int llama_decode_batch_begin(const teo_llama_desc* d, const teo_decode_batch_state* s, void* ws, size_t ws_bytes, hipStream_t st) {
    const DecodeBatchWs w = decode_batch_carve(d, s->batch, ws, ws_bytes);
    if (w.total > ws_bytes) {
        set_error("teo_llama_decode_batch_begin: workspace %zu < %zu", ws_bytes, w.total);
        return TEO_ERR_WORKSPACE;
    }
    if (batch_uses_skinny(d, s->batch))                   // also hand layer 0's RMSNorm its inputs (SkinnyFuse)
        return embed_token_emit(s->d_token, d->embed, w.h, d->hidden, d->dtype, st, s->batch, d->in_norm_w[0], w.hg, w.ssq, w.nparts);
    return embed_token(s->d_token, d->embed, w.h, d->hidden, d->dtype, st, s->batch);
}

int llama_decode_batch_step(const teo_llama_desc* d, const teo_decode_batch_state* s, void* ws, size_t ws_bytes, hipStream_t st) {
    const int B = s->batch;
    const DecodeBatchWs w = decode_batch_carve(d, B, ws, ws_bytes);
    if (w.total > ws_bytes) {
        set_error("teo_llama_decode_batch_step: workspace %zu < %zu", ws_bytes, w.total);
        return TEO_ERR_WORKSPACE;
    }
    const int dt = d->dtype;
    const int D = d->hidden, H = d->heads, Hk = d->kv_heads, hd = d->head_dim, F = d->inter;
    const int QKV = (H + 2 * Hk) * hd;
    const bool w8 = d->qkv_w8 != nullptr, h8 = d->lm_head8 != nullptr;
    const bool skinny = batch_uses_skinny(d, B);
    if (!skinny && (s->w_tiled || s->gateup_block8)) {
        set_error("teo_llama_decode_batch_step: tiled weights need bf16 activations and K %% %d == 0", w8 ? 64 : 32);
        return TEO_ERR_UNSUPPORTED;
    }
    const unsigned tl = s->w_tiled ? TEO_GEMM_WTILED : 0u;
    AttnBatch bt;
    bt.batch = B; bt.q_stride = QKV; bt.cache_stride = s->cache_stride; bt.o_stride = (long long)H * hd;
    // w.h holds the residual stream; on the skinny path w.hg = bf16(h * g) and w.ssq = partial sum(h^2) of the norm that
    // comes next -- written by the producer of h (embed / o / down GEMM epilogue), consumed by the next GEMM.
    SkinnyFuse take;                                       // consumer side
    take.ssq_in = w.ssq; take.nparts = w.nparts; take.eps = d->eps;
    take.f16 = dt == TEO_F16;
    for (int l = 0; l < d->layers; ++l) {
        const void* qkv_w = w8 ? d->qkv_w8[l] : d->qkv_w[l];
        const void* o_w = w8 ? d->o_w8[l] : d->o_w[l];
        const void* gu_w = w8 ? d->gateup_w8[l] : d->gateup_w[l];
        const void* dn_w = w8 ? d->down_w8[l] : d->down_w[l];
        const float *qkv_s = w8 ? d->qkv_s[l] : nullptr, *o_s = w8 ? d->o_s[l] : nullptr;
        const float *gu_s = w8 ? d->gateup_s[l] : nullptr, *dn_s = w8 ? d->down_s[l] : nullptr;
        prof_class(TEO_PROF_QKV);
        if (skinny) {
            TEO_TRY(skinny_gemm(w.hg, qkv_w, qkv_s, w8, nullptr, 0.f, nullptr, w.qkv, B, QKV, D, D, QKV, tl, dt, st, take));
        } else {
            TEO_TRY(batch_linear_rows(B, w.h, D, qkv_w, qkv_s, w8, d->in_norm_w[l], nullptr, w.qkv, QKV, QKV, D, d->eps, 0, dt, dt, st));
        }
        // RoPE + KV append of the B new tokens inside the attention kernel, each conversation at its own position
        prof_class(TEO_PROF_ATTN);
        TEO_TRY(attn_decode(w.qkv, d->k_cache[l], d->v_cache[l], d->vt_cache[l], d->rope_cos, d->rope_sin, w.attn, w.part,
                            s->d_pos, d->max_seq, H, Hk, hd, 1.0f / sqrtf((float)hd), dt, st, bt));
        if (skinny) {
            SkinnyFuse give;                               // h += attn Wo^T; hand post_norm its inputs
            give.f16 = dt == TEO_F16;
            give.next_g = (const unsigned short*)d->post_norm_w[l]; give.xg_out = (unsigned short*)w.hg; give.ssq_out = w.ssq;
            prof_class(TEO_PROF_O);
            TEO_TRY(skinny_gemm(w.attn, o_w, o_s, w8, nullptr, 0.f, w.h, w.h, B, D, H * hd, H * hd, D, tl, dt, st, give));
            prof_class(TEO_PROF_GATEUP);
            TEO_TRY(skinny_gemm(w.hg, gu_w, gu_s, w8, nullptr, 0.f, nullptr, w.act, B, 2 * F, D, D, F,
                                tl | (s->gateup_block8 ? TEO_GEMM_SWIGLU8 : TEO_GEMM_SWIGLU16), dt, st, take));
            give.next_g = (const unsigned short*)(l + 1 < d->layers ? d->in_norm_w[l + 1] : d->final_norm_w);
            prof_class(TEO_PROF_DOWN);
            TEO_TRY(skinny_gemm(w.act, dn_w, dn_s, w8, nullptr, 0.f, w.h, w.h, B, D, F, F, D, tl, dt, st, give));
        } else {
            prof_class(TEO_PROF_O);
            TEO_TRY(batch_linear_rows(B, w.attn, H * hd, o_w, o_s, w8, nullptr, w.h, w.h, D, D, H * hd, d->eps, 0, dt, dt, st));
            prof_class(TEO_PROF_GATEUP);
            TEO_TRY(batch_linear_rows(B, w.h, D, gu_w, gu_s, w8, d->post_norm_w[l], nullptr, w.act, F, 2 * F, D, d->eps,
                                      TEO_GEMM_SWIGLU16, dt, dt, st));
            prof_class(TEO_PROF_DOWN);
            TEO_TRY(batch_linear_rows(B, w.act, F, dn_w, dn_s, w8, nullptr, w.h, w.h, D, D, F, d->eps, 0, dt, dt, st));
        }
    }
    const void* head_w = h8 ? d->lm_head8 : d->lm_head;
    const float* head_s = h8 ? d->lm_head_s : nullptr;
    prof_class(TEO_PROF_LM_HEAD);
    if (skinny) {
        TEO_TRY(skinny_gemm(w.hg, head_w, head_s, h8, nullptr, 0.f, nullptr, s->d_logits, B, d->vocab, D, D, d->vocab, tl, TEO_F32, st, take));
    } else {
        TEO_TRY(batch_linear_rows(B, w.h, D, head_w, head_s, h8, d->final_norm_w, nullptr, s->d_logits, d->vocab, d->vocab, D, d->eps, 0,
                                  dt, TEO_F32, st));
    }
    const teo_decode_state t = batch_as_state(s);
    prof_class(TEO_PROF_TAIL);
    if (skinny)
        return decode_tail(s->d_logits, &t, d->embed, w.h, d->vocab, D, dt, st, B, s->out_stride, d->in_norm_w[0], w.hg, w.ssq, w.nparts);
    return decode_tail(s->d_logits, &t, d->embed, w.h, d->vocab, D, dt, st, B, s->out_stride);
}

// h <- embed[*d_token]: arms the first step of a generation (later steps get it from the previous step's tail)
int llama_decode_begin(const teo_llama_desc* d, const teo_decode_state* s, void* ws, size_t ws_bytes, hipStream_t st) {
    const DecodeWs w = decode_carve(d, ws, ws_bytes);
    if (w.total > ws_bytes) {
        set_error("teo_llama_decode_begin: workspace %zu < %zu", ws_bytes, w.total);
        return TEO_ERR_WORKSPACE;
    }
    return embed_token(s->d_token, d->embed, w.h, d->hidden, d->dtype, st);
}

// ------------------------------------------------------------------------------------------------
// profiled decode step: every launch timed by its own dispatch timestamps (TEO_KLAUNCH, common.h)
// ------------------------------------------------------------------------------------------------
struct ProfRec { hipEvent_t start, stop; int cls; };
static thread_local std::vector<ProfRec>* g_prof = nullptr;
static thread_local int g_prof_cls = 0;
bool prof_take(hipEvent_t* start, hipEvent_t* stop) {
    if (!g_prof) return false;
    ProfRec r;
    r.cls = g_prof_cls;
    if (hipEventCreate(&r.start) != hipSuccess) return false;
    if (hipEventCreate(&r.stop) != hipSuccess) { (void)hipEventDestroy(r.start); return false; }
    g_prof->push_back(r);
    *start = r.start;
    *stop = r.stop;
    return true;
}
void prof_class(int cls) { g_prof_cls = cls; }
void prof_bump(int delta) { g_prof_cls += delta; }

// holds the stream for `ticks` of the 100 MHz wall clock: the profiled step's ~200 launches are enqueued behind it (the host needs
// ~9 us per timestamped launch, more than the small kernels run) and then execute back to back, as a graph replay does -- a kernel
// that starts on an idle memory system measures 2-3 % faster than the same kernel inside the real step
__global__ void prof_hold_kernel(unsigned long long ticks) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}

template <typename F>
static int profiled_step(F step, float* ms_out, int* count_out, hipStream_t st, size_t reserve) {
    std::vector<ProfRec> recs;
    recs.reserve(reserve);
    prof_hold_kernel<<<1, 64, 0, st>>>(400000ull);            // 4 ms
    g_prof = &recs;
    const int rc = step();
    g_prof = nullptr;
    g_prof_cls = 0;
    hipError_t e = hipStreamSynchronize(st);
    for (int c = 0; c < TEO_PROF_CLASSES; ++c) { ms_out[c] = 0.f; count_out[c] = 0; }
    for (const ProfRec& r : recs) {
        float ms = 0.f;
        if (e == hipSuccess && rc == TEO_OK) e = hipEventElapsedTime(&ms, r.start, r.stop);
        if (r.cls >= 0 && r.cls < TEO_PROF_CLASSES) { ms_out[r.cls] += ms; count_out[r.cls] += 1; }
        (void)hipEventDestroy(r.start);
        (void)hipEventDestroy(r.stop);
    }
    if (rc != TEO_OK) return rc;
    return e == hipSuccess ? TEO_OK : hip_fail(e, "teo_llama_decode_step_profile");
}

int llama_decode_step_profile(const teo_llama_desc* d, const teo_decode_state* s, void* ws, size_t ws_bytes, float* ms_out,
                              int* count_out, hipStream_t st) {
    return profiled_step([&]() { return llama_decode_step(d, s, ws, ws_bytes, st); }, ms_out, count_out, st, 8 * (size_t)d->layers + 8);
}

int llama_decode_batch_step_profile(const teo_llama_desc* d, const teo_decode_batch_state* s, void* ws, size_t ws_bytes, float* ms_out,
                                    int* count_out, hipStream_t st) {
    return profiled_step([&]() { return llama_decode_batch_step(d, s, ws, ws_bytes, st); }, ms_out, count_out, st,
                         8 * (size_t)d->layers * (size_t)std::max(1, s->batch) + 16);
}

// ------------------------------------------------------------------------------------------------
// hipGraph wrapper
// ------------------------------------------------------------------------------------------------
// capture whatever `enqueue` launches on `st` into an instantiated hipGraph
template <typename F>
static int capture_graph(hipStream_t st, F enqueue, teo_graph** out) {
    *out = nullptr;
    hipError_t e = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
    if (e != hipSuccess) return hip_fail(e, "hipStreamBeginCapture");
    const int rc = enqueue();
    hipGraph_t g = nullptr;
    e = hipStreamEndCapture(st, &g);
    if (rc != TEO_OK) {
        if (g) (void)hipGraphDestroy(g);
        return rc;
    }
    if (e != hipSuccess) return hip_fail(e, "hipStreamEndCapture");
    hipGraphExec_t ex = nullptr;
    e = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
    if (e != hipSuccess) {
        (void)hipGraphDestroy(g);
        return hip_fail(e, "hipGraphInstantiate");
    }
    teo_graph* tg = new teo_graph();
    tg->graph = g;
    tg->exec = ex;
    *out = tg;
    return TEO_OK;
}

int decode_graph_create(const teo_llama_desc* d, const teo_decode_state* s, void* ws, size_t ws_bytes, hipStream_t st,
                        teo_graph** out) {
    return capture_graph(st, [&]() { return llama_decode_step(d, s, ws, ws_bytes, st); }, out);
}

int decode_batch_graph_create(const teo_llama_desc* d, const teo_decode_batch_state* s, void* ws, size_t ws_bytes,
                              hipStream_t st, teo_graph** out) {
    return capture_graph(st, [&]() { return llama_decode_batch_step(d, s, ws, ws_bytes, st); }, out);
}

}  // namespace teo
