/*
 * teo_hip.h -- C ABI of libteo_hip.so: the MI355X (gfx950) compute library beneath the
 * videollava-compatible Python surface of teochat_amd.
 *
 * Boundary rules:
 *   - extern "C", plain pointers + sizes, no torch / C++ types in any signature;
 *   - every entry point returns 0 (TEO_OK) or a negative teo_status; it never throws.
 *     teo_last_error() returns a thread-local message for the last failure;
 *   - all pointers named d_* are DEVICE pointers owned by the caller (torch tensors on the
 *     Python side); the library allocates nothing on the device: composed entry points take a
 *     caller-provided workspace whose size comes from the matching *_workspace_bytes() query;
 *   - work is enqueued asynchronously on the caller's stream (hipStream_t passed as void*);
 *   - dtype arguments are teo_dtype; "T" below means the element type that dtype selects.
 *
 * Each entry point cites the reference interface it replaces (paths relative to the reference
 * repo; "tf" = transformers modelling files the reference delegates its arithmetic to,
 * pyproject.toml:17).
 */
#ifndef TEO_HIP_H
#define TEO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TEO_ABI_VERSION 2 /* 2 (round 5): teo_tune blocks instead of process-wide knobs, `tune` field at the end of the descriptors, teo_sizeof */

typedef void* teo_stream_t; /* hipStream_t */

typedef enum { TEO_F32 = 0, TEO_BF16 = 1, TEO_F16 = 2 /* IEEE binary16: the reference's inference dtype (builder.py:105) */ } teo_dtype;
typedef enum { TEO_ACT_NONE = 0, TEO_ACT_GELU_ERF = 1, TEO_ACT_QUICK_GELU = 2 } teo_act;
typedef enum {
    TEO_OK = 0,
    TEO_ERR_ARG = -1,         /* bad argument (null pointer, negative size, misaligned) */
    TEO_ERR_UNSUPPORTED = -2, /* shape / dtype combination not implemented */
    TEO_ERR_HIP = -3,         /* a HIP runtime call failed (message in teo_last_error) */
    TEO_ERR_WORKSPACE = -4    /* workspace too small */
} teo_status;

/* GEMM epilogue flags (teo_gemm `flags`) */
#define TEO_GEMM_SWIGLU16 1u /* W rows are [gate16|up16] interleaved; C[M, N/2] = silu(g) * u */
#define TEO_GEMM_FORCE_SIMPLE 2u /* use the shape-agnostic VALU kernel even when the MFMA kernel applies */
#define TEO_GEMM_SWIGLU8 8u /* teo_gemm_skinny: like SWIGLU16 with gate/up rows interleaved in blocks of 8 */
#define TEO_GEMM_F16 16u /* teo_gemm_skinny: x / W / residual are IEEE half (implied by a TEO_F16 output; needed with a TEO_F32 one) */
#define TEO_GEMM_WTILED 4u /* teo_gemm_skinny: W is stored as 1 KB operand tiles (16 rows x 32 k bf16 / 64 k fp8), see below */

int teo_version(void);
const char* teo_last_error(void);
/* Diagnostics: which kernel family the most recent teo_gemm* / teo_attention call of this thread dispatched to
 * ("gemm_simple", "gemm_mfma_128", "gemm_mfma_128_sk", "gemm_wide", "gemm_wide_sk", "gemm_big", "gemm_big_hybrid", "gemm_fp8_*",
 * "gemm_big_hybrid_cohort", "gemm_narrow_64", "gemm_narrow_128", "gemm_narrow_128w8", "gemm_pipe_64x64", "gemm_pipe_64", "gemm_pipe_64_r4", "gemm_pipe_128x96",
 * "gemm_pipe_128", "gemm_quad_160", "gemm_quad_160_w4", "attn_flash32", "attn_simple").  Lets the parity tests state which production kernel they checked. */
const char* teo_last_kernel(void);
/* Size of a struct of this header as the LIBRARY was built with it (0 for an unknown name): a binding checks its own layout against it
 * at load time -- "teo_vit_desc", "teo_proj_desc", "teo_llama_desc", "teo_decode_state", "teo_decode_batch_state", "teo_attn_args". */
size_t teo_sizeof(const char* struct_name);
/* Performance tuning knobs.  PERF-ONLY: every key selects among kernels / geometries that compute the same values (bit-identical
 * unless noted "fp32 order": the fp32 summation order of a reduction may change, nothing else).  Result- or path-selecting options
 * are NOT here: they are fields of the descriptors (teo_llama_desc.prefill_fp8, .rope_in_attn).
 * NO knob is process-wide state (no globals).  Knobs live in a teo_tune block that belongs to its creator
 * (teo_tune_create / teo_tune_destroy).  The block in effect for a call is
 *   1. the `tune` field of the descriptor for the descriptor-driven entry points (teo_vit_encode, teo_projector, teo_llama_*), when set
 *      -- two engines in one process carry two blocks and never see each other's choices;
 *   2. else the block the CALLING THREAD bound with teo_tune_bind (thread-local; the primitive operators have no descriptor);
 *   3. else the built-in defaults = what ships.
 * A block may be read by any number of threads; changing it while another thread runs a call under it is the caller's race.
 * teo_tune_destroy on a block still bound by ANOTHER thread or named by a live descriptor is a use-after-free of the caller's making.
 * teo_tune_get(NULL, key, &v) reads the shipped default; teo_tune_keys() lists every key, space separated.  Keys (0 / 1 unless said):
 *   decode GEMV   : "gemv_variant" (-1 default; 0..2, 10..13: row-group geometry; fp32 order), "gemv_nt" (non-temporal weight loads),
 *                   "gemv_max_blocks" (workgroup cap), "gemv_small_k" (x prologue sized to K <= 4096), "gemv_splitk_u" (chunks per thread and step of the split-K GEMV: 0 auto, 1/2/3/4/6), "gemv_splitk_r" (its rows per workgroup: 0 auto, 2/4)
 *   prefill GEMM  : "gemm_bm" (tile rows of the plain kernel: 0 auto, 64, 128), "gemm_depth", "gemm_sk" (stream-K: 0 off, 1 auto, 2 force),
 *                   "gemm_wide" (0 off, 1 auto, 2 force), "gemm_wide_sched", "gemm_wide_group", "gemm_big" (0 off, 1 auto, 2 force),
 *                   "gemm_big_group", "gemm_big_hybrid" (0 off, 1 auto, 2 force), "gemm_big_cohort" (stream-K part of the hybrid form as
 *                   XCD-local cohorts: -1 auto, 0 linear ranges, 8 / 16 / 32 workgroups per cohort), "gemm_big_ragged" (a last row block of <= 128
 *                   rows as 128 x 512 tiles instead of a padded 256-row tile: 0 never, 1 auto = where that makes the problem one round, 2 whenever
 *                   the shape allows), "gemm_narrow" (64 / 128 x 128 LDS-DMA
 *                   tiles for few-tile / short-K shapes: 0 off, 1 auto, 2 force), "gemm_narrow_bm" (its tile rows: 0 auto, 64, 128),
 *                   "gemm_narrow_waves" (the 128 x 128 tile on 4 waves or on 8: 0 auto, 4, 8),
 *                   "gemm_narrow_pipe" (the software-pipelined small tiles -- 64 x 64, 64 x 128, 128 x 96, 128 x 128, sized for one workgroup per
 *                   CU -- in place of the LDS-DMA tiles: 0 off, 1 auto (default), 2 with "gemm_narrow" = 2: the forced narrow tile runs in this
 *                   form), "gemm_pipe_bn" (forced form: tile columns, 0 = 128, 64 with 64 rows, 96 with 128 rows), "gemm_pipe_stages" (forced
 *                   form: ring depth, 0 = 3, 3, 4),
 *                   "gemm_quad" (256 x 160 tiles, hand-scheduled K loop with the accumulators in AGPRs, for problems that are one round of them: 0 off, 1 auto,
 *                   2 force), "gemm_quad_waves" (8: two waves per SIMD, the default; 4: one wave per SIMD with the whole register file), "gemm_fp8_wide" (0..3),
 *                   "gemm_fp8_big" (0..2) -- all bit-identical families
 *   rope / caches : "rope_vt_fused" (prefill: RoPE + K append and V / V^T append as one launch where the 16-byte paths apply: 0 two launches, 1 one -- default;
 *                   bit-identical)
 *   prefill attn  : "flash_order" (causal workgroup order of the flash kernel: 0 heavy-first, 1 second dispatch pass mirrored), "flash_pipe"
 *                   (software pipeline inside the wave: -1 auto = causal kernels, 0 one tile at a time, 1 wherever the form exists) --
 *                   same tiles, same arithmetic: bit-identical
 *   decode attn   : "attn_chunk" (keys per decode chunk: 0 auto, 32/64/128/256; fp32 order of the split merge + where P is rounded),
 *                   "attn_whole" (batched decode attention as one workgroup per (conversation, head): 0 off, 1 auto = when conversations x
 *                   heads make whole rounds of the CUs (a multiple of the CU count, or at least 7/4 rounds), 2 whenever the shape allows; bit-identical to the split + combine pair at the same chunk)
 *   batched GEMM  : "skinny_tiles" (0 auto, 1/2/4/8), "skinny_nt", "skinny_stream" (0 off, 1 auto, 2 whenever eligible), "skinny_ring"
 *                   (weight tiles in flight of the streaming form: 0 default, 1 one more), "skinny_unr" (tile kernel steps per register
 *                   set: 0 auto, 4, 8), "skinny_waves" (tile kernel waves per workgroup: 0 auto = 8, 8, 16 -- measured and lost in round 6,
 *                   kept for the A/B), "skinny_grid" (persistent workgroups per CU of the streaming form: 0 auto = 1, 1..3 -- likewise) --
 *                   bit-identical at K = 4096, fp32 order elsewhere
 * teo_tune_set returns TEO_ERR_ARG for an unknown key or a value outside the key's set (message in teo_last_error). */
typedef struct teo_tune teo_tune;
teo_tune* teo_tune_create(void);                      /* a block holding the shipped defaults; NULL when out of memory */
int teo_tune_destroy(teo_tune* tune);                 /* NULL is fine; unbinds it from the calling thread first */
int teo_tune_set(teo_tune* tune, const char* key, int value);
int teo_tune_get(const teo_tune* tune, const char* key, int* value);
int teo_tune_reset(teo_tune* tune);                   /* every key back to the shipped default */
int teo_tune_bind(const teo_tune* tune);              /* the calling thread's block for calls without a descriptor block; NULL unbinds */
const char* teo_tune_keys(void);
/* 1 when the MFMA (fast) kernel would be used for this GEMM, 0 when the generic kernel would. */
int teo_gemm_uses_mfma(int M, int N, int K, int dtype, unsigned flags);

/* ---------------------------------------------------------------------------------------------
 * Primitive operators (each one parity-tested on its own against oracle/teo_oracle.py)
 * ------------------------------------------------------------------------------------------- */

/* y[r,:] = LayerNorm(x[r,:]) * w + b.   Replaces nn.LayerNorm at
 * languagebind/image/modeling_image.py:70,72 (layer_norm1/2) and :601 (pre_layrnorm). */
int teo_layernorm(const void* d_x, const void* d_w, const void* d_b, void* d_y, int rows, int dim, float eps,
                  int dtype, teo_stream_t stream);

/* y[r,:] = x[r,:] * rsqrt(mean(x^2) + eps) * w.   Replaces tf llama LlamaRMSNorm.forward
 * (constructed by LlamaModel, reached from language_model/llava_llama.py:88-99). */
int teo_rmsnorm(const void* d_x, const void* d_w, void* d_y, int rows, int dim, float eps, int dtype,
                teo_stream_t stream);

/* C[M,N] = act(A[M,K] . W[N,K]^T + bias[N]) + residual[M,N]   (bias/residual may be NULL).
 * A rows have stride lda, C/residual rows stride ldc (elements).  out_dtype may be TEO_F32 with
 * bf16 inputs (logits).  Replaces every nn.Linear on the path: tf CLIPAttention q/k/v/out_proj,
 * CLIPMLP fc1/fc2 (constructed at modeling_image.py:69,71), the projector Linear layers
 * (multimodal_projector/builder.py:42-46), tf LlamaAttention q/k/v/o_proj, LlamaMLP, lm_head. */
int teo_gemm(const void* d_A, const void* d_W, const void* d_bias, const void* d_residual, void* d_C, int M,
             int N, int K, int lda, int ldc, int act, unsigned flags, int dtype, int out_dtype,
             teo_stream_t stream);

/* teo_gemm with a scratch workspace (teo_gemm_workspace_bytes() bytes, 256-byte aligned): lets the library run the
 * stream-K form of the MFMA kernel when the static 128 x 128 tiling would leave a ragged last round on the 256 CUs (e.g.
 * M = 2168, N = 4096: 544 tiles on 512 slots).  Results are bit-identical to teo_gemm (same k-order per output element).
 * teo_gemm_workspace_init must run once on a fresh workspace (it zeroes the hand-off flags; the kernels re-arm them). */
size_t teo_gemm_workspace_bytes(void);
int teo_gemm_workspace_init(void* d_workspace, teo_stream_t stream);
/* Contract of a workspace: ONE stream at a time (two streams running teo_gemm_ws / teo_gemm_fp8_ws / teo_vit_encode / teo_llama_prefill
 * on the same workspace would interleave its hand-off flags and slabs).  The persistent forms are sized for the MI355X's 256 CUs; on any
 * other CU count the workspace is ignored and the plain kernels run.  A hand-off that ever timed out (never observed; the producer of
 * a slab is resident before its consumer by construction) sets a STICKY error word instead of continuing silently:
 * teo_gemm_workspace_status copies it to *host_flag after synchronising the stream (0 = fine, 1 = results since the last
 * teo_gemm_workspace_init are invalid). */
int teo_gemm_workspace_status(const void* d_workspace, int* host_flag, teo_stream_t stream);
int teo_gemm_ws(const void* d_A, const void* d_W, const void* d_bias, const void* d_residual, void* d_C, int M, int N, int K,
                int lda, int ldc, int act, unsigned flags, int dtype, int out_dtype, void* d_workspace, teo_stream_t stream);

/* w8a8 GEMM on the block-scaled fp8 MFMA of CDNA4 (v_mfma_scale_f32_16x16x128_f8f6f4, every block scale 2^0; config C5's "fp8
 * weight path on CDNA4 MFMA" for the prefill phase):
 *     C[m, n] = a_scale[m] * w_scale[n] * sum_k A8[m, k] * W8[n, k]  (+ residual[m, n]);  TEO_GEMM_SWIGLU16 as in teo_gemm.
 * A8 [M, lda] and W8 [N, K] hold OCP e4m3 bytes, a_scale [M] / w_scale [N] fp32; C / residual bf16 (or C fp32).  K % 128 == 0,
 * lda % 16 == 0, N % 4 == 0; TEO_ERR_UNSUPPORTED otherwise.  Same Linear layers as teo_gemm (tf LlamaAttention / LlamaMLP,
 * reached from llava_llama.py:88-99) under a weight + activation quantisation the reference does not have (its 8-bit option is
 * bitsandbytes, eval.py:52-53): selectable, never the default.
 * teo_quant_rows_fp8: per-row (per-token) quantiser  q[m, :] = e4m3(f(x[m, :]) / s[m]),  s[m] = max|f(x[m, :])| / 448, with
 * f = identity, or LlamaRMSNorm (d_norm_w != NULL: x * rsqrt(mean(x^2) + eps) * w rounded to bf16, as teo_rmsnorm).  x bf16. */
int teo_gemm_fp8(const void* d_A8, const float* d_a_scale, const void* d_W8, const float* d_w_scale, const void* d_residual, void* d_C,
                 int M, int N, int K, int lda, int ldc, unsigned flags, int out_dtype, teo_stream_t stream);
/* teo_gemm_fp8 with the teo_gemm_ws workspace: stream-K form of the wide fp8 kernel for the one-round-plus shapes; bit-identical. */
int teo_gemm_fp8_ws(const void* d_A8, const float* d_a_scale, const void* d_W8, const float* d_w_scale, const void* d_residual, void* d_C,
                    int M, int N, int K, int lda, int ldc, unsigned flags, int out_dtype, void* d_workspace, teo_stream_t stream);
int teo_quant_rows_fp8(const void* d_x, const void* d_norm_w, void* d_q, float* d_scale, int rows, int K, int ldx, float eps,
                       teo_stream_t stream);

/* Patch extraction for the CLIP patch-embedding conv (kernel = stride = patch, no bias):
 * cols[t*g*g + py*g + px, c*P*P + ky*P + kx] = pixels[t, c, py*P+ky, px*P+kx], zero padded to ldcols.
 * Replaces the im2col half of CLIPVisionEmbeddings.patch_embedding (used at modeling_image.py:602,645). */
/* Fused form of the same convolution (bf16): out[t*g*g + py*g + px, n] = sum_k W[n, k] * pixel(t, k), k = c*P*P + ky*P + kx, the
 * patch pixels gathered straight into the LDS operand image of the MFMA tile (no im2col matrix).  W [dim, ldw] with the columns
 * k >= channels*P*P zero (ldw % 64 == 0).  Bit-identical to teo_im2col_patches + teo_gemm.  Replaces
 * CLIPVisionEmbeddings.patch_embedding (modeling_image.py:602,645). */
int teo_patch_embed(const void* d_pixels, const void* d_weight, void* d_out, int T, int channels, int image, int patch, int ldw,
                    int dim, int dtype, teo_stream_t stream);
int teo_im2col_patches(const void* d_pixels, void* d_cols, int T, int channels, int image, int patch, int ldcols,
                       int dtype, teo_stream_t stream);

/* hidden0[t, 0, :] = LN(cls + pos[0]); hidden0[t, 1+p, :] = LN(patch[t*NP + p] + pos[1+p]).
 * Replaces the cat/+position_embedding of CLIPVisionEmbeddings.forward and pre_layrnorm
 * (modeling_image.py:645-649). */
int teo_vit_embed_ln(const void* d_patch, const void* d_cls, const void* d_pos, const void* d_w, const void* d_b,
                     void* d_out, int T, int n_patches, int dim, float eps, int dtype, teo_stream_t stream);

/* Attention over explicitly strided operands (element strides):
 *   Q[b][h][i][:]  at q  + b*q_bs  + h*q_hs  + i*q_rs          (i < q_len, head_dim contiguous)
 *   K[b][hk][j][:] at k  + b*k_bs  + hk*k_hs + j*k_rs          (j < kv_len)
 *   V[b][hk][j][:] at v  + b*v_bs  + hk*v_hs + j*v_rs          (row-major values)
 *   VT[b][hk][:][j] at vt + b*vt_bs + hk*vt_hs + d*vt_rs + j   (transposed values; may be NULL)
 *   O[b][i][h*head_dim + :] at o + b*o_bs + i*o_rs
 * out = softmax(scale * Q K^T + causal mask) V;  causal: key j visible to query i iff j <= i + (kv_len - q_len).
 * hk = h / (heads / kv_heads).  The MFMA kernel needs VT, bf16 and head_dim in {64,128}; otherwise the
 * generic kernel runs (reads V).  Replaces tf CLIPAttention / LlamaAttention eager attention. */
typedef struct {
    const void* q; const void* k; const void* v; const void* vt; void* o;
    long long q_bs, q_hs, q_rs;
    long long k_bs, k_hs, k_rs;
    long long v_bs, v_hs, v_rs;
    long long vt_bs, vt_hs, vt_rs;
    long long o_bs, o_rs;
    int batch, heads, kv_heads, head_dim, q_len, kv_len;
    int causal;
    float scale;
    unsigned flags; /* TEO_ATTN_FORCE_SIMPLE */
} teo_attn_args;
#define TEO_ATTN_FORCE_SIMPLE 1u
int teo_attention(const teo_attn_args* args, int dtype, teo_stream_t stream);

/* ViT helper: VT[t][h][d][j] = qkv[t*N + j][2*D + h*hd + d], rows padded with zeros to ldv.
 * (layout change only; lets the MFMA attention kernel read key-contiguous values.) */
int teo_vit_value_transpose(const void* d_qkv, void* d_vt, int T, int N, int heads, int head_dim, int ldv,
                            int dtype, teo_stream_t stream);

/* RoPE + KV append for S new positions of one sequence (LLaMA prefill or decode):
 *   qkv[s, :] = [q (H*hd) | k (Hk*hd) | v (Hk*hd)] rows with stride ld_qkv; q is rotated IN PLACE;
 *   k rotated -> K cache [Hk][S_max][hd] at position past+s; v -> V cache (same layout) and,
 *   when d_vt_cache != NULL, VT cache [Hk][hd][S_max].
 *   cos/sin tables: fp32 [max_pos][hd/2] (built on the host exactly as tf LlamaRotaryEmbedding does);
 *   positions: int32 [S] (position_ids of the new tokens).
 * Replaces tf apply_rotary_pos_emb + the KV-cache torch.cat in LlamaAttention.forward. */
int teo_rope_kv_append(void* d_qkv, int ld_qkv, const int* d_positions, const float* d_cos, const float* d_sin,
                       void* d_k_cache, void* d_v_cache, void* d_vt_cache, int S, int past, int S_max, int heads,
                       int kv_heads, int head_dim, int dtype, teo_stream_t stream);

/* Embedding splice (the data movement of prepare_inputs_labels_for_multimodal, llava_arch.py:254-293):
 * out[r,:] = plan[r] >= 0 ? embed[plan[r],:] : (plan[r] == INT32_MIN ? 0 : visual[-plan[r]-1,:]).
 * The int32 plan is built on the host by the bit-exact index logic. */
int teo_embed_splice(const int* d_plan, const void* d_embed, const void* d_visual, void* d_out, int rows, int dim,
                     int dtype, teo_stream_t stream);

/* Decode attention (one new query row per conversation; tf llama eager attention with a KV cache, the H15 row of
 * SURVEY.md section 8a at q_len == 1).  Conversation b < batch: query row at d_q + b*q_stride, caches at + b*cache_stride
 * (elements; K and V are [kv_heads][max_seq][head_dim], V^T [kv_heads][head_dim][max_seq]), position d_pos[b] (= number
 * of cached tokens; attends over d_pos[b] + 1 keys), output [heads*head_dim] at d_out + b*o_stride.
 *   rope_cos == NULL : d_q holds the rotated query [heads*head_dim]; the caches already contain the new token's K/V.
 *   rope_cos != NULL : d_q holds the raw [q | k | v] row of the QKV projection; the kernel applies RoPE at d_pos[b] to q
 *                      and k and appends k, v (and V^T when d_vt_cache != NULL) to the caches before attending.
 * KV-split partials + combine; d_partials is teo_attn_decode_workspace_bytes(...) of scratch. */
size_t teo_attn_decode_workspace_bytes(int heads, int head_dim, int max_seq, int batch);
int teo_attn_decode(const void* d_q, void* d_k_cache, void* d_v_cache, void* d_vt_cache, const float* d_rope_cos,
                    const float* d_rope_sin, void* d_out, float* d_partials, const int* d_pos, int max_seq, int heads,
                    int kv_heads, int head_dim, float scale, int dtype, int batch, long long q_stride, long long cache_stride,
                    long long o_stride, teo_stream_t stream);

/* Training-shape loss: mean over rows with label != ignore_index of (logsumexp(logits[r]) - logits[r, label[r]]) --
 * torch.nn.CrossEntropyLoss as LlamaForCausalLM.forward applies it to the shifted logits/labels (call site
 * videollava/model/language_model/llava_llama.py:88-99).  logits fp32 [rows, ld]; labels int64 [rows] (already shifted
 * by the caller); loss_row fp32 [rows] scratch/output (0 on ignored rows); out fp32 [3] = {mean (nan if nothing is
 * supervised), sum, count}. */
int teo_cross_entropy(const float* d_logits, long long ld, const long long* d_labels, float* d_loss_row, float* d_out, int rows,
                      int vocab, long long ignore_index, teo_stream_t stream);

/* Image preprocessing on the device (languagebind/image/processing_image.py:15-25 get_image_transform):
 * src uint8 [T, H, W, 3] (RGB, as PIL gives it) -> out [T, 3, S, S] = Normalize(CenterCrop(S)(Resize(S, bicubic,
 * antialias)(ToTensor(src))), mean, std) in `dtype`; mean/std are HOST pointers to 3 floats.  The resampling is ATen's
 * anti-aliased bicubic (a = -0.5, align_corners = False); H == W == S is the identity. */
int teo_preprocess_frames(const unsigned char* d_src, void* d_out, int T, int H, int W, int S, const float* mean,
                          const float* std, int dtype, teo_stream_t stream);
/* The same behind expand2square (videollava/mm_utils.py:14-25,28-36: `image_aspect_ratio == 'pad'`): the frame is first pasted,
 * centred (offset (side - n) // 2), on a square canvas of its longer side filled with pad_rgb (HOST pointer to 3 bytes, the
 * reference passes tuple(int(m * 255) for m in image_mean)); the canvas is never materialised. */
int teo_preprocess_frames_pad(const unsigned char* d_src, void* d_out, int T, int H, int W, int S, const float* mean,
                              const float* std, const unsigned char* pad_rgb, int dtype, teo_stream_t stream);

/* out[t, p, :] = in[t, 1+p, :]  (feature_select 'patch', languagebind/__init__.py:121-129) */
int teo_drop_cls(const void* d_in, void* d_out, int T, int n_tokens, int dim, int dtype, teo_stream_t stream);

/* token[r] = argmax_j logits[r, j] (first index on ties, like torch.argmax); int64 out.
 * Replaces the greedy branch of GenerationMixin (call at eval/inference.py:64-72 with do_sample=False). */
int teo_argmax(const float* d_logits, long long* d_token, int rows, int vocab, teo_stream_t stream);

/* token = multinomial(softmax(top_p(top_k(logits / temperature)))) with u = uniform(seed, draw).  Same filter order as HF's
 * TemperatureLogitsWarper -> TopKLogitsWarper -> softmax -> multinomial (the sampled branch the reference uses,
 * eval/inference.py:64-72 with do_sample=True).  top_k <= 0 or >= vocab disables the top-k filter (multinomial over the whole
 * vocabulary); 0 < top_k <= 1024 otherwise (larger: TEO_ERR_UNSUPPORTED).  top_p outside (0, 1) disables the nucleus filter
 * (HF TopPLogitsWarper: ascending cumulative probability <= 1 - top_p is dropped, the most probable token always stays);
 * it acts on the top-k survivors and needs the top-k filter on when vocab > 1024 (else TEO_ERR_UNSUPPORTED -- never a
 * silently truncated distribution). */
int teo_sample_topk(const float* d_logits, long long* d_token, int vocab, float temperature, int top_k, float top_p,
                    unsigned long long seed, unsigned long long draw, teo_stream_t stream);

/* Decode GEMV: y[N] = W[N,K] . f(x) (+ residual), x one row.
 *   norm_w != NULL : f(x) = rmsnorm(x) * norm_w (rounded to dtype), else f(x) = x
 *   flags & TEO_GEMM_SWIGLU16 : W is gate/up interleaved-16; y[N/2] = silu(g)*u
 *   out_dtype TEO_F32 allowed (logits).
 * Same Linear layers as teo_gemm, for the q_len == 1 decode step (HBM-bound, no MFMA). */
int teo_gemv(const void* d_x, const void* d_W, const void* d_norm_w, const void* d_residual, void* d_y, int N, int K,
             float eps, unsigned flags, int dtype, int out_dtype, teo_stream_t stream);

/* teo_gemv with fp8 e4m3 weights: W8 [N,K] bytes, w_scale [N] fp32 (y = scale * (W8 . f(x))); bf16 activations only. */
int teo_gemv_w8(const void* d_x, const void* d_W8, const float* d_w_scale, const void* d_norm_w, const void* d_residual,
                void* d_y, int N, int K, float eps, unsigned flags, int out_dtype, teo_stream_t stream);

/* Batched-decode GEMM: out[b, n] = sum_k x[b, k] * W[n, k] (+ residual[b, n]) for MB <= 16 conversations; the weights
 * are streamed once for the whole batch (HF generate with batch > 1 through LlamaForCausalLM.forward,
 * videollava/model/language_model/llava_llama.py:88-99).  bf16 activations x [MB, ldx]; W bf16 [N, K] or fp8 e4m3
 * (w_fp8 = 1) with w_scale [N]; out bf16 / f32 [MB, ldo]; TEO_GEMM_SWIGLU16 as in teo_gemm (out [MB, N/2]).
 * K % 32 == 0 (64 for fp8); returns TEO_ERR_UNSUPPORTED otherwise.
 * d_norm_w != NULL fuses LlamaRMSNorm of the rows: out = rsqrt(mean(x^2) + eps) * (W . bf16(x * norm_w)) -- the row
 * factor commutes with the product, so x is read once and no normalised copy exists (rounding: x*g once to bf16,
 * the factor applied in fp32; HF rounds x/rms first: same order of error).
 * TEO_GEMM_WTILED: W holds ceil(N/16) * (K/KS) tiles of 1 KB (KS = 32 k for bf16, 64 k for fp8; rows past N zero);
 * tile (n/16, k/KS) starts at ((n/16) * (K/KS) + k/KS) KB and element (n, k) sits in it at 16-byte lane
 * ((k % KS) / CH) * 16 + n % 16, position k % CH (CH = 8 bf16 / 16 fp8 per lane) -- the v_mfma_f32_16x16x32_bf16
 * operand order, so every wave load is 1 KB contiguous. */
int teo_gemm_skinny(const void* d_x, const void* d_W, const float* d_w_scale, int w_fp8, const void* d_norm_w, float eps,
                    const void* d_residual, void* d_out, int MB, int N, int K, int ldx, int ldo, unsigned flags, int out_dtype,
                    teo_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Composed runtime entry points (the layer loops live in C++, not Python)
 * ------------------------------------------------------------------------------------------- */
typedef struct {
    int hidden, heads, inter, layers_run; /* layers_run = number of encoder layers actually executed */
    int image, patch, channels;
    int act;        /* teo_act of CLIPMLP */
    float eps;
    int dtype;
    int k_pad;      /* padded K of the patch GEMM (multiple of 64) */
    const void* patch_w;  /* [hidden, k_pad] (zero padded) */
    const void* cls;      /* [hidden] */
    const void* pos;      /* [n_patches+1, hidden] */
    const void* pre_ln_w; const void* pre_ln_b;
    /* per layer arrays of device pointers (host arrays of length layers_run) */
    const void* const* ln1_w; const void* const* ln1_b;
    const void* const* qkv_w; const void* const* qkv_b;   /* fused [3*hidden, hidden], [3*hidden] */
    const void* const* out_w; const void* const* out_b;
    const void* const* ln2_w; const void* const* ln2_b;
    const void* const* fc1_w; const void* const* fc1_b;
    const void* const* fc2_w; const void* const* fc2_b;
    int keep_cls;   /* feature_select (languagebind/__init__.py:121-129): 0 = 'patch' (drop the CLS row), 1 = 'cls_patch' */
    const teo_tune* tune; /* performance knobs of this engine (NULL: the calling thread's bound block, else the defaults) */
} teo_vit_desc;

size_t teo_vit_workspace_bytes(const teo_vit_desc* d, int T);
/* features[T, n_patches, hidden] = hidden_states[layers_run][:, 1:]  (H7-H11 of SURVEY.md section 8a):
 * LanguageBindImageTower.forward + feature_select (languagebind/__init__.py:121-146).  With d->keep_cls the CLS row
 * stays: features[T, n_patches + 1, hidden] = hidden_states[layers_run]  ('cls_patch'). */
int teo_vit_encode(const teo_vit_desc* d, const void* d_pixels, int T, void* d_features, void* d_workspace,
                   size_t workspace_bytes, teo_stream_t stream);

typedef struct {
    int in_dim, out_dim, depth; /* depth 1 = linear, 2 = mlp2x_gelu */
    int dtype;
    const void* w[4]; const void* b[4];
    const teo_tune* tune; /* as teo_vit_desc.tune */
} teo_proj_desc;
size_t teo_projector_workspace_bytes(const teo_proj_desc* d, int rows);
/* build_vision_projector (multimodal_projector/builder.py:33-51) applied to [rows, in_dim]. */
int teo_projector(const teo_proj_desc* d, const void* d_x, int rows, void* d_y, void* d_workspace,
                  size_t workspace_bytes, teo_stream_t stream);

typedef struct {
    int hidden, heads, kv_heads, head_dim, inter, layers, vocab;
    float eps;
    int dtype;
    int max_seq;           /* S_max of the KV cache */
    const void* embed;     /* [vocab, hidden] */
    const void* final_norm_w;
    const void* lm_head;   /* [vocab, hidden] */
    const float* rope_cos; const float* rope_sin;   /* [max_pos, head_dim/2] fp32 */
    int max_pos;
    const void* const* in_norm_w;
    const void* const* qkv_w;      /* fused [(heads+2*kv_heads)*head_dim, hidden] */
    const void* const* o_w;
    const void* const* post_norm_w;
    const void* const* gateup_w;   /* [2*inter, hidden], gate/up interleaved in blocks of 16 rows */
    const void* const* down_w;
    /* KV cache, per layer device pointers */
    void* const* k_cache;  /* [kv_heads][max_seq][head_dim] */
    void* const* v_cache;  /* [kv_heads][max_seq][head_dim] */
    void* const* vt_cache; /* [kv_heads][head_dim][max_seq] */
    /* Optional fp8 (OCP e4m3fn) copies of the Linear weights for the DECODE path (config C5): same row layouts as
     * above, one fp32 scale per output row (W = fp8 * scale; power-of-two scales make the bf16 weights above, used by
     * prefill, exactly equal to the dequantised fp8 weights).  All NULL -> decode streams the bf16/f32 weights. */
    const void* const* qkv_w8;    const float* const* qkv_s;
    const void* const* o_w8;      const float* const* o_s;
    const void* const* gateup_w8; const float* const* gateup_s;
    const void* const* down_w8;   const float* const* down_s;
    const void* lm_head8;         const float* lm_head_s;
    /* Per-engine options (they select a path or change results, so they live here and not in teo_tune_set): */
    int prefill_fp8;   /* 1: prefill Linear layers as w8a8 on the fp8 MFMA (activations quantised per token; needs the *_w8 copies,
                        *    bf16).  Lossy beyond the weight quantisation: selectable, never a default.  0: bf16 / f32 GEMMs */
    int rope_in_attn;  /* single-conversation decode step: 0 = RoPE + KV append in the QKV GEMV epilogue (default), 1 = inside the
                        *    decode attention kernel (same values; the batched step always uses 1) */
    const teo_tune* tune; /* performance knobs of this engine (as teo_vit_desc.tune); a captured decode graph keeps the choices made at capture */
} teo_llama_desc;

size_t teo_llama_prefill_workspace_bytes(const teo_llama_desc* d, int S);
/* LlamaModel.forward + lm_head over S new positions given inputs_embeds (llava_llama.py:88-99).
 *   d_embeds [S, hidden] (dtype), d_positions int32 [S], past = tokens already in the cache.
 *   logits_rows: 0 -> logits for all S positions [S, vocab] fp32; 1 -> last position only [1, vocab].
 *   d_hidden_states (may be NULL): `output_hidden_states` of the kept forward signature (llava_llama.py:56-69, 88-99) --
 *   [layers + 1][S][hidden] in the model dtype: snapshot 0 = the input embeddings, l = the residual stream after layer l - 1, the last
 *   one after the final RMSNorm (what LlamaModel.forward collects: lm_head(hidden_states[-1]) == logits). */
int teo_llama_prefill(const teo_llama_desc* d, const void* d_embeds, const int* d_positions, int S, int past,
                      int last_only, float* d_logits, void* d_workspace, size_t workspace_bytes,
                      teo_stream_t stream, void* d_hidden_states);
/* teo_llama_prefill that also returns the attention maps: `output_attentions` of the kept forward signature (llava_llama.py:65,95 ->
 * LlamaAttention's eager softmax).  d_attentions: [layers][heads][S][past + S] in the model dtype -- softmax over the visible (causal)
 * keys of scale * q k^T from the rotated q and the cached k the attention kernel itself reads, statistics in fp32, one rounding, exact
 * zeros for masked keys.  The fused attention kernels never materialise these maps; a plain kernel writes them beside the unchanged
 * forward (logits, cache and hidden states are the ones teo_llama_prefill produces).  Not a performance path: layers x heads x S x (past + S)
 * elements (C3: 32 x 32 x 2168 x 2168 x 2 B = 9.6 GB). */
int teo_llama_prefill_attentions(const teo_llama_desc* d, const void* d_embeds, const int* d_positions, int S, int past,
                                 int last_only, float* d_logits, void* d_workspace, size_t workspace_bytes, teo_stream_t stream,
                                 void* d_hidden_states, void* d_attentions);

/* Persistent decode state: greedy decode of one sequence with everything (token, position, stop flag) on the device. */
typedef struct {
    long long* d_token;      /* [1] current token id (input of the step; overwritten with the next token) */
    int* d_pos;              /* [1] position of d_token == number of tokens in the cache */
    long long* d_out_tokens; /* [max_new] generated ids, written at d_out_count */
    int* d_out_count;        /* [1] */
    int* d_stop;             /* [1] set to 1 when the generated tail equals d_stop_ids (id-suffix match) */
    const long long* d_stop_ids; int n_stop_ids; /* may be NULL/0 */
    float* d_logits;         /* [vocab] fp32 logits of the last step */
    /* sampling (N1): do_sample = 0 -> argmax.  Otherwise logits/temperature -> top-k -> softmax -> multinomial with a
     * counter-based generator: draw i of a generation uses (seed, i).  d_rng = {seed, draws so far} on the device. */
    int do_sample; int top_k; float temperature;
    unsigned long long* d_rng; /* [2] */
    float top_p;             /* nucleus filter after top-k; outside (0, 1) = off */
} teo_decode_state;

/* Prefill of nseq NEW conversations at once (batched generate): d_embeds holds the spliced embedding rows of all
 * sequences back to back ([sum(seq_lens), hidden]); seq_lens is a HOST array.  Norms and GEMMs run over all rows, RoPE /
 * KV append / causal attention per sequence into cache slot b (= the descriptor's cache pointers + b*cache_stride
 * elements).  last_only = 1: d_logits [nseq, vocab] fp32, last position of each sequence (batched generate); last_only = 0:
 * d_logits [sum(seq_lens), vocab], every row (the training-shape forward with B > 1, llava_llama.py:88-99 called from
 * train.py:840-901).  Workspace: teo_llama_prefill_workspace_bytes(d, sum(seq_lens)).  Row for row the results equal
 * teo_llama_prefill's. */
int teo_llama_prefill_batch(const teo_llama_desc* d, const void* d_embeds, const int* seq_lens, int nseq, long long cache_stride,
                            int last_only, float* d_logits, void* d_workspace, size_t workspace_bytes, teo_stream_t stream,
                            void* d_hidden_states /* NULL or [layers + 1][sum(seq_lens)][hidden], as teo_llama_prefill */);

size_t teo_llama_decode_workspace_bytes(const teo_llama_desc* d);
/* Arm a generation: workspace.h <- embed[*d_token] (call once after filling d_token/d_pos; every step's tail then
 * prepares the next step's embedding itself). */
int teo_llama_decode_begin(const teo_llama_desc* d, const teo_decode_state* st, void* d_workspace,
                           size_t workspace_bytes, teo_stream_t stream);

/* One greedy decode step (32 layers -> lm_head -> argmax -> append -> next embedding), all on device.
 * Replaces one iteration of GenerationMixin's loop around LlavaLlamaForCausalLM.forward with
 * input_ids [1,1] (llava_arch.py:154-163 decode branch; position = past length). */
int teo_llama_decode_step(const teo_llama_desc* d, const teo_decode_state* st, void* d_workspace,
                          size_t workspace_bytes, teo_stream_t stream);

/* hipGraph form of the same step: capture once, replay per token. */
typedef struct teo_graph teo_graph;
/* The sticky hand-off error word (teo_gemm_workspace_status) of the GEMM workspace carved inside a teo_vit_encode /
 * teo_llama_prefill (_batch: pass the total row count) workspace. */
int teo_vit_workspace_status(const teo_vit_desc* d, int T, void* d_workspace, size_t workspace_bytes, int* host_flag, teo_stream_t stream);
int teo_llama_prefill_workspace_status(const teo_llama_desc* d, int S, void* d_workspace, size_t workspace_bytes, int* host_flag,
                                       teo_stream_t stream);
/* Measurement aid: ONE decode step (plain launches, not a graph replay) in which every kernel launch carries its own start / stop
 * events (hipExtLaunchKernel: the dispatch's execution timestamps -- kernel time only, what rocprofv3 --kernel-trace reports).
 * ms_out[c] = summed kernel milliseconds of class c over the step, count_out[c] = launches of that class.  Same arithmetic and
 * state changes as teo_llama_decode_step (it IS that step); synchronises the stream before returning. */
enum { TEO_PROF_QKV = 0, TEO_PROF_ATTN = 1, TEO_PROF_ATTN_COMBINE = 2, TEO_PROF_O = 3, TEO_PROF_GATEUP = 4, TEO_PROF_DOWN = 5,
       TEO_PROF_LM_HEAD = 6, TEO_PROF_TAIL = 7, TEO_PROF_CLASSES = 8 };
int teo_llama_decode_step_profile(const teo_llama_desc* d, const teo_decode_state* st, void* d_workspace, size_t workspace_bytes,
                                  float* ms_out /* [TEO_PROF_CLASSES] */, int* count_out /* [TEO_PROF_CLASSES] */,
                                  teo_stream_t stream);
int teo_llama_decode_graph_create(const teo_llama_desc* d, const teo_decode_state* st, void* d_workspace,
                                  size_t workspace_bytes, teo_stream_t stream, teo_graph** out);
int teo_graph_launch(teo_graph* g, int n_times, teo_stream_t stream);
int teo_graph_destroy(teo_graph* g);

/* ---- batched decode (config C5's B conversations per GPU; HF generate with batch > 1) -------------------------
 * B <= TEO_MAX_DECODE_BATCH conversations advance one token per step.  Every weight matrix is streamed from HBM once
 * per step (teo_gemm_skinny), each conversation attends over its own KV cache at its own position, and the tail
 * (argmax / sampler, append, stop test, next embedding) runs per conversation.  The descriptor is the one used for
 * prefill except that k_cache/v_cache/vt_cache[l] point at conversation 0 of a [B][...] allocation and
 * `cache_stride` (elements) separates consecutive conversations; with w_tiled = 1 its weight matrices (qkv/o/gateup/
 * down/lm_head, bf16 or fp8) are in the TEO_GEMM_WTILED layout.  Finished conversations keep stepping (their d_stop
 * is set; the host truncates), exactly like the single-conversation loop. */
#define TEO_MAX_DECODE_BATCH 16
typedef struct {
    int batch;                /* 1..TEO_MAX_DECODE_BATCH */
    int out_stride;           /* d_out_tokens is [batch][out_stride] */
    long long cache_stride;   /* elements between conversations in each layer's K, V and V^T cache */
    int w_tiled;              /* 1: the descriptor's decode weight matrices are TEO_GEMM_WTILED */
    int gateup_block8;        /* 1: the descriptor's gate/up matrices (and their fp8 scales) interleave gate/up rows in blocks
                               * of 8 (TEO_GEMM_SWIGLU8) instead of 16 -- one row tile per workgroup also for the SwiGLU GEMM */
    long long* d_token;       /* [batch] */
    int* d_pos;               /* [batch] */
    long long* d_out_tokens;  /* [batch][out_stride] */
    int* d_out_count;         /* [batch] */
    int* d_stop;              /* [batch] */
    const long long* d_stop_ids; int n_stop_ids; /* shared by the batch; may be NULL/0 */
    float* d_logits;          /* [batch][vocab] */
    int do_sample; int top_k; float temperature;
    unsigned long long* d_rng; /* [batch][2] = {seed, draws so far} per conversation */
    float top_p;              /* nucleus filter after top-k; outside (0, 1) = off */
} teo_decode_batch_state;
size_t teo_llama_decode_batch_workspace_bytes(const teo_llama_desc* d, int batch);
int teo_llama_decode_batch_begin(const teo_llama_desc* d, const teo_decode_batch_state* st, void* d_workspace,
                                 size_t workspace_bytes, teo_stream_t stream);
int teo_llama_decode_batch_step(const teo_llama_desc* d, const teo_decode_batch_state* st, void* d_workspace,
                                size_t workspace_bytes, teo_stream_t stream);
/* Measurement aid, the batched counterpart of teo_llama_decode_step_profile: ONE batched step with every launch timed by its own
 * dispatch timestamps; the same TEO_PROF_* classes (the attention class covers the whole-context kernel, or the split kernel with
 * its combine in TEO_PROF_ATTN_COMBINE). */
int teo_llama_decode_batch_step_profile(const teo_llama_desc* d, const teo_decode_batch_state* st, void* d_workspace, size_t workspace_bytes,
                                        float* ms_out /* [TEO_PROF_CLASSES] */, int* count_out /* [TEO_PROF_CLASSES] */, teo_stream_t stream);
int teo_llama_decode_batch_graph_create(const teo_llama_desc* d, const teo_decode_batch_state* st, void* d_workspace,
                                        size_t workspace_bytes, teo_stream_t stream, teo_graph** out);

/* ---- multi-GPU context and the one collective of the path ---------------------------
 * One process per GPU.  The T-frame tower shards over the ranks (frames are independent through the tower: T is the batch
 * dimension, languagebind/image/modeling_image.py:641-643); rank r encodes a contiguous block of frames and ONE all-gather
 * of the visual tokens (tower width, before the projector) rebuilds the full, chronologically ordered tensor on every rank.
 * The reference has no counterpart (single GPU: scripts/eval_teochat.sh:9-10); it stacks all frames on one device
 * (llava_arch.py:194).  teo_ctx is opaque: it holds the rank, the device properties and the RCCL communicator.
 *   teo_comm_unique_id : 128 bytes (ncclUniqueId) generated on ONE rank; the host code hands them to every rank by any
 *                        control-plane means (a torch.distributed store, MPI, a file) -- never the data path.
 *   teo_ctx_create     : collective over all ranks (ncclCommInitRank); world_size == 1 accepts unique_id == NULL.
 *   teo_allgather_visual : out[r * rows_per_rank + i, :] = local_of_rank_r[i, :] on every rank (ncclAllGather over xGMI),
 *                        enqueued on `stream`; every rank passes the same rows_per_rank (ragged blocks are padded by the
 *                        caller).  dtype TEO_F32 / TEO_BF16. */
#define TEO_COMM_ID_BYTES 128
typedef struct teo_ctx teo_ctx;
int teo_comm_unique_id(void* out_id /* TEO_COMM_ID_BYTES */);
int teo_ctx_create(int rank, int world_size, const void* unique_id, int device, teo_ctx** out);
int teo_ctx_destroy(teo_ctx* ctx);
int teo_ctx_info(const teo_ctx* ctx, int* rank, int* world_size, int* cu_count, size_t* hbm_bytes);
/* The knob block a context owns (created with it, destroyed with it): the natural home of an engine's knobs when the engine has a
 * context -- point the descriptors' `tune` at it. */
teo_tune* teo_ctx_tune(teo_ctx* ctx);
int teo_allgather_visual(teo_ctx* ctx, const void* d_local, void* d_out, int rows_per_rank, int dim, int dtype,
                         teo_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* TEO_HIP_H */
