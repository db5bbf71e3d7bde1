// Internal C++ entry points of the kernels (one per .hip file); the extern "C" surface is in abi.hip.
#pragma once
#include "common.h"

struct teo_graph {
    hipGraph_t graph;
    hipGraphExec_t exec;
};

namespace teo {

int layernorm(const void* x, const void* w, const void* b, void* y, int rows, int dim, float eps, int dtype, hipStream_t st);
int rmsnorm(const void* x, const void* w, void* y, int rows, int dim, float eps, int dtype, hipStream_t st);
int vit_embed_ln(const void* patch, const void* cls, const void* pos, const void* w, const void* b, void* out, int T,
                 int NP, int dim, float eps, int dtype, hipStream_t st);

bool gemm_mfma_ok(int M, int N, int K, int lda, int ldc, int dtype, unsigned flags, const void* A, const void* W,
                  const void* bias, const void* res, const void* C);
int gemm(const void* A, const void* W, const void* bias, const void* res, void* C, int M, int N, int K, int lda,
         int ldc, int act, unsigned flags, int dtype, int out_dtype, hipStream_t st, void* sk_ws = nullptr);

int attention(const teo_attn_args* a, int dtype, hipStream_t st);
int attention_flash32(const teo_attn_args& a, hipStream_t st, bool f16 = false);
int attention_probs(const teo_attn_args* a, int dtype, void* probs, hipStream_t st);      // the softmax maps themselves (`output_attentions`)
size_t attn_decode_ws_bytes(int heads, int hd, int S_max, int batch = 1);
struct AttnBatch {          // per-conversation strides (elements) of a batched decode step; {1, 0, 0, 0} = one conversation
    int batch = 1;
    long long q_stride = 0, cache_stride = 0, o_stride = 0;
};
int attn_decode(const void* q, void* kc, void* vc, void* vtc, const float* rope_cos, const float* rope_sin, void* o,
                float* part, const int* d_pos, int S_max, int heads, int kv_heads, int hd, float scale, int dtype,
                hipStream_t st, AttnBatch bt = AttnBatch());

int rope_kv_append(void* qkv, int ld, const int* positions, const float* cs, const float* sn, void* kc, void* vc,
                   void* vtc, int S, int past, const int* d_past, int S_max, int heads, int kv_heads, int hd, int dtype,
                   hipStream_t st);
int vit_value_transpose(const void* qkv, void* vt, int T, int N, int heads, int hd, int ldv, int dtype, hipStream_t st);
int im2col_patches(const void* px, void* cols, int T, int C, int img, int P, int ld, int dtype, hipStream_t st);
int embed_splice(const int* plan, const void* embed, const void* visual, void* out, int rows, int dim, int dtype,
                 hipStream_t st);
int drop_cls(const void* in, void* out, int T, int ntok, int dim, int dtype, hipStream_t st);
int argmax(const float* logits, long long* tok, int rows, int vocab, hipStream_t st);
int decode_advance(const teo_decode_state* s, hipStream_t st);
// TEO_OK when the device sampler implements this (top_k, top_p) combination exactly as HF's warpers would apply it
int sampler_check(int vocab, int top_k, float top_p);
int sample_topk(const float* logits, long long* tok, int vocab, float temperature, int top_k, float top_p, unsigned long long seed,
                unsigned long long draw, hipStream_t st);
int decode_tail(const float* logits, const teo_decode_state* s, const void* embed, void* h, int vocab, int dim, int dtype,
                hipStream_t st, int batch = 1, int out_stride = 0, const void* g0 = nullptr, void* hg = nullptr,
                float* ssq = nullptr, int nparts = 0);
int embed_token(const long long* tok, const void* embed, void* h, int dim, int dtype, hipStream_t st, int batch = 1);
int embed_token_emit(const long long* tok, const void* embed, void* h, int dim, int dtype, hipStream_t st, int batch,
                     const void* g, void* hg, float* ssq, int nparts);

int gemv(const void* x, const void* W, const void* norm_w, const void* res, void* y, int N, int K, float eps,
         unsigned flags, int dtype, int out_dtype, hipStream_t st);

int preprocess_frames(const unsigned char* src, void* out, int T, int H, int W, int S, const float* mean, const float* stdv,
                      int dtype, hipStream_t st, const unsigned char* pad_rgb = nullptr);
int cross_entropy(const float* logits, long long ld, const long long* labels, float* loss_row, float* out, int rows, int vocab,
                  long long ignore_index, hipStream_t st);
bool patch_embed_ok(int C, int img, int P, int ldw, int D, int dtype, const void* px, const void* W, const void* out);
int patch_embed(const void* px, const void* W, void* out, int T, int C, int img, int P, int ldw, int D, hipStream_t st, bool f16 = false);
bool gemm_big_hybrid_fits(int M, int N, int K);
int gemm_big_ragged_tiles(int M, int N, int K = 0);    // gemm_big.hip: 128 x 512 tiles over a last row block of <= 128 rows (0: none)
long long gemm_big_tile_count(int M, int N, int K = 0); // equal-cost tiles of the 256 x 256 family for an M x N problem
bool skinny_gemm_ok(int MB, int N, int K, int ldx, int w_fp8, unsigned flags, const void* x, const void* W);
// Producer-side RMSNorm hand-off between the GEMMs of a batched decode step.  A residual-producing GEMM (o / down
// projection, one row tile per workgroup) also emits xg_out = bf16(h * next_g) and ssq_out[b][workgroup] = its 16
// columns' share of sum(h[b]^2); the consumer GEMM takes x = xg_out as a plain operand and rebuilds 1/rms per row from
// the `nparts` partial sums (ssq_in) -- the norm costs no launch, no extra pass over x and no per-step VALU work.
struct SkinnyFuse {
    const unsigned short* next_g = nullptr;   // [N] norm weight the consumer would have applied
    unsigned short* xg_out = nullptr;         // [MB][ldo]
    float* ssq_out = nullptr;                 // [MB][gridDim.x]
    const float* ssq_in = nullptr;            // [MB][nparts]
    int nparts = 0;
    float eps = 0.f;
    bool f16 = false;                         // activations, 16-bit weights and 16-bit outputs are IEEE binary16 instead of bfloat16 (no fp8 weights)
    unsigned long long* trace = nullptr;      // probe builds only (tools/skinny_probe.hip): [workgroups][SK_TRACE_SLOTS] wall-clock marks
};
constexpr int SK_TRACE_SLOTS = 16;
int skinny_gemm(const void* x, const void* W, const float* wscale, int w_fp8, const void* norm_w, float eps, const void* res,
                void* out, int MB, int N, int K, int ldx, int ldo, unsigned flags, int out_dtype, hipStream_t st,
                SkinnyFuse fuse = SkinnyFuse());
// launch helpers of the tile families; f16: the operands are IEEE binary16 (else bfloat16)
int gemm_big_launch(const void* A, const void* W, const void* bias, const void* res, void* C, int M, int N, int K, int lda, int ldc,
                    int act, bool swiglu, bool of32, bool f16, hipStream_t st, void* sk_ws, size_t flags_offset);
int gemm_wide_launch(const void* A, const void* W, const void* bias, const void* res, void* C, int M, int N, int K, int lda, int ldc,
                     int act, bool swiglu, bool of32, bool f16, hipStream_t st);
int gemm_narrow_launch(const void* A, const void* W, const void* bias, const void* res, void* C, int M, int N, int K, int lda, int ldc,
                       int act, bool of32, bool f16, int bm, hipStream_t st, bool waves8 = false);       // gemm_narrow.hip: bm 64 or 128 (waves8: the 8-wave 128 x 128 form)
int gemm_quad_launch(const void* A, const void* W, const void* bias, const void* res, void* C, int M, int N, int K, int lda, int ldc,
                     int act, bool of32, bool f16, hipStream_t st);                 // gemm_quad.hip: 256 x 160 tiles, 4 waves
int gemm_pipe_launch(const void* A, const void* W, const void* bias, const void* res, void* C, int M, int N, int K, int lda, int ldc,
                     int act, bool of32, bool f16, int bm, int tn, int ns, hipStream_t st, bool swiglu = false);   // gemm_quad.hip: the same K loop on 64 x 64 / 64 x 128 / 128 x 128 tiles
constexpr size_t GEMM_SK_SLAB_BYTES = (size_t)64 << 20;   // slab area of the stream-K workspaces (largest user: 256 x 256 KB)
constexpr int GEMM_SK_FLAG_INTS = 1024;          // hand-off flags (<= 512 used) + the sticky error word
constexpr int GEMM_SK_ERR_SLOT = GEMM_SK_FLAG_INTS - 1;   // set to 1 by a hand-off that timed out (results of that GEMM are invalid)
size_t gemm_sk_workspace_bytes();
int gemm_sk_workspace_status(const void* ws, int* host_flag, hipStream_t st);
// w8a8 prefill GEMM on the scaled fp8 MFMA (gemm_fp8.hip) and the per-token activation quantiser (norm_w != NULL: RMSNorm first)
int gemm_fp8(const void* A8, const float* a_scale, const void* W8, const float* w_scale, const void* res, void* C, int M, int N, int K,
             int lda, int ldc, unsigned flags, int out_dtype, hipStream_t st, void* sk_ws = nullptr);
int quant_rows_fp8(const void* x, const void* norm_w, void* q, float* s, int M, int K, int ldx, float eps, hipStream_t st);
int gemm_sk_workspace_init(void* ws, hipStream_t st);
int gemv_qkv_rope(const void* x, const void* W, const float* wscale, int w_fp8, const void* norm_w, void* qout,
                  const float* cs, const float* sn, const int* d_pos, void* kc, void* vc, void* vtc, int S_max, int H, int Hk,
                  int hd, int K, float eps, int dtype, hipStream_t st);
int gemv_w(const void* x, const void* W, const float* wscale, int w_fp8, const void* norm_w, const void* res, void* y, int N,
           int K, float eps, unsigned flags, int dtype, int out_dtype, hipStream_t st);

inline size_t esize(int dtype) { return dtype == TEO_F32 ? 4 : 2; }
inline size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

}  // namespace teo
