// Batched-decode ("skinny") GEMM for gfx950:  out[b][n] = sum_k x[b][k] * W[n][k],  b < MB <= 16 conversations.
//
// One decode step of B conversations multiplies the SAME weight matrices with B activation rows, so the weights
// are streamed from HBM once per step instead of once per conversation (config C5's batched-decode variant;
// reference: HF generate with batch>1 through LlamaForCausalLM.forward, videollava/model/language_model/
// llava_llama.py:88-99).  The kernel is HBM-bound like the GEMV; the arithmetic goes to the matrix cores only
// because 16 dot products per weight element would otherwise saturate the VALU:
//
//   * v_mfma_f32_16x16x32_bf16 with the WEIGHT tile as the first operand (16 rows x 32 k, read straight from
//     global memory into the operand registers -- no LDS staging: each weight byte is used exactly once; with the
//     TEO_GEMM_WTILED layout every load instruction reads 1 KB contiguous, +25..35 % over row-major) and the
//     activations x^T as the second operand (lane = (conversation b, k-group)); columns b >= MB compute garbage
//     that is never written.
//   * a workgroup = 8 waves = RT row tiles (16 weight rows each) x 8/RT contiguous K slices; a wave streams one
//     tile row over its slice.  The partial 16x16 tiles are reduced through LDS in a fixed order (deterministic),
//     then scale (fp8), SwiGLU, residual and rounding happen once.  x comes from L2 (K*MB*2 bytes per workgroup).
//   * fp8-e4m3 weights: a lane's 16-byte chunk covers 16 k of one row = two MFMAs (the k-permutation inside a
//     64-k step is the same for W and x, so the dot product is unchanged); fp8 -> bf16 is exact.
#include "common.h"
#include "ops.h"

namespace teo {

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef unsigned char fp8_t;

constexpr int SK_WAVES = 8;
constexpr int SK_THREADS = SK_WAVES * 64;
constexpr int SK_TP = 16 * 17;      // padded 16x16 partial tile in LDS: [b][i] at b*17 + i

static int g_sk_tiles = 0;          // 0 = auto
static int g_sk_nt = 1;
static int g_sk_stream = 1;         // 0 = never, 1 = auto, 2 = whenever the streaming form is eligible
void skinny_tune_reset() { g_sk_tiles = 0; g_sk_nt = 1; g_sk_stream = 1; }
int skinny_tune_set(const char* key, int value) {
    if (!strcmp(key, "skinny_tiles") && (value == 0 || value == 1 || value == 2 || value == 4 || value == 8)) { g_sk_tiles = value; return 0; }
    if (!strcmp(key, "skinny_nt")) { g_sk_nt = value != 0; return 0; }
    if (!strcmp(key, "skinny_stream") && value >= 0 && value <= 2) { g_sk_stream = value; return 0; }
    return -1;
}

// Timeline marks of the probe build (tools/skinny_probe.hip instantiates TRACE = true; the library only TRACE = false):
// slot s of workgroup blockIdx.x <- the 100 MHz wall clock, written by one lane.
template <bool TRACE>
__device__ __forceinline__ void sk_mark(const SkinnyFuse& fuse, int slot) {
    if constexpr (TRACE) {
        if ((threadIdx.x & 63) == 0) fuse.trace[(long long)blockIdx.x * SK_TRACE_SLOTS + slot] = wall_clock64();
    }
}

template <bool NT>
__device__ __forceinline__ u32x4 sk_ldw(const void* p) {
    if (NT) return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
    return *reinterpret_cast<const u32x4*>(p);
}

// 8 fp8 e4m3 (two dwords) -> 8 bf16, exact (3 mantissa bits; every e4m3 value is a bf16 value): four
// v_cvt_scalef32_pk_bf16_fp8 (two values per instruction, scale 1)
typedef __bf16 sk_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bf16x8 fp8x8_to_bf16x8(unsigned a, unsigned b) {
    u32x4 r;
    r.x = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(a, 1.0f, false));
    r.y = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(a, 1.0f, true));
    r.z = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(b, 1.0f, false));
    r.w = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(b, 1.0f, true));
    return __builtin_bit_cast(bf16x8, r);
}

// x (8 bf16) * g (8 bf16) -> 8 bf16 (one rounding), and the sum of squares of x
__device__ __forceinline__ u32x4 sk_scale_frag(const u32x4& xv, const u32x4& gv, float& ssq) {
    u32x4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float x0 = __uint_as_float(xv[i] << 16), x1 = __uint_as_float(xv[i] & 0xffff0000u);
        const float g0 = __uint_as_float(gv[i] << 16), g1 = __uint_as_float(gv[i] & 0xffff0000u);
        ssq = fmaf(x0, x0, ssq);
        ssq = fmaf(x1, x1, ssq);
        r[i] = pack_bf2(x0 * g0, x1 * g1);
    }
    return r;
}

// NORM: fused RMSNorm of the activation rows.  out[b][n] = inv_rms[b] * sum_k W[n][k] * bf16(x[b][k] * g[k]): the
// per-row factor 1/rms commutes with the GEMM, so the kernel streams x once, accumulates sum(x^2) from the very
// fragments it feeds to the matrix cores and applies inv_rms in the epilogue -- no separate norm launch, no
// normalised copy of x.  (g is staged in LDS once per workgroup.)
template <typename WT, int UNR, bool NT, bool SWIGLU, bool NORM, bool TRACE = false>
__global__ __launch_bounds__(SK_THREADS) void skinny_gemm_kernel(const bf16_t* __restrict__ x, const WT* __restrict__ W,
                                                                 const float* __restrict__ wscale,
                                                                 const bf16_t* __restrict__ norm_w, float eps,
                                                                 const bf16_t* res, void* outv,
                                                                 int MB, int N, int K, int ldx, int ldo, int ldr, int tiled,
                                                                 int out_f32, int RT, SkinnyFuse fuse, int sw8) {
    constexpr bool F8 = sizeof(WT) == 1;
    constexpr int KS = F8 ? 64 : 32;                     // k elements per step (one 16-byte chunk per lane)
    constexpr int CH = F8 ? 16 : 8;                      // k elements per lane chunk
    constexpr int XL = F8 ? 2 : 1;                       // 16-byte activation loads per step
    __shared__ float red[SK_WAVES][SK_TP];
    __shared__ float ssq_part[SK_WAVES][16];
    __shared__ float inv_s[16];
    extern __shared__ __attribute__((aligned(16))) unsigned char sk_dyn[];      // NORM: g[K] bf16
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    if (wid == 0) sk_mark<TRACE>(fuse, 0);
    if (NORM) {
        for (int i = tid; i < K / 8; i += SK_THREADS)
            reinterpret_cast<u32x4*>(sk_dyn)[i] = reinterpret_cast<const u32x4*>(norm_w)[i];
        __syncthreads();
    }
    const bf16_t* gs = reinterpret_cast<const bf16_t*>(sk_dyn) + fg * (sizeof(WT) == 1 ? 16 : 8);
    float ssq = 0.f;
    // the 8 waves of a workgroup = RT row tiles (16 weight rows each) x KSPLIT contiguous K slices
    const int KSPLIT = SK_WAVES / RT;
    const int rt = wid / KSPLIT, ks = wid % KSPLIT;
    const int n0 = blockIdx.x * 16 * RT;
    const int nsteps = K / KS;
    const int per = (nsteps + KSPLIT - 1) / KSPLIT;
    const int s0 = ks * per, s1 = min(s0 + per, nsteps);

    // weight addressing.  Row-major [N][K]: a wave instruction touches 16 rows x 64 B.  Tiled (TEO_GEMM_WTILED): the
    // matrix is stored as 1 KB tiles of 16 rows x KS k in operand order (tile (n/16, k/KS) at ((n/16)*(K/KS) + k/KS) KB,
    // lane l = (k%KS)/CH*16 + n%16 owns bytes [16 l, 16 l + 16)), so one instruction reads 1 KB contiguous.
    const WT* wp;
    long long pstep;
    if (tiled) {
        wp = W + ((long long)min(n0 / 16 + rt, (N + 15) / 16 - 1) * nsteps) * (64 * CH) + lane * CH;
        pstep = 64 * CH;
    } else {
        wp = W + (long long)min(n0 + rt * 16 + fr, N - 1) * K + fg * CH;
        pstep = KS;
    }
    const bf16_t* xp = x + (long long)min(fr, MB - 1) * ldx + fg * CH;

    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};

    // Epilogue operands of this thread's first output (fp8 row scales, residual, the next norm's weight), requested AHEAD of
    // the weight stream: after the reduction barrier they would be a serial global round trip at the end of every workgroup
    // (1.5-2 us of a 6-14 us o / down launch).
    float pf_s0 = 1.f, pf_s1 = 1.f, pf_res = 0.f, pf_g = 0.f;
    {
        const int OUTC = SWIGLU ? 8 * RT : 16 * RT;
        const int b = tid / OUTC, c = tid % OUTC;
        if (tid < 16 * OUTC && b < MB) {
            if (SWIGLU) {
                const int tp = sw8 ? (c >> 3) : (c >> 4), i = sw8 ? (c & 7) : (c & 15);
                const int ng = n0 + (sw8 ? tp * 16 : tp * 32) + i, nu = ng + (sw8 ? 8 : 16);
                if (ng < N && wscale) { pf_s0 = wscale[ng]; pf_s1 = wscale[nu]; }
            } else {
                const int col = n0 + c;
                if (col < N) {
                    if (wscale) pf_s0 = wscale[col];
                    if (res) pf_res = bf2f(res[(long long)b * ldr + col]);
                    if (fuse.xg_out) pf_g = bf2f(fuse.next_g[col]);
                }
            }
        }
    }

    // Software pipeline over the wave's K slice: two register sets of UNR steps each; every load is unconditional
    // (step index clamped to the slice, the x fragment of an out-of-range step is zeroed) so hipcc keeps counting
    // vmcnt instead of draining the queue at a control-flow merge.
#define TEO_SK_LOAD(WR, XR, BASE)                                                                              \
    _Pragma("unroll") for (int u = 0; u < UNR; ++u) {                                                          \
        const long long si = min((BASE) + u, s1 - 1);                                                          \
        WR[u] = sk_ldw<NT>(wp + si * pstep);                                                                   \
        _Pragma("unroll") for (int j = 0; j < XL; ++j)                                                         \
            XR[u][j] = *reinterpret_cast<const u32x4*>(xp + si * KS + j * 8);                                  \
    }
#define TEO_SK_COMP(WR, XR, BASE)                                                                              \
    _Pragma("unroll") for (int u = 0; u < UNR; ++u) {                                                          \
        const bool ok = (BASE) + u < s1;                                                                       \
        u32x4 x0 = XR[u][0], x1 = XR[u][XL - 1];                                                               \
        if (!ok) { x0 = (u32x4){0u, 0u, 0u, 0u}; x1 = x0; }                                                    \
        if (NORM) {                                                                                            \
            const long long sg = min((BASE) + u, s1 - 1);                                                      \
            x0 = sk_scale_frag(x0, *reinterpret_cast<const u32x4*>(gs + sg * KS), ssq);                        \
            if (F8) x1 = sk_scale_frag(x1, *reinterpret_cast<const u32x4*>(gs + sg * KS + 8), ssq);            \
        }                                                                                                      \
        if (F8) {                                                                                              \
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fp8x8_to_bf16x8(WR[u].x, WR[u].y),                   \
                                                          __builtin_bit_cast(bf16x8, x0), acc, 0, 0, 0);       \
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fp8x8_to_bf16x8(WR[u].z, WR[u].w),                   \
                                                          __builtin_bit_cast(bf16x8, x1), acc, 0, 0, 0);       \
        } else {                                                                                               \
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, WR[u]),                   \
                                                          __builtin_bit_cast(bf16x8, x0), acc, 0, 0, 0);       \
        }                                                                                                      \
    }

#define TEO_SK_SSQ_LOAD \
        float sq0 = 0.f, sq1 = 0.f; \
        if (fuse.ssq_in) { \
            const float* r0 = fuse.ssq_in + (long long)min(wid, MB - 1) * fuse.nparts; \
            const float* r1 = fuse.ssq_in + (long long)min(wid + 8, MB - 1) * fuse.nparts; \
            for (int p0 = lane; p0 < fuse.nparts; p0 += 256) { \
                float a0[4], a1[4]; \
_Pragma("unroll") \
                for (int i = 0; i < 4; ++i) { \
                    const int p_ = min(p0 + 64 * i, fuse.nparts - 1); \
                    a0[i] = r0[p_]; \
                    a1[i] = r1[p_]; \
                } \
_Pragma("unroll") \
                for (int i = 0; i < 4; ++i) { \
                    const bool in = p0 + 64 * i < fuse.nparts; \
                    sq0 += in ? a0[i] : 0.f; \
                    sq1 += in ? a1[i] : 0.f; \
                } \
            } \
        }
#define TEO_SK_SSQ_REDUCE \
        if (fuse.ssq_in) { \
            sq0 = wave_sum(sq0); \
            sq1 = wave_sum(sq1); \
            if (lane == 0) { \
                if (wid < MB) inv_s[wid] = rsqrtf(sq0 / (float)K + fuse.eps); \
                if (wid + 8 < MB) inv_s[wid + 8] = rsqrtf(sq1 / (float)K + fuse.eps); \
            } \
        }

    if (s0 < s1) {
        u32x4 wa[UNR], xa[UNR][XL], wb[UNR], xb[UNR][XL];
        int s = s0;
        // producer-side RMSNorm (SkinnyFuse): 1/rms of the rows from the producer's partial sums.  Wave w owns rows w
        // and w + 8.  The partials are requested FIRST and the wave's first weight batch right behind them, so the
        // reduction's L2 round trip runs under the HBM latency of the weight stream (vmcnt counts in order: waiting for
        // the older ssq loads does not wait for the weights).
        TEO_SK_SSQ_LOAD
        TEO_SK_LOAD(wa, xa, s)
        TEO_SK_SSQ_REDUCE
        if (wid == 0) sk_mark<TRACE>(fuse, 1);
        for (; s + 2 * UNR < s1; s += 2 * UNR) {
            TEO_SK_LOAD(wb, xb, s + UNR)
            TEO_SK_COMP(wa, xa, s)
            TEO_SK_LOAD(wa, xa, s + 2 * UNR)
            TEO_SK_COMP(wb, xb, s + UNR)
        }
        TEO_SK_LOAD(wb, xb, s + UNR)
        TEO_SK_COMP(wa, xa, s)
        TEO_SK_COMP(wb, xb, s + UNR)
    } else {                                             // K shorter than 8 slices: this wave only owes its rows' 1/rms
        TEO_SK_SSQ_LOAD
        TEO_SK_SSQ_REDUCE
    }
#undef TEO_SK_LOAD
#undef TEO_SK_COMP

    // partial tiles -> LDS: lane holds out[b = fr][row i = fg*4 + r] of its tile
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wid][fr * 17 + fg * 4 + r] = acc[r];
    if (NORM) {                                          // lanes (fr, fg = 0..3) hold row fr's partial sum of squares
        ssq += __shfl_xor(ssq, 16, 64);
        ssq += __shfl_xor(ssq, 32, 64);
        if (fg == 0) ssq_part[wid][fr] = ssq;
    }
    if (wid == 0) sk_mark<TRACE>(fuse, 2);
    __syncthreads();
    if (wid == 0) sk_mark<TRACE>(fuse, 3);

    const int OUTC = SWIGLU ? 8 * RT : 16 * RT;          // output columns of this workgroup
    float emit_v = 0.f;
    bool emit_ok = false;
    for (int o = tid; o < 16 * OUTC; o += SK_THREADS) {
        const int b = o / OUTC, c = o % OUTC;
        if (b >= MB) break;
        const bool first = o == tid;                     // this thread's prefetched operands apply
        float inv = 1.f;
        if (fuse.ssq_in) inv = inv_s[b];
        if (NORM) {                                      // the K slices of row tile 0 cover the whole row
            float t = 0.f;
            for (int w = 0; w < KSPLIT; ++w) t += ssq_part[w][b];
            inv = rsqrtf(t / (float)K + eps);
        }
        float v;
        int col;
        if (SWIGLU) {
            // 16-row interleave: tile pair (gate tile 2 tp, up tile 2 tp + 1); 8-row interleave: rows i / i + 8 of one tile
            const int tp = sw8 ? (c >> 3) : (c >> 4), i = sw8 ? (c & 7) : (c & 15);
            const int tg = sw8 ? tp : 2 * tp, tu = sw8 ? tp : 2 * tp + 1, iu = sw8 ? i + 8 : i;
            const int ng = n0 + (sw8 ? tp * 16 : tp * 32) + i, nu = ng + (sw8 ? 8 : 16);
            if (ng >= N) continue;
            float g = 0.f, u = 0.f;
            for (int w = 0; w < KSPLIT; ++w) {
                g += red[tg * KSPLIT + w][b * 17 + i];
                u += red[tu * KSPLIT + w][b * 17 + iu];
            }
            g *= inv; u *= inv;
            if (wscale) { g *= first ? pf_s0 : wscale[ng]; u *= first ? pf_s1 : wscale[nu]; }
            v = silu(g) * u;
            col = (n0 >> 1) + c;
        } else {
            const int t = c >> 4, i = c & 15;
            col = n0 + c;
            if (col >= N) continue;
            v = 0.f;
            for (int w = 0; w < KSPLIT; ++w) v += red[t * KSPLIT + w][b * 17 + i];
            v *= inv;
            if (wscale) v *= first ? pf_s0 : wscale[col];
        }
        if (res) v += first ? pf_res : bf2f(res[(long long)b * ldr + col]);
        if (out_f32) reinterpret_cast<float*>(outv)[(long long)b * ldo + col] = v;
        else {
            const bf16_t hb = f2bf(v);
            reinterpret_cast<bf16_t*>(outv)[(long long)b * ldo + col] = hb;
            emit_v = bf2f(hb);
            emit_ok = true;
        }
    }
    if (!SWIGLU && fuse.xg_out && tid < 256) {
        // RT == 1 (host-enforced): thread = (b = tid / 16, column c = tid % 16) and it has just produced h[b][n0 + c].
        // Emit what the NEXT layer's RMSNorm needs: bf16(h * g_next) and this workgroup's share of sum(h^2) per row.
        const int b = tid >> 4, c = tid & 15, col = n0 + c;
        float sq = emit_ok ? emit_v * emit_v : 0.f;
        if (emit_ok) fuse.xg_out[(long long)b * ldo + col] = f2bf(emit_v * pf_g);   // (b, c) = the prefetch's mapping at RT == 1
        sq += __shfl_xor(sq, 8, 64);
        sq += __shfl_xor(sq, 4, 64);
        sq += __shfl_xor(sq, 2, 64);
        sq += __shfl_xor(sq, 1, 64);
        if (c == 0 && b < MB) fuse.ssq_out[(long long)b * gridDim.x + blockIdx.x] = sq;
    }
    if (wid == 0) sk_mark<TRACE>(fuse, 4);
}

// ---- streaming form --------------------------------------------------------------------------------------------------
// The kernel above pays its fixed costs once per 16 weight rows: the workgroup start, the activation fragments (as many
// bytes as the fp8 weights of the tile, twice that for 16 conversations), the drain into the LDS reduction and the
// epilogue, during which the workgroup has no weight request in flight.  Measured on the 7B shapes the weight stream of a
// batched step runs at 3.7-4.2 TB/s (fp8) / 5.5 TB/s (bf16, 8 rows) / 4.2 TB/s (bf16, 16 rows) against 6.4 for a plain
// read; with the activation loads removed (probe) the 16-row case alone gains 40 %.
//
// Here a workgroup is PERSISTENT: 8 streaming waves own one K slice each for the whole launch and walk the row tiles
// blockIdx.x, blockIdx.x + gridDim.x, ...:
//   * the slice's activation fragments are loaded ONCE into registers (PER steps x 16 B [x 2 for fp8] per lane = 64
//     VGPRs at K = 4096) -- no activation traffic after the first tile, and only weight loads in the vmcnt queue;
//   * the weight stream is one flat double-buffered pipeline ACROSS tiles: the first loads of tile i + 1 are in flight
//     while tile i's partial sums go to LDS;
//   * a ninth wave does every epilogue (fixed-order reduction of the 8 partial tiles, 1/rms, fp8 scale, SwiGLU,
//     residual, rounding, the next norm's hand-off): the streaming waves never issue another global load, so nothing
//     they wait for sits behind the prefetched weights, and their only synchronisation is one s_barrier per tile
//     (partial tiles double-buffered; the epilogue of tile i overlaps the stream of tile i + 1).
// Same arithmetic as the kernel above (same K partition when K / KS is a multiple of 8, same reduction order).
constexpr int SS_NW = 8;                       // streaming waves; wave SS_NW is the epilogue wave
constexpr int SS_THREADS = (SS_NW + 1) * 64;
constexpr int SS_TP = 16 * 20;                 // partial tile in LDS: [b][i] at b*20 + i (16-byte aligned rows)

template <typename WT, int UNR, int SPT, bool SW8, bool TRACE = false>
__global__ __launch_bounds__(SS_THREADS) void skinny_stream_kernel(const bf16_t* __restrict__ x, const WT* __restrict__ W,
                                                                   const float* __restrict__ wscale, const bf16_t* res, void* outv,
                                                                   int MB, int N, int K, int ldx, int ldo, int tiled, int out_f32,
                                                                   SkinnyFuse fuse) {
    constexpr bool F8 = sizeof(WT) == 1;
    constexpr int KS = F8 ? 64 : 32, CH = F8 ? 16 : 8, XL = F8 ? 2 : 1;
    constexpr int PER = UNR * SPT;                       // steps of one wave per tile
    constexpr int TU = (SPT & 1) ? 2 : 1;                // tiles per trip of the unrolled loop (an even number of sets)
    __shared__ __attribute__((aligned(16))) float red[2][SS_NW][SS_TP];
    __shared__ float inv_s[16];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int ntiles = (N + 15) / 16, nsteps = K / KS, G = gridDim.x;

    if (wid == 0 || wid == SS_NW) sk_mark<TRACE>(fuse, wid == 0 ? 0 : 8);
    if (wid < SS_NW) {
        const int fr = lane & 15, fg = lane >> 4;
        const int s0 = wid * PER;
        int ntrace = 1;
        TEO_SK_SSQ_LOAD
        // the slice's activations, once
        u32x4 xr[PER][XL];
        {
            const bf16_t* xp = x + (long long)min(fr, MB - 1) * ldx + fg * CH;
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const long long sc = min(s0 + i, nsteps - 1);
#pragma unroll
                for (int j = 0; j < XL; ++j) xr[i][j] = *reinterpret_cast<const u32x4*>(xp + sc * KS + j * 8);
            }
        }
        const long long pstep = tiled ? 64 * CH : KS;
        const long long tstride = (long long)nsteps * (64 * CH);
        u32x4 w[2][UNR];
        // weight loads of set ls of tile t (t clamped: a trailing prefetch re-reads the last tile and is dropped)
#define TEO_SS_LOADW(BUF, T, LS)                                                                               \
        {                                                                                                      \
            const int tc = min((T), ntiles - 1);                                                               \
            const WT* wt = tiled ? W + (long long)tc * tstride + lane * CH                                     \
                                 : W + (long long)min(tc * 16 + fr, N - 1) * K + fg * CH;                      \
            _Pragma("unroll") for (int u = 0; u < UNR; ++u)                                                    \
                w[BUF][u] = sk_ldw<true>(wt + (long long)min(s0 + (LS) * UNR + u, nsteps - 1) * pstep);        \
        }
        int t = blockIdx.x;
        TEO_SS_LOADW(0, t, 0)
        TEO_SK_SSQ_REDUCE
#pragma unroll
        for (int i = 0; i < PER; ++i)                    // steps past the end of K contribute nothing
            if (s0 + i >= nsteps) {
#pragma unroll
                for (int j = 0; j < XL; ++j) xr[i][j] = (u32x4){0u, 0u, 0u, 0u};
            }
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
        int par = 0;
        for (; t < ntiles; t += TU * G) {
#pragma unroll
            for (int q = 0; q < TU * SPT; ++q) {
                const int tq = t + (q / SPT) * G, ls = q % SPT;
                if (q + 1 < TU * SPT) TEO_SS_LOADW((q + 1) & 1, t + ((q + 1) / SPT) * G, (q + 1) % SPT)
                else                  TEO_SS_LOADW(0, t + TU * G, 0)
                {
                    // (a trailing tile past the end is computed on the clamped re-read and dropped: the loads above must
                    // stay unconditional users, or the compiler sinks them below the previous tile's barrier)
#pragma unroll
                    for (int u = 0; u < UNR; ++u) {
                        const int i = ls * UNR + u;
                        if (F8) {
                            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fp8x8_to_bf16x8(w[q & 1][u].x, w[q & 1][u].y),
                                                                          __builtin_bit_cast(bf16x8, xr[i][0]), acc, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fp8x8_to_bf16x8(w[q & 1][u].z, w[q & 1][u].w),
                                                                          __builtin_bit_cast(bf16x8, xr[i][XL - 1]), acc, 0, 0, 0);
                        } else {
                            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w[q & 1][u]),
                                                                          __builtin_bit_cast(bf16x8, xr[i][0]), acc, 0, 0, 0);
                        }
                    }
                    if (ls == SPT - 1 && (TU == 1 || q < SPT || tq < ntiles)) {   // tile done: lane holds out[b = fr][rows fg*4 .. +3]
                        *reinterpret_cast<f32x4*>(&red[par][wid][fr * 20 + fg * 4]) = acc;
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        __builtin_amdgcn_s_barrier();
                        if (wid == 0 && ntrace < 8) sk_mark<TRACE>(fuse, ntrace++);
                        acc = (f32x4){0.f, 0.f, 0.f, 0.f};
                        par ^= 1;
                    }
                }
            }
        }
#undef TEO_SS_LOADW
    } else {
        // epilogue wave: lane = (conversation b, 4 consecutive rows of the tile).  The tile's scales / residual / next-norm
        // weights are requested one tile AHEAD (a global round trip under a saturated HBM pipe is about one tile period:
        // taken after the barrier it would make this wave the bottleneck of the workgroup)
        const int b = lane >> 2, i4 = lane & 3;
        const bool bok = b < MB;
        const long long rrow = (long long)min(b, MB - 1) * ldo;
        f32x4 sc_n = (f32x4){1.f, 1.f, 1.f, 1.f};
        uint2 rv_n = make_uint2(0u, 0u), gv_n = make_uint2(0u, 0u);
#define TEO_SS_PREFETCH(T)                                                                                     \
        {                                                                                                      \
            const int tp = min((T), ntiles - 1), np = tp * 16 + i4 * 4;                                        \
            if (tp * 16 + 16 <= N) {                                                                           \
                if (wscale) sc_n = *reinterpret_cast<const f32x4*>(wscale + np);                               \
                if (!SW8 && res) rv_n = *reinterpret_cast<const uint2*>(res + rrow + np);                      \
                if (!SW8 && fuse.xg_out) gv_n = *reinterpret_cast<const uint2*>(fuse.next_g + np);             \
            }                                                                                                  \
        }
        TEO_SS_PREFETCH(blockIdx.x)
        int par = 0, ntrace = 9;
        for (int t = blockIdx.x; t < ntiles; t += G) {
            asm volatile("" ::: "memory");               // the previous tile's LDS reads and stores stay BEFORE this barrier ...
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");               // ... and this tile's partial sums are read AFTER it (s_barrier alone does not order them for the compiler)
            const f32x4 sc = sc_n;
            const uint2 rv = rv_n, gv = gv_n;
            TEO_SS_PREFETCH(t + G)
            f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int wv = 0; wv < SS_NW; ++wv) v += *reinterpret_cast<const f32x4*>(&red[par][wv][b * 20 + i4 * 4]);
            par ^= 1;
            const int n0 = t * 16, nb = n0 + i4 * 4;    // first weight row of the lane
            const float inv = fuse.ssq_in ? inv_s[b] : 1.f;
            const bool full = n0 + 16 <= N;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] *= inv;
            if (wscale) {
                if (full) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] *= sc[r];
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] *= wscale[min(nb + r, N - 1)];
                }
            }
            if (SW8) {
                // rows 0..7 of the tile are gate rows, 8..15 their up rows: lanes i4 < 2 pair with lane + 2
                f32x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = silu(v[r]) * __shfl_down(v[r], 2, 64);
                if (bok && i4 < 2) {
                    const long long at = (long long)b * ldo + t * 8 + i4 * 4;
                    if (out_f32) *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(outv) + at) = o;
                    else {
                        uint2 pk;
                        pk.x = pack_bf2(o[0], o[1]);
                        pk.y = pack_bf2(o[2], o[3]);
                        *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(outv) + at) = pk;
                    }
                }
            } else {
                const long long at = (long long)b * ldo + nb;
                float sq = 0.f;
                if (full) {
                    if (res) {
                        v[0] += __uint_as_float(rv.x << 16); v[1] += __uint_as_float(rv.x & 0xffff0000u);
                        v[2] += __uint_as_float(rv.y << 16); v[3] += __uint_as_float(rv.y & 0xffff0000u);
                    }
                    if (out_f32) { if (bok) *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(outv) + at) = v; }
                    else {
                        uint2 pk;
                        pk.x = pack_bf2(v[0], v[1]);
                        pk.y = pack_bf2(v[2], v[3]);
                        if (bok) *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(outv) + at) = pk;
                        if (fuse.xg_out) {
                            // what the NEXT layer's RMSNorm needs: bf16(h * g_next) and this tile's share of sum(h^2) per row
                            const float h0 = __uint_as_float(pk.x << 16), h1 = __uint_as_float(pk.x & 0xffff0000u);
                            const float h2 = __uint_as_float(pk.y << 16), h3 = __uint_as_float(pk.y & 0xffff0000u);
                            uint2 xg;
                            xg.x = pack_bf2(h0 * __uint_as_float(gv.x << 16), h1 * __uint_as_float(gv.x & 0xffff0000u));
                            xg.y = pack_bf2(h2 * __uint_as_float(gv.y << 16), h3 * __uint_as_float(gv.y & 0xffff0000u));
                            if (bok) *reinterpret_cast<uint2*>(fuse.xg_out + at) = xg;
                            sq = h0 * h0 + h1 * h1 + h2 * h2 + h3 * h3;
                        }
                    }
                } else {                                 // ragged last tile: element-wise
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int col = nb + r;
                        if (col < N && bok) {
                            float y = v[r];
                            if (res) y += bf2f(res[at + r]);
                            if (out_f32) reinterpret_cast<float*>(outv)[at + r] = y;
                            else {
                                const bf16_t hb = f2bf(y);
                                reinterpret_cast<bf16_t*>(outv)[at + r] = hb;
                                if (fuse.xg_out) {
                                    const float h = bf2f(hb);
                                    fuse.xg_out[at + r] = f2bf(h * bf2f(fuse.next_g[col]));
                                    sq += h * h;
                                }
                            }
                        }
                    }
                }
                if (fuse.xg_out) {
                    sq += __shfl_xor(sq, 1, 64);
                    sq += __shfl_xor(sq, 2, 64);
                    if (i4 == 0 && bok) fuse.ssq_out[(long long)b * ntiles + t] = sq;
                }
            }
            if (ntrace < 16) sk_mark<TRACE>(fuse, ntrace++);
        }
#undef TEO_SS_PREFETCH
    }
}
#undef TEO_SK_SSQ_LOAD
#undef TEO_SK_SSQ_REDUCE

bool skinny_gemm_ok(int MB, int N, int K, int ldx, int w_fp8, unsigned flags, const void* x, const void* W) {
    const int ks = w_fp8 ? 64 : 32;
    if (MB < 1 || MB > 16 || N < 1 || K < ks || K % ks != 0 || ldx % 8 != 0) return false;
    if ((reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(W) & 15)) return false;
    if ((flags & TEO_GEMM_SWIGLU16) && N % 32 != 0) return false;
    if ((flags & TEO_GEMM_SWIGLU8) && (N % 16 != 0 || (flags & TEO_GEMM_SWIGLU16))) return false;
    return true;
}

// bf16 activations; W bf16 or fp8 e4m3 (+ per-row scales), row-major or TEO_GEMM_WTILED; out bf16 or f32;
// res (bf16, may alias out) optional
// norm_w != NULL: fused RMSNorm of x (see the kernel).  fuse: producer-side norm hand-off (ops.h SkinnyFuse).
int skinny_gemm(const void* x, const void* W, const float* wscale, int w_fp8, const void* norm_w, float eps, const void* res,
                void* out, int MB, int N, int K, int ldx, int ldo, unsigned flags, int out_dtype, hipStream_t st, SkinnyFuse fuse) {
    const bool swiglu = (flags & (TEO_GEMM_SWIGLU16 | TEO_GEMM_SWIGLU8)) != 0;
    const int tiled = (flags & TEO_GEMM_WTILED) ? 1 : 0;
    if (!skinny_gemm_ok(MB, N, K, ldx, w_fp8, flags, x, W)) {
        set_error("skinny_gemm: unsupported MB=%d N=%d K=%d ldx=%d", MB, N, K, ldx);
        return TEO_ERR_UNSUPPORTED;
    }
    TEO_CHECK_ARG(!w_fp8 || wscale, "skinny_gemm: fp8 weights need per-row scales");
    TEO_CHECK_ARG(!(swiglu && res), "skinny_gemm: SWIGLU16 takes no residual");
    TEO_CHECK_ARG(!norm_w || (K <= 16384 && (reinterpret_cast<uintptr_t>(norm_w) & 15) == 0), "skinny_gemm: fused norm needs K <= 16384 and an aligned weight");
    // row tiles per workgroup (8 waves = RT row tiles x 8/RT K slices)
    TEO_CHECK_ARG(!(fuse.xg_out && (swiglu || out_dtype == TEO_F32 || !fuse.next_g || !fuse.ssq_out)),
                  "skinny_gemm: the norm hand-off needs a plain bf16 output, next_g and ssq_out");
    TEO_CHECK_ARG(!(fuse.ssq_in && norm_w), "skinny_gemm: ssq_in and norm_w are exclusive");
    TEO_CHECK_ARG(!fuse.ssq_in || fuse.nparts >= 1, "skinny_gemm: nparts %d", fuse.nparts);
    int rt = g_sk_tiles;
    if (rt == 0 || fuse.xg_out) rt = 1;   // measured: one row tile per workgroup (most waves in flight) wins at every N
    const int sw8 = (flags & TEO_GEMM_SWIGLU8) ? 1 : 0;   // gate/up interleaved in blocks of 8 rows: a pair fits one tile
    if (swiglu && !sw8 && rt < 2) rt = 2; // 16-row interleave: the gate tile and its up tile meet in the epilogue
    const int ldr = ldo, of = out_dtype == TEO_F32;
    // streaming form (persistent workgroups, activations in registers): K <= 4096, one row tile per workgroup, vector
    // epilogue accesses aligned
    {
        const int nsteps = K / (w_fp8 ? 64 : 32), ntiles = cdiv(N, 16);
        const auto al = [](const void* p, uintptr_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; };
        bool ok = g_sk_stream != 0 && !norm_w && !(flags & TEO_GEMM_SWIGLU16) && (g_sk_tiles == 0 || g_sk_tiles == 1) &&
                  nsteps <= (w_fp8 ? 64 : 128) && ldo % 4 == 0 && al(out, 16) && (!res || al(res, 8)) && (!wscale || al(wscale, 16)) &&
                  (!fuse.xg_out || (al(fuse.xg_out, 8) && al(fuse.next_g, 8)));
        // auto: at least two tiles per workgroup, and where it measures faster (fp8 weights: -15 % at 8 rows, -25 % at 16;
        // bf16: -10..20 % above 8 rows, a tie at 8 or fewer where the duplicate activation rows coalesce)
        if (ok && g_sk_stream == 1) ok = ntiles >= 512 && (w_fp8 || MB > 8);
        if (ok) {
            const int cus = device_cu_count();
            const int grid = std::max(1, std::min(ntiles, cus > 0 ? cus : 256));
#define TEO_SS(WW, UN, SP, SW)                                                                              \
            skinny_stream_kernel<WW, UN, SP, SW><<<grid, SS_THREADS, 0, st>>>((const bf16_t*)x, (const WW*)W, wscale, (const bf16_t*)res, \
                                                                              out, MB, N, K, ldx, ldo, tiled, of, fuse)
            if (w_fp8) { if (sw8) TEO_SS(fp8_t, 8, 1, true); else TEO_SS(fp8_t, 8, 1, false); }
            else       { if (sw8) TEO_SS(bf16_t, 8, 2, true); else TEO_SS(bf16_t, 8, 2, false); }
#undef TEO_SS
            note_kernel("skinny_stream");
            TEO_LAUNCH_CHECK("skinny_gemm (stream)");
            return TEO_OK;
        }
    }
    const int blocks = cdiv(N, 16 * rt);
    const size_t dyn = norm_w ? (size_t)K * 2 : 0;
#define TEO_SK(WW, NTV, SW, NM)                                                                            \
    skinny_gemm_kernel<WW, 4, NTV, SW, NM><<<blocks, SK_THREADS, dyn, st>>>(                               \
        (const bf16_t*)x, (const WW*)W, wscale, (const bf16_t*)norm_w, eps, (const bf16_t*)res, out, MB, N, K, ldx, ldo, ldr, \
        tiled, of, rt, fuse, sw8)
#define TEO_SK_N(WW, NTV, SW) if (norm_w) { TEO_SK(WW, NTV, SW, true); } else { TEO_SK(WW, NTV, SW, false); }
#define TEO_SK_F(WW, NTV) if (swiglu) { TEO_SK_N(WW, NTV, true) } else { TEO_SK_N(WW, NTV, false) }
    if (w_fp8) { if (g_sk_nt) { TEO_SK_F(fp8_t, true) } else { TEO_SK_F(fp8_t, false) } }
    else       { if (g_sk_nt) { TEO_SK_F(bf16_t, true) } else { TEO_SK_F(bf16_t, false) } }
#undef TEO_SK_F
#undef TEO_SK_N
#undef TEO_SK
    note_kernel("skinny_gemm");
    TEO_LAUNCH_CHECK("skinny_gemm");
    return TEO_OK;
}

}  // namespace teo
