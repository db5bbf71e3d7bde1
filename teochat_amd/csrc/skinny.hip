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

// tune().skinny_tiles (default 0): 0 = auto
// tune().skinny_nt (default 1): non-temporal weight loads
// tune().skinny_stream (default 1): 0 = never, 1 = auto, 2 = whenever the streaming form is eligible
// tune().skinny_ring (default 0): streaming form, weight tiles in flight per wave: 0 = default (2 fp8 / 1 bf16: measured best end to end,
                                    // B = 8 fp8 step 3.455 vs 3.54 ms), 1 = one more (3 / 2)
// tune().skinny_unr (default 0): tile kernel, steps per register set: 0 = auto (8 for long K slices without SwiGLU / in-kernel norm), 4, 8
// tune().skinny_waves (default 0): tile kernel, waves per workgroup: 0 = auto, 8, 16 (16: K split 16 ways; plain / residual epilogues, one row tile per workgroup)
// tune().skinny_grid (default 0): streaming form, persistent workgroups per CU: 0 = auto (1), 1, 2, 3

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

// x (8 values) * g (8 values) -> 8 values of the same 16-bit format (one rounding), and the sum of squares of x
template <bool F16>
__device__ __forceinline__ u32x4 sk_scale_frag(const u32x4& xv, const u32x4& gv, float& ssq) {
    u32x4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float x0 = h_lo<F16>(xv[i]), x1 = h_hi<F16>(xv[i]);
        const float g0 = h_lo<F16>(gv[i]), g1 = h_hi<F16>(gv[i]);
        ssq = fmaf(x0, x0, ssq);
        ssq = fmaf(x1, x1, ssq);
        r[i] = pack_h2<F16>(x0 * g0, x1 * g1);
    }
    return r;
}

// NORM: fused RMSNorm of the activation rows.  out[b][n] = inv_rms[b] * sum_k W[n][k] * bf16(x[b][k] * g[k]): the
// per-row factor 1/rms commutes with the GEMM, so the kernel streams x once, accumulates sum(x^2) from the very
// fragments it feeds to the matrix cores and applies inv_rms in the epilogue -- no separate norm launch, no
// normalised copy of x.  (g is staged in LDS once per workgroup.)
template <typename WT, int UNR, bool NT, bool SWIGLU, bool NORM, bool TRACE = false, int WV = SK_WAVES>
__global__ __launch_bounds__(WV * 64) void skinny_gemm_kernel(const WT* __restrict__ W, const bf16_t* __restrict__ x,
                                                                 int MB, int N, int K, int ldx, int tiled, int RT,
                                                                 const float* __restrict__ wscale, const bf16_t* res,
                                                                 const bf16_t* __restrict__ norm_w, float eps, void* outv,
                                                                 int ldo, int ldr, int out_f32, SkinnyFuse fuse, int sw8) {
    // (argument order: what the first weight / activation requests need sits in the 14 dwords that arrive preloaded in SGPRs)
    constexpr bool F8 = sizeof(WT) == 1;
    constexpr bool F16 = IsF16<WT>::v;                   // WT = f16_t: IEEE binary16 activations, weights and 16-bit outputs (bf16_t / fp8: bfloat16)
    constexpr int KS = F8 ? 64 : 32;                     // k elements per step (one 16-byte chunk per lane)
    constexpr int CH = F8 ? 16 : 8;                      // k elements per lane chunk
    constexpr int XL = F8 ? 2 : 1;                       // 16-byte activation loads per step
    // WV waves per workgroup: 8, or 16 (round 6: the one-tile-per-CU launches -- o / down -- split K over twice as many waves, i.e.
    // twice the weight requests of a CU in flight from the first cycle; `skinny_waves`)
    constexpr int WG_THREADS = WV * 64;
    __shared__ float red[WV][SK_TP];
    __shared__ float ssq_part[WV][16];
    __shared__ float inv_s[16];
    extern __shared__ __attribute__((aligned(16))) unsigned char sk_dyn[];      // NORM: g[K] bf16
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    if (wid == 0) sk_mark<TRACE>(fuse, 0);
    if (NORM) {
        for (int i = tid; i < K / 8; i += WG_THREADS)
            reinterpret_cast<u32x4*>(sk_dyn)[i] = reinterpret_cast<const u32x4*>(norm_w)[i];
        __syncthreads();
    }
    const bf16_t* gs = reinterpret_cast<const bf16_t*>(sk_dyn) + fg * (sizeof(WT) == 1 ? 16 : 8);
    float ssq = 0.f;
    // the 8 waves of a workgroup = RT row tiles (16 weight rows each) x KSPLIT contiguous K slices
    // (RT is 1, 2, 4 or 8 -- host-checked: shifts, no integer division in front of the first weight request)
    const int rt_log2 = __builtin_ctz((unsigned)RT), ks_log2 = __builtin_ctz((unsigned)WV) - rt_log2;
    const int KSPLIT = 1 << ks_log2;
    const int rt = wid >> ks_log2, ks = wid & (KSPLIT - 1);
    const int n0 = blockIdx.x * 16 * RT;
    const int nsteps = K / KS;
    const int per = (nsteps + KSPLIT - 1) >> ks_log2;
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
    // Unconditional: absent operands read the activation buffer instead (a pointer select) and out-of-range threads a clamped
    // index -- behind a branch the compiler waits vmcnt(0) at the merge, i.e. one full memory round trip (1.7 us measured,
    // tools/skinny_probe.py) before the first weight load of the workgroup is even issued.
    float pf_s0, pf_s1, pf_res, pf_g;
    {
        const int OUTC = SWIGLU ? 8 * RT : 16 * RT;
        const int b = min(tid / OUTC, MB - 1), c = tid % OUTC;
        const float* ws_p = wscale ? wscale : reinterpret_cast<const float*>(x);
        const bf16_t* res_p = res ? res : x;
        const bf16_t* g_p = fuse.xg_out ? reinterpret_cast<const bf16_t*>(fuse.next_g) : x;
        const int ws_on = wscale ? 1 : 0, res_on = res ? 1 : 0, g_on = fuse.xg_out ? 1 : 0;
        int n_a, n_b;                                    // SWIGLU: gate row / up row; else the output column (twice)
        if (SWIGLU) {
            const int tp = sw8 ? (c >> 3) : (c >> 4), i = sw8 ? (c & 7) : (c & 15);
            n_a = n0 + (sw8 ? tp * 16 : tp * 32) + i;
            n_b = n_a + (sw8 ? 8 : 16);
        } else {
            n_a = n_b = n0 + c;
        }
        n_a = min(n_a, N - 1);
        n_b = min(n_b, N - 1);
        pf_s0 = ws_p[n_a * ws_on];
        pf_s1 = ws_p[n_b * ws_on];
        pf_res = h2f<F16>(res_p[((long long)b * ldr + n_a) * res_on]);
        pf_g = h2f<F16>(g_p[n_a * g_on]);
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
            x0 = sk_scale_frag<F16>(x0, *reinterpret_cast<const u32x4*>(gs + sg * KS), ssq);                        \
            if (F8) x1 = sk_scale_frag<F16>(x1, *reinterpret_cast<const u32x4*>(gs + sg * KS + 8), ssq);            \
        }                                                                                                      \
        if (F8) {                                                                                              \
            acc = mfma16<F16>(fp8x8_to_bf16x8(WR[u].x, WR[u].y),                   \
                                                          __builtin_bit_cast(bf16x8, x0), acc);       \
            acc = mfma16<F16>(fp8x8_to_bf16x8(WR[u].z, WR[u].w),                   \
                                                          __builtin_bit_cast(bf16x8, x1), acc);       \
        } else {                                                                                               \
            acc = mfma16<F16>(__builtin_bit_cast(bf16x8, WR[u]),                   \
                                                          __builtin_bit_cast(bf16x8, x0), acc);       \
        }                                                                                                      \
    }

    // producer-side RMSNorm (SkinnyFuse): wave w owns rows w and w + WV (WV = waves of the workgroup: 8 -> two rows each, 16 -> one); the first 256 partial sums of both rows are REQUESTED
    // here (unconditional, clamped; a dummy source when the launch takes no partials) and summed by TEO_SK_SSQ_REDUCE after the
    // wave's first weights are in flight -- they are the oldest entries of the in-order vmcnt queue, so waiting for them waits for
    // nothing else
#define TEO_SK_SSQ_LOAD \
        float sq0 = 0.f, sq1 = 0.f; \
        const float* ssq_p_ = fuse.ssq_in ? fuse.ssq_in : reinterpret_cast<const float*>(x); \
        const int ssq_on_ = fuse.ssq_in ? 1 : 0, ssq_n_ = fuse.ssq_in ? fuse.nparts : 1; \
        float ssq_a0_[4], ssq_a1_[4]; \
        { \
            const long long r0_ = (long long)min(wid, MB - 1) * ssq_n_ * ssq_on_, r1_ = (long long)min(wid + WV, MB - 1) * ssq_n_ * ssq_on_; \
_Pragma("unroll") \
            for (int i = 0; i < 4; ++i) { \
                const int p_ = min(lane + 64 * i, ssq_n_ - 1) * ssq_on_; \
                ssq_a0_[i] = ssq_p_[r0_ + p_]; \
                ssq_a1_[i] = ssq_p_[r1_ + p_]; \
            } \
        }
#define TEO_SK_SSQ_REDUCE \
        { \
            /* unconditional (a launch without partials sums its dummy loads into an inv_s nobody reads): inside `if (ssq_in)` the \
               compiler sinks the loads above into the branch, behind the weight loads, and drains the queue for them */ \
_Pragma("unroll") \
            for (int i = 0; i < 4; ++i) { \
                const bool in = lane + 64 * i < ssq_n_; \
                sq0 += in ? ssq_a0_[i] : 0.f; \
                sq1 += in ? ssq_a1_[i] : 0.f; \
            } \
            if (ssq_n_ > 256) {                                         /* more than 256 partials (hidden > 4096): the rare tail */ \
                for (int p0 = 256 + lane; p0 < fuse.nparts; p0 += 64) { \
                    sq0 += fuse.ssq_in[(long long)min(wid, MB - 1) * fuse.nparts + p0]; \
                    sq1 += fuse.ssq_in[(long long)min(wid + WV, MB - 1) * fuse.nparts + p0]; \
                } \
            } \
            sq0 = wave_sum(sq0); \
            sq1 = wave_sum(sq1); \
            if (lane == 0) { \
                if (wid < MB) inv_s[wid] = rsqrtf(fabsf(sq0) / (float)K + fuse.eps); \
                if (wid + WV < MB) inv_s[wid + WV] = rsqrtf(fabsf(sq1) / (float)K + fuse.eps); \
            } \
        }

    if (s0 < s1) {
        u32x4 wa[UNR], xa[UNR][XL], wb[UNR], xb[UNR][XL];
        int s = s0;
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
        ssq = cross_group_sum<16>(ssq);
        if (fg == 0) ssq_part[wid][fr] = ssq;
    }
    if (wid == 0) sk_mark<TRACE>(fuse, 2);
    __syncthreads();
    if (wid == 0) sk_mark<TRACE>(fuse, 3);

    const int OUTC = SWIGLU ? 8 * RT : 16 * RT;          // output columns of this workgroup
    float emit_v = 0.f;
    bool emit_ok = false;
    for (int o = tid; o < 16 * OUTC; o += WG_THREADS) {
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
        if (res) v += first ? pf_res : h2f<F16>(res[(long long)b * ldr + col]);
        if (out_f32) reinterpret_cast<float*>(outv)[(long long)b * ldo + col] = v;
        else {
            const bf16_t hb = f2h<F16>(v);
            reinterpret_cast<bf16_t*>(outv)[(long long)b * ldo + col] = hb;
            emit_v = h2f<F16>(hb);
            emit_ok = true;
        }
    }
    if (!SWIGLU && fuse.xg_out && tid < 256) {
        // RT == 1 (host-enforced): thread = (b = tid / 16, column c = tid % 16) and it has just produced h[b][n0 + c].
        // Emit what the NEXT layer's RMSNorm needs: bf16(h * g_next) and this workgroup's share of sum(h^2) per row.
        const int b = tid >> 4, c = tid & 15, col = n0 + c;
        float sq = emit_ok ? emit_v * emit_v : 0.f;
        if (emit_ok) fuse.xg_out[(long long)b * ldo + col] = f2h<F16>(emit_v * pf_g);   // (b, c) = the prefetch's mapping at RT == 1
        sq = group_sum<16>(sq);
        if (c == 0 && b < MB) fuse.ssq_out[(long long)b * gridDim.x + blockIdx.x] = sq;
    }
    if (wid == 0) sk_mark<TRACE>(fuse, 4);
}

// ---- streaming form --------------------------------------------------------------------------------------------------
// The kernel above pays its fixed costs once per 16 weight rows: the workgroup start, the activation fragments (as many
// bytes as the fp8 weights of the tile, twice that for 16 conversations), the drain into the LDS reduction and the
// epilogue, during which the workgroup has no weight request in flight.
//
// Here a workgroup is PERSISTENT: its 8 waves own one K slice each for the whole launch and walk the row tiles
// blockIdx.x, blockIdx.x + gridDim.x, ...:
//   * the slice's activation fragments are loaded ONCE into registers (PER steps x 16 B [x 2 for fp8] per lane = 64
//     VGPRs at K = 4096) -- no activation traffic after the first tile, and only weight loads in the vmcnt queue;
//   * the weight stream is one flat ring of NS register sets (8 loads each) ACROSS tiles.  Measured (round 4, tools/skinny_probe.py,
//     bench.py --batch 8 --weights fp8): deeper is NOT better -- the memory system returns the sets of all waves interleaved, so
//     with 3 fp8 tiles requested up front the FIRST tile of a workgroup completes after 7.2 us instead of 5.4 and the tiles then
//     finish 1.9 us apart (the conversion + MFMA rate of a CU), 3.54 ms per B = 8 step against 3.455 with 2 tiles: NS = 2 ships;
//   * the epilogue is done by ALL waves, one output per thread (fixed-order sum of the 8 partial tiles from double-buffered
//     LDS, 1/rms, fp8 scale, SwiGLU, residual, rounding, the next norm's hand-off), its operands requested one tile ahead
//     with unconditional (clamped) loads.  Round 3 had a ninth wave for it: its loads sat behind a branch, so the compiler
//     drained vmcnt(0) before every barrier and the whole workgroup waited ~1.6 us per tile for that one wave (ISA + timeline);
//   * one s_barrier per tile.
// Same arithmetic as the kernel above (same K partition when K / KS is a multiple of 8, same reduction order).
constexpr int SS_TP = 16 * 20;                 // partial tile in LDS: [b][i] at b*20 + i (16-byte aligned rows)

template <typename WT, int UNR, int SPT, int NS, bool SW8, bool TRACE = false>
__global__ __launch_bounds__(SK_THREADS) void skinny_stream_kernel(const bf16_t* __restrict__ x, const WT* __restrict__ W,
                                                                   const float* __restrict__ wscale, const bf16_t* res, void* outv,
                                                                   int MB, int N, int K, int ldx, int ldo, int tiled, int out_f32,
                                                                   SkinnyFuse fuse) {
    constexpr bool F8 = sizeof(WT) == 1;
    constexpr bool F16 = IsF16<WT>::v;
    constexpr int KS = F8 ? 64 : 32, CH = F8 ? 16 : 8, XL = F8 ? 2 : 1;
    constexpr int PER = UNR * SPT;                       // steps of one wave per tile
    constexpr int WV = SK_WAVES;
    static_assert(NS % SPT == 0 && NS >= 2, "the ring holds whole tiles");
    __shared__ __attribute__((aligned(16))) float red[2][SK_WAVES][SS_TP];
    __shared__ float inv_s[16];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int ntiles = N >> 4, nsteps = K / KS, G = gridDim.x;          // N % 16 == 0 (host-checked)
    const int my_tiles = (ntiles - (int)blockIdx.x + G - 1) / G;        // >= 1: the grid never exceeds the tile count
    const int total = my_tiles * SPT;                                   // weight sets of this workgroup
    const int s0 = wid * PER;
    if (wid == 0) sk_mark<TRACE>(fuse, 0);
    TEO_SK_SSQ_LOAD

    // ---- epilogue operands: thread (eb = conversation, ec = row of the tile) owns one output per tile; requested one tile ahead,
    // unconditionally (indices clamped: a thread without an output loads somebody else's operands and drops them)
    const int eb = tid >> 4, ec = tid & 15;
    const int ebc = min(eb, MB - 1);
    const bool e_on = SW8 ? (tid < 256 && ec < 8 && eb < MB) : (tid < 256 && eb < MB);
    // (absent operands read the activation buffer instead -- a pointer select, never a branch around a load: behind a branch the
    // compiler drains vmcnt(0), i.e. the whole weight ring, once per tile)
    const float* ws_p = wscale ? wscale : reinterpret_cast<const float*>(x);
    const bf16_t* res_p = (!SW8 && res) ? res : x;
    const bf16_t* g_p = (!SW8 && fuse.xg_out) ? reinterpret_cast<const bf16_t*>(fuse.next_g) : x;
    const int ws_on = wscale ? 1 : 0, res_on = (!SW8 && res) ? 1 : 0, g_on = (!SW8 && fuse.xg_out) ? 1 : 0;
    float sc_n, sc_n2, rv_n, gv_n;
#define TEO_SS_PREFETCH(T)                                                                                     \
    {                                                                                                          \
        const int np = min((T), ntiles - 1) * 16 + (SW8 ? (ec & 7) : ec);                                      \
        sc_n = ws_p[np * ws_on];                                                                               \
        sc_n2 = ws_p[(np + (SW8 ? 8 : 0)) * ws_on];                                                            \
        rv_n = h2f<F16>(res_p[((long long)ebc * ldo + np) * res_on]);                                              \
        gv_n = h2f<F16>(g_p[np * g_on]);                                                                           \
    }
    TEO_SS_PREFETCH((int)blockIdx.x)

    // ---- the slice's activations, once
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
    // ---- weight ring: set g = (tile g / SPT of this workgroup, steps (g % SPT) * UNR ..).  Sets past the end re-read the
    // activation buffer (L2-resident, always mapped) so that every load stays unconditional and costs no HBM traffic.
    const long long pstep = tiled ? 64 * CH : KS;
    const long long tstride = (long long)nsteps * (64 * CH);
    u32x4 w[NS][UNR];
#define TEO_SS_LOADW(SLOT, GI)                                                                                 \
    {                                                                                                          \
        const int g_ = (GI);                                                                                   \
        const int tq = (int)blockIdx.x + (g_ / SPT) * G, ls_ = g_ % SPT;                                       \
        const WT* wt = tiled ? W + (long long)tq * tstride + lane * CH                                         \
                             : W + (long long)(tq * 16 + fr) * K + fg * CH;                                    \
        const bool live_ = g_ < total;                                                                         \
        _Pragma("unroll") for (int u = 0; u < UNR; ++u) {                                                      \
            const WT* pw = wt + (long long)min(s0 + ls_ * UNR + u, nsteps - 1) * pstep;                        \
            w[SLOT][u] = sk_ldw<true>(live_ ? pw : reinterpret_cast<const WT*>(x) + lane * CH);                \
        }                                                                                                      \
    }
#pragma unroll
    for (int r = 0; r < NS - 1; ++r) TEO_SS_LOADW(r, r)
    TEO_SK_SSQ_REDUCE
#pragma unroll
    for (int i = 0; i < PER; ++i)                        // steps past the end of K contribute nothing
        if (s0 + i >= nsteps) {
#pragma unroll
            for (int j = 0; j < XL; ++j) xr[i][j] = (u32x4){0u, 0u, 0u, 0u};
        }
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    int par = 0, ntrace = 1;
    float sc = 1.f, sc2 = 1.f, rv = 0.f, gv = 0.f;       // epilogue operands of the tile being multiplied
    for (int g0 = 0; g0 < total; g0 += NS) {
#pragma unroll
        for (int r = 0; r < NS; ++r) {
            const int g = g0 + r;
            const int ls = r % SPT;                      // NS % SPT == 0: the set's position inside its tile is static
            if (ls == 0) {
                // first set of a tile: take the operands requested one tile ago and request the next tile's -- BEFORE this step's
                // weight set goes out, so that they are older in the (in-order) vmcnt queue than every set still in flight when
                // the epilogue needs them
                sc = sc_n; sc2 = sc_n2; rv = rv_n; gv = gv_n;
                TEO_SS_PREFETCH((int)blockIdx.x + (g / SPT + 1) * G)
            }
            TEO_SS_LOADW((r + NS - 1) % NS, g + NS - 1)
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int i = ls * UNR + u;
                if (F8) {
                    acc = mfma16<F16>(fp8x8_to_bf16x8(w[r][u].x, w[r][u].y),
                                                                  __builtin_bit_cast(bf16x8, xr[i][0]), acc);
                    acc = mfma16<F16>(fp8x8_to_bf16x8(w[r][u].z, w[r][u].w),
                                                                  __builtin_bit_cast(bf16x8, xr[i][XL - 1]), acc);
                } else {
                    acc = mfma16<F16>(__builtin_bit_cast(bf16x8, w[r][u]),
                                                                  __builtin_bit_cast(bf16x8, xr[i][0]), acc);
                }
            }
            if (ls == SPT - 1 && g < total) {            // tile done (uniform): lane holds out[b = fr][rows fg*4 .. +3]
                const int t = (int)blockIdx.x + (g / SPT) * G;
                *reinterpret_cast<f32x4*>(&red[par][wid][fr * 20 + fg * 4]) = acc;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");           // the partial tiles are read AFTER the barrier
                if (wid == 0 && ntrace < 8) sk_mark<TRACE>(fuse, ntrace++);
                acc = (f32x4){0.f, 0.f, 0.f, 0.f};
                // ---- epilogue: one output per thread, partial tiles summed in wave order
                const float inv = fuse.ssq_in ? inv_s[ebc] : 1.f;
                if (SW8) {
                    float gsum = 0.f, usum = 0.f;        // rows 0..7 of the tile are gate rows, 8..15 their up rows
#pragma unroll
                    for (int wv = 0; wv < SK_WAVES; ++wv) {
                        gsum += red[par][wv][ebc * 20 + (ec & 7)];
                        usum += red[par][wv][ebc * 20 + (ec & 7) + 8];
                    }
                    gsum *= inv; usum *= inv;
                    if (wscale) { gsum *= sc; usum *= sc2; }
                    const float v = silu(gsum) * usum;
                    if (e_on) {
                        const long long at = (long long)eb * ldo + t * 8 + ec;
                        if (out_f32) reinterpret_cast<float*>(outv)[at] = v;
                        else reinterpret_cast<bf16_t*>(outv)[at] = f2h<F16>(v);
                    }
                } else {
                    float v = 0.f;
#pragma unroll
                    for (int wv = 0; wv < SK_WAVES; ++wv) v += red[par][wv][ebc * 20 + ec];
                    v *= inv;
                    if (wscale) v *= sc;
                    if (res) v += rv;
                    const long long at = (long long)ebc * ldo + t * 16 + ec;
                    float sq = 0.f;
                    if (out_f32) { if (e_on) reinterpret_cast<float*>(outv)[at] = v; }
                    else {
                        const bf16_t hb = f2h<F16>(v);
                        if (e_on) reinterpret_cast<bf16_t*>(outv)[at] = hb;
                        if (fuse.xg_out) {
                            // what the NEXT layer's RMSNorm needs: bf16(h * g_next) and this tile's share of sum(h^2) per row
                            const float h = h2f<F16>(hb);
                            if (e_on) fuse.xg_out[at] = f2h<F16>(h * gv);
                            sq = h * h;
                        }
                    }
                    if (fuse.xg_out) {
                        sq = group_sum<16>(sq);
                        if (ec == 0 && e_on) fuse.ssq_out[(long long)eb * ntiles + t] = sq;
                    }
                }
                par ^= 1;
                if (wid == 0 && ntrace < 16) sk_mark<TRACE>(fuse, 8 + (ntrace - 1));
            }
        }
    }
#undef TEO_SS_LOADW
#undef TEO_SS_PREFETCH
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
    if (out_dtype == TEO_F16 || (flags & TEO_GEMM_F16)) fuse.f16 = true;      // the runtime sets fuse.f16 itself
    TEO_CHECK_ARG(!(fuse.f16 && w_fp8), "skinny_gemm: fp8 weights go with bfloat16 activations (their power-of-two row scales are exact in bf16 only)");
    TEO_CHECK_ARG(!(swiglu && res), "skinny_gemm: SWIGLU16 takes no residual");
    TEO_CHECK_ARG(!norm_w || (K <= 16384 && (reinterpret_cast<uintptr_t>(norm_w) & 15) == 0), "skinny_gemm: fused norm needs K <= 16384 and an aligned weight");
    // row tiles per workgroup (8 waves = RT row tiles x 8/RT K slices)
    TEO_CHECK_ARG(!(fuse.xg_out && (swiglu || out_dtype == TEO_F32 || !fuse.next_g || !fuse.ssq_out)),
                  "skinny_gemm: the norm hand-off needs a plain bf16 output, next_g and ssq_out");
    TEO_CHECK_ARG(!(fuse.ssq_in && norm_w), "skinny_gemm: ssq_in and norm_w are exclusive");
    TEO_CHECK_ARG(!fuse.ssq_in || fuse.nparts >= 1, "skinny_gemm: nparts %d", fuse.nparts);
    int rt = tune().skinny_tiles;
    if (rt == 0 || fuse.xg_out) rt = 1;   // measured: one row tile per workgroup (most waves in flight) wins at every N
    const int sw8 = (flags & TEO_GEMM_SWIGLU8) ? 1 : 0;   // gate/up interleaved in blocks of 8 rows: a pair fits one tile
    if (swiglu && !sw8 && rt < 2) rt = 2; // 16-row interleave: the gate tile and its up tile meet in the epilogue
    const int ldr = ldo, of = out_dtype == TEO_F32;
    // streaming form (persistent workgroups, activations in registers): K <= 4096, whole 16-row tiles, one row tile at a time
    {
        const int nsteps = K / (w_fp8 ? 64 : 32), ntiles = N / 16;
        const auto al = [](const void* p, uintptr_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; };
        bool ok = tune().skinny_stream != 0 && !norm_w && !(flags & TEO_GEMM_SWIGLU16) && (tune().skinny_tiles == 0 || tune().skinny_tiles == 1) && N % 16 == 0 &&
                  nsteps <= (w_fp8 ? 64 : 128) && al(x, 16);
        // auto: where it measures faster than one tile per workgroup -- at least two tiles per workgroup, and fp8 weights or more
        // than 8 rows (bf16 at <= 8 rows: a tie, the duplicate activation rows of the tile kernel coalesce)
        if (ok && tune().skinny_stream == 1) ok = ntiles >= 512 && (w_fp8 || MB > 8);
        if (ok) {
            const int cus = device_cu_count();
            const int per_cu = tune().skinny_grid > 0 ? tune().skinny_grid : 1;
            const int grid = std::max(1, std::min(ntiles, per_cu * (cus > 0 ? cus : 256)));
#define TEO_SS(WW, UN, SP, NSV, SW)                                                                         \
            TEO_KLAUNCH((skinny_stream_kernel<WW, UN, SP, NSV, SW>), grid, SK_THREADS, 0, st, (const bf16_t*)x, (const WW*)W, wscale, (const bf16_t*)res, \
                        out, MB, N, K, ldx, ldo, tiled, of, fuse)
            if (tune().skinny_ring == 0) {
                if (w_fp8)         { if (sw8) TEO_SS(fp8_t, 8, 1, 2, true); else TEO_SS(fp8_t, 8, 1, 2, false); }
                else if (fuse.f16) { if (sw8) TEO_SS(f16_t, 8, 2, 2, true); else TEO_SS(f16_t, 8, 2, 2, false); }
                else               { if (sw8) TEO_SS(bf16_t, 8, 2, 2, true); else TEO_SS(bf16_t, 8, 2, 2, false); }
            } else {
                if (w_fp8)         { if (sw8) TEO_SS(fp8_t, 8, 1, 3, true); else TEO_SS(fp8_t, 8, 1, 3, false); }
                else if (fuse.f16) { if (sw8) TEO_SS(f16_t, 8, 2, 4, true); else TEO_SS(f16_t, 8, 2, 4, false); }
                else               { if (sw8) TEO_SS(bf16_t, 8, 2, 4, true); else TEO_SS(bf16_t, 8, 2, 4, false); }
            }
#undef TEO_SS
            note_kernel("skinny_stream");
            TEO_LAUNCH_CHECK("skinny_gemm (stream)");
            return TEO_OK;
        }
    }
    const int blocks = cdiv(N, 16 * rt);
    const size_t dyn = norm_w ? (size_t)K * 2 : 0;
    // long K slices (down projection, K = 11008: 21.5 fp8 / 43 bf16 steps per wave): 8 steps per register set = 16 weight + their
    // activation loads in flight per wave instead of 8 -- the launch is latency-bound with 64 KB of weights per CU in flight
    // (3.2 TB/s, tools/skinny_probe.py); fp8 takes 6 steps (its activation fragments are twice the weights: 8 steps spill); 150-200
    // VGPRs: one workgroup per CU, which is all N = 4096 has anyway
    // 16 waves per workgroup (round 6): the launches with about one 16-row tile per CU (o, down: N = 4096 -> 256 workgroups) are bound by
    // how many weight requests a CU has in flight, not by its waves' arithmetic -- 16 K slices put twice the requests out from the first cycle
    {
        const int steps16 = K / (w_fp8 ? 64 : 32) / 16;
        const bool w16 = !swiglu && !norm_w && rt == 1 && (tune().skinny_nt != 0) && steps16 >= 2 &&
                         (tune().skinny_waves == 16 || (tune().skinny_waves == 0 && false));
        if (w16) {
#define TEO_SK16(WW, UN)                                                                                    \
            TEO_KLAUNCH((skinny_gemm_kernel<WW, UN, true, false, false, false, 16>), blocks, 1024, 0, st, (const WW*)W, (const bf16_t*)x, MB, N, K, ldx,  \
                        tiled, rt, wscale, (const bf16_t*)res, (const bf16_t*)nullptr, eps, out, ldo, ldr, of, fuse, sw8)
            // steps per register set: two sets cover the wave's whole K slice where the registers allow (every request of the launch out at once)
            // (fp8: at most 4 -- its activation fragments are twice the weights, UNR = 6 spills 67 VGPRs under the 128-register cap of a
            // 1024-thread workgroup: tools/kernel_meta.py)
            const int un = steps16 <= 4 ? 2 : ((steps16 <= 8 || w_fp8) ? 4 : 6);
            if (w_fp8)         { if (un == 2) TEO_SK16(fp8_t, 2); else TEO_SK16(fp8_t, 4); }
            else if (fuse.f16) { if (un == 2) TEO_SK16(f16_t, 2); else if (un == 4) TEO_SK16(f16_t, 4); else TEO_SK16(f16_t, 6); }
            else               { if (un == 2) TEO_SK16(bf16_t, 2); else if (un == 4) TEO_SK16(bf16_t, 4); else TEO_SK16(bf16_t, 6); }
#undef TEO_SK16
            note_kernel("skinny_gemm_w16");
            TEO_LAUNCH_CHECK("skinny_gemm");
            return TEO_OK;
        }
    }
    const int steps_per_wave = K / (w_fp8 ? 64 : 32) / (SK_WAVES / rt);
    const bool unr8 = !swiglu && !norm_w && (tune().skinny_nt != 0) && (tune().skinny_unr == 8 || (tune().skinny_unr == 0 && steps_per_wave >= 16 && blocks <= 2 * std::max(device_cu_count(), 1)));
    if (unr8) {
        if (w_fp8) TEO_KLAUNCH((skinny_gemm_kernel<fp8_t, 6, true, false, false>), blocks, SK_THREADS, 0, st, (const fp8_t*)W, (const bf16_t*)x, MB, N, K, ldx, tiled, rt,
                               wscale, (const bf16_t*)res, (const bf16_t*)nullptr, eps, out, ldo, ldr, of, fuse, sw8);
        else if (fuse.f16) TEO_KLAUNCH((skinny_gemm_kernel<f16_t, 8, true, false, false>), blocks, SK_THREADS, 0, st, (const f16_t*)W, (const bf16_t*)x, MB, N, K, ldx, tiled, rt,
                               wscale, (const bf16_t*)res, (const bf16_t*)nullptr, eps, out, ldo, ldr, of, fuse, sw8);
        else       TEO_KLAUNCH((skinny_gemm_kernel<bf16_t, 8, true, false, false>), blocks, SK_THREADS, 0, st, (const bf16_t*)W, (const bf16_t*)x, MB, N, K, ldx, tiled, rt,
                               wscale, (const bf16_t*)res, (const bf16_t*)nullptr, eps, out, ldo, ldr, of, fuse, sw8);
        note_kernel("skinny_gemm_u8");
        TEO_LAUNCH_CHECK("skinny_gemm");
        return TEO_OK;
    }
#define TEO_SK(WW, NTV, SW, NM)                                                                            \
    TEO_KLAUNCH((skinny_gemm_kernel<WW, 4, NTV, SW, NM>), blocks, SK_THREADS, dyn, st,                     \
        (const WW*)W, (const bf16_t*)x, MB, N, K, ldx, tiled, rt, wscale, (const bf16_t*)res, (const bf16_t*)norm_w, eps, out, ldo, ldr, \
        of, fuse, sw8)
#define TEO_SK_N(WW, NTV, SW) if (norm_w) { TEO_SK(WW, NTV, SW, true); } else { TEO_SK(WW, NTV, SW, false); }
#define TEO_SK_F(WW, NTV) if (swiglu) { TEO_SK_N(WW, NTV, true) } else { TEO_SK_N(WW, NTV, false) }
    if (w_fp8)         { if ((tune().skinny_nt != 0)) { TEO_SK_F(fp8_t, true) } else { TEO_SK_F(fp8_t, false) } }
    else if (fuse.f16) { TEO_SK_F(f16_t, true) }              // (non-temporal weight loads only: one instantiation set for the second format)
    else               { if ((tune().skinny_nt != 0)) { TEO_SK_F(bf16_t, true) } else { TEO_SK_F(bf16_t, false) } }
#undef TEO_SK_F
#undef TEO_SK_N
#undef TEO_SK
    note_kernel("skinny_gemm");
    TEO_LAUNCH_CHECK("skinny_gemm");
    return TEO_OK;
}

}  // namespace teo
