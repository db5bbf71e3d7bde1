// Decode GEMV: y[N] = W[N,K] . f(x) for ONE activation row (q_len == 1 decode step).
//
// HBM-bound: every weight byte is read exactly once per token, so the kernel is a pure weight stream:
//   - no MFMA, no LDS staging of W (each byte is used once; an LDS round trip would be pure overhead);
//   - 16-byte non-temporal loads (the stream must not evict x / the KV cache from L2), R rows per wave in
//     flight at once, K loop unrolled so >= 8 loads per lane are outstanding before the first use;
//   - x (after the optional fused RMSNorm) is staged once per workgroup in LDS as fp32 and re-read with
//     conflict-free ds_read_b128;
//   - epilogues fused: residual add, SwiGLU on the interleaved-16 gate/up layout, fp32 logits.
// Roofline: HBM (8 TB/s spec); algorithmic bytes per launch = N*K*sizeof(T) (+ K + N elements, negligible).
#include "common.h"

namespace teo {

constexpr int GV_WAVES = 4;           // waves per workgroup
constexpr int GV_THREADS = GV_WAVES * 64;

template <typename T> struct Vec16;   // 16 bytes of T -> floats
template <> struct Vec16<bf16_t> {
    static constexpr int N = 8;
    __device__ static __forceinline__ void cvt(const uint4& r, float* f) {
        f[0] = __uint_as_float(r.x << 16); f[1] = __uint_as_float(r.x & 0xffff0000u);
        f[2] = __uint_as_float(r.y << 16); f[3] = __uint_as_float(r.y & 0xffff0000u);
        f[4] = __uint_as_float(r.z << 16); f[5] = __uint_as_float(r.z & 0xffff0000u);
        f[6] = __uint_as_float(r.w << 16); f[7] = __uint_as_float(r.w & 0xffff0000u);
    }
};
template <> struct Vec16<float> {
    static constexpr int N = 4;
    __device__ static __forceinline__ void cvt(const uint4& r, float* f) {
        f[0] = __uint_as_float(r.x); f[1] = __uint_as_float(r.y); f[2] = __uint_as_float(r.z); f[3] = __uint_as_float(r.w);
    }
};

__device__ __forceinline__ uint4 ld_nt16(const void* p) {
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    const v4u v = __builtin_nontemporal_load(reinterpret_cast<const v4u*>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}

// Tunables (teo_tune_set): which <R,U,DB,PF> instantiation runs, non-temporal loads on/off, workgroup cap.
struct GemvTune { int variant = -1; int nt = 1; int max_blocks = 1024; };
static GemvTune g_tune;
int gemv_tune_set(const char* key, int value) {
    if (!strcmp(key, "gemv_variant")) g_tune.variant = value;
    else if (!strcmp(key, "gemv_nt")) g_tune.nt = value;
    else if (!strcmp(key, "gemv_max_blocks")) g_tune.max_blocks = value;
    else return -1;
    return 0;
}

// R rows per wave per pass, U 16-byte chunks per lane per row per K-iteration (R*U loads issued back to back).
//   DB: the next iteration's R*U loads are issued before the current ones are consumed (register double buffer);
//   PF: the first R*U loads are issued before the x staging / RMSNorm prologue (the weight stream does not depend on x);
//   NT: non-temporal weight loads.
// SWIGLU: rows come in (gate, up) pairs 16 apart inside 32-row blocks.
template <typename T, typename TO, int R, int U, bool DB, bool PF, bool NT, bool SWIGLU>
__global__ __launch_bounds__(GV_THREADS) void gemv_kernel(const T* __restrict__ x, const T* __restrict__ W,
                                                          const T* __restrict__ norm_w, const T* __restrict__ res,
                                                          TO* __restrict__ y, int N, int K, float eps) {
    extern __shared__ __attribute__((aligned(16))) float xs[];     // [K] fp32 (+8 floats of reduction scratch)
    constexpr int VE = Vec16<T>::N;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int nchunk = K / VE;                    // 16-byte chunks per row
    const int kpad = (K + 3) & ~3;
    float* red = xs + kpad;
    const int wave_global = blockIdx.x * GV_WAVES + wid;
    const int nwaves = gridDim.x * GV_WAVES;
    const int ngroups = SWIGLU ? (N / 2 + (R / 2) - 1) / (R / 2) : (N + R - 1) / R;

    auto row_of = [&](int grp, int r) -> long long {
        if (SWIGLU) {
            const int j = min(grp * (R / 2) + (r >> 1), N / 2 - 1);     // output column
            return (long long)((j >> 4) * 32 + (j & 15) + ((r & 1) ? 16 : 0));
        }
        return (long long)min(grp * R + r, N - 1);
    };
    // full = every lane's chunk is in range (no per-load exec masking in the steady-state loop)
    auto issue = [&](uint4 (&w)[U][R], int grp, int c0, bool full) {
        if (full) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const T* p = W + row_of(grp, r) * K + (long long)(c0 + u * 64 + lane) * VE;
                    w[u][r] = NT ? ld_nt16(p) : *reinterpret_cast<const uint4*>(p);
                }
            }
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int c = c0 + u * 64 + lane;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const T* p = W + row_of(grp, r) * K + (long long)c * VE;
                    w[u][r] = (c < nchunk) ? (NT ? ld_nt16(p) : *reinterpret_cast<const uint4*>(p)) : make_uint4(0, 0, 0, 0);
                }
            }
        }
    };

    // ---- x first (the weight stream does not depend on it, but vmcnt retires in order: loads issued after the
    // weights would have to wait for them), then the first weight block, then the prologue under the weight latency
    constexpr int XPT = 6;                        // 16-byte x chunks a thread can hold: K <= 6*256*VE (12288 for bf16)
    uint4 xr[XPT], nr[XPT];
    const bool x_in_regs = nchunk <= XPT * GV_THREADS;
    if (x_in_regs) {
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
            const int c = tid + i * GV_THREADS;
            xr[i] = (c < nchunk) ? *reinterpret_cast<const uint4*>(x + (long long)c * VE) : make_uint4(0, 0, 0, 0);
        }
        if (norm_w) {                             // same round trip as x: no second dependent global-load phase
#pragma unroll
            for (int i = 0; i < XPT; ++i) {
                const int c = tid + i * GV_THREADS;
                nr[i] = (c < nchunk) ? *reinterpret_cast<const uint4*>(norm_w + (long long)c * VE) : make_uint4(0, 0, 0, 0);
            }
        }
    }
    uint4 wa[U][R];
    int grp = wave_global;
    constexpr int STEP0 = 64 * U;
    const bool pf = PF && grp < ngroups && nchunk >= STEP0;   // unconditional loads only: masked ones get drained at once
    if (pf) issue(wa, grp, 0, true);

    float ss = 0.f;
    if (x_in_regs) {
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
            const int c = tid + i * GV_THREADS;
            if (c < nchunk) {
                float f[VE];
                Vec16<T>::cvt(xr[i], f);
#pragma unroll
                for (int e = 0; e < VE; ++e) ss = fmaf(f[e], f[e], ss);
            }
        }
        if (norm_w) {
            const float rr = rsqrtf(block_sum<GV_THREADS>(ss, red) / K + eps);
#pragma unroll
            for (int i = 0; i < XPT; ++i) {
                const int c = tid + i * GV_THREADS;
                if (c < nchunk) {
                    float f[VE], g[VE];
                    Vec16<T>::cvt(xr[i], f);
                    Vec16<T>::cvt(nr[i], g);
#pragma unroll
                    for (int e = 0; e < VE; ++e) xs[c * VE + e] = Elem<T>::round(f[e] * rr * g[e]);
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < XPT; ++i) {
                const int c = tid + i * GV_THREADS;
                if (c < nchunk) {
                    float f[VE];
                    Vec16<T>::cvt(xr[i], f);
#pragma unroll
                    for (int e = 0; e < VE; ++e) xs[c * VE + e] = f[e];
                }
            }
        }
    } else {
        for (int c = tid; c < nchunk; c += GV_THREADS) {
            const uint4 raw = *reinterpret_cast<const uint4*>(x + (long long)c * VE);
            float f[VE];
            Vec16<T>::cvt(raw, f);
#pragma unroll
            for (int e = 0; e < VE; ++e) { xs[c * VE + e] = f[e]; ss = fmaf(f[e], f[e], ss); }
        }
        if (norm_w) {
            const float rr = rsqrtf(block_sum<GV_THREADS>(ss, red) / K + eps);
            for (int c = tid; c < nchunk; c += GV_THREADS) {
                const uint4 raw = *reinterpret_cast<const uint4*>(norm_w + (long long)c * VE);
                float f[VE];
                Vec16<T>::cvt(raw, f);
#pragma unroll
                for (int e = 0; e < VE; ++e) xs[c * VE + e] = Elem<T>::round(xs[c * VE + e] * rr * f[e]);
            }
        }
    }
    __syncthreads();

    auto consume = [&](const uint4 (&w)[U][R], int c0, float (&acc)[R], bool full) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int c = c0 + u * 64 + lane;
            float xv[VE];
            if (full || c < nchunk) {
#pragma unroll
                for (int e = 0; e < VE; e += 4) {
                    const float4 t = *reinterpret_cast<const float4*>(xs + c * VE + e);
                    xv[e] = t.x; xv[e + 1] = t.y; xv[e + 2] = t.z; xv[e + 3] = t.w;
                }
            } else {
#pragma unroll
                for (int e = 0; e < VE; ++e) xv[e] = 0.f;
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                float f[VE];
                Vec16<T>::cvt(w[u][r], f);
#pragma unroll
                for (int e = 0; e < VE; ++e) acc[r] = fmaf(f[e], xv[e], acc[r]);
            }
        }
    };

    constexpr int STEP = 64 * U;
    bool have = pf;                              // wa already holds block 0 of the first group
    for (; grp < ngroups; grp += nwaves) {
        float acc[R];
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = 0.f;
        if (DB) {
            uint4 wb[U][R];
            if (!have) issue(wa, grp, 0, false);
            for (int c0 = 0; c0 < nchunk; c0 += 2 * STEP) {
                const int c1 = c0 + STEP, c2 = c0 + 2 * STEP;
                if (c1 < nchunk) issue(wb, grp, c1, false);
                consume(wa, c0, acc, false);
                if (c2 < nchunk) issue(wa, grp, c2, false);
                if (c1 < nchunk) consume(wb, c1, acc, false);
            }
            have = false;
        } else {
            int c0 = 0;
            if (have) { consume(wa, 0, acc, true); c0 = STEP; have = false; }
            const int cfull = (nchunk / STEP) * STEP;          // steady state: no bounds checks, no exec masking
            for (; c0 < cfull; c0 += STEP) {
                issue(wa, grp, c0, true);
                consume(wa, c0, acc, true);
            }
            if (c0 < nchunk) {
                issue(wa, grp, c0, false);
                consume(wa, c0, acc, false);
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = wave_sum(acc[r]);
        if (lane == 0) {
            if (SWIGLU) {
#pragma unroll
                for (int p = 0; p < R / 2; ++p) {
                    const int j = grp * (R / 2) + p;
                    if (j < N / 2) Elem<TO>::st(y + j, silu(acc[2 * p]) * acc[2 * p + 1]);
                }
            } else {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int n = grp * R + r;
                    if (n < N) {
                        float v = acc[r];
                        if (res) v += Elem<T>::ld(res + n);
                        Elem<TO>::st(y + n, v);
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Split-K form for few, long rows (o / down projections: N = 4096): a workgroup owns R rows and its 4 waves take
// interleaved 1-KiB chunks of every row, so 4x more waves stream than with one wave per row group; partial sums meet
// in LDS.  x is read by each wave for its own chunks only (no norm on these layers -> no full-vector prologue).
// ------------------------------------------------------------------------------------------------
template <typename T, typename TO, int R, int U, bool NT>
__global__ __launch_bounds__(GV_THREADS) void gemv_splitk_kernel(const T* __restrict__ x, const T* __restrict__ W,
                                                                 const T* __restrict__ res, TO* __restrict__ y, int N,
                                                                 int K) {
    constexpr int VE = Vec16<T>::N;
    __shared__ float part[GV_WAVES][R];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int nchunk = K / VE;
    const int row0 = blockIdx.x * R;
    float acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = 0.f;
    // chunk index for (iteration it, unroll u): c = (it*U + u) * 256 + wid*64 + lane
    const T* wrow[R];
#pragma unroll
    for (int r = 0; r < R; ++r) wrow[r] = W + (long long)min(row0 + r, N - 1) * K;
    const int cfull = (nchunk / (256 * U)) * (256 * U);       // steady state: every lane in range, no exec masking
    int cb = 0;
    for (; cb < cfull; cb += 256 * U) {
        uint4 xr[U];
        uint4 w[U][R];
#pragma unroll
        for (int u = 0; u < U; ++u) xr[u] = *reinterpret_cast<const uint4*>(x + (long long)(cb + u * 256 + wid * 64 + lane) * VE);
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const T* p = wrow[r] + (long long)(cb + u * 256 + wid * 64 + lane) * VE;
                w[u][r] = NT ? ld_nt16(p) : *reinterpret_cast<const uint4*>(p);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float xv[VE];
            Vec16<T>::cvt(xr[u], xv);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                float f[VE];
                Vec16<T>::cvt(w[u][r], f);
#pragma unroll
                for (int e = 0; e < VE; ++e) acc[r] = fmaf(f[e], xv[e], acc[r]);
            }
        }
    }
    for (; cb < nchunk; cb += 256 * U) {
        uint4 xr[U];
        uint4 w[U][R];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int c = cb + u * 256 + wid * 64 + lane;
            xr[u] = (c < nchunk) ? *reinterpret_cast<const uint4*>(x + (long long)c * VE) : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int c = cb + u * 256 + wid * 64 + lane;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const T* p = wrow[r] + (long long)c * VE;
                w[u][r] = (c < nchunk) ? (NT ? ld_nt16(p) : *reinterpret_cast<const uint4*>(p)) : make_uint4(0, 0, 0, 0);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float xv[VE];
            Vec16<T>::cvt(xr[u], xv);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                float f[VE];
                Vec16<T>::cvt(w[u][r], f);
#pragma unroll
                for (int e = 0; e < VE; ++e) acc[r] = fmaf(f[e], xv[e], acc[r]);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = wave_sum(acc[r]);
    if (lane == 0) {
#pragma unroll
        for (int r = 0; r < R; ++r) part[wid][r] = acc[r];
    }
    __syncthreads();
    if (tid < R) {
        const int n = row0 + tid;
        if (n < N) {
            float v = (part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid]);
            if (res) v += Elem<T>::ld(res + n);
            Elem<TO>::st(y + n, v);
        }
    }
}

template <typename T, typename TO, int R, int U>
static int gemv_launch_splitk(const void* x, const void* W, const void* res, void* y, int N, int K, hipStream_t st) {
    const int blocks = cdiv(N, R);
    if (g_tune.nt)
        gemv_splitk_kernel<T, TO, R, U, true><<<blocks, GV_THREADS, 0, st>>>((const T*)x, (const T*)W, (const T*)res, (TO*)y, N, K);
    else
        gemv_splitk_kernel<T, TO, R, U, false><<<blocks, GV_THREADS, 0, st>>>((const T*)x, (const T*)W, (const T*)res, (TO*)y, N, K);
    TEO_LAUNCH_CHECK("gemv_splitk");
    return TEO_OK;
}

template <typename T, typename TO, int R, int U, bool DB, bool PF>
static int gemv_launch_ru(const void* x, const void* W, const void* norm_w, const void* res, void* y, int N, int K,
                          float eps, bool swiglu, hipStream_t st) {
    const int ngroups = swiglu ? cdiv(N / 2, R / 2) : cdiv(N, R);
    int blocks = cdiv(ngroups, GV_WAVES);
    if (blocks > g_tune.max_blocks) blocks = g_tune.max_blocks;
    const size_t lds = (size_t)(((K + 3) & ~3) + 8) * sizeof(float);
#define TEO_GV(NTV, SW)                                                                                              \
    gemv_kernel<T, TO, R, U, DB, PF, NTV, SW><<<blocks, GV_THREADS, lds, st>>>((const T*)x, (const T*)W, (const T*)norm_w, \
                                                                                (const T*)res, (TO*)y, N, K, eps)
    if (g_tune.nt) { if (swiglu) TEO_GV(true, true); else TEO_GV(true, false); }
    else           { if (swiglu) TEO_GV(false, true); else TEO_GV(false, false); }
#undef TEO_GV
    TEO_LAUNCH_CHECK("gemv");
    return TEO_OK;
}

template <typename T, typename TO>
static int gemv_launch(const void* x, const void* W, const void* norm_w, const void* res, void* y, int N, int K, float eps,
                       bool swiglu, hipStream_t st) {
    // few long rows without a fused norm (o / down projections): split-K workgroups, 2 rows each (measured best)
    if (!swiglu && norm_w == nullptr && N <= 8192) return gemv_launch_splitk<T, TO, 2, 2>(x, W, res, y, N, K, st);
    // >= 4 KiB contiguous per row per step streams ~7 % faster than 2 KiB (tools/stream_probe.py, profiles/r01_gemv_variant_sweep.txt)
    // and the first weight block is issued (unconditionally) before the RMSNorm prologue so that it hides under the latency
    return gemv_launch_ru<T, TO, 2, 4, false, true>(x, W, norm_w, res, y, N, K, eps, swiglu, st);
}

// tuning sweep: bf16 -> bf16 only
static int gemv_launch_variant(int v, const void* x, const void* W, const void* norm_w, const void* res, void* y, int N,
                               int K, float eps, bool swiglu, hipStream_t st) {
#define TEO_V(id, R, U, DB, PF) \
    case id: return gemv_launch_ru<bf16_t, bf16_t, R, U, DB, PF>(x, W, norm_w, res, y, N, K, eps, swiglu, st)
    switch (v) {
        TEO_V(0, 4, 2, false, false);
        TEO_V(1, 4, 2, false, true);
        TEO_V(2, 4, 2, true, true);
        TEO_V(3, 2, 2, false, true);
        TEO_V(4, 2, 4, false, true);
        TEO_V(5, 4, 1, false, true);
        TEO_V(6, 8, 1, false, true);
        TEO_V(7, 4, 4, false, true);
        TEO_V(8, 2, 2, true, true);
        TEO_V(9, 8, 2, false, true);
        TEO_V(10, 2, 1, false, true);
        TEO_V(11, 2, 8, false, true);
        TEO_V(18, 2, 4, false, false);
        TEO_V(19, 4, 4, false, false);
        TEO_V(20, 2, 8, false, false);
        TEO_V(21, 2, 2, false, false);
        default: break;
    }
#undef TEO_V
    if (v >= 12 && v <= 17 && !swiglu && norm_w == nullptr) {
        switch (v) {
            case 12: return gemv_launch_splitk<bf16_t, bf16_t, 4, 1>(x, W, res, y, N, K, st);
            case 13: return gemv_launch_splitk<bf16_t, bf16_t, 4, 2>(x, W, res, y, N, K, st);
            case 14: return gemv_launch_splitk<bf16_t, bf16_t, 2, 2>(x, W, res, y, N, K, st);
            case 15: return gemv_launch_splitk<bf16_t, bf16_t, 2, 4>(x, W, res, y, N, K, st);
            case 16: return gemv_launch_splitk<bf16_t, bf16_t, 8, 1>(x, W, res, y, N, K, st);
            case 17: return gemv_launch_splitk<bf16_t, bf16_t, 1, 4>(x, W, res, y, N, K, st);
        }
    }
    if (v >= 12 && v <= 17) return gemv_launch_ru<bf16_t, bf16_t, 4, 2, false, false>(x, W, norm_w, res, y, N, K, eps, swiglu, st);
    set_error("gemv: unknown variant %d", v);
    return TEO_ERR_ARG;
}

// ------------------------------------------------------------------------------------------------
// Decode QKV projection with the RoPE rotation and the KV-cache append fused into the epilogue.
//   rows of Wqkv: [q: H*hd | k: Hk*hd | v: Hk*hd].  A wave owns two rotation pairs (i, i + hd/2) of one q/k head
//   (4 rows), or 4 consecutive v rows.  q is written rotated to qout[H*hd]; k rotated to K cache[hk][pos][:];
//   v to V cache[hk][pos][:] and V^T cache[hk][:][pos].  pos (= cache slot = rotary position) is read from device
//   memory so the launch can be replayed from a hipGraph.  Rounding points are those of the unfused path:
//   round(linear) -> rotate in fp32 -> round.
// ------------------------------------------------------------------------------------------------
template <typename T, bool NT>
__global__ __launch_bounds__(GV_THREADS) void gemv_qkv_rope_kernel(const T* __restrict__ x, const T* __restrict__ W,
                                                                   const T* __restrict__ norm_w, T* __restrict__ qout,
                                                                   const float* __restrict__ cs, const float* __restrict__ sn,
                                                                   const int* __restrict__ d_pos, T* __restrict__ kc,
                                                                   T* __restrict__ vc, T* __restrict__ vtc, int S_max, int H,
                                                                   int Hk, int hd, int K, float eps) {
    extern __shared__ __attribute__((aligned(16))) float xs[];
    constexpr int VE = Vec16<T>::N;
    constexpr int R = 2, U = 4;                         // one rotation pair (rows i, i + hd/2) or two v rows per wave
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int nchunk = K / VE, kpad = (K + 3) & ~3;
    float* red = xs + kpad;
    const int half = hd >> 1;
    const int qk_groups = (H + Hk) * half;              // rotation pairs
    const int v_groups = (Hk * hd) / 2;
    const int ngroups = qk_groups + v_groups;
    const int nwaves = gridDim.x * GV_WAVES;

    const int pos = *d_pos;                              // issued first: everything that depends on it is far downstream
    auto rows_of = [&](int grp, long long (&rows)[R], int& head, int& i0) -> bool {
        const bool qk = grp < qk_groups;
        if (qk) {
            head = grp / half;
            i0 = grp % half;
            rows[0] = (long long)head * hd + i0;
            rows[1] = rows[0] + half;
        } else {
            rows[0] = (long long)(H + Hk) * hd + (grp - qk_groups) * 2;
            rows[1] = rows[0] + 1;
        }
        return qk;
    };
    // x and norm_w in one round trip, the wave's first weight block right behind them (it hides under the prologue),
    // then the normalised row (rounded to T) in LDS as fp32
    constexpr int XPT = 6;
    float ss = 0.f;
    uint4 wpre[U][R];
    const int grp0 = blockIdx.x * GV_WAVES + wid;
    bool have = false;
    if (nchunk <= XPT * GV_THREADS) {
        uint4 xr[XPT], nr[XPT];
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
            const int c = tid + i * GV_THREADS;
            xr[i] = (c < nchunk) ? *reinterpret_cast<const uint4*>(x + (long long)c * VE) : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
            const int c = tid + i * GV_THREADS;
            nr[i] = (c < nchunk) ? *reinterpret_cast<const uint4*>(norm_w + (long long)c * VE) : make_uint4(0, 0, 0, 0);
        }
        if (grp0 < ngroups && nchunk >= 64 * U) {
            long long rows[R];
            int head, i0;
            rows_of(grp0, rows, head, i0);
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const T* p = W + rows[r] * K + (long long)(u * 64 + lane) * VE;
                    wpre[u][r] = NT ? ld_nt16(p) : *reinterpret_cast<const uint4*>(p);
                }
            have = true;
        }
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
            float f[VE];
            Vec16<T>::cvt(xr[i], f);
#pragma unroll
            for (int e = 0; e < VE; ++e) ss = fmaf(f[e], f[e], ss);
        }
        const float rr = rsqrtf(block_sum<GV_THREADS>(ss, red) / K + eps);
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
            const int c = tid + i * GV_THREADS;
            if (c < nchunk) {
                float f[VE], g[VE];
                Vec16<T>::cvt(xr[i], f);
                Vec16<T>::cvt(nr[i], g);
#pragma unroll
                for (int e = 0; e < VE; ++e) xs[c * VE + e] = Elem<T>::round(f[e] * rr * g[e]);
            }
        }
    } else {
        for (int c = tid; c < nchunk; c += GV_THREADS) {
            const uint4 raw = *reinterpret_cast<const uint4*>(x + (long long)c * VE);
            float f[VE];
            Vec16<T>::cvt(raw, f);
#pragma unroll
            for (int e = 0; e < VE; ++e) { xs[c * VE + e] = f[e]; ss = fmaf(f[e], f[e], ss); }
        }
        const float rr = rsqrtf(block_sum<GV_THREADS>(ss, red) / K + eps);
        for (int c = tid; c < nchunk; c += GV_THREADS) {
            const uint4 raw = *reinterpret_cast<const uint4*>(norm_w + (long long)c * VE);
            float f[VE];
            Vec16<T>::cvt(raw, f);
#pragma unroll
            for (int e = 0; e < VE; ++e) xs[c * VE + e] = Elem<T>::round(xs[c * VE + e] * rr * f[e]);
        }
    }
    __syncthreads();

    for (int grp = grp0; grp < ngroups; grp += nwaves) {
        long long rows[R];
        int head = 0, i0 = 0;
        const bool is_qk = rows_of(grp, rows, head, i0);
        // rotation coefficients of this pair: loaded before the weight stream so the epilogue never waits on memory
        float rc = 1.f, rs = 0.f;
        if (is_qk) { rc = cs[(long long)pos * half + i0]; rs = sn[(long long)pos * half + i0]; }
        float acc[R];
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = 0.f;
        const int cfull = (nchunk / (64 * U)) * (64 * U);
        int c0 = 0;
        if (have) {                                           // block 0 of the first group was prefetched
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int c = u * 64 + lane;
                float xv[VE];
#pragma unroll
                for (int e = 0; e < VE; e += 4) {
                    const float4 t = *reinterpret_cast<const float4*>(xs + c * VE + e);
                    xv[e] = t.x; xv[e + 1] = t.y; xv[e + 2] = t.z; xv[e + 3] = t.w;
                }
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    float f[VE];
                    Vec16<T>::cvt(wpre[u][r], f);
#pragma unroll
                    for (int e = 0; e < VE; ++e) acc[r] = fmaf(f[e], xv[e], acc[r]);
                }
            }
            c0 = 64 * U;
            have = false;
        }
        for (; c0 < cfull; c0 += 64 * U) {                   // steady state: no bounds checks
            uint4 w[U][R];
#pragma unroll
            for (int u = 0; u < U; ++u) {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const T* p = W + rows[r] * K + (long long)(c0 + u * 64 + lane) * VE;
                    w[u][r] = NT ? ld_nt16(p) : *reinterpret_cast<const uint4*>(p);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int c = c0 + u * 64 + lane;
                float xv[VE];
#pragma unroll
                for (int e = 0; e < VE; e += 4) {
                    const float4 t = *reinterpret_cast<const float4*>(xs + c * VE + e);
                    xv[e] = t.x; xv[e + 1] = t.y; xv[e + 2] = t.z; xv[e + 3] = t.w;
                }
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    float f[VE];
                    Vec16<T>::cvt(w[u][r], f);
#pragma unroll
                    for (int e = 0; e < VE; ++e) acc[r] = fmaf(f[e], xv[e], acc[r]);
                }
            }
        }
        for (; c0 < nchunk; c0 += 64 * U) {
            uint4 w[U][R];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int c = c0 + u * 64 + lane;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const T* p = W + rows[r] * K + (long long)c * VE;
                    w[u][r] = (c < nchunk) ? (NT ? ld_nt16(p) : *reinterpret_cast<const uint4*>(p)) : make_uint4(0, 0, 0, 0);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int c = c0 + u * 64 + lane;
                float xv[VE];
#pragma unroll
                for (int e = 0; e < VE; ++e) xv[e] = 0.f;
                if (c < nchunk) {
#pragma unroll
                    for (int e = 0; e < VE; e += 4) {
                        const float4 t = *reinterpret_cast<const float4*>(xs + c * VE + e);
                        xv[e] = t.x; xv[e + 1] = t.y; xv[e + 2] = t.z; xv[e + 3] = t.w;
                    }
                }
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    float f[VE];
                    Vec16<T>::cvt(w[u][r], f);
#pragma unroll
                    for (int e = 0; e < VE; ++e) acc[r] = fmaf(f[e], xv[e], acc[r]);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = wave_sum(acc[r]);
        if (lane == 0) {
            if (is_qk) {
                const float x1 = Elem<T>::round(acc[0]), x2 = Elem<T>::round(acc[1]);
                const float y1 = x1 * rc - x2 * rs, y2 = x2 * rc + x1 * rs;
                if (head < H) {
                    Elem<T>::st(qout + head * hd + i0, y1);
                    Elem<T>::st(qout + head * hd + i0 + half, y2);
                } else {
                    T* dst = kc + ((long long)(head - H) * S_max + pos) * hd;
                    Elem<T>::st(dst + i0, y1);
                    Elem<T>::st(dst + i0 + half, y2);
                }
            } else {
                const int v0 = (grp - qk_groups) * 2;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int hk = (v0 + r) / hd, d = (v0 + r) % hd;
                    T val;
                    Elem<T>::st(&val, acc[r]);
                    vc[((long long)hk * S_max + pos) * hd + d] = val;
                    if (vtc) vtc[((long long)hk * hd + d) * S_max + pos] = val;
                }
            }
        }
    }
}

int gemv_qkv_rope(const void* x, const void* W, const void* norm_w, void* qout, const float* cs, const float* sn,
                  const int* d_pos, void* kc, void* vc, void* vtc, int S_max, int H, int Hk, int hd, int K, float eps,
                  int dtype, hipStream_t st) {
    const int ve = dtype == TEO_F32 ? 4 : 8;
    TEO_CHECK_ARG(K % ve == 0 && hd % 2 == 0, "gemv_qkv_rope: K=%d hd=%d", K, hd);
    TEO_CHECK_ARG((size_t)(K + 16) * 4 <= 64 * 1024, "gemv_qkv_rope: K=%d too large for LDS staging", K);
    const int ngroups = (H + Hk) * (hd / 2) + Hk * hd / 2;
    int blocks = cdiv(ngroups, GV_WAVES);
    if (blocks > g_tune.max_blocks) blocks = g_tune.max_blocks;
    const size_t lds = (size_t)(((K + 3) & ~3) + 8) * sizeof(float);
#define TEO_QR(TT, NTV)                                                                                              \
    gemv_qkv_rope_kernel<TT, NTV><<<blocks, GV_THREADS, lds, st>>>((const TT*)x, (const TT*)W, (const TT*)norm_w,     \
                                                                   (TT*)qout, cs, sn, d_pos, (TT*)kc, (TT*)vc, (TT*)vtc, \
                                                                   S_max, H, Hk, hd, K, eps)
    if (dtype == TEO_F32) { if (g_tune.nt) TEO_QR(float, true); else TEO_QR(float, false); }
    else                  { if (g_tune.nt) TEO_QR(bf16_t, true); else TEO_QR(bf16_t, false); }
#undef TEO_QR
    TEO_LAUNCH_CHECK("gemv_qkv_rope");
    return TEO_OK;
}

int gemv(const void* x, const void* W, const void* norm_w, const void* res, void* y, int N, int K, float eps,
         unsigned flags, int dtype, int out_dtype, hipStream_t st) {
    if (N == 0) return TEO_OK;
    const bool swiglu = flags & TEO_GEMM_SWIGLU16;
    const int ve = dtype == TEO_F32 ? 4 : 8;
    TEO_CHECK_ARG(K % ve == 0, "teo_gemv: K=%d must be a multiple of %d", K, ve);
    TEO_CHECK_ARG((reinterpret_cast<uintptr_t>(W) & 15) == 0, "teo_gemv: W must be 16-byte aligned");
    TEO_CHECK_ARG((size_t)(K + 16) * 4 <= 64 * 1024, "teo_gemv: K=%d too large for LDS staging", K);
    TEO_CHECK_ARG((reinterpret_cast<uintptr_t>(x) & 15) == 0 && (norm_w == nullptr || (reinterpret_cast<uintptr_t>(norm_w) & 15) == 0), "teo_gemv: x / norm_w must be 16-byte aligned");
    if (swiglu) TEO_CHECK_ARG(N % 32 == 0 && !res, "teo_gemv: SWIGLU16 needs N %% 32 == 0 and no residual");
    if (dtype == TEO_F32) {
        TEO_CHECK_ARG(out_dtype == TEO_F32, "teo_gemv: f32 inputs need f32 output");
        return gemv_launch<float, float>(x, W, norm_w, res, y, N, K, eps, swiglu, st);
    }
    if (dtype == TEO_BF16) {
        if (out_dtype == TEO_F32) return gemv_launch<bf16_t, float>(x, W, norm_w, res, y, N, K, eps, swiglu, st);
        if (g_tune.variant >= 0) return gemv_launch_variant(g_tune.variant, x, W, norm_w, res, y, N, K, eps, swiglu, st);
        return gemv_launch<bf16_t, bf16_t>(x, W, norm_w, res, y, N, K, eps, swiglu, st);
    }
    set_error("teo_gemv: unknown dtype %d", dtype);
    return TEO_ERR_UNSUPPORTED;
}

}  // namespace teo
