// Decode GEMV: y[N] = W[N,K] . f(x) for ONE activation row (q_len == 1 decode step).
//
// HBM-bound: every weight byte is read exactly once per token, so the kernels are pure weight streams:
//   - no MFMA, no LDS staging of W (each byte is used once; an LDS round trip would be pure overhead, and LDS-DMA
//     streaming measured no faster than non-temporal loads to VGPRs: tools/stream_probe.py);
//   - 16-byte non-temporal loads, 2 rows x 4 KiB contiguous per wave per step (8 loads in flight per lane);
//   - x (and the RMSNorm weight) are loaded in ONE round trip, the wave's first weight block is issued right behind
//     them UNCONDITIONALLY (exec-masked loads make hipcc wait vmcnt(0) immediately) so it hides under the prologue;
//     f(x) is staged once per workgroup in LDS as fp32 and re-read with conflict-free ds_read_b128;
//   - few-long-rows layers (o / down, N = 4096) use split-K workgroups: 4 waves share 2 rows;
//   - epilogues fused: residual add, SwiGLU on the interleaved-16 gate/up layout, fp32 logits, RoPE + KV append;
//   - weights may be bf16/f32 (same type as x) or fp8 e4m3 (OCP) with one fp32 scale per output row.
// Roofline: HBM (8 TB/s spec); algorithmic bytes per launch = N*K*sizeof(WT) (+ K + N elements, negligible).
#include <type_traits>

#include "ops.h"

namespace teo {

constexpr int GV_WAVES = 4;           // waves per workgroup
constexpr int GV_THREADS = GV_WAVES * 64;
typedef unsigned char fp8_t;          // OCP e4m3fn bits
typedef float f32x2 __attribute__((ext_vector_type(2)));

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
template <> struct Vec16<f16_t> {
    static constexpr int N = 8;
    __device__ static __forceinline__ void cvt(const uint4& r, float* f) {
        f[0] = h_lo<true>(r.x); f[1] = h_hi<true>(r.x); f[2] = h_lo<true>(r.y); f[3] = h_hi<true>(r.y);
        f[4] = h_lo<true>(r.z); f[5] = h_hi<true>(r.z); f[6] = h_lo<true>(r.w); f[7] = h_hi<true>(r.w);
    }
};
template <> struct Vec16<float> {
    static constexpr int N = 4;
    __device__ static __forceinline__ void cvt(const uint4& r, float* f) {
        f[0] = __uint_as_float(r.x); f[1] = __uint_as_float(r.y); f[2] = __uint_as_float(r.z); f[3] = __uint_as_float(r.w);
    }
};
template <> struct Vec16<fp8_t> {
    static constexpr int N = 16;
    __device__ static __forceinline__ void cvt(const uint4& r, float* f) {
        const unsigned w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x2 lo = __builtin_amdgcn_cvt_pk_f32_fp8(w[i], false);   // bytes 0,1
            const f32x2 hi = __builtin_amdgcn_cvt_pk_f32_fp8(w[i], true);    // bytes 2,3
            f[4 * i] = lo.x; f[4 * i + 1] = lo.y; f[4 * i + 2] = hi.x; f[4 * i + 3] = hi.y;
        }
    }
};

// acc += <16 bytes of weights> . xv  -- fp8 is consumed word by word so only 4 converted values are live at a time
template <typename WT>
__device__ __forceinline__ float dot16(const uint4& w, const float* xv, float acc) {
    constexpr int VE = Vec16<WT>::N;
    float f[VE];
    Vec16<WT>::cvt(w, f);
#pragma unroll
    for (int e = 0; e < VE; ++e) acc = fmaf(f[e], xv[e], acc);
    return acc;
}
template <>
__device__ __forceinline__ float dot16<fp8_t>(const uint4& w, const float* xv, float acc) {
    const unsigned ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const f32x2 lo = __builtin_amdgcn_cvt_pk_f32_fp8(ww[i], false);
        const f32x2 hi = __builtin_amdgcn_cvt_pk_f32_fp8(ww[i], true);
        acc = fmaf(lo.x, xv[4 * i], acc);
        acc = fmaf(lo.y, xv[4 * i + 1], acc);
        acc = fmaf(hi.x, xv[4 * i + 2], acc);
        acc = fmaf(hi.y, xv[4 * i + 3], acc);
    }
    return acc;
}

// ---- bf16 activations: x lives in LDS as bf16 (it IS bf16: the input row, or the normalised row after its one rounding)
// and meets a 16-byte weight chunk through v_dot2c_f32_bf16 (two exact products + fp32 accumulate per instruction): 4
// VALU instructions and one ds_read_b128 per 8 weights instead of 16 unpack + 8 FMA and two reads of an fp32 image.
typedef __bf16 gv_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float dot2bf(unsigned a, unsigned b, float acc) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(gv_bf16x2, a), __builtin_bit_cast(gv_bf16x2, b), acc, false);
}
// xv: VE/8 chunks of 8 bf16 activations matching the weight chunk
template <typename WT>
__device__ __forceinline__ float dotb(const uint4& w, const uint4* xv, float acc) {      // 16-bit weights of the activations' format
    constexpr bool F16 = IsF16<WT>::v;
    acc = dot2h<F16>(w.x, xv[0].x, acc); acc = dot2h<F16>(w.y, xv[0].y, acc);
    acc = dot2h<F16>(w.z, xv[0].z, acc); acc = dot2h<F16>(w.w, xv[0].w, acc);
    return acc;
}
template <>
__device__ __forceinline__ float dotb<fp8_t>(const uint4& w, const uint4* xv, float acc) {   // 16 fp8 weights, exact -> bf16
    const unsigned ww[4] = {w.x, w.y, w.z, w.w};
    const unsigned xx[8] = {xv[0].x, xv[0].y, xv[0].z, xv[0].w, xv[1].x, xv[1].y, xv[1].z, xv[1].w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned lo = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(ww[i], 1.0f, false));
        const unsigned hi = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(ww[i], 1.0f, true));
        acc = dot2bf(lo, xx[2 * i], acc);
        acc = dot2bf(hi, xx[2 * i + 1], acc);
    }
    return acc;
}
template <typename T> struct IsBf { static constexpr bool v = false; };
template <> struct IsBf<bf16_t> { static constexpr bool v = true; };
template <typename T> struct Is16 { static constexpr bool v = IsBf<T>::v || IsF16<T>::v; };      // a 16-bit activation format
// Row-group kernels keep x in LDS.  Measured on one box: the bf16 image + v_dot2c is a win for fp8 weights (VALU-bound:
// qkv 12.2 -> 11.7 us, gate/up 17.0 -> 16.5) but a loss for bf16 weights (gate/up 29.3 -> 30.8 us: v_dot2c_f32_bf16 issues
// slower than the unpack + FMA pairs it replaces), so bf16 weights keep the fp32 image.  The split-K kernel reads x from
// global memory as bf16 and uses v_dot2c for both (down projection 16.6 -> 15.6 us).
template <typename T, typename WT> struct BfImage { static constexpr bool v = IsBf<T>::v && sizeof(WT) == 1; };
// Round 6, IEEE half weights in the row-group kernel (gate/up, lm_head): the fp32 image costs 8 v_cvt_f32_f16 (four of them SDWA forms,
// with their s_nop hazards) + 4 v_pk_fma_f32 + the register moves that pair them per 8 weights -- 20 VALU instructions where bfloat16
// needs 16 (shift / mask unpack) -- and measured 1.2-2.6 % behind bf16 whatever the data (tools/fp16_probe.py).  The 16-bit image +
// v_dot2_f32_f16 is 4 instructions and one ds_read_b128 per 8 weights.  (The QKV + RoPE kernel keeps the fp32 image: it ties there.)
template <typename T, typename WT> struct RowsImage16 { static constexpr bool v = BfImage<T, WT>::v || (IsF16<T>::v && IsF16<WT>::v); };

template <bool NT>
__device__ __forceinline__ uint4 ld16(const void* p) {
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    const v4u v = NT ? __builtin_nontemporal_load(reinterpret_cast<const v4u*>(p)) : *reinterpret_cast<const v4u*>(p);
    return make_uint4(v.x, v.y, v.z, v.w);
}

// Tunables (teo_tune, tune.h): gemv_nt non-temporal loads on/off, gemv_max_blocks workgroup cap, gemv_variant of the row-group kernel, ...

// LDS image of f(x) for weight chunks of VE elements: the lanes of a wave read chunk (cb*64 + lane), so the image is
// lane-linear per 4-float group -- [cb][g = e/4][lane][4 floats] -- and every ds_read_b128 of a wave is one contiguous
// KiB (conflict-free).  (A plain xs[k] image makes lanes stride by VE*4 bytes: 2-way conflicts for bf16 chunks, 4-way
// for fp8 chunks.)  k must be a multiple of 4.
template <int VE>
__device__ __forceinline__ int xs_off(int k) {
    const int c = k / VE, g = (k % VE) >> 2;
    return (((c >> 6) * (VE / 4) + g) << 8) + ((c & 63) << 2);
}
template <int VE>
static size_t xs_lds_bytes(int K) { return ((size_t)((K / VE + 63) / 64) * 64 * VE + 8) * sizeof(float); }
// bf16 image: 16-byte pieces, lane-linear per piece index g = (k % VE) / 8 (VE = 8: the plain row)
template <int VE>
__device__ __forceinline__ int xb_off(int k) {
    const int c = k / VE, g = (k % VE) >> 3;
    return ((((c >> 6) * (VE / 8) + g) << 6) + (c & 63)) * 8 + (k & 7);
}
template <int VE>
static size_t xb_lds_bytes(int K) { return (size_t)((K / VE + 63) / 64) * 64 * VE * 2 + 8 * sizeof(float); }
template <bool XB, int VE>
static size_t x_image_bytes(int K) { return XB ? xb_lds_bytes<VE>(K) : xs_lds_bytes<VE>(K); }
// scratch floats behind the image
template <bool XB, int VE>
__device__ __forceinline__ float* x_image_end(float* xs, int nchunk) {
    const int padded = ((nchunk + 63) / 64) * 64 * VE;
    return XB ? reinterpret_cast<float*>(reinterpret_cast<bf16_t*>(xs) + padded) : xs + padded;
}

// ------------------------------------------------------------------------------------------------
// prologue shared by the row-group kernels: xs (fp32, LDS, xs_off layout) = f(x), f = identity or rmsnorm(x)*norm_w rounded to T.
// `after_loads()` runs once the x / norm_w loads are in flight (the caller issues its weight prefetch there).
// ------------------------------------------------------------------------------------------------
template <typename T, int VE, bool XB, int XPT, typename F>
__device__ __forceinline__ void stage_x(const T* __restrict__ x, const T* __restrict__ norm_w, float* xs, float* red, int K,
                                        float eps, F after_loads) {
    constexpr int VX = Vec16<T>::N;
    // XPT = 16-byte x chunks a thread holds in registers: K <= XPT*256*VX (6: 12288 for bf16; 2: 4096).  The kernel's VGPR
    // allocation is the maximum over the whole kernel, and with XPT = 6 this prologue IS the maximum (fp8 row-group kernel:
    // 128 VGPRs = 4 waves per SIMD against 61 = 8 with XPT = 2): the host picks the small form whenever K allows.
    const int tid = threadIdx.x;
    const int nx = K / VX;
    float ss = 0.f;
    if (nx <= XPT * GV_THREADS) {
        uint4 xr[XPT], nr[XPT];
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
            const int c = tid + i * GV_THREADS;
            xr[i] = (c < nx) ? *reinterpret_cast<const uint4*>(x + (long long)c * VX) : make_uint4(0, 0, 0, 0);
        }
        if (norm_w) {                             // same round trip as x: no second dependent global-load phase
#pragma unroll
            for (int i = 0; i < XPT; ++i) {
                const int c = tid + i * GV_THREADS;
                nr[i] = (c < nx) ? *reinterpret_cast<const uint4*>(norm_w + (long long)c * VX) : make_uint4(0, 0, 0, 0);
            }
        }
        after_loads();
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
            float f[VX];
            Vec16<T>::cvt(xr[i], f);
#pragma unroll
            for (int e = 0; e < VX; ++e) ss = fmaf(f[e], f[e], ss);
        }
        float rr = 1.f;
        if (norm_w) rr = rsqrtf(block_sum<GV_THREADS>(ss, red) / K + eps);
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
            const int c = tid + i * GV_THREADS;
            if (c < nx) {
                float f[VX], g[VX];
                Vec16<T>::cvt(xr[i], f);
                if (norm_w) {
                    Vec16<T>::cvt(nr[i], g);
#pragma unroll
                    for (int e = 0; e < VX; ++e) f[e] = Elem<T>::round(f[e] * rr * g[e]);
                }
                if constexpr (XB) {          // bf16 image: the raw row, or the rounded normalised values repacked
                    uint4 o = xr[i];
                    if (norm_w) o = make_uint4(pack_h2<IsF16<T>::v>(f[0], f[1]), pack_h2<IsF16<T>::v>(f[2], f[3]), pack_h2<IsF16<T>::v>(f[4], f[5]),
                                               pack_h2<IsF16<T>::v>(f[6], f[7]));
                    *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(xs) + xb_off<VE>(c * VX)) = o;
                } else {
#pragma unroll
                    for (int e = 0; e < VX; e += 4)
                        *reinterpret_cast<float4*>(xs + xs_off<VE>(c * VX + e)) = make_float4(f[e], f[e + 1], f[e + 2], f[e + 3]);
                }
            }
        }
    } else {
        after_loads();
        for (int c = tid; c < nx; c += GV_THREADS) {
            const uint4 raw = *reinterpret_cast<const uint4*>(x + (long long)c * VX);
            float f[VX];
            Vec16<T>::cvt(raw, f);
#pragma unroll
            for (int e = 0; e < VX; ++e) ss = fmaf(f[e], f[e], ss);
            if constexpr (XB) {
                *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(xs) + xb_off<VE>(c * VX)) = raw;
            } else {
#pragma unroll
                for (int e = 0; e < VX; e += 4)
                    *reinterpret_cast<float4*>(xs + xs_off<VE>(c * VX + e)) = make_float4(f[e], f[e + 1], f[e + 2], f[e + 3]);
            }
        }
        if (norm_w) {
            const float rr = rsqrtf(block_sum<GV_THREADS>(ss, red) / K + eps);
            for (int c = tid; c < nx; c += GV_THREADS) {
                float f[VX];
                Vec16<T>::cvt(*reinterpret_cast<const uint4*>(norm_w + (long long)c * VX), f);
                if constexpr (XB) {
                    uint4* p4 = reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(xs) + xb_off<VE>(c * VX));
                    float v[VX];
                    Vec16<T>::cvt(*p4, v);
#pragma unroll
                    for (int e = 0; e < VX; ++e) v[e] = v[e] * rr * f[e];
                    *p4 = make_uint4(pack_h2<IsF16<T>::v>(v[0], v[1]), pack_h2<IsF16<T>::v>(v[2], v[3]), pack_h2<IsF16<T>::v>(v[4], v[5]),
                                     pack_h2<IsF16<T>::v>(v[6], v[7]));
                } else {
#pragma unroll
                    for (int e = 0; e < VX; e += 4) {
                        float4* p4 = reinterpret_cast<float4*>(xs + xs_off<VE>(c * VX + e));
                        float4 v = *p4;
                        v.x = Elem<T>::round(v.x * rr * f[e]); v.y = Elem<T>::round(v.y * rr * f[e + 1]);
                        v.z = Elem<T>::round(v.z * rr * f[e + 2]); v.w = Elem<T>::round(v.w * rr * f[e + 3]);
                        *p4 = v;
                    }
                }
            }
        }
    }
    __syncthreads();
}

// one block of R rows x U chunks: loads and the dot-product update
template <typename WT, int R, int U, bool NT, bool FULL>
__device__ __forceinline__ void issue_block(uint4 (&w)[U][R], const WT* const (&rowp)[R], int c0, int lane, int nchunk) {
    constexpr int VE = Vec16<WT>::N;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int c = c0 + u * 64 + lane;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (FULL) w[u][r] = ld16<NT>(rowp[r] + (long long)c * VE);
            else w[u][r] = (c < nchunk) ? ld16<NT>(rowp[r] + (long long)c * VE) : make_uint4(0, 0, 0, 0);
        }
    }
}
template <typename T, typename WT, int R, int U, bool FULL, bool XB16 = BfImage<T, WT>::v>
__device__ __forceinline__ void consume_block(const uint4 (&w)[U][R], const float* xs, int c0, int lane, int nchunk,
                                              float (&acc)[R]) {
    constexpr int VE = Vec16<WT>::N;
    if constexpr (XB16) {
        // one accumulator per (row, chunk): v_dot2c chains of 4 instead of 4 U (the instruction accumulates in place, so
        // a single accumulator per row serialises the whole step)
        float part[U][R];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int c = c0 + u * 64 + lane;
            uint4 xb[VE / 8];
#pragma unroll
            for (int j = 0; j < VE / 8; ++j)
                xb[j] = (FULL || c < nchunk) ? *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(xs) + xb_off<VE>(c * VE + j * 8))
                                             : make_uint4(0, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < R; ++r) part[u][r] = dotb<WT>(w[u][r], xb, 0.f);
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float t = part[0][r];
#pragma unroll
            for (int u = 1; u < U; ++u) t += part[u][r];
            acc[r] += t;
        }
        return;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int c = c0 + u * 64 + lane;
        float xv[VE];
        if (FULL || c < nchunk) {
#pragma unroll
            for (int e = 0; e < VE; e += 4) {
                const float4 t = *reinterpret_cast<const float4*>(xs + xs_off<VE>(c * VE + e));
                xv[e] = t.x; xv[e + 1] = t.y; xv[e + 2] = t.z; xv[e + 3] = t.w;
            }
        } else {
#pragma unroll
            for (int e = 0; e < VE; ++e) xv[e] = 0.f;
        }
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = dot16<WT>(w[u][r], xv, acc[r]);
        // 16-element chunks (fp8): keep the scheduler from hoisting every chunk's LDS reads and conversions to the top
        // of the step (that costs 170-250 VGPRs and the occupancy with it)
        if (VE > 8) __builtin_amdgcn_sched_barrier(0);
    }
}

// ------------------------------------------------------------------------------------------------
// Row-group kernel: a wave owns R rows (SWIGLU: R/2 (gate, up) pairs 16 apart inside 32-row blocks).
// ------------------------------------------------------------------------------------------------
template <typename T, typename TO, typename WT, int R, int U, bool PF, bool NT, bool SWIGLU, int XPT>
__global__ __launch_bounds__(GV_THREADS) void gemv_kernel(const T* __restrict__ x, const WT* __restrict__ W,
                                                          const float* __restrict__ wscale, const T* __restrict__ norm_w,
                                                          const T* res, TO* y, int N, int K,
                                                          float eps) {
    extern __shared__ __attribute__((aligned(16))) float xs[];     // [K] fp32 (+8 floats of reduction scratch)
    constexpr int VE = Vec16<WT>::N;
    constexpr int STEP = 64 * U;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int nchunk = K / VE;                    // 16-byte weight chunks per row
    float* red = x_image_end<RowsImage16<T, WT>::v, VE>(xs, nchunk);
    const int nwaves = gridDim.x * GV_WAVES;
    const int ngroups = SWIGLU ? (N / 2 + (R / 2) - 1) / (R / 2) : (N + R - 1) / R;

    auto row_of = [&](int grp, int r) -> long long {
        if (SWIGLU) {
            const int j = min(grp * (R / 2) + (r >> 1), N / 2 - 1);     // output column
            return (long long)((j >> 4) * 32 + (j & 15) + ((r & 1) ? 16 : 0));
        }
        return (long long)min(grp * R + r, N - 1);
    };

    uint4 wa[U][R];
    int grp = blockIdx.x * GV_WAVES + wid;
    // PF (host guarantees nchunk >= STEP): NO branch around the prefetch -- at a control-flow merge hipcc waits vmcnt(0),
    // which would drain the weights before the prologue.  Waves past the last group prefetch a clamped (valid) row.
    stage_x<T, VE, RowsImage16<T, WT>::v, XPT>(x, norm_w, xs, red, K, eps, [&]() {
        if (PF) {
            const int gp = min(grp, ngroups - 1);
            const WT* rowp[R];
#pragma unroll
            for (int r = 0; r < R; ++r) rowp[r] = W + row_of(gp, r) * K;
            issue_block<WT, R, U, NT, true>(wa, rowp, 0, lane, nchunk);
        }
    });

    bool have = PF;
    for (; grp < ngroups; grp += nwaves) {
        const WT* rowp[R];
        long long rows[R];
#pragma unroll
        for (int r = 0; r < R; ++r) { rows[r] = row_of(grp, r); rowp[r] = W + rows[r] * K; }
        float sc[R];                                 // row scales: loaded ahead of the weight stream, never waited on later
#pragma unroll
        for (int r = 0; r < R; ++r) sc[r] = wscale ? wscale[rows[r]] : 1.f;
        float acc[R];
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = 0.f;
        int c0 = 0;
        if (have) { consume_block<T, WT, R, U, true, RowsImage16<T, WT>::v>(wa, xs, 0, lane, nchunk, acc); c0 = STEP; have = false; }
        const int cfull = (nchunk / STEP) * STEP;          // steady state: no bounds checks, no exec masking
        for (; c0 < cfull; c0 += STEP) {
            issue_block<WT, R, U, NT, true>(wa, rowp, c0, lane, nchunk);
            consume_block<T, WT, R, U, true, RowsImage16<T, WT>::v>(wa, xs, c0, lane, nchunk, acc);
        }
        if (c0 < nchunk) {
            issue_block<WT, R, U, NT, false>(wa, rowp, c0, lane, nchunk);
            consume_block<T, WT, R, U, false, RowsImage16<T, WT>::v>(wa, xs, c0, lane, nchunk, acc);
        }
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = wave_sum(acc[r]);
        if (lane == 0) {
#pragma unroll
            for (int r = 0; r < R; ++r) acc[r] *= sc[r];
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
// interleaved chunks of every row, so 4x more waves stream than with one wave per row group; partial sums meet in LDS.
// x is read by each wave for its own chunks only (no norm on these layers -> no full-vector prologue).
// ------------------------------------------------------------------------------------------------
template <typename T, typename TO, typename WT, int R, int U, bool NT>
__global__ __launch_bounds__(GV_THREADS) void gemv_splitk_kernel(const T* __restrict__ x, const WT* __restrict__ W,
                                                                 const float* __restrict__ wscale, const T* res,
                                                                 TO* y, int N, int K) {
    constexpr int VE = Vec16<WT>::N, VX = Vec16<T>::N;
    constexpr int XL = VE / VX;                           // 16-byte x loads per weight chunk (1, or 2 for fp8 weights)
    __shared__ float part[GV_WAVES][R];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int nchunk = K / VE;
    const int row0 = blockIdx.x * R;
    const float my_scale = (wscale && tid < R && row0 + tid < N) ? wscale[row0 + tid] : 1.f;   // ahead of the stream
    // the residual too: read after the reduction barrier it is a dependent global round trip at the tail of EVERY workgroup
    // (all of them are resident at once, so the whole launch ends one memory latency later); only this thread writes y[n]
    const float my_res = (res && tid < R && row0 + tid < N) ? Elem<T>::ld(res + row0 + tid) : 0.f;
    float acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = 0.f;
    const WT* wrow[R];
#pragma unroll
    for (int r = 0; r < R; ++r) wrow[r] = W + (long long)min(row0 + r, N - 1) * K;
    // chunk index for (iteration, unroll u): c = cb + u*256 + wid*64 + lane
    auto step = [&](int cb, auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
        uint4 xr[U][XL];
        uint4 w[U][R];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int c = cb + u * 256 + wid * 64 + lane;
#pragma unroll
            for (int j = 0; j < XL; ++j)
                xr[u][j] = (FULL || c < nchunk) ? *reinterpret_cast<const uint4*>(x + (long long)c * VE + j * VX) : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int c = cb + u * 256 + wid * 64 + lane;
#pragma unroll
            for (int r = 0; r < R; ++r)
                w[u][r] = (FULL || c < nchunk) ? ld16<NT>(wrow[r] + (long long)c * VE) : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if constexpr (Is16<T>::v) {                  // raw 16-bit activations straight into v_dot2_f32_{bf16,f16}
#pragma unroll
                for (int r = 0; r < R; ++r) acc[r] = dotb<WT>(w[u][r], xr[u], acc[r]);
            } else {
                float xv[VE];
#pragma unroll
                for (int j = 0; j < XL; ++j) Vec16<T>::cvt(xr[u][j], xv + j * VX);
#pragma unroll
                for (int r = 0; r < R; ++r) acc[r] = dot16<WT>(w[u][r], xv, acc[r]);
            }
        }
    };
    const int cfull = (nchunk / (256 * U)) * (256 * U);       // steady state: every lane in range, no exec masking
    int cb = 0;
    for (; cb < cfull; cb += 256 * U) step(cb, std::true_type{});
    for (; cb < nchunk; cb += 256 * U) step(cb, std::false_type{});
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
            v *= my_scale;
            v += my_res;
            Elem<TO>::st(y + n, v);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Decode QKV projection with the RoPE rotation and the KV-cache append fused into the epilogue.
//   rows of Wqkv: [q: H*hd | k: Hk*hd | v: Hk*hd].  A wave owns one rotation pair (i, i + hd/2) of a q/k head, or 2
//   consecutive v rows.  q is written rotated to qout[H*hd]; k rotated to K cache[hk][pos][:]; v to V cache[hk][pos][:]
//   and V^T cache[hk][:][pos].  pos (= cache slot = rotary position) is read from device memory so the launch can be
//   replayed from a hipGraph.  Rounding points are those of the unfused path: round(linear) -> rotate in fp32 -> round.
// ------------------------------------------------------------------------------------------------
template <typename T, typename WT, bool NT, bool PF, int U, int XPT>
__global__ __launch_bounds__(GV_THREADS) void gemv_qkv_rope_kernel(const WT* __restrict__ W, const T* __restrict__ x,
                                                                   const T* __restrict__ norm_w, const float* __restrict__ wscale,
                                                                   const int* __restrict__ d_pos, int K, int H, int Hk, int hd,
                                                                   T* __restrict__ qout, const float* __restrict__ cs,
                                                                   const float* __restrict__ sn, T* __restrict__ kc,
                                                                   T* __restrict__ vc, T* __restrict__ vtc, int S_max, float eps) {
    // (argument order: the weight stream's operands and the position pointer first -- those 14 dwords arrive preloaded in SGPRs)
    extern __shared__ __attribute__((aligned(16))) float xs[];
    constexpr int VE = Vec16<WT>::N;
    constexpr int R = 2, STEP = 64 * U;                  // U chunks per row per step (4 for 2-byte weights, 2 for fp8)
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int nchunk = K / VE;
    float* red = x_image_end<BfImage<T, WT>::v, VE>(xs, nchunk);
    const int half = hd >> 1, half_log2 = __builtin_ctz((unsigned)hd) - 1;
    const int qk_groups = (H + Hk) * half;              // rotation pairs
    const int ngroups = qk_groups + (Hk * hd) / 2;
    const int nwaves = gridDim.x * GV_WAVES;
    const int pos = *d_pos;                              // issued first: everything that depends on it is far downstream

    auto rows_of = [&](int grp, long long (&rows)[R], int& head, int& i0) -> bool {
        const bool qk = grp < qk_groups;
        if (qk) {
            head = grp >> half_log2;                     // hd is a power of two (host-checked): no integer division per group
            i0 = grp & (half - 1);
            rows[0] = (long long)head * hd + i0;
            rows[1] = rows[0] + half;
        } else {
            rows[0] = (long long)(H + Hk) * hd + (grp - qk_groups) * 2;
            rows[1] = rows[0] + 1;
        }
        return qk;
    };

    uint4 wa[U][R];
    const int grp0 = blockIdx.x * GV_WAVES + wid;
    stage_x<T, VE, BfImage<T, WT>::v, XPT>(x, norm_w, xs, red, K, eps, [&]() {
        if (PF) {                                         // branch-free (see gemv_kernel)
            long long rows[R];
            int head, i0;
            rows_of(min(grp0, ngroups - 1), rows, head, i0);
            const WT* rowp[R] = {W + rows[0] * K, W + rows[1] * K};
            issue_block<WT, R, U, NT, true>(wa, rowp, 0, lane, nchunk);
        }
    });

    bool have = PF;
    for (int grp = grp0; grp < ngroups; grp += nwaves) {
        long long rows[R];
        int head = 0, i0 = 0;
        const bool is_qk = rows_of(grp, rows, head, i0);
        const WT* rowp[R] = {W + rows[0] * K, W + rows[1] * K};
        // rotation coefficients of this pair: loaded before the weight stream so the epilogue never waits on memory
        float rc = 1.f, rs = 0.f;
        if (is_qk) { rc = cs[(long long)pos * half + i0]; rs = sn[(long long)pos * half + i0]; }
        const float sc0 = wscale ? wscale[rows[0]] : 1.f, sc1 = wscale ? wscale[rows[1]] : 1.f;
        float acc[R] = {0.f, 0.f};
        int c0 = 0;
        if (have) { consume_block<T, WT, R, U, true>(wa, xs, 0, lane, nchunk, acc); c0 = STEP; have = false; }
        const int cfull = (nchunk / STEP) * STEP;
        for (; c0 < cfull; c0 += STEP) {
            issue_block<WT, R, U, NT, true>(wa, rowp, c0, lane, nchunk);
            consume_block<T, WT, R, U, true>(wa, xs, c0, lane, nchunk, acc);
        }
        if (c0 < nchunk) {
            issue_block<WT, R, U, NT, false>(wa, rowp, c0, lane, nchunk);
            consume_block<T, WT, R, U, false>(wa, xs, c0, lane, nchunk, acc);
        }
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = wave_sum(acc[r]);
        if (lane == 0) {
            acc[0] *= sc0; acc[1] *= sc1;
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
                    const int hk = (v0 + r) >> (half_log2 + 1), d = (v0 + r) & (hd - 1);
                    T val;
                    Elem<T>::st(&val, acc[r]);
                    vc[((long long)hk * S_max + pos) * hd + d] = val;
                    if (vtc) vtc[((long long)hk * hd + d) * S_max + pos] = val;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// host dispatch  (kernels: `res` and the output may be the same buffer -- in-place residual add -- so neither is __restrict__)
// ------------------------------------------------------------------------------------------------
template <typename T, typename TO, typename WT, int R, int U, bool PF, int XPT = 6>
static int launch_rows(const void* x, const void* W, const float* ws, const void* norm_w, const void* res, void* y, int N,
                       int K, float eps, bool swiglu, hipStream_t st) {
    if (PF && K / Vec16<WT>::N < 64 * U)          // the unconditional prefetch needs one full step per row
        return launch_rows<T, TO, WT, R, U, false>(x, W, ws, norm_w, res, y, N, K, eps, swiglu, st);   // (short rows: the default prologue)
    const int ngroups = swiglu ? cdiv(N / 2, R / 2) : cdiv(N, R);
    int blocks = cdiv(ngroups, GV_WAVES);
    if (blocks > tune().gemv_max_blocks) blocks = tune().gemv_max_blocks;
    const size_t lds = x_image_bytes<RowsImage16<T, WT>::v, Vec16<WT>::N>(K);
#define TEO_GV(NTV, SW)                                                                                              \
    TEO_KLAUNCH((gemv_kernel<T, TO, WT, R, U, PF, NTV, SW, XPT>), blocks, GV_THREADS, lds, st, (const T*)x, (const WT*)W, ws, (const T*)norm_w, \
                (const T*)res, (TO*)y, N, K, eps)
    if (tune().gemv_nt) { if (swiglu) TEO_GV(true, true); else TEO_GV(true, false); }
    else           { if (swiglu) TEO_GV(false, true); else TEO_GV(false, false); }
#undef TEO_GV
    TEO_LAUNCH_CHECK("gemv");
    return TEO_OK;
}

template <typename T, typename TO, typename WT, int R, int U>
static int launch_splitk(const void* x, const void* W, const float* ws, const void* res, void* y, int N, int K, hipStream_t st) {
    const int blocks = cdiv(N, R);
    if (tune().gemv_nt)
        TEO_KLAUNCH((gemv_splitk_kernel<T, TO, WT, R, U, true>), blocks, GV_THREADS, 0, st, (const T*)x, (const WT*)W, ws, (const T*)res, (TO*)y, N, K);
    else
        TEO_KLAUNCH((gemv_splitk_kernel<T, TO, WT, R, U, false>), blocks, GV_THREADS, 0, st, (const T*)x, (const WT*)W, ws, (const T*)res, (TO*)y, N, K);
    TEO_LAUNCH_CHECK("gemv_splitk");
    return TEO_OK;
}

template <typename T, typename TO, typename WT>
static int gemv_launch(const void* x, const void* W, const float* ws, const void* norm_w, const void* res, void* y, int N, int K,
                       float eps, bool swiglu, hipStream_t st) {
    // few long rows without a fused norm (o / down projections): split-K workgroups, 2 rows each (measured best).
    // U is chosen so that ONE step covers the whole row (256*U chunks): every load of the workgroup is in flight at once
    // instead of 2-3 dependent steps (down projection, K = 11008: 3 steps of U = 2 -> 1 step of U = 6).
    if (!swiglu && norm_w == nullptr && N <= 8192 && tune().gemv_variant < 0) {
        const int nchunk = K / Vec16<WT>::N;
        // round 4 sweep (tools/bench_kernels.py gemv_splitk_sweep -> profiles/r04_gemv_splitk_sweep.txt): a row of <= 256 chunks (fp8 o
        // projection) takes U = 1 -- with U = 2 half of the step's instructions are masked (5.39 -> 4.95 us alone, 5.24 -> 4.72 in
        // the step); 4 rows per workgroup are within noise alone and lose in the step (fp8 down 9.94 -> 10.53 us).  The chunk ->
        // (wave, lane) map and every lane's order of accumulation do not depend on R or U: all forms give the same bits (tested)
        const int u = tune().gemv_splitk_u > 0 ? tune().gemv_splitk_u : (nchunk <= 256 ? 1 : (nchunk <= 512 ? 2 : (nchunk <= 1024 ? 4 : (nchunk <= 1536 ? 6 : 2))));
        const int r = tune().gemv_splitk_r > 0 ? tune().gemv_splitk_r : 2;
#define TEO_SPK(RR, UU) if (r == RR && u == UU) return launch_splitk<T, TO, WT, RR, UU>(x, W, ws, res, y, N, K, st)
        TEO_SPK(2, 1); TEO_SPK(2, 3); TEO_SPK(2, 4); TEO_SPK(2, 6);
        TEO_SPK(4, 1); TEO_SPK(4, 2); TEO_SPK(4, 3); TEO_SPK(4, 4); TEO_SPK(4, 6);
#undef TEO_SPK
        return launch_splitk<T, TO, WT, 2, 2>(x, W, ws, res, y, N, K, st);
    }
    switch (tune().gemv_variant) {          // tuning sweep (tools/bench_kernels.py): R rows x U chunks, prefetch on/off
        case 0: return launch_rows<T, TO, WT, 4, 2, false>(x, W, ws, norm_w, res, y, N, K, eps, swiglu, st);
        case 1: return launch_rows<T, TO, WT, 2, 4, false>(x, W, ws, norm_w, res, y, N, K, eps, swiglu, st);
        case 2: return launch_rows<T, TO, WT, 2, 8, true>(x, W, ws, norm_w, res, y, N, K, eps, swiglu, st);
        default: break;
    }
    const bool small_k = K / Vec16<T>::N <= 2 * GV_THREADS && (tune().gemv_small_k != 0);   // the x prologue fits 2 chunks per thread (stage_x)
    // >= 4 KiB contiguous per row per step streams ~7 % faster than 2 KiB; first block prefetched under the prologue.
    // fp8 rows are half as long: 4 rows per wave keep the same bytes in flight per lane
    if constexpr (sizeof(WT) == 1) {
        // fp8 rows are half as long.  With the small prologue (61-95 VGPRs) 2 rows x 4 chunks = 8 KB per wave per step wins
        // (gate/up 16.7 -> 15.3 us, qkv 11.3 -> 10.0, lm_head 23 -> 21.7: at the streaming floor); with the K <= 12288
        // prologue the registers are the prologue's and 2 x 2 keeps the occupancy
        if constexpr (IsBf<T>::v) {
            if (small_k) {
                switch (tune().gemv_variant) {
                    case 11: return launch_rows<T, TO, WT, 4, 2, true, 2>(x, W, ws, norm_w, res, y, N, K, eps, swiglu, st);
                    case 12: return launch_rows<T, TO, WT, 2, 2, true, 2>(x, W, ws, norm_w, res, y, N, K, eps, swiglu, st);
                    case 13: return launch_rows<T, TO, WT, 4, 4, true, 2>(x, W, ws, norm_w, res, y, N, K, eps, swiglu, st);
                    default: return launch_rows<T, TO, WT, 2, 4, true, 2>(x, W, ws, norm_w, res, y, N, K, eps, swiglu, st);
                }
            }
        }
        switch (tune().gemv_variant) {
            case 10: return launch_rows<T, TO, WT, 2, 4, true>(x, W, ws, norm_w, res, y, N, K, eps, swiglu, st);
            case 11: return launch_rows<T, TO, WT, 4, 2, true>(x, W, ws, norm_w, res, y, N, K, eps, swiglu, st);
            case 13: return launch_rows<T, TO, WT, 4, 4, true>(x, W, ws, norm_w, res, y, N, K, eps, swiglu, st);
            default: break;
        }
        return launch_rows<T, TO, WT, 2, 2, true>(x, W, ws, norm_w, res, y, N, K, eps, swiglu, st);
    }
    if constexpr (Is16<T>::v) { if (small_k) return launch_rows<T, TO, WT, 2, 4, true, 2>(x, W, ws, norm_w, res, y, N, K, eps, swiglu, st); }
    return launch_rows<T, TO, WT, 2, 4, true>(x, W, ws, norm_w, res, y, N, K, eps, swiglu, st);
}

// w_dtype: -1 = same as dtype, 2 = fp8 e4m3 with per-row fp32 scales (bf16 activations only)
int gemv_w(const void* x, const void* W, const float* wscale, int w_fp8, const void* norm_w, const void* res, void* y, int N,
           int K, float eps, unsigned flags, int dtype, int out_dtype, hipStream_t st) {
    if (N == 0) return TEO_OK;
    const bool swiglu = flags & TEO_GEMM_SWIGLU16;
    const int ve = w_fp8 ? 16 : (dtype == TEO_F32 ? 4 : 8);
    TEO_CHECK_ARG(K % ve == 0, "teo_gemv: K=%d must be a multiple of %d", K, ve);
    TEO_CHECK_ARG((reinterpret_cast<uintptr_t>(W) & 15) == 0, "teo_gemv: W must be 16-byte aligned");
    TEO_CHECK_ARG((size_t)(K + 1040) * 4 <= 64 * 1024, "teo_gemv: K=%d too large for LDS staging", K);
    TEO_CHECK_ARG((reinterpret_cast<uintptr_t>(x) & 15) == 0 && (norm_w == nullptr || (reinterpret_cast<uintptr_t>(norm_w) & 15) == 0),
                  "teo_gemv: x / norm_w must be 16-byte aligned");
    if (swiglu) TEO_CHECK_ARG(N % 32 == 0 && !res, "teo_gemv: SWIGLU16 needs N %% 32 == 0 and no residual");
    if (w_fp8) {
        TEO_CHECK_ARG(dtype == TEO_BF16 && wscale != nullptr, "teo_gemv: fp8 weights need bf16 activations and per-row scales");
        if (out_dtype == TEO_F32) return gemv_launch<bf16_t, float, fp8_t>(x, W, wscale, norm_w, res, y, N, K, eps, swiglu, st);
        return gemv_launch<bf16_t, bf16_t, fp8_t>(x, W, wscale, norm_w, res, y, N, K, eps, swiglu, st);
    }
    if (dtype == TEO_F32) {
        TEO_CHECK_ARG(out_dtype == TEO_F32, "teo_gemv: f32 inputs need f32 output");
        return gemv_launch<float, float, float>(x, W, nullptr, norm_w, res, y, N, K, eps, swiglu, st);
    }
    if (dtype == TEO_BF16) {
        if (out_dtype == TEO_F32) return gemv_launch<bf16_t, float, bf16_t>(x, W, nullptr, norm_w, res, y, N, K, eps, swiglu, st);
        return gemv_launch<bf16_t, bf16_t, bf16_t>(x, W, nullptr, norm_w, res, y, N, K, eps, swiglu, st);
    }
    if (dtype == TEO_F16) {
        if (out_dtype == TEO_F32) return gemv_launch<f16_t, float, f16_t>(x, W, nullptr, norm_w, res, y, N, K, eps, swiglu, st);
        return gemv_launch<f16_t, f16_t, f16_t>(x, W, nullptr, norm_w, res, y, N, K, eps, swiglu, st);
    }
    set_error("teo_gemv: unknown dtype %d", dtype);
    return TEO_ERR_UNSUPPORTED;
}

int gemv(const void* x, const void* W, const void* norm_w, const void* res, void* y, int N, int K, float eps,
         unsigned flags, int dtype, int out_dtype, hipStream_t st) {
    return gemv_w(x, W, nullptr, 0, norm_w, res, y, N, K, eps, flags, dtype, out_dtype, st);
}

int gemv_qkv_rope(const void* x, const void* W, const float* wscale, int w_fp8, const void* norm_w, void* qout,
                  const float* cs, const float* sn, const int* d_pos, void* kc, void* vc, void* vtc, int S_max, int H, int Hk,
                  int hd, int K, float eps, int dtype, hipStream_t st) {
    const int ve = w_fp8 ? 16 : (dtype == TEO_F32 ? 4 : 8);
    TEO_CHECK_ARG(K % ve == 0 && hd >= 2 && (hd & (hd - 1)) == 0, "gemv_qkv_rope: K=%d hd=%d (head_dim must be a power of two)", K, hd);
    TEO_CHECK_ARG((size_t)(K + 1040) * 4 <= 64 * 1024, "gemv_qkv_rope: K=%d too large for LDS staging", K);
    TEO_CHECK_ARG(!w_fp8 || (dtype == TEO_BF16 && wscale), "gemv_qkv_rope: fp8 weights need bf16 activations and scales");
    const int ngroups = (H + Hk) * (hd / 2) + Hk * hd / 2;
    int blocks = cdiv(ngroups, GV_WAVES);
    if (blocks > tune().gemv_max_blocks) blocks = tune().gemv_max_blocks;
    const size_t lds = w_fp8 ? xb_lds_bytes<16>(K) : (dtype == TEO_F32 ? xs_lds_bytes<4>(K) : xs_lds_bytes<8>(K));
    // small x prologue (2 register chunks per thread) whenever K allows: fewer VGPRs, more waves per SIMD (see stage_x); fp8 rows then
    // take 4 chunks per step like the row-group kernel (8 KB per wave in flight)
    const bool small_k = (tune().gemv_small_k != 0) && K / (dtype == TEO_F32 ? 4 : 8) <= 2 * GV_THREADS;
    const int uu = (w_fp8 && !small_k) ? 2 : 4;
    const bool pf = K / ve >= 64 * uu;
#define TEO_QR2(TT, WW, NTV, XP, UU)                                                                                      \
    if (pf) TEO_KLAUNCH((gemv_qkv_rope_kernel<TT, WW, NTV, true, UU, XP>), blocks, GV_THREADS, lds, st, (const WW*)W, (const TT*)x, (const TT*)norm_w, wscale, \
                                                                       d_pos, K, H, Hk, hd, (TT*)qout, cs, sn, (TT*)kc, (TT*)vc, (TT*)vtc, \
                                                                       S_max, eps);                                          \
    else TEO_KLAUNCH((gemv_qkv_rope_kernel<TT, WW, NTV, false, UU, XP>), blocks, GV_THREADS, lds, st, (const WW*)W, (const TT*)x, (const TT*)norm_w, wscale, \
                                                                       d_pos, K, H, Hk, hd, (TT*)qout, cs, sn, (TT*)kc, (TT*)vc, (TT*)vtc, \
                                                                       S_max, eps)
#define TEO_QR(TT, WW, NTV) if (small_k) { TEO_QR2(TT, WW, NTV, 2, 4); } else { TEO_QR2(TT, WW, NTV, 6, 4); }
#define TEO_QR6(TT, WW, NTV) TEO_QR2(TT, WW, NTV, 6, 4)
#define TEO_QR8(TT, WW, NTV) if (small_k) { TEO_QR2(TT, WW, NTV, 2, 4); } else { TEO_QR2(TT, WW, NTV, 6, 2); }
    if (w_fp8)                 { if (tune().gemv_nt) { TEO_QR8(bf16_t, fp8_t, true); } else { TEO_QR8(bf16_t, fp8_t, false); } }
    else if (dtype == TEO_F32) { if (tune().gemv_nt) { TEO_QR6(float, float, true); } else { TEO_QR6(float, float, false); } }
    else if (dtype == TEO_F16) { if (tune().gemv_nt) { TEO_QR(f16_t, f16_t, true); } else { TEO_QR(f16_t, f16_t, false); } }
    else                       { if (tune().gemv_nt) { TEO_QR(bf16_t, bf16_t, true); } else { TEO_QR(bf16_t, bf16_t, false); } }
#undef TEO_QR
#undef TEO_QR6
#undef TEO_QR8
#undef TEO_QR2
    TEO_LAUNCH_CHECK("gemv_qkv_rope");
    return TEO_OK;
}

}  // namespace teo
