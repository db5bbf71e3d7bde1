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

// R rows per wave per pass.  SWIGLU: rows come in (gate, up) pairs 16 apart inside 32-row blocks.
template <typename T, typename TO, int R, bool SWIGLU>
__global__ __launch_bounds__(GV_THREADS) void gemv_kernel(const T* __restrict__ x, const T* __restrict__ W,
                                                          const T* __restrict__ norm_w, const T* __restrict__ res,
                                                          TO* __restrict__ y, int N, int K, float eps) {
    extern __shared__ __attribute__((aligned(16))) float xs[];     // [K] fp32 (+4 floats of reduction scratch)
    constexpr int VE = Vec16<T>::N;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    float* red = xs + K;
    // ---- stage f(x) in LDS
    if (norm_w) {
        float ss = 0.f;
        for (int i = tid; i < K; i += GV_THREADS) {
            const float v = Elem<T>::ld(x + i);
            xs[i] = v;
            ss += v * v;
        }
        const float r = rsqrtf(block_sum<GV_THREADS>(ss, red) / K + eps);
        for (int i = tid; i < K; i += GV_THREADS) xs[i] = Elem<T>::round(xs[i] * r * Elem<T>::ld(norm_w + i));
    } else {
        for (int i = tid; i < K; i += GV_THREADS) xs[i] = Elem<T>::ld(x + i);
    }
    __syncthreads();

    const int nchunk = K / VE;                    // 16-byte chunks per row
    const int wave_global = blockIdx.x * GV_WAVES + wid;
    const int nwaves = gridDim.x * GV_WAVES;
    // row groups: plain -> R consecutive rows; swiglu -> R/2 (gate, up) pairs
    const int ngroups = SWIGLU ? (N / 2 + (R / 2) - 1) / (R / 2) : (N + R - 1) / R;
    for (int grp = wave_global; grp < ngroups; grp += nwaves) {
        int rows[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (SWIGLU) {
                const int j = grp * (R / 2) + (r >> 1);                 // output column
                const int jr = min(j, N / 2 - 1);
                rows[r] = (jr >> 4) * 32 + (jr & 15) + ((r & 1) ? 16 : 0);
            } else {
                rows[r] = min(grp * R + r, N - 1);
            }
        }
        float acc[R];
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = 0.f;
        for (int c = lane; c < nchunk; c += 128) {
            // two chunks per lane per iteration -> 2R loads in flight
            const int c2 = c + 64;
            const bool has2 = c2 < nchunk;
            uint4 w0[R], w1[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const T* wr = W + (long long)rows[r] * K;
                w0[r] = ld_nt16(wr + (long long)c * VE);
                w1[r] = has2 ? ld_nt16(wr + (long long)c2 * VE) : make_uint4(0, 0, 0, 0);
            }
            float xa[VE], xb[VE];
#pragma unroll
            for (int e = 0; e < VE; e += 4) {
                const float4 t = *reinterpret_cast<const float4*>(xs + c * VE + e);
                xa[e] = t.x; xa[e + 1] = t.y; xa[e + 2] = t.z; xa[e + 3] = t.w;
                if (has2) {
                    const float4 u = *reinterpret_cast<const float4*>(xs + c2 * VE + e);
                    xb[e] = u.x; xb[e + 1] = u.y; xb[e + 2] = u.z; xb[e + 3] = u.w;
                } else {
                    xb[e] = xb[e + 1] = xb[e + 2] = xb[e + 3] = 0.f;
                }
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                float f[VE], g[VE];
                Vec16<T>::cvt(w0[r], f);
                Vec16<T>::cvt(w1[r], g);
#pragma unroll
                for (int e = 0; e < VE; ++e) acc[r] = fmaf(f[e], xa[e], acc[r]);
#pragma unroll
                for (int e = 0; e < VE; ++e) acc[r] = fmaf(g[e], xb[e], acc[r]);
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

template <typename T, typename TO>
static int gemv_launch(const void* x, const void* W, const void* norm_w, const void* res, void* y, int N, int K, float eps,
                       bool swiglu, hipStream_t st) {
    constexpr int R = 4;
    const int ngroups = swiglu ? cdiv(N / 2, R / 2) : cdiv(N, R);
    int blocks = cdiv(ngroups, GV_WAVES);
    if (blocks > 2048) blocks = 2048;
    const size_t lds = (size_t)(K + 4) * sizeof(float);
    if (swiglu)
        gemv_kernel<T, TO, R, true><<<blocks, GV_THREADS, lds, st>>>((const T*)x, (const T*)W, (const T*)norm_w,
                                                                     (const T*)res, (TO*)y, N, K, eps);
    else
        gemv_kernel<T, TO, R, false><<<blocks, GV_THREADS, lds, st>>>((const T*)x, (const T*)W, (const T*)norm_w,
                                                                      (const T*)res, (TO*)y, N, K, eps);
    TEO_LAUNCH_CHECK("gemv");
    return TEO_OK;
}

int gemv(const void* x, const void* W, const void* norm_w, const void* res, void* y, int N, int K, float eps,
         unsigned flags, int dtype, int out_dtype, hipStream_t st) {
    if (N == 0) return TEO_OK;
    const bool swiglu = flags & TEO_GEMM_SWIGLU16;
    const int ve = dtype == TEO_F32 ? 4 : 8;
    TEO_CHECK_ARG(K % ve == 0, "teo_gemv: K=%d must be a multiple of %d", K, ve);
    TEO_CHECK_ARG((reinterpret_cast<uintptr_t>(W) & 15) == 0, "teo_gemv: W must be 16-byte aligned");
    TEO_CHECK_ARG((size_t)(K + 4) * 4 <= 160 * 1024, "teo_gemv: K=%d too large for LDS staging", K);
    if (swiglu) TEO_CHECK_ARG(N % 32 == 0 && !res, "teo_gemv: SWIGLU16 needs N %% 32 == 0 and no residual");
    if (dtype == TEO_F32) {
        TEO_CHECK_ARG(out_dtype == TEO_F32, "teo_gemv: f32 inputs need f32 output");
        return gemv_launch<float, float>(x, W, norm_w, res, y, N, K, eps, swiglu, st);
    }
    if (dtype == TEO_BF16) {
        if (out_dtype == TEO_F32) return gemv_launch<bf16_t, float>(x, W, norm_w, res, y, N, K, eps, swiglu, st);
        return gemv_launch<bf16_t, bf16_t>(x, W, norm_w, res, y, N, K, eps, swiglu, st);
    }
    set_error("teo_gemv: unknown dtype %d", dtype);
    return TEO_ERR_UNSUPPORTED;
}

}  // namespace teo
