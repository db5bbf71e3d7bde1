// Row-wise normalisation kernels: LayerNorm (CLIP ViT), RMSNorm (LLaMA), fused ViT embedding assemble + pre-LN.
// HBM-bound, one 256-thread workgroup per row, fp32 statistics, two passes over an L1/L2-hot row.
#include "common.h"

namespace teo {

// 16-byte vector access helpers (rows are 16-byte aligned whenever dim % VE == 0 and the base pointer is)
template <typename T> struct V16;
template <> struct V16<bf16_t> {
    static constexpr int N = 8;
    __device__ static __forceinline__ void unpack(const uint4& r, float* f) {
        f[0] = __uint_as_float(r.x << 16); f[1] = __uint_as_float(r.x & 0xffff0000u);
        f[2] = __uint_as_float(r.y << 16); f[3] = __uint_as_float(r.y & 0xffff0000u);
        f[4] = __uint_as_float(r.z << 16); f[5] = __uint_as_float(r.z & 0xffff0000u);
        f[6] = __uint_as_float(r.w << 16); f[7] = __uint_as_float(r.w & 0xffff0000u);
    }
    __device__ static __forceinline__ uint4 pack(const float* f) {
        return make_uint4(pack_bf2(f[0], f[1]), pack_bf2(f[2], f[3]), pack_bf2(f[4], f[5]), pack_bf2(f[6], f[7]));
    }
};
template <> struct V16<f16_t> {
    static constexpr int N = 8;
    __device__ static __forceinline__ void unpack(const uint4& r, float* f) {
        f[0] = h_lo<true>(r.x); f[1] = h_hi<true>(r.x); f[2] = h_lo<true>(r.y); f[3] = h_hi<true>(r.y);
        f[4] = h_lo<true>(r.z); f[5] = h_hi<true>(r.z); f[6] = h_lo<true>(r.w); f[7] = h_hi<true>(r.w);
    }
    __device__ static __forceinline__ uint4 pack(const float* f) {
        return make_uint4(pack_f16x2(f[0], f[1]), pack_f16x2(f[2], f[3]), pack_f16x2(f[4], f[5]), pack_f16x2(f[6], f[7]));
    }
};
template <> struct V16<float> {
    static constexpr int N = 4;
    __device__ static __forceinline__ void unpack(const uint4& r, float* f) {
        f[0] = __uint_as_float(r.x); f[1] = __uint_as_float(r.y); f[2] = __uint_as_float(r.z); f[3] = __uint_as_float(r.w);
    }
    __device__ static __forceinline__ uint4 pack(const float* f) {
        return make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3]));
    }
};

// VEC: the row is held in registers as VPT 16-byte vectors per thread (one global read), else scalar 3-pass fallback
template <typename T, bool VEC, int VPT>
__global__ __launch_bounds__(256) void layernorm_kernel(const T* __restrict__ x, const T* __restrict__ w,
                                                        const T* __restrict__ b, T* __restrict__ y, int dim,
                                                        float eps) {
    __shared__ float red[4];
    const long long row = blockIdx.x;
    const T* xr = x + row * dim;
    T* yr = y + row * dim;
    if (VEC) {
        constexpr int VE = V16<T>::N;
        const int nv = dim / VE;
        float v[VPT][VE];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int c = threadIdx.x + i * 256;
            if (c < nv) {
                V16<T>::unpack(*reinterpret_cast<const uint4*>(xr + (long long)c * VE), v[i]);
#pragma unroll
                for (int e = 0; e < VE; ++e) s += v[i][e];
            }
        }
        const float mean = block_sum<256>(s, red) / dim;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int c = threadIdx.x + i * 256;
            if (c < nv) {
#pragma unroll
                for (int e = 0; e < VE; ++e) { const float d = v[i][e] - mean; q += d * d; }
            }
        }
        const float rstd = rsqrtf(block_sum<256>(q, red) / dim + eps);
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int c = threadIdx.x + i * 256;
            if (c < nv) {
                float wv[VE], bv[VE], o[VE];
                V16<T>::unpack(*reinterpret_cast<const uint4*>(w + (long long)c * VE), wv);
                V16<T>::unpack(*reinterpret_cast<const uint4*>(b + (long long)c * VE), bv);
#pragma unroll
                for (int e = 0; e < VE; ++e) o[e] = (v[i][e] - mean) * rstd * wv[e] + bv[e];
                *reinterpret_cast<uint4*>(yr + (long long)c * VE) = V16<T>::pack(o);
            }
        }
        return;
    }
    float s = 0.f;
    for (int i = threadIdx.x; i < dim; i += 256) s += Elem<T>::ld(xr + i);
    const float mean = block_sum<256>(s, red) / dim;
    float v = 0.f;
    for (int i = threadIdx.x; i < dim; i += 256) {
        float d = Elem<T>::ld(xr + i) - mean;
        v += d * d;
    }
    const float rstd = rsqrtf(block_sum<256>(v, red) / dim + eps);
    for (int i = threadIdx.x; i < dim; i += 256)
        Elem<T>::st(yr + i, (Elem<T>::ld(xr + i) - mean) * rstd * Elem<T>::ld(w + i) + Elem<T>::ld(b + i));
}

// NT threads per row: 256, or 128 for many-row launches (16 instead of 8 workgroups per CU: a 2168-row prefill norm fits
// one round of resident workgroups instead of 1.06)
template <typename T, bool VEC, int VPT, int NT>
__global__ __launch_bounds__(NT) void rmsnorm_kernel(const T* __restrict__ x, const T* __restrict__ w,
                                                      T* __restrict__ y, int dim, float eps) {
    __shared__ float red[4];
    const long long row = blockIdx.x;
    const T* xr = x + row * dim;
    T* yr = y + row * dim;
    if (VEC) {
        constexpr int VE = V16<T>::N;
        const int nv = dim / VE;
        float v[VPT][VE];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int c = threadIdx.x + i * NT;
            if (c < nv) {
                V16<T>::unpack(*reinterpret_cast<const uint4*>(xr + (long long)c * VE), v[i]);
#pragma unroll
                for (int e = 0; e < VE; ++e) s = fmaf(v[i][e], v[i][e], s);
            }
        }
        const float r = rsqrtf(block_sum<NT>(s, red) / dim + eps);
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int c = threadIdx.x + i * NT;
            if (c < nv) {
                float wv[VE], o[VE];
                V16<T>::unpack(*reinterpret_cast<const uint4*>(w + (long long)c * VE), wv);
#pragma unroll
                for (int e = 0; e < VE; ++e) o[e] = v[i][e] * r * wv[e];
                *reinterpret_cast<uint4*>(yr + (long long)c * VE) = V16<T>::pack(o);
            }
        }
        return;
    }
    float s = 0.f;
    for (int i = threadIdx.x; i < dim; i += NT) {
        float v = Elem<T>::ld(xr + i);
        s += v * v;
    }
    const float r = rsqrtf(block_sum<NT>(s, red) / dim + eps);
    for (int i = threadIdx.x; i < dim; i += NT) Elem<T>::st(yr + i, Elem<T>::ld(xr + i) * r * Elem<T>::ld(w + i));
}

// row (t, n): n == 0 -> cls + pos[0]; else patch[t*NP + n-1] + pos[n]; the sum is rounded to T (the
// reference materialises the embeddings tensor) and then layer-normalised.
template <typename T>
__global__ __launch_bounds__(256) void vit_embed_ln_kernel(const T* __restrict__ patch, const T* __restrict__ cls,
                                                           const T* __restrict__ pos, const T* __restrict__ w,
                                                           const T* __restrict__ b, T* __restrict__ out, int NP,
                                                           int dim, float eps) {
    __shared__ float red[4];
    extern __shared__ __attribute__((aligned(16))) float rowbuf[];
    const int t = blockIdx.x / (NP + 1), n = blockIdx.x % (NP + 1);
    const T* src = (n == 0) ? cls : patch + ((long long)t * NP + (n - 1)) * dim;
    const T* pr = pos + (long long)n * dim;
    float s = 0.f;
    for (int i = threadIdx.x; i < dim; i += 256) {
        float v = Elem<T>::round(Elem<T>::ld(src + i) + Elem<T>::ld(pr + i));
        rowbuf[i] = v;
        s += v;
    }
    const float mean = block_sum<256>(s, red) / dim;
    float v2 = 0.f;
    for (int i = threadIdx.x; i < dim; i += 256) {
        float d = rowbuf[i] - mean;
        v2 += d * d;
    }
    const float rstd = rsqrtf(block_sum<256>(v2, red) / dim + eps);
    T* yr = out + (long long)blockIdx.x * dim;
    for (int i = threadIdx.x; i < dim; i += 256)
        Elem<T>::st(yr + i, (rowbuf[i] - mean) * rstd * Elem<T>::ld(w + i) + Elem<T>::ld(b + i));
}

static bool vec_ok(int dim, int ve, int vpt, const void* a, const void* b_, const void* c, const void* d) {
    auto al = [](const void* p) { return p == nullptr || (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    return dim % ve == 0 && dim / ve <= vpt * 256 && al(a) && al(b_) && al(c) && al(d);
}

int layernorm(const void* x, const void* w, const void* b, void* y, int rows, int dim, float eps, int dtype,
              hipStream_t st) {
    if (rows == 0) return TEO_OK;
    if (dtype == TEO_F32) {
        if (vec_ok(dim, 4, 4, x, w, b, y))
            layernorm_kernel<float, true, 4><<<rows, 256, 0, st>>>((const float*)x, (const float*)w, (const float*)b, (float*)y, dim, eps);
        else
            layernorm_kernel<float, false, 1><<<rows, 256, 0, st>>>((const float*)x, (const float*)w, (const float*)b, (float*)y, dim, eps);
    } else if (dtype == TEO_F16) {
        if (vec_ok(dim, 8, 2, x, w, b, y))
            layernorm_kernel<f16_t, true, 2><<<rows, 256, 0, st>>>((const f16_t*)x, (const f16_t*)w, (const f16_t*)b, (f16_t*)y, dim, eps);
        else
            layernorm_kernel<f16_t, false, 1><<<rows, 256, 0, st>>>((const f16_t*)x, (const f16_t*)w, (const f16_t*)b, (f16_t*)y, dim, eps);
    } else {
        if (vec_ok(dim, 8, 2, x, w, b, y))
            layernorm_kernel<bf16_t, true, 2><<<rows, 256, 0, st>>>((const bf16_t*)x, (const bf16_t*)w, (const bf16_t*)b, (bf16_t*)y, dim, eps);
        else
            layernorm_kernel<bf16_t, false, 1><<<rows, 256, 0, st>>>((const bf16_t*)x, (const bf16_t*)w, (const bf16_t*)b, (bf16_t*)y, dim, eps);
    }
    TEO_LAUNCH_CHECK("layernorm");
    return TEO_OK;
}

int rmsnorm(const void* x, const void* w, void* y, int rows, int dim, float eps, int dtype, hipStream_t st) {
    if (rows == 0) return TEO_OK;
    if (dtype == TEO_F32) {
        if (vec_ok(dim, 4, 4, x, w, nullptr, y))
            rmsnorm_kernel<float, true, 4, 256><<<rows, 256, 0, st>>>((const float*)x, (const float*)w, (float*)y, dim, eps);
        else
            rmsnorm_kernel<float, false, 1, 256><<<rows, 256, 0, st>>>((const float*)x, (const float*)w, (float*)y, dim, eps);
    } else if (dtype == TEO_F16) {
        if (rows >= 1024 && vec_ok(dim, 8, 2, x, w, nullptr, y))
            rmsnorm_kernel<f16_t, true, 4, 128><<<rows, 128, 0, st>>>((const f16_t*)x, (const f16_t*)w, (f16_t*)y, dim, eps);
        else if (vec_ok(dim, 8, 2, x, w, nullptr, y))
            rmsnorm_kernel<f16_t, true, 2, 256><<<rows, 256, 0, st>>>((const f16_t*)x, (const f16_t*)w, (f16_t*)y, dim, eps);
        else
            rmsnorm_kernel<f16_t, false, 1, 256><<<rows, 256, 0, st>>>((const f16_t*)x, (const f16_t*)w, (f16_t*)y, dim, eps);
    } else {
        if (rows >= 1024 && vec_ok(dim, 8, 2, x, w, nullptr, y))        // dim/8 <= 512 chunks = 4 per thread at 128 threads
            rmsnorm_kernel<bf16_t, true, 4, 128><<<rows, 128, 0, st>>>((const bf16_t*)x, (const bf16_t*)w, (bf16_t*)y, dim, eps);
        else if (vec_ok(dim, 8, 2, x, w, nullptr, y))
            rmsnorm_kernel<bf16_t, true, 2, 256><<<rows, 256, 0, st>>>((const bf16_t*)x, (const bf16_t*)w, (bf16_t*)y, dim, eps);
        else
            rmsnorm_kernel<bf16_t, false, 1, 256><<<rows, 256, 0, st>>>((const bf16_t*)x, (const bf16_t*)w, (bf16_t*)y, dim, eps);
    }
    TEO_LAUNCH_CHECK("rmsnorm");
    return TEO_OK;
}

int vit_embed_ln(const void* patch, const void* cls, const void* pos, const void* w, const void* b, void* out, int T,
                 int NP, int dim, float eps, int dtype, hipStream_t st) {
    const int rows = T * (NP + 1);
    if (rows == 0) return TEO_OK;
    const size_t lds = (size_t)dim * sizeof(float);
    if (dtype == TEO_F32)
        vit_embed_ln_kernel<float><<<rows, 256, lds, st>>>((const float*)patch, (const float*)cls, (const float*)pos,
                                                           (const float*)w, (const float*)b, (float*)out, NP, dim, eps);
    else if (dtype == TEO_F16)
        vit_embed_ln_kernel<f16_t><<<rows, 256, lds, st>>>((const f16_t*)patch, (const f16_t*)cls, (const f16_t*)pos,
                                                           (const f16_t*)w, (const f16_t*)b, (f16_t*)out, NP, dim, eps);
    else
        vit_embed_ln_kernel<bf16_t><<<rows, 256, lds, st>>>((const bf16_t*)patch, (const bf16_t*)cls, (const bf16_t*)pos,
                                                            (const bf16_t*)w, (const bf16_t*)b, (bf16_t*)out, NP, dim, eps);
    TEO_LAUNCH_CHECK("vit_embed_ln");
    return TEO_OK;
}

}  // namespace teo
