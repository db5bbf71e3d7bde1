// Row-wise normalisation kernels: LayerNorm (CLIP ViT), RMSNorm (LLaMA), fused ViT embedding assemble + pre-LN.
// HBM-bound, one 256-thread workgroup per row, fp32 statistics, two passes over an L1/L2-hot row.
#include "common.h"

namespace teo {

template <typename T>
__global__ __launch_bounds__(256) void layernorm_kernel(const T* __restrict__ x, const T* __restrict__ w,
                                                        const T* __restrict__ b, T* __restrict__ y, int dim,
                                                        float eps) {
    __shared__ float red[4];
    const long long row = blockIdx.x;
    const T* xr = x + row * dim;
    T* yr = y + row * dim;
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

template <typename T>
__global__ __launch_bounds__(256) void rmsnorm_kernel(const T* __restrict__ x, const T* __restrict__ w,
                                                      T* __restrict__ y, int dim, float eps) {
    __shared__ float red[4];
    const long long row = blockIdx.x;
    const T* xr = x + row * dim;
    T* yr = y + row * dim;
    float s = 0.f;
    for (int i = threadIdx.x; i < dim; i += 256) {
        float v = Elem<T>::ld(xr + i);
        s += v * v;
    }
    const float r = rsqrtf(block_sum<256>(s, red) / dim + eps);
    for (int i = threadIdx.x; i < dim; i += 256) Elem<T>::st(yr + i, Elem<T>::ld(xr + i) * r * Elem<T>::ld(w + i));
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

int layernorm(const void* x, const void* w, const void* b, void* y, int rows, int dim, float eps, int dtype,
              hipStream_t st) {
    if (rows == 0) return TEO_OK;
    if (dtype == TEO_F32)
        layernorm_kernel<float><<<rows, 256, 0, st>>>((const float*)x, (const float*)w, (const float*)b, (float*)y, dim, eps);
    else
        layernorm_kernel<bf16_t><<<rows, 256, 0, st>>>((const bf16_t*)x, (const bf16_t*)w, (const bf16_t*)b, (bf16_t*)y, dim, eps);
    TEO_LAUNCH_CHECK("layernorm");
    return TEO_OK;
}

int rmsnorm(const void* x, const void* w, void* y, int rows, int dim, float eps, int dtype, hipStream_t st) {
    if (rows == 0) return TEO_OK;
    if (dtype == TEO_F32)
        rmsnorm_kernel<float><<<rows, 256, 0, st>>>((const float*)x, (const float*)w, (float*)y, dim, eps);
    else
        rmsnorm_kernel<bf16_t><<<rows, 256, 0, st>>>((const bf16_t*)x, (const bf16_t*)w, (bf16_t*)y, dim, eps);
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
    else
        vit_embed_ln_kernel<bf16_t><<<rows, 256, lds, st>>>((const bf16_t*)patch, (const bf16_t*)cls, (const bf16_t*)pos,
                                                            (const bf16_t*)w, (const bf16_t*)b, (bf16_t*)out, NP, dim, eps);
    TEO_LAUNCH_CHECK("vit_embed_ln");
    return TEO_OK;
}

}  // namespace teo
