// w8a8 prefill GEMM on the block-scaled fp8 MFMA of CDNA4 (config C5: "fp8 weight path on CDNA4 MFMA"):
//
//     C[m, n] = act_scale[m] * w_scale[n] * sum_k A8[m, k] * W8[n, k]        (+ residual, or the SwiGLU pairing)
//
// A8 = activations quantised per TOKEN to OCP e4m3 (quant_rows_fp8 / rmsnorm_quant_fp8 below), W8 = the per-output-row e4m3
// weights the decode path already streams (teo_llama_desc *_w8 / *_s).  The product runs on
// v_mfma_scale_f32_16x16x128_f8f6f4 with every block scale = 2^0 (E8M0 0x7F): the instruction is used for its K = 128 depth
// -- twice the dense rate of the bf16 MFMA and half the operand bytes -- and the two fp32 scale vectors are applied once in
// the epilogue (they factor out of the sum).  fp8 x fp8 products are exact in fp32, so the only rounding is the fp32
// accumulation: against the dequantised operands the kernel is exact to accumulation order.
//
// Same skeleton as gemm_mfma_bf16_kernel (gemm.hip): 128 x 128 workgroup tile, 4 waves of 64 x 64, K tile = 128 bytes per row
// (the same 128-byte LDS rows, 16-byte chunks XOR-swizzled by row & 7), register-staged global loads one K tile ahead, LDS
// double buffer, XCD-aware tile order, swapped operands (a lane owns 4 consecutive n of one m row).
// MFMA operand layout (16 x 128 per operand): lane l holds row l & 15, bytes 32 * (l >> 4) .. + 32 of the K tile.
// Roofline: MFMA fp8 dense (~5 PFLOP/s); algorithmic FLOPs = 2 * M * N * K.
#include "common.h"
#include "ops.h"

namespace teo {

typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f8_f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int f8_u32x4;

constexpr int F8_BM = 128, F8_BN = 128, F8_BK = 128;          // BK in bytes == elements
constexpr int F8_TILE = F8_BM * F8_BK;                        // 16 KiB per operand tile

__device__ __forceinline__ int f8_xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    const int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    return base + (bid >> 3);
}

template <bool SWIGLU, bool OUT_F32>
__global__ __launch_bounds__(256, 2) void gemm_mfma_fp8_kernel(const unsigned char* __restrict__ A, const float* __restrict__ a_scale,
                                                            const unsigned char* __restrict__ W, const float* __restrict__ w_scale,
                                                            const bf16_t* res, void* Cv, int M, int N, int K, int lda, int ldc,
                                                            int tiles_m, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int tile = f8_xcd_remap(blockIdx.x, tiles_m * tiles_n);
    const int tm = tile % tiles_m, tn = tile / tiles_m;
    const int m0 = tm * F8_BM, n0 = tn * F8_BN;

    const unsigned char* ag[4];
    const unsigned char* wg[4];
    int soff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int id = tid + 256 * i;
        const int row = id >> 3, c = id & 7;
        ag[i] = A + (long long)min(m0 + row, M - 1) * lda + c * 16;
        wg[i] = W + (long long)min(n0 + row, N - 1) * K + c * 16;
        soff[i] = row * F8_BK + ((c ^ (row & 7)) << 4);
    }
    f8_u32x4 ra[4], rb[4];
    f8_f32x4 acc[4][4];   // [ni][mi]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f8_f32x4){0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fg = lane >> 4;
    const int nk = K / F8_BK;

#define TEO_F8_GLOAD(KT)                                                                     \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                          \
        ra[i] = *reinterpret_cast<const f8_u32x4*>(ag[i] + (long long)(KT) * F8_BK);         \
        rb[i] = *reinterpret_cast<const f8_u32x4*>(wg[i] + (long long)(KT) * F8_BK);         \
    }
#define TEO_F8_SWRITE(BUF)                                                                   \
    {                                                                                        \
        unsigned char* sa_ = smem + (BUF) * (2 * F8_TILE);                                   \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                      \
            *reinterpret_cast<f8_u32x4*>(sa_ + soff[i]) = ra[i];                             \
            *reinterpret_cast<f8_u32x4*>(sa_ + F8_TILE + soff[i]) = rb[i];                   \
        }                                                                                    \
    }
#define TEO_F8_FRAG(BASE, ROW)                                                               \
    ({                                                                                       \
        const unsigned char* rp_ = (BASE) + (ROW) * F8_BK;                                   \
        const f8_u32x4 lo_ = *reinterpret_cast<const f8_u32x4*>(rp_ + (((2 * fg) ^ ((ROW) & 7)) << 4));      \
        const f8_u32x4 hi_ = *reinterpret_cast<const f8_u32x4*>(rp_ + (((2 * fg + 1) ^ ((ROW) & 7)) << 4));  \
        i32x8 f_;                                                                            \
        f_[0] = (int)lo_[0]; f_[1] = (int)lo_[1]; f_[2] = (int)lo_[2]; f_[3] = (int)lo_[3];  \
        f_[4] = (int)hi_[0]; f_[5] = (int)hi_[1]; f_[6] = (int)hi_[2]; f_[7] = (int)hi_[3];  \
        f_;                                                                                  \
    })
#define TEO_F8_COMPUTE(BUF)                                                                  \
    {                                                                                        \
        const unsigned char* sA = smem + (BUF) * (2 * F8_TILE);                              \
        const unsigned char* sB = sA + F8_TILE;                                              \
        i32x8 af[4], wf[4];                                                                  \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                      \
            af[i] = TEO_F8_FRAG(sA, wm * 64 + i * 16 + fr);                                  \
            wf[i] = TEO_F8_FRAG(sB, wn * 64 + i * 16 + fr);                                  \
        }                                                                                    \
        _Pragma("unroll") for (int ni = 0; ni < 4; ++ni)                                     \
            _Pragma("unroll") for (int mi = 0; mi < 4; ++mi)                                 \
                acc[ni][mi] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wf[ni], af[mi], acc[ni][mi], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F); \
    }

    TEO_F8_GLOAD(0);
    TEO_F8_SWRITE(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) TEO_F8_GLOAD(kt + 1);
        if (kt & 1) { TEO_F8_COMPUTE(1); } else { TEO_F8_COMPUTE(0); }
        if (kt + 1 < nk) {
            if (kt & 1) { TEO_F8_SWRITE(0); } else { TEO_F8_SWRITE(1); }
        }
        __syncthreads();
    }
#undef TEO_F8_GLOAD
#undef TEO_F8_SWRITE
#undef TEO_F8_FRAG
#undef TEO_F8_COMPUTE

    // epilogue: lane holds C[m = mw + mi*16 + fr][n = nw + ni*16 + fg*4 + r], r = 0..3
    const int mw = m0 + wm * 64, nw = n0 + wn * 64;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        const int m = mw + mi * 16 + fr;
        if (m >= M) continue;
        const float sa = a_scale[m];
        if (SWIGLU) {
#pragma unroll
            for (int ni = 0; ni < 4; ni += 2) {
                const int ng = nw + ni * 16 + fg * 4;              // gate rows; up rows are +16 (N % 32 == 0)
                if (ng >= N) continue;
                const int oc = (nw >> 1) + (ni >> 1) * 16 + fg * 4;
                const float4 sg = *reinterpret_cast<const float4*>(w_scale + ng);
                const float4 su = *reinterpret_cast<const float4*>(w_scale + ng + 16);
                float o[4];
                o[0] = silu(acc[ni][mi][0] * (sa * sg.x)) * (acc[ni + 1][mi][0] * (sa * su.x));
                o[1] = silu(acc[ni][mi][1] * (sa * sg.y)) * (acc[ni + 1][mi][1] * (sa * su.y));
                o[2] = silu(acc[ni][mi][2] * (sa * sg.z)) * (acc[ni + 1][mi][2] * (sa * su.z));
                o[3] = silu(acc[ni][mi][3] * (sa * sg.w)) * (acc[ni + 1][mi][3] * (sa * su.w));
                if (OUT_F32) *reinterpret_cast<float4*>(reinterpret_cast<float*>(Cv) + (long long)m * ldc + oc) = make_float4(o[0], o[1], o[2], o[3]);
                else *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(Cv) + (long long)m * ldc + oc) = make_uint2(pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3]));
            }
        } else {
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                const int n = nw + ni * 16 + fg * 4;
                if (n >= N) continue;                              // N % 4 == 0: whole group in or out
                const float4 sw = *reinterpret_cast<const float4*>(w_scale + n);
                float o[4] = {acc[ni][mi][0] * (sa * sw.x), acc[ni][mi][1] * (sa * sw.y), acc[ni][mi][2] * (sa * sw.z), acc[ni][mi][3] * (sa * sw.w)};
                if (res) {
                    const uint2 q = *reinterpret_cast<const uint2*>(res + (long long)m * ldc + n);
                    o[0] += bf2f((bf16_t)(q.x & 0xffff)); o[1] += bf2f((bf16_t)(q.x >> 16));
                    o[2] += bf2f((bf16_t)(q.y & 0xffff)); o[3] += bf2f((bf16_t)(q.y >> 16));
                }
                if (OUT_F32) *reinterpret_cast<float4*>(reinterpret_cast<float*>(Cv) + (long long)m * ldc + n) = make_float4(o[0], o[1], o[2], o[3]);
                else *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(Cv) + (long long)m * ldc + n) = make_uint2(pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3]));
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// wide-tile form (the structure of gemm_wide.hip): 128 (M) x 256 (N) workgroup tile, 8 waves of 64 x 64, K tile = 128 fp8
// elements = the same 128-byte LDS rows; LDS-DMA operand staging into a three-stage ring (3 x 48 KB), counted vmcnt + one raw
// barrier per K tile.  Same k-order as gemm_mfma_fp8_kernel -> bit-identical results.
// ------------------------------------------------------------------------------------------------
constexpr int F8W_BM = 128, F8W_BN = 256;
constexpr int F8W_A_BYTES = F8W_BM * F8_BK, F8W_W_BYTES = F8W_BN * F8_BK, F8W_STAGE = F8W_A_BYTES + F8W_W_BYTES;
constexpr int F8W_PIECES = F8W_STAGE / 1024 / 8;          // 6 one-KiB DMA pieces per wave per K tile

template <bool SWIGLU, bool OUT_F32>
__global__ __launch_bounds__(512, 2) void gemm_mfma_fp8_wide_kernel(const unsigned char* __restrict__ A, const float* __restrict__ a_scale,
                                                                 const unsigned char* __restrict__ W, const float* __restrict__ w_scale,
                                                                 const bf16_t* res, void* Cv, int M, int N, int K, int lda, int ldc,
                                                                 int tiles_m, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 2, wn = wid & 3;
    const int tile = f8_xcd_remap(blockIdx.x, tiles_m * tiles_n);
    const int tm = tile % tiles_m, tn = tile / tiles_m;
    const int m0 = tm * F8W_BM, n0 = tn * F8W_BN;
    const int fr = lane & 15, fg = lane >> 4;
    const int nk = K / F8_BK;
    const unsigned char* src[F8W_PIECES];
#pragma unroll
    for (int j = 0; j < F8W_PIECES; ++j) {
        const int g = wid * F8W_PIECES + j;
        const int rl = lane >> 3, c = (lane & 7) ^ rl;
        if (g < 16) src[j] = A + (long long)min(m0 + g * 8 + rl, M - 1) * lda + c * 16;
        else src[j] = W + (long long)min(n0 + (g - 16) * 8 + rl, N - 1) * K + c * 16;
    }
#define TEO_F8W_STAGE(KT, ST)                                                                                               \
    _Pragma("unroll") for (int j = 0; j < F8W_PIECES; ++j)                                                                  \
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[j] + (long long)(KT) * F8_BK), \
                                         (__attribute__((address_space(3))) void*)(smem + (ST) * F8W_STAGE + (wid * F8W_PIECES + j) * 1024), 16, 0, 0);
#define TEO_F8W_FRAG(BASE, ROW)                                                              \
    ({                                                                                       \
        const unsigned char* rp_ = (BASE) + (ROW) * F8_BK;                                   \
        const f8_u32x4 lo_ = *reinterpret_cast<const f8_u32x4*>(rp_ + (((2 * fg) ^ ((ROW) & 7)) << 4));      \
        const f8_u32x4 hi_ = *reinterpret_cast<const f8_u32x4*>(rp_ + (((2 * fg + 1) ^ ((ROW) & 7)) << 4));  \
        i32x8 f_;                                                                            \
        f_[0] = (int)lo_[0]; f_[1] = (int)lo_[1]; f_[2] = (int)lo_[2]; f_[3] = (int)lo_[3];  \
        f_[4] = (int)hi_[0]; f_[5] = (int)hi_[1]; f_[6] = (int)hi_[2]; f_[7] = (int)hi_[3];  \
        f_;                                                                                  \
    })
    f8_f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f8_f32x4){0.f, 0.f, 0.f, 0.f};
    TEO_F8W_STAGE(0, 0)
    if (nk > 1) TEO_F8W_STAGE(1, 1)
    int st = 0;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const int st2 = st == 0 ? 2 : st - 1;
        const unsigned char* sA = smem + st * F8W_STAGE;
        const unsigned char* sB = sA + F8W_A_BYTES;
        // skewed halves (as gw_ktile in gemm_wide.hip): waves 4-7 issue their DMA pieces before their fragment reads, waves 0-3
        // between their two MFMA blocks -- one wave of a SIMD multiplies while its partner loads
        const bool late = wid < 4;
        if (!late && kt + 2 < nk) { TEO_F8W_STAGE(kt + 2, st2) }
        __builtin_amdgcn_sched_barrier(0);
        i32x8 af[4], wf[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            af[i] = TEO_F8W_FRAG(sA, wm * 64 + i * 16 + fr);
            wf[i] = TEO_F8W_FRAG(sB, wn * 64 + i * 16 + fr);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
                acc[ni][mi] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wf[ni], af[mi], acc[ni][mi], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
        __builtin_amdgcn_sched_barrier(0);
        if (late && kt + 2 < nk) { TEO_F8W_STAGE(kt + 2, st2) }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ni = 2; ni < 4; ++ni)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
                acc[ni][mi] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wf[ni], af[mi], acc[ni][mi], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
        st = st == 2 ? 0 : st + 1;
    }
#undef TEO_F8W_STAGE
#undef TEO_F8W_FRAG
    const int mw = m0 + wm * 64, nw = n0 + wn * 64;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        const int m = mw + mi * 16 + fr;
        if (m >= M) continue;
        const float sa = a_scale[m];
        if (SWIGLU) {
#pragma unroll
            for (int ni = 0; ni < 4; ni += 2) {
                const int ng = nw + ni * 16 + fg * 4;
                if (ng >= N) continue;
                const int oc = (nw >> 1) + (ni >> 1) * 16 + fg * 4;
                const float4 sg = *reinterpret_cast<const float4*>(w_scale + ng);
                const float4 su = *reinterpret_cast<const float4*>(w_scale + ng + 16);
                float o[4];
                o[0] = silu(acc[ni][mi][0] * (sa * sg.x)) * (acc[ni + 1][mi][0] * (sa * su.x));
                o[1] = silu(acc[ni][mi][1] * (sa * sg.y)) * (acc[ni + 1][mi][1] * (sa * su.y));
                o[2] = silu(acc[ni][mi][2] * (sa * sg.z)) * (acc[ni + 1][mi][2] * (sa * su.z));
                o[3] = silu(acc[ni][mi][3] * (sa * sg.w)) * (acc[ni + 1][mi][3] * (sa * su.w));
                if (OUT_F32) *reinterpret_cast<float4*>(reinterpret_cast<float*>(Cv) + (long long)m * ldc + oc) = make_float4(o[0], o[1], o[2], o[3]);
                else *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(Cv) + (long long)m * ldc + oc) = make_uint2(pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3]));
            }
        } else {
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                const int n = nw + ni * 16 + fg * 4;
                if (n >= N) continue;
                const float4 sw = *reinterpret_cast<const float4*>(w_scale + n);
                float o[4] = {acc[ni][mi][0] * (sa * sw.x), acc[ni][mi][1] * (sa * sw.y), acc[ni][mi][2] * (sa * sw.z), acc[ni][mi][3] * (sa * sw.w)};
                if (res) {
                    const uint2 q = *reinterpret_cast<const uint2*>(res + (long long)m * ldc + n);
                    o[0] += bf2f((bf16_t)(q.x & 0xffff)); o[1] += bf2f((bf16_t)(q.x >> 16));
                    o[2] += bf2f((bf16_t)(q.y & 0xffff)); o[3] += bf2f((bf16_t)(q.y >> 16));
                }
                if (OUT_F32) *reinterpret_cast<float4*>(reinterpret_cast<float*>(Cv) + (long long)m * ldc + n) = make_float4(o[0], o[1], o[2], o[3]);
                else *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(Cv) + (long long)m * ldc + n) = make_uint2(pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3]));
            }
        }
    }
}

// stream-K form of the wide fp8 kernel (one-round-plus shapes: o / down at M = 2168), the scheme of gemm_wide.hip
constexpr int F8W_SLAB_FLOATS = F8W_BM * F8W_BN;

template <bool OUT_F32>
__global__ __launch_bounds__(512, 2) void gemm_mfma_fp8_wide_sk_kernel(const unsigned char* __restrict__ A, const float* __restrict__ a_scale,
                                                                    const unsigned char* __restrict__ W, const float* __restrict__ w_scale,
                                                                    const bf16_t* res, void* Cv, int M, int N, int K, int lda, int ldc,
                                                                    int tiles_m, int tiles_n, int per, float* slabs, int* flags) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 2, wn = wid & 3;
    const int fr = lane & 15, fg = lane >> 4;
    const int nk = K / F8_BK;
    const long long total = (long long)tiles_m * tiles_n * nk;
    const int q = f8_xcd_remap(blockIdx.x, gridDim.x);
    const long long it0 = (long long)q * per, it1 = min(it0 + per, total);
    if (it0 >= total) return;
    const int t_first = (int)(it0 / nk), k_first = (int)(it0 % nk);
    const int t_last = (int)((it1 - 1) / nk), k_end = (int)(it1 - (long long)t_last * nk);
    const int has_head = k_first != 0, has_tail = k_end != nk;
    const int t_full0 = t_first + has_head, n_full = (t_last + 1 - has_tail) - t_full0;
    const int nseg = has_tail + n_full + has_head;
    f8_f32x4 acc[4][4];
    for (int sgi = 0; sgi < nseg; ++sgi) {
        const bool is_tail = has_tail && sgi == 0;
        const bool is_head = has_head && sgi == nseg - 1;
        const int pos = is_tail ? t_last : (is_head ? t_first : t_full0 + (sgi - has_tail));
        const int kb = is_head ? k_first : 0, ke = is_tail ? k_end : nk;
        const int tm = pos % tiles_m, tn = pos / tiles_m;
        const int m0 = tm * F8W_BM, n0 = tn * F8W_BN;
        const unsigned char* src[F8W_PIECES];
#pragma unroll
        for (int j = 0; j < F8W_PIECES; ++j) {
            const int g = wid * F8W_PIECES + j;
            const int rl = lane >> 3, c = (lane & 7) ^ rl;
            if (g < 16) src[j] = A + (long long)min(m0 + g * 8 + rl, M - 1) * lda + c * 16;
            else src[j] = W + (long long)min(n0 + (g - 16) * 8 + rl, N - 1) * K + c * 16;
        }
#define TEO_F8W_STAGE(KT, ST)                                                                                               \
    _Pragma("unroll") for (int j = 0; j < F8W_PIECES; ++j)                                                                  \
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[j] + (long long)(KT) * F8_BK), \
                                         (__attribute__((address_space(3))) void*)(smem + (ST) * F8W_STAGE + (wid * F8W_PIECES + j) * 1024), 16, 0, 0);
#define TEO_F8W_FRAG(BASE, ROW)                                                              \
    ({                                                                                       \
        const unsigned char* rp_ = (BASE) + (ROW) * F8_BK;                                   \
        const f8_u32x4 lo_ = *reinterpret_cast<const f8_u32x4*>(rp_ + (((2 * fg) ^ ((ROW) & 7)) << 4));      \
        const f8_u32x4 hi_ = *reinterpret_cast<const f8_u32x4*>(rp_ + (((2 * fg + 1) ^ ((ROW) & 7)) << 4));  \
        i32x8 f_;                                                                            \
        f_[0] = (int)lo_[0]; f_[1] = (int)lo_[1]; f_[2] = (int)lo_[2]; f_[3] = (int)lo_[3];  \
        f_[4] = (int)hi_[0]; f_[5] = (int)hi_[1]; f_[6] = (int)hi_[2]; f_[7] = (int)hi_[3];  \
        f_;                                                                                  \
    })
        if (is_head) {
            if (tid == 0) {
                int spins = 0;
                while (__hip_atomic_load(flags + q - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
                    __builtin_amdgcn_s_sleep(8);
                    if (++spins > (1 << 24)) { __hip_atomic_store(flags + GEMM_SK_ERR_SLOT, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }   // never reached (the producer wrote its slab first thing); a miss is STICKY: teo_gemm_workspace_status
                }
                __hip_atomic_store(flags + q - 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __builtin_amdgcn_s_barrier();
            const auto sl = __builtin_amdgcn_make_buffer_rsrc(slabs + (size_t)(q - 1) * F8W_SLAB_FLOATS, 0, F8W_SLAB_FLOATS * 4, 0x00020000);
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
                    acc[ni][mi] = __builtin_bit_cast(f8_f32x4, __builtin_amdgcn_raw_buffer_load_b128(sl, ((ni * 4 + mi) * 512 + tid) * 16, 0, 16));
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = (f8_f32x4){0.f, 0.f, 0.f, 0.f};
        }
        TEO_F8W_STAGE(kb, 0)
        if (kb + 1 < ke) TEO_F8W_STAGE(kb + 1, 1)
        int st = 0;
        for (int kt = kb; kt < ke; ++kt) {
            if (kt + 1 < ke) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const int st2 = st == 0 ? 2 : st - 1;
            const unsigned char* sA = smem + st * F8W_STAGE;
            const unsigned char* sB = sA + F8W_A_BYTES;
            const bool late = wid < 4;                                   // skewed halves, as in the plain wide kernel
            if (!late && kt + 2 < ke) { TEO_F8W_STAGE(kt + 2, st2) }
            __builtin_amdgcn_sched_barrier(0);
            i32x8 af[4], wf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                af[i] = TEO_F8W_FRAG(sA, wm * 64 + i * 16 + fr);
                wf[i] = TEO_F8W_FRAG(sB, wn * 64 + i * 16 + fr);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
                    acc[ni][mi] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wf[ni], af[mi], acc[ni][mi], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
            __builtin_amdgcn_sched_barrier(0);
            if (late && kt + 2 < ke) { TEO_F8W_STAGE(kt + 2, st2) }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ni = 2; ni < 4; ++ni)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
                    acc[ni][mi] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wf[ni], af[mi], acc[ni][mi], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
            st = st == 2 ? 0 : st + 1;
        }
#undef TEO_F8W_STAGE
#undef TEO_F8W_FRAG
        __builtin_amdgcn_s_barrier();
        if (is_tail) {
            const auto sl = __builtin_amdgcn_make_buffer_rsrc(slabs + (size_t)q * F8W_SLAB_FLOATS, 0, F8W_SLAB_FLOATS * 4, 0x00020000);
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(f8_u32x4, acc[ni][mi]), sl, ((ni * 4 + mi) * 512 + tid) * 16, 0, 16);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (tid == 0) __hip_atomic_store(flags + q, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            continue;
        }
        const int mw = m0 + wm * 64, nw = n0 + wn * 64;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            const int m = mw + mi * 16 + fr;
            if (m >= M) continue;
            const float sa = a_scale[m];
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                const int n = nw + ni * 16 + fg * 4;
                if (n >= N) continue;
                const float4 sw = *reinterpret_cast<const float4*>(w_scale + n);
                float o[4] = {acc[ni][mi][0] * (sa * sw.x), acc[ni][mi][1] * (sa * sw.y), acc[ni][mi][2] * (sa * sw.z), acc[ni][mi][3] * (sa * sw.w)};
                if (res) {
                    const uint2 qv = *reinterpret_cast<const uint2*>(res + (long long)m * ldc + n);
                    o[0] += bf2f((bf16_t)(qv.x & 0xffff)); o[1] += bf2f((bf16_t)(qv.x >> 16));
                    o[2] += bf2f((bf16_t)(qv.y & 0xffff)); o[3] += bf2f((bf16_t)(qv.y >> 16));
                }
                if (OUT_F32) *reinterpret_cast<float4*>(reinterpret_cast<float*>(Cv) + (long long)m * ldc + n) = make_float4(o[0], o[1], o[2], o[3]);
                else *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(Cv) + (long long)m * ldc + n) = make_uint2(pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3]));
            }
        }
    }
}


// ------------------------------------------------------------------------------------------------
// 256 x 256 form (the geometry of gemm_big.hip: 8 waves of 128 x 64, two-stage LDS-DMA ring of 2 x 64 KB, skewed DMA issue):
// 33 % fewer L2 -> LDS bytes and 25 % fewer fragment bytes per FLOP than the 128 x 256 kernel.  One K tile = 128 bytes = ONE scaled
// MFMA per fragment pair, so there is no second half to carry: 24 ds_read_b128 then 32 MFMAs per wave per K tile, the MFMAs
// ordered row-fragment-major so they start when the weight fragments and the first activation fragment have arrived.
// Bit-identical to the other fp8 kernels (same k order per output element).
// ------------------------------------------------------------------------------------------------
constexpr int F8B_BM = 256, F8B_BN = 256;
constexpr int F8B_A_BYTES = F8B_BM * F8_BK, F8B_STAGE = F8B_A_BYTES + F8B_BN * F8_BK;     // 32 KB + 32 KB

template <bool SWIGLU, bool OUT_F32>
__global__ __launch_bounds__(512) void gemm_mfma_fp8_big_kernel(const unsigned char* __restrict__ A, const float* __restrict__ a_scale,
                                                              const unsigned char* __restrict__ W, const float* __restrict__ w_scale,
                                                              const bf16_t* res, void* Cv, int M, int N, int K, int lda, int ldc,
                                                              int tiles_m, int tiles_n, int group) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 2, wn = wid & 3;
    const int fr = lane & 15, fg = lane >> 4;
    const int nk = K / F8_BK;
    const int tile = f8_xcd_remap(blockIdx.x, tiles_m * tiles_n);
    const int gsz = group * tiles_m, sup = tile / gsz, rem = tile - sup * gsz;       // grouped tile walk (gemm_big.hip)
    const int gn = min(group, tiles_n - sup * group);
    const int tm = rem / gn, tn = sup * group + rem % gn;
    const int m0 = tm * F8B_BM, n0 = tn * F8B_BN;
    // DMA: waves 0-3 bring A rows 64 w .. + 63 (8 pieces of 8 rows), waves 4-7 W rows; lane l: row l >> 3, logical chunk (l & 7) ^ (l >> 3)
    const bool isA = wid < 4;
    const unsigned char* base = isA ? A : W;
    const int ld = isA ? lda : K;
    const int row0 = isA ? m0 + wid * 64 : n0 + (wid - 4) * 64;
    const int rmax = (isA ? M : N) - 1;
    unsigned off[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int rl = lane >> 3, c = (lane & 7) ^ rl;
        off[j] = (unsigned)min(row0 + j * 8 + rl, rmax) * (unsigned)ld + c * 16;
    }
    const int lds_piece0 = (isA ? wid * 8 : 32 + (wid - 4) * 8) * 1024;
#define TEO_F8B_STAGE(KT, ST)                                                                                                \
    _Pragma("unroll") for (int j = 0; j < 8; ++j)                                                                            \
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + off[j] + (unsigned)(KT) * F8_BK), \
                                         (__attribute__((address_space(3))) void*)(smem + (ST) * F8B_STAGE + lds_piece0 + j * 1024), 16, 0, 0);
#define TEO_F8B_FRAG(BASE, ROW)                                                              \
    ({                                                                                       \
        const unsigned char* rp_ = (BASE) + (ROW) * F8_BK;                                   \
        const f8_u32x4 lo_ = *reinterpret_cast<const f8_u32x4*>(rp_ + (((2 * fg) ^ ((ROW) & 7)) << 4));      \
        const f8_u32x4 hi_ = *reinterpret_cast<const f8_u32x4*>(rp_ + (((2 * fg + 1) ^ ((ROW) & 7)) << 4));  \
        i32x8 f_;                                                                            \
        f_[0] = (int)lo_[0]; f_[1] = (int)lo_[1]; f_[2] = (int)lo_[2]; f_[3] = (int)lo_[3];  \
        f_[4] = (int)hi_[0]; f_[5] = (int)hi_[1]; f_[6] = (int)hi_[2]; f_[7] = (int)hi_[3];  \
        f_;                                                                                  \
    })
    f8_f32x4 acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = (f8_f32x4){0.f, 0.f, 0.f, 0.f};
    TEO_F8B_STAGE(0, 0)
    const bool late = wid < 4;
    for (int kt = 0; kt < nk; ++kt) {
        const int st = kt & 1;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (!late && kt + 1 < nk) { TEO_F8B_STAGE(kt + 1, st ^ 1) }
        __builtin_amdgcn_sched_barrier(0);
        const unsigned char* sA = smem + st * F8B_STAGE;
        const unsigned char* sB = sA + F8B_A_BYTES;
        i32x8 af[8], wf[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) wf[i] = TEO_F8B_FRAG(sB, wn * 64 + i * 16 + fr);
#pragma unroll
        for (int i = 0; i < 8; ++i) af[i] = TEO_F8B_FRAG(sA, wm * 128 + i * 16 + fr);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
                acc[ni][mi] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wf[ni], af[mi], acc[ni][mi], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
        __builtin_amdgcn_sched_barrier(0);
        if (late && kt + 1 < nk) { TEO_F8B_STAGE(kt + 1, st ^ 1) }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mi = 4; mi < 8; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
                acc[ni][mi] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wf[ni], af[mi], acc[ni][mi], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
        __builtin_amdgcn_sched_barrier(0);
    }
#undef TEO_F8B_STAGE
#undef TEO_F8B_FRAG
    const int mw = m0 + wm * 128, nw = n0 + wn * 64;
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) {
        const int m = mw + mi * 16 + fr;
        if (m >= M) continue;
        const float sa = a_scale[m];
        if (SWIGLU) {
#pragma unroll
            for (int ni = 0; ni < 4; ni += 2) {
                const int ng = nw + ni * 16 + fg * 4;
                if (ng >= N) continue;
                const int oc = (nw >> 1) + (ni >> 1) * 16 + fg * 4;
                const float4 sg = *reinterpret_cast<const float4*>(w_scale + ng);
                const float4 su = *reinterpret_cast<const float4*>(w_scale + ng + 16);
                float o[4];
                o[0] = silu(acc[ni][mi][0] * (sa * sg.x)) * (acc[ni + 1][mi][0] * (sa * su.x));
                o[1] = silu(acc[ni][mi][1] * (sa * sg.y)) * (acc[ni + 1][mi][1] * (sa * su.y));
                o[2] = silu(acc[ni][mi][2] * (sa * sg.z)) * (acc[ni + 1][mi][2] * (sa * su.z));
                o[3] = silu(acc[ni][mi][3] * (sa * sg.w)) * (acc[ni + 1][mi][3] * (sa * su.w));
                if (OUT_F32) *reinterpret_cast<float4*>(reinterpret_cast<float*>(Cv) + (long long)m * ldc + oc) = make_float4(o[0], o[1], o[2], o[3]);
                else *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(Cv) + (long long)m * ldc + oc) = make_uint2(pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3]));
            }
        } else {
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                const int n = nw + ni * 16 + fg * 4;
                if (n >= N) continue;
                const float4 sw = *reinterpret_cast<const float4*>(w_scale + n);
                float o[4] = {acc[ni][mi][0] * (sa * sw.x), acc[ni][mi][1] * (sa * sw.y), acc[ni][mi][2] * (sa * sw.z), acc[ni][mi][3] * (sa * sw.w)};
                if (res) {
                    const uint2 q = *reinterpret_cast<const uint2*>(res + (long long)m * ldc + n);
                    o[0] += bf2f((bf16_t)(q.x & 0xffff)); o[1] += bf2f((bf16_t)(q.x >> 16));
                    o[2] += bf2f((bf16_t)(q.y & 0xffff)); o[3] += bf2f((bf16_t)(q.y >> 16));
                }
                if (OUT_F32) *reinterpret_cast<float4*>(reinterpret_cast<float*>(Cv) + (long long)m * ldc + n) = make_float4(o[0], o[1], o[2], o[3]);
                else *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(Cv) + (long long)m * ldc + n) = make_uint2(pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3]));
            }
        }
    }
}

// tune().gemm_fp8_big (default 1): 256 x 256 kernel: 0 off, 1 auto (rounds model), 2 forced
constexpr double F8_BIG_ROUND_COST = 1.66;   // measured: 58.4 us per round of 256 x 256 tiles vs 35.3 us per round of 128 x 256 (gate/up at M = 17344)
// tune().gemm_fp8_wide (default 1): 0: 128 x 128 kernel only, 1: by the rounds model, 2: wide wherever K has two tiles

bool gemm_fp8_ok(int M, int N, int K, int lda, int ldc, unsigned flags, const void* A, const void* W, const void* res, const void* C) {
    if (M < 1 || N < 1 || K % F8_BK != 0 || lda % 16 != 0 || ldc % 4 != 0 || N % 4 != 0) return false;
    if ((flags & TEO_GEMM_SWIGLU16) && (N % 32 != 0 || res)) return false;
    auto al = [](const void* p, size_t a) { return p == nullptr || (reinterpret_cast<uintptr_t>(p) % a) == 0; };
    return al(A, 16) && al(W, 16) && al(res, 8) && al(C, 16);
}

int gemm_fp8(const void* A8, const float* a_scale, const void* W8, const float* w_scale, const void* res, void* C, int M, int N, int K,
             int lda, int ldc, unsigned flags, int out_dtype, hipStream_t st, void* sk_ws) {
    if (M == 0 || N == 0) return TEO_OK;
    if (sk_ws) {                                             // the stream-K form is sized for 256 CUs
        if (device_cu_count() != 256) sk_ws = nullptr;
    }
    if (!gemm_fp8_ok(M, N, K, lda, ldc, flags, A8, W8, res, C)) {
        set_error("teo_gemm_fp8: needs K %% 128 == 0, lda %% 16 == 0, N %% 4 == 0 (32 with SWIGLU16, no residual) and 16-byte aligned operands "
                  "(M %d N %d K %d lda %d ldc %d)", M, N, K, lda, ldc);
        return TEO_ERR_UNSUPPORTED;
    }
    const bool swiglu = flags & TEO_GEMM_SWIGLU16;
    const bool of32 = out_dtype == TEO_F32;
    {   // wide tiles when they need fewer (cost-weighted) rounds: same model as the bf16 kernel (gemm.hip gemm_wide_wins)
        const long long t_wide = (long long)cdiv(M, F8W_BM) * cdiv(N, F8W_BN), t_plain = (long long)cdiv(M, F8_BM) * cdiv(N, F8_BN);
        const long long rem = t_plain % 512;
        const double plain = (double)(t_plain / 512) + (rem == 0 ? 0.0 : (rem <= 256 ? 0.66 : 1.0));
        const double wide = (double)cdiv(t_wide, 256) * 0.80;   // measured: a wide fp8 round costs ~0.8 of a 128 x 128 round (o: 76 vs 81 us, gate/up 219 vs 272)
        const bool sk_shape = sk_ws && tune().gemm_fp8_wide && !swiglu && K >= 2 * F8_BK && t_wide > 256 && (tune().gemm_fp8_wide == 3 || t_wide <= 256 + 256 / 6);
        const long long t_big = (long long)cdiv(M, F8B_BM) * cdiv(N, F8B_BN);
        if (K >= 2 * F8_BK && (tune().gemm_fp8_big == 2 || (tune().gemm_fp8_big == 1 && tune().gemm_fp8_wide == 1 && t_big >= 160 && !sk_shape &&
                                                  cdiv(t_big, 256) * F8_BIG_ROUND_COST < (double)cdiv(t_wide, 256)))) {
            const int tiles_m = cdiv(M, F8B_BM), tiles_n = cdiv(N, F8B_BN);
            const size_t lds = 2 * F8B_STAGE;
            const int group = tiles_m >= 16 ? 4 : 1;
#define TEO_F8B_LAUNCH(SW, OF)                                                                                                  \
    {                                                                                                                           \
        static unsigned long long attr_mask = 0;                                                                                \
        if (int e = lds_attr_once(reinterpret_cast<const void*>(&gemm_mfma_fp8_big_kernel<SW, OF>), (int)lds, &attr_mask, "gemm_fp8 big")) return e; \
        gemm_mfma_fp8_big_kernel<SW, OF><<<tiles_m * tiles_n, 512, lds, st>>>((const unsigned char*)A8, a_scale, (const unsigned char*)W8, \
                                                                             w_scale, (const bf16_t*)res, C, M, N, K, lda, ldc, tiles_m, tiles_n, group); \
    }
            if (swiglu) { if (of32) TEO_F8B_LAUNCH(true, true) else TEO_F8B_LAUNCH(true, false) }
            else { if (of32) TEO_F8B_LAUNCH(false, true) else TEO_F8B_LAUNCH(false, false) }
#undef TEO_F8B_LAUNCH
            note_kernel("gemm_fp8_big"); TEO_LAUNCH_CHECK("gemm_mfma_fp8_big");
            return TEO_OK;
        }
        if (sk_shape) {
            // just over one round of wide tiles: persistent stream-K grid (slabs: 256 x 128 KB, flags behind GEMM_SK_SLAB_BYTES as in gemm.hip)
            const int tiles_m = cdiv(M, F8W_BM), tiles_n = cdiv(N, F8W_BN);
            const long long total = (long long)tiles_m * tiles_n * (K / F8_BK);
            const int per = (int)((total + 255) / 256);
            const size_t lds = 3 * F8W_STAGE;
            float* slabs = (float*)sk_ws;
            int* flg = (int*)((unsigned char*)sk_ws + GEMM_SK_SLAB_BYTES);
#define TEO_F8SK_LAUNCH(OF)                                                                                                     \
    {                                                                                                                           \
        static unsigned long long attr_mask = 0;                                                                                \
        if (int e = lds_attr_once(reinterpret_cast<const void*>(&gemm_mfma_fp8_wide_sk_kernel<OF>), (int)lds, &attr_mask, "gemm_fp8 wide sk")) return e; \
        gemm_mfma_fp8_wide_sk_kernel<OF><<<256, 512, lds, st>>>((const unsigned char*)A8, a_scale, (const unsigned char*)W8, w_scale, \
                                                               (const bf16_t*)res, C, M, N, K, lda, ldc, tiles_m, tiles_n, per, slabs, flg); \
    }
            if (of32) TEO_F8SK_LAUNCH(true) else TEO_F8SK_LAUNCH(false)
#undef TEO_F8SK_LAUNCH
            note_kernel("gemm_fp8_wide_sk"); TEO_LAUNCH_CHECK("gemm_mfma_fp8_wide_sk");
            return TEO_OK;
        }
        if (K >= 2 * F8_BK && (tune().gemm_fp8_wide >= 2 || (tune().gemm_fp8_wide == 1 && wide < plain && (t_wide >= 256 || (t_wide >= 144 && t_plain > 256))))) {
            const int tiles_m = cdiv(M, F8W_BM), tiles_n = cdiv(N, F8W_BN);
            const size_t lds = 3 * F8W_STAGE;
#define TEO_F8W_LAUNCH(SW, OF)                                                                                                  \
    {                                                                                                                           \
        static unsigned long long attr_mask = 0;                                                                                \
        if (int e = lds_attr_once(reinterpret_cast<const void*>(&gemm_mfma_fp8_wide_kernel<SW, OF>), (int)lds, &attr_mask, "gemm_fp8 wide")) return e; \
        gemm_mfma_fp8_wide_kernel<SW, OF><<<tiles_m * tiles_n, 512, lds, st>>>((const unsigned char*)A8, a_scale, (const unsigned char*)W8, \
                                                                              w_scale, (const bf16_t*)res, C, M, N, K, lda, ldc, tiles_m, tiles_n); \
    }
            if (swiglu) { if (of32) TEO_F8W_LAUNCH(true, true) else TEO_F8W_LAUNCH(true, false) }
            else { if (of32) TEO_F8W_LAUNCH(false, true) else TEO_F8W_LAUNCH(false, false) }
#undef TEO_F8W_LAUNCH
            note_kernel("gemm_fp8_wide"); TEO_LAUNCH_CHECK("gemm_mfma_fp8_wide");
            return TEO_OK;
        }
    }
    const int tiles_m = cdiv(M, F8_BM), tiles_n = cdiv(N, F8_BN);
    const int nwg = tiles_m * tiles_n;
    const size_t lds = 4 * F8_TILE;
#define TEO_F8_LAUNCH(SW, OF)                                                                                               \
    gemm_mfma_fp8_kernel<SW, OF><<<nwg, 256, lds, st>>>((const unsigned char*)A8, a_scale, (const unsigned char*)W8, w_scale,  \
                                                        (const bf16_t*)res, C, M, N, K, lda, ldc, tiles_m, tiles_n)
    if (swiglu) { if (of32) TEO_F8_LAUNCH(true, true); else TEO_F8_LAUNCH(true, false); }
    else { if (of32) TEO_F8_LAUNCH(false, true); else TEO_F8_LAUNCH(false, false); }
#undef TEO_F8_LAUNCH
    note_kernel("gemm_fp8_128"); TEO_LAUNCH_CHECK("gemm_mfma_fp8");
    return TEO_OK;
}

// ------------------------------------------------------------------------------------------------
// per-token activation quantisation: q[m, :] = e4m3(x[m, :] / s[m]), s[m] = max|x[m, :]| / 448 (1 for an all-zero row).
// NORM: x is first passed through LlamaRMSNorm exactly as the bf16 path does it (fp32 statistics, value rounded to bf16):
// the quantiser sees the tensor the bf16 GEMM would have consumed.  One workgroup per row, row kept in registers.
// ------------------------------------------------------------------------------------------------
template <bool NORM>
__global__ __launch_bounds__(256) void quant_rows_fp8_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ g, unsigned char* __restrict__ q,
                                                          float* __restrict__ s, int K, int ldx, float eps) {
    constexpr int MAXV = 6;                          // 8-element vectors per thread: K <= 256 * 8 * 6 = 12288
    __shared__ float red[4];
    const int m = blockIdx.x, tid = threadIdx.x;
    const bf16_t* xr = x + (long long)m * ldx;
    float v[MAXV][8];
    const int nv = K >> 3;                           // K % 16 == 0 (host)
    float ssq = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = tid + 256 * i;
        if (c < nv) {
            const uint4 r = *reinterpret_cast<const uint4*>(xr + c * 8);
            v[i][0] = __uint_as_float(r.x << 16); v[i][1] = __uint_as_float(r.x & 0xffff0000u);
            v[i][2] = __uint_as_float(r.y << 16); v[i][3] = __uint_as_float(r.y & 0xffff0000u);
            v[i][4] = __uint_as_float(r.z << 16); v[i][5] = __uint_as_float(r.z & 0xffff0000u);
            v[i][6] = __uint_as_float(r.w << 16); v[i][7] = __uint_as_float(r.w & 0xffff0000u);
#pragma unroll
            for (int e = 0; e < 8; ++e) ssq = fmaf(v[i][e], v[i][e], ssq);
        }
    }
    if (NORM) {
        const float tot = block_sum<256>(ssq, red);
        const float inv = rsqrtf(tot / (float)K + eps);
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c = tid + 256 * i;
            if (c < nv) {
                const uint4 r = *reinterpret_cast<const uint4*>(g + c * 8);
                const float gw[8] = {__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u), __uint_as_float(r.y << 16),
                                     __uint_as_float(r.y & 0xffff0000u), __uint_as_float(r.z << 16), __uint_as_float(r.z & 0xffff0000u),
                                     __uint_as_float(r.w << 16), __uint_as_float(r.w & 0xffff0000u)};
#pragma unroll
                for (int e = 0; e < 8; ++e) v[i][e] = bf2f(f2bf(v[i][e] * inv * gw[e]));   // x * r * w in fp32, one rounding to bf16: as norm.hip
            }
        }
    }
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = tid + 256 * i;
        if (c < nv) {
#pragma unroll
            for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf(v[i][e]));
        }
    }
    amax = wave_max(amax);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = amax;
    __syncthreads();
    amax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float scale = amax > 0.f ? amax * (1.0f / 448.0f) : 1.0f;
    const float inv_s = 1.0f / scale;
    if (tid == 0) s[m] = scale;
    unsigned char* qr = q + (long long)m * K;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = tid + 256 * i;
        if (c < nv) {
            int w0 = 0, w1 = 0;
            w0 = __builtin_amdgcn_cvt_pk_fp8_f32(v[i][0] * inv_s, v[i][1] * inv_s, w0, false);
            w0 = __builtin_amdgcn_cvt_pk_fp8_f32(v[i][2] * inv_s, v[i][3] * inv_s, w0, true);
            w1 = __builtin_amdgcn_cvt_pk_fp8_f32(v[i][4] * inv_s, v[i][5] * inv_s, w1, false);
            w1 = __builtin_amdgcn_cvt_pk_fp8_f32(v[i][6] * inv_s, v[i][7] * inv_s, w1, true);
            *reinterpret_cast<uint2*>(qr + c * 8) = make_uint2((unsigned)w0, (unsigned)w1);
        }
    }
}

int quant_rows_fp8(const void* x, const void* norm_w, void* q, float* s, int M, int K, int ldx, float eps, hipStream_t st) {
    if (M == 0) return TEO_OK;
    if (K % 16 != 0 || K > 256 * 8 * 6 || ldx % 8 != 0) {
        set_error("teo_quant_rows_fp8: K %d must be a multiple of 16 and <= 12288 (ldx %d a multiple of 8)", K, ldx);
        return TEO_ERR_UNSUPPORTED;
    }
    if (norm_w) quant_rows_fp8_kernel<true><<<M, 256, 0, st>>>((const bf16_t*)x, (const bf16_t*)norm_w, (unsigned char*)q, s, K, ldx, eps);
    else quant_rows_fp8_kernel<false><<<M, 256, 0, st>>>((const bf16_t*)x, nullptr, (unsigned char*)q, s, K, ldx, eps);
    TEO_LAUNCH_CHECK("quant_rows_fp8");
    return TEO_OK;
}

}  // namespace teo
