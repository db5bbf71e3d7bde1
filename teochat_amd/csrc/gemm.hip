// C[M,N] = act(A[M,K] . W[N,K]^T + bias) + residual        (nn.Linear layout: both operands K-contiguous)
//
//  gemm_simple_kernel<T,TO> : shape-agnostic 64x64x16 VALU tile kernel, fp32 accumulate.  It is the fp32
//                             parity path and the fallback for shapes the MFMA kernel does not take.
//  gemm_mfma_bf16_kernel    : bf16 in, fp32 accumulate on v_mfma_f32_16x16x32_bf16.  128x128x64 workgroup tile,
//                             4 waves (2x2) of 64x64, LDS double buffer (64 KiB), register-staged global loads
//                             issued one K-tile ahead, 16-byte XOR-swizzled LDS rows (ds_read_b128 <= 2-way),
//                             XCD-aware tile order (m fastest so one XCD's L2 keeps a W panel).
//                             Operands are swapped (D^T = W . A^T) so a lane owns 4 consecutive N of one M row:
//                             8-byte stores, vector bias/residual loads, and the gate/up pair of SwiGLU sits in
//                             the same lane (W rows interleaved in blocks of 16 at load time).
//
// Roofline: MFMA bf16 dense (2.5 PFLOP/s); algorithmic FLOPs = 2*M*N*K.
#include "common.h"
#include <algorithm>

#include "ops.h"

namespace teo {

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;


// ------------------------------------------------------------------------------------------------
// generic kernel
// ------------------------------------------------------------------------------------------------
template <typename T, typename TO, bool SWIGLU>
__global__ __launch_bounds__(256) void gemm_simple_kernel(const T* __restrict__ A, const T* __restrict__ W,
                                                          const T* __restrict__ bias, const T* res,
                                                          TO* __restrict__ C, int M, int N, int K, int lda, int ldc,
                                                          int act) {
    __shared__ float As[16][65];
    __shared__ float Ws[16][65];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;

    for (int k0 = 0; k0 < K; k0 += 16) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int id = threadIdx.x + i * 256;
            const int r = id >> 4, kk = id & 15;
            const bool kin = (k0 + kk) < K;
            As[kk][r] = (kin && (m0 + r) < M) ? Elem<T>::ld(A + (long long)(m0 + r) * lda + k0 + kk) : 0.f;
            Ws[kk][r] = (kin && (n0 + r) < N) ? Elem<T>::ld(W + (long long)(n0 + r) * K + k0 + kk) : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            float a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = As[kk][ty + 16 * i];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = Ws[kk][tx + 16 * j];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + ty + 16 * i;
        if (m >= M) continue;
        if (SWIGLU) {
            // columns tx+16j: (j=0,1) and (j=2,3) are (gate, up) pairs of the interleaved-16 layout
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int ng = n0 + tx + 32 * p;       // gate column
                if (ng < N) {   // N % 32 == 0, so the whole (gate, up) block is in range
                    const float g = acc[i][2 * p], u = acc[i][2 * p + 1];
                    const int oc = (n0 >> 1) + 16 * p + tx;
                    Elem<TO>::st(C + (long long)m * ldc + oc, silu(g) * u);
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = n0 + tx + 16 * j;
                if (n >= N) continue;
                float v = acc[i][j];
                if (bias) v += Elem<T>::ld(bias + n);
                v = act_apply(v, act);
                if (res) v += Elem<T>::ld(res + (long long)m * ldc + n);
                Elem<TO>::st(C + (long long)m * ldc + n, v);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// MFMA kernel
// ------------------------------------------------------------------------------------------------
constexpr int BM = 128, BN = 128, BK = 64;
constexpr double GEMM_WIDE_ROUND_COST = 0.88;  // one round of 256 wide tiles (one per CU) / one round of 512 128 x 128 tiles (two per CU):
                                               // 62-69 us against 71-83 us at K = 4096 (tools/bench_kernels.py gemm_wide)
constexpr int TILE_BYTES = BM * BK * 2;   // 16 KiB per operand tile

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    // blocks are dispatched round-robin over the 8 XCDs; give each XCD a contiguous range of tiles (bijective form)
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    const int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    return base + (bid >> 3);
}

// MF = 16-row fragments per wave along M: workgroup tile = (32*MF) x 128 (MF = 4: 128 x 128; MF = 2: 64 x 128 for shapes
// whose 128-row tiles quantise badly over the 512 resident workgroup slots, e.g. M = 2168, N = 4096 -> 544 tiles).
template <bool SWIGLU, bool OUT_F32, int DEPTH, int MF, bool F16 = false>   // F16: IEEE binary16 operands / outputs instead of bfloat16 (common.h)
__global__ __launch_bounds__(256, 2) void gemm_mfma_bf16_kernel(const bf16_t* __restrict__ A,
                                                             const bf16_t* __restrict__ W,
                                                             const bf16_t* __restrict__ bias,
                                                             const bf16_t* res, void* Cv,
                                                             int M, int N, int K, int lda, int ldc, int act,
                                                             int tiles_m, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int tile = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    // an XCD owns a contiguous run of the tile sequence, m fastest: the run shares a few W panels (its L2 keeps them).  (n fastest --
    // sharing A row blocks and streaming W -- was measured for the shapes where W is the smaller operand, ViT out_proj / fc2: 1-5 %
    // slower, tools/vit_gemm_probe.py round 4; not kept.)
    const int tm = tile % tiles_m, tn = tile / tiles_m;
    constexpr int BMT = 32 * MF;
    const int m0 = tm * BMT, n0 = tn * BN;

    // staging: thread owns 4 chunks (16 B) of each operand tile: chunk id = tid + 256*i -> row id>>3, chunk id&7.
    // NOTE: plain arrays + fully unrolled loops only -- lambdas capturing these arrays made hipcc spill them to scratch.
    // 32-bit element offsets from the (wave-uniform) base pointers: half the address registers of per-thread pointers, which is
    // what lets the SwiGLU variant keep two register stages without spilling (host checks M * lda and N * K < 2^31)
    unsigned ag[MF];                           // the A tile has 32*MF rows: MF chunks per thread
    unsigned wg[4];
    int soff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int id = tid + 256 * i;
        const int row = id >> 3, c = id & 7;
        const int gn = min(n0 + row, N - 1);
        if (i < MF) ag[i] = (unsigned)min(m0 + row, M - 1) * (unsigned)lda + c * 8;
        wg[i] = (unsigned)gn * (unsigned)K + c * 8;
        soff[i] = row * (BK * 2) + ((c ^ (row & 7)) << 4);
    }
    u32x4 ra0[MF], rb0[4], ra1[MF], rb1[4];   // two register stages: tiles kt+1 and kt+2 are in flight during compute(kt)

#define TEO_GLOAD(RA, RB, KT)                                                   \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                             \
        if (i < MF) RA[i] = *reinterpret_cast<const u32x4*>(A + (ag[i] + (unsigned)(KT) * BK));  \
        RB[i] = *reinterpret_cast<const u32x4*>(W + (wg[i] + (unsigned)(KT) * BK));  \
    }
#define TEO_SWRITE(RA, RB, BUF)                                                 \
    {                                                                           \
        unsigned char* sa_ = smem + (BUF) * (2 * TILE_BYTES);                   \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                         \
            if (i < MF) *reinterpret_cast<u32x4*>(sa_ + soff[i]) = RA[i];       \
            *reinterpret_cast<u32x4*>(sa_ + TILE_BYTES + soff[i]) = RB[i];      \
        }                                                                       \
    }

    f32x4 acc[4][MF];   // [ni][mi]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < MF; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // fragment read offsets: row = base + i*16 + (lane&15); logical chunk = ks*4 + (lane>>4)
    const int fr = lane & 15, fg = lane >> 4;
    const int nk = K / BK;

#define TEO_COMPUTE(BUF)                                                                                             \
    {                                                                                                                \
        const unsigned char* sA = smem + (BUF) * (2 * TILE_BYTES);                                                   \
        const unsigned char* sB = sA + TILE_BYTES;                                                                   \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                                           \
            bf16x8 af[MF], wf[4];                                                                                    \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                          \
                if (i < MF) {                                                                                        \
                    const int ra_ = wm * (16 * MF) + i * 16 + fr;                                                    \
                    af[i] = *reinterpret_cast<const bf16x8*>(sA + ra_ * (BK * 2) + (((ks * 4 + fg) ^ (ra_ & 7)) << 4)); \
                }                                                                                                    \
                const int rw_ = wn * 64 + i * 16 + fr;                                                               \
                wf[i] = *reinterpret_cast<const bf16x8*>(sB + rw_ * (BK * 2) + (((ks * 4 + fg) ^ (rw_ & 7)) << 4));  \
            }                                                                                                        \
            _Pragma("unroll") for (int ni = 0; ni < 4; ++ni)                                                         \
                _Pragma("unroll") for (int mi = 0; mi < MF; ++mi)                                                    \
                    acc[ni][mi] = mfma16<F16>(wf[ni], af[mi], acc[ni][mi]);     \
        }                                                                                                            \
    }

    if (DEPTH == 2) {
        TEO_GLOAD(ra0, rb0, 0);
        if (nk > 1) TEO_GLOAD(ra1, rb1, 1);
        TEO_SWRITE(ra0, rb0, 0);
        __syncthreads();
        for (int kt = 0; kt < nk; kt += 2) {
            // even step: tile kt in LDS buffer 0, tile kt+1 in flight in set 1
            if (kt + 2 < nk) TEO_GLOAD(ra0, rb0, kt + 2);
            TEO_COMPUTE(0);
            if (kt + 1 < nk) TEO_SWRITE(ra1, rb1, 1);
            __syncthreads();
            if (kt + 1 >= nk) break;
            // odd step: tile kt+1 in LDS buffer 1, tile kt+2 in flight in set 0
            if (kt + 3 < nk) TEO_GLOAD(ra1, rb1, kt + 3);
            TEO_COMPUTE(1);
            if (kt + 2 < nk) TEO_SWRITE(ra0, rb0, 0);
            __syncthreads();
        }
    } else {
        TEO_GLOAD(ra0, rb0, 0);
        TEO_SWRITE(ra0, rb0, 0);
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + 1 < nk) TEO_GLOAD(ra0, rb0, kt + 1);
            if (kt & 1) { TEO_COMPUTE(1); } else { TEO_COMPUTE(0); }
            if (kt + 1 < nk) {
                if (kt & 1) { TEO_SWRITE(ra0, rb0, 0); } else { TEO_SWRITE(ra0, rb0, 1); }
            }
            __syncthreads();
        }
    }
#undef TEO_GLOAD
#undef TEO_SWRITE
#undef TEO_COMPUTE

    // epilogue: lane holds C[m = mw + mi*16 + fr][n = nw + ni*16 + fg*4 + r], r = 0..3
    // (the tile origin goes through an empty asm so that no epilogue address can be computed -- and kept live -- ahead of the
    // K loop: that hoisting cost the SwiGLU variant its second register stage)
    int mw = m0 + wm * (16 * MF), nw = n0 + wn * 64;
    asm volatile("" : "+v"(mw), "+v"(nw));
#pragma unroll
    for (int mi = 0; mi < MF; ++mi) {
        const int m = mw + mi * 16 + fr;
        if (m >= M) continue;
        if (SWIGLU) {
#pragma unroll
            for (int ni = 0; ni < 4; ni += 2) {
                const int ng = nw + ni * 16 + fg * 4;              // gate rows; up rows are +16
                if (ng >= N) continue;                               // N % 32 == 0: whole block in or out
                const int oc = (nw >> 1) + (ni >> 1) * 16 + fg * 4;
                float o[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = silu(acc[ni][mi][r]) * acc[ni + 1][mi][r];
                if (OUT_F32) {
                    *reinterpret_cast<float4*>(reinterpret_cast<float*>(Cv) + (long long)m * ldc + oc) =
                        make_float4(o[0], o[1], o[2], o[3]);
                } else {
                    *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(Cv) + (long long)m * ldc + oc) =
                        make_uint2(pack_h2<F16>(o[0], o[1]), pack_h2<F16>(o[2], o[3]));
                }
                // keep the 16 (gate, up) blocks in program order: hoisting all 64 expf expansions at once is what pushed this
                // variant to 226 VGPRs (depth 1) / spills (depth 2)
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                const int n = nw + ni * 16 + fg * 4;
                if (n >= N) continue;
                float o[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = acc[ni][mi][r];
                if (n + 3 < N) {
                    if (bias) {
                        const uint2 b = *reinterpret_cast<const uint2*>(bias + n);
                        o[0] += h_lo<F16>(b.x); o[1] += h_hi<F16>(b.x);
                        o[2] += h_lo<F16>(b.y); o[3] += h_hi<F16>(b.y);
                    }
                    if (act != TEO_ACT_NONE) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) o[r] = act_apply(o[r], act);
                    }
                    if (res) {
                        const uint2 q = *reinterpret_cast<const uint2*>(res + (long long)m * ldc + n);
                        o[0] += h_lo<F16>(q.x); o[1] += h_hi<F16>(q.x);
                        o[2] += h_lo<F16>(q.y); o[3] += h_hi<F16>(q.y);
                    }
                    if (OUT_F32) {
                        *reinterpret_cast<float4*>(reinterpret_cast<float*>(Cv) + (long long)m * ldc + n) =
                            make_float4(o[0], o[1], o[2], o[3]);
                    } else {
                        *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(Cv) + (long long)m * ldc + n) =
                            make_uint2(pack_h2<F16>(o[0], o[1]), pack_h2<F16>(o[2], o[3]));
                    }
                } else {
                    for (int r = 0; r < 4 && n + r < N; ++r) {
                        float v = o[r];
                        if (bias) v += h2f<F16>(bias[n + r]);
                        v = act_apply(v, act);
                        if (res) v += h2f<F16>(res[(long long)m * ldc + n + r]);
                        if (OUT_F32) reinterpret_cast<float*>(Cv)[(long long)m * ldc + n + r] = v;
                        else reinterpret_cast<bf16_t*>(Cv)[(long long)m * ldc + n + r] = f2h<F16>(v);
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// stream-K form of the same kernel (128 x 128 x 64 tiles, same k-order, same MFMA chains)
//
// Static tilings quantise badly on 256 CUs x 2 resident workgroups: at M = 2168 the N = 4096 layers (o, down) have 544
// tiles for 512 slots -- two rounds, the second 6 % full -- and qkv 1632 tiles = 3.19 rounds.  Here the grid is PERSISTENT
// (one workgroup per slot) and the flattened (tile, k-tile) iteration space is cut into equal contiguous ranges, so every
// workgroup does the same amount of MFMA work whatever M and N are.  A range that starts or ends inside a tile shares that
// tile with its neighbour:
//     tail  (range ends inside tile t): k-tiles [0, e) of t are accumulated FIRST in the workgroup's timeline, the fp32
//           accumulators go to this workgroup's slab in the workspace, a flag is raised;
//     full tiles: as in the plain kernel;
//     head  (range starts inside tile t): done LAST: the neighbour's slab (raised long ago: it was the first thing that
//           workgroup did) is loaded AS THE INITIAL ACCUMULATOR and k-tiles [b, nk) are accumulated on top of it.
// The k-order of every output element is therefore exactly the sequential one: results are BIT-IDENTICAL to the plain kernel
// (tests/test_kernels_gpu.py), and independent of M, N and the grid size -- the bitwise invariances the model tests rely on
// (frame-sharded tower == unsharded, batched prefill == per-slot prefill, truncation == prefix) still hold.
// Hand-off across workgroups (other XCDs, non-coherent L2s): slab and flag are written with relaxed agent-scope atomic
// stores (write-through, `sc1`), every storing wave drains `vmcnt(0)`, barrier, then the flag; the reader polls the flag
// relaxed and reads the slab with agent-scope atomic loads (`sc1`: past L1, coherent with the write-through stores).  No
// fences.  A waiter always waits on a workgroup that produced its slab at the very start of its life.
// ------------------------------------------------------------------------------------------------
template <bool SWIGLU, bool OUT_F32, bool F16 = false>
__device__ __forceinline__ void gemm_tile_epilogue(f32x4 (&acc)[4][4], const bf16_t* __restrict__ bias, const bf16_t* res, void* Cv,
                                                   int M, int N, int ldc, int act, int m0, int n0, int wm, int wn, int fr, int fg) {
    const int mw = m0 + wm * 64, nw = n0 + wn * 64;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        const int m = mw + mi * 16 + fr;
        if (m >= M) continue;
        if (SWIGLU) {
#pragma unroll
            for (int ni = 0; ni < 4; ni += 2) {
                const int ng = nw + ni * 16 + fg * 4;
                if (ng >= N) continue;
                const int oc = (nw >> 1) + (ni >> 1) * 16 + fg * 4;
                float o[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = silu(acc[ni][mi][r]) * acc[ni + 1][mi][r];
                if (OUT_F32) {
                    *reinterpret_cast<float4*>(reinterpret_cast<float*>(Cv) + (long long)m * ldc + oc) = make_float4(o[0], o[1], o[2], o[3]);
                } else {
                    *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(Cv) + (long long)m * ldc + oc) =
                        make_uint2(pack_h2<F16>(o[0], o[1]), pack_h2<F16>(o[2], o[3]));
                }
            }
        } else {
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                const int n = nw + ni * 16 + fg * 4;
                if (n >= N) continue;
                float o[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = acc[ni][mi][r];
                if (n + 3 < N) {
                    if (bias) {
                        const uint2 b = *reinterpret_cast<const uint2*>(bias + n);
                        o[0] += h_lo<F16>(b.x); o[1] += h_hi<F16>(b.x);
                        o[2] += h_lo<F16>(b.y); o[3] += h_hi<F16>(b.y);
                    }
                    if (act != TEO_ACT_NONE) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) o[r] = act_apply(o[r], act);
                    }
                    if (res) {
                        const uint2 q = *reinterpret_cast<const uint2*>(res + (long long)m * ldc + n);
                        o[0] += h_lo<F16>(q.x); o[1] += h_hi<F16>(q.x);
                        o[2] += h_lo<F16>(q.y); o[3] += h_hi<F16>(q.y);
                    }
                    if (OUT_F32) {
                        *reinterpret_cast<float4*>(reinterpret_cast<float*>(Cv) + (long long)m * ldc + n) = make_float4(o[0], o[1], o[2], o[3]);
                    } else {
                        *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(Cv) + (long long)m * ldc + n) =
                            make_uint2(pack_h2<F16>(o[0], o[1]), pack_h2<F16>(o[2], o[3]));
                    }
                } else {
                    for (int r = 0; r < 4 && n + r < N; ++r) {
                        float v = o[r];
                        if (bias) v += h2f<F16>(bias[n + r]);
                        v = act_apply(v, act);
                        if (res) v += h2f<F16>(res[(long long)m * ldc + n + r]);
                        if (OUT_F32) reinterpret_cast<float*>(Cv)[(long long)m * ldc + n + r] = v;
                        else reinterpret_cast<bf16_t*>(Cv)[(long long)m * ldc + n + r] = f2h<F16>(v);
                    }
                }
            }
        }
    }
}

constexpr int SK_SLAB_FLOATS = BM * BN;          // one 128 x 128 fp32 accumulator tile per workgroup

template <bool SWIGLU, bool OUT_F32, bool F16 = false>
__global__ __launch_bounds__(256, 2) void gemm_mfma_bf16_sk_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W,
                                                                const bf16_t* __restrict__ bias, const bf16_t* res, void* Cv,
                                                                int M, int N, int K, int lda, int ldc, int act, int tiles_m,
                                                                int tiles_n, int per, float* slabs, int* flags) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int fr = lane & 15, fg = lane >> 4;
    const int nk = K / BK;
    const long long total = (long long)tiles_m * tiles_n * nk;
    const int q = xcd_remap(blockIdx.x, gridDim.x);                 // logical position: one XCD owns a contiguous run of ranges
    const long long it0 = (long long)q * per, it1 = min(it0 + per, total);
    if (it0 >= total) return;

    const bf16_t* ag[4];
    const bf16_t* wg[4];
    int soff[4];
    f32x4 acc[4][4];
    u32x4 ra0[4], rb0[4];

    // one segment: k-tiles [kb, ke) of tile `tile`, accumulated into acc (the LDS double buffer is free on entry and on exit)
    // position s of the flattened tile sequence -> tile: the sequence is laid out in P = ceil(per / nk) interleaved runs, so
    // that the workgroups of an XCD, which sit ~per/nk positions apart, work on NEIGHBOURING tiles at any moment (a few W
    // panels in that XCD's L2 instead of a dozen)
    const int ntiles_ = tiles_m * tiles_n;
    const int P_ = (per + nk - 1) / nk, pfull_ = ntiles_ / P_, prem_ = ntiles_ % P_;
#define TEO_SK_SETUP(POS)                                                                                         \
    const int pr_ = (POS) % P_, pc_ = (POS) / P_;                                                                 \
    const int tile_ = pr_ * pfull_ + min(pr_, prem_) + pc_;                                                       \
    const int tm_ = tile_ % tiles_m, tn_ = tile_ / tiles_m;                                                       \
    const int m0 = tm_ * BM, n0 = tn_ * BN;                                                                       \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                               \
        const int id = tid + 256 * i;                                                                             \
        const int row = id >> 3, c = id & 7;                                                                      \
        ag[i] = A + (long long)min(m0 + row, M - 1) * lda + c * 8;                                                \
        wg[i] = W + (long long)min(n0 + row, N - 1) * K + c * 8;                                                  \
        soff[i] = row * (BK * 2) + ((c ^ (row & 7)) << 4);                                                        \
    }
#define TEO_SK_GLOAD(KT)                                                                                          \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                               \
        ra0[i] = *reinterpret_cast<const u32x4*>(ag[i] + (long long)(KT) * BK);                                   \
        rb0[i] = *reinterpret_cast<const u32x4*>(wg[i] + (long long)(KT) * BK);                                   \
    }
#define TEO_SK_SWRITE(BUF)                                                                                        \
    {                                                                                                             \
        unsigned char* sa_ = smem + (BUF) * (2 * TILE_BYTES);                                                     \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                           \
            *reinterpret_cast<u32x4*>(sa_ + soff[i]) = ra0[i];                                                    \
            *reinterpret_cast<u32x4*>(sa_ + TILE_BYTES + soff[i]) = rb0[i];                                       \
        }                                                                                                         \
    }
#define TEO_SK_COMPUTE(BUF)                                                                                       \
    {                                                                                                             \
        const unsigned char* sA = smem + (BUF) * (2 * TILE_BYTES);                                                \
        const unsigned char* sB = sA + TILE_BYTES;                                                                \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                                        \
            bf16x8 af[4], wf[4];                                                                                  \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                       \
                const int ra_ = wm * 64 + i * 16 + fr;                                                            \
                af[i] = *reinterpret_cast<const bf16x8*>(sA + ra_ * (BK * 2) + (((ks * 4 + fg) ^ (ra_ & 7)) << 4)); \
                const int rw_ = wn * 64 + i * 16 + fr;                                                            \
                wf[i] = *reinterpret_cast<const bf16x8*>(sB + rw_ * (BK * 2) + (((ks * 4 + fg) ^ (rw_ & 7)) << 4)); \
            }                                                                                                     \
            _Pragma("unroll") for (int ni = 0; ni < 4; ++ni)                                                      \
                _Pragma("unroll") for (int mi = 0; mi < 4; ++mi)                                                  \
                    acc[ni][mi] = mfma16<F16>(wf[ni], af[mi], acc[ni][mi]);  \
        }                                                                                                         \
    }
#define TEO_SK_KLOOP(KB, KE)                                                                                      \
    {                                                                                                             \
        TEO_SK_GLOAD(KB);                                                                                         \
        TEO_SK_SWRITE(0);                                                                                         \
        __syncthreads();                                                                                          \
        for (int kt = (KB); kt < (KE); ++kt) {                                                                    \
            const int par = (kt - (KB)) & 1;                                                                      \
            if (kt + 1 < (KE)) TEO_SK_GLOAD(kt + 1);                                                              \
            if (par) { TEO_SK_COMPUTE(1); } else { TEO_SK_COMPUTE(0); }                                           \
            if (kt + 1 < (KE)) {                                                                                  \
                if (par) { TEO_SK_SWRITE(0); } else { TEO_SK_SWRITE(1); }                                         \
            }                                                                                                     \
            __syncthreads();                                                                                      \
        }                                                                                                         \
    }
#define TEO_SK_ZERO()                                                                                             \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                 \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int t_first = (int)(it0 / nk), k_first = (int)(it0 % nk);           // head: tile t_first from k-tile k_first (if > 0)
    const int t_last = (int)((it1 - 1) / nk), k_end = (int)(it1 - (long long)t_last * nk);   // tail: tile t_last up to k_end (if < nk)
    const bool has_head = k_first != 0;
    // per >= nk (host): a range that starts inside a tile always reaches that tile's end, so a tile has at most two owners
    // slabs go through buffer descriptors: 16-byte stores / loads with the sc1 bit (write-through / past L1); 8-byte agent
    // atomics would be one fabric transaction each (measured: the hand-off then costs ~40 us per workgroup)
    const auto my_slab = __builtin_amdgcn_make_buffer_rsrc(slabs + (size_t)q * SK_SLAB_FLOATS, 0, SK_SLAB_FLOATS * 4, 0x00020000);

    // ---- (1) tail first: its partial sums are what the next range's owner continues from
    if (k_end != nk) {
        TEO_SK_SETUP(t_last)
        TEO_SK_ZERO()
        TEO_SK_KLOOP(0, k_end)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[ni][mi]), my_slab, ((ni * 4 + mi) * 256 + tid) * 16, 0,
                                                       /*aux: sc1*/ 16);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                    // every storing wave: write-through stores landed
        __syncthreads();
        if (tid == 0) __hip_atomic_store(flags + q, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // ---- (2) the tiles that lie entirely inside the range
    {
        const int tb = has_head ? t_first + 1 : t_first;
        const int te = (k_end != nk) ? t_last : t_last + 1;
        for (int t = tb; t < te; ++t) {
            TEO_SK_SETUP(t)
            TEO_SK_ZERO()
            TEO_SK_KLOOP(0, nk)
            gemm_tile_epilogue<SWIGLU, OUT_F32, F16>(acc, bias, res, Cv, M, N, ldc, act, m0, n0, wm, wn, fr, fg);
        }
    }
    // ---- (3) head last: continue the previous range's partial sums in the same k-order
    if (has_head) {
        if (tid == 0) {
            int spins = 0;
            while (__hip_atomic_load(flags + q - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
                __builtin_amdgcn_s_sleep(8);
                if (++spins > (1 << 24)) { __hip_atomic_store(flags + GEMM_SK_ERR_SLOT, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }   // never reached (the producer wrote its slab first thing); a miss is STICKY: teo_gemm_workspace_status
            }
            __hip_atomic_store(flags + q - 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // re-armed for the next launch
        }
        __syncthreads();
        const auto src = __builtin_amdgcn_make_buffer_rsrc(slabs + (size_t)(q - 1) * SK_SLAB_FLOATS, 0, SK_SLAB_FLOATS * 4, 0x00020000);
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
                acc[ni][mi] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(src, ((ni * 4 + mi) * 256 + tid) * 16, 0, /*aux: sc1*/ 16));
        TEO_SK_SETUP(t_first)
        TEO_SK_KLOOP(k_first, nk)
        gemm_tile_epilogue<SWIGLU, OUT_F32, F16>(acc, bias, res, Cv, M, N, ldc, act, m0, n0, wm, wn, fr, fg);
    }
#undef TEO_SK_SETUP
#undef TEO_SK_GLOAD
#undef TEO_SK_SWRITE
#undef TEO_SK_COMPUTE
#undef TEO_SK_KLOOP
#undef TEO_SK_ZERO
}

// ------------------------------------------------------------------------------------------------
// host dispatch
// ------------------------------------------------------------------------------------------------
constexpr double GEMM_BIG_ROUND_COST = 1.75;  // measured: 100 us per round of 256 x 256 tiles vs 58 us per round of 128 x 256 (gate/up at M = 17344; 8192^3: 195 vs 111)
// tune().gemm_big (default 1): 256 x 256 LDS-DMA kernel (gemm_big.hip): 0 off, 1 auto (rounds model), 2 forced
// tune().gemm_wide (default 1): wide-tile LDS-DMA kernel (gemm_wide.hip): 0 off, 1 auto, 2 forced wherever its shape constraints hold
// tune().gemm_sk (default 1): 1: stream-K kernel when a workspace is given and the static tiling would leave a ragged last round
constexpr int SK_MAX_GRID = 512;   // 256 CUs x 2 resident workgroups (64 KiB LDS, <= 256 VGPRs each)
// slabs of every stream-K form share the first GEMM_SK_SLAB_BYTES (512 x 64 KB here, 256 x 128 KB in gemm_wide.hip / gemm_fp8.hip,
// 256 x 256 KB in gemm_big.hip); the hand-off flags live behind them
size_t gemm_sk_workspace_bytes() { return GEMM_SK_SLAB_BYTES + GEMM_SK_FLAG_INTS * sizeof(int); }
// flags live behind the slabs; they must be zero before the first stream-K launch on a workspace (the kernel re-arms them)
int gemm_sk_workspace_init(void* ws, hipStream_t st) {
    hipError_t e = hipMemsetAsync((unsigned char*)ws + GEMM_SK_SLAB_BYTES, 0, GEMM_SK_FLAG_INTS * sizeof(int), st);
    return e == hipSuccess ? TEO_OK : hip_fail(e, "gemm_sk_workspace_init");
}
// the sticky error word of a workspace (0 = every hand-off since teo_gemm_workspace_init arrived); synchronises the stream
int gemm_sk_workspace_status(const void* ws, int* host_flag, hipStream_t st) {
    hipError_t e = hipMemcpyAsync(host_flag, (const unsigned char*)ws + GEMM_SK_SLAB_BYTES + GEMM_SK_ERR_SLOT * sizeof(int), sizeof(int),
                                  hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    return e == hipSuccess ? TEO_OK : hip_fail(e, "gemm_sk_workspace_status");
}
// the persistent forms assume the MI355X's 256 CUs (grids of 256 / 512 resident workgroups): elsewhere the plain kernels run
static bool sk_grid_fits_device() {
    return device_cu_count() == 256;
}
// tune().gemm_depth (default 0): 0 = auto: 2-deep register prefetch, 1-deep for the SwiGLU epilogue (register budget)
// tune().gemm_bm (default 0): 0 = auto (by wave quantisation over the resident workgroup slots), 64 or 128

bool gemm_mfma_ok(int M, int N, int K, int lda, int ldc, int dtype, unsigned flags, const void* A, const void* W,
                  const void* bias, const void* res, const void* C) {
    if ((dtype != TEO_BF16 && dtype != TEO_F16) || (flags & TEO_GEMM_FORCE_SIMPLE)) return false;
    if (K % BK != 0 || lda % 8 != 0 || ldc % 4 != 0 || N % 4 != 0) return false;
    if ((long long)M * lda >= (1ll << 31) || (long long)N * K >= (1ll << 31)) return false;      // 32-bit element offsets in the kernel
    if (M < 1 || N < 1) return false;
    if ((flags & TEO_GEMM_SWIGLU16) && (N % 32 != 0)) return false;
    auto al = [](const void* p, size_t a) { return p == nullptr || (reinterpret_cast<uintptr_t>(p) % a) == 0; };
    return al(A, 16) && al(W, 16) && al(bias, 8) && al(res, 8) && al(C, 16);
}

int gemm_wide_sk_launch(const void* A, const void* W, const void* bias, const void* res, void* C, int M, int N, int K, int lda, int ldc,
                        int act, bool of32, bool f16, void* sk_ws, size_t flags_offset, hipStream_t st);          // gemm_wide.hip

// 128 x 256 tiles on 256 slots (one 8-wave workgroup per CU) against 128 x 128 tiles on 512 slots: rounds of equal-length tiles
static bool gemm_wide_wins(int M, int N, int K, bool forced, bool swiglu) {
    if (K < 2 * BK) return false;
    const long long t_wide = (long long)cdiv(M, 128) * cdiv(N, 256), t_plain = (long long)cdiv(M, 128) * cdiv(N, 128);
    if (forced) return true;
    // not enough tiles to fill the chip once (240 at M = 638, N = 12288: 57 vs 79 us); with a short K loop (K <= 1024: the tower's
    // qkv at M = 2056, 204 wide tiles) the single round of wide tiles wins from 192 on (23.9 vs 29.6 us, tools/vit_gemm_probe.py)
    // a long K loop on half a round of wide tiles still beats a whole round of 128 x 128 ones (the tower's fc2 at T = 16: M = 4112, N = 1024,
    // K = 4096, 132 wide tiles: 52.6 us against 61-65 us on either 128 x 128 kernel; tools/vit_gemm_probe.py, round 5)
    if (K >= 4096 && t_wide >= 128 && t_wide <= 256) return true;
    // round 6 (tools/dispatch_monotone.py): gate/up below one round of wide tiles (M <= 256: 68.7 us at M = 128 against 78.4 on the 128 x 128 tile --
    // the SwiGLU epilogue has no narrow family), and the short-K shapes from 144 wide tiles on (the tower's fc1 / the projector at T = 4 .. 5:
    // 26.1-26.9 us against 27.7-28.7 on 128 x 128 tiles)
    if (swiglu && t_wide >= 64 && t_wide <= 256) return true;
    if (t_wide < 208 && !(K <= 1024 && t_wide >= 144)) return false;
    // cost in rounds of the plain kernel; its ragged last round runs faster when it leaves one workgroup per CU (x 0.66, measured)
    const long long rem = t_plain % 512;
    const double plain = (double)(t_plain / 512) + (rem == 0 ? 0.0 : (rem <= 256 ? 0.66 : 1.0));
    const double wide = (double)cdiv(t_wide, 256) * GEMM_WIDE_ROUND_COST;
    return wide < plain;
}

template <typename T, typename TO>
static void launch_simple(const void* A, const void* W, const void* bias, const void* res, void* C, int M, int N,
                          int K, int lda, int ldc, int act, bool swiglu, hipStream_t st) {
    dim3 grid(cdiv(N, 64), cdiv(M, 64));
    if (swiglu)
        gemm_simple_kernel<T, TO, true><<<grid, 256, 0, st>>>((const T*)A, (const T*)W, (const T*)bias,
                                                              (const T*)res, (TO*)C, M, N, K, lda, ldc, act);
    else
        gemm_simple_kernel<T, TO, false><<<grid, 256, 0, st>>>((const T*)A, (const T*)W, (const T*)bias,
                                                               (const T*)res, (TO*)C, M, N, K, lda, ldc, act);
}

// the 128 x 128 (or 64 x 128) tile kernel, one workgroup per tile
static int gemm_plain_launch(const void* A, const void* W, const void* bias, const void* res, void* C, int M, int N, int K, int lda, int ldc,
                             int act, bool swiglu, bool of32, bool f16, int bm, hipStream_t st) {
    const int tiles_n = cdiv(N, BN), tiles_m = cdiv(M, bm);
    const int nwg = tiles_m * tiles_n;
    const size_t lds = 4 * TILE_BYTES;
#define TEO_GEMM_KF(SW, OF, DP, MFV, FV)                                                                              \
    gemm_mfma_bf16_kernel<SW, OF, DP, MFV, FV><<<nwg, 256, lds, st>>>((const bf16_t*)A, (const bf16_t*)W, (const bf16_t*)bias, \
                                                                      (const bf16_t*)res, C, M, N, K, lda, ldc, act, tiles_m, \
                                                                      tiles_n)
#define TEO_GEMM_K(SW, OF, DP, MFV) do { if (f16) TEO_GEMM_KF(SW, OF, DP, MFV, true); else TEO_GEMM_KF(SW, OF, DP, MFV, false); } while (0)
#define TEO_GEMM_LAUNCH(SW, OF)                                                                                      \
    if (bm == 64) { TEO_GEMM_K(SW, OF, 2, 2); }                                                                       \
    else if ((tune().gemm_depth == 0 && !(SW)) || tune().gemm_depth == 2) { TEO_GEMM_K(SW, OF, 2, 4); }                         \
    else { TEO_GEMM_K(SW, OF, 1, 4); }
        if (swiglu) { if (of32) { TEO_GEMM_LAUNCH(true, true) } else { TEO_GEMM_LAUNCH(true, false) } }
        else        { if (of32) { TEO_GEMM_LAUNCH(false, true) } else { TEO_GEMM_LAUNCH(false, false) } }
#undef TEO_GEMM_K
#undef TEO_GEMM_KF
#undef TEO_GEMM_LAUNCH
        note_kernel(bm == 64 ? "gemm_mfma_64" : "gemm_mfma_128");
    TEO_LAUNCH_CHECK("gemm_mfma_bf16");
    return TEO_OK;
}

int gemm(const void* A, const void* W, const void* bias, const void* res, void* C, int M, int N, int K, int lda,
         int ldc, int act, unsigned flags, int dtype, int out_dtype, hipStream_t st, void* sk_ws) {
    if (M == 0 || N == 0) return TEO_OK;
    if (sk_ws && !sk_grid_fits_device()) sk_ws = nullptr;        // stream-K / hybrid grids are sized for 256 CUs
    const bool swiglu = flags & TEO_GEMM_SWIGLU16;
    if (swiglu && (bias || res || act != TEO_ACT_NONE || N % 32 != 0)) {
        set_error("teo_gemm: SWIGLU16 needs N %% 32 == 0 and no bias/residual/act");
        return TEO_ERR_ARG;
    }
    const bool f16 = dtype == TEO_F16;                       // the 16-bit format of this call, handed to the launch helper of every tile family
    if (gemm_mfma_ok(M, N, K, lda, ldc, dtype, flags, A, W, bias, res, C)) {
        // tile height: 128 rows; 64 rows only for small problems whose 128-row tiling leaves more than half of the 512
        // resident workgroup slots empty (ViT o / fc2: 136 tiles; +7 % there).  Measured at M = 2168: 64-row tiles lose
        // 10-25 % on every LLaMA shape (half the weight reuse per tile), wave quantisation notwithstanding.
        const int tiles_n = cdiv(N, BN);
        int bm = tune().gemm_bm;
        if (bm == 0) bm = (cdiv(M, 128) * tiles_n <= 256 && !swiglu) ? 64 : 128;
        const int tiles_m = cdiv(M, bm);
        const int nwg = tiles_m * tiles_n;
        const size_t lds = 4 * TILE_BYTES;
        const bool of32 = out_dtype == TEO_F32;
        // narrow LDS-DMA tiles (gemm_narrow.hip, round 5): forced here; the automatic rule sits below, after the families it competes with
        if (tune().gemm_narrow == 2 && tune().gemm_narrow_pipe == 2 && swiglu)          // forced: the software-pipelined small tiles with the SwiGLU epilogue
            return gemm_pipe_launch(A, W, bias, res, C, M, N, K, lda, ldc, act, of32, f16, tune().gemm_narrow_bm == 128 ? 128 : 64, tune().gemm_pipe_bn,
                                    tune().gemm_pipe_stages, st, true);
        if (tune().gemm_narrow == 2 && !swiglu)
            return gemm_narrow_launch(A, W, bias, res, C, M, N, K, lda, ldc, act, of32, f16, tune().gemm_narrow_bm == 128 ? 128 : 64, st,
                                      tune().gemm_narrow_bm == 128 && tune().gemm_narrow_waves == 8);
        // gate/up + SwiGLU at M <= 128 (a text-only prompt): one or two row tiles of 128 x 256 leave two thirds of the CUs idle (86 / 172 tiles: 66-68 us);
        // the software-pipelined small tiles carry the SwiGLU epilogue too -- 64 x 128 at M <= 64 (344 tiles: 42 us), 128 x 128 at M <= 128 (172
        // tiles: 50 us); from M = 129 on the 128 x 256 / 256 x 256 tiles are ahead again (tools/dispatch_monotone.py, round 6)
        if (swiglu && M <= 128 && K >= 2048 && tune().gemm_narrow == 1 && tune().gemm_narrow_pipe == 1 && tune().gemm_bm == 0 && tune().gemm_wide == 1 &&
            tune().gemm_big == 1)
            return gemm_pipe_launch(A, W, bias, res, C, M, N, K, lda, ldc, act, of32, f16, M <= 64 ? 64 : 128, 128, 4, st, true);
        if (tune().gemm_quad == 2 && !swiglu) return gemm_quad_launch(A, W, bias, res, C, M, N, K, lda, ldc, act, of32, f16, st);      // forced (its rule: below)
        // automatic: wherever the 64-row register-staged kernel was the choice (few tiles: the tower's out_proj / fc2, every tower GEMM and
        // the LLaMA o / down projections of config C2) the 64 x 128 LDS-DMA tile runs instead -- tools/vit_gemm_probe.py (round 5, us):
        // fc2 47.1 -> 36.5, out_proj 17.2 -> 14.6 (T = 8); at T = 2: fc2 42.6 -> 32.6, fc1 23.0 -> 18.6, LLaMA o 60.8 -> 39.2, down 148 -> 93
        // round 6: with 192 .. 256 tiles of 128 x 128 (one per CU, three quarters of the chip or more), a long K loop and a wide N -- LLaMA
        // o / down at M = 641 .. 1024 -- the EIGHT-wave 128 x 128 tile (two waves per SIMD on one tile per CU) beats the two four-wave 64 x 128
        // tiles per CU: o 41.5-43.1 vs 53.2-55.0 us, down 107-109 vs 134-136 (tools/vit_gemm_probe.py); it loses at 160 tiles (M = 638: 41.0 /
        // 103.4 vs 39.9 / 93.0) and on the tower's N = 1024 shapes (fc2 39.8 vs 37.0), which stay on the 64 x 128 tile
        // round 6, late: the software-pipelined K loop on tiles sized for about ONE workgroup per CU (gemm_quad.hip gemm_pipe_launch; every
        // candidate on every shape, weights from HBM: profiles/r06_pipe_candidates.txt; us, against the LDS-DMA tiles of gemm_narrow.hip).
        // t64 / t128 / t96 / tq: tiles of 64 x 64 / 64 x 128 / 128 x 96 / 128 x 128.
        //   t64 <= 288 (more with a short K loop): 64 x 64 -- LLaMA o / down at M <= 256 20 / 55 against 34 / 86; the tower at T <= 4: fc2 32-33 ->
        //       15-18, out_proj 10.8 -> 6.3-6.6, qkv 11 -> 6.1-8.7, fc1 16 -> 9-11.6;
        //   t128 <= 256: 64 x 128, ring of 4 (one per CU) for a long K loop -- o / down at M = 257 .. 512 28-31 / 79 against 35 / 86; fc2 at
        //       T = 5 .. 7 27-31 against 33-35; out_proj at T = 5 .. 7 and qkv at T = 2 9.1-11.5 against 11.2-12.7;
        //   t96 <= 256: 128 x 96 -- the tower's fc2 / out_proj at T = 8 .. 11 (187 .. 253 tiles) 31-36 / 12.4-13.4 against 41-43 / 15.3-16.1, qkv
        //       at T = 3 .. 4 11 against 13.5-14; LLaMA o / down at M = 513 .. 640 (215 tiles, ring of 4) 30-36 / 77-90 against 41 / 90-92;
        //   else, wide N and long K from 192 tiles of 128 x 128 on (o / down at M = 641 .. 1024): 128 x 128 -- 39-42 / 94-102 against 43-44 /
        //       97-104 on the eight-wave LDS-DMA tile.
        //   Not taken: a short K loop with an activation epilogue and more than 256 tiles of 64 x 128 (fc1 + GELU at T = 3 .. 4: one wave per
        //   SIMD evaluates its 64-128 erf alone; the old 64 x 128 kernel's two workgroups per CU alternate: 18-19 against 20-25), and the
        //   tower's N = 1024 shapes beyond 256 tiles of 128 x 96 (T >= 12: within 5 % either way).
        if (tune().gemm_narrow == 1 && tune().gemm_narrow_pipe == 1 && tune().gemm_bm == 0 && tune().gemm_narrow_waves == 0 && bm == 64) {
            const long long cm64 = cdiv(M, 64), cm128 = cdiv(M, 128);
            const long long t64 = cm64 * cdiv(N, 64), t128 = cm64 * cdiv(N, 128), t96 = cm128 * cdiv(N, 96), tq = cm128 * tiles_n;
            const bool long_wide = K >= 2048 && N >= 2048;
            if (t64 <= 288 || (K <= 1024 && (t64 <= 384 || (N >= 2048 && t64 <= 512))))
                return gemm_pipe_launch(A, W, bias, res, C, M, N, K, lda, ldc, act, of32, f16, 64, 64, 4, st);
            if (t128 <= 256)
                return gemm_pipe_launch(A, W, bias, res, C, M, N, K, lda, ldc, act, of32, f16, 64, 128, K >= 2048 ? 4 : 3, st);
            if (!(K <= 1024 && act != TEO_ACT_NONE)) {
                if (t96 <= 256)
                    return gemm_pipe_launch(A, W, bias, res, C, M, N, K, lda, ldc, act, of32, f16, 128, 96, long_wide ? 4 : 3, st);
                if (long_wide && tq >= 192)
                    return gemm_pipe_launch(A, W, bias, res, C, M, N, K, lda, ldc, act, of32, f16, 128, 128, 3, st);
            }
        }
        if (tune().gemm_narrow == 1 && tune().gemm_bm == 0 && tune().gemm_narrow_waves != 4 && bm == 64 && K >= 2048 && N >= 2048 &&
            (long long)cdiv(M, 128) * tiles_n >= 192)
            return gemm_narrow_launch(A, W, bias, res, C, M, N, K, lda, ldc, act, of32, f16, 128, st, true);
        if (tune().gemm_narrow == 1 && tune().gemm_bm == 0 && bm == 64)
            return gemm_narrow_launch(A, W, bias, res, C, M, N, K, lda, ldc, act, of32, f16, 64, st);
        const long long t_wide_ = (long long)cdiv(M, 128) * cdiv(N, 256), t_big = gemm_big_tile_count(M, N, K);  // (counts a ragged last row block as 128 x 512 tiles)
        // 256 x 160 tiles (gemm_quad.hip, round 5): automatic, and only while no other family is forced, where the problem is ONE round of
        // them but more than one round of 128 x 256 tiles: M = 2056 .. 2304 against N = 4096 (272 wide tiles, 234 of these): LLaMA o / down at
        // config C3, the tower's fc1.  tools/vit_gemm_probe.py (us, real epilogues): o 79.7 -> 71.3, down 178.6 -> 170.7, fc1 + GELU 39.5 -> 34.2
        // on the eight-wave form (the default); the four-wave form (one wave per SIMD) ties it on o / down and loses 17 us on fc1 + GELU
        const bool one_round_160 = !swiglu && tune().gemm_quad == 1 && tune().gemm_bm == 0 && tune().gemm_wide == 1 && tune().gemm_big == 1 &&
                                   tune().gemm_sk == 1 && tune().gemm_narrow == 1 && K >= 8 * BK && t_wide_ > 256 &&
                                   (long long)cdiv(M, 256) * cdiv(N, 160) <= std::min(device_cu_count(), 256);
        if (one_round_160) return gemm_quad_launch(A, W, bias, res, C, M, N, K, lda, ldc, act, of32, f16, st);
        // just over one round of wide tiles -- or, for a short K loop (K <= 1024: the tower's fc1 at T = 16, 528 tiles), just over two:
        // there a ragged third round costs a third of the launch (wide 86.7 us, its stream-K form 65.5; tools/vit_gemm_probe.py, round 5)
        // round 6 (tools/shape_sweep.py, tools/dispatch_probe.py over M = 767 .. 3328): with a LONG K loop (K >= 2048: the LLaMA shapes) the
        // stream-K form keeps winning up to 1.375 tiles per workgroup -- o / down at M = 2305 .. 2816 (304 .. 352 wide tiles) 86-97 / 208-242 us
        // against 113-136 / 298-343 us on the 128 x 128 tiles the rounds model fell back to, qkv at M = 769 .. 896 (336 tiles) 93-95 vs 128-132
        const long long sk_wide_rem = K >= 2048 ? 96 : 256 / 6;
        const long long sk_wide_max = K <= 1024 ? 2 * 256 + 256 / 6 : 256 + sk_wide_rem;
        const bool sk_wide_fit = t_wide_ > 256 && t_wide_ <= sk_wide_max && (t_wide_ % 256) != 0 && (t_wide_ % 256) <= sk_wide_rem;
        const bool sk_wide_shape = sk_ws && tune().gemm_sk && tune().gemm_wide && !swiglu && sk_wide_fit;
        // 256 x 256 tiles: a round of them costs GEMM_BIG_ROUND_COST rounds of the 128 x 256 kernel for twice the area (measured
        // 1.45-1.7 us against 0.875 us per K tile); taken when that beats the wide kernel's round count and the chip is filled
        // (with a workspace its hybrid form has no ragged last round: fractional rounds + a hand-off allowance)
        const double big_rounds = (sk_ws && t_big > 256 && t_big % 256 != 0 && gemm_big_hybrid_fits(M, N, K)) ? (double)t_big / 256.0 + 0.12
                                                                                                                 : (double)cdiv(t_big, 256);
        // round 6: three quarters of a round of 256 x 256 tiles already beats the alternatives when the K loop is long (K >= 2048): qkv at
        // M = 897 .. 1024 (192 tiles) 97-99 us vs 137-139 on 128 x 128 tiles, o / down at M = 2817 .. 3328 (192 / 208 tiles) 102-108 / 252-257 vs
        // 122-142 / 300-365; and at EQUAL modelled cost the 256 x 256 tile is the one that measures ahead (gate/up at M = 2305 .. 2560: four
        // rounds of them 393 us, seven rounds of 128 x 256 tiles 434-443) -- hence <=
        // (tools/dispatch_monotone.py, round 6: a GEMM with more rows cannot be faster -- every inversion it found was a threshold here.)  192 tiles for
        // every K (the tower's qkv at T = 15 / 16: 32.8 vs 37.6 us; fc1 / the projector at T = 12: 46.8 vs 52.1, 48.2 vs 50.0); 160 for the SwiGLU
        // epilogue, whose only other families are the 128 x 256 tile and the register-staged 128 x 128 one (gate/up at M = 257 .. 512, 172 tiles:
        // 92-94 us against 103-150)
        // (late round 6, weights from HBM: 156 for the other epilogues too -- the tower's fc1 at T = 10 .. 11 (160 / 176 tiles) 47 us against 49-53 on the
        // two-stage 128 x 128 LDS-DMA tile, qkv at T = 13 .. 15 (156 .. 180 tiles of 256 x 256: 13-15 x 12) 32-33 against 33-35)
        const long long t_big_min = swiglu ? 160 : 156;
        if (bm == 128 && K >= 2 * BK && (tune().gemm_big == 2 || (tune().gemm_big == 1 && tune().gemm_wide == 1 && t_big >= t_big_min && !sk_wide_shape &&
                                                               big_rounds * GEMM_BIG_ROUND_COST <= (double)cdiv(t_wide_, 256))))
            return gemm_big_launch(A, W, bias, res, C, M, N, K, lda, ldc, act, swiglu, of32, f16, st, sk_ws, GEMM_SK_SLAB_BYTES);
        {   // just over one round of WIDE tiles (272 on 256 CUs: o / down at M = 2168): the stream-K form of the wide kernel
            const long long t_wide = (long long)cdiv(M, 128) * cdiv(N, 256);
            if (sk_ws && tune().gemm_sk && tune().gemm_wide && bm == 128 && !swiglu && K >= 2 * BK && t_wide > 256 &&
                (tune().gemm_sk == 2 || sk_wide_fit))
                return gemm_wide_sk_launch(A, W, bias, res, C, M, N, K, lda, ldc, act, of32, f16, sk_ws, GEMM_SK_SLAB_BYTES, st);
        }
        if (tune().gemm_wide && bm == 128 && gemm_wide_wins(M, N, K, tune().gemm_wide == 2, swiglu))
            return gemm_wide_launch(A, W, bias, res, C, M, N, K, lda, ldc, act, swiglu, of32, f16, st);
        // stream-K where it was measured to win: just over ONE round of tiles (544 tiles on 512 slots at M = 2168, N = 4096:
        // o 114 -> 93 us, down 297 -> 250 us).  With several tiles per workgroup the contiguous ranges spread an XCD's
        // concurrent tiles over three times as many W panels as the plain kernel's rolling window does and the L2 misses
        // cost more than the idle tail of the last round saves (qkv, 3.19 rounds: 255 -> 330 us; 1.5 rounds: 118 -> 130 us).
        if (sk_ws && tune().gemm_sk && bm == 128 && nwg > SK_MAX_GRID &&
            (tune().gemm_sk == 2 || (nwg < 2 * SK_MAX_GRID && (nwg % SK_MAX_GRID) <= SK_MAX_GRID / 6))) {
            const int nk = K / BK;
            const long long total = (long long)nwg * nk;
            const int per = (int)((total + SK_MAX_GRID - 1) / SK_MAX_GRID);          // >= nk because nwg > SK_MAX_GRID
            float* slabs = (float*)sk_ws;
            int* flg = (int*)((unsigned char*)sk_ws + GEMM_SK_SLAB_BYTES);
#define TEO_SK_LAUNCH_F(SW, OF, FV)                                                                                   \
    gemm_mfma_bf16_sk_kernel<SW, OF, FV><<<SK_MAX_GRID, 256, lds, st>>>((const bf16_t*)A, (const bf16_t*)W, (const bf16_t*)bias,  \
                                                                         (const bf16_t*)res, C, M, N, K, lda, ldc, act, tiles_m, \
                                                                         tiles_n, per, slabs, flg)
#define TEO_SK_LAUNCH(SW, OF) do { if (f16) TEO_SK_LAUNCH_F(SW, OF, true); else TEO_SK_LAUNCH_F(SW, OF, false); } while (0)
            if (swiglu) { if (of32) TEO_SK_LAUNCH(true, true); else TEO_SK_LAUNCH(true, false); }
            else { if (of32) TEO_SK_LAUNCH(false, true); else TEO_SK_LAUNCH(false, false); }
#undef TEO_SK_LAUNCH
#undef TEO_SK_LAUNCH_F
            note_kernel("gemm_mfma_128_sk");
            TEO_LAUNCH_CHECK("gemm_mfma_bf16_sk");
            return TEO_OK;
        }
        // what no other family took: the 128 x 128 LDS-DMA tile instead of the register-staged one (the tower's fc2 / out_proj at T = 16:
        // 75.5 -> 59.0 us, 25.8 -> 23.8); the register-staged kernel keeps the SwiGLU epilogue and stays the reference form of the tests
        if (tune().gemm_narrow == 1 && tune().gemm_bm == 0 && bm == 128 && !swiglu)
            return gemm_narrow_launch(A, W, bias, res, C, M, N, K, lda, ldc, act, of32, f16, 128, st);
        return gemm_plain_launch(A, W, bias, res, C, M, N, K, lda, ldc, act, swiglu, of32, f16, bm, st);
    }
    if (dtype == TEO_F32) {
        if (out_dtype != TEO_F32) { set_error("teo_gemm: f32 inputs need f32 output"); return TEO_ERR_UNSUPPORTED; }
        launch_simple<float, float>(A, W, bias, res, C, M, N, K, lda, ldc, act, swiglu, st);
    } else if (dtype == TEO_BF16) {
        if (out_dtype == TEO_F32) launch_simple<bf16_t, float>(A, W, bias, res, C, M, N, K, lda, ldc, act, swiglu, st);
        else launch_simple<bf16_t, bf16_t>(A, W, bias, res, C, M, N, K, lda, ldc, act, swiglu, st);
    } else if (dtype == TEO_F16) {
        if (out_dtype == TEO_F32) launch_simple<f16_t, float>(A, W, bias, res, C, M, N, K, lda, ldc, act, swiglu, st);
        else launch_simple<f16_t, f16_t>(A, W, bias, res, C, M, N, K, lda, ldc, act, swiglu, st);
    } else {
        set_error("teo_gemm: unknown dtype %d", dtype);
        return TEO_ERR_UNSUPPORTED;
    }
    note_kernel("gemm_simple");
    TEO_LAUNCH_CHECK("gemm_simple");
    return TEO_OK;
}

}  // namespace teo
