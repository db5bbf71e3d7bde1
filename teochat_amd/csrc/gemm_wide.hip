// Wide-tile form of the bf16 MFMA GEMM for the big prefill shapes: 128 (M) x 256 (N) x 64 workgroup tile, 8 waves (2 x 4) of
// 64 x 64, operands brought in by LDS-DMA (global_load_lds_dwordx4: no VGPR round trip, no ds_write pass) into a THREE-stage
// LDS ring (3 x 48 KB of the CU's 160 KB) with two K tiles in flight and ONE barrier per K tile:
//
//     for kt:   s_waitcnt vmcnt(6)      <- this wave's pieces of tile kt have landed (tile kt+1's 6 pieces may still fly)
//               s_barrier               <- every wave's pieces of kt landed; everybody is done reading stage (kt-1) % 3
//               issue tile kt+2 -> stage (kt+2) % 3 == (kt-1) % 3
//               16 ds_read_b128 + 32 MFMAs per wave on stage kt % 3
//
// Why (ablation of the 128 x 128 register-staged kernel at M = 2168, N = 12288, K = 4096; gemm.hip): its global loads ALONE
// take 157 us (3.4 GB per GEMM through L2 at ~22 TB/s), its LDS writes alone 91 us, its ds_read + MFMA loop alone 150 us (58 %
// of the MFMA peak), all three together 252-278 us.  This kernel moves 25 % fewer bytes per FLOP (48 KB per 4.2 MFLOP instead
// of 32 KB per 2.1), has no LDS write instructions at all, keeps two tiles in flight without spending registers, and M = 2168
// still tiles without waste (17 x 128).  LDS image and fragment reads are the plain kernel's: 128-byte rows, 16-byte chunk c of
// row r at position c ^ (r & 7) -- with LDS-DMA the swizzle is applied to the per-lane SOURCE address (the destination is
// wave-linear: 1 KB = 8 rows per instruction), which keeps every 128-byte row a single coalesced request.
// Same k-order per output element as the plain kernel -> bit-identical results (tests/test_kernels_gpu.py).
#include "common.h"
#include "gemm_epilogue.h"
#include "ops.h"

namespace teo {

typedef __attribute__((ext_vector_type(8))) short gw_bf16x8;
typedef __attribute__((ext_vector_type(4))) float gw_f32x4;

constexpr int GW_BM = 128, GW_BN = 256, GW_BK = 64;
constexpr int GW_A_BYTES = GW_BM * GW_BK * 2;            // 16 KiB
constexpr int GW_W_BYTES = GW_BN * GW_BK * 2;            // 32 KiB
constexpr int GW_STAGE = GW_A_BYTES + GW_W_BYTES;        // 48 KiB
constexpr int GW_PIECES = GW_STAGE / 1024 / 8;           // 1-KiB DMA pieces per wave per K tile = 6

// tune().gemm_wide_sched (default 1: the skewed / carried K-loop order), tune().gemm_wide_group (default 0: chosen from the tile grid)

__device__ __forceinline__ int gw_xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    const int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    return base + (bid >> 3);
}

// One K tile of a wave: 16 fragment reads + 32 MFMAs on stage `st`, and the DMA issue of tile kt+2 (`dma()`), in the order that
// measured best (tools/bench_kernels.py gemm_wide_sched; qkv 256 -> 225-234 us, gate/up 370 -> 345 us, bit-identical):
//  * skewed halves: waves 4-7 (the second wave of every SIMD) issue their six DMA pieces first, waves 0-3 only after their first
//    MFMA block -- one wave of a SIMD multiplies while its partner is in its load segment (both in lockstep: 44 % MFMA-busy;
//    skewed: 51 %).  s_setprio on either half loses, DMA pieces interleaved one by one between MFMAs change nothing;
//  * carried half: the second-half fragments of tile kt are multiplied right after the barrier of tile kt+1, from registers,
//    while the first-half fragments of kt+1 are on their way from LDS -- no wave sits behind a barrier without MFMA work.
// The MFMA chain of every output element is still k-ascending: h0(kt-1), h1(kt-1), h0(kt), ... -> results unchanged.
// Ablation at qkv (M = 2168, N = 12288, K = 4096; us): all 234 | no DMA 196 | no MFMA 169 | no fragment reads 173 | MFMA + barrier
// only 150 | DMA only 146 (3.4 GB / 146 us = 22 TB/s out of L2): the L2 -> LDS delivery is as long as the MFMA work.
template <bool F16, typename DMA>
__device__ __forceinline__ void gw_ktile(const unsigned char* sA, const unsigned char* sB, int wid, int wm, int wn, int fr, int fg,
                                         bool first, gw_f32x4 (&acc)[4][4], gw_bf16x8 (&caf)[4], gw_bf16x8 (&cwf)[4], DMA&& dma) {
    const bool late = wid < 4;
    if (!late) dma();
    __builtin_amdgcn_sched_barrier(0);
    gw_bf16x8 af[4], wf[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ra_ = wm * 64 + i * 16 + fr;
        af[i] = *reinterpret_cast<const gw_bf16x8*>(sA + ra_ * 128 + ((fg ^ (ra_ & 7)) << 4));
        const int rw_ = wn * 64 + i * 16 + fr;
        wf[i] = *reinterpret_cast<const gw_bf16x8*>(sB + rw_ * 128 + ((fg ^ (rw_ & 7)) << 4));
    }
    __builtin_amdgcn_sched_barrier(0);
    if (!first) {
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
                acc[ni][mi] = mfma16<F16>(cwf[ni], caf[mi], acc[ni][mi]);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (late) dma();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ra_ = wm * 64 + i * 16 + fr;
        caf[i] = *reinterpret_cast<const gw_bf16x8*>(sA + ra_ * 128 + (((4 + fg) ^ (ra_ & 7)) << 4));
        const int rw_ = wn * 64 + i * 16 + fr;
        cwf[i] = *reinterpret_cast<const gw_bf16x8*>(sB + rw_ * 128 + (((4 + fg) ^ (rw_ & 7)) << 4));
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
            acc[ni][mi] = mfma16<F16>(wf[ni], af[mi], acc[ni][mi]);
}
// the carried second half of the last K tile of a segment
template <bool F16>
__device__ __forceinline__ void gw_flush(gw_f32x4 (&acc)[4][4], const gw_bf16x8 (&caf)[4], const gw_bf16x8 (&cwf)[4]) {
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
            acc[ni][mi] = mfma16<F16>(cwf[ni], caf[mi], acc[ni][mi]);
}

template <bool SWIGLU, bool OUT_F32, int SCHED, bool F16 = false>
__global__ __launch_bounds__(512, 2) void gemm_mfma_bf16_wide_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W,
                                                                  const bf16_t* __restrict__ bias, const bf16_t* res, void* Cv,
                                                                  int M, int N, int K, int lda, int ldc, int act, int tiles_m,
                                                                  int tiles_n, int group) {
    constexpr int sched = SCHED;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 2, wn = wid & 3;
    const int tile = gw_xcd_remap(blockIdx.x, tiles_m * tiles_n);
    // tiles walk `group` N panels at a time, N fastest: the 32 workgroups an XCD runs together then cover ~(32 / group) x group
    // tiles (8 x 4: 256 KB of distinct operand rows per K tile instead of 336 KB with whole columns) -- better L2 hit rate
    const int gsz = group * tiles_m, sup = tile / gsz, rem = tile - sup * gsz;
    const int gn = min(group, tiles_n - sup * group);
    const int tm = rem / gn, tn = sup * group + rem % gn;
    const int m0 = tm * GW_BM, n0 = tn * GW_BN;
    const int fr = lane & 15, fg = lane >> 4;
    const int nk = K / GW_BK;

    // DMA pieces of this wave: piece g = wid * 6 + j covers 8 rows (g < 16: A rows 8g.., else W rows 8 (g - 16)..); lane l brings
    // row (l >> 3) of the piece, logical chunk (l & 7) ^ (l >> 3), to LDS byte g * 1024 + l * 16 of the stage
    const bf16_t* src[GW_PIECES];
#pragma unroll
    for (int j = 0; j < GW_PIECES; ++j) {
        const int g = wid * GW_PIECES + j;
        const int rl = lane >> 3, c = (lane & 7) ^ rl;
        if (g < 16) src[j] = A + (long long)min(m0 + g * 8 + rl, M - 1) * lda + c * 8;
        else src[j] = W + (long long)min(n0 + (g - 16) * 8 + rl, N - 1) * K + c * 8;
    }
#define TEO_GW_STAGE(KT, ST)                                                                                              \
    _Pragma("unroll") for (int j = 0; j < GW_PIECES; ++j)                                                                 \
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[j] + (long long)(KT) * GW_BK), \
                                         (__attribute__((address_space(3))) void*)(smem + (ST) * GW_STAGE + (wid * GW_PIECES + j) * 1024), 16, 0, 0);

    uint2 bv[4];                                         // bias of the lane's columns: requested now, used after the K loop (gemm_epilogue.h)
    gemm_bias_load<4>(SWIGLU ? nullptr : bias, n0 + wn * 64, fg, N, bv);
    gw_f32x4 acc[4][4];   // [ni][mi]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (gw_f32x4){0.f, 0.f, 0.f, 0.f};

    TEO_GW_STAGE(0, 0)
    if (nk > 1) TEO_GW_STAGE(1, 1)
    gw_bf16x8 caf[4], cwf[4];                            // second-half fragments carried over the next barrier (gw_ktile)
    int st = 0;                                          // stage of tile kt
    for (int kt = 0; kt < nk; ++kt) {
        // tile kt landed (this wave), then for everyone; the barrier also says stage (kt + 2) % 3 is no longer being read
        if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const int st2 = st == 0 ? 2 : st - 1;             // (kt + 2) % 3
        const unsigned char* sA = smem + st * GW_STAGE;
        const unsigned char* sB = sA + GW_A_BYTES;
        if (sched == 0) {                                 // the plain order (kept for A/B: gemm_wide_sched = 0)
            if (kt + 2 < nk) { TEO_GW_STAGE(kt + 2, st2) }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                gw_bf16x8 af[4], wf[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int ra_ = wm * 64 + i * 16 + fr;
                    af[i] = *reinterpret_cast<const gw_bf16x8*>(sA + ra_ * 128 + (((ks * 4 + fg) ^ (ra_ & 7)) << 4));
                    const int rw_ = wn * 64 + i * 16 + fr;
                    wf[i] = *reinterpret_cast<const gw_bf16x8*>(sB + rw_ * 128 + (((ks * 4 + fg) ^ (rw_ & 7)) << 4));
                }
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi)
                        acc[ni][mi] = mfma16<F16>(wf[ni], af[mi], acc[ni][mi]);
            }
        } else {
            gw_ktile<F16>(sA, sB, wid, wm, wn, fr, fg, kt == 0, acc, caf, cwf, [&]() { if (kt + 2 < nk) { TEO_GW_STAGE(kt + 2, st2) } });
        }
        st = st == 2 ? 0 : st + 1;
    }
#undef TEO_GW_STAGE
    if (sched != 0 && nk > 0) gw_flush<F16>(acc, caf, cwf);

    gemm_epilogue<4, 4, 2, SWIGLU, OUT_F32, F16>(acc, bv, bias != nullptr, res, Cv, M, N, ldc, act, m0 + wm * 64, n0 + wn * 64, fr, fg);
}

// ------------------------------------------------------------------------------------------------
// stream-K form of the wide kernel for shapes that are just over ONE round of wide tiles (272 tiles on 256 CUs at M = 2168,
// N = 4096: o and down).  Same scheme as gemm_mfma_bf16_sk_kernel (gemm.hip): a persistent grid of one workgroup per CU, the
// flattened (tile, k-tile) space cut into equal contiguous ranges; a range that ends inside a tile accumulates that part FIRST
// and hands its fp32 accumulators (128 KB) to the next range's owner, which loads them as its INITIAL accumulators and
// continues the k-loop LAST in its own timeline -- sequential k-order per output element, bit-identical results.  Hand-off by
// 16-byte sc1 buffer stores / loads + a relaxed agent-scope flag (no fences).  One K-loop body serves all three kinds of segment.
// ------------------------------------------------------------------------------------------------
constexpr int GW_SLAB_FLOATS = GW_BM * GW_BN;            // 128 KB of fp32 per workgroup

template <bool OUT_F32, bool F16 = false>
__global__ __launch_bounds__(512, 2) void gemm_mfma_bf16_wide_sk_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W,
                                                                     const bf16_t* __restrict__ bias, const bf16_t* res, void* Cv,
                                                                     int M, int N, int K, int lda, int ldc, int act, int tiles_m,
                                                                     int tiles_n, int per, float* slabs, int* flags) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 2, wn = wid & 3;
    const int fr = lane & 15, fg = lane >> 4;
    const int nk = K / GW_BK;
    const long long total = (long long)tiles_m * tiles_n * nk;
    const int q = gw_xcd_remap(blockIdx.x, gridDim.x);
    const long long it0 = (long long)q * per, it1 = min(it0 + per, total);
    if (it0 >= total) return;
    const int t_first = (int)(it0 / nk), k_first = (int)(it0 % nk);
    const int t_last = (int)((it1 - 1) / nk), k_end = (int)(it1 - (long long)t_last * nk);
    const int has_head = k_first != 0, has_tail = k_end != nk;           // per >= nk: a tile has at most two owners
    const int t_full0 = t_first + has_head, n_full = (t_last + 1 - has_tail) - t_full0;
    const int nseg = has_tail + n_full + has_head;

    gw_f32x4 acc[4][4];
    gw_bf16x8 caf[4], cwf[4];
    for (int sgi = 0; sgi < nseg; ++sgi) {
        // order: tail (its partial sums are needed by the neighbour), the full tiles, head (the neighbour's partial sums are long there)
        const bool is_tail = has_tail && sgi == 0;
        const bool is_head = has_head && sgi == nseg - 1;
        const int pos = is_tail ? t_last : (is_head ? t_first : t_full0 + (sgi - has_tail));
        const int kb = is_head ? k_first : 0, ke = is_tail ? k_end : nk;
        const int tm = pos % tiles_m, tn = pos / tiles_m;
        const int m0 = tm * GW_BM, n0 = tn * GW_BN;
        const bf16_t* src[GW_PIECES];
#pragma unroll
        for (int j = 0; j < GW_PIECES; ++j) {
            const int g = wid * GW_PIECES + j;
            const int rl = lane >> 3, c = (lane & 7) ^ rl;
            if (g < 16) src[j] = A + (long long)min(m0 + g * 8 + rl, M - 1) * lda + c * 8;
            else src[j] = W + (long long)min(n0 + (g - 16) * 8 + rl, N - 1) * K + c * 8;
        }
#define TEO_GW_STAGE(KT, ST)                                                                                              \
    _Pragma("unroll") for (int j = 0; j < GW_PIECES; ++j)                                                                 \
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[j] + (long long)(KT) * GW_BK), \
                                         (__attribute__((address_space(3))) void*)(smem + (ST) * GW_STAGE + (wid * GW_PIECES + j) * 1024), 16, 0, 0);
        if (is_head) {
            if (tid == 0) {
                int spins = 0;
                while (__hip_atomic_load(flags + q - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
                    __builtin_amdgcn_s_sleep(8);
                    if (++spins > (1 << 24)) { __hip_atomic_store(flags + GEMM_SK_ERR_SLOT, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }   // never reached (the producer wrote its slab first thing); a miss is STICKY: teo_gemm_workspace_status
                }
                __hip_atomic_store(flags + q - 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // re-armed for the next launch
            }
            __builtin_amdgcn_s_barrier();
            const auto sl = __builtin_amdgcn_make_buffer_rsrc(slabs + (size_t)(q - 1) * GW_SLAB_FLOATS, 0, GW_SLAB_FLOATS * 4, 0x00020000);
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
                    acc[ni][mi] = __builtin_bit_cast(gw_f32x4, __builtin_amdgcn_raw_buffer_load_b128(sl, ((ni * 4 + mi) * 512 + tid) * 16, 0, /*sc1*/ 16));
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = (gw_f32x4){0.f, 0.f, 0.f, 0.f};
        }
        // bias of the lane's columns (segments that end in an epilogue): requested ahead of the K loop (gemm_epilogue.h)
        uint2 bv[4];
        gemm_bias_load<4>(bias, n0 + wn * 64, fg, N, bv);
        // the ring is primed AFTER the slab loads were issued: vmcnt retires loads in order, so the counted waits below still
        // mean "this wave's pieces of tile kt have landed" (the 16 slab loads of a head segment are older and retire first)
        TEO_GW_STAGE(kb, 0)
        if (kb + 1 < ke) TEO_GW_STAGE(kb + 1, 1)
        int st = 0;
        for (int kt = kb; kt < ke; ++kt) {
            if (kt + 1 < ke) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const int st2 = st == 0 ? 2 : st - 1;
            const unsigned char* sA = smem + st * GW_STAGE;
            const unsigned char* sB = sA + GW_A_BYTES;
            gw_ktile<F16>(sA, sB, wid, wm, wn, fr, fg, kt == kb, acc, caf, cwf, [&]() { if (kt + 2 < ke) { TEO_GW_STAGE(kt + 2, st2) } });
            st = st == 2 ? 0 : st + 1;
        }
#undef TEO_GW_STAGE
        if (ke > kb) gw_flush<F16>(acc, caf, cwf);
        __builtin_amdgcn_s_barrier();                       // every wave is done reading the ring before the next segment refills it
        if (is_tail) {
            const auto sl = __builtin_amdgcn_make_buffer_rsrc(slabs + (size_t)q * GW_SLAB_FLOATS, 0, GW_SLAB_FLOATS * 4, 0x00020000);
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned int, acc[ni][mi]), sl,
                                                           ((ni * 4 + mi) * 512 + tid) * 16, 0, /*sc1*/ 16);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // every storing wave: write-through stores landed
            __builtin_amdgcn_s_barrier();
            if (tid == 0) __hip_atomic_store(flags + q, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            continue;
        }
        // epilogue (no SwiGLU in this form: it serves the N = hidden layers, o and down)
        gemm_epilogue<4, 4, 2, false, OUT_F32, F16>(acc, bv, bias != nullptr, res, Cv, M, N, ldc, act, m0 + wm * 64, n0 + wn * 64, fr, fg);
    }
}

// slabs: 256 x 128 KB (the workspace of gemm_sk_workspace_bytes() holds 512 x 64 KB) + flags behind them
int gemm_wide_sk_launch(const void* A, const void* W, const void* bias, const void* res, void* C, int M, int N, int K, int lda, int ldc,
                        int act, bool of32, bool f16, void* sk_ws, size_t flags_offset, hipStream_t st) {
    const int tiles_m = cdiv(M, GW_BM), tiles_n = cdiv(N, GW_BN);
    const int nk = K / GW_BK;
    const long long total = (long long)tiles_m * tiles_n * nk;
    const int grid = 256;
    const int per = (int)((total + grid - 1) / grid);                    // >= nk: the caller only comes here with more than 256 tiles
    const size_t lds = 3 * GW_STAGE;
    float* slabs = (float*)sk_ws;
    int* flg = (int*)((unsigned char*)sk_ws + flags_offset);
#define TEO_GWSK_LAUNCH(OF) { if (f16) TEO_GWSK_LAUNCH_F(OF, true) else TEO_GWSK_LAUNCH_F(OF, false) }
#define TEO_GWSK_LAUNCH_F(OF, FV)                                                                                                 \
    {                                                                                                                             \
        static unsigned long long attr_mask = 0;                                                                                  \
        if (int e = lds_attr_once(reinterpret_cast<const void*>(&gemm_mfma_bf16_wide_sk_kernel<OF, FV>), (int)lds, &attr_mask, "gemm_wide_sk")) return e; \
        gemm_mfma_bf16_wide_sk_kernel<OF, FV><<<grid, 512, lds, st>>>((const bf16_t*)A, (const bf16_t*)W, (const bf16_t*)bias,         \
                                                                  (const bf16_t*)res, C, M, N, K, lda, ldc, act, tiles_m, tiles_n, \
                                                                  per, slabs, flg);                                               \
    }
    if (of32) TEO_GWSK_LAUNCH(true) else TEO_GWSK_LAUNCH(false)
#undef TEO_GWSK_LAUNCH
#undef TEO_GWSK_LAUNCH_F
    note_kernel("gemm_wide_sk"); TEO_LAUNCH_CHECK("gemm_mfma_bf16_wide_sk");
    return TEO_OK;
}

int gemm_wide_launch(const void* A, const void* W, const void* bias, const void* res, void* C, int M, int N, int K, int lda, int ldc,
                     int act, bool swiglu, bool of32, bool f16, hipStream_t st) {
    const int tiles_m = cdiv(M, GW_BM), tiles_n = cdiv(N, GW_BN);
    const int nwg = tiles_m * tiles_n;
    const size_t lds = 3 * GW_STAGE;
#define TEO_GW_LAUNCH_S(SW, OF, SC) { if (f16) TEO_GW_LAUNCH_SF(SW, OF, SC, true) else TEO_GW_LAUNCH_SF(SW, OF, SC, false) }
#define TEO_GW_LAUNCH_SF(SW, OF, SC, FV)                                                                                          \
    {                                                                                                                             \
        static unsigned long long attr_mask = 0;                                                                                  \
        if (int e = lds_attr_once(reinterpret_cast<const void*>(&gemm_mfma_bf16_wide_kernel<SW, OF, SC, FV>), (int)lds, &attr_mask, "gemm_wide")) return e; \
        gemm_mfma_bf16_wide_kernel<SW, OF, SC, FV><<<nwg, 512, lds, st>>>((const bf16_t*)A, (const bf16_t*)W, (const bf16_t*)bias,     \
                                                                      (const bf16_t*)res, C, M, N, K, lda, ldc, act, tiles_m, tiles_n, tune().gemm_wide_group ? tune().gemm_wide_group : (tiles_m >= 32 ? 4 : 1)); \
    }
#define TEO_GW_LAUNCH(SW, OF)                                                                                                     \
    {                                                                                                                             \
        if (tune().gemm_wide_sched == 0) TEO_GW_LAUNCH_S(SW, OF, 0)                                                                         \
        else TEO_GW_LAUNCH_S(SW, OF, 1)                                                                                           \
    }
    if (swiglu) { if (of32) TEO_GW_LAUNCH(true, true) else TEO_GW_LAUNCH(true, false) }
    else { if (of32) TEO_GW_LAUNCH(false, true) else TEO_GW_LAUNCH(false, false) }
#undef TEO_GW_LAUNCH_S
#undef TEO_GW_LAUNCH_SF
#undef TEO_GW_LAUNCH
    note_kernel("gemm_wide"); TEO_LAUNCH_CHECK("gemm_mfma_bf16_wide");
    return TEO_OK;
}

}  // namespace teo
