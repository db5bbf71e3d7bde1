// Narrow-tile form of the 16-bit MFMA GEMM (round 5): TBM (64 or 128) x 128 x 64 workgroup tile, 4 waves (2 x 2) of (TBM / 2) x 64
// (tile and wave layout are template parameters: see the template's comment),
// operands brought in by LDS-DMA (global_load_lds_dwordx4) into a ring of NS stages, ONE barrier per K tile -- the K loop of the
// 128 x 256 kernel (gemm_wide.hip) on a tile small enough for problems that have FEW tiles (the ViT tower: M = 2056, N = 1024 ->
// 136 tiles of 128 x 128) or a SHORT K loop (K = 1024: 16 K tiles, prologue and epilogue as long as the loop).
//
// Why: those shapes ran on the register-staged 64 x 128 / 128 x 128 kernel (gemm.hip), whose K step is one serial chain
// global load -> wait -> ds_write -> barrier -> ds_read -> MFMA with a 2-deep register prefetch: 0.78 us per K step of a 64 x 128
// tile, latency-bound (tools/vit_probe.py, round 3).  Here nothing passes through registers, NS - 1 K tiles are in flight per
// workgroup, and two workgroups share a CU (72 KB of LDS at 64 x 128 x 3 stages, 64 KB at 128 x 128 x 2 stages), so one workgroup's
// prologue / epilogue runs under the other's loop.  tools/gemm_lab.hip (`narrow`, profiles/r05_gemm_lab_narrow.txt), random data, cold
// weights: fc2 (M = 2056, N = 1024, K = 4096) 36.2 us against 50.6, out_proj (K = 1024) 12.1 against 17.4.
// Same LDS image (128-byte rows, 16-byte chunk c of row r at c ^ (r & 7), applied through the DMA source address), same fragment
// reads and the same k-ascending MFMA chain per output element as every other tile family -> bit-identical results (tested).
#include "common.h"
#include "gemm_epilogue.h"
#include "ops.h"

namespace teo {

typedef __attribute__((ext_vector_type(8))) short gn_bf16x8;
typedef __attribute__((ext_vector_type(4))) float gn_f32x4;

constexpr int GN_BK = 64;

__device__ __forceinline__ int gn_xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    const int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    return base + (bid >> 3);
}

// TBM x TBN tile, WM x WN waves of (TBM / WM) x (TBN / WN); NS stages.  Shipped: 64 x 128 and 128 x 128 on 2 x 2 waves (two workgroups per
// CU).  (Measured late in round 5 and not kept: 256 x 160 on 4 x 2 waves, 3 stages -- the hand-scheduled form of that tile is gemm_quad.hip,
// 1-12 us ahead -- and 128 x 160 on 2 x 2 waves with two workgroups per CU: profiles/r05_gemm_experiments.md 6b.)
template <int TBM, int TBN, int WM, int WN, int NS, bool OUT_F32, bool F16>
__global__ __launch_bounds__(WM * WN * 64, (WM * WN == 4 ? 2 : 1)) void gemm_mfma_bf16_narrow_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W,
                                                                     const bf16_t* __restrict__ bias, const bf16_t* res, void* Cv, int M,
                                                                     int N, int K, int lda, int ldc, int act, int tiles_m, int tiles_n) {
    constexpr int NW = WM * WN;
    constexpr int A_BYTES = TBM * GN_BK * 2, STAGE = A_BYTES + TBN * GN_BK * 2;
    constexpr int PT = STAGE / 1024;              // 1-KiB DMA pieces per K tile (8 rows each): TBM / 8 of A, then TBN / 8 of W
    constexpr int NP = (PT + NW - 1) / NW;        // pieces per wave: 6 (64 x 128), 8 (128 x 128), 7 (256 x 160: 56 slots for 52 pieces -- the
                                                  // last wave brings piece 51 five times, to the same LDS bytes: +8 % L2 reads, one wait count)
    constexpr int MI = TBM / WM / 16, NI = TBN / WN / 16;      // 16-row A / W fragments per wave
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid / WN, wn = wid % WN;
    const int fr = lane & 15, fg = lane >> 4;
    const int nk = K / GN_BK;
    const int tile = gn_xcd_remap(blockIdx.x, gridDim.x);      // m fastest: the workgroups an XCD runs together share a W panel
    const int tm = tile % tiles_m, tn = tile / tiles_m;
    const int m0 = tm * TBM, n0 = tn * TBN;

    // DMA pieces of this wave: piece g = min(wid * NP + j, PT - 1) covers 8 rows (g < TBM / 8: A rows, else W rows); lane l brings row
    // (l >> 3) of the piece, logical chunk (l & 7) ^ (l >> 3), to LDS byte g * 1024 + l * 16 of the stage
    const bf16_t* src[NP];
    int gofs[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const int g = min(wid * NP + j, PT - 1);
        const int rl = lane >> 3, c = (lane & 7) ^ rl;
        if (g < TBM / 8) src[j] = A + (long long)min(m0 + g * 8 + rl, M - 1) * lda + c * 8;
        else src[j] = W + (long long)min(n0 + (g - TBM / 8) * 8 + rl, N - 1) * K + c * 8;
        gofs[j] = g * 1024;                       // (wave-uniform)
    }
#define TEO_GN_STAGE(KT, ST)                                                                                                   \
    _Pragma("unroll") for (int j = 0; j < NP; ++j)                                                                             \
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[j] + (long long)(KT) * GN_BK),   \
                                         (__attribute__((address_space(3))) void*)(smem + (ST) * STAGE + gofs[j]), 16, 0, 0);
    uint2 bv[NI];                                 // the lane's bias values: requested now, used after the K loop (gemm_epilogue.h)
    gemm_bias_load<NI>(bias, n0 + wn * (NI * 16), fg, N, bv);
    gn_f32x4 acc[NI][MI];   // [ni][mi]
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < MI; ++j) acc[i][j] = (gn_f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s0 = 0; s0 < NS - 1; ++s0)
        if (s0 < nk) { TEO_GN_STAGE(s0, s0) }
    int st = 0;
    for (int kt = 0; kt < nk; ++kt) {
        // this wave's pieces of tile kt have landed (up to NS - 2 later tiles may still fly), then everybody's; the barrier also says
        // nobody reads the stage of tile kt - 1 any more -- it takes tile kt + NS - 1
        const int ahead = min(nk - 1 - kt, NS - 2);
        if (NS >= 3 && ahead >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const int stn = st == 0 ? NS - 1 : st - 1;
        if (kt + NS - 1 < nk) { TEO_GN_STAGE(kt + NS - 1, stn) }
        __builtin_amdgcn_sched_barrier(0);
        const unsigned char* sA = smem + st * STAGE;
        const unsigned char* sB = sA + A_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            gn_bf16x8 af[MI], wf[NI];
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int rw_ = wn * (NI * 16) + i * 16 + fr;
                wf[i] = *reinterpret_cast<const gn_bf16x8*>(sB + rw_ * 128 + (((ks * 4 + fg) ^ (rw_ & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int ra_ = wm * (MI * 16) + i * 16 + fr;
                af[i] = *reinterpret_cast<const gn_bf16x8*>(sA + ra_ * 128 + (((ks * 4 + fg) ^ (ra_ & 7)) << 4));
            }
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = mfma16<F16>(wf[ni], af[mi], acc[ni][mi]);
        }
        st = st + 1 == NS ? 0 : st + 1;
    }
#undef TEO_GN_STAGE
    static_assert(NS == 2 || NS == 3, "the counted wait above allows one later tile in flight");

    gemm_epilogue<NI, MI, MI, false, OUT_F32, F16>(acc, bv, bias != nullptr, res, Cv, M, N, ldc, act, m0 + wm * (MI * 16), n0 + wn * (NI * 16), fr, fg);
}

// bm: 64 -> 64 x 128 tiles, 3 stages (72 KB: two workgroups per CU); 128 -> 128 x 128 tiles, 2 stages (64 KB: two per CU).
// No SwiGLU form (the shapes this family serves carry bias / activation / residual epilogues)
// bm = 128 with waves8: the same 128 x 128 tile on EIGHT waves (2 x 4 of 64 x 32), 3 stages (96 KB: one workgroup per CU, two waves per
// SIMD) -- round 6, for problems with at most ONE such tile per CU (LLaMA o / down at M <= 1024, the tower's fc2 / out_proj): a lone
// four-wave workgroup leaves every SIMD one wave whose MFMA chain waits for its own fragment reads (0.95 us per K step measured on a
// 128 x 128 tile alone on its CU); with two waves per SIMD one reads while the other multiplies.
int gemm_narrow_launch(const void* A, const void* W, const void* bias, const void* res, void* C, int M, int N, int K, int lda, int ldc,
                       int act, bool of32, bool f16, int bm, hipStream_t st, bool waves8) {
    // forced: the software-pipelined form of the same tiles (gemm_quad.hip gemm_pipe_launch; its automatic rule is in gemm.hip)
    if (tune().gemm_narrow_pipe == 2)
        return gemm_pipe_launch(A, W, bias, res, C, M, N, K, lda, ldc, act, of32, f16, bm, tune().gemm_pipe_bn, tune().gemm_pipe_stages, st);
    const int bn = 128;
    const int tiles_m = cdiv(M, bm), tiles_n = cdiv(N, bn);
    const int nwg = tiles_m * tiles_n;
#define TEO_GN_LAUNCH_T(TBM, TBN, WM, WN, NS, OF, FV)                                                                             \
    {                                                                                                                             \
        constexpr size_t lds = (size_t)(NS) * ((TBM) * GN_BK * 2 + (TBN) * GN_BK * 2);                                            \
        static unsigned long long attr_mask = 0;                                                                                  \
        if (int e = lds_attr_once(reinterpret_cast<const void*>(&gemm_mfma_bf16_narrow_kernel<TBM, TBN, WM, WN, NS, OF, FV>), (int)lds, &attr_mask, "gemm_narrow")) return e; \
        gemm_mfma_bf16_narrow_kernel<TBM, TBN, WM, WN, NS, OF, FV><<<nwg, (WM) * (WN) * 64, lds, st>>>((const bf16_t*)A, (const bf16_t*)W, (const bf16_t*)bias,  \
                                                                          (const bf16_t*)res, C, M, N, K, lda, ldc, act, tiles_m, tiles_n); \
    }
#define TEO_GN_LAUNCH_F(TBM, TBN, WM, WN, NS, OF) { if (f16) TEO_GN_LAUNCH_T(TBM, TBN, WM, WN, NS, OF, true) else TEO_GN_LAUNCH_T(TBM, TBN, WM, WN, NS, OF, false) }
#define TEO_GN_LAUNCH(TBM, TBN, WM, WN, NS) { if (of32) TEO_GN_LAUNCH_F(TBM, TBN, WM, WN, NS, true) else TEO_GN_LAUNCH_F(TBM, TBN, WM, WN, NS, false) }
    if (bm == 64) TEO_GN_LAUNCH(64, 128, 2, 2, 3) else if (waves8) TEO_GN_LAUNCH(128, 128, 2, 4, 3) else TEO_GN_LAUNCH(128, 128, 2, 2, 2)
#undef TEO_GN_LAUNCH
#undef TEO_GN_LAUNCH_F
#undef TEO_GN_LAUNCH_T
    note_kernel(bm == 64 ? "gemm_narrow_64" : (waves8 ? "gemm_narrow_128w8" : "gemm_narrow_128"));
    TEO_LAUNCH_CHECK("gemm_mfma_bf16_narrow");
    return TEO_OK;
}

}  // namespace teo
