// Hand-scheduled form of the 16-bit MFMA GEMM on a 256 x 160 x 64 workgroup tile (round 5, late): operands by LDS-DMA into a ring of 3 stages
// (3 x 52 KB), accumulators pinned to AGPRs, the K loop's issue order written out.  Two wave layouts of the same loop:
//   WM = 4 (default): EIGHT waves (4 x 2) of 64 x 80, two per SIMD -- 80 accumulator registers, 40 MFMAs + 7 DMA pieces + 18 reads per K tile and wave;
//   WM = 2: FOUR waves (2 x 2) of 128 x 80, ONE per SIMD with the whole 512-register file -- 160 accumulator registers, 80 + 13 + 26.
//
// Why this shape: M = 2056 .. 2304 rows against N = 4096 (LLaMA o / down at config C3, the tower's fc1) is 272 tiles of 128 x 256 -- one
// round of the 256 CUs plus a ragged sixteenth -- and 144 of 256 x 256, which leaves 112 CUs idle.  256 x 160 tiles make it 9 x 26 = 234
// workgroups: ONE round, 91 % of the CUs busy, and the tile is still large enough to amortise its operand traffic.  It is the tile the
// vendor library picks for the same shapes (profiles/r05_library_yardstick.txt: MT256x160 / MT160x256, 256 threads); measured here
// (tools/gemm_lab.hip `w4n`, profiles/r05_gemm_lab_w4n.txt, four waves, no epilogue, cold weights): o 66 us against 78 (128 x 256 stream-K)
// and the library's 71, down 159 against 174 / 169, fc1 24 against 31 / 22.  A 2-stage ring loses 28 % on down (K = 11008 from HBM): the
// third stage is the point.  With the real epilogues (tools/vit_gemm_probe.py): o 79.7 -> 71.3 (8 waves) / 72.4 (4), down 178.6 -> 170.7 / 170.0,
// fc1 + GELU 39.5 -> 34.2 / 51.1 -- one wave per SIMD issues the 160 erf evaluations of a lane alone, two alternate; hence the default.
//
// K tile = two phases, ONE barrier:
//     A: MFMA k-half 0 (fragments X) || ds_read k-half 1 -> Y
//        -- lgkmcnt(0): my reads of this stage are done; vmcnt: my pieces of tile kt + 1 have landed (tile kt + 2 may fly) -- s_barrier --
//     B: MFMA k-half 1 (Y) || DMA of tile kt + 3 into the stage just released || ds_read k-half 0 of tile kt + 1 -> X
// The MFMAs are inline asm with "+a" accumulators: with the builtin and 160+ accumulators the register allocator moves them between AGPRs
// and VGPRs (~1000 v_accvgpr moves per K tile at 256 x 256); as volatile asm they also keep SOURCE order against the LDS reads and the DMA
// builtins, so the interleave below (4 MFMAs, then the group's DMA pieces and two reads) is the issue order.  hipcc still places the
// s_waitcnt for the fragment registers itself (it sees the asm operands); the explicit lgkmcnt(0) at the end of a K tile sits where nothing
// is outstanding.  Same LDS image (128-byte rows, 16-byte chunk c of row r at c ^ (r & 7), applied through the DMA source address), same
// fragment reads and the same k-ascending MFMA chain per output element as every other tile family -> bit-identical results (tested).
#include "common.h"
#include "gemm_epilogue.h"
#include "ops.h"

namespace teo {

constexpr int GQ_BM = 256, GQ_BK = 64;
// s_waitcnt immediate of gfx9: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt[5:4] << 14 -- "vmcnt(N) lgkmcnt(0)" with expcnt left alone
#define GQ_WAIT_VM_LGKM0(N) ((((N) & 15) | (7 << 4) | ((((N) >> 4) & 3) << 14)))

__device__ __forceinline__ int gq_xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    const int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    return base + (bid >> 3);
}

template <bool F16> __device__ __forceinline__ void gq_mfma(teo_f32x4& c, const teo_h16x8& a, const teo_h16x8& b) {
    if constexpr (F16) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}

// TBM: rows of the workgroup tile (256; 64 / 128: the few-tile shapes' form, see gemm_pipe_launch below).
// WM: waves along M (2: four waves of (TBM / 2) x 16 NI, one per SIMD; 4: EIGHT waves of (TBM / 4) x 16 NI, two per SIMD)
// OCC: waves per SIMD the register budget is cut for (WM / 2 for one workgroup per CU; 2 with WM = 2: two four-wave workgroups per CU)
template <int TBM, int NI, int NS, int WM, int OCC, bool OUT_F32, bool F16, bool SWIGLU = false>
__global__ __launch_bounds__(WM * 128) __attribute__((amdgpu_waves_per_eu(WM / 2, OCC))) void gemm_mfma_bf16_quad_kernel(
    const bf16_t* __restrict__ A, const bf16_t* __restrict__ W, const bf16_t* __restrict__ bias, const bf16_t* res, void* Cv, int M, int N, int K,
    int lda, int ldc, int act, int tiles_m, int tiles_n) {
    constexpr int NW = WM * 2, MI = TBM / 16 / WM;     // waves; 16-row A fragments per wave (8 or 4 at 256 rows)
    constexpr int PA = TBM / 8;                  // DMA pieces of A per stage
    constexpr int TN = 32 * NI, STG = (TBM + TN) * 128, PT = PA + 4 * NI, NPW = (PT + NW - 1) / NW;
    static_assert(NS >= 2 && (NI * MI) % 4 == 0 && MI >= 1 && MI <= 8 && NI <= 8, "MFMA groups of four");
    constexpr int G = NI * MI / 4;               // MFMA groups of 4 per phase (NI x MI fragment pairs)
    constexpr int R = MI + NI;                   // fragment reads per phase
    // 256-row tiles: 4 MFMAs, then two reads, group after group (LDS and MFMA busy side by side through the phase).  The small tiles have 2 / 4
    // groups per phase: their reads go FIRST and into the first half of the groups, so that every read has at least four MFMAs behind it
    // before the phase's wait (with "MFMAs, then reads" the last group's reads would be issued right in front of the wait)
    constexpr bool RF = TBM != GQ_BM;
    constexpr int RG = RF ? (G + 1) / 2 : G;     // groups that issue reads
    constexpr int RPG = (R + RG - 1) / RG;       // fragment reads per such group (2 at 256 x 160; 6 at 64 x 128, 4 at 128 x 128)
    constexpr int PPG = (NPW + G - 1) / G;       // DMA pieces per group
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 1, wn = wid & 1;
    const int fr = lane & 15, fg = lane >> 4;
    const int nk = K / GQ_BK;
    const int tile = gq_xcd_remap(blockIdx.x, gridDim.x);      // m fastest: the workgroups an XCD runs together share a W panel
    const int tm = tile % tiles_m, tn = tile / tiles_m;
    const int m0 = tm * TBM, n0 = tn * TN;

    // DMA pieces of a stage: TBM / 8 of A (8 rows each), then 4 NI of W; wave w brings pieces w NPW .. w NPW + NPW - 1.  Lane l brings row
    // (l >> 3) of the piece, logical chunk (l & 7) ^ (l >> 3), to LDS byte piece * 1024 + l * 16 of the stage
    const bf16_t* src[NPW];
#pragma unroll
    for (int j = 0; j < NPW; ++j) {
        const int p = min(wid * NPW + j, PT - 1);   // (eight waves: 56 slots for 52 pieces -- the last wave brings piece 51 again, same bytes)
        const int rl = lane >> 3, c = (lane & 7) ^ rl;
        src[j] = p < PA ? A + (long long)min(m0 + p * 8 + rl, M - 1) * lda + c * 8 : W + (long long)min(n0 + (p - PA) * 8 + rl, N - 1) * K + c * 8;
    }
#define TEO_GQ_PIECE(KT, ST, J)                                                                                                  \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[J] + (long long)(KT) * GQ_BK),         \
                                     (__attribute__((address_space(3))) void*)(smem + (ST) * STG + min(wid * NPW + (J), PT - 1) * 1024), 16, 0, 0);
    uint2 bv[NI];                                 // the lane's bias values: requested now, used after the K loop (gemm_epilogue.h)
    gemm_bias_load<NI>(bias, n0 + wn * (NI * 16), fg, N, bv);
    teo_f32x4 acc[NI][MI];   // [ni][mi]
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < MI; ++j) acc[i][j] = (teo_f32x4){0.f, 0.f, 0.f, 0.f};
    teo_h16x8 xf[16], yf[16];                    // [0 .. NI - 1]: W fragments, [8 .. 15]: A fragments
    const int rw0 = wn * (NI * 16) + fr, ra0 = wm * (MI * 16) + fr;
    // read RI of a phase: RI < MI -> A fragment RI, else W fragment RI - MI (the first MFMA groups need every A fragment, group g only W[4 g / MI])
#define TEO_GQ_READ(F, ST, KS, RI)                                                                                \
    {                                                                                                             \
        const int isw_ = (RI) >= MI;                                                                              \
        const int r_ = (isw_ ? rw0 + ((RI) - MI) * 16 : ra0 + (RI) * 16);                                         \
        F[isw_ ? (RI) - MI : 8 + (RI)] =                                                                          \
            *reinterpret_cast<const teo_h16x8*>(smem + (ST) * STG + (isw_ ? TBM * 128 : 0) + r_ * 128 + ((((KS) * 4 + fg) ^ (r_ & 7)) << 4)); \
    }
#define TEO_GQ_MFMA4(F, GI)                                                                                       \
    _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) {                                                            \
        const int p_ = (GI) * 4 + q_;                                                                             \
        const int ni_ = p_ / MI, mi_ = p_ % MI;                                                                   \
        gq_mfma<F16>(acc[ni_][mi_], F[ni_], F[8 + mi_]);                                                          \
    }
#pragma unroll
    for (int t = 0; t < NS; ++t)
        if (t < nk) {
#pragma unroll
            for (int j = 0; j < NPW; ++j) { TEO_GQ_PIECE(t, t, j) }
        }
    // (the bias request above sits in front of the pieces in the in-order vmcnt queue: every count below covers it)
    if (nk >= NS) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 1) * NPW) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int r = 0; r < R; ++r) { TEO_GQ_READ(xf, 0, 0, r) }
    __builtin_amdgcn_s_waitcnt(0xC07F);         // lgkmcnt(0), as a builtin: hipcc's wait pass does not read inline asm, and a wait it cannot see
                                                // in the loop's preheader makes it put its own lgkmcnt(0) in front of the loop's first MFMA
    int st = 0;
    // MORE: a tile kt + NS exists (its pieces go into the stage this tile releases); MORE1: a tile kt + 1 exists
#define TEO_GQ_KTILE(MORE, MORE1)                                                                                 \
    {                                                                                                             \
        const int stn = st + 1 == NS ? 0 : st + 1;                                                                \
        _Pragma("unroll") for (int g = 0; g < G; ++g) {                                                           \
            if (RF) {                                                                                             \
                _Pragma("unroll") for (int t = 0; t < RPG; ++t)                                                   \
                    if (g * RPG + t < R) { TEO_GQ_READ(yf, st, 1, g * RPG + t) }                                  \
            }                                                                                                     \
            TEO_GQ_MFMA4(xf, g)                                                                                   \
            if (!RF) {                                                                                            \
                _Pragma("unroll") for (int t = 0; t < RPG; ++t)                                                   \
                    if (g * RPG + t < R) { TEO_GQ_READ(yf, st, 1, g * RPG + t) }                                  \
            }                                                                                                     \
        }                                                                                                         \
        /* builtin waits (hipcc's wait pass cannot read an asm one and would add its own lgkmcnt(0) behind the next reads) */ \
        if (MORE) __builtin_amdgcn_s_waitcnt(GQ_WAIT_VM_LGKM0((NS - 2) * NPW));                                   \
        else __builtin_amdgcn_s_waitcnt(GQ_WAIT_VM_LGKM0(0));                                                     \
        asm volatile("" ::: "memory");                                                                            \
        __builtin_amdgcn_s_barrier();                                                                             \
        _Pragma("unroll") for (int g = 0; g < G; ++g) {                                                           \
            if (RF && MORE1) {                                                                                    \
                _Pragma("unroll") for (int t = 0; t < RPG; ++t)                                                   \
                    if (g * RPG + t < R) { TEO_GQ_READ(xf, stn, 0, g * RPG + t) }                                 \
            }                                                                                                     \
            TEO_GQ_MFMA4(yf, g)                                                                                   \
            _Pragma("unroll") for (int t = 0; t < PPG; ++t)                                                       \
                if (g * PPG + t < NPW && MORE) { TEO_GQ_PIECE(kt + NS, st, g * PPG + t) }                         \
            if (!RF && MORE1) {                                                                                   \
                _Pragma("unroll") for (int t = 0; t < RPG; ++t)                                                   \
                    if (g * RPG + t < R) { TEO_GQ_READ(xf, stn, 0, g * RPG + t) }                                 \
            }                                                                                                     \
        }                                                                                                         \
        __builtin_amdgcn_s_waitcnt(0xC07F);   /* lgkmcnt(0): landed long ago; as a builtin hipcc's wait pass sees it */ \
        st = stn;                                                                                                 \
    }
    int kt = 0;
    for (; kt + NS < nk; ++kt) TEO_GQ_KTILE(true, true)
    for (; kt + 1 < nk; ++kt) TEO_GQ_KTILE(false, true)
    TEO_GQ_KTILE(false, false)
#undef TEO_GQ_KTILE
#undef TEO_GQ_PIECE
#undef TEO_GQ_READ
#undef TEO_GQ_MFMA4
    // the asm MFMAs are invisible to hipcc's hazard recognizer: cover the last results' write-back before the epilogue reads the AGPRs
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");

    // (measured and not kept: the same epilogue through LDS -- 64 rows x 80 columns of f32 per wave parked in the free ring, read back as whole
    // rows by a rolled loop, 8-byte stores covering a row's 160 contiguous bytes: bit-identical, o 73.1 -> 89.8 us, down 175.9 -> 187.2, fc1 + GELU
    // 51.2 -> 52.2.  The activation's cost here is VALU issue with ONE wave per SIMD, not code size: 160 erf per lane, no second wave to
    // alternate with -- which is why the eight-wave layout is the default.)
    static_assert(!SWIGLU || NI % 2 == 0, "(gate 16 | up 16) column blocks per wave");
    gemm_epilogue<NI, MI, (MI % 4 == 0 ? 4 : MI), SWIGLU, OUT_F32, F16>(acc, bv, bias != nullptr, res, Cv, M, N, ldc, act, m0 + wm * (MI * 16), n0 + wn * (NI * 16), fr, fg);
}

// 256 x 160 tiles, 3 stages (156 KB: one workgroup per CU), eight or four waves (tune().gemm_quad_waves).  No SwiGLU form (gate / up run
// several rounds: the 256 x 256 hybrid's shapes)
int gemm_quad_launch(const void* A, const void* W, const void* bias, const void* res, void* C, int M, int N, int K, int lda, int ldc,
                     int act, bool of32, bool f16, hipStream_t st) {
    const bool eight = tune().gemm_quad_waves != 4;          // default: eight waves (two per SIMD); 4: one wave per SIMD, 128 x 80 each
    constexpr int NI = 5, NS = 3, TN = 32 * NI;
    const int tiles_m = cdiv(M, GQ_BM), tiles_n = cdiv(N, TN);
    const int nwg = tiles_m * tiles_n;
    constexpr size_t lds = (size_t)NS * (GQ_BM + TN) * 128;
#define TEO_GQ_LAUNCH_W(OF, FV, WMV)                                                                                              \
    {                                                                                                                             \
        static unsigned long long attr_mask = 0;                                                                                  \
        if (int e = lds_attr_once(reinterpret_cast<const void*>(&gemm_mfma_bf16_quad_kernel<GQ_BM, NI, NS, WMV, (WMV) / 2, OF, FV>), (int)lds, &attr_mask, "gemm_quad")) return e; \
        gemm_mfma_bf16_quad_kernel<GQ_BM, NI, NS, WMV, (WMV) / 2, OF, FV><<<nwg, (WMV) * 128, lds, st>>>((const bf16_t*)A, (const bf16_t*)W, (const bf16_t*)bias, \
                                                                          (const bf16_t*)res, C, M, N, K, lda, ldc, act, tiles_m, tiles_n); \
    }
#define TEO_GQ_LAUNCH_T(OF, FV) { if (eight) TEO_GQ_LAUNCH_W(OF, FV, 4) else TEO_GQ_LAUNCH_W(OF, FV, 2) }
#define TEO_GQ_LAUNCH_F(OF) { if (f16) TEO_GQ_LAUNCH_T(OF, true) else TEO_GQ_LAUNCH_T(OF, false) }
    if (of32) TEO_GQ_LAUNCH_F(true) else TEO_GQ_LAUNCH_F(false)
#undef TEO_GQ_LAUNCH_F
#undef TEO_GQ_LAUNCH_T
#undef TEO_GQ_LAUNCH_W
    note_kernel(eight ? "gemm_quad_160" : "gemm_quad_160_w4");
    TEO_LAUNCH_CHECK("gemm_mfma_bf16_quad");
    return TEO_OK;
}

// The same K loop on the few-tile shapes' tiles (round 6): 64 x 64 and 64 x 128 (four waves of 32 x 32 / 32 x 64), 128 x 96 and 128 x 128 (four waves
// of 64 x 48 / 64 x 64), rings of 3 or 4 stages; the even-NI tiles also with the SwiGLU epilogue (gate/up at M <= 128).  Why: where a launch has at most one or two workgroups per CU (LLaMA o / down below ~1000 rows, every tower
// GEMM at T <= 4, fc2 / out_proj at every T) a SIMD holds ONE wave, and gemm_narrow.hip's loop as hipcc schedules it is
// ds_read x 4 -> s_waitcnt lgkmcnt(0) -> 4 MFMAs, four times per K tile: the whole LDS latency is exposed four times (0.5 us per K tile of a
// 64 x 128 tile alone on its CU; its 16 MFMAs are 0.1 us).  Here the reads of k-half 1 fly under the MFMAs of k-half 0 and the next tile's under
// k-half 1's.  And the tile is chosen so that the launch has about one workgroup per CU: a CU's LDS-DMA stream stops scaling at 50-70 GB/s
// (a ring of 6 stages is no faster than one of 4), so a problem with 64 tiles of 64 x 128 runs on 64 CUs' worth of memory parallelism -- as
// 128 tiles of 64 x 64 it is 30 % faster (LLaMA o at M <= 256: 33.5 -> 29 (pipelined) -> 21 us; the tower's fc2 at T = 2: 31 -> 22 -> 15).
// Same LDS image, fragment reads and k-ascending chain: bit-identical to every other family (tests/test_gemm_fuzz_gpu.py).
// bm 64: tn 64 (ring of 4 = 64 KB: two per CU) or 128 (ring of 3 = 72 KB: two per CU; ring of 4 = 96 KB: one per CU, for launches of at most
// one workgroup per CU with a long K loop); bm 128: tn 96 (ring of 3 = 84 KB or 4 = 112 KB) or 128 (ring of 3 = 96 KB): one per CU
int gemm_pipe_launch(const void* A, const void* W, const void* bias, const void* res, void* C, int M, int N, int K, int lda, int ldc,
                     int act, bool of32, bool f16, int bm, int tn, int ns, hipStream_t st, bool swiglu) {
    if (bm != 128) bm = 64;
    if (swiglu && tn == 96) tn = 128;                    // the SwiGLU epilogue pairs 16-column blocks: an even number of them per wave
    if (bm == 128) tn = tn == 96 ? 96 : 128;
    else if (tn != 64) tn = 128;
    ns = (bm == 128) ? (tn == 96 && ns == 4 ? 4 : 3) : (tn == 64 ? 4 : (ns == 4 ? 4 : 3));
    const int tiles_m = cdiv(M, bm), tiles_n = cdiv(N, tn);
    const int nwg = tiles_m * tiles_n;
#define TEO_GP_LAUNCH_S(OF, FV, BMV, NIV, NSV, OCCV, SWV)                                                                         \
    {                                                                                                                             \
        constexpr size_t lds = (size_t)(NSV) * ((BMV) + 32 * (NIV)) * 128;                                                        \
        static unsigned long long attr_mask = 0;                                                                                  \
        if (int e = lds_attr_once(reinterpret_cast<const void*>(&gemm_mfma_bf16_quad_kernel<BMV, NIV, NSV, 2, OCCV, OF, FV, SWV>), (int)lds, &attr_mask, "gemm_pipe")) return e; \
        gemm_mfma_bf16_quad_kernel<BMV, NIV, NSV, 2, OCCV, OF, FV, SWV><<<nwg, 256, lds, st>>>((const bf16_t*)A, (const bf16_t*)W, (const bf16_t*)bias, \
                                                                          (const bf16_t*)res, C, M, N, K, lda, ldc, act, tiles_m, tiles_n); \
    }
#define TEO_GP_LAUNCH_W(OF, FV, BMV, NIV, NSV, OCCV) { if (swiglu) TEO_GP_LAUNCH_S(OF, FV, BMV, NIV, NSV, OCCV, true) else TEO_GP_LAUNCH_S(OF, FV, BMV, NIV, NSV, OCCV, false) }
#define TEO_GP_LAUNCH_T(OF, FV)                                                                                                   \
    {                                                                                                                             \
        if (bm == 128 && tn == 96) { if (ns == 4) TEO_GP_LAUNCH_S(OF, FV, 128, 3, 4, 1, false) else TEO_GP_LAUNCH_S(OF, FV, 128, 3, 3, 1, false) } \
        else if (bm == 128) TEO_GP_LAUNCH_W(OF, FV, 128, 4, 3, 1)                                                                 \
        else if (tn == 64) TEO_GP_LAUNCH_W(OF, FV, 64, 2, 4, 2)                                                                   \
        else if (ns == 4) TEO_GP_LAUNCH_W(OF, FV, 64, 4, 4, 2)                                                                    \
        else TEO_GP_LAUNCH_W(OF, FV, 64, 4, 3, 2)                                                                                 \
    }
#define TEO_GP_LAUNCH_F(OF) { if (f16) TEO_GP_LAUNCH_T(OF, true) else TEO_GP_LAUNCH_T(OF, false) }
    if (of32) TEO_GP_LAUNCH_F(true) else TEO_GP_LAUNCH_F(false)
#undef TEO_GP_LAUNCH_F
#undef TEO_GP_LAUNCH_T
#undef TEO_GP_LAUNCH_W
#undef TEO_GP_LAUNCH_S
    note_kernel(bm == 128 ? (tn == 96 ? "gemm_pipe_128x96" : "gemm_pipe_128") : (tn == 64 ? "gemm_pipe_64x64" : (ns == 4 ? "gemm_pipe_64_r4" : "gemm_pipe_64")));
    TEO_LAUNCH_CHECK("gemm_mfma_bf16_pipe");
    return TEO_OK;
}

}  // namespace teo
