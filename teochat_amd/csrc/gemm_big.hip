// 256 (M) x 256 (N) x 64 form of the bf16 MFMA GEMM: 8 waves (2 x 4) of 128 x 64, operands by LDS-DMA into a two-stage ring
// (2 x 64 KB).  Against the 128 x 256 kernel (gemm_wide.hip) a workgroup moves 33 % fewer bytes per FLOP out of L2 (64 KB per
// 8.4 MFLOP instead of 48 KB per 4.2) and a wave reads 25 % fewer fragment bytes per MFMA (12 ds_read_b128 per 32 MFMAs instead of
// 8 per 16) -- the two things the ablation of the wide kernel found as long as its MFMA work.  Same LDS image (128-byte rows,
// 16-byte chunk c of row r at c ^ (r & 7), applied through the DMA source address), same MFMA chain per output element
// (k ascending) -> bit-identical results.  K loop as gw_ktile: skewed DMA issue between the two waves of a SIMD, second-half
// fragments carried over the next barrier (245 VGPRs, 2 waves per SIMD).  The price is tile quantisation: 9 row tiles at
// M = 2168 (6 % padding) and 256-column panels; the dispatch (gemm.hip) only comes here when the rounds model says it pays.
// Measured (tools/bench_kernels.py gemm_big): 1.45-1.7 us per K tile of a workgroup (MFMA alone would be 0.94); qkv at M = 2168
// 195 us vs 223, gate/up at M = 4208 608 vs 668, at M = 17344 2434 vs 2682, 4096^3 108 vs 118; gate/up at M = 2168 (774 tiles =
// 3.02 rounds) loses, 371 vs 340.  Ablation (qkv): MFMA + barriers alone 159 us, DMA alone 140 (one tile in flight: a two-stage
// ring is latency-bound at ~1.1 us per 64 KB), fragment reads alone 81.  A second barrier per K tile that frees the stage early
// (1.5 tiles of DMA lead) makes the DMA alone faster (106 us) and the kernel slower (246 us): not kept.
#include "common.h"
#include "gemm_epilogue.h"
#include "ops.h"

namespace teo {

typedef __attribute__((ext_vector_type(8))) short gb_bf16x8;
typedef __attribute__((ext_vector_type(4))) float gb_f32x4;

constexpr int GB_BM = 256, GB_BN = 256, GB_BK = 64;
constexpr int GB_A_BYTES = GB_BM * GB_BK * 2;            // 32 KiB
constexpr int GB_STAGE = GB_A_BYTES + GB_BN * GB_BK * 2; // 64 KiB
constexpr int GB_SLAB_FLOATS = GB_BM * GB_BN;          // 256 KB of fp32 accumulators per workgroup
constexpr int GB_PIECES = 8;                             // 1-KiB DMA pieces per wave per K tile (waves 0-3: A, waves 4-7: W)
constexpr int GB_RAG_STAGE = (128 + 512) * GB_BK * 2;    // 80 KiB: the 128 x 512 tile over a ragged last row block (round 6)
constexpr int GB_RAG_PIECES = 11;                        // its W pieces per wave (waves 2-7: 66 slots for 64 pieces)

__device__ __forceinline__ int gb_xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    const int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    return base + (bid >> 3);
}

// HYBRID = false: one workgroup per tile.  HYBRID = true: a persistent grid of 256 workgroups (one per CU); the first `dp_rounds`
// x 256 tiles are taken whole, one per workgroup per round (all workgroups of an XCD at the same k at the same time: the L2 sharing
// of the plain kernel), the remaining r tiles (256 <= r < 512) are cut stream-K style into 256 equal contiguous (tile, k) ranges
// with the sequential hand-off of gemm_mfma_bf16_sk_kernel: a range that ends inside a tile runs that part FIRST and hands its fp32
// accumulators (256 KB slab, 16-byte sc1 stores + flag) to the next range's owner, which continues from them LAST in its own
// timeline -- k-order per output element unchanged, results bit-identical, no ragged last round.
template <bool SWIGLU, bool OUT_F32, bool HYBRID, bool F16 = false>
__global__ __launch_bounds__(512) void gemm_mfma_bf16_big_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W,
                                                               const bf16_t* __restrict__ bias, const bf16_t* res, void* Cv, int M,
                                                               int N, int K, int lda, int ldc, int act, int tiles_m, int tiles_n,
                                                               int group, int dp_rounds, int per, float* slabs, int* flags, int cohort,
                                                               int rag_tiles) {
    // rag_tiles > 0 (round 6): `tiles_m` counts the FULL 256-row tiles only; the last M % 256 <= 128 rows are covered by rag_tiles =
    // tiles_n / 2 tiles of 128 x 512 appended to the tile list (tile index >= tiles_m * tiles_n): the same eight waves of 128 x 64, laid out
    // 1 x 8 instead of 2 x 4, over a stage of 128 A rows + 512 W rows (80 KB) -- the same MFMA work, fragment reads and K loop per tile
    // as a full tile, so every tile still costs the same (the stream-K partition stays unweighted), but the row padding is gone: L = 256 T +
    // (prompt - T) always leaves 112 .. 127 rows in the last tile, which a ninth 256-row tile covered with half its MFMA work on padding
    // (M = 2168: qkv 432 -> 408 tiles, gate/up 774 -> 731).  Same k-ascending MFMA chain per output element: bit-identical (tested).
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fg = lane >> 4;
    const int nk = K / GB_BK;
    const int q = gb_xcd_remap(blockIdx.x, gridDim.x);
    // segments of this workgroup: [tail piece] [dp_rounds whole tiles] [whole tiles of its stream-K range] [head piece]
    // The stream-K ranges come in two arrangements.  cohort == 0 (round 2): workgroup q owns the contiguous (tile, k) range
    // [q per, (q + 1) per) -- neighbours sit at different k of different tiles, nothing of the stream-K part is shared through L2
    // (its K tiles take up to twice as long as the data-parallel ones once W comes from HBM).  cohort == C (round 5): the
    // remaining tiles form columns of C tiles (position p = column, tile = p C + slot), workgroup q = (chain q / C, slot q % C);
    // the C workgroups of a chain link -- neighbours on one XCD after the remap -- all own the SAME (column, k) range and walk it
    // in step, so they share W panels and A row tiles through their L2 exactly as in a data-parallel round; the hand-off goes
    // to the same slot of the next link (slab q -> q + C).  Same sequential k-order per tile either way.
    int t_first = 0, k_first = 0, t_last = 0, k_end = nk, has_head = 0, has_tail = 0, n_full = 0, t_full0 = 0, nseg = 1;
    const int t_reg = tiles_m * tiles_n, t_all = t_reg + rag_tiles;
    const int sk_tile0 = dp_rounds * 256;
    const int sk_tiles = t_all - sk_tile0;
    const int slot = cohort ? (q & (cohort - 1)) : 0, pstride = cohort ? cohort : 1;          // position p -> tile sk_tile0 + p * pstride + slot
    const int pred = q - pstride;
    if (HYBRID) {
        const long long total = cohort ? (long long)((sk_tiles + cohort - 1) / cohort) * nk : (long long)sk_tiles * nk;
        const long long it0 = (long long)(cohort ? q / cohort : q) * per, it1 = min(it0 + per, total);
        if (it0 < total) {
            t_first = (int)(it0 / nk); k_first = (int)(it0 % nk);
            t_last = (int)((it1 - 1) / nk); k_end = (int)(it1 - (long long)t_last * nk);
            has_head = k_first != 0; has_tail = k_end != nk;                 // per >= nk: a tile has at most two owners
            t_full0 = t_first + has_head; n_full = (t_last + 1 - has_tail) - t_full0;
        }
        nseg = has_tail + dp_rounds + n_full + has_head;
    }
    gb_f32x4 acc[4][8];   // [ni][mi]
    for (int sgi = 0; sgi < nseg; ++sgi) {
    bool is_tail = false, is_head = false;
    int tile, kb = 0, ke = nk;
    if (HYBRID) {
        is_tail = has_tail && sgi == 0;
        is_head = has_head && sgi == nseg - 1;
        const int j = sgi - has_tail;                                        // index among the whole tiles
        int pos = -1;
        if (is_tail) { pos = t_last; ke = k_end; }
        else if (is_head) { pos = t_first; kb = k_first; }
        else if (j >= dp_rounds) pos = t_full0 + (j - dp_rounds);
        tile = pos < 0 ? j * 256 + q : sk_tile0 + pos * pstride + slot;
        if (tile >= t_all) continue;                                         // a column's unused slots (both owners skip alike: no hand-off)
    } else {
        tile = q;
    }
    const bool rag = tile >= t_reg;                                          // (workgroup-uniform) a 128 x 512 tile over the ragged last rows
    // tiles walk `group` N panels at a time, N fastest (see gemm_wide.hip): with many row tiles the 32 workgroups of an XCD would
    // otherwise share one W panel and stream all of A (sq8192: every XCD reads the whole A once per round)
    const int gsz = group * tiles_m, sup = rag ? 0 : tile / max(gsz, 1), rem = tile - sup * gsz;
    const int gn = min(group, tiles_n - sup * group);
    const int tm = rem / gn, tn = sup * group + rem % gn;
    const int m0 = rag ? tiles_m * GB_BM : tm * GB_BM, n0 = rag ? (tile - t_reg) * (2 * GB_BN) : tn * GB_BN;
    const int wm = rag ? 0 : wid >> 2, wn = rag ? wid : wid & 3;              // the wave's 128 x 64 block of the tile
    const int a_bytes = rag ? GB_A_BYTES / 2 : GB_A_BYTES, stage_bytes = rag ? GB_RAG_STAGE : GB_STAGE;

    // DMA, full tile: wave w < 4 brings A rows 64 w .. 64 w + 63 (8 pieces of 8 rows), wave w >= 4 W rows 64 (w - 4) ..; 128 x 512 tile:
    // waves 0 / 1 the 128 A rows, waves 2 .. 7 eleven pieces each of the 64 W pieces (66 slots: the last wave brings piece 63 three times, same
    // bytes).  Lane l carries row (l >> 3) of a piece, logical chunk (l & 7) ^ (l >> 3).  32-bit element offsets from one wave-uniform base.
    const bool isA = rag ? wid < 2 : wid < 4;
    const bf16_t* base = isA ? A : W;
    const int ld = isA ? lda : K;
    const int rmax = (isA ? M : N) - 1;
    const int np = (rag && !isA) ? GB_RAG_PIECES : GB_PIECES;
    const int p0 = isA ? wid * 8 : (rag ? (wid - 2) * GB_RAG_PIECES : (wid - 4) * 8), pmax = isA ? (rag ? 15 : 31) : (rag ? 63 : 31);
    unsigned off[GB_RAG_PIECES];
#pragma unroll
    for (int j = 0; j < GB_RAG_PIECES; ++j) {
        const int rl = lane >> 3, c = (lane & 7) ^ rl;
        off[j] = (unsigned)min((isA ? m0 : n0) + min(p0 + j, pmax) * 8 + rl, rmax) * (unsigned)ld + c * 8;
    }
    const int lds_base = isA ? 0 : a_bytes;
#define TEO_GB_STAGE(KT, ST)                                                                                                 \
    _Pragma("unroll") for (int j = 0; j < GB_RAG_PIECES; ++j)                                                                \
        if (j < np)                                                                                                          \
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + off[j] + (unsigned)(KT) * GB_BK), \
                                             (__attribute__((address_space(3))) void*)(smem + (ST) * stage_bytes + lds_base + min(p0 + j, pmax) * 1024), 16, 0, 0);

    if (HYBRID && is_head) {
        if (tid == 0) {
            int spins = 0;
            while (__hip_atomic_load(flags + pred, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
                __builtin_amdgcn_s_sleep(8);
                if (++spins > (1 << 24)) { __hip_atomic_store(flags + GEMM_SK_ERR_SLOT, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }   // never reached (the producer wrote its slab first thing); a miss is STICKY: teo_gemm_workspace_status
            }
            __hip_atomic_store(flags + pred, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // re-armed for the next launch
        }
        __builtin_amdgcn_s_barrier();
        const auto sl = __builtin_amdgcn_make_buffer_rsrc(slabs + (size_t)pred * GB_SLAB_FLOATS, 0, GB_SLAB_FLOATS * 4, 0x00020000);
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int mi = 0; mi < 8; ++mi)
                acc[ni][mi] = __builtin_bit_cast(gb_f32x4, __builtin_amdgcn_raw_buffer_load_b128(sl, ((ni * 8 + mi) * 512 + tid) * 16, 0, /*sc1*/ 16));
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = (gb_f32x4){0.f, 0.f, 0.f, 0.f};
    }

    // (a head segment's 32 slab loads are older than the DMA pieces below: vmcnt retires in order, the waits keep their meaning)
    TEO_GB_STAGE(kb, 0)
    const bool late = wid < 4;                            // skewed DMA issue between the two waves of a SIMD (see gw_ktile)
    gb_bf16x8 af[8], wf[4], caf[8], cwf[4];               // first-half fragments of this K tile; second half carried over the next barrier
#define TEO_GB_READ(AF, WF, KS)                                                                                   \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                               \
        const int rw_ = wn * 64 + i * 16 + fr;                                                                    \
        WF[i] = *reinterpret_cast<const gb_bf16x8*>(sB + rw_ * 128 + ((((KS) * 4 + fg) ^ (rw_ & 7)) << 4));        \
    }                                                                                                             \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                               \
        const int ra_ = wm * 128 + i * 16 + fr;                                                                   \
        AF[i] = *reinterpret_cast<const gb_bf16x8*>(sA + ra_ * 128 + ((((KS) * 4 + fg) ^ (ra_ & 7)) << 4));        \
    }
#define TEO_GB_MFMA(AF, WF, N0, N1)                                                                               \
    _Pragma("unroll") for (int ni = N0; ni < N1; ++ni)                                                            \
        _Pragma("unroll") for (int mi = 0; mi < 8; ++mi)                                                          \
            acc[ni][mi] = mfma16<F16>(WF[ni], AF[mi], acc[ni][mi]);
    for (int kt = kb; kt < ke; ++kt) {
        const int st = (kt - kb) & 1;
        // this wave's pieces of tile kt have landed and its fragment reads of tile kt-1 have returned; then everybody's
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (!late && kt + 1 < ke) { TEO_GB_STAGE(kt + 1, st ^ 1) }
        __builtin_amdgcn_sched_barrier(0);
        const unsigned char* sA = smem + st * stage_bytes;
        const unsigned char* sB = sA + a_bytes;
        TEO_GB_READ(af, wf, 0)
        __builtin_amdgcn_sched_barrier(0);
        if (kt > kb) { TEO_GB_MFMA(caf, cwf, 0, 4) }       // second half of tile kt-1, from registers, under the reads above
        __builtin_amdgcn_sched_barrier(0);
        if (late && kt + 1 < ke) { TEO_GB_STAGE(kt + 1, st ^ 1) }
        __builtin_amdgcn_sched_barrier(0);
        TEO_GB_READ(caf, cwf, 1)
        __builtin_amdgcn_sched_barrier(0);
        TEO_GB_MFMA(af, wf, 0, 4)
        __builtin_amdgcn_sched_barrier(0);
    }
    if (ke > kb) { TEO_GB_MFMA(caf, cwf, 0, 4) }
#undef TEO_GB_READ
#undef TEO_GB_MFMA
#undef TEO_GB_STAGE
    if (HYBRID) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                       // every wave is done reading the ring before the next segment refills it
        if (is_tail) {
            const auto sl = __builtin_amdgcn_make_buffer_rsrc(slabs + (size_t)q * GB_SLAB_FLOATS, 0, GB_SLAB_FLOATS * 4, 0x00020000);
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int mi = 0; mi < 8; ++mi)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned int, acc[ni][mi]), sl,
                                                           ((ni * 8 + mi) * 512 + tid) * 16, 0, /*sc1*/ 16);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // every storing wave: write-through stores landed
            __builtin_amdgcn_s_barrier();
            if (tid == 0) __hip_atomic_store(flags + q, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            continue;
        }
    }

    // epilogue (gemm_epilogue.h); the bias is fetched here, not ahead of the K loop: this kernel has no registers to spare
    {
        uint2 bv[4];
        gemm_bias_load<4>(SWIGLU ? nullptr : bias, n0 + wn * 64, fg, N, bv);
        gemm_epilogue<4, 8, 1, SWIGLU, OUT_F32, F16>(acc, bv, bias != nullptr, res, Cv, M, N, ldc, act, m0 + wm * 128, n0 + wn * 64, fr, fg);
    }
    }   // segments
}

// tune().gemm_big_group (default 0): N panels per tile group (0: from the tile grid)
// tune().gemm_big_cohort (default -1): stream-K part in XCD-local cohorts of this many workgroups: -1 auto (16 behind data-parallel rounds, else linear), 0 linear ranges, 8 / 16 / 32
// tune().gemm_big_hybrid (default 1): data-parallel rounds + stream-K remainder when a workspace is given (1: if it fits MALL, 2: always)
// ... or there are at most 1.5 tiles per workgroup (a pure stream-K grid whose workgroups mostly stay on one tile: down at M = 4208,
// 272 tiles, 183 MB: 311 us against 421 us for three ragged rounds of 128 x 256 tiles)
// tune().gemm_big_ragged (default 1): a last row block of <= 128 rows as 128 x 512 tiles (round 6): 0 = never (a padded 256-row tile), 1 = auto,
// 2 = whenever the shape allows.  Auto = only where it turns the problem into ONE round of tiles: gate/up at M = 638 (config C2) is 258
// padded tiles = a round and two tiles, 215 with the ragged form -- 104.9 us against 119.3 (hybrid) / 173.5 (two rounds).  Elsewhere the
// tile count drops by 5.6 % but not the number of rounds, and a 128 x 512 tile moves 80 KB per K step instead of 64: measured
// (tools/dispatch_probe.py, M = 2168 / 4208) qkv 193.8 vs 193.8 / 348 vs 356, gate/up 340-372 vs 333 / 620 vs 604 us -- a wash or a loss,
// because the hybrid form's stream-K part grows when a data-parallel round disappears (731 tiles = 1 round + 475 instead of 2 + 262).
// (bytes / tile-count rule of the hybrid form: see gemm_big_hybrid_fits below)
static bool gb_hybrid_rule(long long T, long long bytes) { return bytes <= (160ll << 20) || T <= 384 || (T <= 800 && bytes <= (208ll << 20)); }
// K > 0 adds the second automatic case: the ragged form saves a whole ROUND of the plain (non-hybrid) kernel where the hybrid form does not
// apply to the padded problem anyway -- gate/up at M = 4353 .. 4480 (1548 -> 1505 tiles: seven rounds -> six; 637 us against 654 padded, 710 on
// the 128 x 256 tile the rounds model fell back to)
int gemm_big_ragged_tiles(int M, int N, int K) {
    const int rows = M % GB_BM, tiles_n = (N + GB_BN - 1) / GB_BN, mode = tune().gemm_big_ragged;
    if (!(mode && rows > 0 && rows <= GB_BM / 2 && tiles_n % 2 == 0 && M >= GB_BM)) return 0;
    const long long t_rag = (long long)(M / GB_BM) * tiles_n + tiles_n / 2, t_full = (long long)(M / GB_BM + 1) * tiles_n;
    bool take = mode == 2 || (t_rag <= 256 && t_full > 256);
    if (!take && K > 0 && (t_rag + 255) / 256 < (t_full + 255) / 256) {
        const bool hybrid_would_run = t_full > 256 && t_full % 256 != 0 && gb_hybrid_rule(t_full, ((long long)M + N) * K * 2);
        take = !hybrid_would_run;
    }
    return take ? tiles_n / 2 : 0;
}
long long gemm_big_tile_count(int M, int N, int K) {
    const int rt = gemm_big_ragged_tiles(M, N, K);
    return (long long)(rt ? M / GB_BM : (M + GB_BM - 1) / GB_BM) * ((N + GB_BN - 1) / GB_BN) + rt;
}

bool gemm_big_hybrid_fits(int M, int N, int K) {
    const long long T = gemm_big_tile_count(M, N, K);
    const long long bytes = ((long long)M + N) * K * 2;
    // round 3 (tools/split_probe.py): gate/up at M = 2168 (198 MB, 774 tiles = 3.02 rounds) runs 337 / 342 us (warm / cold weights) in
    // the hybrid form against 358 / 363 on the 128 x 256 kernel -- with only three rounds the ragged one costs more than the
    // unshared stream-K part; at M = 4208 (214 MB, 5.7 rounds) the hybrid form loses (712 vs 580 us)
    return gb_hybrid_rule(T, bytes);
}

int gemm_big_launch(const void* A, const void* W, const void* bias, const void* res, void* C, int M, int N, int K, int lda, int ldc,
                    int act, bool swiglu, bool of32, bool f16, hipStream_t st, void* sk_ws, size_t flags_offset) {
    const int tiles_n = cdiv(N, GB_BN);
    const int rag_tiles = gemm_big_ragged_tiles(M, N, K);  // > 0: the last M % 256 <= 128 rows as tiles_n / 2 tiles of 128 x 512 (see the kernel)
    const int tiles_m = rag_tiles ? M / GB_BM : cdiv(M, GB_BM);
    const int T = tiles_m * tiles_n + rag_tiles, nk = K / GB_BK;
    const size_t lds = rag_tiles ? 2 * (size_t)GB_RAG_STAGE : 2 * (size_t)GB_STAGE;
    const int group = tune().gemm_big_group ? tune().gemm_big_group : (tiles_m >= 16 ? 4 : 1);
    // hybrid form when the tile count is not a whole number of rounds -- and the operands fit the 256 MB Infinity Cache: the
    // stream-K part has every workgroup at its own (tile, k), nothing is shared through L2, and once A + W no longer sit in MALL its
    // K tiles take twice as long as the data-parallel ones (measured: gate/up at M = 4208, 214 MB: 712 us vs 580; at M = 2168 qkv,
    // 118 MB: 195 vs 203; gemm_big_hybrid = 2 forces it)
    const bool hybrid = sk_ws && tune().gemm_big_hybrid && T > 256 && T % 256 != 0 && (tune().gemm_big_hybrid == 2 || gemm_big_hybrid_fits(M, N, K));
    const int dp_rounds = hybrid ? T / 256 - 1 : 0;
    // cohort form: columns of `cohort` tiles, 256 / cohort chain links; a link's range must cover a whole tile (per >= nk), else linear.
    // Measured (tools/bench_kernels.py gemm_cohort, cold weights, us; linear / 8 / 16 / 32): gate/up at M = 2168 (2 rounds + 262 tiles)
    // 342.9 / 324.9 / 322.1 / 329.4, qkv at M = 4208 (2 + 304) 392.6 / 366.1 / 360.7 / 369.4, gate/up at M = 4208 (4 + 438) 723.6 / 609.1 /
    // 601.3 / 596.9; with NO data-parallel round in front the linear ranges stay ahead or level (qkv at M = 2168, 432 tiles: 197.6 /
    // 208.1 / 202.1 / 212.1; gate/up at M = 638, 258 tiles: 117.2 / 123.8 / 126.9 / 135.0; down at M = 4208, 272 tiles: 305.8 / 301.1 /
    // 295.1 / 308.3) -> auto = 16 behind at least one data-parallel round
    int cohort = !hybrid ? 0 : (tune().gemm_big_cohort >= 0 ? tune().gemm_big_cohort : (dp_rounds >= 1 ? 16 : 0));
    if (cohort && cdiv(T - dp_rounds * 256, cohort) < 256 / cohort) cohort = 0;
    const int per = !hybrid ? 0 : cohort ? cdiv((long long)cdiv(T - dp_rounds * 256, cohort) * nk, 256 / cohort)
                                         : (int)(((long long)(T - dp_rounds * 256) * nk + 255) / 256);
    float* slabs = (float*)sk_ws;
    int* flg = hybrid ? (int*)((unsigned char*)sk_ws + flags_offset) : nullptr;
#define TEO_GB_LAUNCH_H(SW, OF, HY) { if (f16) TEO_GB_LAUNCH_HF(SW, OF, HY, true) else TEO_GB_LAUNCH_HF(SW, OF, HY, false) }
#define TEO_GB_LAUNCH_HF(SW, OF, HY, FV)                                                                                          \
    {                                                                                                                             \
        static unsigned long long attr_mask = 0;                                                                                  \
        if (int e = lds_attr_once(reinterpret_cast<const void*>(&gemm_mfma_bf16_big_kernel<SW, OF, HY, FV>), 2 * GB_RAG_STAGE, &attr_mask, "gemm_big")) return e; \
        gemm_mfma_bf16_big_kernel<SW, OF, HY, FV><<<(HY) ? 256 : T, 512, lds, st>>>((const bf16_t*)A, (const bf16_t*)W, (const bf16_t*)bias, \
                                                                               (const bf16_t*)res, C, M, N, K, lda, ldc, act, tiles_m,  \
                                                                               tiles_n, group, dp_rounds, per, slabs, flg, cohort, rag_tiles); \
    }
#define TEO_GB_LAUNCH(SW, OF) { if (hybrid) TEO_GB_LAUNCH_H(SW, OF, true) else TEO_GB_LAUNCH_H(SW, OF, false) }
    if (swiglu) { if (of32) TEO_GB_LAUNCH(true, true) else TEO_GB_LAUNCH(true, false) }
    else { if (of32) TEO_GB_LAUNCH(false, true) else TEO_GB_LAUNCH(false, false) }
#undef TEO_GB_LAUNCH
#undef TEO_GB_LAUNCH_H
#undef TEO_GB_LAUNCH_HF
    note_kernel(hybrid ? (cohort ? "gemm_big_hybrid_cohort" : "gemm_big_hybrid") : "gemm_big");
    TEO_LAUNCH_CHECK("gemm_mfma_bf16_big");
    return TEO_OK;
}

}  // namespace teo
