// Prefill / tower attention on v_mfma_f32_32x32x16_bf16 (gfx950), D in {64, 128}, causal or not.
//
// Workgroup = 4 waves = 128 queries of one head (32 per wave); KV tiles of 64 keys staged by LDS-DMA (global_load_lds_dwordx4: no
// staging registers, no ds_write pass) into two slots (K row-major, V^T key-contiguous; 2 x 32 KB at D = 128 -> two workgroups per
// CU), ONE barrier per tile.  Round 4: the loop is a software pipeline inside the wave (causal kernels):
//     iteration t:  DMA V^T(t+1), K(t+2) into the free slots
//                   S^T(t+1) = K(t+1) . Q^T (16 MFMAs) issued in one basic block with the online softmax of tile t (registers)
//                   O^T += V^T(t) . P(t)^T (16 MFMAs; the compiler sinks the exponentials between them)
//                   vmcnt(0), barrier
// The score MFMA is SWAPPED (S^T = K . Q^T): a lane owns ONE query (column l & 31) and 16 of the 32 keys of a block, so the
// row max / row sum are in-lane reductions plus one v_permlane32_swap, the rescale factor is a per-lane scalar, and the
// exponentiated block is already the B operand of O^T += V^T . P^T -- P never touches LDS.  The two lanes of a query trade half of
// their packed P (four v_permlane32_swap per 32-key block) so that each holds 8 CONSECUTIVE keys per k-step: V^T stays in natural
// key order, which a DMA can write, and the A operand is one ds_read_b128 per lane.  (Rounds 2-3 staged through registers and
// stored V^T as [g0 g2 g1 g3] per 16 keys instead.)
// 32 x 32 x 16 instead of 16 x 16 x 32: one K / V^T fragment read feeds 32 queries instead of 16 -- half the LDS bytes per
// FLOP of the round-1 kernel, which was LDS-bound (52 % of its wave cycles parked, 33 % of its LDS cycles bank conflicts).
// LDS images (bank maths: MI355X_MICROARCH.md section LDS; ds_read_b128 is served in 16-lane groups over a 256-byte row), written
// wave-linear by the DMA with the permutation applied through the SOURCE address:
//   K   [64 keys][D], 16-byte chunk c of row r at c ^ ((r / RPB) & (CH - 1)), RPB = rows per 256 bytes  -> conflict-free
//   V^T [D][64 keys], 16-byte chunk c of row d at c ^ ((d >> 1) & 7)                                     -> conflict-free
// Arithmetic = oracle attention_core mode "flash64": 64-key tiles from key 0, running max, P = exp2(s * scale * log2e - m),
// P rounded to bf16 per tile for the PV product, normaliser from the unrounded P, fp32 rescale (skipped, bit-identically,
// when no row maximum of the wave moved).
// Roofline: MFMA bf16 dense (2.5 PFLOP/s); algorithmic FLOPs = 4 * q_len * kv_len * D * heads (half of it when causal).
#include <string.h>

#include "common.h"
#include "ops.h"

namespace teo {

typedef __attribute__((ext_vector_type(8))) short fa_bf16x8;      // 8 raw 16-bit operands (bf16 or fp16: the F16 template flag)
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int fa_u32x4;
constexpr int FA_TRACE_ITERS = 40;            // TRACE: iterations recorded per wave (tools/flash_probe.hip)

__device__ __forceinline__ float fa_other_half_max(float x) {
    // max of this lane's value and lane ^ 32's: v_permlane32_swap exchanges the upper half of vdst with the lower half of src
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

// pair_c (causal only): workgroup slots per XCD that the dispatcher fills in its first pass (CUs per XCD); 0 = plain heavy-first order
// TRACE (tools/flash_probe.hip only): 100 MHz wall-clock marks of every loop phase of the heaviest workgroup's waves
// PIPE: the in-wave software pipeline (score MFMAs of tile t+1 issued with the softmax of tile t); false = one tile at a time
// NB: (K, V^T) tile pairs in LDS.  2 everywhere but the short non-causal D = 64 case (the tower: 5 tiles), which takes four and requests
// a tile three iterations ahead: there an iteration is shorter than a DMA round trip
template <int D, bool CAUSAL, bool F16 = false, bool TRACE = false, bool PIPE = true, int NB = 2>
__global__ __launch_bounds__(256, 2) void attn_flash32_kernel(teo_attn_args a, int pair_c, unsigned long long* trace) {
    constexpr int CH = D / 8;                 // 16-byte chunks per K row
    constexpr int KROW = D * 2;               // bytes per K row
    constexpr int RPB = 256 / KROW > 0 ? 256 / KROW : 1;
    constexpr int KT_BYTES = 64 * KROW;       // K tile
    constexpr int VROW = 128;                 // V^T row: 64 keys in natural order, 16-byte chunk c of row d at c ^ ((d >> 1) & 7)
    constexpr int VT_BYTES = D * VROW;        // V^T tile: D rows x 64 keys
    constexpr int VBASE = NB * KT_BYTES;      // K tile t at (t % NB) * KT_BYTES, V^T tile t at VBASE + (t % NB) * VT_BYTES

    constexpr int RPP = 1024 / KROW;          // K rows per 1 KB DMA piece
    constexpr int NPK = (64 / RPP) / 4;       // K pieces per wave
    constexpr int NPV = (D / 8) / 4;          // V^T pieces (8 rows each) per wave
    constexpr int GRP = NPK + NPV;            // DMA instructions per wave per tile
    static_assert(!PIPE || NB == 2, "the in-wave pipeline stages K one tile ahead of V^T in two buffers");
    constexpr int NDB = D / 32;               // 32-row d-blocks of O^T
    constexpr int NKK = D / 16;               // k-steps of the score MFMA
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int ql = lane & 31, hi = lane >> 5;
    // 1-D grid -> (batch*head, query block).  Workgroups are dispatched round-robin over the 8 XCDs (each with its own L2):
    // XCD x takes the heads = x (mod 8), so every query block of a head streams that head's K / V^T through ONE L2, and
    // inside an XCD consecutive slots walk the heads at equal query block, late (heavy, causal) query blocks first.
    // Two workgroups share a CU (LDS): the dispatcher's first pass puts one workgroup on every CU of the XCD (slots 0 .. C-1, the
    // heaviest), its second pass the next C in the same CU order -- those are taken in MIRRORED weight order, so the heaviest of
    // the first pass is joined by the lightest of the second (causal work 34 + 4, 32 + 6, ... key tiles at L = 2168 instead of
    // 34 + 18 ... 20 + 4); what is left (the lightest of all) fills the slots that free up first.  Same tiles, same arithmetic.
    const int nqb = (a.q_len + 127) >> 7, nbh = a.heads * a.batch;
    int bh, qidx;
    if ((nbh & 7) == 0) {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, hpx = nbh >> 3;
        int j = slot;
        if (CAUSAL && pair_c > 0) {
            const int n = hpx * nqb, e2 = min(2 * pair_c, n);
            if (slot >= pair_c && slot < e2) j = pair_c + e2 - 1 - slot;
        }
        bh = xcd + 8 * (j % hpx);
        qidx = j / hpx;
    } else {
        bh = blockIdx.x % nbh;
        qidx = blockIdx.x / nbh;
    }
    const int h = bh % a.heads, b = bh / a.heads;
    const int hk = h / (a.heads / a.kv_heads);
    const int qb = (nqb - 1 - qidx) * 128;
    const int off = a.kv_len - a.q_len;
    const bf16_t* Q = (const bf16_t*)a.q + b * a.q_bs + h * a.q_hs;
    const bf16_t* K = (const bf16_t*)a.k + b * a.k_bs + hk * a.k_hs;
    const bf16_t* VT = (const bf16_t*)a.vt + b * a.vt_bs + hk * a.vt_hs;

    // this lane's query row as the B operand of S^T = K . Q^T: Q[q][kk*16 + hi*8 .. +8]
    const int qi = qb + wid * 32 + ql;
    const int qrow = min(qi, a.q_len - 1);
    fa_bf16x8 qf[NKK];
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk)
        qf[kk] = __builtin_bit_cast(fa_bf16x8, *reinterpret_cast<const fa_u32x4*>(Q + (long long)qrow * a.q_rs + kk * 16 + hi * 8));
    const int qpos = qi + off;                                          // last key this query may see (causal)
    const int wave_qpos_max = qb + wid * 32 + 31 + off;                 // ... of the wave's last query
    const int wave_qpos_min = qb + wid * 32 + off;

    f32x16 acc_o[NDB];
#pragma unroll
    for (int i = 0; i < NDB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_o[i][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const float sl2 = a.scale * 1.44269504088896340736f;               // scores in log2 units

    int kv_end = a.kv_len;
    if (CAUSAL) kv_end = min(a.kv_len, qb + 127 + off + 1);
    const int ntiles = (kv_end + 63) >> 6;

    // ---- staging by LDS-DMA (global_load_lds_dwordx4: a wave instruction moves 64 x 16 bytes from per-lane global addresses to 1 KB of
    // LDS, wave-linear; no staging VGPRs, no ds_write pass, no address arithmetic in the loop).  The LDS images are the ones the
    // fragment reads want; the permutation goes through the SOURCE address:
    //   K   piece g = 1 KB = RPP rows: lane l brings row r = g * RPP + l / CH, global chunk (l % CH) ^ ((r / RPB) & (CH - 1))
    //   V^T piece g = 8 rows of 128 B:  lane l brings row d = 8 g + (l >> 3), global chunk (l & 7) ^ ((d >> 1) & 7)
    // V^T sits in NATURAL key order.  The score MFMA leaves a lane (query q, half hi) with P of keys {0-3, 8-11, 16-19, 24-27} + 4 hi of
    // a 32-key block; four v_permlane32_swap on the packed P (the two lanes of a query trade keys 8-11 <-> 4-7 and 24-27 <-> 20-23)
    // turn that into keys 0-7 / 16-23 (hi = 0) and 8-15 / 24-31 (hi = 1): the k-slots of O^T += V^T . P^T then run over 16 CONSECUTIVE
    // keys, and the A operand of lane (d, hi) is the 16-byte chunk 2 u + hi of row d -- one conflict-free ds_read_b128 (the K image's
    // swizzle for 128-byte rows), from an image that a DMA can write.
    const int wv = __builtin_amdgcn_readfirstlane(wid);
    // 32-bit element offsets (one head's K / V^T is far below 2^31 elements): uniform tile term on the scalar unit, no 64-bit multiplies
    unsigned koff[NPK];                                    // row r of tile 0, this lane's chunk (tile T: + T * 64 rows)
    const bf16_t* vsrc[NPV];
    int krow[NPK];
    const unsigned k_rs32 = (unsigned)a.k_rs;
#pragma unroll
    for (int i = 0; i < NPK; ++i) {
        const int r = (wv * NPK + i) * RPP + lane / CH;
        krow[i] = r;
        koff[i] = (unsigned)r * k_rs32 + (unsigned)(((lane % CH) ^ ((r / RPB) & (CH - 1))) * 8);
    }
    const unsigned ktile = 64u * k_rs32;
#pragma unroll
    for (int i = 0; i < NPV; ++i) {
        const int d = (wv * NPV + i) * 8 + (lane >> 3);
        vsrc[i] = VT + (long long)d * a.vt_rs + ((lane & 7) ^ ((d >> 1) & 7)) * 8;
    }
    // The DMA pieces are INLINE ASM (round 6, late): to hipcc's wait pass a builtin LDS-DMA is a pending FLAT operation, and while one is pending
    // every wait it derives for a fragment register is forced to lgkmcnt(0) -- in the steady state (pieces of the next tiles requested at the
    // top of the iteration) each K / V^T fragment read was followed by a full wait in front of its MFMA, where the prologue (nothing pending)
    // compiles to a 2-deep read pipeline with lgkmcnt(2) / (1).  Invisible to the pass, the pieces only add events to the in-order vmcnt queue
    // (the loop's own vmcnt(0) is written in TEO_FA_PUBLISH); M0 = LDS byte address of the wave's 1 KB piece.
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)smem);
#define TEO_FA_DMA16(GPTR, LDS_OFF)                                                                               \
    {                                                                                                             \
        const bf16_t* gp_ = (GPTR);                                                                               \
        const unsigned la_ = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(LDS_OFF));                          \
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gp_), "s"(la_) : "memory"); \
    }
#define TEO_FA_DMA_K(T_, SLOT)                                                                                        \
    _Pragma("unroll") for (int i = 0; i < NPK; ++i) {                                                             \
        /* rows past kv_len - 1 (only in the tile that crosses it) re-read the last row: their scores are masked */ \
        const unsigned back_ = (unsigned)max((T_) * 64 + krow[i] - (a.kv_len - 1), 0);                            \
        TEO_FA_DMA16(K + (koff[i] + (unsigned)(T_) * ktile - back_ * k_rs32), (SLOT) * KT_BYTES + (wv * NPK + i) * 1024)      \
    }
#define TEO_FA_DMA_V(T_, SLOT)                                                                                        \
    _Pragma("unroll") for (int i = 0; i < NPV; ++i)                                                               \
        TEO_FA_DMA16(vsrc[i] + (T_) * 64, VBASE + (SLOT) * VT_BYTES + (wv * NPV + i) * 1024)
    // the tile that crosses kv_len: keys >= kv_len of the V^T image are zeroed after it landed (P is 0 there, the cache row may hold anything)
#define TEO_FA_TAIL_FIX(T_, SLOT)                                                                                 \
    if ((T_) < ntiles && (T_) * 64 + 64 > a.kv_len) {                                                             \
        unsigned char* sV_ = smem + VBASE + (SLOT) * VT_BYTES;                                                    \
        const int valid = a.kv_len - (T_) * 64;                  /* 1 .. 63 keys of this tile exist */            \
        for (int id = tid; id < D * 8; id += 256) {              /* one 16-byte chunk (8 keys) of one row */      \
            const int d = id >> 3, c = id & 7;                                                                    \
            fa_u32x4* cp = reinterpret_cast<fa_u32x4*>(sV_ + d * VROW + ((c ^ ((d >> 1) & 7)) << 4));             \
            const int keep = valid - c * 8;                      /* keys of this chunk that exist */               \
            if (keep <= 0) *cp = (fa_u32x4){0u, 0u, 0u, 0u};                                                      \
            else if (keep < 8) {                                                                                  \
                fa_u32x4 v_ = *cp;                                                                                \
                _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                   \
                    if (2 * e >= keep) v_[e] = 0u;                                                                \
                    else if (2 * e + 1 >= keep) v_[e] &= 0xffffu;                                                 \
                }                                                                                                 \
                *cp = v_;                                                                                         \
            }                                                                                                     \
        }                                                                                                         \
        __syncthreads();                                                                                          \
    }
    // S^T = K . Q^T of tile T_ (its K in buffer T_ & 1) into SX: 2 key blocks x NKK k-steps.  No conditions inside: the caller decides.
#define TEO_FA_SCORES(SLOT, SX)                                                                                   \
    {                                                                                                             \
        const unsigned char* sK = smem + (SLOT) * KT_BYTES;                                                       \
        _Pragma("unroll") for (int kb = 0; kb < 2; ++kb) {                                                        \
            _Pragma("unroll") for (int r = 0; r < 16; ++r) SX[kb][r] = 0.f;                                       \
            const int row = kb * 32 + ql;                                                                         \
            const unsigned char* rp = sK + row * KROW;                                                            \
            const int sw = (row / RPB) & (CH - 1);                                                                \
            _Pragma("unroll") for (int kk = 0; kk < NKK; ++kk) {                                                  \
                const fa_bf16x8 kf = __builtin_bit_cast(fa_bf16x8, *reinterpret_cast<const fa_u32x4*>(rp + (((2 * kk + hi) ^ sw) << 4))); \
                SX[kb] = mfma32<F16>(kf, qf[kk], SX[kb]);                                                         \
            }                                                                                                     \
        }                                                                                                         \
    }
    // online softmax of tile T_ whose scores sit in SX (lane holds S[key = j0 + kb*32 + (r&3) + 8*(r>>2) + 4*hi][query ql]); leaves P in SX.
    // MASKED: tiles that cross kv_len or the causal diagonal of this wave
#define TEO_FA_SOFTMAX(T_, SX, MASKED)                                                                            \
    float tmax = -INFINITY;                                                                                       \
    if (MASKED) {                                                                                                 \
        _Pragma("unroll") for (int kb = 0; kb < 2; ++kb)                                                          \
            _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                      \
                const int key = (T_) * 64 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;                            \
                const bool ok = (key < a.kv_len) && (!CAUSAL || key <= qpos);                                     \
                const float v = ok ? SX[kb][r] * sl2 : -INFINITY;                                                 \
                SX[kb][r] = v;                                                                                    \
                tmax = fmaxf(tmax, v);                                                                            \
            }                                                                                                     \
    } else {                                                                                                      \
        _Pragma("unroll") for (int kb = 0; kb < 2; ++kb)                                                          \
            _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                      \
                const float v = SX[kb][r] * sl2;                                                                  \
                SX[kb][r] = v;                                                                                    \
                tmax = fmaxf(tmax, v);                                                                            \
            }                                                                                                     \
    }                                                                                                             \
    tmax = fa_other_half_max(tmax);                                                                               \
    const float m_new = fmaxf(m_run, tmax);                                                                       \
    const float m_use = (m_new == -INFINITY) ? 0.f : m_new;                                                       \
    float psum = 0.f;                                                                                             \
    _Pragma("unroll") for (int kb = 0; kb < 2; ++kb)                                                              \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                          \
            const float p = __builtin_amdgcn_exp2f(SX[kb][r] - m_use);     /* v_exp_f32 (results below 2^-126 flush to 0) */ \
            psum += p;                                                                                            \
            SX[kb][r] = p;                                                                                        \
        }
    // rescale when some row maximum of the wave moved, then O^T += V^T . P^T : 4 blocks of 16 keys x NDB d-blocks (V^T in buffer T_ & 1)
#define TEO_FA_PV(SLOT, SX)                                                                                       \
    {                                                                                                             \
        if (!__all(m_new == m_run)) {                                                                             \
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_use);   /* m_run = -inf -> 0 */                  \
            l_run *= alpha;                                                                                       \
            _Pragma("unroll") for (int i = 0; i < NDB; ++i)                                                       \
                _Pragma("unroll") for (int r = 0; r < 16; ++r) acc_o[i][r] *= alpha;                              \
        }                                                                                                         \
        l_run += psum;                                                                                            \
        m_run = m_new;                                                                                            \
        const unsigned char* sV = smem + VBASE + (SLOT) * VT_BYTES;                                               \
        unsigned pk[2][8];                                  /* P rounded to the storage type, two keys per register */ \
        _Pragma("unroll") for (int kb = 0; kb < 2; ++kb) {                                                        \
            _Pragma("unroll") for (int j = 0; j < 8; ++j) pk[kb][j] = pack_h2<F16>(SX[kb][2 * j], SX[kb][2 * j + 1]); \
            _Pragma("unroll") for (int g = 0; g < 2; ++g) {     /* keys 8-11 of hi = 0 <-> keys 4-7 of hi = 1 (and + 16) */ \
                _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                   \
                    const auto sw_ = __builtin_amdgcn_permlane32_swap(pk[kb][4 * g + j], pk[kb][4 * g + 2 + j], false, false); \
                    pk[kb][4 * g + j] = sw_[0];                                                                   \
                    pk[kb][4 * g + 2 + j] = sw_[1];                                                               \
                }                                                                                                 \
            }                                                                                                     \
        }                                                                                                         \
        _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                                           \
            const int kb = u >> 1, r0 = (u & 1) * 4;                                                              \
            union { fa_bf16x8 v; unsigned w[4]; } pf;                                                             \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) pf.w[j] = pk[kb][r0 + j];                               \
            _Pragma("unroll") for (int db = 0; db < NDB; ++db) {                                                  \
                const int d = db * 32 + ql;                                                                       \
                const fa_bf16x8 vf = __builtin_bit_cast(fa_bf16x8, *reinterpret_cast<const fa_u32x4*>(sV + d * VROW + (((2 * u + hi) ^ ((d >> 1) & 7)) << 4))); \
                acc_o[db] = mfma32<F16>(vf, pf.v, acc_o[db]);                                                     \
            }                                                                                                     \
        }                                                                                                         \
    }
    // TRACE: marks 0 loop top, 1 DMA issued, 2 scores (+ softmax) done, 3 PV done, 4 DMA landed, 5 past the barrier
#define TEO_FA_MARK(T_, PH)                                                                                        \
    if constexpr (TRACE) {                                                                                         \
        if (blockIdx.x == 0 && lane == 0 && (T_) < FA_TRACE_ITERS) trace[((wid * FA_TRACE_ITERS + (T_)) * 6) + (PH)] = wall_clock64(); \
    }
    // every wave waits for its own DMA pieces, the barrier publishes everyone's (and says the buffers read in this iteration are free)
#define TEO_FA_PUBLISH()                                                                                          \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                              \
    __syncthreads();
    // ---- The loop is a two-stage software pipeline INSIDE a wave (PIPE).  The serial chain of one iteration -- score MFMAs, softmax VALU,
    // PV MFMAs -- is what bounds the heaviest causal workgroup (tools/flash_probe.py), so in the steady state iteration t issues the
    // score MFMAs of tile t+1, which depend on nothing in flight, IN THE SAME BASIC BLOCK as the softmax of tile t: the matrix pipe
    // works in the shadow of the exponentials.  K is therefore staged one tile ahead of V^T:
    //     iteration t:  DMA V^T(t+1), K(t+2)  |  [ S(t+1) = K(t+1) . Q^T  ||  softmax(t) ]  ->  O^T += V^T(t) . P(t)^T  |  vmcnt(0), barrier
    // K(t) and V^T(t) live in slot t % NB; every DMA targets a slot whose last readers finished before the previous barrier.
    // Tiles that need masking (the causal diagonal, the tail of kv_len), the last tile and waves with nothing left take the plain
    // sequence; masked tiles are the LAST tiles of a wave, so a plain score pass never reads a K buffer that a DMA of the same
    // iteration is refilling (it only happens from ntiles - 2 on, where no K(t+2) exists).  Same tiles, same expressions, same order
    // per output as the one-tile-at-a-time loop (!PIPE: DMA of tile t+NB-1 at the top of iteration t).
    f32x16 sa[2], sb[2];
    bool have_s = false;                                    // SC of the coming iteration already holds its scores
    int kslot = 0;                                          // LDS slot of tile t (K and V^T) at the top of iteration t; tile t+1 sits in the next one
    // waits until at most `groups` of the most recently requested tiles are still in flight (vmcnt counts instructions, in order)
#define TEO_FA_WAIT_GROUPS(G_)                                                                                    \
    {                                                                                                             \
        const int g_ = (G_);                                                                                      \
        if (g_ <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                             \
        else if (g_ == 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(GRP) : "memory");                             \
        else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * GRP) : "memory");                                      \
    }
    if (PIPE) {
        TEO_FA_DMA_K(0, 0)
        TEO_FA_DMA_V(0, 0)
        if (ntiles > 1) TEO_FA_DMA_K(1, 1)
        TEO_FA_PUBLISH()
        TEO_FA_TAIL_FIX(0, 0)
        TEO_FA_SCORES(0, sa)
        have_s = true;
        __syncthreads();                                    // K(0)'s buffer is refilled by iteration 0's DMA of K(2)
    } else {
        const int npro = min(NB - 1, ntiles);               // tiles 0 .. NB-2 up front, then one per iteration
        for (int t = 0; t < npro; ++t) { TEO_FA_DMA_K(t, t) TEO_FA_DMA_V(t, t) }
        TEO_FA_WAIT_GROUPS(min(npro - 1, 2))
        __syncthreads();
        TEO_FA_TAIL_FIX(0, 0)
    }
#define TEO_FA_ITER(T_, SC, SN)                                                                                   \
    {                                                                                                             \
        TEO_FA_MARK(T_, 0)                                                                                        \
        const int knext_ = kslot + 1 == NB ? 0 : kslot + 1;                                                       \
        if (PIPE) {                                                                                               \
            if ((T_) + 1 < ntiles) TEO_FA_DMA_V((T_) + 1, knext_)                                                 \
            if ((T_) + 2 < ntiles) TEO_FA_DMA_K((T_) + 2, kslot)      /* into the slot of K(t): last read in iteration t-1 */ \
        } else if ((T_) + NB - 1 < ntiles) {                /* into the slot of tile t-1 */                          \
            const int kprev_ = kslot == 0 ? NB - 1 : kslot - 1;                                                   \
            TEO_FA_DMA_K((T_) + NB - 1, kprev_)                                                                   \
            TEO_FA_DMA_V((T_) + NB - 1, kprev_)                                                                   \
        }                                                                                                         \
        TEO_FA_MARK(T_, 1)                                                                                        \
        const int j0_ = (T_) * 64;                                                                                \
        if (!CAUSAL || j0_ <= wave_qpos_max) {              /* else: this wave's queries all lie before the tile */ \
            const bool need_mask = (j0_ + 64 > a.kv_len) || (CAUSAL && j0_ + 63 > wave_qpos_min);                 \
            const bool next_too = PIPE && (T_) + 1 < ntiles && (!CAUSAL || j0_ + 64 <= wave_qpos_max);            \
            if (!have_s) TEO_FA_SCORES(kslot, SC)                                                                 \
            if (!PIPE) __builtin_amdgcn_sched_barrier(0);   /* one tile at a time: keep the PV fragment reads behind the scores (registers) */ \
            if (next_too && !need_mask) {                                                                         \
                TEO_FA_SCORES(knext_, SN)                                                                         \
                TEO_FA_SOFTMAX(T_, SC, false)                                                                     \
                TEO_FA_MARK(T_, 2)                                                                                \
                TEO_FA_PV(kslot, SC)                                                                              \
                have_s = true;                                                                                    \
            } else if (need_mask) {                                                                               \
                TEO_FA_SOFTMAX(T_, SC, true)                                                                      \
                TEO_FA_MARK(T_, 2)                                                                                \
                TEO_FA_PV(kslot, SC)                                                                              \
                have_s = false;                                                                                   \
            } else {                                                                                              \
                TEO_FA_SOFTMAX(T_, SC, false)                                                                     \
                TEO_FA_MARK(T_, 2)                                                                                \
                TEO_FA_PV(kslot, SC)                                                                              \
                have_s = false;                                                                                   \
            }                                                                                                     \
        }                                                                                                         \
        TEO_FA_MARK(T_, 3)                                                                                        \
        /* what the next iteration reads must have landed; tiles requested further ahead may stay in flight */   \
        TEO_FA_WAIT_GROUPS(PIPE ? 0 : min(ntiles - 1, (T_) + NB - 1) - ((T_) + 1))                                \
        TEO_FA_MARK(T_, 4)                                                                                        \
        __syncthreads();                                                                                          \
        TEO_FA_TAIL_FIX((T_) + 1, knext_)                                                                         \
        kslot = knext_;                                                                                           \
        TEO_FA_MARK(T_, 5)                                                                                        \
    }
    if constexpr (PIPE) {
_Pragma("nounroll")
        for (int t = 0; t < ntiles; t += 2) {               // unrolled by two: the score buffers swap roles by name, not by copy
            TEO_FA_ITER(t, sa, sb)
            if (t + 1 < ntiles) TEO_FA_ITER(t + 1, sb, sa)
        }
    } else {
_Pragma("nounroll")
        for (int t = 0; t < ntiles; ++t) TEO_FA_ITER(t, sa, sa)
    }
#undef TEO_FA_ITER
#undef TEO_FA_WAIT_GROUPS
#undef TEO_FA_PUBLISH
#undef TEO_FA_MARK
#undef TEO_FA_SCORES
#undef TEO_FA_SOFTMAX
#undef TEO_FA_PV
#undef TEO_FA_DMA_K
#undef TEO_FA_DMA_V
#undef TEO_FA_TAIL_FIX
    // ---- finish: l over the two lanes that share a query, normalise, store O[q][h*D + db*32 + (r&3) + 8*(r>>2) + 4*hi]
    {
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(l_run), __float_as_uint(l_run), false, false);
        l_run = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    if (qi < a.q_len) {
        const float inv = 1.0f / l_run;
        bf16_t* o = (bf16_t*)a.o + b * a.o_bs + (long long)qi * a.o_rs + h * D;
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const uint2 pk = make_uint2(pack_h2<F16>(acc_o[db][4 * g] * inv, acc_o[db][4 * g + 1] * inv),
                                            pack_h2<F16>(acc_o[db][4 * g + 2] * inv, acc_o[db][4 * g + 3] * inv));
                *reinterpret_cast<uint2*>(o + db * 32 + 8 * g + 4 * hi) = pk;
            }
    }
}

// tune().flash_pipe (default -1): in-wave pipeline: -1 auto (where the second score buffer fits the register budget), 0 off, 1 on
// tune().flash_order (default 1): causal workgroup order: 0 heavy-first, 1 heavy-first with the second dispatch pass mirrored (see the kernel)

int attention_flash32(const teo_attn_args& a, hipStream_t st, bool f16) {
    dim3 grid(cdiv(a.q_len, 128) * a.heads * a.batch);
    const int cus = device_cu_count();
    const int pair_c = (tune().flash_order == 1 && a.causal && cus >= 8) ? cus / 8 : 0;
    size_t lds = 2 * (size_t)(64 * a.head_dim * 2 + a.head_dim * 128);
    // the in-wave pipeline where its second score buffer fits the 256-register budget of two waves per SIMD (D = 128 non-causal spills)
#define TEO_FA(DD, CC, PP) { if (f16) attn_flash32_kernel<DD, CC, true, false, PP><<<grid, 256, lds, st>>>(a, pair_c, nullptr); else attn_flash32_kernel<DD, CC, false, false, PP><<<grid, 256, lds, st>>>(a, pair_c, nullptr); }
    const bool pipe = tune().flash_pipe < 0 ? (a.causal != 0) : tune().flash_pipe == 1;
    if (!pipe && a.head_dim == 64 && !a.causal) {           // the tower's shape: four tile pairs (64 KB), requests three iterations ahead
        lds = 4 * (size_t)(64 * 64 * 2 + 64 * 128);
        if (f16) attn_flash32_kernel<64, false, true, false, false, 4><<<grid, 256, lds, st>>>(a, pair_c, nullptr);
        else     attn_flash32_kernel<64, false, false, false, false, 4><<<grid, 256, lds, st>>>(a, pair_c, nullptr);
        note_kernel("attn_flash32"); TEO_LAUNCH_CHECK("attn_flash32");
        return TEO_OK;
    }
    if (a.head_dim == 128) { if (a.causal) { if (pipe) TEO_FA(128, true, true) else TEO_FA(128, true, false) } else TEO_FA(128, false, false) }     // (non-causal D = 128 with the pipeline spills: never built)
    else { if (a.causal) { if (pipe) TEO_FA(64, true, true) else TEO_FA(64, true, false) } else { if (pipe) TEO_FA(64, false, true) else TEO_FA(64, false, false) } }
#undef TEO_FA
    note_kernel("attn_flash32"); TEO_LAUNCH_CHECK("attn_flash32");
    return TEO_OK;
}

}  // namespace teo
