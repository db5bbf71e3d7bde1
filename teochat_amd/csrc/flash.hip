// Prefill / tower attention on v_mfma_f32_32x32x16_bf16 (gfx950), D in {64, 128}, causal or not.
//
// Workgroup = 4 waves = 128 queries of one head (32 per wave); KV tiles of 64 keys, double-buffered in LDS (K row-major,
// V^T key-contiguous; 2 x 32 KB at D = 128 -> two workgroups per CU), ONE barrier per tile:
//     iteration t:  write tile t+1 (in registers since iteration t-1) to the other LDS buffer
//                   issue the global loads of tile t+2
//                   S^T = K . Q^T (16 MFMAs) -> online softmax in registers -> O^T += V^T . P^T (16 MFMAs)
//                   barrier
// The score MFMA is SWAPPED (S^T = K . Q^T): a lane owns ONE query (column l & 31) and 16 of the 32 keys of a block, so the
// row max / row sum are in-lane reductions plus one v_permlane32_swap, the rescale factor is a per-lane scalar, and the
// exponentiated block is already the B operand of O^T += V^T . P^T -- P never touches LDS.  The k-slots of that product are
// matched to the keys a lane holds by the ORDER in which V^T is staged (inside every 16-key block the 4-key groups sit as
// [g0 g2 g1 g3]), so the A operand is one ds_read_b128 per lane.
// 32 x 32 x 16 instead of 16 x 16 x 32: one K / V^T fragment read feeds 32 queries instead of 16 -- half the LDS bytes per
// FLOP of the round-1 kernel (attn_mfma_kernel, kept for A/B as attn_flash = 0), which was LDS-bound (52 % of its wave
// cycles parked, 33 % of its LDS cycles bank conflicts, 7.8 % MFMA busy).
// LDS images (bank maths: MI355X_MICROARCH.md section LDS; ds_read_b128 is served in 16-lane groups over a 256-byte row):
//   K   [64 keys][D] bf16, 16-byte chunk c of row r at c ^ ((r / RPB) & (CH - 1)), RPB = rows per 256 bytes  -> conflict-free
//   V^T [D][64 keys] bf16, rows padded to 144 bytes (9 x 16: odd, so 16 rows fan out over all 16 slots of a bank row, and
//       two neighbouring rows' 8-byte staging writes fall into different halves of the 128-byte write period) -> conflict-free
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
template <int D, bool CAUSAL, bool F16 = false, bool TRACE = false>
__global__ __launch_bounds__(256, 2) void attn_flash32_kernel(teo_attn_args a, int pair_c, unsigned long long* trace) {
    constexpr int CH = D / 8;                 // 16-byte chunks per K row
    constexpr int KROW = D * 2;               // bytes per K row
    constexpr int RPB = 256 / KROW > 0 ? 256 / KROW : 1;
    constexpr int KT_BYTES = 64 * KROW;       // K tile
    constexpr int VROW = 144;                 // V^T row: 64 keys (128 B) + 16 B pad -> reads AND the 8-byte staging writes conflict-free
    constexpr int VT_BYTES = D * VROW;        // V^T tile: D rows x 64 keys
    constexpr int BUF = KT_BYTES + VT_BYTES;
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

    // ---- staging: thread owns NKC 16-byte chunks of the K tile and NVC of the V^T tile
    constexpr int NKC = (64 * CH) / 256, NVC = (D * 8) / 256;
    fa_u32x4 rk[NKC], rv[NVC];
#define TEO_FA_LOAD(T_)                                                                                           \
    {                                                                                                             \
        const int jt = (T_) * 64;                                                                                 \
        _Pragma("unroll") for (int i = 0; i < NKC; ++i) {                                                         \
            const int id = tid + 256 * i;                                                                         \
            const int r = id / CH, c = id % CH;                                                                   \
            const int gj = min(jt + r, a.kv_len - 1);                                                             \
            rk[i] = *reinterpret_cast<const fa_u32x4*>(K + (long long)gj * a.k_rs + c * 8);                       \
        }                                                                                                         \
        _Pragma("unroll") for (int i = 0; i < NVC; ++i) {                                                         \
            const int id = tid + 256 * i;                                                                         \
            const int d = id >> 3, c = id & 7;                                                                    \
            rv[i] = *reinterpret_cast<const fa_u32x4*>(VT + (long long)d * a.vt_rs + jt + c * 8);                 \
        }                                                                                                         \
    }
    // V^T chunk c = 2b + e of a row (keys 8c .. 8c+7 = groups g_{2e}, g_{2e+1} of 16-key block b): half hf goes to chunk
    // 2b + hf, 8-byte slot e -- the [g0 g2 | g1 g3] order the P operand holds its keys in.  Keys >= kv_len read as zero.
#define TEO_FA_WRITE(T_, BUFI)                                                                                    \
    {                                                                                                             \
        unsigned char* sK_ = smem + (BUFI) * BUF;                                                                 \
        unsigned char* sV_ = sK_ + KT_BYTES;                                                                      \
        const int jt = (T_) * 64;                                                                                 \
        _Pragma("unroll") for (int i = 0; i < NKC; ++i) {                                                         \
            const int id = tid + 256 * i;                                                                         \
            const int r = id / CH, c = id % CH;                                                                   \
            *reinterpret_cast<fa_u32x4*>(sK_ + r * KROW + ((c ^ ((r / RPB) & (CH - 1))) << 4)) = rk[i];           \
        }                                                                                                         \
        const bool tail = jt + 64 > a.kv_len;                                                                     \
        _Pragma("unroll") for (int i = 0; i < NVC; ++i) {                                                         \
            const int id = tid + 256 * i;                                                                         \
            const int d = id >> 3, c = id & 7;                                                                    \
            fa_u32x4 val = rv[i];                                                                                 \
            if (tail) {                                                                                           \
                const int valid = a.kv_len - (jt + c * 8);                                                        \
                _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                   \
                    if (2 * e >= valid) val[e] = 0u;                                                              \
                    else if (2 * e + 1 >= valid) val[e] &= 0xffffu;                                               \
                }                                                                                                 \
            }                                                                                                     \
            const int blk = c >> 1, e = c & 1;                                                                    \
            unsigned char* rowp = sV_ + d * VROW + (e << 3);                                                      \
            *reinterpret_cast<uint2*>(rowp + ((2 * blk) << 4)) = make_uint2(val[0], val[1]);                      \
            *reinterpret_cast<uint2*>(rowp + ((2 * blk + 1) << 4)) = make_uint2(val[2], val[3]);                  \
        }                                                                                                         \
    }

    TEO_FA_LOAD(0)
    TEO_FA_WRITE(0, 0)
    if (ntiles > 1) TEO_FA_LOAD(1)
    __syncthreads();

    // TRACE: marks 0 loop top, 1 staged (LDS writes + global loads issued), 2 scores done, 3 softmax done, 4 PV done, 5 past the barrier
#define TEO_FA_MARK(PH)                                                                                            \
    if constexpr (TRACE) {                                                                                         \
        if (blockIdx.x == 0 && lane == 0 && t < FA_TRACE_ITERS) trace[((wid * FA_TRACE_ITERS + t) * 6) + (PH)] = wall_clock64(); \
    }
    for (int t = 0; t < ntiles; ++t) {
        const int j0 = t * 64;
        const int cur = t & 1;
        TEO_FA_MARK(0)
        if (t + 1 < ntiles) {
            TEO_FA_WRITE(t + 1, cur ^ 1)                 // that buffer was last read in iteration t-1 (barrier since)
            if (t + 2 < ntiles) TEO_FA_LOAD(t + 2)
        }
        TEO_FA_MARK(1)
        // a wave whose queries all lie before this tile has nothing to add (causal); it still staged and meets the barrier
        if (!CAUSAL || j0 <= wave_qpos_max) {
            const unsigned char* sK = smem + cur * BUF;
            const unsigned char* sV = sK + KT_BYTES;
            // ---- S^T = K . Q^T : 2 key blocks x NKK k-steps
            f32x16 s[2];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
                for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
                const int row = kb * 32 + ql;
                const unsigned char* rp = sK + row * KROW;
                const int sw = (row / RPB) & (CH - 1);
#pragma unroll
                for (int kk = 0; kk < NKK; ++kk) {
                    const fa_bf16x8 kf = __builtin_bit_cast(fa_bf16x8, *reinterpret_cast<const fa_u32x4*>(rp + (((2 * kk + hi) ^ sw) << 4)));
                    s[kb] = mfma32<F16>(kf, qf[kk], s[kb]);
                }
            }
            if constexpr (TRACE) { asm volatile("s_nop 0" :: "v"(s[0][0]), "v"(s[1][0])); }      // the marks below wait for the results
            TEO_FA_MARK(2)
            // lane holds S[key = j0 + kb*32 + (r&3) + 8*(r>>2) + 4*hi][query ql]
            const bool need_mask = (j0 + 64 > a.kv_len) || (CAUSAL && j0 + 63 > wave_qpos_min);
            float tmax = -INFINITY;
            if (need_mask) {
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = j0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                        const bool ok = (key < a.kv_len) && (!CAUSAL || key <= qpos);
                        const float v = ok ? s[kb][r] * sl2 : -INFINITY;
                        s[kb][r] = v;
                        tmax = fmaxf(tmax, v);
                    }
            } else {
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float v = s[kb][r] * sl2;
                        s[kb][r] = v;
                        tmax = fmaxf(tmax, v);
                    }
            }
            tmax = fa_other_half_max(tmax);
            const float m_new = fmaxf(m_run, tmax);
            const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
            float psum = 0.f;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float p = __builtin_amdgcn_exp2f(s[kb][r] - m_use);     // v_exp_f32 (results below 2^-126 flush to 0)
                    psum += p;
                    s[kb][r] = p;
                }
            if (!__all(m_new == m_run)) {                               // some row maximum of the wave moved: rescale
                const float alpha = __builtin_amdgcn_exp2f(m_run - m_use);   // m_run = -inf -> 0
                l_run *= alpha;
#pragma unroll
                for (int i = 0; i < NDB; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc_o[i][r] *= alpha;
            }
            l_run += psum;
            m_run = m_new;
            if constexpr (TRACE) { asm volatile("s_nop 0" :: "v"(l_run), "v"(s[0][15]), "v"(s[1][15])); }
            TEO_FA_MARK(3)
            // ---- O^T += V^T . P^T : 4 blocks of 16 keys x NDB d-blocks
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int kb = u >> 1, r0 = (u & 1) * 8;
                union { fa_bf16x8 v; unsigned w[4]; } pf;
#pragma unroll
                for (int j = 0; j < 4; ++j) pf.w[j] = pack_h2<F16>(s[kb][r0 + 2 * j], s[kb][r0 + 2 * j + 1]);
#pragma unroll
                for (int db = 0; db < NDB; ++db) {
                    const int d = db * 32 + ql;
                    const fa_bf16x8 vf = __builtin_bit_cast(fa_bf16x8, *reinterpret_cast<const fa_u32x4*>(sV + d * VROW + ((2 * u + hi) << 4)));
                    acc_o[db] = mfma32<F16>(vf, pf.v, acc_o[db]);
                }
            }
            if constexpr (TRACE) { asm volatile("s_nop 0" :: "v"(acc_o[0][0]), "v"(acc_o[NDB - 1][15])); }
        }
        TEO_FA_MARK(4)
        __syncthreads();
        TEO_FA_MARK(5)
    }
#undef TEO_FA_MARK
#undef TEO_FA_LOAD
#undef TEO_FA_WRITE
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

static int g_flash_order = 1;      // causal workgroup order: 0 heavy-first, 1 heavy-first with the second dispatch pass mirrored (see the kernel)
void flash_tune_reset() { g_flash_order = 1; }
int flash_tune_set(const char* key, int value) {
    if (!strcmp(key, "flash_order") && value >= 0 && value <= 1) { g_flash_order = value; return 0; }
    return -1;
}

int attention_flash32(const teo_attn_args& a, hipStream_t st, bool f16) {
    dim3 grid(cdiv(a.q_len, 128) * a.heads * a.batch);
    const int cus = device_cu_count();
    const int pair_c = (g_flash_order == 1 && a.causal && cus >= 8) ? cus / 8 : 0;
    const size_t lds = 2 * (size_t)(64 * a.head_dim * 2 + a.head_dim * 144);
#define TEO_FA(DD, CC) { if (f16) attn_flash32_kernel<DD, CC, true><<<grid, 256, lds, st>>>(a, pair_c, nullptr); else attn_flash32_kernel<DD, CC, false><<<grid, 256, lds, st>>>(a, pair_c, nullptr); }
    if (a.head_dim == 128) { if (a.causal) TEO_FA(128, true) else TEO_FA(128, false) }
    else { if (a.causal) TEO_FA(64, true) else TEO_FA(64, false) }
#undef TEO_FA
    note_kernel("attn_flash32"); TEO_LAUNCH_CHECK("attn_flash32");
    return TEO_OK;
}

}  // namespace teo
