// Attention kernels.
//
//  attn_simple_kernel<T>   : one wave per (query, head); scores in LDS, exact two-pass softmax.  fp32 parity path,
//                            fallback for head dims the MFMA kernel does not take, and the on-GPU cross-check of it.
//  attn_mfma_kernel<D,C>   : flash-style bf16 kernel on v_mfma_f32_16x16x32_bf16, D in {64,128}, causal or not.
//                            One workgroup = 64 queries of one head (4 waves x 16 queries), KV tiles of 64 keys staged
//                            in LDS (K row-major XOR-swizzled for ds_read_b128; V^T key-contiguous, swizzled for
//                            ds_read_b64).  The score MFMA is SWAPPED (S^T = K . Q^T): a lane then owns ONE query and
//                            16 keys of the tile, so the online-softmax row max/sum is an in-lane reduction plus two
//                            cross-lane steps, the rescale factor is a per-lane scalar, and the exponentiated tile
//                            is already in the B-operand layout of O^T += V^T . P^T -- no LDS round trip for P.
//  attn_decode_*           : q_len == 1 (decode).  Split over the KV length (grid heads x splits, fixed for hipGraph
//                            replay; kv_len comes from device memory), then a combine kernel.  HBM-bound:
//                            algorithmic bytes = 2 * kv_heads * kv_len * head_dim * sizeof(T) per layer.
#include "common.h"
#include "ops.h"

namespace teo {

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

// ------------------------------------------------------------------------------------------------
// generic kernel
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(64) void attn_simple_kernel(teo_attn_args a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* qs = sm;                   // [head_dim]
    float* sc = sm + a.head_dim;      // [kv_len]
    const int i = blockIdx.x, h = blockIdx.y, b = blockIdx.z, lane = threadIdx.x;
    const int hk = h / (a.heads / a.kv_heads);
    const int hd = a.head_dim;
    const T* q = (const T*)a.q + b * a.q_bs + h * a.q_hs + (long long)i * a.q_rs;
    const T* k = (const T*)a.k + b * a.k_bs + hk * a.k_hs;
    const T* v = (const T*)a.v + b * a.v_bs + hk * a.v_hs;
    for (int d = lane; d < hd; d += 64) qs[d] = Elem<T>::ld(q + d);
    __syncthreads();
    const int lim = a.causal ? min(a.kv_len, i + (a.kv_len - a.q_len) + 1) : a.kv_len;
    float mx = -INFINITY;
    for (int j = lane; j < lim; j += 64) {
        const T* kr = k + (long long)j * a.k_rs;
        float s = 0.f;
        for (int d = 0; d < hd; ++d) s = fmaf(qs[d], Elem<T>::ld(kr + d), s);
        s *= a.scale;
        sc[j] = s;
        mx = fmaxf(mx, s);
    }
    mx = wave_max(mx);
    float sum = 0.f;
    for (int j = lane; j < lim; j += 64) {
        const float p = expf(sc[j] - mx);
        sum += p;
        sc[j] = Elem<T>::round(p);    // P is fed to the PV product in the storage type
    }
    sum = wave_sum(sum);
    __syncthreads();
    T* o = (T*)a.o + b * a.o_bs + (long long)i * a.o_rs + h * hd;
    for (int d = lane; d < hd; d += 64) {
        float acc = 0.f;
        for (int j = 0; j < lim; ++j) acc = fmaf(sc[j], Elem<T>::ld(v + (long long)j * a.v_rs + d), acc);
        Elem<T>::st(o + d, acc / sum);
    }
}

// ------------------------------------------------------------------------------------------------
// MFMA flash kernel
// ------------------------------------------------------------------------------------------------
template <int D, bool CAUSAL>
__global__ __launch_bounds__(256) void attn_mfma_kernel(teo_attn_args a) {
    constexpr int CH = D / 8;              // 16-byte chunks per K row
    constexpr int KROW = D * 2;            // bytes per K row in LDS
    constexpr int KT_BYTES = 64 * KROW;    // K tile
    constexpr int NDF = D / 16;            // d-fragments of the output
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sK = smem;              // [64 keys][D] bf16, chunk c of row r at c ^ (r & (CH-1))
    unsigned char* sV = smem + KT_BYTES;   // [D][64 keys] bf16 (128 B rows), 16-B chunk c of row d at c ^ ((d>>1)&7)

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int h = blockIdx.y, b = blockIdx.z;
    const int hk = h / (a.heads / a.kv_heads);
    const int qb = blockIdx.x * 64;
    const int off = a.kv_len - a.q_len;
    const bf16_t* Q = (const bf16_t*)a.q + b * a.q_bs + h * a.q_hs;
    const bf16_t* K = (const bf16_t*)a.k + b * a.k_bs + hk * a.k_hs;
    const bf16_t* VT = (const bf16_t*)a.vt + b * a.vt_bs + hk * a.vt_hs;

    // this lane's query (B operand of the swapped score MFMA): Q[q][kk*32 + fg*8 .. +8]
    const int qi = qb + wid * 16 + fr;
    const int qrow = min(qi, a.q_len - 1);
    bf16x8 qf[D / 32];
#pragma unroll
    for (int kk = 0; kk < D / 32; ++kk)
        qf[kk] = *reinterpret_cast<const bf16x8*>(Q + (long long)qrow * a.q_rs + kk * 32 + fg * 8);
    const int qpos = qi + off;             // last visible key (causal)

    f32x4 acc_o[NDF];
#pragma unroll
    for (int i = 0; i < NDF; ++i) acc_o[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;
    const float sl2 = a.scale * 1.44269504088896340736f;   // scores in log2 units

    int kv_end = a.kv_len;
    if (CAUSAL) kv_end = min(a.kv_len, qb + 63 + off + 1);
    const int ntiles = (kv_end + 63) >> 6;

    // Register prefetch: tile t+1 is requested from global memory before tile t is computed, so the HBM/L2 latency of
    // the K / V^T stream hides under the MFMAs (the loads are unconditional: the tile index is clamped).
    constexpr int NKC = (64 * CH) / 256, NVC = (D * 8) / 256;
    u32x4 rk[NKC], rv[NVC];
#define TEO_FA_LOAD(T_)                                                                                           \
    {                                                                                                             \
        const int jt = (T_) * 64;                                                                                 \
        _Pragma("unroll") for (int i = 0; i < NKC; ++i) {                                                         \
            const int id = tid + 256 * i;                                                                         \
            const int r = id / CH, c = id % CH;                                                                   \
            const int gj = min(jt + r, a.kv_len - 1);                                                             \
            rk[i] = *reinterpret_cast<const u32x4*>(K + (long long)gj * a.k_rs + c * 8);                          \
        }                                                                                                         \
        _Pragma("unroll") for (int i = 0; i < NVC; ++i) {                                                         \
            const int id = tid + 256 * i;                                                                         \
            const int d = id >> 3, c = id & 7;                                                                    \
            rv[i] = *reinterpret_cast<const u32x4*>(VT + (long long)d * a.vt_rs + jt + c * 8);                    \
        }                                                                                                         \
    }
    TEO_FA_LOAD(0)
    for (int t = 0; t < ntiles; ++t) {
        const int j0 = t * 64;
        __syncthreads();   // previous tile fully consumed
        // ---- stage K tile: 64 rows x CH chunks
#pragma unroll
        for (int i = 0; i < NKC; ++i) {
            const int id = tid + 256 * i;
            const int r = id / CH, c = id % CH;
            *reinterpret_cast<u32x4*>(sK + r * KROW + ((c ^ (r & (CH - 1))) << 4)) = rk[i];
        }
        // ---- stage V^T tile: D rows x 64 keys; keys >= kv_len must read as zero.  Inside each 32-key block the keys
        // are stored in the order the P operand of the PV MFMA holds them (lane group fg owns keys fg*4..+4 and
        // 16+fg*4..+4), so the MFMA reads one 16-byte chunk per lane: global chunk cc of a block -> 8-byte units
        // (cc%2)*4 + cc/2 and that + 2.
#pragma unroll
        for (int i = 0; i < NVC; ++i) {
            const int id = tid + 256 * i;
            const int d = id >> 3, c = id & 7;
            u32x4 val = rv[i];
            const int valid = a.kv_len - (j0 + c * 8);     // number of valid keys in this chunk
            if (valid < 8) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (2 * e >= valid) val[e] = 0u;
                    else if (2 * e + 1 >= valid) val[e] &= 0xffffu;
                }
            }
            const int kb = c >> 2, cc = c & 3;
            const int u0 = (cc & 1) * 4 + (cc >> 1), u1 = u0 + 2;
            unsigned char* rowp = sV + d * 128;
            const int sw = d & 7;
            *reinterpret_cast<uint2*>(rowp + (((kb * 4 + (u0 >> 1)) ^ sw) << 4) + ((u0 & 1) << 3)) = make_uint2(val[0], val[1]);
            *reinterpret_cast<uint2*>(rowp + (((kb * 4 + (u1 >> 1)) ^ sw) << 4) + ((u1 & 1) << 3)) = make_uint2(val[2], val[3]);
        }
        __syncthreads();
        if (t + 1 < ntiles) TEO_FA_LOAD(t + 1)

        // ---- S^T = K . Q^T : 4 key fragments x (D/32) k-steps
        f32x4 s[4];
#pragma unroll
        for (int f = 0; f < 4; ++f) s[f] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < D / 32; ++kk) {
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                const int r = f * 16 + fr;
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(sK + r * KROW + (((kk * 4 + fg) ^ (r & (CH - 1))) << 4));
                s[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[kk], s[f], 0, 0, 0);
            }
        }
        // lane holds S[key = j0 + f*16 + fg*4 + r][query = fr]
        float tmax = -INFINITY;
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = j0 + f * 16 + fg * 4 + r;
                float v = s[f][r] * sl2;
                const bool ok = (key < a.kv_len) && (!CAUSAL || key <= qpos);
                v = ok ? v : -INFINITY;
                s[f][r] = v;
                tmax = fmaxf(tmax, v);
            }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float m_new = fmaxf(m_run, tmax);
        const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
        const float alpha = exp2f(m_run - m_use);           // m_run = -inf -> 0
        float psum = 0.f;
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = exp2f(s[f][r] - m_use);
                psum += p;
                s[f][r] = p;
            }
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int i = 0; i < NDF; ++i) {
            acc_o[i][0] *= alpha; acc_o[i][1] *= alpha; acc_o[i][2] *= alpha; acc_o[i][3] *= alpha;
        }
        // ---- O^T += V^T . P^T : 2 key blocks of 32 x NDF d-fragments
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            union { bf16x8 v; unsigned u[4]; } pf;
            pf.u[0] = pack_bf2(s[2 * kb][0], s[2 * kb][1]);
            pf.u[1] = pack_bf2(s[2 * kb][2], s[2 * kb][3]);
            pf.u[2] = pack_bf2(s[2 * kb + 1][0], s[2 * kb + 1][1]);
            pf.u[3] = pack_bf2(s[2 * kb + 1][2], s[2 * kb + 1][3]);
#pragma unroll
            for (int df = 0; df < NDF; ++df) {
                const int d = df * 16 + fr;
                union { bf16x8 v; unsigned u[4]; } vf;
                vf.v = *reinterpret_cast<const bf16x8*>(sV + d * 128 + (((kb * 4 + fg) ^ (d & 7)) << 4));
                acc_o[df] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf.v, pf.v, acc_o[df], 0, 0, 0);
            }
        }
    }
#undef TEO_FA_LOAD
    // ---- finish: l over the four lanes that share a query, normalise, store O[q][h*D + df*16 + fg*4 + r]
    l_run += __shfl_xor(l_run, 16, 64);
    l_run += __shfl_xor(l_run, 32, 64);
    if (qi < a.q_len) {
        const float inv = 1.0f / l_run;
        bf16_t* o = (bf16_t*)a.o + b * a.o_bs + (long long)qi * a.o_rs + h * D;
#pragma unroll
        for (int df = 0; df < NDF; ++df) {
            const uint2 pk = make_uint2(pack_bf2(acc_o[df][0] * inv, acc_o[df][1] * inv),
                                        pack_bf2(acc_o[df][2] * inv, acc_o[df][3] * inv));
            *reinterpret_cast<uint2*>(o + df * 16 + fg * 4) = pk;
        }
    }
}

static int g_attn_flash = 1;      // 1: attn_flash32_kernel (flash.hip, 32x32x16 MFMA, 128 queries per workgroup); 0: the round-1 kernel

bool attn_mfma_ok(const teo_attn_args& a, int dtype) {
    if (dtype != TEO_BF16 || (a.flags & TEO_ATTN_FORCE_SIMPLE) || a.vt == nullptr) return false;
    if (a.head_dim != 64 && a.head_dim != 128) return false;
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    if (!al16(a.q) || !al16(a.k) || !al16(a.vt) || (reinterpret_cast<uintptr_t>(a.o) & 7)) return false;
    if (a.q_rs % 8 || a.q_hs % 8 || a.q_bs % 8 || a.k_rs % 8 || a.k_hs % 8 || a.k_bs % 8) return false;
    if (a.vt_rs % 8 || a.vt_hs % 8 || a.vt_bs % 8 || a.o_rs % 4 || a.o_bs % 4) return false;
    // V^T rows are read in whole 64-key tiles
    if (a.vt_rs < (long long)((a.kv_len + 63) / 64) * 64) return false;
    return true;
}

int attention(const teo_attn_args* ap, int dtype, hipStream_t st) {
    const teo_attn_args& a = *ap;
    if (a.q_len == 0 || a.batch == 0) return TEO_OK;
    TEO_CHECK_ARG(a.heads % a.kv_heads == 0, "teo_attention: heads %d not a multiple of kv_heads %d", a.heads, a.kv_heads);
    TEO_CHECK_ARG(a.kv_len >= a.q_len || !a.causal, "teo_attention: causal needs kv_len >= q_len");
    if (attn_mfma_ok(a, dtype) && g_attn_flash) return attention_flash32(a, st);
    if (attn_mfma_ok(a, dtype)) {
        dim3 grid(cdiv(a.q_len, 64), a.heads, a.batch);
        const size_t lds = 64 * a.head_dim * 2 + a.head_dim * 128;
        if (a.head_dim == 128) {
            if (a.causal) attn_mfma_kernel<128, true><<<grid, 256, lds, st>>>(a);
            else attn_mfma_kernel<128, false><<<grid, 256, lds, st>>>(a);
        } else {
            if (a.causal) attn_mfma_kernel<64, true><<<grid, 256, lds, st>>>(a);
            else attn_mfma_kernel<64, false><<<grid, 256, lds, st>>>(a);
        }
        note_kernel("attn_mfma"); TEO_LAUNCH_CHECK("attn_mfma");
        return TEO_OK;
    }
    TEO_CHECK_ARG(a.v != nullptr, "teo_attention: generic kernel needs row-major V");
    const size_t lds = (size_t)(a.head_dim + a.kv_len) * sizeof(float);
    if (lds > 64 * 1024) {
        set_error("teo_attention: generic kernel supports kv_len + head_dim <= 16384 (got %d)", a.kv_len + a.head_dim);
        return TEO_ERR_UNSUPPORTED;
    }
    dim3 grid(a.q_len, a.heads, a.batch);
    if (dtype == TEO_F32) attn_simple_kernel<float><<<grid, 64, lds, st>>>(a);
    else attn_simple_kernel<bf16_t><<<grid, 64, lds, st>>>(a);
    note_kernel("attn_simple"); TEO_LAUNCH_CHECK("attn_simple");
    return TEO_OK;
}

// ------------------------------------------------------------------------------------------------
// decode: one query row, kv_len read from device memory (hipGraph replays the same launch each token)
//
// grid (heads, nsplit), 256 threads; a workgroup owns DEC_CHUNK keys, each wave a quarter of them.  Rows of K and V
// are streamed with 16-byte loads, LPR = head_dim*sizeof(T)/16 lanes per row; ALL loads of the chunk (K and V) are
// issued before the first use so one lane has 2*DEC_CHUNK/4/(64/LPR) loads in flight (16 for bf16, d=128).
// scores: per-row partial dot + xor-shuffle over the LPR lanes; PV: each lane accumulates its 16-byte column slice
// over its keys, lanes of different rows are summed with two xor-shuffles at the end.  No MFMA: 1 query row.
// ------------------------------------------------------------------------------------------------
static int g_dec_chunk = 0;       // keys per workgroup (32 / 64 / 128 / 256); 0 = auto: 64 for one conversation (2.83 vs 2.86 ms/token
                                  // at 128, 2.87 at 32, 2.94 at 256), 128 for a batched step (4.74 vs 4.79 ms/step at 64)
static int g_fused_combine = 0;   // 1: the last workgroup of a head merges the KV splits (no combine launch).  Measured: the
                                  // agent-scope release/acquire fences cost far more than the launch they save (2.89 -> 3.51
                                  // ms/token; batch 8: 5.2 -> 13 ms/step), so it stays off -- kept as a tested experiment.
int g_rope_in_attn = -1;          // decode RoPE + KV append: 0 = in the QKV GEMV epilogue, 1 = inside the attention kernel,
                                   // -1 = auto = 0.  Measured end to end on one box: bf16 weights 2.926 vs 2.995 ms/token in favour of 0;
                                   // fp8 weights were 2.216 vs 2.234 in favour of 1 until the fp8 QKV+RoPE GEMV got the small prologue
                                   // and 4 chunks per step (round 3): now 1.815 vs 1.856 in favour of 0 as well
static int g_attn_fat = 0;        // decode attention for bf16 / head_dim 128: 1 = the fat-split kernel of attn_fat.hip + record merge in the o-projection
                                  // GEMV (no combine launch).  Measured on MI355X (round 3, profiles/r03_decode_attention_ab.md): a tie at
                                  // ctx 2300 (2.713 vs 2.706 ms/token), slower at ctx 700 (2.643 vs 2.582) and 4250 (2.957 vs 2.886) -- off.
bool attn_fat_enabled() { return g_attn_fat != 0 && g_fused_combine == 0; }
int attn_tune_set(const char* key, int value) {
    if (!strcmp(key, "attn_fat")) { g_attn_fat = value != 0; return 0; }
    if (!strcmp(key, "attn_flash")) { g_attn_flash = value != 0; return 0; }
    if (!strcmp(key, "attn_chunk") && (value == 0 || value == 32 || value == 64 || value == 128 || value == 256)) { g_dec_chunk = value; return 0; }
    if (!strcmp(key, "attn_fused_combine")) { g_fused_combine = value != 0; return 0; }
    if (!strcmp(key, "rope_in_attn") && (value >= -1 && value <= 1)) { g_rope_in_attn = value; return 0; }
    return -1;
}

template <typename T> struct Cvt16;
template <> struct Cvt16<bf16_t> {
    static constexpr int N = 8;
    __device__ static __forceinline__ void cvt(const uint4& r, float* f) {
        f[0] = __uint_as_float(r.x << 16); f[1] = __uint_as_float(r.x & 0xffff0000u);
        f[2] = __uint_as_float(r.y << 16); f[3] = __uint_as_float(r.y & 0xffff0000u);
        f[4] = __uint_as_float(r.z << 16); f[5] = __uint_as_float(r.z & 0xffff0000u);
        f[6] = __uint_as_float(r.w << 16); f[7] = __uint_as_float(r.w & 0xffff0000u);
    }
    __device__ static __forceinline__ uint4 pack(const float* f) {
        return make_uint4(pack_bf2(f[0], f[1]), pack_bf2(f[2], f[3]), pack_bf2(f[4], f[5]), pack_bf2(f[6], f[7]));
    }
};
template <> struct Cvt16<float> {
    static constexpr int N = 4;
    __device__ static __forceinline__ void cvt(const uint4& r, float* f) {
        f[0] = __uint_as_float(r.x); f[1] = __uint_as_float(r.y); f[2] = __uint_as_float(r.z); f[3] = __uint_as_float(r.w);
    }
    __device__ static __forceinline__ uint4 pack(const float* f) {
        return make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3]));
    }
};

typedef __bf16 attn_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float attn_dot2(unsigned a, unsigned b, float acc) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(attn_bf16x2, a), __builtin_bit_cast(attn_bf16x2, b), acc, false);
}

// K/V rows are read once per step: non-temporal 16-byte loads (streamed past L2 like the GEMV weight stream)
typedef __attribute__((ext_vector_type(4))) unsigned int kv_u32x4;
__device__ __forceinline__ uint4 ld_kv(const void* p) {
    const kv_u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const kv_u32x4*>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}

// q: [heads*hd] (already rotated), K/V cache [kv_heads][S_max][hd]; partial: [heads][nsplit][hd + 2] fp32 (m, l, o[hd])
// ROPE: `q` is the raw [q | k | v] row of the new token (GEMV output, not yet rotated).  The kernel rotates q on load,
// and the one lane group that owns key `pos` rotates the new k, takes the new v, appends both to the caches (K, V, V^T)
// and uses them directly -- RoPE + KV append cost no launch and no pass of their own.
template <typename T, int LPR, int DEC_CHUNK, bool ROPE>
__global__ __launch_bounds__(256) void attn_decode_partial_kernel(const T* __restrict__ q, T* __restrict__ kc,
                                                                  T* __restrict__ vc, T* __restrict__ vtc,
                                                                  const float* __restrict__ cs, const float* __restrict__ sn,
                                                                  float* __restrict__ part, const int* __restrict__ d_pos,
                                                                  int S_max, int heads, int kv_heads, float scale, int nsplit,
                                                                  AttnBatch bt, int* __restrict__ counters, T* __restrict__ o_out) {
    constexpr int VE = Cvt16<T>::N;
    constexpr int HD = LPR * VE;
    {   // conversation blockIdx.z of a batched step: its own query row, caches, position and partial slab
        const long long bz = blockIdx.z;
        if (counters) { counters += bz * heads; o_out += bz * bt.o_stride; }
        q += bz * bt.q_stride;
        kc += bz * bt.cache_stride;
        vc += bz * bt.cache_stride;
        if (vtc) vtc += bz * bt.cache_stride;
        d_pos += bz;
        part += bz * (long long)heads * nsplit * (HD + 2);
    }
    constexpr int RPI = 64 / LPR;                       // rows (keys) per wave-wide load instruction
    constexpr int KPW = DEC_CHUNK / 4;                  // keys per wave
    constexpr int NI = KPW / RPI;                       // load instructions per wave per operand
    __shared__ float sc[DEC_CHUNK];
    __shared__ float red[8];
    __shared__ float obuf[4][HD];
    const int h = blockIdx.x, sp = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int hk = h / (heads / kv_heads);
    const int kv_len = *d_pos + 1;
    const int c0 = sp * DEC_CHUNK;
    float* out = part + ((long long)h * nsplit + sp) * (HD + 2);
    if (c0 >= kv_len) {                                 // nothing here: neutral partial
        if (counters) return;                           // fused combine only looks at the splits that hold keys
        if (tid == 0) { out[0] = -INFINITY; out[1] = 0.f; }
        for (int d = tid; d < HD; d += 256) out[2 + d] = 0.f;
        return;
    }
    const int sub = lane % LPR, grp = lane / LPR;
    const T* kb = kc + (long long)hk * S_max * HD + sub * VE;
    const T* vb = vc + (long long)hk * S_max * HD + sub * VE;
    const int kw0 = c0 + wid * KPW;                     // first key of this wave
    uint4 kr[NI], vr[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int j = min(kw0 + i * RPI + grp, kv_len - 1);
        kr[i] = ld_kv(kb + (long long)j * HD);
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int j = min(kw0 + i * RPI + grp, kv_len - 1);
        vr[i] = ld_kv(vb + (long long)j * HD);
    }
    float qf[VE];
    float knew[VE], vnew[VE];
    const int pos = kv_len - 1;
    if (ROPE) {
        // rotate-half RoPE on 16-byte chunks: this lane owns elements sub*VE..+VE of the head, its partner (sub ^ LPR/2)
        // the matching elements of the other half; coefficient index = element index mod hd/2
        constexpr int HL = LPR / 2;
        const int psub = sub ^ HL, ci = (sub % HL) * VE;
        const float sgn = (sub < HL) ? -1.f : 1.f;
        float cf[VE], sf[VE];
#pragma unroll
        for (int e = 0; e < VE; e += 4) {
            const float4 c4 = *reinterpret_cast<const float4*>(cs + (long long)pos * (HD / 2) + ci + e);
            const float4 s4 = *reinterpret_cast<const float4*>(sn + (long long)pos * (HD / 2) + ci + e);
            cf[e] = c4.x; cf[e + 1] = c4.y; cf[e + 2] = c4.z; cf[e + 3] = c4.w;
            sf[e] = s4.x; sf[e + 1] = s4.y; sf[e + 2] = s4.z; sf[e + 3] = s4.w;
        }
        float own[VE], oth[VE];
        Cvt16<T>::cvt(*reinterpret_cast<const uint4*>(q + h * HD + sub * VE), own);
        Cvt16<T>::cvt(*reinterpret_cast<const uint4*>(q + h * HD + psub * VE), oth);
#pragma unroll
        for (int e = 0; e < VE; ++e) qf[e] = Elem<T>::round(own[e] * cf[e] + sgn * oth[e] * sf[e]);
        const T* kraw = q + (long long)(heads + hk) * HD;
        Cvt16<T>::cvt(*reinterpret_cast<const uint4*>(kraw + sub * VE), own);
        Cvt16<T>::cvt(*reinterpret_cast<const uint4*>(kraw + psub * VE), oth);
#pragma unroll
        for (int e = 0; e < VE; ++e) knew[e] = Elem<T>::round(own[e] * cf[e] + sgn * oth[e] * sf[e]);
        Cvt16<T>::cvt(*reinterpret_cast<const uint4*>(q + (long long)(heads + kv_heads + hk) * HD + sub * VE), vnew);
    } else {
        const uint4 qraw = *reinterpret_cast<const uint4*>(q + h * HD + sub * VE);
        Cvt16<T>::cvt(qraw, qf);
    }
    // bf16: q (bf16-exact after its rounding) stays packed and meets the raw key chunk through v_dot2c_f32_bf16
    // (4 instructions per 8 elements instead of 8 unpacks + 8 FMAs)
    constexpr bool DOT2 = sizeof(T) == 2;
    uint4 qpk = make_uint4(0, 0, 0, 0);
    if (DOT2) qpk = Cvt16<T>::pack(qf);
    // ---- scores
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        float kf[VE];
        uint4 kraw = kr[i];
        const int j = kw0 + i * RPI + grp;
        if (!DOT2) Cvt16<T>::cvt(kr[i], kf);
        if (ROPE && j == pos) {                         // the new token's key: not in the cache yet
            if (DOT2) kraw = Cvt16<T>::pack(knew);
#pragma unroll
            for (int e = 0; e < VE; ++e) kf[e] = knew[e];
            if (h % (heads / kv_heads) == 0) {          // one q head per kv head appends
                *reinterpret_cast<uint4*>(kc + ((long long)hk * S_max + pos) * HD + sub * VE) = Cvt16<T>::pack(knew);
                *reinterpret_cast<uint4*>(vc + ((long long)hk * S_max + pos) * HD + sub * VE) = Cvt16<T>::pack(vnew);
                if (vtc) {
                    const uint4 pv = Cvt16<T>::pack(vnew);
                    const T* pe = reinterpret_cast<const T*>(&pv);
#pragma unroll
                    for (int e = 0; e < VE; ++e) vtc[((long long)hk * HD + sub * VE + e) * S_max + pos] = pe[e];
                }
            }
        }
        float s = 0.f;
        if (DOT2) {
            s = attn_dot2(kraw.x, qpk.x, s); s = attn_dot2(kraw.y, qpk.y, s);
            s = attn_dot2(kraw.z, qpk.z, s); s = attn_dot2(kraw.w, qpk.w, s);
        } else {
#pragma unroll
            for (int e = 0; e < VE; ++e) s = fmaf(qf[e], kf[e], s);
        }
#pragma unroll
        for (int o = LPR >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (sub == 0) sc[j - c0] = (j < kv_len) ? s * scale : -INFINITY;
    }
    __syncthreads();
    // ---- chunk max / exp / sum (every wave redundantly over the chunk's scores: DEC_CHUNK/64 per lane)
    constexpr int SPL = (DEC_CHUNK + 63) / 64;
    float sv[SPL];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < SPL; ++i) {
        sv[i] = (DEC_CHUNK >= 64 || lane + 64 * i < DEC_CHUNK) ? sc[(lane + 64 * i) % DEC_CHUNK] : -INFINITY;
        mx = fmaxf(mx, sv[i]);
    }
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < SPL; ++i) { sv[i] = expf(sv[i] - mx); sum += sv[i]; }      // -inf -> 0
    sum = wave_sum(sum);
    __syncthreads();
    if (wid == 0) {
#pragma unroll
        for (int i = 0; i < SPL; ++i)
            if (DEC_CHUNK >= 64 || lane + 64 * i < DEC_CHUNK) sc[lane + 64 * i] = Elem<T>::round(sv[i]);
    }
    __syncthreads();
    // ---- PV on this wave's keys
    float acc[VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) acc[e] = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const float p = sc[wid * KPW + i * RPI + grp];           // 0 for keys >= kv_len
        float vf[VE];
        Cvt16<T>::cvt(vr[i], vf);
        if (ROPE && kw0 + i * RPI + grp == pos) {
#pragma unroll
            for (int e = 0; e < VE; ++e) vf[e] = vnew[e];
        }
#pragma unroll
        for (int e = 0; e < VE; ++e) acc[e] = fmaf(p, vf[e], acc[e]);
    }
#pragma unroll
    for (int o = LPR; o < 64; o <<= 1) {
#pragma unroll
        for (int e = 0; e < VE; ++e) acc[e] += __shfl_xor(acc[e], o, 64);
    }
    if (grp == 0) {
#pragma unroll
        for (int e = 0; e < VE; ++e) obuf[wid][sub * VE + e] = acc[e];
    }
    __syncthreads();
    if (!counters) {
        for (int d = tid; d < HD; d += 256) out[2 + d] = obuf[0][d] + obuf[1][d] + obuf[2][d] + obuf[3][d];
        if (tid == 0) { out[0] = mx; out[1] = sum; }
        return;
    }
    // ---- fused combine: the LAST workgroup of this head to finish merges the splits (no combine launch, no fences).
    // Hand-off protocol (MI355X: per-XCD L2s are not coherent, a CU's L1 is never refreshed by other CUs' stores):
    //   producer : the partial record is stored WRITE-THROUGH (relaxed agent-scope atomic stores = `global_store ... sc1`),
    //              every storing wave drains its stores (`s_waitcnt vmcnt(0)`), the workgroup meets at a barrier, ONE lane
    //              takes a ticket with a relaxed agent-scope fetch_add;
    //   consumer : the workgroup that draws the last ticket reads every record with sc1 loads (relaxed agent-scope atomic
    //              loads: they bypass this CU's L1 and are coherent with the write-through stores of the other XCDs).
    // No release/acquire fence anywhere: an agent-scope release writes back the XCD's whole dirty L2 (that is what made the
    // round-1 form of this path 20 % slower than a separate combine launch).  The records are re-used every token, so a
    // stale line would show up at once: the result is compared bit for bit with the two-launch path in the tests (the merge
    // below adds the splits in exactly the order attn_decode_combine_kernel does).
    for (int d = tid; d < HD; d += 256)
        __hip_atomic_store(out + 2 + d, obuf[0][d] + obuf[1][d] + obuf[2][d] + obuf[3][d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid == 0) {
        __hip_atomic_store(out, mx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(out + 1, sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // EVERY storing wave: its write-through stores have landed
    __shared__ int s_last;
    __shared__ float wgt[256];
    __shared__ float accs[512];
    __syncthreads();
    const int nact = (kv_len + DEC_CHUNK - 1) / DEC_CHUNK;
    if (tid == 0) {
        const int ticket = __hip_atomic_fetch_add(&counters[h], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = ticket == nact - 1;
        if (last) __hip_atomic_store(&counters[h], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // re-armed for the next launch
        s_last = last;
    }
    __syncthreads();
    if (!s_last) return;
    const int stride = HD + 2;
    const float* pb = part + (long long)h * nsplit * stride;
#define TEO_LD_SC1(ptr) __hip_atomic_load((ptr), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
    // Same arithmetic, same order as attn_decode_combine_kernel (G = 512 / HD split groups per column, group g sums the
    // splits g, g + G, ... in order, the G group sums are added in order): a thread here carries two of those groups.
    constexpr int G = 512 / HD;                                  // HD is a power of two in [32, 256] -> G in [2, 16]
    constexpr int GP = G / (256 / HD) > 0 ? G / (256 / HD) : 1;  // groups per thread (256 threads cover 256 / HD groups at once)
    const int dcol = tid % HD, g0 = tid / HD;                    // thread owns groups g0 + j * (256 / HD), j < GP
    float v0[GP][8];
#pragma unroll
    for (int j = 0; j < GP; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i)
            v0[j][i] = TEO_LD_SC1(pb + (long long)min(g0 + j * (256 / HD) + i * G, nact - 1) * stride + 2 + dcol);
    float m0 = -INFINITY, l0 = 0.f;
    if (tid < nact) { m0 = TEO_LD_SC1(pb + tid * stride); l0 = TEO_LD_SC1(pb + tid * stride + 1); }
    float M = wave_max(m0);
    if (lane == 0) red[wid] = M;
    __syncthreads();
    M = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float w0 = (m0 == -INFINITY) ? 0.f : expf(m0 - M);
    wgt[tid] = w0;
    float Ls = wave_sum(l0 * w0);
    if (lane == 0) red[4 + wid] = Ls;
    __syncthreads();
    const float inv = 1.0f / ((red[4] + red[5]) + (red[6] + red[7]));
#pragma unroll
    for (int j = 0; j < GP; ++j) {
        const int g = g0 + j * (256 / HD);
        float a = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) a += (g + i * G < nact) ? v0[j][i] * wgt[min(g + i * G, 255)] : 0.f;
        for (int s0 = g + 8 * G; s0 < nact; s0 += 8 * G) {
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = TEO_LD_SC1(pb + (long long)min(s0 + i * G, nact - 1) * stride + 2 + dcol);
#pragma unroll
            for (int i = 0; i < 8; ++i) a += (s0 + i * G < nact) ? v[i] * wgt[min(s0 + i * G, 255)] : 0.f;
        }
        accs[g * HD + dcol] = a;
    }
#undef TEO_LD_SC1
    __syncthreads();
    if (tid < HD) {
        float t = 0.f;
        for (int k = 0; k < G; ++k) t += accs[tid + k * HD];
        Elem<T>::st(o_out + h * HD + tid, t * inv);
    }
}

// one workgroup (512 threads) per head.  Split weights: one thread per split (parallel loads).  Output: thread =
// (column d, split group g); a thread owns every G-th split (G = 512 / hd) and keeps 8 loads in flight, the G partial
// sums of a column meet in LDS -- one or two L2 round trips instead of a dependent chain over all splits.
template <typename T>
__global__ __launch_bounds__(512) void attn_decode_combine_kernel(const float* __restrict__ part, T* __restrict__ o,
                                                                  const int* __restrict__ d_pos, int hd, int nsplit, int chunk,
                                                                  long long o_stride) {
    __shared__ float w[256];
    __shared__ float red[16];
    __shared__ float accs[512];
    const int h = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int stride = hd + 2;
    part += (long long)blockIdx.y * gridDim.x * nsplit * stride;
    o += (long long)blockIdx.y * o_stride;
    d_pos += blockIdx.y;
    const float* pb = part + (long long)h * nsplit * stride;
    const int nact = min(nsplit, (*d_pos + 1 + chunk - 1) / chunk);       // splits that hold keys (<= 256)
    const int G = 512 / hd;                                                 // hd is a power of two <= 256
    const int g = tid / hd, d = tid % hd;
    // the first NB splits of this thread are requested together with the split statistics (they do not depend on them): ONE round
    // trip covers NB * G splits (48 at head_dim 128 = ctx 3072 at 64 keys per split; round 2 had NB = 8 and a second, dependent
    // batch from 33 splits on, i.e. for every C3 step)
    constexpr int NB = 12;
    float v0[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) v0[i] = pb[(long long)min(g + i * G, nact - 1) * stride + 2 + d];
    float m0 = -INFINITY, l0 = 0.f;
    if (tid < nact) { m0 = pb[tid * stride]; l0 = pb[tid * stride + 1]; }
    float M = wave_max(m0);
    if (lane == 0) red[wid] = M;
    __syncthreads();
    M = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));               // splits live in threads 0..255 = waves 0..3
    const float w0 = (m0 == -INFINITY) ? 0.f : expf(m0 - M);
    if (tid < 256) w[tid] = w0;
    float L = wave_sum(l0 * w0);
    if (lane == 0) red[8 + wid] = L;
    __syncthreads();
    const float inv = 1.0f / ((red[8] + red[9]) + (red[10] + red[11]));
    float a = 0.f;
#pragma unroll
    for (int i = 0; i < NB; ++i) a += (g + i * G < nact) ? v0[i] * w[min(g + i * G, 255)] : 0.f;
    for (int s0 = g + NB * G; s0 < nact; s0 += 8 * G) {
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = pb[(long long)min(s0 + i * G, nact - 1) * stride + 2 + d];
#pragma unroll
        for (int i = 0; i < 8; ++i) a += (s0 + i * G < nact) ? v[i] * w[min(s0 + i * G, 255)] : 0.f;
    }
    accs[tid] = a;
    __syncthreads();
    if (g == 0) {
        float t = 0.f;
        for (int k = 0; k < G; ++k) t += accs[d + k * hd];
        Elem<T>::st(o + h * hd + d, t * inv);
    }
}

size_t attn_decode_ws_bytes(int heads, int hd, int S_max, int batch) {
    const int nsplit = cdiv(S_max, 32);      // sized for the smallest chunk
    const size_t small = (size_t)batch * heads * nsplit * (hd + 2) * sizeof(float);
    const size_t fat = (size_t)batch * heads * ATTN_FAT_MAX_SPLITS * ATTN_FAT_REC * sizeof(float);      // attn_fat.hip records
    return small > fat ? small : fat;
}
// the exported primitive (teo_attn_decode) keeps the arrival counters of the fused combine behind the partial records
size_t attn_decode_counters_offset(int heads, int hd, int S_max, int batch) {
    return (attn_decode_ws_bytes(heads, hd, S_max, batch) + 255) / 256 * 256;
}
bool attn_decode_fused_enabled() { return g_fused_combine != 0; }

template <typename T, int LPR>
static void attn_decode_launch(const void* q, void* kc, void* vc, void* vtc, const float* cs, const float* sn, void* o,
                               float* part, const int* d_pos, int S_max, int heads, int kv_heads, int hd, float scale,
                               int nsplit, int chunk, bool rope, AttnBatch bt, int* counters, hipStream_t st) {
    dim3 grid(heads, nsplit, bt.batch);
#define TEO_PART(CH, RP)                                                                                              \
    TEO_KLAUNCH((attn_decode_partial_kernel<T, LPR, CH, RP>), grid, 256, 0, st, (const T*)q, (T*)kc, (T*)vc, (T*)vtc, cs, sn, part, \
                                                                     d_pos, S_max, heads, kv_heads, scale, nsplit, bt, \
                                                                     counters, (T*)o)
#define TEO_PART_R(CH) if (rope) { TEO_PART(CH, true); } else { TEO_PART(CH, false); }
    if constexpr (32 / 4 >= 64 / LPR) {
        if (chunk == 32) { TEO_PART_R(32) } else if (chunk == 64) { TEO_PART_R(64) } else if (chunk == 256) { TEO_PART_R(256) } else { TEO_PART_R(128) }
    } else if constexpr (64 / 4 >= 64 / LPR) {
        if (chunk == 64) { TEO_PART_R(64) } else if (chunk == 256) { TEO_PART_R(256) } else { TEO_PART_R(128) }
    } else {
        if (chunk == 256) { TEO_PART_R(256) } else { TEO_PART_R(128) }
    }
#undef TEO_PART_R
#undef TEO_PART
    if (!counters) {
        prof_bump(1);
        TEO_KLAUNCH((attn_decode_combine_kernel<T>), dim3(heads, bt.batch), 512, 0, st, part, (T*)o, d_pos, hd, nsplit, chunk, bt.o_stride);
        prof_bump(-1);
    }
}

// rope_cos != NULL: q is the raw qkv row; RoPE and the KV append of the new token happen inside the kernel
// bt: batched step (bt.batch conversations: q/o rows, caches, positions and partial slabs strided per conversation)
int attn_decode(const void* q, void* kc, void* vc, void* vtc, const float* rope_cos, const float* rope_sin, void* o,
                float* part, const int* d_pos, int S_max, int heads, int kv_heads, int hd, float scale, int dtype,
                hipStream_t st, AttnBatch bt, int* counters) {
    const bool rope = rope_cos != nullptr;
    if (g_attn_fat && !g_fused_combine && bt.batch == 1 && attn_fat_ok(hd, dtype, S_max))
        return attn_decode_fat(q, kc, vc, vtc, rope_cos, rope_sin, o, part, d_pos, S_max, heads, kv_heads, scale, st, bt, true);
    if (!g_fused_combine) counters = nullptr;
    int chunk = g_dec_chunk ? g_dec_chunk : (bt.batch > 1 ? 128 : 64);
    const int esz = dtype == TEO_F32 ? 4 : 2;
    const int lpr = hd * esz / 16;
    if (chunk / 4 < 64 / lpr) chunk = 4 * (64 / lpr);          // every wave needs at least one load instruction of keys
    if (chunk != 32 && chunk != 64 && chunk != 128 && chunk != 256) chunk = 128;
    while (chunk < 256 && cdiv(S_max, chunk) > 256) chunk *= 2;        // the combine handles at most 256 splits
    const int nsplit = cdiv(S_max, chunk);
    if (nsplit > 256 || (hd * esz) % 16 != 0 || (lpr != 2 && lpr != 4 && lpr != 8 && lpr != 16 && lpr != 32)) {
        set_error("attn_decode: unsupported head_dim %d / max_seq %d", hd, S_max);
        return TEO_ERR_UNSUPPORTED;
    }
#define TEO_DEC(TT, LL) attn_decode_launch<TT, LL>(q, kc, vc, vtc, rope_cos, rope_sin, o, part, d_pos, S_max, heads, kv_heads, hd, scale, nsplit, chunk, rope, bt, counters, st)
    if (dtype == TEO_F32) {
        switch (lpr) { case 2: TEO_DEC(float, 2); break; case 4: TEO_DEC(float, 4); break; case 8: TEO_DEC(float, 8); break;
                       case 16: TEO_DEC(float, 16); break; default: TEO_DEC(float, 32); }
    } else {
        switch (lpr) { case 2: TEO_DEC(bf16_t, 2); break; case 4: TEO_DEC(bf16_t, 4); break; case 8: TEO_DEC(bf16_t, 8); break;
                       case 16: TEO_DEC(bf16_t, 16); break; default: TEO_DEC(bf16_t, 32); }
    }
#undef TEO_DEC
    TEO_LAUNCH_CHECK("attn_decode");
    return TEO_OK;
}

}  // namespace teo
