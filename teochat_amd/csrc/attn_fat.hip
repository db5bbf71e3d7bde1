// Decode attention, one query row per conversation, "fat split" form (round 3).
//
// Replaces the H15 row of SURVEY.md section 8a at q_len == 1 (tf llama eager attention over the KV cache, called through
// llava_llama.py:88-99) for bf16 / head_dim 128 -- the same arithmetic contract as attention.hip's split-KV kernels (independent
// key chunks: chunk max, P = exp(s - max) rounded to bf16 for the PV product, normaliser from the unrounded P, fp32 merge of the
// chunk records), with a launch shape chosen for the memory system instead of for the key count:
//
//   grid = heads x NS workgroups of 1024 threads with NS = ceil(S_max / 320) -> 256 workgroups for LLaMA-2-7B at S_max 2560: ONE
//   workgroup per CU, every CU streams the same number of keys (the key range of a split is a fraction of the CURRENT context,
//   computed on the device), 16 waves x (5 K + 5 V) 16-byte non-temporal loads per lane = up to 160 KB in flight per CU, all
//   issued before the first use.  The round-2 form used 64-key workgroups: 1152 small workgroups at ctx 2300, 36 records per
//   head for the combine launch (8.4 + 4.5 us per layer in the real step).  Here a head leaves NS <= 16 records, small enough
//   that the consumer of the attention output -- the o-projection GEMV -- merges them in its prologue (gemv_o_combine in gemv.hip):
//   the combine launch disappears from the decode step.
//
// ROPE = true: d_q is the raw [q | k | v] row of the QKV GEMV; the kernel rotates q, and the lane group that owns key `pos`
// rotates the new k, takes the new v and appends both to the caches (attention.hip's scheme).
#include "ops.h"

namespace teo {

typedef __bf16 fat_bf16x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((ext_vector_type(4))) unsigned int fat_u32x4;

__device__ __forceinline__ float fat_dot2(unsigned a, unsigned b, float acc) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(fat_bf16x2, a), __builtin_bit_cast(fat_bf16x2, b), acc, false);
}
__device__ __forceinline__ uint4 fat_ld_nt(const void* p) {
    const fat_u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const fat_u32x4*>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void fat_unpack(const uint4& r, float* f) {
    f[0] = __uint_as_float(r.x << 16); f[1] = __uint_as_float(r.x & 0xffff0000u);
    f[2] = __uint_as_float(r.y << 16); f[3] = __uint_as_float(r.y & 0xffff0000u);
    f[4] = __uint_as_float(r.z << 16); f[5] = __uint_as_float(r.z & 0xffff0000u);
    f[6] = __uint_as_float(r.w << 16); f[7] = __uint_as_float(r.w & 0xffff0000u);
}
__device__ __forceinline__ uint4 fat_pack(const float* f) {
    return make_uint4(pack_bf2(f[0], f[1]), pack_bf2(f[2], f[3]), pack_bf2(f[4], f[5]), pack_bf2(f[6], f[7]));
}

constexpr int FAT_HD = 128, FAT_LPR = 16, FAT_VE = 8, FAT_RPI = 4;     // 16 lanes x 8 bf16 per key row, 4 rows per wave instruction
// waves per workgroup x row-group loads per wave and operand: 16 x 5 (1024 threads) or 8 x 10 (512 threads, <= 128 VGPRs: the form
// of the overlapped decode step, which wants the next kernel of the chain resident beside it); 320 keys per workgroup either way
constexpr int FAT_KEYS_MAX = 320;

int attn_fat_nsplit(int S_max) { return cdiv(S_max, FAT_KEYS_MAX); }
bool attn_fat_ok(int hd, int dtype, int S_max) { return hd == FAT_HD && dtype == TEO_BF16 && attn_fat_nsplit(S_max) <= ATTN_FAT_MAX_SPLITS; }

// partial record of (head h, split sp): part[(h * NS + sp) * ATTN_FAT_REC + {0: chunk max, 1: sum of unrounded P, 4..131: sum P_bf16 * v}]
template <bool ROPE, int NW, int NI>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(4, 8))) void attn_decode_fat_kernel(const bf16_t* __restrict__ q, bf16_t* __restrict__ kc,
                                                               bf16_t* __restrict__ vc, bf16_t* __restrict__ vtc,
                                                               const float* __restrict__ cs, const float* __restrict__ sn,
                                                               float* __restrict__ part, const int* __restrict__ d_pos, int S_max,
                                                               int heads, int kv_heads, float scale, int nsplit, AttnBatch bt,
                                                               int pos_arg, Chain ch) {
    constexpr int HD = FAT_HD, LPR = FAT_LPR, VE = FAT_VE, RPI = FAT_RPI;
    constexpr int FAT_KPW_MAX = NI * RPI;
    static_assert(NW * NI * RPI == FAT_KEYS_MAX, "320 keys per workgroup");
    {
        const long long bz = blockIdx.z;
        q += bz * bt.q_stride;
        kc += bz * bt.cache_stride;
        vc += bz * bt.cache_stride;
        if (vtc) vtc += bz * bt.cache_stride;
        d_pos += bz;
        part += bz * (long long)heads * nsplit * ATTN_FAT_REC;
    }
    __shared__ float sc[FAT_KEYS_MAX];
    __shared__ float red[2 * NW];
    __shared__ float obuf[NW][HD];
    const int h = blockIdx.x, sp = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int hk = h / (heads / kv_heads);
    const bool coh = ch.on != 0;
    const int kv_len = (pos_arg >= 0 ? pos_arg : *d_pos) + 1;          // chained launch: the position is an argument (d_pos may still be in flight)
    // keys of this split: an equal share of the current context, a whole number of row groups per wave
    const int kpw = (int)((((long long)kv_len + nsplit - 1) / nsplit + NW - 1) / NW);   // keys per wave, <= FAT_KPW_MAX because nsplit >= S_max / 320
    const int per = kpw * NW;
    const int c0 = sp * per;
    float* out = part + ((long long)h * nsplit + sp) * ATTN_FAT_REC;
    if (c0 >= kv_len) {                                        // short context: this split holds no key -> neutral record
        chain_wait(ch);                                        // (the previous reader of the records must be done before any store)
        if (tid == 0) { act_st4(out, 0, __float_as_uint(-INFINITY), coh); act_st4(out, 4, 0u, coh); }
        if (tid < HD) act_st4(out, (4 + tid) * 4, 0u, coh);
        chain_signal(ch);
        return;
    }
    const int sub = lane % LPR, grp = lane / LPR;
    const bf16_t* kb = kc + (long long)hk * S_max * HD + sub * VE;
    const bf16_t* vb = vc + (long long)hk * S_max * HD + sub * VE;
    const int kw0 = c0 + wid * kpw;                            // first key of this wave
    uint4 kr[NI], vr[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {                             // every load of the workgroup in flight before the first use
        const int j = min(kw0 + min(i * RPI + grp, kpw - 1), kv_len - 1);      // dead slots re-read the wave's last row (masked below)
        kr[i] = fat_ld_nt(kb + (long long)j * HD);
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int j = min(kw0 + min(i * RPI + grp, kpw - 1), kv_len - 1);
        vr[i] = fat_ld_nt(vb + (long long)j * HD);
    }
    chain_wait(ch);                                            // K / V of the cached keys are in flight; q comes from the predecessor
    float qf[VE], knew[VE], vnew[VE];
    const int pos = kv_len - 1;
    if (ROPE) {
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
        fat_unpack(act_ld16(q, (unsigned)(h * HD + sub * VE) * 2u, coh), own);
        fat_unpack(act_ld16(q, (unsigned)(h * HD + psub * VE) * 2u, coh), oth);
#pragma unroll
        for (int e = 0; e < VE; ++e) qf[e] = Elem<bf16_t>::round(own[e] * cf[e] + sgn * oth[e] * sf[e]);
        const unsigned kraw_off = (unsigned)((heads + hk) * HD) * 2u;
        fat_unpack(act_ld16(q, kraw_off + (unsigned)(sub * VE) * 2u, coh), own);
        fat_unpack(act_ld16(q, kraw_off + (unsigned)(psub * VE) * 2u, coh), oth);
#pragma unroll
        for (int e = 0; e < VE; ++e) knew[e] = Elem<bf16_t>::round(own[e] * cf[e] + sgn * oth[e] * sf[e]);
        fat_unpack(act_ld16(q, (unsigned)((heads + kv_heads + hk) * HD + sub * VE) * 2u, coh), vnew);
    } else {
        fat_unpack(act_ld16(q, (unsigned)(h * HD + sub * VE) * 2u, coh), qf);
    }
    const uint4 qpk = fat_pack(qf);                           // q is bf16-exact: packed operand of v_dot2c_f32_bf16
    // ---- scores of this wave's keys
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const bool live = i * RPI + grp < kpw;                 // this slot holds one of the wave's keys
        const int j = kw0 + i * RPI + grp;
        uint4 kraw = kr[i];
        if (ROPE && live && j == pos) {                        // the new token's key: not in the cache yet
            kraw = fat_pack(knew);
            if (h % (heads / kv_heads) == 0) {                 // one q head per kv head appends
                // (write-through in a chained launch: later steps read these rows with ordinary loads from lines nobody cached)
                const unsigned row_off = (unsigned)((((long long)hk * S_max + pos) * HD + sub * VE) * 2);
                act_st16(kc, row_off, kraw, coh);
                const uint4 pv = fat_pack(vnew);
                act_st16(vc, row_off, pv, coh);
                if (vtc) {
                    const bf16_t* pe = reinterpret_cast<const bf16_t*>(&pv);
#pragma unroll
                    for (int e = 0; e < VE; ++e) act_st2(vtc, (unsigned)((((long long)hk * HD + sub * VE + e) * S_max + pos) * 2), pe[e], coh);
                }
            }
        }
        float s = 0.f;
        s = fat_dot2(kraw.x, qpk.x, s); s = fat_dot2(kraw.y, qpk.y, s);
        s = fat_dot2(kraw.z, qpk.z, s); s = fat_dot2(kraw.w, qpk.w, s);
#pragma unroll
        for (int o = LPR >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (sub == 0) sc[wid * FAT_KPW_MAX + i * RPI + grp] = (live && j < kv_len) ? s * scale : -INFINITY;
    }
    __syncthreads();
    // ---- split max / exp / sum: every wave redundantly over the 320 score slots (5 per lane)
    constexpr int SPL = FAT_KEYS_MAX / 64;
    float sv[SPL];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < SPL; ++i) { sv[i] = sc[lane + 64 * i]; mx = fmaxf(mx, sv[i]); }
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < SPL; ++i) { sv[i] = expf(sv[i] - mx); sum += sv[i]; }      // -inf -> 0
    sum = wave_sum(sum);
    __syncthreads();
    if (wid == 0) {
#pragma unroll
        for (int i = 0; i < SPL; ++i) sc[lane + 64 * i] = Elem<bf16_t>::round(sv[i]);
    }
    __syncthreads();
    // ---- PV on this wave's keys
    float acc[VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) acc[e] = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const float p = sc[wid * FAT_KPW_MAX + i * RPI + grp];         // 0 for dead slots
        float vf[VE];
        fat_unpack(vr[i], vf);
        if (ROPE && i * RPI + grp < kpw && kw0 + i * RPI + grp == pos) {
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
    if (tid < HD) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) t += obuf[w][tid];
        act_st4(out, (4 + tid) * 4, __float_as_uint(t), coh);
    }
    if (tid == 0) { act_st4(out, 0, __float_as_uint(mx), coh); act_st4(out, 4, __float_as_uint(sum), coh); }
    chain_signal(ch);
}

// merge of the NS records of a head (also what gemv_o_combine does in its prologue, in the same order):
//   M = max_s m_s;  w_s = exp(m_s - M);  o[d] = (sum_s w_s * o_s[d]) / (sum_s w_s * l_s), both sums in split order
__global__ __launch_bounds__(128) void attn_decode_fat_combine_kernel(const float* __restrict__ part, bf16_t* __restrict__ o, int nsplit,
                                                                      long long o_stride, Chain ch) {
    const int h = blockIdx.x, d = threadIdx.x;
    const bool coh = ch.on != 0;
    part += (long long)blockIdx.y * gridDim.x * nsplit * ATTN_FAT_REC;
    o += (long long)blockIdx.y * o_stride;
    const float* pb = part + (long long)h * nsplit * ATTN_FAT_REC;
    chain_wait(ch);
    float v;
    if (coh) {                                     // same chains as attn_fat_merge, records through coherent loads
        float M = -INFINITY;
        float m[ATTN_FAT_MAX_SPLITS], l[ATTN_FAT_MAX_SPLITS], ov[ATTN_FAT_MAX_SPLITS];
#pragma unroll
        for (int s2 = 0; s2 < ATTN_FAT_MAX_SPLITS; ++s2) {
            const unsigned off = (unsigned)(min(s2, nsplit - 1) * ATTN_FAT_REC) * 4u;
            m[s2] = __uint_as_float(act_ld4(pb, off, true));
            l[s2] = __uint_as_float(act_ld4(pb, off + 4u, true));
            ov[s2] = __uint_as_float(act_ld4(pb, off + (unsigned)(4 + d) * 4u, true));
        }
#pragma unroll
        for (int s2 = 0; s2 < ATTN_FAT_MAX_SPLITS; ++s2) M = fmaxf(M, s2 < nsplit ? m[s2] : -INFINITY);
        float Ls = 0.f, a = 0.f;
#pragma unroll
        for (int s2 = 0; s2 < ATTN_FAT_MAX_SPLITS; ++s2) {
            if (s2 < nsplit) {
                const float w = (m[s2] == -INFINITY) ? 0.f : expf(m[s2] - M);
                Ls = fmaf(w, l[s2], Ls);
                a = fmaf(w, ov[s2], a);
            }
        }
        v = a / Ls;
    } else {
        v = attn_fat_merge(pb, nsplit, d);
    }
    act_st2(o, (unsigned)(h * FAT_HD + d) * 2u, f2bf(v), coh);
    chain_signal(ch);
}

int attn_decode_fat(const void* q, void* kc, void* vc, void* vtc, const float* rope_cos, const float* rope_sin, void* o, float* part,
                    const int* d_pos, int S_max, int heads, int kv_heads, float scale, hipStream_t st, AttnBatch bt, bool with_combine,
                    int pos_arg, const Chain* ch_attn, const Chain* ch_combine) {
    const int ns = attn_fat_nsplit(S_max);
    dim3 grid(heads, ns, bt.batch);
    Chain off;
    memset(&off, 0, sizeof(off));
    const Chain c1 = ch_attn ? *ch_attn : off, c2 = ch_combine ? *ch_combine : off;
    const bool slim = ch_attn != nullptr;          // chained step: 8 waves x 10 row groups (512 threads, half of a CU's registers)
#define TEO_FAT(RP, NWW, NII) TEO_KLAUNCH((attn_decode_fat_kernel<RP, NWW, NII>), grid, NWW * 64, 0, st, (const bf16_t*)q, (bf16_t*)kc, (bf16_t*)vc, \
                                          (bf16_t*)vtc, rope_cos, rope_sin, part, d_pos, S_max, heads, kv_heads, scale, ns, bt, pos_arg, c1)
    if (rope_cos) { if (slim) TEO_FAT(true, 8, 10); else TEO_FAT(true, 16, 5); }
    else { if (slim) TEO_FAT(false, 8, 10); else TEO_FAT(false, 16, 5); }
#undef TEO_FAT
    if (with_combine) {
        prof_bump(1);
        TEO_KLAUNCH(attn_decode_fat_combine_kernel, dim3(heads, bt.batch), 128, 0, st, (const float*)part, (bf16_t*)o, ns, bt.o_stride, c2);
        prof_bump(-1);
    }
    TEO_LAUNCH_CHECK("attn_decode_fat");
    return TEO_OK;
}

}  // namespace teo
