// Attention kernels.
//
//  attn_simple_kernel<T>   : one wave per (query, head); scores in LDS, exact two-pass softmax.  fp32 parity path,
//                            fallback for head dims the MFMA kernel does not take, and the on-GPU cross-check of it.
//  (the MFMA flash kernel, attn_flash32_kernel, lives in flash.hip)
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

// `output_attentions` of the kept forward signature (llava_llama.py:65,95 -> LlamaAttention's eager softmax): the [heads][q_len][kv_len]
// maps themselves, which the flash kernel never materialises.  NOT on the performance path -- a caller that asks for the maps gets them
// from this plain kernel beside the (unchanged) attention: one wave per (query, head), scores = scale * q . k in fp32 from the operands the
// attention kernel reads (rotated q, cached k), softmax statistics in fp32, ONE rounding to the output type, exact zeros for masked keys.
template <typename T>
__global__ __launch_bounds__(64) void attn_probs_kernel(teo_attn_args a, T* __restrict__ probs) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* qs = sm;                   // [head_dim]
    float* sc = sm + a.head_dim;      // [kv_len]
    const int i = blockIdx.x, h = blockIdx.y, b = blockIdx.z, lane = threadIdx.x;
    const int hk = h / (a.heads / a.kv_heads);
    const int hd = a.head_dim;
    const T* q = (const T*)a.q + b * a.q_bs + h * a.q_hs + (long long)i * a.q_rs;
    const T* k = (const T*)a.k + b * a.k_bs + hk * a.k_hs;
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
        sc[j] = p;
    }
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
    T* out = probs + (((long long)b * a.heads + h) * a.q_len + i) * a.kv_len;
    for (int j = lane; j < a.kv_len; j += 64) Elem<T>::st(out + j, j < lim ? sc[j] * inv : 0.f);
}

// probs: [batch][heads][q_len][kv_len] in the model dtype
int attention_probs(const teo_attn_args* ap, int dtype, void* probs, hipStream_t st) {
    const teo_attn_args& a = *ap;
    if (a.q_len == 0 || a.batch == 0) return TEO_OK;
    TEO_CHECK_ARG(a.heads % a.kv_heads == 0 && (a.kv_len >= a.q_len || !a.causal), "attention_probs: heads %d kv_heads %d q_len %d kv_len %d", a.heads, a.kv_heads, a.q_len, a.kv_len);
    const size_t lds = (size_t)(a.head_dim + a.kv_len) * sizeof(float);
    if (lds > 64 * 1024) {
        set_error("attention_probs: kv_len + head_dim <= 16384 (got %d)", a.kv_len + a.head_dim);
        return TEO_ERR_UNSUPPORTED;
    }
    dim3 grid(a.q_len, a.heads, a.batch);
    if (dtype == TEO_F32) attn_probs_kernel<float><<<grid, 64, lds, st>>>(a, (float*)probs);
    else if (dtype == TEO_F16) attn_probs_kernel<f16_t><<<grid, 64, lds, st>>>(a, (f16_t*)probs);
    else attn_probs_kernel<bf16_t><<<grid, 64, lds, st>>>(a, (bf16_t*)probs);
    TEO_LAUNCH_CHECK("attn_probs");
    return TEO_OK;
}

bool attn_mfma_ok(const teo_attn_args& a, int dtype) {
    if ((dtype != TEO_BF16 && dtype != TEO_F16) || (a.flags & TEO_ATTN_FORCE_SIMPLE) || a.vt == nullptr) return false;
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
    if (attn_mfma_ok(a, dtype)) return attention_flash32(a, st, dtype == TEO_F16);
    TEO_CHECK_ARG(a.v != nullptr, "teo_attention: generic kernel needs row-major V");
    const size_t lds = (size_t)(a.head_dim + a.kv_len) * sizeof(float);
    if (lds > 64 * 1024) {
        set_error("teo_attention: generic kernel supports kv_len + head_dim <= 16384 (got %d)", a.kv_len + a.head_dim);
        return TEO_ERR_UNSUPPORTED;
    }
    dim3 grid(a.q_len, a.heads, a.batch);
    if (dtype == TEO_F32) attn_simple_kernel<float><<<grid, 64, lds, st>>>(a);
    else if (dtype == TEO_F16) attn_simple_kernel<f16_t><<<grid, 64, lds, st>>>(a);
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
// tune().attn_chunk (default 0): keys per workgroup (32 / 64 / 128 / 256); 0 = auto: 64 for one conversation (2.83 vs 2.86 ms/token
                                  // at 128, 2.87 at 32, 2.94 at 256), 128 for a batched step (4.74 vs 4.79 ms/step at 64)
// tune().attn_whole (default 1): batched steps: whole-context kernel (one workgroup per (conversation, head), no combine launch): 0 off, 1 auto
                                  // (batch * heads >= half the CUs), 2 whenever the shape allows.  Bit-identical to the split + combine pair.

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
template <> struct Cvt16<f16_t> {
    static constexpr int N = 8;
    __device__ static __forceinline__ void cvt(const uint4& r, float* f) {
        f[0] = h_lo<true>(r.x); f[1] = h_hi<true>(r.x); f[2] = h_lo<true>(r.y); f[3] = h_hi<true>(r.y);
        f[4] = h_lo<true>(r.z); f[5] = h_hi<true>(r.z); f[6] = h_lo<true>(r.w); f[7] = h_hi<true>(r.w);
    }
    __device__ static __forceinline__ uint4 pack(const float* f) {
        return make_uint4(pack_f16x2(f[0], f[1]), pack_f16x2(f[2], f[3]), pack_f16x2(f[4], f[5]), pack_f16x2(f[6], f[7]));
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

// two packed 16-bit products + fp32 accumulate: v_dot2_f32_bf16 / v_dot2_f32_f16 by the element type
#define attn_dot2 dot2h<IsF16<T>::v>

// acc[e] += p * v[e] over the 8 (16-bit) / 4 (fp32) values of a 16-byte V chunk.  IEEE half (round 6): v_fma_mix_f32 takes the half operand
// straight from either word of the register (op_sel picks the word, op_sel_hi marks it f16) -- 8 instructions instead of 8 converts (four of
// them SDWA forms with their s_nop hazards) + 8 FMAs; fma(float(v), p, acc) with one rounding: the same bits.  The fp16 decode attention
// kernel measured 8.38 us against bf16's 8.04 inside real steps before it (profiles/r06_bench_fp16_kernel_stats.md).
template <typename T>
__device__ __forceinline__ void attn_pv_fma(const uint4& vr, float p, float (&acc)[Cvt16<T>::N]) {
    if constexpr (IsF16<T>::v) {
        const unsigned w[4] = {vr.x, vr.y, vr.z, vr.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(acc[2 * i]) : "v"(w[i]), "v"(p));
            asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(acc[2 * i + 1]) : "v"(w[i]), "v"(p));
        }
    } else {
        float vf[Cvt16<T>::N];
        Cvt16<T>::cvt(vr, vf);
#pragma unroll
        for (int e = 0; e < Cvt16<T>::N; ++e) acc[e] = fmaf(p, vf[e], acc[e]);
    }
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
__global__ __launch_bounds__(256) void attn_decode_partial_kernel(const int* __restrict__ d_pos, T* __restrict__ kc, T* __restrict__ vc,
                                                                  const T* __restrict__ q, int S_max, int heads, int kv_heads, int nsplit,
                                                                  T* __restrict__ vtc, const float* __restrict__ cs,
                                                                  const float* __restrict__ sn, float* __restrict__ part, float scale,
                                                                  AttnBatch bt) {
    // argument order: what the first K / V request needs (the position, the caches, the geometry) sits in the 12 dwords that arrive
    // preloaded in SGPRs (Makefile: -amdgpu-kernarg-preload-count) -- the position load is the first instruction, not the second round trip
    constexpr int VE = Cvt16<T>::N;
    constexpr int HD = LPR * VE;
    const int kv_len = d_pos[blockIdx.z] + 1;           // requested first: the K / V requests wait for it, the rest of the arguments load beside it
    {   // conversation blockIdx.z of a batched step: its own query row, caches, position and partial slab
        const long long bz = blockIdx.z;
        q += bz * bt.q_stride;
        kc += bz * bt.cache_stride;
        vc += bz * bt.cache_stride;
        if (vtc) vtc += bz * bt.cache_stride;
        part += bz * (long long)heads * nsplit * (HD + 2);
    }
    constexpr int RPI = 64 / LPR;                       // rows (keys) per wave-wide load instruction
    constexpr int KPW = DEC_CHUNK / 4;                  // keys per wave
    constexpr int NI = KPW / RPI;                       // load instructions per wave per operand
    __shared__ float sc[DEC_CHUNK];
    __shared__ float pstrip[4][DEC_CHUNK];
    __shared__ float red[8];
    __shared__ float obuf[4][HD];
    const int h = blockIdx.x, sp = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int hk = (heads == kv_heads) ? h : h / (heads / kv_heads);      // MHA: no integer division in front of the first request
    const int c0 = sp * DEC_CHUNK;
    float* out = part + ((long long)h * nsplit + sp) * (HD + 2);
    if (c0 >= kv_len) {                                 // nothing here: neutral partial
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
        s = group_sum<LPR>(s);
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
    // every wave holds the chunk's rounded probabilities (same values in all four): each writes its own strip and reads it back --
    // ordered inside the wave, so no workgroup barrier between the softmax and PV
#pragma unroll
    for (int i = 0; i < SPL; ++i)
        if (DEC_CHUNK >= 64 || lane + 64 * i < DEC_CHUNK) pstrip[wid][lane + 64 * i] = Elem<T>::round(sv[i]);
    // ---- PV on this wave's keys
    float acc[VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) acc[e] = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const float p = pstrip[wid][wid * KPW + i * RPI + grp];      // 0 for keys >= kv_len
        if constexpr (ROPE) {
            float vf[VE];
            Cvt16<T>::cvt(vr[i], vf);
            if (kw0 + i * RPI + grp == pos) {
#pragma unroll
                for (int e = 0; e < VE; ++e) vf[e] = vnew[e];
            }
#pragma unroll
            for (int e = 0; e < VE; ++e) acc[e] = fmaf(p, vf[e], acc[e]);
        } else {
            attn_pv_fma<T>(vr[i], p, acc);
        }
    }
#pragma unroll
    for (int e = 0; e < VE; ++e) acc[e] = cross_group_sum<LPR>(acc[e]);
    if (grp == 0) {
#pragma unroll
        for (int e = 0; e < VE; ++e) obuf[wid][sub * VE + e] = acc[e];
    }
    __syncthreads();
    for (int d = tid; d < HD; d += 256) out[2 + d] = obuf[0][d] + obuf[1][d] + obuf[2][d] + obuf[3][d];
    if (tid == 0) { out[0] = mx; out[1] = sum; }
}

// Merge of one head's split records (m, l, o[hd]) by 512 threads: shared by the combine launch (records in global memory)
// and the whole-context kernel (records in LDS) so both evaluate the same expressions in the same order.
// Split weights: one thread per split (parallel loads).  Output: thread = (column d, split group g); a thread owns every G-th
// split (G = 512 / hd) and keeps its loads in flight, the G partial sums of a column meet in LDS -- one or two round trips
// instead of a dependent chain over all splits.  pb: records of this head, `stride` floats apart; nact <= 256 splits hold keys.
template <typename T>
__device__ __forceinline__ void attn_merge_records(const float* __restrict__ pb, int stride, int nact, int hd_log2, T* __restrict__ o_row,
                                                   float* w /* [256] */, float* red /* [16] */, float* accs /* [512] */) {
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const bool on = tid < 512;                                              // a larger workgroup: the other waves only join the barriers
    const int hd = 1 << hd_log2;                                            // hd is a power of two <= 256: shifts, no integer division
    const int G = 512 >> hd_log2;
    const int g = on ? tid >> hd_log2 : 0, d = tid & (hd - 1);
    // the first NB splits of this thread are requested together with the split statistics (they do not depend on them): ONE round
    // trip covers NB * G splits (48 at head_dim 128 = ctx 3072 at 64 keys per split).  32-bit element offsets: a head's records are
    // at most 256 x 258 floats
    constexpr int NB = 12;
    float v0[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) v0[i] = pb[(unsigned)(min(g + i * G, nact - 1) * stride + 2 + d)];
    float m0 = -INFINITY, l0 = 0.f;
    if (tid < nact) { m0 = pb[(unsigned)(tid * stride)]; l0 = pb[(unsigned)(tid * stride + 1)]; }
    float M = wave_max(m0);
    if (lane == 0 && on) red[wid] = M;
    __syncthreads();
    M = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));               // splits live in threads 0..255 = waves 0..3
    const float w0 = (m0 == -INFINITY) ? 0.f : expf(m0 - M);
    if (tid < 256) w[tid] = w0;
    float L = wave_sum(l0 * w0);
    if (lane == 0 && on) red[8 + wid] = L;
    __syncthreads();
    const float inv = 1.0f / ((red[8] + red[9]) + (red[10] + red[11]));
    float a = 0.f;
    // (the product is formed for every slot -- a clamped slot re-reads a valid record -- and a select decides what is added: no branch
    // per split; adding +0 to the running sum changes nothing)
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const float t = v0[i] * w[min(g + i * G, 255)];
        a += (g + i * G < nact) ? t : 0.f;
    }
    for (int s0 = g + NB * G; s0 < nact; s0 += 8 * G) {
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = pb[(unsigned)(min(s0 + i * G, nact - 1) * stride + 2 + d)];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float t = v[i] * w[min(s0 + i * G, 255)];
            a += (s0 + i * G < nact) ? t : 0.f;
        }
    }
    if (on) accs[tid] = a;
    __syncthreads();
    if (on && g == 0) {
        float t = 0.f;
        for (int k = 0; k < G; ++k) t += accs[d + k * hd];
        Elem<T>::st(o_row + d, t * inv);
    }
}

// one workgroup (512 threads) per head.  chunk and hd are powers of two (host-checked): passed as shifts
template <typename T>
__global__ __launch_bounds__(512) void attn_decode_combine_kernel(const int* __restrict__ d_pos, const float* __restrict__ part,
                                                                  T* __restrict__ o, int hd_log2, int nsplit, int chunk_log2,
                                                                  long long o_stride) {
    __shared__ float w[256];
    __shared__ float red[16];
    __shared__ float accs[512];
    const int pos = d_pos[blockIdx.y];                                      // requested first
    const int h = blockIdx.x;
    const int hd = 1 << hd_log2;
    const int stride = hd + 2;
    part += (long long)blockIdx.y * gridDim.x * nsplit * stride;
    o += (long long)blockIdx.y * o_stride;
    const float* pb = part + (long long)h * nsplit * stride;
    const int nact = min(nsplit, (pos + (1 << chunk_log2)) >> chunk_log2);  // splits that hold keys (<= 256)
    attn_merge_records<T>(pb, stride, nact, hd_log2, o + h * hd, w, red, accs);
}

// ------------------------------------------------------------------------------------------------
// Whole-context form for BATCHED steps: one workgroup (8 waves) per (conversation, head) walks the conversation's whole
// context -- with batch x heads >= the CU count every CU streams ONE long K/V range instead of ~20 short launches' worth of
// 64 KB workgroups, and the split records never leave the CU: no partial-record traffic, no combine launch (4.9 us per layer
// of a batched step, pure latency).
//   * a WAVE owns a chunk of DEC_CHUNK keys at a time (chunks wid, wid + 8, ...), processed as the four quarter-chunks the
//     four waves of the split kernel take: same score expression, same chunk max / exp / sum, same four partial PV sums added in
//     the same order -> the chunk record (m, l, o[hd]) is bit-identical to attn_decode_partial_kernel's;
//   * no workgroup barrier in the stream: scores go through a per-wave LDS strip; the loads run as a 4-slot register ring
//     (K0 K1 K2 K3 V0 V1 V2 V3 K0' ...: a slot is refilled with the next quarter as soon as it is consumed) so every wave keeps
//     3-4 quarters (24-32 KB) in flight across chunk boundaries;
//   * the records stay in LDS and are merged by the same code as the combine launch (attn_merge_records) -> the output is
//     bit-identical to the two-launch path with the same chunk size (tested).
// ------------------------------------------------------------------------------------------------
template <typename T, int LPR, int DEC_CHUNK, bool ROPE, int AW_WAVES, bool PROBE = false>   // PROBE (tools/attn_probe.hip only): loads without the arithmetic
__global__ __launch_bounds__(AW_WAVES * 64) void attn_decode_whole_kernel(const T* __restrict__ q, T* __restrict__ kc, T* __restrict__ vc,
                                                                          T* __restrict__ vtc, const float* __restrict__ cs,
                                                                          const float* __restrict__ sn, T* __restrict__ o,
                                                                          const int* __restrict__ d_pos, int S_max, int heads, int kv_heads,
                                                                          float scale, AttnBatch bt) {
    constexpr int VE = Cvt16<T>::N;
    constexpr int HD = LPR * VE;
    constexpr int RPI = 64 / LPR;                       // keys per wave-wide load instruction
    constexpr int KPQ = DEC_CHUNK / 4;                  // keys per quarter-chunk (= keys per wave of the split kernel)
    constexpr int NI = KPQ / RPI;                       // load instructions per quarter
    constexpr int SPL = (DEC_CHUNK + 63) / 64;
    constexpr int STRIDE = HD + 2;
    extern __shared__ __attribute__((aligned(16))) float aw_lds[];
    float* sc = aw_lds + (threadIdx.x >> 6) * DEC_CHUNK;               // this wave's score strip
    float* rec = aw_lds + AW_WAVES * DEC_CHUNK;                         // [nact][HD + 2]
    {
        const long long bz = blockIdx.y;
        q += bz * bt.q_stride;
        kc += bz * bt.cache_stride;
        vc += bz * bt.cache_stride;
        if (vtc) vtc += bz * bt.cache_stride;
        d_pos += bz;
        o += bz * bt.o_stride;
    }
    const int h = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int hk = (heads == kv_heads) ? h : h / (heads / kv_heads);      // MHA: no integer division in front of the first request
    const int kv_len = *d_pos + 1;
    const int nact = (kv_len + DEC_CHUNK - 1) / DEC_CHUNK;
    const int sub = lane % LPR, grp = lane / LPR;
    // uniform head bases + 32-bit byte offsets (one head's cache is far below 4 GB): SGPR-base addressing, no 64-bit address per load
    const char* kb = reinterpret_cast<const char*>(kc + (long long)hk * S_max * HD);
    const char* vb = reinterpret_cast<const char*>(vc + (long long)hk * S_max * HD);
    const unsigned lane_off = (unsigned)(sub * VE * sizeof(T));

    // quarter `qt` of chunk `c` into a ring slot.  Unconditional: a wave without a next chunk re-reads row 0 of the head (cached)
    uint4 r0[NI], r1[NI], r2[NI], r3[NI];
#define TEO_AW_ISSUE(SLOT, BASE, C, QT, LIVE)                                                                          \
    _Pragma("unroll") for (int i = 0; i < NI; ++i) {                                                                   \
        const int j = (LIVE) ? min((C) * DEC_CHUNK + (QT) * KPQ + i * RPI + grp, kv_len - 1) : 0;                      \
        SLOT[i] = ld_kv((BASE) + ((unsigned)j * (unsigned)(HD * sizeof(T)) + lane_off));                               \
    }
    int c = wid;
    {
        const bool live = c < nact;
        TEO_AW_ISSUE(r0, kb, c, 0, live)
        TEO_AW_ISSUE(r1, kb, c, 1, live)
        TEO_AW_ISSUE(r2, kb, c, 2, live)
        TEO_AW_ISSUE(r3, kb, c, 3, live)
    }
    float qf[VE];
    uint4 knew_pk = make_uint4(0, 0, 0, 0), vnew_pk = make_uint4(0, 0, 0, 0);     // the new token's rotated key / value, in the storage type
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
        Cvt16<T>::cvt(*reinterpret_cast<const uint4*>(q + h * HD + sub * VE), own);
        Cvt16<T>::cvt(*reinterpret_cast<const uint4*>(q + h * HD + psub * VE), oth);
#pragma unroll
        for (int e = 0; e < VE; ++e) qf[e] = Elem<T>::round(own[e] * cf[e] + sgn * oth[e] * sf[e]);
        const T* kraw = q + (long long)(heads + hk) * HD;
        Cvt16<T>::cvt(*reinterpret_cast<const uint4*>(kraw + sub * VE), own);
        Cvt16<T>::cvt(*reinterpret_cast<const uint4*>(kraw + psub * VE), oth);
        float knew[VE];
#pragma unroll
        for (int e = 0; e < VE; ++e) knew[e] = Elem<T>::round(own[e] * cf[e] + sgn * oth[e] * sf[e]);
        knew_pk = Cvt16<T>::pack(knew);                    // exact: the values are already rounded to T
        vnew_pk = *reinterpret_cast<const uint4*>(q + (long long)(heads + kv_heads + hk) * HD + sub * VE);
        // KV append of the new token, once per kv head (the stream below never reads row `pos`: it substitutes these registers)
        if (wid == 0 && grp == 0 && h % (heads / kv_heads) == 0) {
            *reinterpret_cast<uint4*>(kc + ((long long)hk * S_max + pos) * HD + sub * VE) = knew_pk;
            *reinterpret_cast<uint4*>(vc + ((long long)hk * S_max + pos) * HD + sub * VE) = vnew_pk;
            if (vtc) {
                const uint4 pv = vnew_pk;
                const T* pe = reinterpret_cast<const T*>(&pv);
#pragma unroll
                for (int e = 0; e < VE; ++e) vtc[((long long)hk * HD + sub * VE + e) * S_max + pos] = pe[e];
            }
        }
    } else {
        const uint4 qraw = *reinterpret_cast<const uint4*>(q + h * HD + sub * VE);
        Cvt16<T>::cvt(qraw, qf);
    }
    constexpr bool DOT2 = sizeof(T) == 2;
    uint4 qpk = make_uint4(0, 0, 0, 0);
    if (DOT2) qpk = Cvt16<T>::pack(qf);

    // scores of quarter QT (keys c0 + QT*KPQ ..) from ring slot SLOT into the wave's strip
#define TEO_AW_SCORES(SLOT, QT)                                                                                        \
    _Pragma("unroll") for (int i = 0; i < NI; ++i) {                                                                   \
        const int j = c0 + (QT) * KPQ + i * RPI + grp;                                                                 \
        const bool isnew = ROPE && j == pos;               /* the new token's key: registers, not the cache row */     \
        uint4 kraw = SLOT[i];                                                                                          \
        kraw.x = isnew ? knew_pk.x : kraw.x; kraw.y = isnew ? knew_pk.y : kraw.y;                                      \
        kraw.z = isnew ? knew_pk.z : kraw.z; kraw.w = isnew ? knew_pk.w : kraw.w;                                      \
        float s = 0.f;                                                                                                 \
        if (DOT2) {                                                                                                    \
            s = attn_dot2(kraw.x, qpk.x, s); s = attn_dot2(kraw.y, qpk.y, s);                                          \
            s = attn_dot2(kraw.z, qpk.z, s); s = attn_dot2(kraw.w, qpk.w, s);                                          \
        } else {                                                                                                       \
            float kf[VE];                                                                                              \
            Cvt16<T>::cvt(kraw, kf);                                                                                   \
            _Pragma("unroll") for (int e = 0; e < VE; ++e) s = fmaf(qf[e], kf[e], s);                                  \
        }                                                                                                              \
        s = group_sum<LPR>(s);                                                                                         \
        if (sub == 0) sc[j - c0] = (j < kv_len) ? s * scale : -INFINITY;                                               \
    }
    // PV of quarter QT from ring slot SLOT: the split kernel's per-wave partial sum, reduced over the key groups
#define TEO_AW_PV(SLOT, QT, OUT)                                                                                       \
    {                                                                                                                  \
        float acc[VE];                                                                                                 \
        _Pragma("unroll") for (int e = 0; e < VE; ++e) acc[e] = 0.f;                                                   \
        _Pragma("unroll") for (int i = 0; i < NI; ++i) {                                                               \
            const float p = sc[(QT) * KPQ + i * RPI + grp];                                                            \
            const bool isnew = ROPE && c0 + (QT) * KPQ + i * RPI + grp == pos;                                         \
            uint4 vraw = SLOT[i];                                                                                      \
            vraw.x = isnew ? vnew_pk.x : vraw.x; vraw.y = isnew ? vnew_pk.y : vraw.y;                                  \
            vraw.z = isnew ? vnew_pk.z : vraw.z; vraw.w = isnew ? vnew_pk.w : vraw.w;                                  \
            attn_pv_fma<T>(vraw, p, acc);                                                                              \
        }                                                                                                              \
        _Pragma("unroll") for (int e = 0; e < VE; ++e) acc[e] = cross_group_sum<LPR>(acc[e]);                          \
        if ((QT) == 0) { _Pragma("unroll") for (int e = 0; e < VE; ++e) OUT[e] = acc[e]; }                             \
        else           { _Pragma("unroll") for (int e = 0; e < VE; ++e) OUT[e] = OUT[e] + acc[e]; }                    \
    }

    unsigned probe_acc = 0;
#define TEO_AW_TOUCH(SLOT) _Pragma("unroll") for (int i = 0; i < NI; ++i) probe_acc ^= SLOT[i].x ^ SLOT[i].y ^ SLOT[i].z ^ SLOT[i].w;
    if constexpr (PROBE) {
        for (; c < nact; c += AW_WAVES) {
            const int cn = c + AW_WAVES;
            const bool nlive = cn < nact;
            TEO_AW_TOUCH(r0) TEO_AW_ISSUE(r0, vb, c, 0, true)
            TEO_AW_TOUCH(r1) TEO_AW_ISSUE(r1, vb, c, 1, true)
            TEO_AW_TOUCH(r2) TEO_AW_ISSUE(r2, vb, c, 2, true)
            TEO_AW_TOUCH(r3) TEO_AW_ISSUE(r3, vb, c, 3, true)
            TEO_AW_TOUCH(r0) TEO_AW_ISSUE(r0, kb, cn, 0, nlive)
            TEO_AW_TOUCH(r1) TEO_AW_ISSUE(r1, kb, cn, 1, nlive)
            TEO_AW_TOUCH(r2) TEO_AW_ISSUE(r2, kb, cn, 2, nlive)
            TEO_AW_TOUCH(r3) TEO_AW_ISSUE(r3, kb, cn, 3, nlive)
        }
        if (probe_acc == 0x12345678u) o[h * HD + (tid & (HD - 1))] = T(0);
        return;
    }
#undef TEO_AW_TOUCH
    for (; c < nact; c += AW_WAVES) {
        const int c0 = c * DEC_CHUNK;
        const int cn = c + AW_WAVES;
        const bool nlive = cn < nact;
        // ---- scores, quarter by quarter; each consumed K slot is refilled with the same quarter of V
        TEO_AW_SCORES(r0, 0)
        TEO_AW_ISSUE(r0, vb, c, 0, true)
        TEO_AW_SCORES(r1, 1)
        TEO_AW_ISSUE(r1, vb, c, 1, true)
        TEO_AW_SCORES(r2, 2)
        TEO_AW_ISSUE(r2, vb, c, 2, true)
        TEO_AW_SCORES(r3, 3)
        TEO_AW_ISSUE(r3, vb, c, 3, true)
        __builtin_amdgcn_wave_barrier();
        // ---- chunk max / exp / sum: the split kernel's expressions over the same strip
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
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < SPL; ++i)
            if (DEC_CHUNK >= 64 || lane + 64 * i < DEC_CHUNK) sc[lane + 64 * i] = Elem<T>::round(sv[i]);
        __builtin_amdgcn_wave_barrier();
        // ---- PV, quarter by quarter; each consumed V slot is refilled with the next chunk's K quarter
        float ot[VE];                                      // ((q0 + q1) + q2) + q3, the order the split kernel adds its four waves
        TEO_AW_PV(r0, 0, ot)
        TEO_AW_ISSUE(r0, kb, cn, 0, nlive)
        TEO_AW_PV(r1, 1, ot)
        TEO_AW_ISSUE(r1, kb, cn, 1, nlive)
        TEO_AW_PV(r2, 2, ot)
        TEO_AW_ISSUE(r2, kb, cn, 2, nlive)
        TEO_AW_PV(r3, 3, ot)
        TEO_AW_ISSUE(r3, kb, cn, 3, nlive)
        __builtin_amdgcn_wave_barrier();                   // the strip is rewritten by the next chunk's scores
        float* rc = rec + (long long)c * STRIDE;
        if (grp == 0) {
#pragma unroll
            for (int e = 0; e < VE; ++e) rc[2 + sub * VE + e] = ot[e];
        }
        if (lane == 0) { rc[0] = mx; rc[1] = sum; }
    }
#undef TEO_AW_ISSUE
#undef TEO_AW_SCORES
#undef TEO_AW_PV
    __syncthreads();
    float* w = rec + (long long)nact * STRIDE;             // merge scratch behind the records: w[256], red[16], accs[512]
    attn_merge_records<T>(rec, STRIDE, nact, __builtin_ctz((unsigned)HD), o + h * HD, w, w + 256, w + 272);
}

size_t attn_decode_ws_bytes(int heads, int hd, int S_max, int batch) {
    const int nsplit = cdiv(S_max, 32);      // sized for the smallest chunk
    return (size_t)batch * heads * nsplit * (hd + 2) * sizeof(float);
}

template <typename T, int LPR>
static void attn_decode_launch(const void* q, void* kc, void* vc, void* vtc, const float* cs, const float* sn, void* o,
                               float* part, const int* d_pos, int S_max, int heads, int kv_heads, int hd, float scale,
                               int nsplit, int chunk, bool rope, AttnBatch bt, hipStream_t st) {
    dim3 grid(heads, nsplit, bt.batch);
#define TEO_PART(CH, RP)                                                                                              \
    TEO_KLAUNCH((attn_decode_partial_kernel<T, LPR, CH, RP>), grid, 256, 0, st, d_pos, (T*)kc, (T*)vc, (const T*)q, S_max, heads, kv_heads,      \
                                                                     nsplit, (T*)vtc, cs, sn, part, scale, bt)
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
    prof_bump(1);
    TEO_KLAUNCH((attn_decode_combine_kernel<T>), dim3(heads, bt.batch), 512, 0, st, d_pos, part, (T*)o, __builtin_ctz((unsigned)hd), nsplit,
                __builtin_ctz((unsigned)chunk), bt.o_stride);
    prof_bump(-1);
}

template <typename T, int LPR, int NW>
static int attn_whole_launch_t(const void* q, void* kc, void* vc, void* vtc, const float* cs, const float* sn, void* o, const int* d_pos,
                               int S_max, int heads, int kv_heads, float scale, int chunk, bool rope, AttnBatch bt, size_t lds, hipStream_t st) {
    constexpr int RPI = 64 / LPR;
    dim3 grid(heads, bt.batch);
#define TEO_AW(CH, RP)                                                                                                          \
    {                                                                                                                           \
        static unsigned long long attr_mask = 0;                                                                                \
        if (lds > 48 * 1024) if (int e = lds_attr_once(reinterpret_cast<const void*>(&attn_decode_whole_kernel<T, LPR, CH, RP, NW>), 96 * 1024, &attr_mask, "attn_decode_whole")) return e; \
        TEO_KLAUNCH((attn_decode_whole_kernel<T, LPR, CH, RP, NW>), grid, NW * 64, lds, st, (const T*)q, (T*)kc, (T*)vc, (T*)vtc, cs, sn, \
                    (T*)o, d_pos, S_max, heads, kv_heads, scale, bt);                                                           \
    }
#define TEO_AW_R(CH) { if (rope) TEO_AW(CH, true) else TEO_AW(CH, false) }
    // quarter-chunks of 1..8 load instructions: chunk / 4 / RPI in [1, 8]
    if (chunk == 128) { if constexpr (128 / 4 / RPI >= 1 && 128 / 4 / RPI <= 8) TEO_AW_R(128) else return TEO_ERR_UNSUPPORTED; }
    else if (chunk == 64) { if constexpr (64 / 4 / RPI >= 1 && 64 / 4 / RPI <= 8) TEO_AW_R(64) else return TEO_ERR_UNSUPPORTED; }
    else if (chunk == 32) { if constexpr (32 / 4 / RPI >= 1 && 32 / 4 / RPI <= 8) TEO_AW_R(32) else return TEO_ERR_UNSUPPORTED; }
    else return TEO_ERR_UNSUPPORTED;
#undef TEO_AW_R
#undef TEO_AW
    note_kernel("attn_decode_whole");
    TEO_LAUNCH_CHECK("attn_decode_whole");
    return TEO_OK;
}

static int attn_whole_launch(const void* q, void* kc, void* vc, void* vtc, const float* cs, const float* sn, void* o, const int* d_pos, int S_max,
                             int heads, int kv_heads, int hd, float scale, int chunk, int lpr, int dtype, AttnBatch bt, size_t lds, hipStream_t st) {
    const bool rope = cs != nullptr;
#define TEO_AWL(TT, LL) return attn_whole_launch_t<TT, LL, 8>(q, kc, vc, vtc, cs, sn, o, d_pos, S_max, heads, kv_heads, scale, chunk, rope, bt, lds, st)
    if (dtype == TEO_F32) {
        switch (lpr) { case 4: TEO_AWL(float, 4); case 8: TEO_AWL(float, 8); case 16: TEO_AWL(float, 16); case 32: TEO_AWL(float, 32); default: return TEO_ERR_UNSUPPORTED; }
    }
    if (dtype == TEO_F16) {
        switch (lpr) { case 4: TEO_AWL(f16_t, 4); case 8: TEO_AWL(f16_t, 8); case 16: TEO_AWL(f16_t, 16); case 32: TEO_AWL(f16_t, 32); default: return TEO_ERR_UNSUPPORTED; }
    }
    switch (lpr) { case 4: TEO_AWL(bf16_t, 4); case 8: TEO_AWL(bf16_t, 8); case 16: TEO_AWL(bf16_t, 16); case 32: TEO_AWL(bf16_t, 32); default: return TEO_ERR_UNSUPPORTED; }
#undef TEO_AWL
}

// rope_cos != NULL: q is the raw qkv row; RoPE and the KV append of the new token happen inside the kernel
// bt: batched step (bt.batch conversations: q/o rows, caches, positions and partial slabs strided per conversation)
int attn_decode(const void* q, void* kc, void* vc, void* vtc, const float* rope_cos, const float* rope_sin, void* o,
                float* part, const int* d_pos, int S_max, int heads, int kv_heads, int hd, float scale, int dtype,
                hipStream_t st, AttnBatch bt) {
    const bool rope = rope_cos != nullptr;
    int chunk = tune().attn_chunk ? tune().attn_chunk : (bt.batch > 1 ? 128 : 64);
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
    // whole-context form (batched steps with at least one workgroup per CU: B >= 8 at 32 heads; with the reductions on DPP the split form
    // wins below that -- B = 4: 30.1 vs 31.8 us, profiles/r04_attn_probe_b4.txt).  Its default chunk is 64 keys (quarter-chunks of 4 load instructions: a 4-slot ring of 64
    // VGPRs; with 128-key chunks the bf16 / head_dim 128 kernel needs more than 256 registers), "attn_chunk" forces another.
    {
        int cw = tune().attn_chunk ? tune().attn_chunk : 64;
        if (cw / 4 < 64 / lpr) cw = 4 * (64 / lpr);
        while (cw < 128 && cdiv(S_max, cw) > 256) cw *= 2;
        const int nsw = cdiv(S_max, cw);
        const int ni = cw / 4 / (64 / lpr);
        const size_t lds = ((size_t)8 * cw + (size_t)nsw * (hd + 2) + 784) * sizeof(float);
        const int cus = device_cu_count();
        const bool fits = ni >= 1 && ni <= 8 && (hd & (hd - 1)) == 0 && hd <= 256 && lds <= 96 * 1024 && nsw <= 256 && (cw == 32 || cw == 64 || cw == 128);
        // one workgroup per (conversation, head): whole rounds of the CUs only -- with 9..13 conversations x 32 heads the CUs that get a second
        // workgroup finish 1.3x later than the rest and the split form wins (B = 9: 4.45 vs 4.85 ms per step, B = 10: 4.55 vs 4.99,
        // B = 13: 5.25 vs 5.34; from 7/4 rounds on the imbalance amortises: B = 14: 5.52 vs 5.56; profiles/r06_batch_sweep_attn_{whole,split}.md)
        const int ncu = cus > 0 ? cus : 256, wgs = bt.batch * heads;
        const bool balanced = wgs >= ncu && (wgs % ncu == 0 || 4 * wgs >= 7 * ncu);
        if (fits && (tune().attn_whole == 2 || (tune().attn_whole == 1 && bt.batch > 1 && balanced))) {
            const int rcw = attn_whole_launch(q, kc, vc, vtc, rope_cos, rope_sin, o, d_pos, S_max, heads, kv_heads, hd, scale, cw, lpr, dtype, bt, lds, st);
            if (rcw != TEO_ERR_UNSUPPORTED) return rcw;
        }
    }
#define TEO_DEC(TT, LL) attn_decode_launch<TT, LL>(q, kc, vc, vtc, rope_cos, rope_sin, o, part, d_pos, S_max, heads, kv_heads, hd, scale, nsplit, chunk, rope, bt, st)
    if (dtype == TEO_F32) {
        switch (lpr) { case 2: TEO_DEC(float, 2); break; case 4: TEO_DEC(float, 4); break; case 8: TEO_DEC(float, 8); break;
                       case 16: TEO_DEC(float, 16); break; default: TEO_DEC(float, 32); }
    } else if (dtype == TEO_F16) {
        switch (lpr) { case 2: TEO_DEC(f16_t, 2); break; case 4: TEO_DEC(f16_t, 4); break; case 8: TEO_DEC(f16_t, 8); break;
                       case 16: TEO_DEC(f16_t, 16); break; default: TEO_DEC(f16_t, 32); }
    } else {
        switch (lpr) { case 2: TEO_DEC(bf16_t, 2); break; case 4: TEO_DEC(bf16_t, 4); break; case 8: TEO_DEC(bf16_t, 8); break;
                       case 16: TEO_DEC(bf16_t, 16); break; default: TEO_DEC(bf16_t, 32); }
    }
#undef TEO_DEC
    TEO_LAUNCH_CHECK("attn_decode");
    return TEO_OK;
}

}  // namespace teo
