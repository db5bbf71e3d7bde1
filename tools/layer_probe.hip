// Persistent decode-layer probe: is ONE launch that walks the five weight streams of a LLaMA layer (qkv, KV cache, o, gate/up,
// down) with grid barriers between them faster than five launches of the same code?  ("why the decode step
// stays at 0.66".)  Weights are streamed exactly as the production GEMVs do (2 rows x 4 KiB per wave per step, non-temporal
// 16-byte loads, fp32 LDS image of x, wave reduction); every stage writes its outputs write-through (sc1) and the next stage
// reads them with sc1 loads after a counter barrier -- no fences.  Each wave issues the first weight block of the NEXT stage
// before it waits at the barrier, so the HBM stream keeps running across the seam.  Every spin is bounded.
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/layer_probe.hip -o tools/liblayer_probe.so
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned v4u __attribute__((ext_vector_type(4)));
constexpr int NSTAGE_MAX = 8;
constexpr int U = 4, R = 2, STEP = 64 * U;         // one step: 2 rows x 256 chunks of 16 B

struct Stage {
    const unsigned short* W;   // [layers][N][K] bf16
    int N, K, split;           // K = elements per row; `split` waves-groups share a row (partials summed by the consumer)
    int norm;                  // consumer-side sum of squares + scale (RMSNorm stand-in)
    long long layer_stride;    // elements between layers
};
struct Params {
    Stage st[NSTAGE_MAX];
    int nstage, layers;
    float* act[2];             // ping-pong activation buffers: [split][N] fp32 partials
    unsigned* counter;         // monotonic arrival counter (zeroed by the host)
    int* err;
    int prefetch;              // issue the next stage's first block before the barrier
    int sleep;                 // s_sleep argument while polling
};

__device__ __forceinline__ int xs_off(int k) {     // lane-linear fp32 image for 8-element chunks (see gemv.hip)
    const int c = k >> 3, g = (k & 7) >> 2;
    return ((((c >> 6) * 2 + g) << 8) + ((c & 63) << 2)) + (k & 3);
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float dot16(const v4u& w, const float* xv, float acc) {
    acc = fmaf(__uint_as_float(w.x << 16), xv[0], acc); acc = fmaf(__uint_as_float(w.x & 0xffff0000u), xv[1], acc);
    acc = fmaf(__uint_as_float(w.y << 16), xv[2], acc); acc = fmaf(__uint_as_float(w.y & 0xffff0000u), xv[3], acc);
    acc = fmaf(__uint_as_float(w.z << 16), xv[4], acc); acc = fmaf(__uint_as_float(w.z & 0xffff0000u), xv[5], acc);
    acc = fmaf(__uint_as_float(w.w << 16), xv[6], acc); acc = fmaf(__uint_as_float(w.w & 0xffff0000u), xv[7], acc);
    return acc;
}

template <bool FULL>
__device__ __forceinline__ void issue_block(v4u (&w)[U][R], const unsigned short* const (&rowp)[R], int c0, int lane, int nchunk) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int c = c0 + u * 64 + lane;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (FULL) w[u][r] = __builtin_nontemporal_load(reinterpret_cast<const v4u*>(rowp[r] + (long long)c * 8));
            else w[u][r] = (c < nchunk) ? __builtin_nontemporal_load(reinterpret_cast<const v4u*>(rowp[r] + (long long)c * 8)) : (v4u){0, 0, 0, 0};
        }
    }
}
template <bool FULL>
__device__ __forceinline__ void consume_block(const v4u (&w)[U][R], const float* xs, int xbase, int c0, int lane, int nchunk, float (&acc)[R]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int c = c0 + u * 64 + lane;
        float xv[8];
        if (FULL || c < nchunk) {
            const float4 a = *reinterpret_cast<const float4*>(xs + xs_off(xbase + c * 8));
            const float4 b = *reinterpret_cast<const float4*>(xs + xs_off(xbase + c * 8 + 4));
            xv[0] = a.x; xv[1] = a.y; xv[2] = a.z; xv[3] = a.w; xv[4] = b.x; xv[5] = b.y; xv[6] = b.z; xv[7] = b.w;
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) xv[e] = 0.f;
        }
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = dot16(w[u][r], xv, acc[r]);
    }
}

// group g of a stage -> (piece, first row); rows 2q, 2q+1 share the x slice `piece`
__device__ __forceinline__ void group_rows(const Stage& s, const unsigned short* W, int g, int ngroups, const unsigned short* (&rowp)[R],
                                           int& piece, int& row0) {
    const int gg = min(g, ngroups - 1);
    piece = gg % s.split;
    row0 = (gg / s.split) * R;
    const int kp = s.K / s.split;
#pragma unroll
    for (int r = 0; r < R; ++r) rowp[r] = W + (long long)min(row0 + r, s.N - 1) * s.K + (long long)piece * kp;
}

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void layer_probe_kernel(Params P, int only_layer, int only_stage) {
    extern __shared__ __attribute__((aligned(16))) float xs[];
    __shared__ float red[WAVES];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int nwaves = gridDim.x * WAVES;
    const int wave = blockIdx.x * WAVES + wid;
    const bool persistent = only_stage < 0;
    const int l0 = persistent ? 0 : only_layer, l1 = persistent ? P.layers : only_layer + 1;
    const int s0 = persistent ? 0 : only_stage, s1 = persistent ? P.nstage : only_stage + 1;

    v4u wa[U][R];
    bool have = false;
    unsigned epoch = 0;
    bool dead = false;
    for (int l = l0; l < l1; ++l) {
        for (int si = s0; si < s1; ++si) {
            const Stage s = P.st[si];
            const unsigned short* W = s.W + (long long)l * s.layer_stride;
            const int kp = s.K / s.split, nchunk = kp / 8;
            const int ngroups = ((s.N + R - 1) / R) * s.split;
            const int gidx = l * P.nstage + si;                       // global stage index: input = act[gidx & 1], output = act[~gidx & 1]
            const float* xin = P.act[gidx & 1];
            float* yout = P.act[(gidx + 1) & 1];
            if (!have && P.prefetch) {                                 // first stage of the launch: prefetch behind nothing
                const unsigned short* rowp[R];
                int piece, row0;
                group_rows(s, W, wave, ngroups, rowp, piece, row0);
                issue_block<true>(wa, rowp, 0, lane, nchunk);
                have = true;
            }
            if (persistent && gidx > 0) {                              // wait until every workgroup finished the previous stage
                if (tid == 0) {
                    const unsigned target = epoch * gridDim.x;
                    int spins = 0;
                    while (!dead && __hip_atomic_load(P.counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                        __builtin_amdgcn_s_sleep(1);
                        if (++spins > (1 << 17)) { atomicExch(P.err, 1); dead = true; }
                    }
                }
                __syncthreads();
            }
            // x = sum of the producer's `split` partials (sc1 loads: the producers wrote through to memory), optional norm
            const int psplit = P.st[(si + P.nstage - 1) % P.nstage].split;
            const int pn = P.st[(si + P.nstage - 1) % P.nstage].N;
            {
                const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xin), 0, psplit * pn * 4, 0x00020000);
                float ss = 0.f;
                for (int k4 = tid * 4; k4 < s.K; k4 += WAVES * 64 * 4) {
                    v4u t[4];
#pragma unroll
                    for (int p = 0; p < 4; ++p)                        // all pieces in flight at once (clamped index, weight 0 past psplit)
                        t[p] = __builtin_amdgcn_raw_buffer_load_b128(rs, (min(p, psplit - 1) * pn + (k4 % pn)) * 4, 0, 16);
                    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        const float m = p < psplit ? 1.f : 0.f;
                        a.x += m * __uint_as_float(t[p].x); a.y += m * __uint_as_float(t[p].y);
                        a.z += m * __uint_as_float(t[p].z); a.w += m * __uint_as_float(t[p].w);
                    }
                    ss += a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w;
                    *reinterpret_cast<float4*>(xs + xs_off(k4)) = a;
                }
                if (s.norm) {
                    ss = wave_sum(ss);
                    if (lane == 0) red[wid] = ss;
                    __syncthreads();
                    float t = 0.f;
#pragma unroll
                    for (int w = 0; w < WAVES; ++w) t += red[w];
                    const float rr = rsqrtf(t / s.K + 1e-5f);
                    for (int k4 = tid * 4; k4 < s.K; k4 += WAVES * 64 * 4) {
                        float4* q = reinterpret_cast<float4*>(xs + xs_off(k4));
                        float4 a = *q;
                        *q = make_float4(a.x * rr, a.y * rr, a.z * rr, a.w * rr);
                    }
                }
            }
            __syncthreads();
            for (int g = wave; g < ngroups; g += nwaves) {
                const unsigned short* rowp[R];
                int piece, row0;
                group_rows(s, W, g, ngroups, rowp, piece, row0);
                const int xbase = piece * kp;
                float acc[R] = {0.f, 0.f};
                int c0 = 0;
                if (have) { consume_block<true>(wa, xs, xbase, 0, lane, nchunk, acc); c0 = STEP; have = false; }
                const int cfull = (nchunk / STEP) * STEP;
                for (; c0 < cfull; c0 += STEP) {
                    issue_block<true>(wa, rowp, c0, lane, nchunk);
                    consume_block<true>(wa, xs, xbase, c0, lane, nchunk, acc);
                }
                if (c0 < nchunk) {
                    issue_block<false>(wa, rowp, c0, lane, nchunk);
                    consume_block<false>(wa, xs, xbase, c0, lane, nchunk, acc);
                }
#pragma unroll
                for (int r = 0; r < R; ++r) acc[r] = wave_sum(acc[r]);
                if (lane == 0) {
#pragma unroll
                    for (int r = 0; r < R; ++r)
                        if (row0 + r < s.N)
                            __hip_atomic_store(reinterpret_cast<unsigned*>(yout) + piece * s.N + row0 + r, __float_as_uint(acc[r] * 1e-3f),
                                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            have = false;
            if (persistent) {
                // outputs visible (write-through acknowledged) before this workgroup arrives at the barrier
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const bool last = (l == l1 - 1) && (si == s1 - 1);
                if (!last && P.prefetch) {                             // next stage's first block: in flight across the barrier
                    const int sn = (si + 1) % P.nstage, ln = l + (si + 1 == P.nstage);
                    const Stage n = P.st[sn];
                    const unsigned short* rowp[R];
                    int piece, row0;
                    group_rows(n, n.W + (long long)ln * n.layer_stride, wave, ((n.N + R - 1) / R) * n.split, rowp, piece, row0);
                    issue_block<true>(wa, rowp, 0, lane, n.K / n.split / 8);
                    have = true;
                }
                __syncthreads();                                        // the LDS image is free again, every wave's stores are out
                ++epoch;
                if (tid == 0) __hip_atomic_fetch_add(P.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

extern "C" int layer_probe_run(const Params* p, int waves, int blocks, int only_layer, int only_stage, size_t lds, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (waves == 16) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&layer_probe_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        layer_probe_kernel<16><<<blocks, 1024, lds, st>>>(*p, only_layer, only_stage);
    } else if (waves == 8) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&layer_probe_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        layer_probe_kernel<8><<<blocks, 512, lds, st>>>(*p, only_layer, only_stage);
    } else if (waves == 4) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&layer_probe_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        layer_probe_kernel<4><<<blocks, 256, lds, st>>>(*p, only_layer, only_stage);
    } else return -1;
    return (int)hipGetLastError();
}

// the same code as one launch per (layer, stage)
extern "C" int layer_probe_run_multi(const Params* p, int waves, int blocks, size_t lds, void* stream) {
    for (int l = 0; l < p->layers; ++l)
        for (int s = 0; s < p->nstage; ++s) {
            const int rc = layer_probe_run(p, waves, blocks, l, s, lds, stream);
            if (rc) return rc;
        }
    return 0;
}
