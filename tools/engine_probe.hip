// Loader / consumer decode-layer engine probe (the structure MI355X_MICROARCH.md prices at 0.87-0.89x of five launches for a 1B
// model): ONE persistent launch, one workgroup per CU, wave 0 streams the CU's share of every weight matrix of the layer through an
// LDS ring with LDS-DMA (non-temporal), waves 1-3 consume ring slots (v_dot2c_f32_bf16 against a bf16 image of x in LDS) and publish
// their outputs as 8-byte {tag, 2 x bf16} granules with write-through stores; before the next matrix one consumer wave gathers the
// whole output vector of the previous one by polling the granules (sc1 loads) into the x image.  The loader never waits for an edge:
// it runs ahead as far as the ring allows.  Synthetic chain qkv -> KV stream -> o -> gate/up -> down at LLaMA-2-7B sizes (the
// attention arithmetic is replaced by a GEMV over the same bytes); every spin is bounded (error flag instead of a hang).
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/engine_probe.hip -o tools/libengine_probe.so
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

constexpr int NOP_MAX = 8;
#ifndef PROBE_FILL_PIECES
#define PROBE_FILL_PIECES 16
#endif
constexpr int FILL_PIECES = PROBE_FILL_PIECES;          // 1-KiB pieces per fill (build with -DPROBE_FILL_PIECES=8 for 8 KB fills)
constexpr int SLOT_BYTES = FILL_PIECES * 1024;
constexpr int NSLOT = 131072 / SLOT_BYTES;               // 128 KB ring: 8 x 16 KB or 16 x 8 KB
constexpr int XIMG_BYTES = 24576;        // bf16 image of x, K <= 12288

struct Op {
    const unsigned short* W;             // [layers][N][K] bf16, K * 2 bytes a multiple of 1024
    long long layer_stride;              // elements
    int N, K;                            // rows, row length
    int k_in;                            // elements of x this op consumes (= K)
};
struct EParams {
    Op op[NOP_MAX];
    int nop, layers;
    unsigned long long* vec[2];          // granule vectors (ping-pong by global op index parity), 16384 granules each
    int* err;
    int use_nt;
    int mode;                            // 1: consumers release slots without reading them (loader-bound rate)
};

__device__ __forceinline__ float dot2(unsigned a, unsigned b, float acc) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, a), __builtin_bit_cast(bf16x2, b), acc, false);
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ void fail(int* err, int code, int a, int b, int c) {
    if (atomicCAS(err, 0, code) == 0) { err[1] = a; err[2] = b; err[3] = c; }
}

constexpr int NLOAD = 2, NCONS = 6;   // 8 waves: a single loader wave issues at most ~15 GB/s of LDS-DMA (one 1-KiB piece per ~150 cycles)

__global__ __launch_bounds__(512) void engine_probe_kernel(EParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* ring = smem;
    unsigned char* ximg = smem + NSLOT * SLOT_BYTES;
    volatile int* ready = reinterpret_cast<volatile int*>(ximg + XIMG_BYTES);      // [NSLOT]: (fill index + 1) landed in the slot
    int* done = const_cast<int*>(ready) + 16;                                      // [NSLOT]: pieces consumed from the slot, cumulative
    volatile int* xready = reinterpret_cast<volatile int*>(const_cast<int*>(ready) + 32);   // epoch of the x image
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cu = blockIdx.x, ncu = gridDim.x;
    if (tid < 40) const_cast<int*>(ready)[tid] = 0;
    __syncthreads();

    if (wid < NLOAD) {
        // ---------------------------------------------------------------- loader
        int f = 0;                                                 // global fill index of this CU
        int lwait = 0, lstalls = 0;
        int last1 = -1, last2 = -1, last3 = -1;                    // this wave's most recent fills (not yet published)
        for (int l = 0; l < P.layers; ++l) {
            for (int o = 0; o < P.nop; ++o) {
                const Op op = P.op[o];
                const int R = op.N / ncu, ppr = op.K * 2 / 1024;
                const int npieces = R * ppr, nfill = (npieces + FILL_PIECES - 1) / FILL_PIECES;
                const unsigned char* base = reinterpret_cast<const unsigned char*>(op.W + (long long)l * op.layer_stride + (long long)cu * R * op.K);
                for (int i = 0; i < nfill; ++i, ++f) {
                    if ((f % NLOAD) != wid) continue;             // loaders take the fills round-robin
                    const int slot = f % NSLOT;
                    const int need = (f / NSLOT) * FILL_PIECES;      // all earlier uses of the slot fully consumed
                    int spins = 0;
                    if (P.mode < 2 && __hip_atomic_load(done + slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < need) {
                        // about to wait for the consumers: first publish everything of ours that is in flight (they may be waiting for it)
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        if (lane == 0) {
                            if (last3 >= 0) ready[last3 % NSLOT] = last3 + 1;
                            if (last2 >= 0) ready[last2 % NSLOT] = last2 + 1;
                            if (last1 >= 0) ready[last1 % NSLOT] = last1 + 1;
                        }
                        last1 = last2 = last3 = -1;
                        while (__hip_atomic_load(done + slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < need) {
                            __builtin_amdgcn_s_sleep(1);
                            if (++spins > (1 << 20)) { fail(P.err, 1, f, done[slot], need); return; }
                        }
                        lwait += spins; ++lstalls;
                    }
                    const int valid = min(FILL_PIECES, npieces - i * FILL_PIECES);
#pragma unroll
                    for (int j = 0; j < FILL_PIECES; ++j) {
                        const int piece = min(i * FILL_PIECES + j, npieces - 1);
                        const unsigned char* src = base + (long long)piece * 1024 + lane * 16;
                        if (P.use_nt)
                            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                             (__attribute__((address_space(3))) void*)(ring + slot * SLOT_BYTES + j * 1024), 16, 0, 2);
                        else
                            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                             (__attribute__((address_space(3))) void*)(ring + slot * SLOT_BYTES + j * 1024), 16, 0, 0);
                    }
                    if (valid < FILL_PIECES && lane == 0) __hip_atomic_fetch_add(done + slot, FILL_PIECES - valid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    // fills are retired in order: with this one issued, fill f-2 has landed
                    // this wave's fills are retired in order: with this one issued, its fill two turns back has landed
                    if (FILL_PIECES == 16) {
                        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");  // one own fill of 16 loads may fly behind this one: the one before has landed
                        if (last1 >= 0 && lane == 0) ready[last1 % NSLOT] = last1 + 1;
                        last1 = f;
                    } else {
                        asm volatile("s_waitcnt vmcnt(24)" ::: "memory");  // three own fills of 8 loads may fly
                        if (last3 >= 0 && lane == 0) ready[last3 % NSLOT] = last3 + 1;
                        last3 = last2; last2 = last1; last1 = f;
                    }
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) {
            if (last3 >= 0) ready[last3 % NSLOT] = last3 + 1;
            if (last2 >= 0) ready[last2 % NSLOT] = last2 + 1;
            if (last1 >= 0) ready[last1 % NSLOT] = last1 + 1;
            if (cu == 0 && wid == 0) { P.err[4] = lwait; P.err[5] = lstalls; P.err[6] = f; }
        }
        return;
    }

    // -------------------------------------------------------------------- consumers (waves 1..3)
    if (P.mode >= 2) return;                                       // modes 2 / 3: the loader alone, free-running over the ring
    const int cw = wid - NLOAD;
    int fbase = 0;                                                 // first fill index of the current op
    int cwait = 0, cfills = 0;
    for (int l = 0; l < P.layers; ++l) {
        for (int o = 0; o < P.nop; ++o) {
            const Op op = P.op[o];
            const int gidx = l * P.nop + o;                        // global op index; epoch = gidx + 1
            const int R = op.N / ncu, ppr = op.K * 2 / 1024;
            const int npieces = R * ppr, nfill = (npieces + FILL_PIECES - 1) / FILL_PIECES;
            // ---- x image: outputs of the previous op, gathered by consumer 0 (the very first op reads whatever is there)
            if (gidx > 0) {
                // all three consumer waves sweep a third of the granule vector each, 16 polls in flight per lane
                const unsigned long long* vin = P.vec[(gidx - 1) & 1];
                const int ngran = op.k_in / 2;
                const int third = ((ngran + NCONS - 1) / NCONS + 63) / 64 * 64;
                const int g_lo = cw * third, g_hi = min(ngran, g_lo + third);
                for (int g0 = g_lo; g0 < g_hi; g0 += 64 * 16) {
                    v2u v[16];
                    int spins = 0;
                    for (;;) {
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            const unsigned long long* q = vin + min(g0 + i * 64 + lane, g_hi - 1);
                            asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=v"(v[i]) : "v"(q) : "memory");
                        }
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        bool all = true;
#pragma unroll
                        for (int i = 0; i < 16; ++i) all = all && (v[i].x == (unsigned)gidx);
                        if (__all(all)) break;
                        if (++spins > (1 << 18)) { fail(P.err, 2, gidx, g0, (int)v[0].x); break; }
                        __builtin_amdgcn_s_sleep(1);
                    }
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int g = g0 + i * 64 + lane;
                        if (g < g_hi) *reinterpret_cast<unsigned*>(ximg + g * 4) = v[i].y;
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane == 0) __hip_atomic_fetch_add(const_cast<int*>(xready), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                int spins = 0;
                while (*xready < NCONS * gidx) {                      // every consumer wave has written its third
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > (1 << 22)) { fail(P.err, 3, gidx, *xready, cw); return; }
                }
            }
            // ---- rows of this CU: pairs (2p, 2p+1), p = cw, cw + 3, ...
            unsigned long long* vout = P.vec[gidx & 1];
            for (int p = cw; 2 * p < R; p += NCONS) {
                float acc[2] = {0.f, 0.f};
                int cur_fill = -1, cur_cnt = 0;
                for (int q = 0; q < 2 * ppr; ++q) {                 // the pair's pieces are contiguous: rows 2p, 2p+1
                    const int rr = q >= ppr, j = q - rr * ppr;
                    const int piece = 2 * p * ppr + q;
                    const int fill = fbase + piece / FILL_PIECES, slot = fill % NSLOT;
                    if (fill != cur_fill) {                           // one flag poll and one release per fill, not per piece
                        if (cur_fill >= 0) {
                            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // its bytes are in registers before the slot is released
                            if (lane == 0) __hip_atomic_fetch_add(done + cur_fill % NSLOT, cur_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                        cur_fill = fill; cur_cnt = 0;
                        int spins = 0;
                        while (ready[slot] < fill + 1) {
                            __builtin_amdgcn_s_sleep(1);
                            if (++spins > (1 << 22)) { fail(P.err, 4, fill, ready[slot], cu * 8 + cw); return; }
                        }
                        cwait += spins; ++cfills;
                    }
                    ++cur_cnt;
                    if (P.mode != 1) {
                        const v4u w = *reinterpret_cast<const v4u*>(ring + slot * SLOT_BYTES + (piece % FILL_PIECES) * 1024 + lane * 16);
                        const v4u x = *reinterpret_cast<const v4u*>(ximg + (j * 512 + lane * 8) * 2);
                        float a = acc[rr];
                        a = dot2(w.x, x.x, a); a = dot2(w.y, x.y, a); a = dot2(w.z, x.z, a); a = dot2(w.w, x.w, a);
                        acc[rr] = a;
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane == 0) __hip_atomic_fetch_add(done + cur_fill % NSLOT, cur_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                const float a0 = wave_sum(acc[0]) * 1e-3f, a1 = wave_sum(acc[1]) * 1e-3f;
                if (lane == 0) {
                    const unsigned pay = (__float_as_uint(a0) >> 16) | (__float_as_uint(a1) & 0xffff0000u);
                    const unsigned long long gran = ((unsigned long long)pay << 32) | (unsigned)(gidx + 1);
                    unsigned long long* dst = vout + (size_t)cu * (R / 2) + p;
                    asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(dst), "v"(gran) : "memory");
                }
            }
            fbase += nfill;
        }
    }
    if (cu == 0 && cw == 0 && lane == 0) { P.err[7] = cwait; P.err[8] = cfills; }
}

extern "C" int engine_probe_run(const EParams* p, int blocks, void* stream) {
    const size_t lds = NSLOT * SLOT_BYTES + XIMG_BYTES + 256;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&engine_probe_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    engine_probe_kernel<<<blocks, 512, lds, (hipStream_t)stream>>>(*p);
    return (int)hipGetLastError();
}
