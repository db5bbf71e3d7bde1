// All-gather hand-off probe for a persistent decode layer (groundwork for round 2).
//
// 256 workgroups (one per CU) stay resident and run `iters` rounds.  In round e every workgroup publishes `gpw` 8-byte
// granules {payload, tag = e} of a shared vector with write-through (sc1) stores, then gathers the WHOLE vector
// (256*gpw granules) by polling each granule with sc1 loads until its tag equals e.  No fences: a granule carries its own
// epoch, 8-byte accesses are single-copy atomic.  Every spin is bounded (give-up -> error flag), so a scheduling
// assumption that does not hold ends the kernel instead of hanging the GPU.
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/handoff_probe.hip -o tools/libhandoff_probe.so
#include <hip/hip_runtime.h>
#include <stdint.h>

__device__ __forceinline__ void st_sc1(unsigned long long* p, unsigned long long v) {
    asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ unsigned long long ld_sc1(const unsigned long long* p) {
    unsigned long long v;
    asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

extern "C" __global__ __launch_bounds__(256) void handoff_allgather(unsigned long long* vec, int gpw, int iters, int* err,
                                                                    unsigned long long* sink, int batched) {
    const int nb = gridDim.x, b = blockIdx.x, tid = threadIdx.x;
    const int total = nb * gpw;
    unsigned long long acc = 0;
    for (int e = 1; e <= iters; ++e) {
        // publish this workgroup's granules (lane-parallel)
        for (int g = tid; g < gpw; g += 256)
            st_sc1(vec + (size_t)b * gpw + g, ((unsigned long long)(unsigned)e << 32) | (unsigned)(b * gpw + g + e));
        // gather everything: 8 polls in flight per lane (mode 1) or one at a time (mode 0)
        if (batched) {
            for (int g0 = tid; g0 < total; g0 += 256 * 8) {
                unsigned long long v[8];
                int spins = 0;
                for (;;) {
                    // all 8 polls are issued back to back (clamped index), one wait for the batch
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const unsigned long long* q = vec + min(g0 + i * 256, total - 1);
                        asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=v"(v[i]) : "v"(q) : "memory");
                    }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    bool all = true;
#pragma unroll
                    for (int i = 0; i < 8; ++i) all = all && ((unsigned)(v[i] >> 32) >= (unsigned)e);
                    if (all) break;
                    if (++spins > 2000000) { atomicExch(err, 1); break; }
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) acc += v[i] & 0xffffffffull;
            }
        } else
        for (int g = tid; g < total; g += 256) {
            unsigned long long v = 0;
            int spins = 0;
            for (;;) {
                v = ld_sc1(vec + g);
                if ((unsigned)(v >> 32) >= (unsigned)e) break;   // a faster workgroup may already be in a later round
                if (++spins > 2000000) { atomicExch(err, 1); break; }
                __builtin_amdgcn_s_sleep(1);
            }
            acc += v & 0xffffffffull;
        }
        __syncthreads();
        if (*err) break;
    }
    if (acc == 0x1234567ull) *sink = acc;
}

extern "C" int handoff_run(void* vec, int blocks, int gpw, int iters, void* err, void* sink, void* stream, int batched) {
    hipLaunchKernelGGL(handoff_allgather, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (unsigned long long*)vec, gpw, iters,
                       (int*)err, (unsigned long long*)sink, batched);
    return (int)hipGetLastError();
}
