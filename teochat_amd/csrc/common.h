// Shared device/host helpers for libteo_hip (gfx950 only: wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/teo_hip.h"

namespace teo {

typedef unsigned short bf16_t;   // raw bf16 bits

// ---- error plumbing -------------------------------------------------------------------------
void set_error(const char* fmt, ...);
int hip_fail(hipError_t e, const char* what);
// diagnostics: name of the kernel family the last dispatching entry point (GEMM / attention) chose -- teo_last_kernel()
void note_kernel(const char* name);

#define TEO_CHECK_ARG(cond, ...)                   \
    do {                                           \
        if (!(cond)) {                             \
            teo::set_error(__VA_ARGS__);           \
            return TEO_ERR_ARG;                    \
        }                                          \
    } while (0)

#define TEO_LAUNCH_CHECK(name)                                   \
    do {                                                         \
        hipError_t e__ = hipGetLastError();                      \
        if (e__ != hipSuccess) return teo::hip_fail(e__, name);  \
    } while (0)

// ---- launches of the decode path -------------------------------------------------------------
// Every kernel of the decode step is launched through TEO_KLAUNCH so that teo_llama_decode_step_profile can time each launch
// by its own dispatch timestamps (hipExtLaunchKernel start / stop events: kernel execution only, what rocprofv3 reports) and so
// that launch flags (hipExtAnyOrderLaunch experiments) can be applied in one place.  Off by default: a plain <<<>>> launch.
bool prof_take(hipEvent_t* start, hipEvent_t* stop);      // false unless a profiled step is being recorded on this thread
void prof_class(int cls);                                 // class of the launches that follow
void prof_bump(int delta);
extern thread_local unsigned g_launch_flags;
#define TEO_KLAUNCH(kern, grid, block, lds, st, ...)                                                                       \
    do {                                                                                                                   \
        hipEvent_t ps__ = nullptr, pe__ = nullptr;                                                                         \
        if (teo::prof_take(&ps__, &pe__) || teo::g_launch_flags)                                                           \
            hipExtLaunchKernelGGL(kern, dim3(grid), dim3(block), (uint32_t)(lds), st, ps__, pe__, teo::g_launch_flags, __VA_ARGS__); \
        else                                                                                                               \
            hipLaunchKernelGGL(kern, dim3(grid), dim3(block), (uint32_t)(lds), st, __VA_ARGS__);                             \
    } while (0)

// ---- launch chain of the overlapped decode step ------------------------------------------------
// The decode step is a strictly linear chain of kernels (each consumes what its predecessor wrote).  Launched with the AQL
// barrier bit cleared (hipExtAnyOrderLaunch), a queue still DISPATCHES its packets in order -- all workgroups of kernel i are
// placed before the first workgroup of kernel i+1 -- but kernel i+1 no longer waits for kernel i to finish: its workgroups take
// the CU slots the tail of kernel i frees, request their first block of weights (independent of kernel i) and only then wait
// for kernel i's completion word.  The HBM pipe does not drain at the seam.  Deadlock-free by the in-order dispatch: whatever
// a waiting workgroup waits for has all its workgroups resident already.
//   progress : one monotonic word = number of completed launches of the chain; launch `seq` waits for progress >= seq
//   tickets  : ring of arrival counters; the workgroup that draws the last ticket of launch `seq` re-arms the counter and
//              publishes progress = seq + 1
// Visibility without fences (per-XCD L2s are not coherent): everything a later kernel of the chain reads is stored write-through
// (`sc1`) and loaded with `sc1` loads; a storing wave drains its stores (s_waitcnt vmcnt(0)) before the workgroup's ticket.
struct Chain {
    unsigned* progress;
    unsigned* tickets;
    int* err;            // set to 1 when a wait gives up (bounded spin: never hangs the GPU)
    unsigned seq;
    unsigned on;         // 0: plain kernel (ordinary launch order, plain loads / stores)
};
constexpr unsigned CHAIN_RING = 64;
__device__ __forceinline__ void chain_wait(const Chain& c) {
    if (!c.on) return;
    if (threadIdx.x == 0) {
        int spins = 0;
        while ((int)(__hip_atomic_load(c.progress, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - c.seq) < 0) {
            __builtin_amdgcn_s_sleep(16);          // ~0.4 us between polls: a few hundred pollers must not hammer one memory channel
            if (++spins > (1 << 22)) { __hip_atomic_store(c.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
        }
    }
    __syncthreads();
}
__device__ __forceinline__ void chain_signal(const Chain& c) {        // every thread of the workgroup calls it, after its stores
    if (!c.on) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned n = gridDim.x * gridDim.y * gridDim.z;
        unsigned* tk = c.tickets + (c.seq % CHAIN_RING);
        const unsigned t = __hip_atomic_fetch_add(tk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t == n - 1) {
            __hip_atomic_store(tk, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(c.progress, c.seq + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
// activations that cross kernels of the chain: `coh` selects the write-through / L2-bypassing form (buffer ops with the sc1 bit)
typedef unsigned teo_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned teo_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint4 act_ld16(const void* base, unsigned byte_off, bool coh) {
    if (coh) {
        const teo_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(__builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7ffffffc, 0x00020000),
                                                                  byte_off, 0, 16);
        return make_uint4(v.x, v.y, v.z, v.w);
    }
    return *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned char*>(base) + byte_off);
}
__device__ __forceinline__ unsigned act_ld4(const void* base, unsigned byte_off, bool coh) {
    if (coh) return __builtin_amdgcn_raw_buffer_load_b32(__builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7ffffffc, 0x00020000), byte_off, 0, 16);
    return *reinterpret_cast<const unsigned*>(reinterpret_cast<const unsigned char*>(base) + byte_off);
}
__device__ __forceinline__ unsigned short act_ld2(const void* base, unsigned byte_off, bool coh) {
    if (coh) return (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(__builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7ffffffc, 0x00020000), byte_off, 0, 16);
    return *reinterpret_cast<const unsigned short*>(reinterpret_cast<const unsigned char*>(base) + byte_off);
}
__device__ __forceinline__ void act_st16(void* base, unsigned byte_off, const uint4& v, bool coh) {
    if (coh) { const teo_u32x4 t = {v.x, v.y, v.z, v.w};
               __builtin_amdgcn_raw_buffer_store_b128(t, __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7ffffffc, 0x00020000), byte_off, 0, 16); }
    else *reinterpret_cast<uint4*>(reinterpret_cast<unsigned char*>(base) + byte_off) = v;
}
__device__ __forceinline__ void act_st8(void* base, unsigned byte_off, const uint2& v, bool coh) {
    if (coh) { const teo_u32x2 t = {v.x, v.y};
               __builtin_amdgcn_raw_buffer_store_b64(t, __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7ffffffc, 0x00020000), byte_off, 0, 16); }
    else *reinterpret_cast<uint2*>(reinterpret_cast<unsigned char*>(base) + byte_off) = v;
}
__device__ __forceinline__ void act_st4(void* base, unsigned byte_off, unsigned v, bool coh) {
    if (coh) __builtin_amdgcn_raw_buffer_store_b32(v, __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7ffffffc, 0x00020000), byte_off, 0, 16);
    else *reinterpret_cast<unsigned*>(reinterpret_cast<unsigned char*>(base) + byte_off) = v;
}
__device__ __forceinline__ void act_st2(void* base, unsigned byte_off, unsigned short v, bool coh) {
    if (coh) __builtin_amdgcn_raw_buffer_store_b16((short)v, __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7ffffffc, 0x00020000), byte_off, 0, 16);
    else *reinterpret_cast<unsigned short*>(reinterpret_cast<unsigned char*>(base) + byte_off) = v;
}

// ---- bf16 <-> f32 -----------------------------------------------------------------------------
__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((unsigned)v) << 16); }
// round-to-nearest-even (v_cvt_pk_bf16_f32: one instruction for two values, same rounding as torch's float->bfloat16)
typedef float teo_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 teo_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf2(float lo, float hi) {
    const teo_f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, teo_bf16x2));
}
__device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)(pack_bf2(f, 0.f) & 0xffffu); }

template <typename T> struct Elem;
template <> struct Elem<float> {
    static constexpr int kDtype = TEO_F32;
    __device__ static __forceinline__ float ld(const float* p) { return *p; }
    __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
    __device__ static __forceinline__ float round(float v) { return v; }
};
template <> struct Elem<bf16_t> {
    static constexpr int kDtype = TEO_BF16;
    __device__ static __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
    __device__ static __forceinline__ void st(bf16_t* p, float v) { *p = f2bf(v); }
    __device__ static __forceinline__ float round(float v) { return bf2f(f2bf(v)); }
};

// ---- wave / block reductions ----------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// block-wide sum for blockDim.x == NT (multiple of 64); `red` has NT/64 floats of LDS
template <int NT>
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    __syncthreads();
    if (l == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < NT / 64; ++i) t += red[i];
    return t;
}

// ---- activations -------------------------------------------------------------------------------
__device__ __forceinline__ float act_apply(float x, int act) {
    if (act == TEO_ACT_GELU_ERF) return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
    if (act == TEO_ACT_QUICK_GELU) return x / (1.0f + expf(-1.702f * x));
    return x;
}
__device__ __forceinline__ float silu(float x) { return x / (1.0f + expf(-x)); }

inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

}  // namespace teo
