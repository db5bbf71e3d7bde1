// Read-bandwidth probe: plain 16-byte global loads to VGPRs vs LDS-DMA (global_load_lds) + ds_read, same bytes.
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/stream_probe.hip -o tools/libstream_probe.so
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned v4u __attribute__((ext_vector_type(4)));

// each wave streams `per_wave` bytes (multiple of 8 KiB) starting at base + wave_id * per_wave
template <bool NT>
__global__ __launch_bounds__(256) void probe_vgpr(const unsigned char* __restrict__ base, float* __restrict__ out,
                                                  long long per_wave) {
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const unsigned char* p = base + wave * per_wave + lane * 16;
    float acc = 0.f;
    for (long long off = 0; off < per_wave; off += 8192) {
        v4u r[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const v4u* q = reinterpret_cast<const v4u*>(p + off + i * 1024);
            r[i] = NT ? __builtin_nontemporal_load(q) : *q;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) acc += __uint_as_float(r[i].x) + __uint_as_float(r[i].w);
    }
    if (acc == 123.456f) out[0] = acc;
}

// LDS-DMA: each wave owns a ring of SLOTS x 1 KiB; 8 DMA pieces (8 KiB) are issued per step, consumed one step later
template <int SLOTS>
__global__ __launch_bounds__(256) void probe_glds(const unsigned char* __restrict__ base, float* __restrict__ out,
                                                  long long per_wave) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const long long wave = (long long)blockIdx.x * 4 + wid;
    const unsigned char* p = base + wave * per_wave + lane * 16;
    unsigned char* ring = smem + wid * (SLOTS * 1024);
    float acc = 0.f;
    const long long nsteps = per_wave / 8192;
    // prologue: step 0
#pragma unroll
    for (int i = 0; i < 8; ++i)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p + i * 1024),
                                         (__attribute__((address_space(3))) void*)(ring + i * 1024), 16, 0, 0);
    for (long long s = 0; s < nsteps; ++s) {
        const int cur = (int)(s & 1) * 8, nxt = 8 - cur;
        if (s + 1 < nsteps) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void*)(p + (s + 1) * 8192 + i * 1024),
                    (__attribute__((address_space(3))) void*)(ring + (nxt + i) * 1024), 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const v4u r = *reinterpret_cast<const v4u*>(ring + (cur + i) * 1024 + lane * 16);
            acc += __uint_as_float(r.x) + __uint_as_float(r.w);
        }
    }
    if (acc == 123.456f) out[0] = acc;
}

// GEMV-like access: matrix [N][rowbytes]; a wave owns R rows and reads U consecutive 1-KiB chunks of each per step
template <int R, int U>
__global__ __launch_bounds__(256) void probe_rows(const unsigned char* __restrict__ base, float* __restrict__ out,
                                                  int N, long long rowbytes) {
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long nwaves = (long long)gridDim.x * 4;
    float acc = 0.f;
    for (long long g = wave; g * R < N; g += nwaves) {
        const unsigned char* p = base + g * R * rowbytes + lane * 16;
        for (long long off = 0; off < rowbytes; off += 1024 * U) {
            v4u r[R][U];
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int i = 0; i < R; ++i)
                    r[i][u] = __builtin_nontemporal_load(reinterpret_cast<const v4u*>(p + i * rowbytes + off + u * 1024));
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int i = 0; i < R; ++i) acc += __uint_as_float(r[i][u].x) + __uint_as_float(r[i][u].w);
        }
    }
    if (acc == 123.456f) out[0] = acc;
}

extern "C" int probe_rows_run(int R, int U, const void* base, float* out, int N, long long rowbytes, int blocks, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    const unsigned char* b = (const unsigned char*)base;
    if (R == 4 && U == 2) probe_rows<4, 2><<<blocks, 256, 0, st>>>(b, out, N, rowbytes);
    else if (R == 2 && U == 4) probe_rows<2, 4><<<blocks, 256, 0, st>>>(b, out, N, rowbytes);
    else if (R == 1 && U == 8) probe_rows<1, 8><<<blocks, 256, 0, st>>>(b, out, N, rowbytes);
    else if (R == 8 && U == 1) probe_rows<8, 1><<<blocks, 256, 0, st>>>(b, out, N, rowbytes);
    else if (R == 2 && U == 2) probe_rows<2, 2><<<blocks, 256, 0, st>>>(b, out, N, rowbytes);
    else return -1;
    return (int)hipGetLastError();
}

extern "C" int probe_run(int mode, const void* base, float* out, long long total_bytes, int blocks, void* stream) {
    const long long per_wave = total_bytes / ((long long)blocks * 4) / 8192 * 8192;
    hipStream_t st = (hipStream_t)stream;
    if (mode == 0) probe_vgpr<false><<<blocks, 256, 0, st>>>((const unsigned char*)base, out, per_wave);
    else if (mode == 1) probe_vgpr<true><<<blocks, 256, 0, st>>>((const unsigned char*)base, out, per_wave);
    else probe_glds<16><<<blocks, 256, 4 * 16 * 1024, st>>>((const unsigned char*)base, out, per_wave);
    return (int)hipGetLastError();
}
