// On-box MFMA ceiling: every wave issues independent MFMAs back to back from registers (no LDS, no memory, no barriers).
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/mfma_probe.hip -o tools/libmfma_probe.so
#include <hip/hip_runtime.h>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) int i32x8;

template <int MODE>
__global__ __launch_bounds__(256) void mfma_probe(float* out, int iters) {
    const int lane = threadIdx.x;
    if (MODE == 0) {            // v_mfma_f32_16x16x32_bf16, 8 independent accumulators
        bf16x8 a = {(short)lane, 1, 2, 3, 4, 5, 6, 7}, b = {7, 6, 5, 4, 3, 2, 1, (short)lane};
        f32x4 acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
        if (s == 123.f) out[0] = s;
    } else if (MODE == 1) {     // v_mfma_f32_32x32x16_bf16, 4 independent accumulators
        bf16x8 a = {(short)lane, 1, 2, 3, 4, 5, 6, 7}, b = {7, 6, 5, 4, 3, 2, 1, (short)lane};
        f32x16 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) s += acc[i][0];
        if (s == 123.f) out[0] = s;
    } else {                    // v_mfma_scale_f32_16x16x128_f8f6f4 (fp8 e4m3 operands, unit scales)
        i32x8 a = {lane, 1, 2, 3, 4, 5, 6, 7}, b = {7, 6, 5, 4, 3, 2, 1, lane};
        f32x4 acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc[i], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) s += acc[i][0];
        if (s == 123.f) out[0] = s;
    }
}

// returns FLOPs issued by the launch; the caller times it
extern "C" double mfma_probe_run(int mode, float* out, int blocks, int iters, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    const double waves = (double)blocks * 4;
    if (mode == 0) { mfma_probe<0><<<blocks, 256, 0, st>>>(out, iters); return waves * iters * 8 * 2.0 * 16 * 16 * 32; }
    if (mode == 1) { mfma_probe<1><<<blocks, 256, 0, st>>>(out, iters); return waves * iters * 4 * 2.0 * 32 * 32 * 16; }
    mfma_probe<2><<<blocks, 256, 0, st>>>(out, iters);
    return waves * iters * 8 * 2.0 * 16 * 16 * 128;
}
