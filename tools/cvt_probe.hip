// Issue cost of the conversion / pack / MFMA instructions the batched fp8 decode GEMMs are made of (gfx950): every wave issues a
// long run of independent copies of ONE instruction from registers (no memory); the host times the launch.
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/cvt_probe.hip -o tools/libcvt_probe.so
#include <hip/hip_runtime.h>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) int i32x8;

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int MODE>
__global__ void cvt_probe(unsigned* out, int iters) {
    const unsigned lane = threadIdx.x;
    unsigned s0 = lane * 0x01010101u + 0x3c3c3c3cu, s1 = 0x3f800000u;
    unsigned d[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) d[i] = lane + i;
    f32x4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8 a = {(short)lane, 1, 2, 3, 4, 5, 6, 7}, b = {7, 6, 5, 4, 3, 2, 1, (short)lane};
    i32x8 a8 = {(int)lane, 1, 2, 3, 4, 5, 6, 7}, b8 = {7, 6, 5, 4, 3, 2, 1, (int)lane};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (MODE == 0) {          // baseline: v_and_b32
#define X(i) asm volatile("v_and_b32 %0, %1, %2" : "=v"(d[i]) : "v"(s0), "v"(s1));
                REP8(X)
#undef X
            } else if (MODE == 1) {   // v_cvt_scalef32_pk_bf16_fp8 (2 fp8 -> 2 bf16, scale operand)
#define X(i) asm volatile("v_cvt_scalef32_pk_bf16_fp8 %0, %1, %2" : "=v"(d[i]) : "v"(s0), "v"(s1));
                REP8(X)
#undef X
            } else if (MODE == 2) {   // v_cvt_pk_f32_fp8 (2 fp8 -> 2 f32)
                unsigned long long q[8];
#define X(i) asm volatile("v_cvt_pk_f32_fp8 %0, %1" : "=v"(q[i]) : "v"(s0));
                REP8(X)
#undef X
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("" :: "v"(q[i]));
            } else if (MODE == 3) {   // v_cvt_pk_bf16_f32 (2 f32 -> 2 bf16)
#define X(i) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(d[i]) : "v"(s0), "v"(s1));
                REP8(X)
#undef X
            } else if (MODE == 4) {   // v_perm_b32
#define X(i) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(d[i]) : "v"(s0), "v"(s1), "v"(lane));
                REP8(X)
#undef X
            } else if (MODE == 5) {   // v_cvt_scalef32_pk_f16_fp8
#define X(i) asm volatile("v_cvt_scalef32_pk_f16_fp8 %0, %1, %2" : "=v"(d[i]) : "v"(s0), "v"(s1));
                REP8(X)
#undef X
            } else if (MODE == 6) {   // v_lshl_or_b32
#define X(i) asm volatile("v_lshl_or_b32 %0, %1, 4, %2" : "=v"(d[i]) : "v"(s0), "v"(s1));
                REP8(X)
#undef X
            } else if (MODE == 7) {   // v_mfma_f32_16x16x32_bf16
#define X(i) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
                REP8(X)
#undef X
            } else if (MODE == 8) {   // v_mfma_scale_f32_16x16x128_f8f6f4, fp8 x fp8
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8, b8, acc[i], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
            } else if (MODE == 9) {   // the same with a bf8 (e5m2) B operand
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8, b8, acc[i], 0, 1, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
            } else if (MODE == 10) {  // v_cvt_scalef32_pk_bf8_bf16 (2 bf16 -> 2 bf8; the producer side of a split activation)
#define X(i) asm volatile("v_cvt_scalef32_pk_bf8_bf16 %0, %1, %2" : "+v"(d[i]) : "v"(s0), "v"(s1));
                REP8(X)
#undef X
            } else if (MODE == 11) {  // v_dot2c_f32_bf16
                float f[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) f[i] = 0.f;
#define X(i) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(f[i]) : "v"(s0), "v"(s1));
                REP8(X)
#undef X
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("" :: "v"(f[i]));
            }
        }
    }
    unsigned x = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) x ^= d[i] ^ __float_as_uint(acc[i][0]);
    if (x == 0x12345u) out[0] = x;
}

extern "C" int cvt_probe_run(int mode, unsigned* out, int blocks, int threads, int iters, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    switch (mode) {
#define C_(m) case m: cvt_probe<m><<<blocks, threads, 0, st>>>(out, iters); break;
        C_(0) C_(1) C_(2) C_(3) C_(4) C_(5) C_(6) C_(7) C_(8) C_(9) C_(10) C_(11)
#undef C_
        default: return -1;
    }
    return 32;      // instructions of the kind per loop iteration
}
