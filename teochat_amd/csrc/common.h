// Shared device/host helpers for libteo_hip (gfx950 only: wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/teo_hip.h"
#include "tune.h"

namespace teo {

typedef unsigned short bf16_t;   // raw bits of a 16-bit float.  The MFMA / GEMV kernels take these pointers for BOTH 16-bit formats and
                                 // select the arithmetic with a template flag (F16 = false: bfloat16, true: IEEE binary16, below)
struct f16_t { unsigned short bits; };   // the same bits as a DISTINCT type, for the kernels templated on the element type (Elem<T>)

// ---- error plumbing -------------------------------------------------------------------------
void set_error(const char* fmt, ...);
int hip_fail(hipError_t e, const char* what);
// compute units of the CURRENT device (cached per device id; 0 when the query fails)
int device_cu_count();
// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE attribute: set once per (kernel, device); `mask` is the call site's own
// static word, one bit per device id (a benign race sets it twice)
int lds_attr_once(const void* kernel, int bytes, unsigned long long* mask, const char* what);
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
// by its own dispatch timestamps (hipExtLaunchKernel start / stop events: kernel execution only, what rocprofv3 reports).
// Outside a profiled step it is a plain <<<>>> launch.
bool prof_take(hipEvent_t* start, hipEvent_t* stop);      // false unless a profiled step is being recorded on this thread
void prof_class(int cls);                                 // class of the launches that follow
void prof_bump(int delta);
#define TEO_KLAUNCH(kern, grid, block, lds, st, ...)                                                                       \
    do {                                                                                                                   \
        hipEvent_t ps__ = nullptr, pe__ = nullptr;                                                                         \
        if (teo::prof_take(&ps__, &pe__))                                                                                  \
            hipExtLaunchKernelGGL(kern, dim3(grid), dim3(block), (uint32_t)(lds), st, ps__, pe__, 0, __VA_ARGS__);         \
        else                                                                                                               \
            hipLaunchKernelGGL(kern, dim3(grid), dim3(block), (uint32_t)(lds), st, __VA_ARGS__);                             \
    } while (0)

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

// ---- IEEE binary16 (fp16): the reference's inference dtype (builder.py:105, eval/inference.py:53) -----------------------------
typedef _Float16 teo_f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float h2f_f16(unsigned short v) { return (float)__builtin_bit_cast(_Float16, v); }
__device__ __forceinline__ unsigned pack_f16x2(float lo, float hi) {      // v_cvt_pk_f16_f32 (RNE), like torch's float -> half
    const teo_f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, teo_f16x2));
}
// format-generic forms: F16 = false -> bfloat16, true -> binary16
template <bool F16> __device__ __forceinline__ float h2f(unsigned short v) { return F16 ? h2f_f16(v) : bf2f(v); }
template <bool F16> __device__ __forceinline__ unsigned pack_h2(float lo, float hi) { return F16 ? pack_f16x2(lo, hi) : pack_bf2(lo, hi); }
template <bool F16> __device__ __forceinline__ unsigned short f2h(float f) { return (unsigned short)(pack_h2<F16>(f, 0.f) & 0xffffu); }
template <bool F16> __device__ __forceinline__ float h_lo(unsigned w) { return F16 ? h2f_f16((unsigned short)(w & 0xffffu)) : __uint_as_float(w << 16); }
template <bool F16> __device__ __forceinline__ float h_hi(unsigned w) { return F16 ? h2f_f16((unsigned short)(w >> 16)) : __uint_as_float(w & 0xffff0000u); }
// matrix / dot instructions on 8 (two: 2) packed 16-bit values
typedef __attribute__((ext_vector_type(8))) short teo_h16x8;          // raw operand registers (what the bf16 intrinsics take)
typedef __attribute__((ext_vector_type(8))) _Float16 teo_f16x8;
typedef __attribute__((ext_vector_type(8))) __bf16 teo_bf16x8v;
typedef __attribute__((ext_vector_type(4))) float teo_f32x4;
typedef __attribute__((ext_vector_type(16))) float teo_f32x16;
template <bool F16> __device__ __forceinline__ teo_f32x4 mfma16(teo_h16x8 a, teo_h16x8 b, teo_f32x4 c) {          // v_mfma_f32_16x16x32_{bf16,f16}
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(teo_f16x8, a), __builtin_bit_cast(teo_f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(teo_bf16x8v, a), __builtin_bit_cast(teo_bf16x8v, b), c, 0, 0, 0);
}
template <bool F16> __device__ __forceinline__ teo_f32x16 mfma32(teo_h16x8 a, teo_h16x8 b, teo_f32x16 c) {       // v_mfma_f32_32x32x16_{bf16,f16}
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(teo_f16x8, a), __builtin_bit_cast(teo_f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(teo_bf16x8v, a), __builtin_bit_cast(teo_bf16x8v, b), c, 0, 0, 0);
}
template <bool F16> __device__ __forceinline__ float dot2h(unsigned a, unsigned b, float acc) {                   // v_dot2_f32_{bf16,f16}
    if constexpr (F16) return __builtin_amdgcn_fdot2(__builtin_bit_cast(teo_f16x2, a), __builtin_bit_cast(teo_f16x2, b), acc, false);
    else return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(teo_bf16x2, a), __builtin_bit_cast(teo_bf16x2, b), acc, false);
}

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

template <> struct Elem<f16_t> {
    static constexpr int kDtype = TEO_F16;
    __device__ static __forceinline__ float ld(const f16_t* p) { return h2f_f16(p->bits); }
    __device__ static __forceinline__ void st(f16_t* p, float v) { p->bits = f2h<true>(v); }
    __device__ static __forceinline__ float round(float v) { return h2f_f16(f2h<true>(v)); }
};
// the F16 template flag of an element type (kernels that take raw 16-bit pointers AND an element type)
template <typename T> struct IsF16 { static constexpr bool v = false; };
template <> struct IsF16<f16_t> { static constexpr bool v = true; };

// ---- wave / block reductions ----------------------------------------------------------------
// The xor butterfly 32, 16, 8, 4, 2, 1 without the LDS crossbar (`__shfl_xor` compiles to ds_bpermute_b32 + 5 address VALU ops per
// step, ~100 cycles of latency each): v_permlane32_swap / v_permlane16_swap broadcast the two halves / the odd and even rows, DPP
// row_ror:8/4/2/1 supplies the rest.  Operand VALUES per lane are those of the xor butterfly at every step (after the step with
// offset 2o every lane's value has period 2o inside its row, so lane (i + o) % 16 holds what lane i ^ o holds; + and max commute),
// hence every lane ends with the bits the __shfl_xor form produced (tools/wave_probe.py checks that on the device).
template <int CTRL>
__device__ __forceinline__ float dpp_lane(float v) {        // CTRL 0x120 + n: row_ror:n (lane i of a 16-lane row reads lane (i + n) % 16)
    return __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), CTRL, 0xF, 0xF, false));
}
#define TEO_WAVE_BUTTERFLY(OP)                                                                                        \
    {                                                                                                                 \
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);        \
        v = OP(__uint_as_float(r[0]), __uint_as_float(r[1]));                      /* lanes i and i ^ 32 */          \
    }                                                                                                                 \
    {                                                                                                                 \
        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);        \
        v = OP(__uint_as_float(r[0]), __uint_as_float(r[1]));                      /* lanes i and i ^ 16 */          \
    }                                                                                                                 \
    v = OP(v, dpp_lane<0x128>(v));                                                                                    \
    v = OP(v, dpp_lane<0x124>(v));                                                                                    \
    v = OP(v, dpp_lane<0x122>(v));                                                                                    \
    v = OP(v, dpp_lane<0x121>(v));
__device__ __forceinline__ float teo_addf(float a, float b) { return a + b; }
__device__ __forceinline__ float wave_sum(float v) {
    TEO_WAVE_BUTTERFLY(teo_addf)
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
    TEO_WAVE_BUTTERFLY(fmaxf)
    return v;
}
#undef TEO_WAVE_BUTTERFLY
// v[i] + v[i ^ O] in every lane, bitwise what `v + __shfl_xor(v, O, 64)` gives: O = 32 / 16 through the permlane swaps, 8 through row_ror:8
// (inside a 16-lane row (i + 8) % 16 == i ^ 8), 2 / 1 through quad_perm; 4 has no exact DPP pattern on this target and stays a shuffle
template <int O>
__device__ __forceinline__ float xor_add(float v) {
    if constexpr (O == 32) {
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        return __uint_as_float(r[0]) + __uint_as_float(r[1]);
    } else if constexpr (O == 16) {
        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        return __uint_as_float(r[0]) + __uint_as_float(r[1]);
    } else if constexpr (O == 8) {
        return v + dpp_lane<0x128>(v);
    } else if constexpr (O == 2) {
        return v + dpp_lane<0x4E>(v);               // quad_perm:[2,3,0,1]
    } else if constexpr (O == 1) {
        return v + dpp_lane<0xB1>(v);               // quad_perm:[1,0,3,2]
    } else {
        return v + __shfl_xor(v, O, 64);
    }
}
// butterfly W/2, ..., 2, 1 inside aligned groups of W lanes (W = 8 .. 64): the bits of the __shfl_xor loop.  The offset-4 step follows the
// offset-8 step whenever W >= 16, so row_ror:4 reads a lane that holds what lane i ^ 4 holds (see wave_sum)
template <int W>
__device__ __forceinline__ float group_sum(float v) {
    if constexpr (W >= 64) v = xor_add<32>(v);
    if constexpr (W >= 32) v = xor_add<16>(v);
    if constexpr (W >= 16) { v = xor_add<8>(v); v = v + dpp_lane<0x124>(v); }
    else if constexpr (W >= 8) v = xor_add<4>(v);
    if constexpr (W >= 4) v = xor_add<2>(v);
    if constexpr (W >= 2) v = xor_add<1>(v);
    return v;
}
// butterfly L, 2L, ..., 32 ACROSS the groups of L lanes (ascending offsets): the bits of `for (o = L; o < 64; o <<= 1) v += __shfl_xor(v, o)`
template <int L>
__device__ __forceinline__ float cross_group_sum(float v) {
    if constexpr (L <= 1) v = xor_add<1>(v);
    if constexpr (L <= 2) v = xor_add<2>(v);
    if constexpr (L <= 4) v = xor_add<4>(v);
    if constexpr (L <= 8) v = xor_add<8>(v);
    if constexpr (L <= 16) v = xor_add<16>(v);
    if constexpr (L <= 32) v = xor_add<32>(v);
    return v;
}
// the 32-bit value lane i ^ O holds, O = 32 / 16 (permlane swaps + a select on the lane's half / row parity)
template <int O>
__device__ __forceinline__ unsigned xor_get(unsigned v) {
    static_assert(O == 32 || O == 16, "xor_get: the exact exchanges");
    const unsigned lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    if constexpr (O == 32) {
        const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);       // r[0] = lower half twice, r[1] = upper half twice
        return (lane & 32u) ? r[0] : r[1];
    } else {
        const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);       // r[0] = rows 0 0 2 2, r[1] = rows 1 1 3 3
        return (lane & 16u) ? r[0] : r[1];
    }
}
// wave-wide argmax, ties to the smaller index: the __shfl_xor butterfly 32 .. 1 of (value, index) pairs.  After every step both partners
// hold the same pair (the rule is symmetric), so the row_ror steps read what lane i ^ o holds (as in wave_sum); NaN-free inputs.
__device__ __forceinline__ void wave_argmax(float& best, int& bi) {
#define TEO_ARGMAX_STEP(OV, OI)                                                                   \
    {                                                                                             \
        const float ov_ = (OV);                                                                   \
        const int oi_ = (OI);                                                                     \
        if (ov_ > best || (ov_ == best && oi_ < bi)) { best = ov_; bi = oi_; }                    \
    }
    TEO_ARGMAX_STEP(__uint_as_float(xor_get<32>(__float_as_uint(best))), (int)xor_get<32>((unsigned)bi))
    TEO_ARGMAX_STEP(__uint_as_float(xor_get<16>(__float_as_uint(best))), (int)xor_get<16>((unsigned)bi))
    TEO_ARGMAX_STEP(dpp_lane<0x128>(best), (int)__float_as_uint(dpp_lane<0x128>(__uint_as_float((unsigned)bi))))
    TEO_ARGMAX_STEP(dpp_lane<0x124>(best), (int)__float_as_uint(dpp_lane<0x124>(__uint_as_float((unsigned)bi))))
    TEO_ARGMAX_STEP(dpp_lane<0x122>(best), (int)__float_as_uint(dpp_lane<0x122>(__uint_as_float((unsigned)bi))))
    TEO_ARGMAX_STEP(dpp_lane<0x121>(best), (int)__float_as_uint(dpp_lane<0x121>(__uint_as_float((unsigned)bi))))
#undef TEO_ARGMAX_STEP
}
// the same reductions through __shfl_xor (reference form of tools/wave_probe.hip)
__device__ __forceinline__ float wave_sum_shfl(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max_shfl(float v) {
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
