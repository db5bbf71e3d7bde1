// Bitwise check of common.h's wave_sum / wave_max (v_permlane32_swap, v_permlane16_swap, DPP row_ror) against the __shfl_xor butterfly they
// replace, in every lane, on random / special inputs; and a latency comparison (a dependent chain of reductions timed with s_memtime).
#include <hip/hip_runtime.h>
#include "../teochat_amd/csrc/common.h"
using namespace teo;
__global__ void wave_probe_kernel(const float* in, unsigned* diff, int n_waves) {
    const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (w >= n_waves) return;
    const float v = in[w * 64 + (threadIdx.x & 63)];
    const float a = wave_sum(v), b = wave_sum_shfl(v), c = wave_max(v), d = wave_max_shfl(v);
    unsigned bad = 0;
    if (__float_as_uint(a) != __float_as_uint(b)) bad |= 1u;
    if (__float_as_uint(c) != __float_as_uint(d) && !(c != c && d != d)) bad |= 2u;
    // the partial butterflies of the decode attention / batched GEMM epilogues
    {
        float e = v, f = v;
        for (int o = 8; o > 0; o >>= 1) f += __shfl_xor(f, o, 64);
        if (__float_as_uint(group_sum<16>(e)) != __float_as_uint(f)) bad |= 4u;
        e = v; f = v;
        for (int o = 16; o > 0; o >>= 1) f += __shfl_xor(f, o, 64);
        if (__float_as_uint(group_sum<32>(e)) != __float_as_uint(f)) bad |= 8u;
        e = v; f = v;
        for (int o = 4; o > 0; o >>= 1) f += __shfl_xor(f, o, 64);
        if (__float_as_uint(group_sum<8>(e)) != __float_as_uint(f)) bad |= 16u;
        e = v; f = v;
        for (int o = 16; o < 64; o <<= 1) f += __shfl_xor(f, o, 64);
        if (__float_as_uint(cross_group_sum<16>(e)) != __float_as_uint(f)) bad |= 32u;
        e = v; f = v;
        for (int o = 8; o < 64; o <<= 1) f += __shfl_xor(f, o, 64);
        if (__float_as_uint(cross_group_sum<8>(e)) != __float_as_uint(f)) bad |= 64u;
        e = v; f = v;
        for (int o = 32; o < 64; o <<= 1) f += __shfl_xor(f, o, 64);
        if (__float_as_uint(cross_group_sum<32>(e)) != __float_as_uint(f)) bad |= 128u;
    }
    {   // argmax butterfly of (value, index) pairs: quantised values so that ties occur
        float b1 = floorf(v * 4.f), b2 = b1;
        int i1 = (int)((threadIdx.x * 37u) & 63u), i2 = i1;
        if (b1 != b1) b1 = b2 = 0.f;
        wave_argmax(b1, i1);
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(b2, o, 64);
            const int oi = __shfl_xor(i2, o, 64);
            if (ov > b2 || (ov == b2 && oi < i2)) { b2 = ov; i2 = oi; }
        }
        if (__float_as_uint(b1) != __float_as_uint(b2) || i1 != i2) bad |= 256u;
    }
    if (bad) atomicOr(diff, bad), atomicAdd(diff + 1, 1u);
}
template <bool NEW>
__global__ void wave_lat_kernel(float* out, unsigned long long* cyc, int iters) {
    float v = (float)threadIdx.x * 1e-3f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) v = (NEW ? wave_sum(v) : wave_sum_shfl(v)) * 1e-2f + (float)(threadIdx.x & 63);
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = v;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
extern "C" int wave_probe(const float* in, unsigned* diff, int n_waves, hipStream_t st) {
    wave_probe_kernel<<<(n_waves + 3) / 4, 256, 0, st>>>(in, diff, n_waves);
    return (int)hipGetLastError();
}
extern "C" int wave_lat(int which, float* out, unsigned long long* cyc, int iters, hipStream_t st) {
    if (which) wave_lat_kernel<true><<<1, 64, 0, st>>>(out, cyc, iters);
    else wave_lat_kernel<false><<<1, 64, 0, st>>>(out, cyc, iters);
    return (int)hipGetLastError();
}
