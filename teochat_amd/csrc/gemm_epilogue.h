// Shared epilogue of the 16-bit MFMA GEMM tile families (gemm.hip, gemm_wide.hip, gemm_big.hip, gemm_narrow.hip): a lane holds
//     C[m = mw + mi * 16 + fr][n = nw + ni * 16 + fg * 4 + r],  r = 0..3,  in acc[ni][mi][r]
// and writes act(acc + bias) + residual (or silu(gate) * up for the SwiGLU16 layout) rounded once to the output type.
//
// Round 5: the epilogues of rounds 1-4 loaded the bias inside the row loop, behind `if (m >= M) continue;` -- sixteen dependent 8-byte
// loads per lane that hipcc could neither hoist nor batch across the branch: +3.0 us on a 20 us GEMM (tools/gelu_probe.py: the tower's
// qkv 19.6 us without a bias, 22.6 us with one).  Here every load is UNCONDITIONAL (clamped row / column, the stores alone are
// predicated), the bias is fetched once per column group -- `gemm_bias_load` may be called before the K loop so that it costs nothing
// -- and the residual of a whole row block is requested before the first value is used.  The arithmetic is unchanged: same bits.
#pragma once
#include "common.h"

namespace teo {

typedef __attribute__((ext_vector_type(4))) float ge_f32x4;

// bias of the lane's NI column groups as packed 16-bit pairs (zeros without a bias).  N % 4 == 0 (gemm_mfma_ok), so a column group that
// starts inside the matrix is whole; groups past the edge read the last whole group (their outputs are never stored).
template <int NI>
__device__ __forceinline__ void gemm_bias_load(const bf16_t* __restrict__ bias, int nw, int fg, int N, uint2 (&bv)[NI]) {
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) bv[ni] = make_uint2(0u, 0u);
    if (bias) {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) bv[ni] = *reinterpret_cast<const uint2*>(bias + min(nw + ni * 16 + fg * 4, N - 4));
    }
}

// MB: row blocks (of 16 rows) whose residual is requested together (register budget of the caller: MB * NI * 2 VGPRs)
template <int NI, int MI, int MB, bool SWIGLU, bool OUT_F32, bool F16>
__device__ __forceinline__ void gemm_epilogue(const ge_f32x4 (&acc)[NI][MI], const uint2 (&bv)[NI], bool has_bias, const bf16_t* res, void* Cv,
                                              int M, int N, int ldc, int act, int mw, int nw, int fr, int fg) {
    static_assert(MI % MB == 0, "row blocks per batch");
    if constexpr (SWIGLU) {
        // columns are (gate 16 | up 16) blocks: acc[ni] = gate, acc[ni + 1] = up of output column (nw / 2) + (ni / 2) * 16 + fg * 4 + r
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int m = mw + mi * 16 + fr;
#pragma unroll
            for (int ni = 0; ni < NI; ni += 2) {
                const int ng = nw + ni * 16 + fg * 4;
                const int oc = (nw >> 1) + (ni >> 1) * 16 + fg * 4;
                float o[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = silu(acc[ni][mi][r]) * acc[ni + 1][mi][r];
                if (m < M && ng < N) {
                    if (OUT_F32) *reinterpret_cast<float4*>(reinterpret_cast<float*>(Cv) + (long long)m * ldc + oc) = make_float4(o[0], o[1], o[2], o[3]);
                    else *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(Cv) + (long long)m * ldc + oc) = make_uint2(pack_h2<F16>(o[0], o[1]), pack_h2<F16>(o[2], o[3]));
                }
            }
        }
    } else {
#pragma unroll
        for (int m0 = 0; m0 < MI; m0 += MB) {
            uint2 rv[MB][NI];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) rv[mb][ni] = make_uint2(0u, 0u);
            if (res) {
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) {
                    const long long row = (long long)min(mw + (m0 + mb) * 16 + fr, M - 1) * ldc;
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) rv[mb][ni] = *reinterpret_cast<const uint2*>(res + row + min(nw + ni * 16 + fg * 4, N - 4));
                }
            }
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                const int mi = m0 + mb;
                const int m = mw + mi * 16 + fr;
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    const int n = nw + ni * 16 + fg * 4;
                    float o[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = acc[ni][mi][r];
                    if (has_bias) {
                        o[0] += h_lo<F16>(bv[ni].x); o[1] += h_hi<F16>(bv[ni].x);
                        o[2] += h_lo<F16>(bv[ni].y); o[3] += h_hi<F16>(bv[ni].y);
                    }
                    if (act != TEO_ACT_NONE) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) o[r] = act_apply(o[r], act);
                    }
                    if (res) {
                        o[0] += h_lo<F16>(rv[mb][ni].x); o[1] += h_hi<F16>(rv[mb][ni].x);
                        o[2] += h_lo<F16>(rv[mb][ni].y); o[3] += h_hi<F16>(rv[mb][ni].y);
                    }
                    if (m < M && n < N) {
                        if (OUT_F32) *reinterpret_cast<float4*>(reinterpret_cast<float*>(Cv) + (long long)m * ldc + n) = make_float4(o[0], o[1], o[2], o[3]);
                        else *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(Cv) + (long long)m * ldc + n) = make_uint2(pack_h2<F16>(o[0], o[1]), pack_h2<F16>(o[2], o[3]));
                    }
                }
            }
        }
    }
}

}  // namespace teo
