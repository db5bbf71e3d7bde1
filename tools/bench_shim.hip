// Bench-only shim (NOT part of the product ABI, not declared in include/teo_hip.h): times chains of the library's public
// decode-GEMV / skinny-GEMM entry points between two HIP events on the caller's stream.  Links against libteo_hip.so and
// uses nothing but include/teo_hip.h -- what any user of the C ABI could write.  Built as tools/libteo_bench.so.
#include <hip/hip_runtime.h>

#include "../include/teo_hip.h"

namespace {
template <typename F>
int timed_chain(int n, int reps, float* avg_ms_out, hipStream_t st, F launch) {
    if (n <= 0 || reps <= 0 || !avg_ms_out) return TEO_ERR_ARG;
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess) return TEO_ERR_HIP;
    if (hipEventCreate(&e1) != hipSuccess) { (void)hipEventDestroy(e0); return TEO_ERR_HIP; }
    int rc = TEO_OK;
    for (int i = 0; i < n && rc == TEO_OK; ++i) rc = launch(i);          // warm-up pass (not timed)
    (void)hipEventRecord(e0, st);
    for (int r = 0; r < reps && rc == TEO_OK; ++r)
        for (int i = 0; i < n && rc == TEO_OK; ++i) rc = launch(i);
    (void)hipEventRecord(e1, st);
    hipError_t e = hipEventSynchronize(e1);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (rc != TEO_OK) return rc;
    if (e != hipSuccess) return TEO_ERR_HIP;
    *avg_ms_out = ms / (float)(n * reps);
    return TEO_OK;
}
}  // namespace

extern "C" {

// The decode GEMV (teo_gemv / teo_gemv_w8) over n weight matrices back to back; average milliseconds per launch.
// d_scales != NULL: the matrices are fp8-e4m3 with per-row scales.
int teo_bench_gemv_chain(const void* d_x, const void* const* d_Ws, const float* const* d_scales, int n, const void* d_norm_w, void* d_y,
                         int N, int K, float eps, unsigned flags, int dtype, int reps, float* avg_ms_out, teo_stream_t stream) {
    return timed_chain(n, reps, avg_ms_out, (hipStream_t)stream, [&](int i) {
        return d_scales ? teo_gemv_w8(d_x, d_Ws[i], d_scales[i], d_norm_w, nullptr, d_y, N, K, eps, flags, dtype, stream)
                        : teo_gemv(d_x, d_Ws[i], d_norm_w, nullptr, d_y, N, K, eps, flags, dtype, dtype, stream);
    });
}

// The same for teo_gemm_skinny with MB activation rows (x [MB, K] bf16, y [MB, N or N/2] bf16).
int teo_bench_skinny_chain(const void* d_x, const void* const* d_Ws, const float* const* d_scales, int n, const void* d_norm_w, void* d_y,
                           int MB, int N, int K, unsigned flags, int reps, float* avg_ms_out, teo_stream_t stream) {
    const int ldo = (flags & (TEO_GEMM_SWIGLU16 | TEO_GEMM_SWIGLU8)) ? N / 2 : N;
    return timed_chain(n, reps, avg_ms_out, (hipStream_t)stream, [&](int i) {
        return teo_gemm_skinny(d_x, d_Ws[i], d_scales ? d_scales[i] : nullptr, d_scales != nullptr, d_norm_w, 1e-5f, nullptr, d_y, MB, N, K, K,
                               ldo, flags, TEO_BF16, stream);
    });
}

}  // extern "C"
