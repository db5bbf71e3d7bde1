// Timeline probe of the batched-decode GEMM kernels (NOT product code): compiles teochat_amd/csrc/skinny.hip into this
// object with the kernels' TRACE template flag on, so every workgroup writes 100 MHz wall-clock marks at its milestones
// (entry, first weights consumed, every tile's barrier, epilogue stores).  Built as tools/libskinny_probe.so by
// tools/skinny_probe.py; the product library only instantiates TRACE = false.
#include <stdarg.h>

#include "../teochat_amd/csrc/skinny.hip"

namespace teo {
void set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
int hip_fail(hipError_t e, const char* what) { fprintf(stderr, "%s: %s\n", what, hipGetErrorString(e)); return TEO_ERR_HIP; }
void note_kernel(const char*) {}
int device_cu_count() { return 256; }
static teo_tune g_probe_tune;                       // the probe's own knob block (the library keeps these in teo_tune blocks: tune.h)
const teo_tune& tune() { return g_probe_tune; }
int lds_attr_once(const void* kernel, int bytes, unsigned long long*, const char*) {
    return hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess ? TEO_OK : TEO_ERR_HIP;
}
bool prof_take(hipEvent_t*, hipEvent_t*) { return false; }
void prof_class(int) {}
void prof_bump(int) {}
}  // namespace teo

using namespace teo;

// mode 0: tile kernel (one 16-row tile per workgroup), 1: streaming kernel (persistent, one workgroup per CU)
extern "C" int skinny_probe_launch(int mode, const void* x, const void* W, const float* wscale, int w_fp8, const void* res, void* out, int MB,
                                   int N, int K, int sw8, const float* ssq_in, int nparts, const void* next_g, void* xg_out, float* ssq_out,
                                   unsigned long long* trace, int grid_cap, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    SkinnyFuse fuse;
    fuse.ssq_in = ssq_in; fuse.nparts = nparts; fuse.eps = 1e-5f; fuse.trace = trace;
    fuse.next_g = (const unsigned short*)next_g; fuse.xg_out = (unsigned short*)xg_out; fuse.ssq_out = ssq_out;
    const int ldo = sw8 ? N / 2 : N;
    const int ntiles = (N + 15) / 16;
    if (mode == 1) {
        const int grid = ntiles < grid_cap ? ntiles : grid_cap;
        if (w_fp8) {
            if (sw8) skinny_stream_kernel<fp8_t, 8, 1, 3, true, true><<<grid, SK_THREADS, 0, st>>>((const bf16_t*)x, (const fp8_t*)W, wscale, (const bf16_t*)res, out, MB, N, K, K, ldo, 1, 0, fuse);
            else     skinny_stream_kernel<fp8_t, 8, 1, 3, false, true><<<grid, SK_THREADS, 0, st>>>((const bf16_t*)x, (const fp8_t*)W, wscale, (const bf16_t*)res, out, MB, N, K, K, ldo, 1, 0, fuse);
        } else {
            if (sw8) skinny_stream_kernel<bf16_t, 8, 2, 4, true, true><<<grid, SK_THREADS, 0, st>>>((const bf16_t*)x, (const bf16_t*)W, wscale, (const bf16_t*)res, out, MB, N, K, K, ldo, 1, 0, fuse);
            else     skinny_stream_kernel<bf16_t, 8, 2, 4, false, true><<<grid, SK_THREADS, 0, st>>>((const bf16_t*)x, (const bf16_t*)W, wscale, (const bf16_t*)res, out, MB, N, K, K, ldo, 1, 0, fuse);
        }
    } else {
        const int blocks = ntiles;
        if (w_fp8) {
            if (sw8) skinny_gemm_kernel<fp8_t, 4, true, true, false, true><<<blocks, SK_THREADS, 0, st>>>((const fp8_t*)W, (const bf16_t*)x, MB, N, K, K, 1, 1, wscale, (const bf16_t*)res, nullptr, 1e-5f, out, ldo, ldo, 0, fuse, 1);
            else     skinny_gemm_kernel<fp8_t, 4, true, false, false, true><<<blocks, SK_THREADS, 0, st>>>((const fp8_t*)W, (const bf16_t*)x, MB, N, K, K, 1, 1, wscale, (const bf16_t*)res, nullptr, 1e-5f, out, ldo, ldo, 0, fuse, 0);
        } else {
            if (sw8) skinny_gemm_kernel<bf16_t, 4, true, true, false, true><<<blocks, SK_THREADS, 0, st>>>((const bf16_t*)W, (const bf16_t*)x, MB, N, K, K, 1, 1, wscale, (const bf16_t*)res, nullptr, 1e-5f, out, ldo, ldo, 0, fuse, 1);
            else     skinny_gemm_kernel<bf16_t, 4, true, false, false, true><<<blocks, SK_THREADS, 0, st>>>((const bf16_t*)W, (const bf16_t*)x, MB, N, K, K, 1, 1, wscale, (const bf16_t*)res, nullptr, 1e-5f, out, ldo, ldo, 0, fuse, 0);
        }
    }
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
