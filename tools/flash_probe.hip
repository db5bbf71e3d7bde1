// Timeline probe of the prefill flash-attention kernel (NOT product code): compiles teochat_amd/csrc/flash.hip into this object
// with the kernel's TRACE template flag on, so the waves of workgroup 0 (the heaviest causal query block) write 100 MHz wall-clock
// marks at every phase of every KV-tile iteration.  Built as tools/libflash_probe.so by tools/flash_probe.py; the product library only
// instantiates TRACE = false.
#include <stdarg.h>
#include <stdio.h>

#include "../teochat_amd/csrc/flash.hip"

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
}  // namespace teo

using namespace teo;

extern "C" int flash_probe_trace_iters() { return FA_TRACE_ITERS; }

// pipe: 1 = the in-wave software pipeline, 0 = one tile at a time (both stage by LDS-DMA)
extern "C" int flash_probe_launch(const teo_attn_args* a, int pair_c, unsigned long long* trace, void* stream, int pipe) {
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(((a->q_len + 127) / 128) * a->heads * a->batch);
    const size_t lds = 2 * (size_t)(64 * a->head_dim * 2 + a->head_dim * 128);
    if (a->head_dim == 64 && !a->causal) {              // the tower's shape: one tile at a time, four tile pairs in LDS
        const size_t lds4 = 4 * (size_t)(64 * 64 * 2 + 64 * 128);
        if (trace) attn_flash32_kernel<64, false, false, true, false, 4><<<grid, 256, lds4, st>>>(*a, pair_c, trace);
        else       attn_flash32_kernel<64, false, false, false, false, 4><<<grid, 256, lds4, st>>>(*a, pair_c, nullptr);
        return hipGetLastError() == hipSuccess ? 0 : -3;
    }
    if (a->head_dim != 128 || !a->causal) return -2;
    if (pipe) {
        if (trace) attn_flash32_kernel<128, true, false, true, true><<<grid, 256, lds, st>>>(*a, pair_c, trace);
        else       attn_flash32_kernel<128, true, false, false, true><<<grid, 256, lds, st>>>(*a, pair_c, nullptr);
    } else {
        if (trace) attn_flash32_kernel<128, true, false, true, false><<<grid, 256, lds, st>>>(*a, pair_c, trace);
        else       attn_flash32_kernel<128, true, false, false, false><<<grid, 256, lds, st>>>(*a, pair_c, nullptr);
    }
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
