// Batched decode attention probe (NOT product code): compiles teochat_amd/csrc/attention.hip into this object and launches the
// whole-context kernel in forms the library does not ship -- 8 / 16 waves per workgroup, 32 / 64-key chunks, and PROBE = true
// (the same loads without the arithmetic: the memory-system ceiling of the access pattern) -- next to the shipped split + combine
// pair, over rotating K/V caches.  Built as tools/libattn_probe.so by tools/attn_probe.py.
#include <stdarg.h>

#include "../teochat_amd/csrc/attention.hip"

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
int attention_flash32(const teo_attn_args&, hipStream_t, bool) { return TEO_ERR_UNSUPPORTED; }
}  // namespace teo

using namespace teo;

// variant: 0 = split + combine (chunk), 1 = whole (chunk, waves), 2 = whole PROBE (chunk, waves)
extern "C" int attn_probe_launch(int variant, int chunk, int waves, const void* q, void* kc, void* vc, void* vtc, const float* cs, const float* sn,
                                 void* o, float* part, const int* d_pos, int S_max, int heads, int batch, long long q_stride,
                                 long long cache_stride, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    AttnBatch bt;
    bt.batch = batch; bt.q_stride = q_stride; bt.cache_stride = cache_stride; bt.o_stride = (long long)heads * 128;
    const float scale = 0.08838834764831845f;
    if (variant == 0) {
        g_probe_tune.attn_whole = 0;
        g_probe_tune.attn_chunk = chunk;
        return attn_decode(q, kc, vc, vtc, cs, sn, o, part, d_pos, S_max, heads, heads, 128, scale, TEO_BF16, st, bt);
    }
    const int nsw = (S_max + chunk - 1) / chunk;
    const size_t lds = ((size_t)waves * chunk + (size_t)nsw * 130 + 784) * sizeof(float);
    dim3 grid(heads, batch);
#define L_(CH, NW, PB)                                                                                                           \
    {                                                                                                                            \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_decode_whole_kernel<bf16_t, 16, CH, true, NW, PB>),        \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);                                        \
        attn_decode_whole_kernel<bf16_t, 16, CH, true, NW, PB><<<grid, NW * 64, lds, st>>>((const bf16_t*)q, (bf16_t*)kc, (bf16_t*)vc, (bf16_t*)vtc, cs, sn, \
                                                                                      (bf16_t*)o, d_pos, S_max, heads, heads, scale, bt); \
    }
#define L2_(CH, NW) { if (variant == 2) L_(CH, NW, true) else L_(CH, NW, false) }
    if (chunk == 64 && waves == 8) L2_(64, 8)
    else if (chunk == 64 && waves == 16) L2_(64, 16)
    else if (chunk == 32 && waves == 8) L2_(32, 8)
    else if (chunk == 32 && waves == 16) L2_(32, 16)
    else if (chunk == 128 && waves == 8) L2_(128, 8)
    else return -2;
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
