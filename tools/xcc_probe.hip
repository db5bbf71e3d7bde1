// Which XCD does workgroup i of a 1-D launch run on?  (NOT product code.)  s_getreg_b32 HW_REG_XCC_ID (id 20, bits 3:0) per workgroup.
#include <hip/hip_runtime.h>
__global__ void xcc_kernel(int* out, int spin) {
    if (threadIdx.x == 0) out[blockIdx.x] = (int)(__builtin_amdgcn_s_getreg(20 | (0 << 6) | ((4 - 1) << 11)) & 15);
    for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(8);
}
extern "C" int xcc_probe(int* d_out, int blocks, int threads, int lds, int spin, void* stream) {
    hipLaunchKernelGGL(xcc_kernel, dim3(blocks), dim3(threads), lds, (hipStream_t)stream, d_out, spin);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
