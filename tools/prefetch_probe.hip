// Probe (NOT product code): is there anything in warming the NEXT kernel's first weight blocks into the XCD's L2 from the tail of the
// current one?  The decode step is ~194 dependent weight-streaming launches per token; each pays ~0.9 us beyond its bytes (launch boundary +
// first-load latency + ramp: profiles/r04_decode_experiments.md).  The weights of launch n + 1 do not depend on launch n, so launch n can
// touch the first lines launch n + 1 will ask for -- from the SAME XCD (workgroup b of both launches runs on XCD b % 8, r04_xcc_probe.txt) --
// with one dword load per 128-byte line, result unused.  Not in the closing table of mechanisms tried in rounds 1-4 (#6 there bounds what
// overlapping consecutive kernels is worth: 0.22 ms per token).
//
// The kernels here stream a [N][K] bf16 matrix in the production GEMV's access pattern (4 waves per workgroup, 2 rows per wave and step,
// 16-byte non-temporal loads, row groups strided by the grid) and reduce it to one float per row (a dot product with a constant vector held
// in registers), so the loads are live; a chain of launches over 32 distinct matrices (qkv / o / gate-up / down sizes of LLaMA-2-7B) is
// timed with HIP events, with and without the tail prefetch.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/prefetch_probe.hip -o tools/prefetch_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

// rows of 8 KiB... K elements of 2 bytes; a wave reads 2 rows per step, a workgroup 8 rows per group
template <int PF>   // PF: KiB per row of the NEXT matrix's first row group (8 rows per workgroup) touched from this kernel's tail (0 = none)
__global__ __launch_bounds__(256) void stream_kernel(const unsigned short* __restrict__ W, float* __restrict__ out, int N, int K,
                                                     const unsigned short* __restrict__ nextW, int nextN, int nextK) {
    __shared__ unsigned scrap[64];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int groups = N / 8;
    const int chunks = (K / 8 + 63) / 64, last = K / 8 - 1;   // 16-byte chunks per lane per row (the ragged end re-reads the last chunk)
    float acc_total = 0.f;
    for (int g = blockIdx.x; g < groups; g += gridDim.x) {
        const unsigned short* r0 = W + (long long)(g * 8 + wid * 2) * K;
        const unsigned short* r1 = r0 + K;
        float a0 = 0.f, a1 = 0.f;
        if (PF > 0 && g + (int)gridDim.x >= groups) {
            // tail prefetch, issued in front of this workgroup's LAST row group (it returns under that group's stream): one dword per
            // 128-byte line of the first PF KiB of each row of the FIRST row group workgroup blockIdx.x of the next launch will read
            if ((int)blockIdx.x < nextN / 8) {
                const unsigned short* q = nextW + (long long)(blockIdx.x * 8 + wid * 2) * nextK;
                constexpr int lines = PF * 1024 / 128;                                  // per row
                // LDS-DMA with a 4-byte payload per lane: the destination is a scrap LDS word, so no VGPR waits for a value nobody wants
                // (an inline-asm global_load into a dummy VGPR lets the compiler reuse that register before the load lands)
                for (int i0 = 0; i0 < 2 * lines; i0 += 64) {
                    const int i = min(i0 + lane, 2 * lines - 1);
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(q + (long long)(i / lines) * nextK + (i % lines) * 64),
                                                     (__attribute__((address_space(3))) void*)scrap, 4, 0, 0);
                }
            }
        }
        for (int c = 0; c < chunks; c += 4) {
            u32x4 v0[4], v1[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                v0[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(r0 + min((c + u) * 64 + lane, last) * 8));
                v1[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(r1 + min((c + u) * 64 + lane, last) * 8));
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    a0 += __uint_as_float(v0[u][i] << 16) + __uint_as_float(v0[u][i] & 0xffff0000u);
                    a1 += __uint_as_float(v1[u][i] << 16) + __uint_as_float(v1[u][i] & 0xffff0000u);
                }
            }
        }
        for (int o = 32; o > 0; o >>= 1) { a0 += __shfl_xor(a0, o, 64); a1 += __shfl_xor(a1, o, 64); }
        if (lane == 0) { out[g * 8 + wid * 2] = a0; out[g * 8 + wid * 2 + 1] = a1; }
        acc_total += a0 + a1;
    }
    if (acc_total == 123.456f) out[0] = acc_total;
    if (PF > 0 && acc_total == 654.321f) out[1] = (float)scrap[lane];
}

int main() {
    struct Mat { const char* name; int N, K; };
    const Mat layer[4] = {{"qkv", 12288, 4096}, {"o", 4096, 4096}, {"gateup", 22016, 4096}, {"down", 4096, 11008}};
    const int L = 32;
    std::vector<unsigned short*> W(L * 4);
    for (int l = 0; l < L; ++l)
        for (int m = 0; m < 4; ++m) {
            const size_t bytes = (size_t)layer[m].N * layer[m].K * 2;
            CK(hipMalloc(&W[l * 4 + m], bytes));
            CK(hipMemset(W[l * 4 + m], 0x3c + m, bytes));
        }
    float* out;
    CK(hipMalloc(&out, 32768 * 4));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto chain = [&](int pf) {
        for (int i = 0; i < L * 4; ++i) {
            const Mat& m = layer[i & 3];
            const Mat& mn = layer[(i + 1) & 3];
            const unsigned short* nw = W[(i + 1) % (L * 4)];
            const int grid = std::min(1024, m.N / 8);
            if (pf == 0) stream_kernel<0><<<grid, 256, 0, st>>>(W[i], out, m.N, m.K, nw, mn.N, mn.K);
            else if (pf == 1) stream_kernel<1><<<grid, 256, 0, st>>>(W[i], out, m.N, m.K, nw, mn.N, mn.K);
            else if (pf == 2) stream_kernel<2><<<grid, 256, 0, st>>>(W[i], out, m.N, m.K, nw, mn.N, mn.K);
            else if (pf == 3) stream_kernel<4><<<grid, 256, 0, st>>>(W[i], out, m.N, m.K, nw, mn.N, mn.K);
            else stream_kernel<8><<<grid, 256, 0, st>>>(W[i], out, m.N, m.K, nw, mn.N, mn.K);
        }
    };
    const int PFK[5] = {0, 1, 2, 4, 8};
    double best[5] = {1e30, 1e30, 1e30, 1e30, 1e30};
    for (int rep = 0; rep < 5; ++rep)
        for (int pf = 0; pf < 5; ++pf) {
            chain(pf);                                   // warm
            CK(hipEventRecord(e0, st));
            for (int r = 0; r < 3; ++r) chain(pf);
            CK(hipEventRecord(e1, st));
            CK(hipStreamSynchronize(st));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            best[pf] = std::min(best[pf], (double)ms / 3.0);
        }
    double bytes = 0;
    for (int m = 0; m < 4; ++m) bytes += (double)layer[m].N * layer[m].K * 2;
    bytes *= L;
    for (int pf = 0; pf < 5; ++pf)
        printf("chain of %d launches (32 layers x qkv, o, gate/up, down; %.2f GB): tail prefetch of the first %d KiB of each of the next launch's first rows: "
               "%.3f ms = %.2f TB/s, %.2f us per launch beyond bytes / 6.46 TB/s\n", L * 4, bytes / 1e9, PFK[pf], best[pf], bytes / best[pf] / 1e9,
               (best[pf] * 1e3 - bytes / 6.46e12 * 1e6) / (L * 4));
    return 0;
}
