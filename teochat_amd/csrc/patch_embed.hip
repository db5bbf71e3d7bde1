// Fused patch embedding of the tower (CLIPVisionEmbeddings.patch_embedding, conv with kernel = stride = P, no bias; used at
// modeling_image.py:602,645 through transformers' CLIPVisionEmbeddings): out[t*g*g + py*g + px, n] = sum_k W[n, k] * pixel(t, k),
// k = c*P*P + ky*P + kx.  The patch pixels are gathered STRAIGHT into the LDS operand image of the MFMA tile (LDS-staged patch
// tiles) -- there is no im2col matrix in HBM.  128 patches x 128 output channels x 64 k per workgroup tile, the LDS image, the
// fragment reads and the MFMA chain (v_mfma_f32_16x16x32_bf16, k ascending, zero padding up to ld_w) of gemm_mfma_bf16_kernel, so
// the result is bit-identical to im2col + GEMM (tests/test_kernels_gpu.py).  A 16-byte operand chunk is 8 consecutive k: 8 pixels
// of one patch row when kx <= P - 8, else the end of one row and the start of the next (P = 14: chunks never line up with rows),
// so the gather is element-wise: 2-byte loads, 32 per thread per K tile, 10 K tiles -- the tower's smallest GEMM (0.3 % of its
// FLOPs), latency-bound either way; what the fusion removes is the 2.6 MB round trip and one launch.
// Roofline: MFMA; algorithmic FLOPs 2 * T * g*g * (C*P*P) * D (0.31 GF per frame at ViT-L/14).
#include "common.h"
#include "ops.h"

namespace teo {

typedef __attribute__((ext_vector_type(8))) short pe_bf16x8;
typedef __attribute__((ext_vector_type(4))) float pe_f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int pe_u32x4;

constexpr int PE_BM = 128, PE_BN = 128, PE_BK = 64, PE_TILE = PE_BM * PE_BK * 2;     // 16 KB per operand tile

template <bool F16>
__global__ __launch_bounds__(256) void patch_embed_mfma_kernel(const bf16_t* __restrict__ px, const bf16_t* __restrict__ W,
                                                              bf16_t* __restrict__ out, int T, int C, int img, int P, int ldw, int D) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int g = img / P, NP = g * g, M = T * NP, KV = C * P * P;
    const int tiles_m = (M + PE_BM - 1) / PE_BM;
    const int tm = blockIdx.x % tiles_m, tn = blockIdx.x / tiles_m;
    const int m0 = tm * PE_BM, n0 = tn * PE_BN;
    // staging: thread owns chunk (row = id >> 3, c = id & 7), id = tid + 256 i, of both operand tiles
    const bf16_t* pbase[4];
    unsigned wg[4];
    int soff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int id = tid + 256 * i;
        const int row = id >> 3, c = id & 7;
        const int p = min(m0 + row, M - 1);
        const int t = p / NP, q = p - t * NP, py = q / g, pxx = q - py * g;
        pbase[i] = px + ((long long)t * C * img + py * P) * img + pxx * P;             // pixel (t, c = 0, py*P, px*P)
        wg[i] = (unsigned)min(n0 + row, D - 1) * (unsigned)ldw + c * 8;
        soff[i] = row * (PE_BK * 2) + ((c ^ (row & 7)) << 4);
    }
    pe_f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (pe_f32x4){0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fg = lane >> 4;
    const int nk = ldw / PE_BK;
    const int PP = P * P;
    for (int kt = 0; kt < nk; ++kt) {
        pe_u32x4 ra[4], rb[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k0 = kt * PE_BK + ((tid + 256 * i) & 7) * 8;
            unsigned short e[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = k0 + j;
                const int kc = min(k, KV - 1);
                const int ch = kc / PP, r = kc - ch * PP, ky = r / P, kx = r - ky * P;
                const unsigned short v = pbase[i][((long long)ch * img + ky) * img + kx];
                e[j] = k < KV ? v : (unsigned short)0;
            }
            ra[i] = (pe_u32x4){(unsigned)e[0] | ((unsigned)e[1] << 16), (unsigned)e[2] | ((unsigned)e[3] << 16),
                               (unsigned)e[4] | ((unsigned)e[5] << 16), (unsigned)e[6] | ((unsigned)e[7] << 16)};
            rb[i] = *reinterpret_cast<const pe_u32x4*>(W + (wg[i] + (unsigned)kt * PE_BK));
        }
        __syncthreads();                               // everybody is done reading the previous tile
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<pe_u32x4*>(smem + soff[i]) = ra[i];
            *reinterpret_cast<pe_u32x4*>(smem + PE_TILE + soff[i]) = rb[i];
        }
        __syncthreads();
        const unsigned char* sA = smem;
        const unsigned char* sB = smem + PE_TILE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            pe_bf16x8 af[4], wf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ra_ = wm * 64 + i * 16 + fr;
                af[i] = *reinterpret_cast<const pe_bf16x8*>(sA + ra_ * (PE_BK * 2) + (((ks * 4 + fg) ^ (ra_ & 7)) << 4));
                const int rw_ = wn * 64 + i * 16 + fr;
                wf[i] = *reinterpret_cast<const pe_bf16x8*>(sB + rw_ * (PE_BK * 2) + (((ks * 4 + fg) ^ (rw_ & 7)) << 4));
            }
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
                    acc[ni][mi] = mfma16<F16>(wf[ni], af[mi], acc[ni][mi]);
        }
    }
    // lane holds out[m = mw + mi*16 + fr][n = nw + ni*16 + fg*4 + r]
    const int mw = m0 + wm * 64, nw = n0 + wn * 64;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        const int m = mw + mi * 16 + fr;
        if (m >= M) continue;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int n = nw + ni * 16 + fg * 4;
            if (n + 3 < D) {
                *reinterpret_cast<uint2*>(out + (long long)m * D + n) =
                    make_uint2(pack_h2<F16>(acc[ni][mi][0], acc[ni][mi][1]), pack_h2<F16>(acc[ni][mi][2], acc[ni][mi][3]));
            } else {
                for (int r = 0; r < 4 && n + r < D; ++r) out[(long long)m * D + n + r] = f2h<F16>(acc[ni][mi][r]);
            }
        }
    }
}

bool patch_embed_ok(int C, int img, int P, int ldw, int D, int dtype, const void* px, const void* W, const void* out) {
    return (dtype == TEO_BF16 || dtype == TEO_F16) && img % P == 0 && ldw % PE_BK == 0 && ldw >= C * P * P && D % 4 == 0 &&
           (reinterpret_cast<uintptr_t>(W) & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & 7) == 0 && px != nullptr;
}

int patch_embed(const void* px, const void* W, void* out, int T, int C, int img, int P, int ldw, int D, hipStream_t st, bool f16) {
    if (T == 0) return TEO_OK;
    const int g = img / P, M = T * g * g;
    const int tiles = cdiv(M, PE_BM) * cdiv(D, PE_BN);
    if (f16) patch_embed_mfma_kernel<true><<<tiles, 256, 2 * PE_TILE, st>>>((const bf16_t*)px, (const bf16_t*)W, (bf16_t*)out, T, C, img, P, ldw, D);
    else patch_embed_mfma_kernel<false><<<tiles, 256, 2 * PE_TILE, st>>>((const bf16_t*)px, (const bf16_t*)W, (bf16_t*)out, T, C, img, P, ldw, D);
    note_kernel("patch_embed_mfma"); TEO_LAUNCH_CHECK("patch_embed_mfma");
    return TEO_OK;
}

}  // namespace teo
