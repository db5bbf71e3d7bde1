// Data-movement and small kernels: RoPE + KV append, V transposes, im2col, embedding splice, CLS drop, argmax,
// decode bookkeeping.  All HBM/latency-bound; none is shaped into a GEMM.
#include "common.h"

namespace teo {

// ------------------------------------------------------------------------------------------------
// RoPE (rotate-half convention, tf llama apply_rotary_pos_emb) + KV append
//   grid (S, heads + 2*kv_heads); block = hd/2 threads (>= 1 wave)
//   d_past (optional) overrides `past` with a device-resident position (decode under hipGraph replay)
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void rope_kv_append_kernel(T* __restrict__ qkv, int ld, const int* __restrict__ positions,
                                      const float* __restrict__ cs, const float* __restrict__ sn, T* __restrict__ kc,
                                      T* __restrict__ vc, T* __restrict__ vtc, int past, const int* __restrict__ d_past,
                                      int S_max, int heads, int kv_heads, int hd, int write_vt) {
    const int s = blockIdx.x, hh = blockIdx.y, half = hd >> 1;
    const int base_pos = d_past ? *d_past : past;
    const int pos = positions ? positions[s] : base_pos + s;   // rotary position of this token
    const int slot = base_pos + s;                             // cache slot
    T* row = qkv + (long long)s * ld;
    for (int i = threadIdx.x; i < half; i += blockDim.x) {
        if (hh < heads + kv_heads) {
            T* x = row + hh * hd;
            const float c = cs[(long long)pos * half + i], sv = sn[(long long)pos * half + i];
            const float x1 = Elem<T>::ld(x + i), x2 = Elem<T>::ld(x + i + half);
            const float y1 = x1 * c - x2 * sv, y2 = x2 * c + x1 * sv;
            if (hh < heads) {
                Elem<T>::st(x + i, y1);
                Elem<T>::st(x + i + half, y2);
            } else {
                T* dst = kc + ((long long)(hh - heads) * S_max + slot) * hd;
                Elem<T>::st(dst + i, y1);
                Elem<T>::st(dst + i + half, y2);
            }
        } else {
            const int hk = hh - heads - kv_heads;
            const T* x = row + (heads + kv_heads + hk) * hd;
            T* dst = vc + ((long long)hk * S_max + slot) * hd;
            const T a = x[i], b = x[i + half];
            dst[i] = a;
            dst[i + half] = b;
            if (write_vt) {
                vtc[((long long)hk * hd + i) * S_max + slot] = a;
                vtc[((long long)hk * hd + i + half) * S_max + slot] = b;
            }
        }
    }
}

// bf16, head_dim % 16 == 0: one thread rotates 8 pairs (16-byte loads of both halves, 32-byte cos / sin rows) or copies
// 2 x 16 bytes of V.  grid (S, ceil((heads + 2 kv_heads) * hd/16 / 256)); the V^T scatter is left to vt_append_kernel.
template <bool F16>
__device__ __forceinline__ void rope_kv_append_vec_body(bf16_t* __restrict__ qkv, int ld, const int* __restrict__ positions,
                                                        const float* __restrict__ cs, const float* __restrict__ sn,
                                                        bf16_t* __restrict__ kc, bf16_t* __restrict__ vc, int past,
                                                        int S_max, int heads, int kv_heads, int hd, int copy_v, int bx, int by) {
    // copy_v == 0 (round 5): the V rows are appended by vt_append_vec_kernel, which has them in hand for the V^T scatter anyway -- V is
    // read once per layer instead of twice
    const int s = bx, half = hd >> 1, tph = half >> 3;
    const int t = by * 256 + threadIdx.x;
    const int hh = t / tph, i0 = (t % tph) * 8;
    if (hh >= heads + (copy_v ? 2 : 1) * kv_heads) return;
    const int pos = positions ? positions[s] : past + s;
    const int slot = past + s;
    bf16_t* x = qkv + (long long)s * ld + hh * hd;
    const uint4 lo = *reinterpret_cast<const uint4*>(x + i0), hi = *reinterpret_cast<const uint4*>(x + i0 + half);
    if (hh >= heads + kv_heads) {
        bf16_t* dst = vc + ((long long)(hh - heads - kv_heads) * S_max + slot) * hd;
        *reinterpret_cast<uint4*>(dst + i0) = lo;
        *reinterpret_cast<uint4*>(dst + i0 + half) = hi;
        return;
    }
    const float4 c0 = *reinterpret_cast<const float4*>(cs + (long long)pos * half + i0);
    const float4 c1 = *reinterpret_cast<const float4*>(cs + (long long)pos * half + i0 + 4);
    const float4 s0 = *reinterpret_cast<const float4*>(sn + (long long)pos * half + i0);
    const float4 s1 = *reinterpret_cast<const float4*>(sn + (long long)pos * half + i0 + 4);
    const float c[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
    const float sv[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
    const unsigned lw[4] = {lo.x, lo.y, lo.z, lo.w}, hw[4] = {hi.x, hi.y, hi.z, hi.w};
    unsigned o1[4], o2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float a0 = h_lo<F16>(lw[j]), a1 = h_hi<F16>(lw[j]);
        const float b0 = h_lo<F16>(hw[j]), b1 = h_hi<F16>(hw[j]);
        o1[j] = pack_h2<F16>(a0 * c[2 * j] - b0 * sv[2 * j], a1 * c[2 * j + 1] - b1 * sv[2 * j + 1]);
        o2[j] = pack_h2<F16>(b0 * c[2 * j] + a0 * sv[2 * j], b1 * c[2 * j + 1] + a1 * sv[2 * j + 1]);
    }
    bf16_t* dst = hh < heads ? x : kc + ((long long)(hh - heads) * S_max + slot) * hd;
    *reinterpret_cast<uint4*>(dst + i0) = make_uint4(o1[0], o1[1], o1[2], o1[3]);
    *reinterpret_cast<uint4*>(dst + i0 + half) = make_uint4(o2[0], o2[1], o2[2], o2[3]);
}
template <bool F16>
__global__ __launch_bounds__(256) void rope_kv_append_vec_kernel(bf16_t* __restrict__ qkv, int ld, const int* __restrict__ positions,
                                                                 const float* __restrict__ cs, const float* __restrict__ sn,
                                                                 bf16_t* __restrict__ kc, bf16_t* __restrict__ vc, int past,
                                                                 int S_max, int heads, int kv_heads, int hd, int copy_v) {
    rope_kv_append_vec_body<F16>(qkv, ld, positions, cs, sn, kc, vc, past, S_max, heads, kv_heads, hd, copy_v, blockIdx.x, blockIdx.y);
}

// tiled V -> V^T append for long prefills: block = (64 positions, one kv head); 128-byte rows out
template <typename T>
__global__ __launch_bounds__(256) void vt_append_kernel(const T* __restrict__ qkv, int ld, T* __restrict__ vtc, int S,
                                                        int past, int S_max, int v_off, int hd) {
    __shared__ T tile[64][129];
    const int s0 = blockIdx.x * 64, hk = blockIdx.y;
    for (int d0 = 0; d0 < hd; d0 += 128) {
        const int dw = min(128, hd - d0);
        for (int id = threadIdx.x; id < 64 * dw; id += 256) {
            const int r = id / dw, d = id % dw;
            tile[r][d] = (s0 + r < S) ? qkv[(long long)(s0 + r) * ld + v_off + hk * hd + d0 + d] : (T)0;
        }
        __syncthreads();
        for (int id = threadIdx.x; id < 64 * dw; id += 256) {
            const int d = id / 64, r = id % 64;
            if (s0 + r < S) vtc[((long long)hk * hd + d0 + d) * S_max + past + s0 + r] = tile[r][d];
        }
        __syncthreads();
    }
}

// bf16 variant with 16-byte global accesses on both sides: block = (64 positions, one kv head, 128 dims per pass).
// Positions >= S inside the last 8-position store chunk are written as zeros (slots past the sequence end, rewritten by
// whoever appends there).  Needs (past + s0) % 8 == 0 for the aligned 16-byte stores.
__device__ __forceinline__ void vt_append_vec_body(const bf16_t* __restrict__ qkv, int ld, bf16_t* __restrict__ vtc, int S,
                                                   int past, int S_max, int v_off, int hd, bf16_t* __restrict__ vc, int bx, int by) {
    __shared__ bf16_t tile[64][136];                       // row stride 272 B: 16-byte aligned rows, column reads spread over banks
    const int s0 = bx * 64, hk = by;
    for (int d0 = 0; d0 < hd; d0 += 128) {
        const int dw = min(128, hd - d0), nc = dw >> 3;    // 16-byte chunks per row
        for (int id = threadIdx.x; id < 64 * nc; id += 256) {
            const int r = id / nc, c = id % nc;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (s0 + r < S) {
                v = *reinterpret_cast<const uint4*>(qkv + (long long)(s0 + r) * ld + v_off + hk * hd + d0 + c * 8);
                if (vc) *reinterpret_cast<uint4*>(vc + ((long long)hk * S_max + past + s0 + r) * hd + d0 + c * 8) = v;      // the V cache row (rope kernel: copy_v = 0)
            }
            *reinterpret_cast<uint4*>(&tile[r][c * 8]) = v;
        }
        __syncthreads();
        for (int id = threadIdx.x; id < dw * 8; id += 256) {
            const int d = id >> 3, pc = id & 7;            // 8 positions pc*8 .. +8 of dim d
            if (s0 + pc * 8 >= S) continue;
            unsigned w[4];
#pragma unroll
            for (int e = 0; e < 4; ++e)
                w[e] = (unsigned)tile[pc * 8 + 2 * e][d] | ((unsigned)tile[pc * 8 + 2 * e + 1][d] << 16);
            *reinterpret_cast<uint4*>(vtc + ((long long)hk * hd + d0 + d) * S_max + past + s0 + pc * 8) = make_uint4(w[0], w[1], w[2], w[3]);
        }
        __syncthreads();
    }
}
__global__ __launch_bounds__(256) void vt_append_vec_kernel(const bf16_t* __restrict__ qkv, int ld, bf16_t* __restrict__ vtc, int S,
                                                            int past, int S_max, int v_off, int hd, bf16_t* __restrict__ vc) {
    vt_append_vec_body(qkv, ld, vtc, S, past, S_max, v_off, hd, vc, blockIdx.x, blockIdx.y);
}
// ONE launch for both (round 6, late): the first n_rope workgroups rotate q / k and append K, the rest append V and scatter V^T -- the two halves
// touch disjoint columns of the qkv rows and were two dependent launches of 6-13 us each per layer
template <bool F16>
__global__ __launch_bounds__(256) void rope_kv_vt_fused_kernel(bf16_t* __restrict__ qkv, int ld, const int* __restrict__ positions,
                                                               const float* __restrict__ cs, const float* __restrict__ sn,
                                                               bf16_t* __restrict__ kc, bf16_t* __restrict__ vc, bf16_t* __restrict__ vtc, int S, int past,
                                                               int S_max, int heads, int kv_heads, int hd, int rope_by, int n_rope, int vt_bx) {
    const int b = blockIdx.x;
    if (b < n_rope) rope_kv_append_vec_body<F16>(qkv, ld, positions, cs, sn, kc, vc, past, S_max, heads, kv_heads, hd, 0, b / rope_by, b % rope_by);
    else vt_append_vec_body(qkv, ld, vtc, S, past, S_max, (heads + kv_heads) * hd, hd, vc, (b - n_rope) % vt_bx, (b - n_rope) / vt_bx);
}

int rope_kv_append(void* qkv, int ld, const int* positions, const float* cs, const float* sn, void* kc, void* vc,
                   void* vtc, int S, int past, const int* d_past, int S_max, int heads, int kv_heads, int hd, int dtype,
                   hipStream_t st) {
    if (S == 0) return TEO_OK;
    TEO_CHECK_ARG(hd % 2 == 0, "rope: odd head_dim %d", hd);
    const int threads = ((hd / 2 + 63) / 64) * 64;
    dim3 grid(S, heads + 2 * kv_heads);
    const bool tiled_vt = vtc && S >= 16 && d_past == nullptr;
    const int wvt = (vtc && !tiled_vt) ? 1 : 0;
    if (dtype == TEO_F32) {
        rope_kv_append_kernel<float><<<grid, threads, 0, st>>>((float*)qkv, ld, positions, cs, sn, (float*)kc, (float*)vc,
                                                               (float*)vtc, past, d_past, S_max, heads, kv_heads, hd, wvt);
        if (tiled_vt)
            vt_append_kernel<float><<<dim3(cdiv(S, 64), kv_heads), 256, 0, st>>>((const float*)qkv, ld, (float*)vtc, S, past,
                                                                                 S_max, (heads + kv_heads) * hd, hd);
    } else {
        const bool vec = hd % 16 == 0 && ld % 8 == 0 && d_past == nullptr && wvt == 0 &&
                         ((reinterpret_cast<uintptr_t>(qkv) | reinterpret_cast<uintptr_t>(kc) | reinterpret_cast<uintptr_t>(vc) |
                           reinterpret_cast<uintptr_t>(cs) | reinterpret_cast<uintptr_t>(sn)) & 15) == 0;
        // the 16-byte V^T kernel also appends the V rows (it reads them for the transpose anyway): the rope kernel then leaves V alone
        const bool vvec = tiled_vt && hd % 8 == 0 && ld % 8 == 0 && past % 8 == 0 && S_max % 8 == 0 &&
                          ((reinterpret_cast<uintptr_t>(qkv) | reinterpret_cast<uintptr_t>(vtc)) & 15) == 0;
        const bool v_by_vt = vec && vvec && (reinterpret_cast<uintptr_t>(vc) & 15) == 0;
        if (vec && v_by_vt && tune().rope_vt_fused != 0) {
            const int nthr = (heads + kv_heads) * (hd / 16), rope_by = cdiv(nthr, 256), n_rope = S * rope_by, vt_bx = cdiv(S, 64);
            const int grid = n_rope + vt_bx * kv_heads;
            if (dtype == TEO_F16)
                rope_kv_vt_fused_kernel<true><<<grid, 256, 0, st>>>((bf16_t*)qkv, ld, positions, cs, sn, (bf16_t*)kc, (bf16_t*)vc, (bf16_t*)vtc, S, past, S_max,
                                                                    heads, kv_heads, hd, rope_by, n_rope, vt_bx);
            else
                rope_kv_vt_fused_kernel<false><<<grid, 256, 0, st>>>((bf16_t*)qkv, ld, positions, cs, sn, (bf16_t*)kc, (bf16_t*)vc, (bf16_t*)vtc, S, past, S_max,
                                                                     heads, kv_heads, hd, rope_by, n_rope, vt_bx);
            TEO_LAUNCH_CHECK("rope_kv_append");
            return TEO_OK;
        }
        if (vec) {
            const int nthr = (heads + (v_by_vt ? 1 : 2) * kv_heads) * (hd / 16);
            if (dtype == TEO_F16)
                rope_kv_append_vec_kernel<true><<<dim3(S, cdiv(nthr, 256)), 256, 0, st>>>((bf16_t*)qkv, ld, positions, cs, sn, (bf16_t*)kc,
                                                                                          (bf16_t*)vc, past, S_max, heads, kv_heads, hd, v_by_vt ? 0 : 1);
            else
                rope_kv_append_vec_kernel<false><<<dim3(S, cdiv(nthr, 256)), 256, 0, st>>>((bf16_t*)qkv, ld, positions, cs, sn, (bf16_t*)kc,
                                                                                           (bf16_t*)vc, past, S_max, heads, kv_heads, hd, v_by_vt ? 0 : 1);
        } else if (dtype == TEO_F16) {
            rope_kv_append_kernel<f16_t><<<grid, threads, 0, st>>>((f16_t*)qkv, ld, positions, cs, sn, (f16_t*)kc,
                                                                   (f16_t*)vc, (f16_t*)vtc, past, d_past, S_max, heads,
                                                                   kv_heads, hd, wvt);
        } else {
            rope_kv_append_kernel<bf16_t><<<grid, threads, 0, st>>>((bf16_t*)qkv, ld, positions, cs, sn, (bf16_t*)kc,
                                                                    (bf16_t*)vc, (bf16_t*)vtc, past, d_past, S_max, heads,
                                                                    kv_heads, hd, wvt);
        }
        if (tiled_vt) {
            if (vvec)
                vt_append_vec_kernel<<<dim3(cdiv(S, 64), kv_heads), 256, 0, st>>>((const bf16_t*)qkv, ld, (bf16_t*)vtc, S, past, S_max,
                                                                                  (heads + kv_heads) * hd, hd, v_by_vt ? (bf16_t*)vc : nullptr);
            else
                vt_append_kernel<bf16_t><<<dim3(cdiv(S, 64), kv_heads), 256, 0, st>>>((const bf16_t*)qkv, ld, (bf16_t*)vtc, S,
                                                                                      past, S_max, (heads + kv_heads) * hd, hd);
        }
    }
    TEO_LAUNCH_CHECK("rope_kv_append");
    return TEO_OK;
}

// ------------------------------------------------------------------------------------------------
// ViT: V^T[t][h][d][j] = qkv[t*N + j][2*D + h*hd + d]; columns j in [N, ldv) are zero
//   grid (ceil(ldv/64), heads, T)
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void vit_vt_kernel(const T* __restrict__ qkv, T* __restrict__ vt, int N, int heads,
                                                     int hd, int ldv) {
    __shared__ T tile[64][65];
    const int j0 = blockIdx.x * 64, h = blockIdx.y, t = blockIdx.z;
    const int D = heads * hd;
    for (int d0 = 0; d0 < hd; d0 += 64) {
        const int dw = min(64, hd - d0);
        for (int id = threadIdx.x; id < 64 * dw; id += 256) {
            const int r = id / dw, d = id % dw;
            tile[r][d] = (j0 + r < N) ? qkv[((long long)t * N + j0 + r) * (3 * D) + 2 * D + h * hd + d0 + d] : (T)0;
        }
        __syncthreads();
        for (int id = threadIdx.x; id < 64 * dw; id += 256) {
            const int d = id / 64, r = id % 64;
            if (j0 + r < ldv) vt[(((long long)t * heads + h) * hd + d0 + d) * ldv + j0 + r] = tile[r][d];
        }
        __syncthreads();
    }
}

// 16-bit variant with 16-byte global accesses on both sides (the shape of vt_append_vec_kernel): block = (64 tokens, one head, one frame),
// hd <= 128; tokens j >= N are written as zeros.  9.5 -> ~4 us per tower layer at T = 8.
__global__ __launch_bounds__(256) void vit_vt_vec_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ vt, int N, int heads, int hd,
                                                         int ldv) {
    __shared__ bf16_t tile[64][136];                       // row stride 272 B: 16-byte aligned rows, column reads spread over banks
    const int j0 = blockIdx.x * 64, h = blockIdx.y, t = blockIdx.z;
    const int D = heads * hd, nc = hd >> 3;                // 16-byte chunks per row
    const bf16_t* src = qkv + ((long long)t * N + j0) * (3 * D) + 2 * D + h * hd;
    for (int id = threadIdx.x; id < 64 * nc; id += 256) {
        const int r = id / nc, c = id % nc;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (j0 + r < N) v = *reinterpret_cast<const uint4*>(src + (long long)r * (3 * D) + c * 8);
        *reinterpret_cast<uint4*>(&tile[r][c * 8]) = v;
    }
    __syncthreads();
    bf16_t* dst = vt + (((long long)t * heads + h) * hd) * ldv + j0;
    for (int id = threadIdx.x; id < hd * 8; id += 256) {
        const int d = id >> 3, pc = id & 7;                // 8 tokens pc*8 .. +8 of dim d
        if (j0 + pc * 8 >= ldv) continue;
        unsigned w[4];
#pragma unroll
        for (int e = 0; e < 4; ++e)
            w[e] = (unsigned)tile[pc * 8 + 2 * e][d] | ((unsigned)tile[pc * 8 + 2 * e + 1][d] << 16);
        *reinterpret_cast<uint4*>(dst + (long long)d * ldv + pc * 8) = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

int vit_value_transpose(const void* qkv, void* vt, int T_, int N, int heads, int hd, int ldv, int dtype, hipStream_t st) {
    if (T_ == 0) return TEO_OK;
    dim3 grid(cdiv(ldv, 64), heads, T_);
    if (dtype != TEO_F32 && hd % 8 == 0 && hd <= 128 && ldv % 8 == 0 && ((reinterpret_cast<uintptr_t>(qkv) | reinterpret_cast<uintptr_t>(vt)) & 15) == 0) {
        vit_vt_vec_kernel<<<grid, 256, 0, st>>>((const bf16_t*)qkv, (bf16_t*)vt, N, heads, hd, ldv);       // bf16 and fp16: a 16-bit move
        TEO_LAUNCH_CHECK("vit_value_transpose");
        return TEO_OK;
    }
    if (dtype == TEO_F32) vit_vt_kernel<float><<<grid, 256, 0, st>>>((const float*)qkv, (float*)vt, N, heads, hd, ldv);
    else vit_vt_kernel<bf16_t><<<grid, 256, 0, st>>>((const bf16_t*)qkv, (bf16_t*)vt, N, heads, hd, ldv);
    TEO_LAUNCH_CHECK("vit_value_transpose");
    return TEO_OK;
}

// ------------------------------------------------------------------------------------------------
// im2col for the patch-embedding conv (kernel = stride = P): one thread per output element
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void im2col_kernel(const T* __restrict__ px, T* __restrict__ cols, int C, int img, int P, int ld,
                              long long total) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int col = (int)(idx % ld);
    const long long row = idx / ld;
    const int g = img / P;
    const int K = C * P * P;
    T v = (T)0;
    if (col < K) {
        const int c = col / (P * P), ky = (col / P) % P, kx = col % P;
        const int t = (int)(row / (g * g)), py = (int)((row / g) % g), pxx = (int)(row % g);
        v = px[(((long long)t * C + c) * img + py * P + ky) * img + pxx * P + kx];
    }
    cols[idx] = v;
}

int im2col_patches(const void* px, void* cols, int T_, int C, int img, int P, int ld, int dtype, hipStream_t st) {
    const int g = img / P;
    const long long total = (long long)T_ * g * g * ld;
    if (total == 0) return TEO_OK;
    TEO_CHECK_ARG(img % P == 0 && ld >= C * P * P, "im2col: image %d patch %d ld %d", img, P, ld);
    const int blocks = cdiv(total, 256);
    if (dtype == TEO_F32) im2col_kernel<float><<<blocks, 256, 0, st>>>((const float*)px, (float*)cols, C, img, P, ld, total);
    else im2col_kernel<bf16_t><<<blocks, 256, 0, st>>>((const bf16_t*)px, (bf16_t*)cols, C, img, P, ld, total);
    TEO_LAUNCH_CHECK("im2col");
    return TEO_OK;
}

// ------------------------------------------------------------------------------------------------
// embedding splice: pure gather/copy driven by the host-built int32 plan
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void embed_splice_kernel(const int* __restrict__ plan, const T* __restrict__ embed,
                                                           const T* __restrict__ visual, T* __restrict__ out, int dim) {
    const long long r = blockIdx.x;
    const int p = plan[r];
    T* dst = out + r * dim;
    if (p == INT32_MIN) {
        for (int i = threadIdx.x; i < dim; i += 256) dst[i] = (T)0;
        return;
    }
    const T* src = (p >= 0) ? embed + (long long)p * dim : visual + (long long)(-(p + 1)) * dim;
    constexpr int VE = 16 / sizeof(T);
    if (dim % VE == 0 && ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0) {
        const uint4* s4 = reinterpret_cast<const uint4*>(src);
        uint4* d4 = reinterpret_cast<uint4*>(dst);
        for (int i = threadIdx.x; i < dim / VE; i += 256) d4[i] = s4[i];
    } else {
        for (int i = threadIdx.x; i < dim; i += 256) dst[i] = src[i];
    }
}

int embed_splice(const int* plan, const void* embed, const void* visual, void* out, int rows, int dim, int dtype,
                 hipStream_t st) {
    if (rows == 0) return TEO_OK;
    if (dtype == TEO_F32)
        embed_splice_kernel<float><<<rows, 256, 0, st>>>(plan, (const float*)embed, (const float*)visual, (float*)out, dim);
    else
        embed_splice_kernel<bf16_t><<<rows, 256, 0, st>>>(plan, (const bf16_t*)embed, (const bf16_t*)visual, (bf16_t*)out, dim);
    TEO_LAUNCH_CHECK("embed_splice");
    return TEO_OK;
}

template <typename T>
__global__ __launch_bounds__(256) void drop_cls_kernel(const T* __restrict__ in, T* __restrict__ out, int ntok, int dim) {
    const int t = blockIdx.x / (ntok - 1), p = blockIdx.x % (ntok - 1);
    const T* src = in + ((long long)t * ntok + 1 + p) * dim;
    T* dst = out + (long long)blockIdx.x * dim;
    for (int i = threadIdx.x; i < dim; i += 256) dst[i] = src[i];
}

int drop_cls(const void* in, void* out, int T_, int ntok, int dim, int dtype, hipStream_t st) {
    const int rows = T_ * (ntok - 1);
    if (rows <= 0) return TEO_OK;
    if (dtype == TEO_F32) drop_cls_kernel<float><<<rows, 256, 0, st>>>((const float*)in, (float*)out, ntok, dim);
    else drop_cls_kernel<bf16_t><<<rows, 256, 0, st>>>((const bf16_t*)in, (bf16_t*)out, ntok, dim);
    TEO_LAUNCH_CHECK("drop_cls");
    return TEO_OK;
}

// ------------------------------------------------------------------------------------------------
// argmax (first index on ties) + greedy bookkeeping
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void argmax_kernel(const float* __restrict__ logits, long long* __restrict__ tok,
                                                      int vocab) {
    __shared__ float sv[16];
    __shared__ int si[16];
    const float* row = logits + (long long)blockIdx.x * vocab;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = threadIdx.x; i < vocab; i += 1024) {
        const float v = row[i];
        if (v > best || bi == 0x7fffffff) { best = v; bi = i; }   // indices ascend per thread: strict > keeps the first
    }
    wave_argmax(best, bi);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sv[w] = best; si[w] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 16; ++k)
            if (sv[k] > best || (sv[k] == best && si[k] < bi)) { best = sv[k]; bi = si[k]; }
        tok[blockIdx.x] = bi;
    }
}

// ------------------------------------------------------------------------------------------------
// Sampler: temperature -> top-k -> softmax -> multinomial, one 1024-thread workgroup, logits stay in L2.
//   k-th largest value on the order-preserving uint image of the floats: vocab <= 32768 -> the row in registers (32 consecutive logits
//   per thread) and a bisection on the key; larger -> 4-pass radix select (LDS histogram).  Survivors (>= threshold, at most TOPK_CAP)
//   gathered in index order, softmax over them in fp32, inverse-CDF draw.  Both forms select the same set and draw the same token.
//   RNG: splitmix64(seed, draw) -> 24-bit uniform in [0,1): counter based, so a hipGraph replay only needs the draw
//   counter in device memory.
// ------------------------------------------------------------------------------------------------
constexpr int TOPK_CAP = 1024;

__device__ __forceinline__ unsigned f2ord(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float uniform01(unsigned long long seed, unsigned long long draw) {
    unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (draw + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (float)(z >> 40) * (1.0f / 16777216.0f);
}

// top-k disabled (top_k <= 0 or >= vocab, HF TopKLogitsWarper is then not applied) and vocab > TOPK_CAP: multinomial over the
// WHOLE vocabulary.  Thread t owns the contiguous index range [t*C, (t+1)*C): softmax in fp32, block scan of the range masses,
// inverse CDF in index order (the order torch.multinomial's CDF walks).  sel_val (>= 1024 floats) is the scan scratch.
__device__ int sample_full_block(const float* __restrict__ logits, int vocab, float temperature, float u, float* sel_val, int* s_misc) {
    const int tid = threadIdx.x;
    const float invt = 1.0f / fmaxf(temperature, 1e-6f);
    const int Cn = (vocab + 1023) / 1024;
    const int lo = min(tid * Cn, vocab), hi = min(lo + Cn, vocab);
    __shared__ float redf[16];
    float mx = -INFINITY;
    for (int i = lo; i < hi; ++i) mx = fmaxf(mx, logits[i] * invt);
    mx = wave_max(mx);
    if ((tid & 63) == 0) redf[tid >> 6] = mx;
    __syncthreads();
    mx = redf[0];
    for (int ww = 1; ww < 16; ++ww) mx = fmaxf(mx, redf[ww]);
    float mass = 0.f;
    for (int i = lo; i < hi; ++i) mass += expf(logits[i] * invt - mx);
    sel_val[tid] = mass;
    __syncthreads();
    if (tid == 0) {
        float tot = 0.f;
        for (int j = 0; j < 1024; ++j) tot += sel_val[j];
        const float target = u * tot;
        float run = 0.f;
        int owner = 1023;
        for (int j = 0; j < 1024; ++j) {
            if (run + sel_val[j] > target) { owner = j; break; }
            run += sel_val[j];
        }
        // walk the owner's range; `run` is the mass before it
        const int olo = min(owner * Cn, vocab), ohi = min(olo + Cn, vocab);
        int pick = max(ohi - 1, 0);
        for (int i = olo; i < ohi; ++i) {
            run += expf(logits[i] * invt - mx);
            if (run > target) { pick = i; break; }
        }
        if (ohi <= olo) pick = vocab - 1;
        s_misc[3] = pick;
    }
    __syncthreads();
    return s_misc[3];
}

// returns the sampled index to every thread; smem: caller provides the shared arrays
__device__ int sample_topk_block(const float* __restrict__ logits, int vocab, float temperature, int top_k, float top_p, float u,
                                 unsigned* hist, int* sel_idx, float* sel_val, int* s_misc) {
    const int tid = threadIdx.x;
    int k = top_k;
    if (k <= 0 || k > vocab) k = vocab;
    if (k == vocab && vocab > TOPK_CAP && !(top_p > 0.f && top_p < 1.f))      // filter off: full-vocabulary multinomial
        return sample_full_block(logits, vocab, temperature, u, sel_val, s_misc);
    if (k > TOPK_CAP) k = TOPK_CAP;                 // host entry points reject this case (teo_sampler_supported)
    // ---- the k-th largest key and the survivors (>= that key, ties kept) in index order
    constexpr int VPT = 32;                         // logits per thread of the register-resident form
    if (vocab <= 1024 * VPT && (reinterpret_cast<uintptr_t>(logits) & 15) == 0) {
        // round 4: every thread keeps its 32 CONSECUTIVE logits in registers (one pass over the 128 KB row instead of 4 radix passes + a
        // 32-strip compaction, and no LDS histogram: the first radix pass put ~all keys into a few exponent bins = serialised atomics).
        // The threshold is found by bisection on the 32-bit ordered key: 32 rounds of (32 compares per thread, one block-wide count).
        unsigned key[VPT];
        const int lo = tid * VPT;
#pragma unroll
        for (int j = 0; j < VPT; j += 4) {
            if (lo + j + 3 < vocab) {
                const float4 v = *reinterpret_cast<const float4*>(logits + lo + j);
                key[j] = f2ord(v.x); key[j + 1] = f2ord(v.y); key[j + 2] = f2ord(v.z); key[j + 3] = f2ord(v.w);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) key[j + e] = lo + j + e < vocab ? f2ord(logits[lo + j + e]) : 0u;    // 0: below every float's key
            }
        }
        // A round's count is taken wave-wide with the scalar unit: v_cmp writes the 64-lane mask, s_bcnt1 counts it -- one VALU
        // instruction per key instead of compare + add per lane and a cross-lane reduction at the end (the rounds are compare-bound:
        // an 8-way section search with 7 thresholds per round, 11 rounds, was slower -- 44 vs 34 us -- because it does 2.4x the compares).
        unsigned lo_k = 1u, hi_k = 0xFFFFFFFFu;       // invariant: count(key >= lo_k) >= k  (k <= vocab real keys, all >= 1)
        for (int it = 0; it < 32 && lo_k < hi_k; ++it) {
            const unsigned mid = lo_k + ((hi_k - lo_k) >> 1) + ((hi_k - lo_k) & 1u);       // upper middle: the range always shrinks
            int cw = 0;                               // wave-uniform
#pragma unroll
            for (int j = 0; j < VPT; ++j) cw += __popcll(__ballot(key[j] >= mid));
            unsigned* cnt = hist + (it & 1) * 16;
            if ((tid & 63) == 0) cnt[tid >> 6] = (unsigned)cw;
            __syncthreads();
            unsigned total = 0;
#pragma unroll
            for (int ww = 0; ww < 16; ++ww) total += cnt[ww];
            if ((int)total >= k) lo_k = mid; else hi_k = mid - 1u;
            if ((int)total == k) break;               // exactly the top k lie at or above mid: the survivor set is already final
        }
        const unsigned thr = lo_k;
        // ordered compaction: exclusive scan of the per-thread survivor counts (wave scan + wave totals)
        int mine = 0;
#pragma unroll
        for (int j = 0; j < VPT; ++j) mine += key[j] >= thr ? 1 : 0;
        const int lane = tid & 63, w = tid >> 6;
        int incl = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int up = __shfl_up(incl, o, 64);
            if (lane >= o) incl += up;
        }
        __syncthreads();                              // the counts of the last bisection round are read
        if (lane == 63) hist[32 + w] = (unsigned)incl;
        __syncthreads();
        int pos = incl - mine;
        int total = 0;
#pragma unroll
        for (int ww = 0; ww < 16; ++ww) {
            const int t = (int)hist[32 + ww];
            if (ww < w) pos += t;
            total += t;
        }
#pragma unroll
        for (int j = 0; j < VPT; ++j) {
            if (key[j] >= thr) {
                if (pos < TOPK_CAP) {
                    sel_idx[pos] = lo + j;
                    sel_val[pos] = __uint_as_float((key[j] & 0x80000000u) ? (key[j] ^ 0x80000000u) : ~key[j]);
                }
                ++pos;
            }
        }
        if (tid == 0) s_misc[2] = total;
        __syncthreads();
    } else {
        // ---- radix select of the k-th largest key
        unsigned prefix = 0, mask = 0;
        int want = k;                                   // rank (1 = largest) still to find inside the current prefix class
        for (int pass = 0; pass < 4; ++pass) {
            const int shift = 24 - 8 * pass;
            for (int i = tid; i < 256; i += 1024) hist[i] = 0;
            __syncthreads();
            for (int i = tid; i < vocab; i += 1024) {
                const unsigned key = f2ord(logits[i]);
                if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1u);
            }
            __syncthreads();
            if (tid == 0) {
                int acc = 0, b = 255;
                for (; b > 0; --b) {
                    if (acc + (int)hist[b] >= want) break;
                    acc += (int)hist[b];
                }
                s_misc[0] = b;
                s_misc[1] = want - acc;
            }
            __syncthreads();
            prefix |= ((unsigned)s_misc[0]) << shift;
            mask |= 255u << shift;
            want = s_misc[1];
            __syncthreads();
        }
        const unsigned thr = prefix;                    // key of the k-th largest logit; ties at thr are all kept (as torch.topk
                                                        // keeps an arbitrary subset, the distribution only differs on exact ties)
        // ---- gather survivors in index order (deterministic): ordered compaction by 1024-element strips
        if (tid == 0) s_misc[2] = 0;
        __syncthreads();
        for (int base = 0; base < vocab; base += 1024) {
            const int i = base + tid;
            const bool keep = i < vocab && f2ord(logits[i]) >= thr;
            const unsigned long long bal = __ballot(keep);
            const int lane = tid & 63, w = tid >> 6;
            if (lane == 0) hist[w] = (unsigned)__popcll(bal);
            __syncthreads();
            int off = s_misc[2];
            for (int ww = 0; ww < w; ++ww) off += (int)hist[ww];
            const int my = off + __popcll(bal & ((1ull << lane) - 1ull));
            if (keep && my < TOPK_CAP) { sel_idx[my] = i; sel_val[my] = logits[i]; }
            __syncthreads();
            if (tid == 0) { int t = 0; for (int ww = 0; ww < 16; ++ww) t += (int)hist[ww]; s_misc[2] += t; }
            __syncthreads();
        }
    }
    const int n = min(s_misc[2], TOPK_CAP);
    // ---- softmax over the survivors at the given temperature, inverse CDF in index order
    const float invt = 1.0f / fmaxf(temperature, 1e-6f);
    float mx = -INFINITY;
    for (int j = tid; j < n; j += 1024) mx = fmaxf(mx, sel_val[j] * invt);
    mx = wave_max(mx);
    __shared__ float redf[16];
    if ((tid & 63) == 0) redf[tid >> 6] = mx;
    __syncthreads();
    mx = redf[0];
    for (int ww = 1; ww < 16; ++ww) mx = fmaxf(mx, redf[ww]);
    __syncthreads();
    for (int j = tid; j < n; j += 1024) sel_val[j] = expf(sel_val[j] * invt - mx);
    __syncthreads();
    if (top_p > 0.f && top_p < 1.f) {
        // nucleus filter (HF TopPLogitsWarper, applied after temperature and top-k): sort ascending by probability, drop
        // every token whose cumulative probability (itself included) is <= 1 - top_p, always keep the most probable one.
        // Survivors are few (<= TOPK_CAP): thread j sums the mass of everything ranked at or below j (ties by index).
        __shared__ float s_tot;
        if (tid == 0) {
            float t = 0.f;
            for (int j = 0; j < n; ++j) t += sel_val[j];
            s_tot = t;
        }
        __syncthreads();
        float keepv = 0.f;
        bool mine = false;
        if (tid < n) {
            const float pj = sel_val[tid];
            float below = 0.f;
            bool is_max = true;
            for (int i = 0; i < n; ++i) {
                const float pi = sel_val[i];
                if (pi < pj || (pi == pj && i <= tid)) below += pi;
                if (pi > pj || (pi == pj && i > tid)) is_max = false;
            }
            mine = true;
            keepv = (below > (1.0f - top_p) * s_tot || is_max) ? pj : 0.f;
        }
        __syncthreads();
        if (mine) sel_val[tid] = keepv;
        __syncthreads();
    }
    if (tid == 0) {
        float tot = 0.f;
        for (int j = 0; j < n; ++j) tot += sel_val[j];
        const float target = u * tot;
        float run = 0.f;
        int pick = sel_idx[n - 1];
        for (int j = 0; j < n; ++j) {
            run += sel_val[j];
            if (run > target) { pick = sel_idx[j]; break; }
        }
        s_misc[3] = pick;
    }
    __syncthreads();
    return s_misc[3];
}

__global__ __launch_bounds__(1024) void sample_topk_kernel(const float* __restrict__ logits, long long* __restrict__ tok,
                                                           int vocab, float temperature, int top_k, float top_p,
                                                           unsigned long long seed, unsigned long long draw) {
    __shared__ __attribute__((aligned(16))) unsigned hist[256];
    __shared__ int sel_idx[TOPK_CAP];
    __shared__ float sel_val[TOPK_CAP];
    __shared__ int s_misc[4];
    const int pick = sample_topk_block(logits, vocab, temperature, top_k, top_p, uniform01(seed, draw), hist, sel_idx, sel_val, s_misc);
    if (threadIdx.x == 0) *tok = pick;
}

int sampler_check(int vocab, int top_k, float top_p) {
    const bool k_off = top_k <= 0 || top_k >= vocab;         // HF: TopKLogitsWarper not applied
    const bool p_on = top_p > 0.f && top_p < 1.f;
    if (!k_off && top_k > TOPK_CAP) {
        set_error("sampler: top_k %d above the device sampler's candidate cap %d (use top_k <= %d, or 0 to disable the filter)",
                  top_k, TOPK_CAP, TOPK_CAP);
        return TEO_ERR_UNSUPPORTED;
    }
    if (k_off && p_on && vocab > TOPK_CAP) {
        set_error("sampler: top_p %g without a top_k <= %d needs a nucleus filter over the whole vocabulary (%d): not implemented",
                  top_p, TOPK_CAP, vocab);
        return TEO_ERR_UNSUPPORTED;
    }
    return TEO_OK;
}

int sample_topk(const float* logits, long long* tok, int vocab, float temperature, int top_k, float top_p, unsigned long long seed,
                unsigned long long draw, hipStream_t st) {
    sample_topk_kernel<<<1, 1024, 0, st>>>(logits, tok, vocab, temperature, top_k, top_p, seed, draw);
    TEO_LAUNCH_CHECK("sample_topk");
    return TEO_OK;
}

// h[:] = embed[t][:]; optionally (batched bf16 step, ops.h SkinnyFuse) also hg = bf16(h * g) and the row's sum of squares
// in ssq[0] (ssq[1..nparts) zeroed) for the first consumer GEMM.  Called by a whole 1024-thread workgroup.
template <typename T>
__device__ __forceinline__ void embed_row_emit(const T* __restrict__ embed, long long t, T* __restrict__ h, int dim,
                                               const T* __restrict__ g, T* __restrict__ hg, float* __restrict__ ssq, int nparts,
                                               float* red16) {
    float sq = 0.f;
    for (int i = threadIdx.x; i < dim; i += 1024) {
        const T e = embed[t * dim + i];
        h[i] = e;
        if (hg) {
            const float ef = Elem<T>::ld(&e);
            Elem<T>::st(hg + i, ef * Elem<T>::ld(g + i));
            sq = fmaf(ef, ef, sq);
        }
    }
    if (hg) {
        sq = wave_sum(sq);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) red16[threadIdx.x >> 6] = sq;
        __syncthreads();
        if (threadIdx.x == 0) {
            float tot = 0.f;
            for (int k = 0; k < 16; ++k) tot += red16[k];
            ssq[0] = tot;
        }
        for (int i = 1 + threadIdx.x; i < nparts; i += 1024) ssq[i] = 0.f;
    }
}

// Decode tail in one launch: argmax over the fp32 logits (float4 loads), then thread 0 appends the token, advances the
// position and runs the id-suffix stop test, then the whole workgroup copies the next token's embedding row into h.
// (embed_next: the NEXT step's embedding lookup is hoisted here; the first step of a generation runs embed_token.)
template <typename T>
__global__ __launch_bounds__(1024) void decode_tail_kernel(const float* __restrict__ logits, teo_decode_state st,
                                                           const T* __restrict__ embed, T* __restrict__ h, int vocab,
                                                           int dim, int out_stride, const T* __restrict__ g0,
                                                           T* __restrict__ hg, float* __restrict__ ssq, int nparts) {
    {   // conversation blockIdx.x of a batched step (out_stride = row length of d_out_tokens)
        const long long b = blockIdx.x;
        logits += b * vocab;
        h += b * dim;
        if (hg) { hg += b * dim; ssq += b * nparts; }
        st.d_token += b; st.d_pos += b; st.d_out_count += b; st.d_stop += b;
        st.d_out_tokens += b * out_stride;
        if (st.d_rng) st.d_rng += 2 * b;
    }
    __shared__ float sv[16];
    __shared__ int si[16];
    __shared__ long long s_tok;
    __shared__ __attribute__((aligned(16))) unsigned hist[256];
    __shared__ int sel_idx[TOPK_CAP];
    __shared__ float sel_val[TOPK_CAP];
    __shared__ int s_misc[4];
    int sampled = -1;
    if (st.do_sample) {
        const unsigned long long draw = st.d_rng[1];
        sampled = sample_topk_block(logits, vocab, st.temperature, st.top_k, st.top_p, uniform01(st.d_rng[0], draw), hist, sel_idx,
                                    sel_val, s_misc);
        if (threadIdx.x == 0) st.d_rng[1] = draw + 1;
    }
    float best = -INFINITY;
    int bi = 0x7fffffff;
    const int nv4 = st.do_sample ? 0 : vocab >> 2;          // the argmax pass only when the token is not sampled
    const float4* l4 = reinterpret_cast<const float4*>(logits);
    for (int i = threadIdx.x; i < nv4; i += 1024) {
        const float4 v = l4[i];
        const int b = i << 2;
        if (v.x > best || bi == 0x7fffffff) { best = v.x; bi = b; }
        if (v.y > best) { best = v.y; bi = b + 1; }
        if (v.z > best) { best = v.z; bi = b + 2; }
        if (v.w > best) { best = v.w; bi = b + 3; }
    }
    for (int i = (nv4 << 2) + threadIdx.x; i < (st.do_sample ? 0 : vocab); i += 1024) {
        const float v = logits[i];
        if (v > best || bi == 0x7fffffff) { best = v; bi = i; }
    }
    wave_argmax(best, bi);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sv[w] = best; si[w] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 16; ++k)
            if (sv[k] > best || (sv[k] == best && si[k] < bi)) { best = sv[k]; bi = si[k]; }
        const long long t = st.do_sample ? sampled : bi;
        *st.d_token = t;
        const int n = *st.d_out_count;
        st.d_out_tokens[n] = t;
        *st.d_out_count = n + 1;
        *st.d_pos = *st.d_pos + 1;
        if (st.d_stop_ids && st.n_stop_ids > 0 && n + 1 >= st.n_stop_ids) {
            bool eq = true;
            for (int k = 0; k < st.n_stop_ids; ++k)
                eq = eq && (st.d_out_tokens[n + 1 - st.n_stop_ids + k] == st.d_stop_ids[k]);
            if (eq) *st.d_stop = 1;
        }
        s_tok = t;
    }
    __syncthreads();
    const long long t = s_tok;
    embed_row_emit<T>(embed, t, h, dim, g0, hg, ssq, nparts, sv);
}

int decode_tail(const float* logits, const teo_decode_state* s, const void* embed, void* h, int vocab, int dim, int dtype,
                hipStream_t st, int batch, int out_stride, const void* g0, void* hg, float* ssq, int nparts) {
    if (dtype == TEO_F32)
        TEO_KLAUNCH((decode_tail_kernel<float>), batch, 1024, 0, st, logits, *s, (const float*)embed, (float*)h, vocab, dim, out_stride,
                    (const float*)g0, (float*)hg, ssq, nparts);
    else if (dtype == TEO_F16)
        TEO_KLAUNCH((decode_tail_kernel<f16_t>), batch, 1024, 0, st, logits, *s, (const f16_t*)embed, (f16_t*)h, vocab, dim, out_stride,
                    (const f16_t*)g0, (f16_t*)hg, ssq, nparts);
    else
        TEO_KLAUNCH((decode_tail_kernel<bf16_t>), batch, 1024, 0, st, logits, *s, (const bf16_t*)embed, (bf16_t*)h, vocab, dim, out_stride,
                    (const bf16_t*)g0, (bf16_t*)hg, ssq, nparts);
    TEO_LAUNCH_CHECK("decode_tail");
    return TEO_OK;
}

int argmax(const float* logits, long long* tok, int rows, int vocab, hipStream_t st) {
    if (rows == 0) return TEO_OK;
    argmax_kernel<<<rows, 1024, 0, st>>>(logits, tok, vocab);
    TEO_LAUNCH_CHECK("argmax");
    return TEO_OK;
}

// after argmax wrote the next token into st.d_token: append it, advance the position, check the stop suffix
__global__ void decode_advance_kernel(teo_decode_state st) {
    if (threadIdx.x != 0) return;
    const int n = *st.d_out_count;
    const long long t = *st.d_token;
    st.d_out_tokens[n] = t;
    *st.d_out_count = n + 1;
    *st.d_pos = *st.d_pos + 1;
    if (st.d_stop_ids && st.n_stop_ids > 0 && n + 1 >= st.n_stop_ids) {
        bool eq = true;
        for (int k = 0; k < st.n_stop_ids; ++k)
            eq = eq && (st.d_out_tokens[n + 1 - st.n_stop_ids + k] == st.d_stop_ids[k]);
        if (eq) *st.d_stop = 1;
    }
}

int decode_advance(const teo_decode_state* s, hipStream_t st) {
    decode_advance_kernel<<<1, 64, 0, st>>>(*s);
    TEO_LAUNCH_CHECK("decode_advance");
    return TEO_OK;
}

// h[:] = embed[token][:]
template <typename T>
__global__ __launch_bounds__(256) void embed_token_kernel(const long long* __restrict__ tok, const T* __restrict__ embed,
                                                          T* __restrict__ h, int dim) {
    const long long t = tok[blockIdx.y];
    h += (long long)blockIdx.y * dim;
    for (int i = threadIdx.x + blockIdx.x * 256; i < dim; i += 256 * gridDim.x) h[i] = embed[t * dim + i];
}

// one 1024-thread workgroup per conversation: h = embed[tok], hg / ssq for the first consumer GEMM (embed_row_emit)
template <typename T>
__global__ __launch_bounds__(1024) void embed_emit_kernel(const long long* __restrict__ tok, const T* __restrict__ embed,
                                                          T* __restrict__ h, int dim, const T* __restrict__ g, T* __restrict__ hg,
                                                          float* __restrict__ ssq, int nparts) {
    __shared__ float red16[16];
    const long long b = blockIdx.x;
    embed_row_emit<T>(embed, tok[b], h + b * dim, dim, g, hg + b * dim, ssq + b * nparts, nparts, red16);
}

int embed_token_emit(const long long* tok, const void* embed, void* h, int dim, int dtype, hipStream_t st, int batch,
                     const void* g, void* hg, float* ssq, int nparts) {
    if (dtype == TEO_F32)
        embed_emit_kernel<float><<<batch, 1024, 0, st>>>(tok, (const float*)embed, (float*)h, dim, (const float*)g, (float*)hg, ssq, nparts);
    else if (dtype == TEO_F16)
        embed_emit_kernel<f16_t><<<batch, 1024, 0, st>>>(tok, (const f16_t*)embed, (f16_t*)h, dim, (const f16_t*)g, (f16_t*)hg, ssq, nparts);
    else
        embed_emit_kernel<bf16_t><<<batch, 1024, 0, st>>>(tok, (const bf16_t*)embed, (bf16_t*)h, dim, (const bf16_t*)g, (bf16_t*)hg, ssq, nparts);
    TEO_LAUNCH_CHECK("embed_token_emit");
    return TEO_OK;
}

int embed_token(const long long* tok, const void* embed, void* h, int dim, int dtype, hipStream_t st, int batch) {
    const dim3 blocks(cdiv(dim, 256), batch);
    if (dtype == TEO_F32) embed_token_kernel<float><<<blocks, 256, 0, st>>>(tok, (const float*)embed, (float*)h, dim);
    else embed_token_kernel<bf16_t><<<blocks, 256, 0, st>>>(tok, (const bf16_t*)embed, (bf16_t*)h, dim);
    TEO_LAUNCH_CHECK("embed_token");
    return TEO_OK;
}

// ---- training-shape loss: CrossEntropyLoss(ignore_index) of LlamaForCausalLM.forward with labels
// (the call at videollava/model/language_model/llava_llama.py:88-99).  One workgroup per row: max, log-sum-exp, picked
// logit -> loss_row[r] (0 for ignored rows); a single workgroup then sums rows and counts in a fixed order (deterministic).
__global__ __launch_bounds__(256) void ce_rows_kernel(const float* __restrict__ logits, long long ld, const long long* __restrict__ labels,
                                                      float* __restrict__ loss_row, int vocab, long long ignore_index) {
    __shared__ float red[4];
    const int r = blockIdx.x;
    const long long lab = labels[r];
    if (lab == ignore_index || lab < 0 || lab >= vocab) {      // out-of-range labels are rejected on the host
        if (threadIdx.x == 0) loss_row[r] = 0.f;
        return;
    }
    const float* row = logits + (long long)r * ld;
    float m = -INFINITY;
    for (int i = threadIdx.x; i < vocab; i += 256) m = fmaxf(m, row[i]);
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float s = 0.f;
    for (int i = threadIdx.x; i < vocab; i += 256) s += expf(row[i] - m);
    s = block_sum<256>(s, red);
    if (threadIdx.x == 0) loss_row[r] = logf(s) + m - row[lab];
}

__global__ __launch_bounds__(256) void ce_reduce_kernel(const float* __restrict__ loss_row, const long long* __restrict__ labels,
                                                        int rows, int vocab, long long ignore_index, float* __restrict__ out) {
    __shared__ float red[4];
    float s = 0.f, n = 0.f;
    for (int i = threadIdx.x; i < rows; i += 256) {
        const long long lab = labels[i];
        if (lab != ignore_index && lab >= 0 && lab < vocab) { s += loss_row[i]; n += 1.f; }
    }
    s = block_sum<256>(s, red);
    n = block_sum<256>(n, red);
    if (threadIdx.x == 0) { out[0] = s / n; out[1] = s; out[2] = n; }     // 0/0 = nan, as torch's mean reduction
}

int cross_entropy(const float* logits, long long ld, const long long* labels, float* loss_row, float* out, int rows, int vocab,
                  long long ignore_index, hipStream_t st) {
    if (rows > 0) {
        ce_rows_kernel<<<rows, 256, 0, st>>>(logits, ld, labels, loss_row, vocab, ignore_index);
        TEO_LAUNCH_CHECK("ce_rows");
    }
    ce_reduce_kernel<<<1, 256, 0, st>>>(loss_row, labels, rows, vocab, ignore_index, out);
    TEO_LAUNCH_CHECK("ce_reduce");
    return TEO_OK;
}

}  // namespace teo
