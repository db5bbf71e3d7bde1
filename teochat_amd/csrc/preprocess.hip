// On-device image preprocessing: the reference's get_image_transform
// (languagebind/image/processing_image.py:15-25) -- ToTensor (uint8 HWC -> float CHW / 255) -> Resize(S, bicubic,
// antialias as torchvision 0.17 does for tensors) -> CenterCrop(S) -> Normalize(mean, std) -- in one kernel, written
// straight in the layout and dtype teo_vit_encode consumes.  The host uploads the raw uint8 frame (4x fewer PCIe bytes
// than the float tensor) and never touches the pixels.
//
// Resampling follows ATen's separable anti-aliased bicubic (UpSampleKernel.cpp `_compute_indices_weights_aa`, a = -0.5):
//   scale = in/out; support = 2*max(scale,1); center = scale*(i+0.5); taps [int(center-support+0.5), int(center+support+0.5))
//   clipped to the image; w_j = cubic((j - center + 0.5) / max(scale,1)), normalised to sum 1.
// Each workgroup owns a 16x16 tile of one output frame: the tile's 16 row-weight and 16 column-weight vectors are built
// once in LDS, then every thread accumulates its pixel for the three channels.
#include "common.h"
#include "ops.h"

namespace teo {

constexpr int PP_MAX_TAPS = 64;          // supports down-scaling by up to ~15x (3360 px -> 224)

__device__ __forceinline__ float cubic_aa(float x) {
    const float a = -0.5f;
    x = fabsf(x);
    if (x < 1.0f) return ((a + 2.0f) * x - (a + 3.0f)) * x * x + 1.0f;
    if (x < 2.0f) return (((x - 5.0f) * x + 8.0f) * x - 4.0f) * a;
    return 0.0f;
}

// taps of output index i along an axis of `in_size` input samples resized to `out_size`
__device__ __forceinline__ void aa_taps(int i, int in_size, int out_size, int& xmin, int& xsize, float* w) {
    const float scale = (float)in_size / (float)out_size;
    const float support = scale >= 1.0f ? 2.0f * scale : 2.0f;
    const float invscale = scale >= 1.0f ? 1.0f / scale : 1.0f;
    const float center = scale * ((float)i + 0.5f);
    xmin = max((int)(center - support + 0.5f), 0);
    xsize = min((int)(center + support + 0.5f), in_size) - xmin;
    xsize = max(min(xsize, PP_MAX_TAPS), 0);
    float total = 0.f;
    for (int j = 0; j < xsize; ++j) {
        const float v = cubic_aa(((float)(j + xmin) - center + 0.5f) * invscale);
        w[j] = v;
        total += v;
    }
    const float inv = total != 0.f ? 1.0f / total : 0.f;
    for (int j = 0; j < xsize; ++j) w[j] *= inv;
}

// PAD: the source is pasted at (oy, ox) into a virtual Hc x Wc canvas filled with (f0, f1, f2) -- expand2square of the
// reference (mm_utils.py:14-25, image_aspect_ratio == 'pad') without materialising the padded image.
template <typename T, bool PAD>
__global__ __launch_bounds__(256) void preprocess_kernel(const unsigned char* __restrict__ src, T* __restrict__ out, int H,
                                                         int W, int nh, int nw, int top, int left, int S, float m0, float m1,
                                                         float m2, float is0, float is1, float is2, int Hs, int Ws, int oy,
                                                         int ox, float f0, float f1, float f2) {
    __shared__ float wy[16][PP_MAX_TAPS], wx[16][PP_MAX_TAPS];
    __shared__ int ymin[16], ysize[16], xmin[16], xsize[16];
    const int t = blockIdx.z, ty0 = blockIdx.y * 16, tx0 = blockIdx.x * 16;
    const int tid = threadIdx.x;
    if (tid < 16) aa_taps(ty0 + tid + top, H, nh, ymin[tid], ysize[tid], wy[tid]);
    else if (tid < 32) aa_taps(tx0 + (tid - 16) + left, W, nw, xmin[tid - 16], xsize[tid - 16], wx[tid - 16]);
    __syncthreads();
    const int ly = tid >> 4, lx = tid & 15;
    const int y = ty0 + ly, x = tx0 + lx;
    if (y >= S || x >= S) return;
    const unsigned char* img = src + (size_t)t * (PAD ? Hs * Ws : H * W) * 3;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    const int y0 = ymin[ly], ny = ysize[ly], x0 = xmin[lx], nx = xsize[lx];
    for (int j = 0; j < ny; ++j) {
        float r0 = 0.f, r1 = 0.f, r2 = 0.f;
        if (PAD) {
            const int sy = y0 + j - oy;                                    // row of the source image, if inside it
            const bool yin = sy >= 0 && sy < Hs;
            const unsigned char* row = img + (size_t)(yin ? sy : 0) * Ws * 3;
            for (int i = 0; i < nx; ++i) {
                const float w = wx[lx][i];
                const int sx = x0 + i - ox;
                const bool in = yin && sx >= 0 && sx < Ws;
                const int sxx = in ? sx : 0;
                r0 = fmaf(w, in ? (float)row[3 * sxx] : f0, r0);
                r1 = fmaf(w, in ? (float)row[3 * sxx + 1] : f1, r1);
                r2 = fmaf(w, in ? (float)row[3 * sxx + 2] : f2, r2);
            }
        } else {
            const unsigned char* row = img + ((size_t)(y0 + j) * W + x0) * 3;
            for (int i = 0; i < nx; ++i) {
                const float w = wx[lx][i];
                r0 = fmaf(w, (float)row[3 * i], r0);
                r1 = fmaf(w, (float)row[3 * i + 1], r1);
                r2 = fmaf(w, (float)row[3 * i + 2], r2);
            }
        }
        const float w = wy[ly][j];
        a0 = fmaf(w, r0, a0); a1 = fmaf(w, r1, a1); a2 = fmaf(w, r2, a2);
    }
    const float k = 1.0f / 255.0f;
    T* o = out + (size_t)t * 3 * S * S + (size_t)y * S + x;
    Elem<T>::st(o, (a0 * k - m0) * is0);
    Elem<T>::st(o + (size_t)S * S, (a1 * k - m1) * is1);
    Elem<T>::st(o + (size_t)2 * S * S, (a2 * k - m2) * is2);
}

int preprocess_frames(const unsigned char* src, void* out, int T_, int Hs, int Ws, int S, const float* mean, const float* stdv,
                      int dtype, hipStream_t st, const unsigned char* pad_rgb) {
    if (T_ == 0) return TEO_OK;
    // pad_rgb: expand2square first -- the resize then sees a side x side canvas with the source pasted centred
    const bool pad = pad_rgb != nullptr && Hs != Ws;
    const int side = Hs > Ws ? Hs : Ws;
    const int H = pad ? side : Hs, W = pad ? side : Ws;
    const int oy = pad ? (side - Hs) / 2 : 0, ox = pad ? (side - Ws) / 2 : 0;
    // Resize(S): shorter edge -> S, the other int(S * long / short) (torchvision); CenterCrop(S): offset round((n - S) / 2)
    int nh, nw;
    if (H <= W) { nh = S; nw = (int)((long long)S * W / H); } else { nw = S; nh = (int)((long long)S * H / W); }
    const float sc = fmaxf((float)H / nh, (float)W / nw);
    if (2.0f * fmaxf(sc, 1.0f) * 2.0f + 2.0f > PP_MAX_TAPS) {
        set_error("teo_preprocess_frames: %dx%d -> %d needs more than %d taps per axis", H, W, S, PP_MAX_TAPS);
        return TEO_ERR_UNSUPPORTED;
    }
    const int top = (int)lrintf((nh - S) / 2.0f), left = (int)lrintf((nw - S) / 2.0f);
    const dim3 grid(cdiv(S, 16), cdiv(S, 16), T_);
    const float f0 = pad ? (float)pad_rgb[0] : 0.f, f1 = pad ? (float)pad_rgb[1] : 0.f, f2 = pad ? (float)pad_rgb[2] : 0.f;
#define TEO_PP(TT, PD)                                                                                                      \
    preprocess_kernel<TT, PD><<<grid, 256, 0, st>>>(src, (TT*)out, H, W, nh, nw, top, left, S, mean[0], mean[1], mean[2],    \
                                                    1.0f / stdv[0], 1.0f / stdv[1], 1.0f / stdv[2], Hs, Ws, oy, ox, f0, f1, f2)
    if (dtype == TEO_F32) { if (pad) TEO_PP(float, true); else TEO_PP(float, false); }
    else if (dtype == TEO_F16) { if (pad) TEO_PP(f16_t, true); else TEO_PP(f16_t, false); }
    else { if (pad) TEO_PP(bf16_t, true); else TEO_PP(bf16_t, false); }
#undef TEO_PP
    TEO_LAUNCH_CHECK("preprocess_frames");
    return TEO_OK;
}

}  // namespace teo
