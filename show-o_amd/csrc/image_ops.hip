// Image pre / post-processing on the device: the steps either side of MAGVITv2.get_code / decode_code
// that the reference runs on the host through torchvision + PIL (training/utils.py:178-185 `image_transform`;
// inference_t2i.py:100-110 inpainting-mask down-sampling; inference_t2i.py:157-159 uint8 conversion).
//
// Byte / integer work, HBM-bound, one thread per output element:
//  * resize = PIL's antialiased bicubic (Pillow src/libImaging/Resample.c: a = -0.5 kernel whose support scales with the
//    down-sampling factor, 22-bit fixed-point coefficients, horizontal pass -> uint8 -> vertical pass -> uint8).  The per-axis
//    coefficient tables are tiny (out_size x ksize) and depend only on the sizes: the host computes them in double exactly as
//    Pillow does (show-o_amd/image_utils.py) and the kernels do the integer accumulation, so the result equals PIL's byte for byte;
//  * the vertical pass also applies CenterCrop + ToTensor (/255) + Normalize ((x - 0.5) / 0.5) and writes fp32 CHW;
//  * uint8 conversion: clamp((x + 1) / 2, 0, 1) * 255 -> truncate, NCHW fp32 -> NHWC uint8;
//  * mask down-sampling: torch's bicubic (A = -0.75, align_corners = False, no antialias) + the 0.5 threshold.
#include "common.h"
#include "../../include/showo_hip.h"

using namespace showo;

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;  // Pillow Resample.c

__device__ inline uint8_t clip8(int v) {  // clip8_lookups[v >> PRECISION_BITS]
    v >>= PRECISION_BITS;
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// in u8 [H, W, C] -> tmp u8 [rows, Wout, C] for rows y0 .. y0+rows-1 of the input
__global__ void resample_h_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ tmp, const int32_t* __restrict__ bounds,
                                  const int32_t* __restrict__ kk, int ksize, int W, int C, int Wout, int y0, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % C);
    const int xo = (int)((i / C) % Wout);
    const int64_t y = i / ((int64_t)C * Wout);
    const int xmin = bounds[2 * xo], xmax = bounds[2 * xo + 1];
    const int32_t* k = kk + (int64_t)xo * ksize;
    const uint8_t* row = in + ((y0 + y) * W + xmin) * C + c;
    int ss = 1 << (PRECISION_BITS - 1);
    for (int x = 0; x < xmax; ++x) ss += (int)row[(int64_t)x * C] * k[x];
    tmp[i] = clip8(ss);
}

// tmp u8 [rows, Wout, C] (row 0 = input row y0) -> out fp32 [C, R, R]: vertical pass for output rows ct..ct+R-1 and columns
// cl..cl+R-1 (the centre crop), then ToTensor / Normalize
__global__ void resample_v_kernel(const uint8_t* __restrict__ tmp, float* __restrict__ out, uint8_t* __restrict__ out_u8,
                                  const int32_t* __restrict__ bounds, const int32_t* __restrict__ kk, int ksize, int C, int Wout, int y0,
                                  int R_h, int R_w, int ct, int cl, int normalize, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int x = (int)(i % R_w);
    const int y = (int)((i / R_w) % R_h);
    const int c = (int)(i / ((int64_t)R_w * R_h));
    const int yo = ct + y;
    const int ymin = bounds[2 * yo], ymax = bounds[2 * yo + 1];
    const int32_t* k = kk + (int64_t)yo * ksize;
    const uint8_t* col = tmp + ((int64_t)(ymin - y0) * Wout + (cl + x)) * C + c;
    int ss = 1 << (PRECISION_BITS - 1);
    for (int t = 0; t < ymax; ++t) ss += (int)col[(int64_t)t * Wout * C] * k[t];
    const uint8_t v = clip8(ss);
    if (out_u8) out_u8[((int64_t)y * R_w + x) * C + c] = v;  // the resized + cropped bytes themselves (HWC), for inspection / parity
    float f = __fdiv_rn((float)v, 255.0f);                   // ToTensor
    if (normalize) f = __fdiv_rn(__fsub_rn(f, 0.5f), 0.5f);  // Normalize(mean 0.5, std 0.5)
    out[i] = f;
}

// images fp32 [B,C,H,W] in [-1,1] -> uint8 [B,H,W,C]: clamp((x + 1) / 2, 0, 1) * 255, truncated (numpy astype(uint8))
__global__ void to_uint8_kernel(const float* __restrict__ x, uint8_t* __restrict__ out, int C, int H, int W, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % C);
    const int w = (int)((i / C) % W);
    const int h = (int)((i / ((int64_t)C * W)) % H);
    const int64_t b = i / ((int64_t)C * W * H);
    float v = __fdiv_rn(__fadd_rn(x[((b * C + c) * H + h) * W + w], 1.0f), 2.0f);
    v = fminf(fmaxf(v, 0.0f), 1.0f);
    v = __fmul_rn(v, 255.0f);
    out[i] = (uint8_t)(int)v;  // NaN -> 0 like the cast of a clamped NaN is undefined in numpy; the decoder never produces one
}

// torch upsample_bicubic2d (A = -0.75, align_corners = False): mask fp32 [S,S] -> bool [s*s] = (value >= 0.5)
__device__ inline float cc1(float x, float A) { return ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f; }
__device__ inline float cc2(float x, float A) { return ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A; }
__global__ void mask_bicubic_threshold_kernel(const float* __restrict__ m, uint8_t* __restrict__ out, float* __restrict__ val, int S, int s) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= s * s) return;
    const int oy = i / s, ox = i % s;
    const float scale = (float)S / (float)s;
    const float A = -0.75f;
    const float ry = scale * ((float)oy + 0.5f) - 0.5f, rx = scale * ((float)ox + 0.5f) - 0.5f;
    const float fy = floorf(ry), fx = floorf(rx);
    const int iy = (int)fy, ix = (int)fx;
    const float ty = ry - fy, tx = rx - fx;
    const float wx[4] = {cc2(tx + 1.f, A), cc1(tx, A), cc1(1.f - tx, A), cc2(1.f - tx + 1.f, A)};
    const float wy[4] = {cc2(ty + 1.f, A), cc1(ty, A), cc1(1.f - ty, A), cc2(1.f - ty + 1.f, A)};
    float rows[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int y = min(max(iy - 1 + a, 0), S - 1);
        float r = 0.f;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int x = min(max(ix - 1 + b, 0), S - 1);
            r += m[(int64_t)y * S + x] * wx[b];
        }
        rows[a] = r;
    }
    const float v = rows[0] * wy[0] + rows[1] * wy[1] + rows[2] * wy[2] + rows[3] * wy[3];
    if (val) val[i] = v;
    out[i] = v >= 0.5f ? 1 : 0;
}

int blocks(int64_t n) { return (int)((n + 255) / 256); }

}  // namespace

// img u8 [H,W,C] (C = 1 or 3); horizontal table (bounds_h int32 [Wout,2], kk_h int32 [Wout,ksize_h]) maps W -> Wout, vertical
// table (bounds_v [Hout,2], kk_v [Hout,ksize_v]) maps H -> Hout (an axis that keeps its size gets the identity table).
// out fp32 [C, crop_h, crop_w] = rows ct.., columns cl.. of the resized image after ToTensor (+ Normalize); tmp u8 scratch of
// at least H*Wout*C bytes; out_u8 (optional) u8 [crop_h, crop_w, C].
extern "C" int showo_image_resize_crop_normalize(const uint8_t* img, int H, int W, int C, int Hout, int Wout, const int32_t* bounds_h,
                                                 const int32_t* kk_h, int ksize_h, const int32_t* bounds_v, const int32_t* kk_v,
                                                 int ksize_v, int ct, int cl, int crop_h, int crop_w, int normalize, uint8_t* tmp,
                                                 float* out, uint8_t* out_u8, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (!img || !out || !tmp) return set_error_msg(1, "image_transform: null argument");
    if ((C != 1 && C != 3) || H <= 0 || W <= 0 || Hout <= 0 || Wout <= 0) return set_error_msg(1, "image_transform: bad shape");
    if (ct < 0 || cl < 0 || ct + crop_h > Hout || cl + crop_w > Wout) return set_error_msg(1, "image_transform: crop outside the resized image");
    if (!bounds_h || !kk_h || !bounds_v || !kk_v || ksize_h < 1 || ksize_v < 1)
        return set_error_msg(1, "image_transform: both coefficient tables are required (identity table = one tap of 1 << 22)");
    // Pillow: horizontal pass first (every input row), uint8 intermediate, then the vertical pass
    const int64_t n1 = (int64_t)H * Wout * C;
    resample_h_kernel<<<dim3(blocks(n1)), dim3(256), 0, s>>>(img, tmp, bounds_h, kk_h, ksize_h, W, C, Wout, 0, n1);
    const int64_t n2 = (int64_t)C * crop_h * crop_w;
    resample_v_kernel<<<dim3(blocks(n2)), dim3(256), 0, s>>>(tmp, out, out_u8, bounds_v, kk_v, ksize_v, C, Wout, 0, crop_h, crop_w, ct, cl,
                                                             normalize, n2);
    SHOWO_CHECK_HIP(hipGetLastError());
    return 0;
}

// x fp32 [B,C,H,W] -> out u8 [B,H,W,C] (inference_t2i.py:157-159)
extern "C" int showo_images_to_uint8(const float* x, uint8_t* out, int B, int C, int H, int W, void* stream) {
    if (B <= 0) return 0;
    if (!x || !out) return set_error_msg(1, "images_to_uint8: null argument");
    const int64_t n = (int64_t)B * C * H * W;
    to_uint8_kernel<<<dim3(blocks(n)), dim3(256), 0, (hipStream_t)stream>>>(x, out, C, H, W, n);
    SHOWO_CHECK_HIP(hipGetLastError());
    return 0;
}

// mask fp32 [S,S] -> out u8 [s*s] = (bicubic down-sample >= 0.5) (inference_t2i.py:100-108); values (optional) fp32 [s*s]
extern "C" int showo_mask_downsample_threshold(const float* mask, int S, int s_out, uint8_t* out, float* values, void* stream) {
    if (!mask || !out || S <= 0 || s_out <= 0) return set_error_msg(1, "mask_downsample: bad argument");
    mask_bicubic_threshold_kernel<<<dim3(blocks((int64_t)s_out * s_out)), dim3(256), 0, (hipStream_t)stream>>>(mask, out, values, S, s_out);
    SHOWO_CHECK_HIP(hipGetLastError());
    return 0;
}
