// Pieces shared by the 3x3 convolution translation units (conv3x3.hip, conv3x3_mid.hip).
#pragma once
#include "ngan_common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

namespace ngan {
// (a named namespace: the launchers that take it cross translation units)
struct ConvArgs {
    const float* x; const float* wp; const float* bias; float* y; float* rn;
    int B, H, W, K, N, tiles_x, tiles_y;
    float slope, eps;
    const float* ay; const float* arn; float* aout;
};
}  // namespace ngan
using ngan::ConvArgs;

namespace {

// Epilogues of the forward / input-gradient kernels (template parameter EPI):
//   0  y = conv + bias
//   1  y = PixelNorm(LeakyReLU(conv + bias)), rn = the per-pixel norm
//   2  the kernel computes an INPUT GRADIENT g (of the next layer); its epilogue applies the backward of the LeakyReLU -> PixelNorm
//      that produced this layer's input:  y = m * (g - ay * mean_c(g * ay)) / arn,  m = ay > 0 ? 1 : slope
//      (ay = that input, i.e. the previous layer's output, arn its norms; same shape as y, also for the pool-adjoint store)
//   3  epilogue 1 followed by ToImage: aout[pixel] = tanh(sum_c ay[c] * y[c])  (ay = the 1x1 colour weights, one colour);
//      y and rn are stored only if y != nullptr
enum { EPI_NONE = 0, EPI_LRELU_PN = 1, EPI_PN_BWD = 2, EPI_TO_IMAGE = 3 };


__device__ __forceinline__ float4 pn_bwd4(float4 g, float4 yy, float s, float inv_r, float slope) {
    return make_float4((g.x - yy.x * s) * inv_r * (yy.x > 0.f ? 1.f : slope), (g.y - yy.y * s) * inv_r * (yy.y > 0.f ? 1.f : slope),
                       (g.z - yy.z * s) * inv_r * (yy.z > 0.f ? 1.f : slope), (g.w - yy.w * s) * inv_r * (yy.w > 0.f ? 1.f : slope));
}

// 4 consecutive channels (starting at ch) of conv-input pixel (gy, gx) of image b, after resampling.
// C = channel count of x.  Out-of-image pixels are the conv's zero padding.
template <int RES>
__device__ __forceinline__ float4 load_resampled(const float* __restrict__ x, int b, int gy, int gx, int ch,
                                                 int H, int W, int C) {
    if (gy < 0 || gy >= H || gx < 0 || gx >= W) return f4zero();
    if (RES == NGAN_RESAMPLE_NONE) {
        return ld4(x + (((long)b * H + gy) * W + gx) * C + ch);
    } else if (RES == NGAN_RESAMPLE_POOL2) {
        const long W2 = 2L * W;
        const float* p = x + (((long)b * 2 * H + 2 * gy) * W2 + 2 * gx) * C + ch;
        float4 v = f4add(f4add(ld4(p), ld4(p + C)), f4add(ld4(p + W2 * C), ld4(p + W2 * C + C)));
        return f4scale(v, 0.25f);
    } else {
        const int h = H >> 1, w = W >> 1;
        int y0, y1, x0, x1; float wy0, wy1, wx0, wx1;
        up2_taps(gy, h, y0, y1, wy0, wy1);
        up2_taps(gx, w, x0, x1, wx0, wx1);
        const float* r0 = x + ((long)b * h + y0) * w * C + ch;
        const float* r1 = x + ((long)b * h + y1) * w * C + ch;
        float4 top = f4fma(ld4(r0 + (long)x1 * C), wx1, f4scale(ld4(r0 + (long)x0 * C), wx0));
        float4 bot = f4fma(ld4(r1 + (long)x1 * C), wx1, f4scale(ld4(r1 + (long)x0 * C), wx0));
        return f4fma(bot, wy1, f4scale(top, wy0));
    }
}

}  // namespace

// conv3x3_mid.hip: split-bf16 kernel for many-channel layers on small images (K, N multiples of 32, up to 128)
namespace ngan {
bool conv3x3_mid_eligible(int B, int H, int W, int K, int N);
bool conv3x3_mid_fuses_epilogue(int B, int H, int W, int K, int N);   // all N channels of a pixel in one workgroup?
int conv3x3_mid_launch(const float* x, const float* packed, const float* bias, float* y, float* rnorm, const float* aux_in,
                       const float* aux_rn, int B, int H, int W, int K, int N, int resample, int epilogue, int out_mode,
                       float slope, float eps, int precision, hipStream_t s);   // precision 0: the exact-fp32 variant
int conv3x3_mid_kernel_name(int B, int H, int W, int K, int N, int resample, int epilogue, int out_mode, int precision, char* buf, int len);
}  // namespace ngan
