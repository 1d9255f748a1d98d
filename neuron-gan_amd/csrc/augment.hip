// On-device input pipeline for the real images: the per-sample transform chain of
// /root/reference/data/NeuronDataset.py:112-126, 149-164 -- RandomAffine(rotation + integer translation, nearest, fill 0) ->
// RandomVerticalFlip -> ColorJitter(brightness, contrast; random order) -> CenterCrop -> Renormalize([0,1] -> [-1,1]) ->
// Resize(stage size, bilinear, antialias) -- as two launches over a whole batch, so that a training step that takes ~0.5 ms per
// image is never fed by a per-image Python/PIL loop.  One colour channel (the reference's images are greyscale).
//   pass 1 (augment_affine_kernel): canvas[b] = flip(affine(src[idx[b]])) (and brightness, clamped, when it comes first) on the
//           padded P x P canvas, plus per-block partial sums: ColorJitter's contrast blends with the mean of the WHOLE canvas.
//   pass 2 (augment_finish_kernel): remaining colour ops per tap, centre crop, renormalise, antialiased down-sampling (separable
//           triangle filter of support `scale`, the aten `_upsample_bilinear2d_aa` weights) straight to the stage resolution.
// Geometry follows torchvision's tensor path (functional.affine -> affine_grid + grid_sample(nearest, zeros, align_corners =
// False)): with centred pixel coordinates, src = R(-angle)^-1-style matrix [cos, sin; -sin, cos] applied to (dst - translate).
#include "ngan_common.h"

namespace {

constexpr int AUG_TILE = 2048;   // pixels per block of pass 1 (256 threads x 8)

struct AugParams {    // one per sample, device memory
    float cosv, sinv, tx, ty;        // rotation (cos, sin of the angle) and translation in pixels
    float brightness, contrast;      // factors (1 = unchanged)
    int flip, contrast_first;        // vertical flip; ColorJitter order
};

__device__ __forceinline__ float clamp01(float v) { return fminf(fmaxf(v, 0.f), 1.f); }

__global__ __launch_bounds__(256) void augment_affine_kernel(const float* __restrict__ src, const int* __restrict__ idx,
                                                             const AugParams* __restrict__ prm, float* __restrict__ canvas,
                                                             float* __restrict__ partial, int P, int nblk) {
    __shared__ float red[4];
    const int b = blockIdx.y, tid = threadIdx.x;
    const AugParams q = prm[b];
    const float* im = src + (long)idx[b] * P * P;
    float* out = canvas + (long)b * P * P;
    const float c0 = 0.5f * (float)(P - 1);
    float s = 0.f;
    for (int k = 0; k < AUG_TILE / 256; ++k) {
        const int e = blockIdx.x * AUG_TILE + k * 256 + tid;
        if (e < P * P) {
            const int yo = e / P, xo = e - yo * P;
            const int ys = q.flip ? P - 1 - yo : yo;                       // RandomVerticalFlip acts on the affine's output
            const float xd = (float)xo - c0 - q.tx, yd = (float)ys - c0 - q.ty;
            const float xs = q.cosv * xd + q.sinv * yd + c0, yv = -q.sinv * xd + q.cosv * yd + c0;
            const int xi = (int)nearbyintf(xs), yi = (int)nearbyintf(yv);   // grid_sample(nearest): round half to even
            float v = (xi >= 0 && xi < P && yi >= 0 && yi < P) ? im[(long)yi * P + xi] : 0.f;
            if (!q.contrast_first) v = clamp01(v * q.brightness);          // adjust_brightness = blend with black, clamped
            out[e] = v;
            s += v;
        }
    }
    s = group_sum<64>(s);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) partial[(long)b * nblk + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// triangle filter weight of input index j for output index i, down-scale factor `scale` >= 1 (aten UpSampleKernel aa weights)
__device__ __forceinline__ float aa_weight(int j, float center, float inv_scale) {
    const float t = fabsf(((float)j - center + 0.5f) * inv_scale);
    return t < 1.f ? 1.f - t : 0.f;
}

__global__ __launch_bounds__(256) void augment_finish_kernel(const float* __restrict__ canvas, const AugParams* __restrict__ prm,
                                                             const float* __restrict__ partial, float* __restrict__ out,
                                                             int P, int R, int S, int nblk) {
    __shared__ float mean_s;
    const int b = blockIdx.y, tid = threadIdx.x;
    const AugParams q = prm[b];
    if (tid < 64) {                                                        // fixed-order sum of pass 1's partials
        float s = 0.f;
        for (int i = tid; i < nblk; i += 64) s += partial[(long)b * nblk + i];
        s = group_sum<64>(s);
        if (tid == 0) mean_s = s / ((float)P * (float)P);
    }
    __syncthreads();
    const float mean = mean_s;
    const float* cv = canvas + (long)b * P * P;
    const int crop = (int)nearbyintf(0.5f * (float)(P - R));               // CenterCrop
    const float scale = (float)R / (float)S, inv_scale = 1.0f / scale;
    const int e = blockIdx.x * 256 + tid;
    if (e >= S * S) return;
    const int yo = e / S, xo = e - yo * S;
    // colour ops that still have to run: brightness first -> only contrast is left; contrast first -> contrast (with the mean of
    // the untouched canvas), then brightness
    auto colour = [&](float v) {
        v = clamp01(q.contrast * v + (1.0f - q.contrast) * mean);
        if (q.contrast_first) v = clamp01(v * q.brightness);
        return 2.0f * v - 1.0f;                                            // Renormalize((-1, 1), (0, 1))
    };
    if (S == R) {
        out[(long)b * S * S + e] = colour(cv[(long)(crop + yo) * P + crop + xo]);
        return;
    }
    const float cy = scale * ((float)yo + 0.5f), cx = scale * ((float)xo + 0.5f);
    const int y0 = max((int)(cy - scale + 0.5f), 0), y1 = min((int)(cy + scale + 0.5f), R);
    const int x0 = max((int)(cx - scale + 0.5f), 0), x1 = min((int)(cx + scale + 0.5f), R);
    float wys = 0.f, wxs = 0.f;
    for (int j = y0; j < y1; ++j) wys += aa_weight(j, cy, inv_scale);
    for (int j = x0; j < x1; ++j) wxs += aa_weight(j, cx, inv_scale);
    float acc = 0.f;
    for (int jy = y0; jy < y1; ++jy) {
        const float wy = aa_weight(jy, cy, inv_scale);
        const float* row = cv + (long)(crop + jy) * P + crop;
        float r = 0.f;
        for (int jx = x0; jx < x1; ++jx) r = fmaf(aa_weight(jx, cx, inv_scale), colour(row[jx]), r);
        acc = fmaf(wy, r, acc);
    }
    out[(long)b * S * S + e] = acc / (wys * wxs);
}

}  // namespace

extern "C" size_t ngan_augment_workspace_bytes(int B, int P) {
    if (B <= 0 || P <= 0) return 0;
    const long nblk = ((long)P * P + AUG_TILE - 1) / AUG_TILE;
    return (size_t)B * ((size_t)P * P + (size_t)nblk) * sizeof(float);
}

extern "C" int ngan_augment_batch(const float* src, const int* idx, const void* params, float* workspace, float* out,
                                  int N, int B, int P, int R, int S, void* stream) {
    NGAN_REQUIRE(src && idx && params && workspace && out, NGAN_ERR_ARG, "augment_batch: null pointer");
    NGAN_REQUIRE(N > 0 && B > 0 && B < 65536 && P > 0 && R > 0 && R <= P && S > 0 && S <= R && R % S == 0 && P <= 16384, NGAN_ERR_SHAPE,
                 "augment_batch: N=%d B=%d P=%d R=%d S=%d unsupported (need S | R <= P)", N, B, P, R, S);
    const int nblk = ngan::ceil_div((long)P * P, AUG_TILE);
    float* canvas = workspace;
    float* partial = workspace + (long)B * P * P;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(augment_affine_kernel, dim3(nblk, B), dim3(256), 0, s, src, idx, reinterpret_cast<const AugParams*>(params),
                       canvas, partial, P, nblk);
    hipLaunchKernelGGL(augment_finish_kernel, dim3(ngan::ceil_div((long)S * S, 256), B), dim3(256), 0, s, canvas,
                       reinterpret_cast<const AugParams*>(params), partial, out, P, R, S, nblk);
    return ngan::launch_status("ngan_augment_batch");
}
