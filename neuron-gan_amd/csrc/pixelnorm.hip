// LeakyReLU -> PixelNorm forward, backward and backward-of-backward on channels-last tensors (fp32, or bf16 storage: template T).
// Replaces the ~1700 decomposed ATen elementwise dispatches per step that /root/reference/models.py:118,126
// (PixelNorm) and models.py:263 (LeakyReLU) issue, including their first- and second-order autograd
// (SURVEY.md 2.1, Appendix C).  HBM-bound: one pixel's C channels are C/4 consecutive lanes holding a float4
// each; the per-pixel channel reductions are wavefront butterfly shuffles, nothing goes through LDS.
#include "ngan_common.h"

namespace {

__device__ __forceinline__ float lrelu_mask(float y, float slope) { return y > 0.f ? 1.f : slope; }

// V float4s (4 V consecutive channels) per lane: 1, or 2 for bf16 storage (16-byte accesses; ngan_common.h)
template <typename T, int LPP, int V>
__global__ __launch_bounds__(256) void pn_fwd_kernel(const T* __restrict__ c, const float* __restrict__ bias,
                                                     T* __restrict__ y, float* __restrict__ rn, long npix, int C,
                                                     float slope, float eps) {
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long pix = gid / LPP;
    const int sub = (int)(gid % LPP);
    const bool ok = pix < npix;
    float4 v[V];
    if (ok) ldav<T, V>(c + pix * C + sub * 4 * V, v);
    float d = 0.f;
#pragma unroll
    for (int i = 0; i < V; ++i) {
        if (!ok) v[i] = f4zero();
        if (bias) v[i] = f4add(v[i], ld4(bias + (sub * V + i) * 4));
        v[i].x = v[i].x > 0.f ? v[i].x : slope * v[i].x; v[i].y = v[i].y > 0.f ? v[i].y : slope * v[i].y;
        v[i].z = v[i].z > 0.f ? v[i].z : slope * v[i].z; v[i].w = v[i].w > 0.f ? v[i].w : slope * v[i].w;
        d += f4dot(v[i], v[i]);
    }
    const float ss = group_sum<LPP>(d);
    const float r = sqrtf(ss / (float)C + eps);
    if (ok) {
#pragma unroll
        for (int i = 0; i < V; ++i) v[i] = f4scale(v[i], 1.0f / r);
        stav<T, V>(y + pix * C + sub * 4 * V, v);
        if (sub == 0) rn[pix] = r;
    }
}

template <typename T, int LPP, int V>
__global__ __launch_bounds__(256) void pn_bwd_kernel(const T* __restrict__ gy, const float* __restrict__ gr,
                                                     const T* __restrict__ y, const float* __restrict__ rn,
                                                     T* __restrict__ gc, long npix, int C, float slope,
                                                     const T* __restrict__ gy2) {
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long pix = gid / LPP;
    const int sub = (int)(gid % LPP);
    const bool ok = pix < npix;
    const long off = pix * C + sub * 4 * V;
    float4 g[V], g2[V], yy[V];
    if (ok) {
        ldav<T, V>(gy + off, g);
        ldav<T, V>(y + off, yy);
        if (gy2) ldav<T, V>(gy2 + off, g2);      // a second contribution to the same gradient, summed here
    }
    float d = 0.f;
#pragma unroll
    for (int i = 0; i < V; ++i) {
        if (!ok) { g[i] = f4zero(); yy[i] = f4zero(); }
        if (gy2 && ok) g[i] = f4add(g[i], g2[i]);
        d += f4dot(g[i], yy[i]);
    }
    const float r = ok ? rn[pix] : 1.f;
    const float inv_c = 1.0f / (float)C;
    const float s = group_sum<LPP>(d) * inv_c;
    const float inv_r = 1.0f / r;
    const float k = (gr && ok) ? gr[pix] * inv_c : 0.f;
    float4 o[V];
#pragma unroll
    for (int i = 0; i < V; ++i) {
        o[i].x = ((g[i].x - yy[i].x * s) * inv_r + k * yy[i].x) * lrelu_mask(yy[i].x, slope);
        o[i].y = ((g[i].y - yy[i].y * s) * inv_r + k * yy[i].y) * lrelu_mask(yy[i].y, slope);
        o[i].z = ((g[i].z - yy[i].z * s) * inv_r + k * yy[i].z) * lrelu_mask(yy[i].z, slope);
        o[i].w = ((g[i].w - yy[i].w * s) * inv_r + k * yy[i].w) * lrelu_mask(yy[i].w, slope);
    }
    if (ok) stav<T, V>(gc + off, o);
}

template <typename T, int LPP, int V>
__global__ __launch_bounds__(256) void pn_bwdbwd_kernel(const T* __restrict__ h, const T* __restrict__ gy,
                                                        const T* __restrict__ y, const float* __restrict__ rn,
                                                        T* __restrict__ ggy, T* __restrict__ gy_out,
                                                        float* __restrict__ gr_out, long npix, int C, float slope) {
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long pix = gid / LPP;
    const int sub = (int)(gid % LPP);
    const bool ok = pix < npix;
    const long off = pix * C + sub * 4 * V;
    float4 hp[V], g[V], yy[V];
    if (ok) {
        ldav<T, V>(h + off, hp);
        ldav<T, V>(gy + off, g);
        ldav<T, V>(y + off, yy);
    }
    float ds = 0.f, dt = 0.f, du = 0.f;
#pragma unroll
    for (int i = 0; i < V; ++i) {
        if (!ok) { hp[i] = f4zero(); g[i] = f4zero(); yy[i] = f4zero(); }
        hp[i].x *= lrelu_mask(yy[i].x, slope); hp[i].y *= lrelu_mask(yy[i].y, slope);
        hp[i].z *= lrelu_mask(yy[i].z, slope); hp[i].w *= lrelu_mask(yy[i].w, slope);
        ds += f4dot(g[i], yy[i]); dt += f4dot(hp[i], yy[i]); du += f4dot(hp[i], g[i]);
    }
    const float r = ok ? rn[pix] : 1.f;
    const float inv_c = 1.0f / (float)C;
    const float s = group_sum<LPP>(ds) * inv_c;
    const float t = group_sum<LPP>(dt) * inv_c;
    const float u = group_sum<LPP>(du) * inv_c;
    const float inv_r = 1.0f / r;
    if (ok) {
        float4 a[V], bq[V];
#pragma unroll
        for (int i = 0; i < V; ++i) {
            a[i].x = (hp[i].x - yy[i].x * t) * inv_r; a[i].y = (hp[i].y - yy[i].y * t) * inv_r;
            a[i].z = (hp[i].z - yy[i].z * t) * inv_r; a[i].w = (hp[i].w - yy[i].w * t) * inv_r;
            bq[i].x = -(s * hp[i].x + t * g[i].x) * inv_r; bq[i].y = -(s * hp[i].y + t * g[i].y) * inv_r;
            bq[i].z = -(s * hp[i].z + t * g[i].z) * inv_r; bq[i].w = -(s * hp[i].w + t * g[i].w) * inv_r;
        }
        stav<T, V>(ggy + off, a);
        stav<T, V>(gy_out + off, bq);
        if (sub == 0) gr_out[pix] = -(float)C * (u - s * t) * inv_r * inv_r;
    }
}

bool lpp_ok(int C) {
    if (C <= 0 || C % 4) return false;
    const int l = C / 4;
    return l <= 64 && (l & (l - 1)) == 0;
}

// lanes per pixel = C / (4 V);  V = 2 (16-byte accesses) for bf16 storage whenever C is a multiple of 8
#define PN_LAUNCH(KERNEL, VV, ...)                                                                                \
    do {                                                                                                          \
        const int lpp = C / (4 * VV);                                                                             \
        const dim3 grid(ngan::ceil_div(npix * lpp, 256)), block(256);                                              \
        hipStream_t s_ = (hipStream_t)stream;                                                                     \
        switch (lpp) {                                                                                            \
            case 1: hipLaunchKernelGGL((KERNEL<T, 1, VV>), grid, block, 0, s_, __VA_ARGS__); break;                 \
            case 2: hipLaunchKernelGGL((KERNEL<T, 2, VV>), grid, block, 0, s_, __VA_ARGS__); break;                 \
            case 4: hipLaunchKernelGGL((KERNEL<T, 4, VV>), grid, block, 0, s_, __VA_ARGS__); break;                 \
            case 8: hipLaunchKernelGGL((KERNEL<T, 8, VV>), grid, block, 0, s_, __VA_ARGS__); break;                 \
            case 16: hipLaunchKernelGGL((KERNEL<T, 16, VV>), grid, block, 0, s_, __VA_ARGS__); break;               \
            case 32: hipLaunchKernelGGL((KERNEL<T, 32, VV>), grid, block, 0, s_, __VA_ARGS__); break;               \
            default: hipLaunchKernelGGL((KERNEL<T, 64, VV>), grid, block, 0, s_, __VA_ARGS__); break;               \
        }                                                                                                         \
    } while (0)
#define PN_DISPATCH(KERNEL, ...)                                                                                  \
    do {                                                                                                          \
        if constexpr (sizeof(T) == 2) {                                                                           \
            if (C % 8 == 0) { PN_LAUNCH(KERNEL, 2, __VA_ARGS__); break; }                                         \
        }                                                                                                         \
        PN_LAUNCH(KERNEL, 1, __VA_ARGS__);                                                                        \
    } while (0)

// T = float: the fp32 contract of include/ngan.h, with csrc/wide.hip behind it for channel counts outside the lane-group kernels'
// range; T = __bf16: the bf16-storage entry points (no wide path)
template <typename T>
int pn_fwd_impl(const T* c, const float* bias, T* y, float* rnorm, long npix, int C, float slope, float eps, void* stream) {
    NGAN_REQUIRE(c && y && rnorm, NGAN_ERR_ARG, "lrelu_pixelnorm_fwd: null pointer");
    if constexpr (sizeof(T) == 4)
        if (npix > 0 && C > 0 && C % 4 == 0 && !lpp_ok(C)) return ngan::wide_pn_fwd(c, bias, y, rnorm, npix, C, slope, eps, (hipStream_t)stream);
    NGAN_REQUIRE(npix > 0 && lpp_ok(C), NGAN_ERR_SHAPE, "lrelu_pixelnorm_fwd: npix=%ld C=%d unsupported", npix, C);
    PN_DISPATCH(pn_fwd_kernel, c, bias, y, rnorm, npix, C, slope, eps);
    return ngan::launch_status("ngan_lrelu_pixelnorm_fwd");
}

template <typename T>
int pn_bwd2_impl(const T* gy, const T* gy2, const float* gr, const T* y, const float* rnorm, T* gc, long npix, int C, float slope, void* stream) {
    NGAN_REQUIRE(gy && y && rnorm && gc, NGAN_ERR_ARG, "lrelu_pixelnorm_bwd: null pointer");
    if constexpr (sizeof(T) == 4)
        if (npix > 0 && C > 0 && C % 4 == 0 && !lpp_ok(C)) return ngan::wide_pn_bwd(gy, gy2, gr, y, rnorm, gc, npix, C, slope, (hipStream_t)stream);
    NGAN_REQUIRE(npix > 0 && lpp_ok(C), NGAN_ERR_SHAPE, "lrelu_pixelnorm_bwd: npix=%ld C=%d unsupported", npix, C);
    PN_DISPATCH(pn_bwd_kernel, gy, gr, y, rnorm, gc, npix, C, slope, gy2);
    return ngan::launch_status("ngan_lrelu_pixelnorm_bwd");
}

template <typename T>
int pn_bwdbwd_impl(const T* h, const T* gy, const T* y, const float* rnorm, T* ggy, T* gy_out, float* gr_out, long npix, int C, float slope,
                   void* stream) {
    NGAN_REQUIRE(h && gy && y && rnorm && ggy && gy_out && gr_out, NGAN_ERR_ARG, "lrelu_pixelnorm_bwdbwd: null pointer");
    if constexpr (sizeof(T) == 4)
        if (npix > 0 && C > 0 && C % 4 == 0 && !lpp_ok(C))
            return ngan::wide_pn_bwdbwd(h, gy, y, rnorm, ggy, gy_out, gr_out, npix, C, slope, (hipStream_t)stream);
    NGAN_REQUIRE(npix > 0 && lpp_ok(C), NGAN_ERR_SHAPE, "lrelu_pixelnorm_bwdbwd: npix=%ld C=%d unsupported", npix, C);
    PN_DISPATCH(pn_bwdbwd_kernel, h, gy, y, rnorm, ggy, gy_out, gr_out, npix, C, slope);
    return ngan::launch_status("ngan_lrelu_pixelnorm_bwdbwd");
}

}  // namespace

#define BF(p) reinterpret_cast<const __bf16*>(p)
#define BFM(p) reinterpret_cast<__bf16*>(p)

extern "C" int ngan_lrelu_pixelnorm_fwd(const float* c, const float* bias, float* y, float* rnorm, long npix, int C,
                                        float slope, float eps, void* stream) {
    return pn_fwd_impl<float>(c, bias, y, rnorm, npix, C, slope, eps, stream);
}
extern "C" int ngan_bf16_lrelu_pixelnorm_fwd(const ngan_bf16* c, const float* bias, ngan_bf16* y, float* rnorm, long npix, int C,
                                             float slope, float eps, void* stream) {
    return pn_fwd_impl<__bf16>(BF(c), bias, BFM(y), rnorm, npix, C, slope, eps, stream);
}

extern "C" int ngan_lrelu_pixelnorm_bwd2(const float* gy, const float* gy2, const float* gr, const float* y, const float* rnorm,
                                         float* gc, long npix, int C, float slope, void* stream) {
    return pn_bwd2_impl<float>(gy, gy2, gr, y, rnorm, gc, npix, C, slope, stream);
}
extern "C" int ngan_bf16_lrelu_pixelnorm_bwd2(const ngan_bf16* gy, const ngan_bf16* gy2, const float* gr, const ngan_bf16* y, const float* rnorm,
                                              ngan_bf16* gc, long npix, int C, float slope, void* stream) {
    return pn_bwd2_impl<__bf16>(BF(gy), BF(gy2), gr, BF(y), rnorm, BFM(gc), npix, C, slope, stream);
}

extern "C" int ngan_lrelu_pixelnorm_bwd(const float* gy, const float* gr, const float* y, const float* rnorm, float* gc,
                                        long npix, int C, float slope, void* stream) {
    return ngan_lrelu_pixelnorm_bwd2(gy, nullptr, gr, y, rnorm, gc, npix, C, slope, stream);
}
extern "C" int ngan_bf16_lrelu_pixelnorm_bwd(const ngan_bf16* gy, const float* gr, const ngan_bf16* y, const float* rnorm, ngan_bf16* gc,
                                             long npix, int C, float slope, void* stream) {
    return ngan_bf16_lrelu_pixelnorm_bwd2(gy, nullptr, gr, y, rnorm, gc, npix, C, slope, stream);
}

extern "C" int ngan_lrelu_pixelnorm_bwdbwd(const float* h, const float* gy, const float* y, const float* rnorm,
                                           float* ggy, float* gy_out, float* gr_out, long npix, int C, float slope,
                                           void* stream) {
    return pn_bwdbwd_impl<float>(h, gy, y, rnorm, ggy, gy_out, gr_out, npix, C, slope, stream);
}
extern "C" int ngan_bf16_lrelu_pixelnorm_bwdbwd(const ngan_bf16* h, const ngan_bf16* gy, const ngan_bf16* y, const float* rnorm,
                                                ngan_bf16* ggy, ngan_bf16* gy_out, float* gr_out, long npix, int C, float slope, void* stream) {
    return pn_bwdbwd_impl<__bf16>(BF(h), BF(gy), BF(y), rnorm, BFM(ggy), BFM(gy_out), gr_out, npix, C, slope, stream);
}
