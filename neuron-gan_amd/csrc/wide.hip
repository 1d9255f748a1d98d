// Channel counts the lane-group kernels of pixelnorm.hip / pointwise.hip do not take -- more than 256, or C / 4 not a power of two:
// the reference's constructors accept any widths and its presets 0004 - 0008 have 512- and 1024-channel blocks on 4x4 .. 32x32
// images (configs/config.py:87-98).  Same operators (formulas: include/ngan.h), one THREAD per pixel walking the channels for the
// per-pixel operators, one thread per channel walking the pixels for the parameter gradients.  A compatibility path for small
// images: no lane-group reductions, no workspaces, fixed summation order (bit-reproducible).
#include "ngan_common.h"

namespace {

__device__ __forceinline__ float lmask(float y, float slope) { return y > 0.f ? 1.f : slope; }

__global__ __launch_bounds__(256) void wide_pn_fwd_kernel(const float* __restrict__ c, const float* __restrict__ bias, float* __restrict__ y,
                                                          float* __restrict__ rn, long npix, int C, float slope, float eps) {
    const long pix = (long)blockIdx.x * 256 + threadIdx.x;
    if (pix >= npix) return;
    const float* src = c + pix * C;
    float ss = 0.f;
    for (int k = 0; k < C; k += 4) {
        float4 v = ld4(src + k);
        if (bias) v = f4add(v, ld4(bias + k));
        v.x = v.x > 0.f ? v.x : slope * v.x; v.y = v.y > 0.f ? v.y : slope * v.y;
        v.z = v.z > 0.f ? v.z : slope * v.z; v.w = v.w > 0.f ? v.w : slope * v.w;
        ss += f4dot(v, v);
    }
    const float r = sqrtf(ss / (float)C + eps), inv = 1.0f / r;
    float* dst = y + pix * C;
    for (int k = 0; k < C; k += 4) {                       // (src may be dst: element k is read before it is written)
        float4 v = ld4(src + k);
        if (bias) v = f4add(v, ld4(bias + k));
        v.x = v.x > 0.f ? v.x : slope * v.x; v.y = v.y > 0.f ? v.y : slope * v.y;
        v.z = v.z > 0.f ? v.z : slope * v.z; v.w = v.w > 0.f ? v.w : slope * v.w;
        st4(dst + k, f4scale(v, inv));
    }
    rn[pix] = r;
}

__global__ __launch_bounds__(256) void wide_pn_bwd_kernel(const float* __restrict__ gy, const float* __restrict__ gy2, const float* __restrict__ gr,
                                                          const float* __restrict__ y, const float* __restrict__ rn, float* __restrict__ gc,
                                                          long npix, int C, float slope) {
    const long pix = (long)blockIdx.x * 256 + threadIdx.x;
    if (pix >= npix) return;
    const long o = pix * C;
    float s = 0.f;
    for (int k = 0; k < C; k += 4) {
        float4 g = ld4(gy + o + k);
        if (gy2) g = f4add(g, ld4(gy2 + o + k));
        s += f4dot(g, ld4(y + o + k));
    }
    const float inv_c = 1.0f / (float)C, inv_r = 1.0f / rn[pix];
    s *= inv_c;
    const float kk = gr ? gr[pix] * inv_c : 0.f;
    for (int k = 0; k < C; k += 4) {                       // (gy may be gc)
        float4 g = ld4(gy + o + k);
        if (gy2) g = f4add(g, ld4(gy2 + o + k));
        const float4 yy = ld4(y + o + k);
        st4(gc + o + k, make_float4(((g.x - yy.x * s) * inv_r + kk * yy.x) * lmask(yy.x, slope), ((g.y - yy.y * s) * inv_r + kk * yy.y) * lmask(yy.y, slope),
                                    ((g.z - yy.z * s) * inv_r + kk * yy.z) * lmask(yy.z, slope), ((g.w - yy.w * s) * inv_r + kk * yy.w) * lmask(yy.w, slope)));
    }
}

__global__ __launch_bounds__(256) void wide_pn_bwdbwd_kernel(const float* __restrict__ h, const float* __restrict__ gy, const float* __restrict__ y,
                                                             const float* __restrict__ rn, float* __restrict__ ggy, float* __restrict__ gy_out,
                                                             float* __restrict__ gr_out, long npix, int C, float slope) {
    const long pix = (long)blockIdx.x * 256 + threadIdx.x;
    if (pix >= npix) return;
    const long o = pix * C;
    float s = 0.f, t = 0.f, u = 0.f;
    for (int k = 0; k < C; k += 4) {
        const float4 g = ld4(gy + o + k), yy = ld4(y + o + k);
        float4 hp = ld4(h + o + k);
        hp.x *= lmask(yy.x, slope); hp.y *= lmask(yy.y, slope); hp.z *= lmask(yy.z, slope); hp.w *= lmask(yy.w, slope);
        s += f4dot(g, yy); t += f4dot(hp, yy); u += f4dot(hp, g);
    }
    const float inv_c = 1.0f / (float)C, inv_r = 1.0f / rn[pix];
    s *= inv_c; t *= inv_c; u *= inv_c;
    for (int k = 0; k < C; k += 4) {
        const float4 g = ld4(gy + o + k), yy = ld4(y + o + k);
        float4 hp = ld4(h + o + k);
        hp.x *= lmask(yy.x, slope); hp.y *= lmask(yy.y, slope); hp.z *= lmask(yy.z, slope); hp.w *= lmask(yy.w, slope);
        st4(ggy + o + k, make_float4((hp.x - yy.x * t) * inv_r, (hp.y - yy.y * t) * inv_r, (hp.z - yy.z * t) * inv_r, (hp.w - yy.w * t) * inv_r));
        st4(gy_out + o + k, make_float4(-(s * hp.x + t * g.x) * inv_r, -(s * hp.y + t * g.y) * inv_r, -(s * hp.z + t * g.z) * inv_r, -(s * hp.w + t * g.w) * inv_r));
    }
    gr_out[pix] = -(float)C * (u - s * t) * inv_r * inv_r;
}

__global__ __launch_bounds__(256) void wide_channel_sum_kernel(const float* __restrict__ g, float* __restrict__ out, long npix, int C, float scale) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    float s = 0.f;
    for (long p = 0; p < npix; ++p) s += g[p * C + c];
    out[c] = s * scale;
}

__global__ __launch_bounds__(256) void wide_to_image_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ t,
                                                                long npix, int C, int Ncol) {
    const long pix = (long)blockIdx.x * 256 + threadIdx.x;
    if (pix >= npix) return;
    for (int k = 0; k < Ncol; ++k) {
        float s = 0.f;
        for (int c = 0; c < C; c += 4) s += f4dot(ld4(x + pix * C + c), ld4(w + k * C + c));
        t[pix * Ncol + k] = tanhf(s);
    }
}

// gx (optionally followed by the LeakyReLU -> PixelNorm backward of the layer that produced x: rn != nullptr)
__global__ __launch_bounds__(256) void wide_to_image_dx_kernel(const float* __restrict__ g, const float* __restrict__ t, const float* __restrict__ x,
                                                               const float* __restrict__ w, float* __restrict__ gx, long npix, int C, int Ncol,
                                                               const float* __restrict__ rn, float slope) {
    const long pix = (long)blockIdx.x * 256 + threadIdx.x;
    if (pix >= npix) return;
    float qv[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < Ncol; ++k) { const float tv = t[pix * Ncol + k]; qv[k] = g[pix * Ncol + k] * (1.0f - tv * tv); }
    float sdot = 0.f;
    if (rn) {
        for (int c = 0; c < C; c += 4) {
            float4 o = f4zero();
            for (int k = 0; k < Ncol; ++k) o = f4fma(ld4(w + k * C + c), qv[k], o);
            sdot += f4dot(o, ld4(x + pix * C + c));
        }
        sdot *= 1.0f / (float)C;
    }
    const float inv_r = rn ? 1.0f / rn[pix] : 1.f;
    for (int c = 0; c < C; c += 4) {
        float4 o = f4zero();
        for (int k = 0; k < Ncol; ++k) o = f4fma(ld4(w + k * C + c), qv[k], o);
        if (rn) {
            const float4 yy = ld4(x + pix * C + c);
            o = make_float4((o.x - yy.x * sdot) * inv_r * lmask(yy.x, slope), (o.y - yy.y * sdot) * inv_r * lmask(yy.y, slope),
                            (o.z - yy.z * sdot) * inv_r * lmask(yy.z, slope), (o.w - yy.w * sdot) * inv_r * lmask(yy.w, slope));
        }
        st4(gx + pix * C + c, o);
    }
}

__global__ __launch_bounds__(256) void wide_to_image_dw_kernel(const float* __restrict__ g, const float* __restrict__ t, const float* __restrict__ x,
                                                               float* __restrict__ gw, long npix, int C, int Ncol) {
    const int i = blockIdx.x * 256 + threadIdx.x;          // (k, c)
    if (i >= Ncol * C) return;
    const int k = i / C, c = i - k * C;
    float s = 0.f;
    for (long p = 0; p < npix; ++p) { const float tv = t[p * Ncol + k]; s = fmaf(x[p * C + c], g[p * Ncol + k] * (1.0f - tv * tv), s); }
    gw[i] = s;
}

__device__ __forceinline__ float wide_img(const float* __restrict__ x, int b, int yy, int xx, int k, int H, int W, int Ncol, int pool) {
    if (!pool) return x[(((long)b * H + yy) * W + xx) * Ncol + k];
    const long W2 = 2L * W;
    const float* p = x + (((long)b * 2 * H + 2 * yy) * W2 + 2 * xx) * Ncol + k;
    return 0.25f * ((p[0] + p[Ncol]) + (p[W2 * Ncol] + p[W2 * Ncol + Ncol]));
}

__global__ __launch_bounds__(256) void wide_from_image_dx_kernel(const float* __restrict__ g, const float* __restrict__ w, float* __restrict__ gx,
                                                                 int B, int H, int W, int Ncol, int C, int pool) {
    const long pix = (long)blockIdx.x * 256 + threadIdx.x;
    if (pix >= (long)B * H * W) return;
    const int xx = (int)(pix % W), yy = (int)((pix / W) % H), b = (int)(pix / ((long)W * H));
    for (int k = 0; k < Ncol; ++k) {
        float s = 0.f;
        for (int c = 0; c < C; ++c) s = fmaf(g[pix * C + c], w[c * Ncol + k], s);
        if (!pool) gx[pix * Ncol + k] = s;
        else {
            const long W2 = 2L * W;
            float* p = gx + (((long)b * 2 * H + 2 * yy) * W2 + 2 * xx) * Ncol + k;
            const float q4 = 0.25f * s;
            p[0] = q4; p[Ncol] = q4; p[W2 * Ncol] = q4; p[W2 * Ncol + Ncol] = q4;
        }
    }
}

__global__ __launch_bounds__(256) void wide_from_image_dw_kernel(const float* __restrict__ x, const float* __restrict__ g, float* __restrict__ gw,
                                                                 float* __restrict__ gb, int B, int H, int W, int Ncol, int C, int pool) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    float acc[4] = {0.f, 0.f, 0.f, 0.f}, sb = 0.f;
    for (int b = 0; b < B; ++b)
        for (int yy = 0; yy < H; ++yy)
            for (int xx = 0; xx < W; ++xx) {
                const float gv = g[(((long)b * H + yy) * W + xx) * C + c];
                sb += gv;
                for (int k = 0; k < Ncol; ++k) acc[k] = fmaf(gv, wide_img(x, b, yy, xx, k, H, W, Ncol, pool), acc[k]);
            }
    for (int k = 0; k < Ncol; ++k) gw[c * Ncol + k] = acc[k];
    if (gb) gb[c] = sb;
}

}  // namespace

namespace ngan {

// (called by the C ABI entry points of pixelnorm.hip / pointwise.hip when C is outside their lane-group kernels' range; C % 4 == 0)
int wide_pn_fwd(const float* c, const float* bias, float* y, float* rn, long npix, int C, float slope, float eps, hipStream_t s) {
    hipLaunchKernelGGL(wide_pn_fwd_kernel, dim3(ceil_div(npix, 256)), dim3(256), 0, s, c, bias, y, rn, npix, C, slope, eps);
    return launch_status("ngan_lrelu_pixelnorm_fwd(wide)");
}
int wide_pn_bwd(const float* gy, const float* gy2, const float* gr, const float* y, const float* rn, float* gc, long npix, int C, float slope, hipStream_t s) {
    hipLaunchKernelGGL(wide_pn_bwd_kernel, dim3(ceil_div(npix, 256)), dim3(256), 0, s, gy, gy2, gr, y, rn, gc, npix, C, slope);
    return launch_status("ngan_lrelu_pixelnorm_bwd(wide)");
}
int wide_pn_bwdbwd(const float* h, const float* gy, const float* y, const float* rn, float* ggy, float* gy_out, float* gr_out, long npix, int C,
                   float slope, hipStream_t s) {
    hipLaunchKernelGGL(wide_pn_bwdbwd_kernel, dim3(ceil_div(npix, 256)), dim3(256), 0, s, h, gy, y, rn, ggy, gy_out, gr_out, npix, C, slope);
    return launch_status("ngan_lrelu_pixelnorm_bwdbwd(wide)");
}
int wide_channel_sum(const float* g, float* out, long npix, int C, float scale, hipStream_t s) {
    hipLaunchKernelGGL(wide_channel_sum_kernel, dim3(ceil_div(C, 256)), dim3(256), 0, s, g, out, npix, C, scale);
    return launch_status("ngan_channel_sum(wide)");
}
int wide_to_image_fwd(const float* x, const float* w, float* t, long npix, int C, int Ncol, hipStream_t s) {
    hipLaunchKernelGGL(wide_to_image_fwd_kernel, dim3(ceil_div(npix, 256)), dim3(256), 0, s, x, w, t, npix, C, Ncol);
    return launch_status("ngan_to_image_fwd(wide)");
}
int wide_to_image_bwd(const float* g, const float* t, const float* x, const float* w, float* gx, float* gw, long npix, int C, int Ncol,
                      const float* rn, float slope, hipStream_t s) {
    hipLaunchKernelGGL(wide_to_image_dw_kernel, dim3(ceil_div((long)Ncol * C, 256)), dim3(256), 0, s, g, t, x, gw, npix, C, Ncol);
    hipLaunchKernelGGL(wide_to_image_dx_kernel, dim3(ceil_div(npix, 256)), dim3(256), 0, s, g, t, x, w, gx, npix, C, Ncol, rn, slope);
    return launch_status("ngan_to_image_bwd(wide)");
}
int wide_from_image_dx(const float* g, const float* w, float* gx, int B, int H, int W, int Ncol, int C, int pool, hipStream_t s) {
    hipLaunchKernelGGL(wide_from_image_dx_kernel, dim3(ceil_div((long)B * H * W, 256)), dim3(256), 0, s, g, w, gx, B, H, W, Ncol, C, pool);
    return launch_status("ngan_from_image_dx(wide)");
}
int wide_from_image_dw(const float* x, const float* g, float* gw, float* gb, int B, int H, int W, int Ncol, int C, int pool, hipStream_t s) {
    hipLaunchKernelGGL(wide_from_image_dw_kernel, dim3(ceil_div(C, 256)), dim3(256), 0, s, x, g, gw, gb, B, H, W, Ncol, C, pool);
    return launch_status("ngan_from_image_dw(wide)");
}

}  // namespace ngan
