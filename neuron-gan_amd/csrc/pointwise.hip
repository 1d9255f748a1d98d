// HBM-bound helpers around the conv stack: 1x1 colour projections (FromImage / ToImage), standalone
// resampling and its adjoint, fade-in arithmetic, gradient-penalty norms, channel sums.
// Reference call sites: /root/reference/models.py:141-149 (ToImage), 161-165 (FromImage), 87-89 (Interpolate),
// 254 (AvgPool2d), 350 / 521 (fade-in); loss_functions.py:171, 176 (x_hat, per-sample gradient norm).
#include "ngan_common.h"

namespace ngan {

// out[i] = scale * sum_j partials[j*stride + i], i < M (outputs i >= M1 go to out2[i - M1]).  One wave per output: lane l adds
// parts l, l + 64, ... (independent loads, all in flight), then a butterfly over the wave: fixed order, deterministic.
__global__ __launch_bounds__(64) void reduce_partials_kernel(const float* __restrict__ partials, int nparts, int M,
                                                             long stride, float* __restrict__ out, int M1,
                                                             float* __restrict__ out2, float scale, int accumulate = 0) {
    const int i = blockIdx.x, lane = threadIdx.x;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int j = lane;
    for (; j + 192 < nparts; j += 256) {
        s0 += partials[(long)j * stride + i];
        s1 += partials[(long)(j + 64) * stride + i];
        s2 += partials[(long)(j + 128) * stride + i];
        s3 += partials[(long)(j + 192) * stride + i];
    }
    for (; j < nparts; j += 64) s0 += partials[(long)j * stride + i];
    const float s = group_sum<64>((s0 + s1) + (s2 + s3));
    if (lane == 0) {            // accumulate: bit 0 out += , bit 1 out2 +=
        if (i < M1) out[i] = (accumulate & 1) ? out[i] + s * scale : s * scale;
        else out2[i - M1] = (accumulate & 2) ? out2[i - M1] + s * scale : s * scale;
    }
}

int reduce_partials_strided(const float* partials, int nparts, int M, long stride, float* out, float scale, hipStream_t s) {
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(M), dim3(64), 0, s, partials, nparts, M, stride, out, M, (float*)nullptr, scale);
    return launch_status("reduce_partials");
}

int reduce_partials_split(const float* partials, int nparts, int M, long stride, float* out, int M1, float* out2, float scale,
                          hipStream_t s) {
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(M), dim3(64), 0, s, partials, nparts, M, stride, out, M1, out2, scale);
    return launch_status("reduce_partials");
}

int reduce_partials_acc(const float* partials, int nparts, int M, long stride, float* out, int M1, float* out2, float scale, int accumulate,
                        hipStream_t s) {
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(M), dim3(64), 0, s, partials, nparts, M, stride, out, M1, out2, scale, accumulate);
    return launch_status("reduce_partials");
}

int reduce_partials(const float* partials, int nparts, int M, float* out, float scale, hipStream_t s) {
    return reduce_partials_strided(partials, nparts, M, M, out, scale, s);
}


}  // namespace ngan

namespace {

using ngan::ceil_div;

__device__ __forceinline__ float4 pn_bwd4_pw(float4 g, float4 yy, float s, float inv_r, float slope) {
    return make_float4((g.x - yy.x * s) * inv_r * (yy.x > 0.f ? 1.f : slope), (g.y - yy.y * s) * inv_r * (yy.y > 0.f ? 1.f : slope),
                       (g.z - yy.z * s) * inv_r * (yy.z > 0.f ? 1.f : slope), (g.w - yy.w * s) * inv_r * (yy.w > 0.f ? 1.f : slope));
}
constexpr int MAX_PARTS = 1024;   // callers size their slab workspaces for 1024 parts

bool pow2_quads(int C) {
    if (C <= 0 || C % 4) return false;
    const int q = C / 4;
    return q <= 64 && (q & (q - 1)) == 0;
}

int stream_blocks(long npix, int Q) {
    long need = (npix * Q + 255) / 256;
    return (int)(need < MAX_PARTS ? need : MAX_PARTS);
}

// block-level column reduction: every thread holds a float4 for channel-quad (tid % Q); threads with the same
// quad are summed; result for quad k is returned to thread k (k < Q).
template <int Q>
__device__ __forceinline__ float4 block_quad_sum(float4 v, float4* red) {
    const int tid = threadIdx.x;
    __syncthreads();
    red[tid] = v;
    __syncthreads();
    float4 s = f4zero();
    if (tid < Q)
        for (int j = tid; j < 256; j += Q) s = f4add(s, red[j]);
    return s;
}

// ------------------------------------------------------------------------------------------------------------
// channel sums (bias gradient)
// ------------------------------------------------------------------------------------------------------------
template <typename T, int Q>
__global__ __launch_bounds__(256) void channel_sum_kernel(const T* __restrict__ g, float* __restrict__ partial,
                                                          long npix, int C) {
    __shared__ float4 red[256];
    const int tid = threadIdx.x, sub = tid % Q;
    const long stride = (long)gridDim.x * (256 / Q);
    float4 acc = f4zero();
    for (long pix = (long)blockIdx.x * (256 / Q) + tid / Q; pix < npix; pix += stride) acc = f4add(acc, lda4(g + pix * C + sub * 4));
    float4 s = block_quad_sum<Q>(acc, red);
    if (tid < Q) st4(partial + (long)blockIdx.x * C + tid * 4, s);
}

// ------------------------------------------------------------------------------------------------------------
// FromImage
// ------------------------------------------------------------------------------------------------------------
template <int POOL>
__device__ __forceinline__ float load_img(const float* __restrict__ x, int b, int yy, int xx, int k, int H, int W, int Ncol) {
    if (POOL == 0) return x[(((long)b * H + yy) * W + xx) * Ncol + k];
    const long W2 = 2L * W;
    const float* p = x + (((long)b * 2 * H + 2 * yy) * W2 + 2 * xx) * Ncol + k;
    return 0.25f * ((p[0] + p[Ncol]) + (p[W2 * Ncol] + p[W2 * Ncol + Ncol]));
}

// one thread per (pixel, channel quad); blockIdx.y = image row (b*H + y), so no per-item division by W or H
// (V float4s per thread: 2 for bf16 storage, i.e. 16-byte stores)
template <typename T, int POOL, int V>
__global__ __launch_bounds__(256) void from_image_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ bias, T* __restrict__ y,
                                                             int B, int H, int W, int Ncol, int C) {
    const unsigned Q = C / (4 * V);
    const unsigned it = blockIdx.x * 256 + threadIdx.x;        // item inside the row: xx * Q + channel group
    if (it >= (unsigned)W * Q) return;
    const int xx = (int)(it / Q);
    const int c0 = (int)(it - (unsigned)xx * Q) * 4 * V;
    const int row = blockIdx.y;
    const int b = row / H, yy = row - b * H;
    float4 o[V];
#pragma unroll
    for (int i = 0; i < V; ++i) o[i] = bias ? ld4(bias + c0 + 4 * i) : f4zero();
    for (int k = 0; k < Ncol; ++k) {
        const float v = load_img<POOL>(x, b, yy, xx, k, H, W, Ncol);
#pragma unroll
        for (int i = 0; i < V; ++i) {
            const int c = c0 + 4 * i;
            o[i].x = fmaf(w[(c + 0) * Ncol + k], v, o[i].x); o[i].y = fmaf(w[(c + 1) * Ncol + k], v, o[i].y);
            o[i].z = fmaf(w[(c + 2) * Ncol + k], v, o[i].z); o[i].w = fmaf(w[(c + 3) * Ncol + k], v, o[i].w);
        }
    }
    stav<T, V>(y + ((long)row * W + xx) * C + c0, o);
}

template <typename T, int Q, int POOL>
__global__ __launch_bounds__(256) void from_image_dx_kernel(const T* __restrict__ g, const float* __restrict__ w,
                                                            float* __restrict__ gx, int B, int H, int W, int Ncol, int C) {
    const unsigned npix = (unsigned)B * H * W;            // host checks that npix * Q fits 31 bits
    const unsigned gid = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned pix = gid / Q;
    const int sub = (int)(gid % Q);
    const bool ok = pix < npix;
    const float4 gv = ok ? lda4(g + (long)pix * C + sub * 4) : f4zero();
    const unsigned row = pix / (unsigned)W;
    const int xx = (int)(pix - row * W);
    const int b = (int)(row / (unsigned)H);
    const int yy = (int)(row - (unsigned)b * H);
    for (int k = 0; k < Ncol; ++k) {
        const int c0 = sub * 4;
        float part = gv.x * w[(c0 + 0) * Ncol + k] + gv.y * w[(c0 + 1) * Ncol + k] + gv.z * w[(c0 + 2) * Ncol + k] +
                     gv.w * w[(c0 + 3) * Ncol + k];
        const float s = group_sum<Q>(part);
        if (ok && sub == 0) {
            if (POOL == 0) {
                gx[(long)pix * Ncol + k] = s;
            } else {
                const long W2 = 2L * W;
                float* p = gx + (((long)b * 2 * H + 2 * yy) * W2 + 2 * xx) * Ncol + k;
                const float q4 = 0.25f * s;
                p[0] = q4; p[Ncol] = q4; p[W2 * Ncol] = q4; p[W2 * Ncol + Ncol] = q4;
            }
        }
    }
}

// partial slab per block: [c*Ncol + k] for k < Ncol, then [C*Ncol + c] for the bias sums
template <typename T, int Q, int POOL>
__global__ __launch_bounds__(256) void from_image_dw_kernel(const float* __restrict__ x, const T* __restrict__ g,
                                                            float* __restrict__ partial, int B, int H, int W, int Ncol, int C,
                                                            int rows_per_block) {
    __shared__ float4 red[256];
    const int tid = threadIdx.x, sub = tid % Q;
    // block = ROWS_PER_BLOCK consecutive image rows (b*H + y); a thread owns a fixed (column phase, channel quad) and walks the
    // columns of each row with stride 256/Q: no division by W or H per item, four pixels in flight per iteration
    const int rows = B * H;
    const int r0 = blockIdx.x * rows_per_block, r1 = min(r0 + rows_per_block, rows);
    constexpr int PPB = 256 / Q;                          // pixels per block-iteration
    float4 acc[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) acc[k] = f4zero();
    for (int row = r0; row < r1; ++row) {
        const int b = row / H, yy = row - b * H;
        const T* grow = g + (long)row * W * C + sub * 4;
        for (int x0 = tid / Q; x0 < W; x0 += 4 * PPB) {
            float4 gv[4];
            float xv[4][4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int xu = x0 + u * PPB;
                const int xx = xu < W ? xu : W - 1;
                gv[u] = lda4(grow + (long)xx * C);
#pragma unroll
                for (int k = 0; k < 4; ++k) xv[u][k] = k < Ncol ? load_img<POOL>(x, b, yy, xx, k, H, W, Ncol) : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (x0 + u * PPB >= W) gv[u] = f4zero();
                acc[4] = f4add(acc[4], gv[u]);
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (k < Ncol) acc[k] = f4fma(gv[u], xv[u][k], acc[k]);
            }
        }
    }
    float* slab = partial + (long)blockIdx.x * C * (Ncol + 1);
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        if (k < Ncol || k == 4) {
            float4 s = block_quad_sum<Q>(acc[k], red);
            if (tid < Q) {
                const int c0 = tid * 4;
                if (k == 4) st4(slab + C * Ncol + c0, s);
                else { slab[(c0 + 0) * Ncol + k] = s.x; slab[(c0 + 1) * Ncol + k] = s.y; slab[(c0 + 2) * Ncol + k] = s.z; slab[(c0 + 3) * Ncol + k] = s.w; }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// ToImage
// ------------------------------------------------------------------------------------------------------------
template <typename T, int Q>
__global__ __launch_bounds__(256) void to_image_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                           float* __restrict__ t, long npix, int C, int Ncol) {
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long pix = gid / Q;
    const int sub = (int)(gid % Q);
    const bool ok = pix < npix;
    const float4 xv = ok ? lda4(x + pix * C + sub * 4) : f4zero();
    for (int k = 0; k < Ncol; ++k) {
        const float s = group_sum<Q>(f4dot(xv, ld4(w + k * C + sub * 4)));
        if (ok && sub == 0) t[pix * Ncol + k] = tanhf(s);
    }
}

// rn != nullptr: x is the output of a LeakyReLU -> PixelNorm with norms rn, and gx receives the gradient w.r.t. that operator's
// INPUT (its backward is applied to the ToImage input-gradient before the store: one pass instead of two over the activation)
// Q lanes per pixel, V float4s (4 V channels) per lane: V = 2 for bf16 storage (16-byte accesses), and two pixels per loop trip, so that
// a wave has 2 KB instead of 512 B of the activation in flight (the bf16 form of the V = 1, one-pixel loop ran at 2.5 TB/s)
// NC: the number of image channels when it is known at compile time (1: the grey-scale images of the reference's dataset), 0 = run time
// (<= 4).  With a run-time count every `k < Ncol` is a branch around a load + wait, and the compiler serialised the small loads of g and t
// in front of the activation's load: the round-4 templated kernel ran 134 - 140 us where round 3's took 113 (profiles/r04_kernel_stats_f32*).
template <typename T, int Q, int V, int NC>
__global__ __launch_bounds__(256) void to_image_bwd_kernel(const float* __restrict__ g, const float* __restrict__ t,
                                                           const T* __restrict__ x, const float* __restrict__ w,
                                                           T* __restrict__ gx, float* __restrict__ partial,
                                                           long npix, int C, int ncol_rt, const float* __restrict__ rn, float slope) {
    __shared__ float4 red[256];
    const int Ncol = NC ? NC : ncol_rt;
    constexpr int U = V;                                  // pixels per loop trip
    const int tid = threadIdx.x, sub = tid % Q;
    const long stride = (long)gridDim.x * (256 / Q);
    float4 acc[4][V];
    float4 wv[4][V];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int i = 0; i < V; ++i) {
            acc[k][i] = f4zero();
            wv[k][i] = k < Ncol ? ld4(w + k * C + (sub * V + i) * 4) : f4zero();
        }
    // (lanes of one pixel always iterate together, which is all the group shuffle below needs)
    for (long pix0 = (long)blockIdx.x * (256 / Q) + tid / Q; pix0 < npix; pix0 += U * stride) {
        float4 xv[U][V];
        float qv[U][4];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long pix = pix0 + u * stride;
            if (U == 1 || pix < npix) {                 // (U = 1: the loop condition already says so -- the fp32 form stays branch-free)
                ldav<T, V>(x + pix * C + sub * 4 * V, xv[u]);
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (k < Ncol) {
                        const float tv = t[pix * Ncol + k];
                        qv[u][k] = g[pix * Ncol + k] * (1.0f - tv * tv);
                    }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long pix = pix0 + u * stride;
            const bool ok = U == 1 || pix < npix;   // (uniform over the Q lanes of a pixel)
            float4 o[V];
            float d = 0.f;
#pragma unroll
            for (int i = 0; i < V; ++i) {
                o[i] = f4zero();
                if (!ok) xv[u][i] = f4zero();
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (k < Ncol && ok) {
                        o[i] = f4fma(wv[k][i], qv[u][k], o[i]);
                        acc[k][i] = f4fma(xv[u][i], qv[u][k], acc[k][i]);
                    }
                d += f4dot(o[i], xv[u][i]);
            }
            if (rn) {
                const float sdot = group_sum<Q>(d) * (1.0f / (float)C);
                const float inv_r = ok ? 1.0f / rn[pix] : 0.f;
#pragma unroll
                for (int i = 0; i < V; ++i) o[i] = pn_bwd4_pw(o[i], xv[u][i], sdot, inv_r, slope);
            }
            if (ok) stav<T, V>(gx + pix * C + sub * 4 * V, o);
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (k < Ncol) {
#pragma unroll
            for (int i = 0; i < V; ++i) {
                float4 s = block_quad_sum<Q>(acc[k][i], red);
                if (tid < Q) st4(partial + (long)blockIdx.x * C * Ncol + k * C + (tid * V + i) * 4, s);
            }
        }
}

// ------------------------------------------------------------------------------------------------------------
// resampling (scalar per element; used on colour images and on feature maps outside fused convs)
// ------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ void up2_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int B, int h, int w, int C) {
    const long total = (long)B * 4 * h * w * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        long r = i / C;
        const int X = (int)(r % (2 * w)); r /= (2 * w);
        const int Y = (int)(r % (2 * h));
        const int b = (int)(r / (2 * h));
        int y0, y1, x0, x1; float wy0, wy1, wx0, wx1;
        up2_taps(Y, h, y0, y1, wy0, wy1);
        up2_taps(X, w, x0, x1, wx0, wx1);
        const T* r0 = x + ((long)b * h + y0) * w * C + c;
        const T* r1 = x + ((long)b * h + y1) * w * C + c;
        const float top = fmaf(lda1(r0 + (long)x1 * C), wx1, lda1(r0 + (long)x0 * C) * wx0);
        const float bot = fmaf(lda1(r1 + (long)x1 * C), wx1, lda1(r1 + (long)x0 * C) * wx0);
        sta1(y + i, fmaf(bot, wy1, top * wy0));
    }
}

// weight with which low-res index i receives from high-res index R (R in 2i-1 .. 2i+2), n = low-res extent
__device__ __forceinline__ float up2_adj_w(int i, int R, int n) {
    if (R < 0 || R > 2 * n - 1) return 0.f;
    const int d = R - 2 * i;
    if (d == -1 || d == 2) return 0.25f;
    if (d == 0) return i == 0 ? 1.0f : 0.75f;
    return i == n - 1 ? 1.0f : 0.75f;  // d == 1
}

template <typename T>
__global__ void up2_adjoint_kernel(const T* __restrict__ gy, T* __restrict__ gx, int B, int h, int w, int C) {
    const long total = (long)B * h * w * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        long r = i / C;
        const int X = (int)(r % w); r /= w;
        const int Y = (int)(r % h);
        const int b = (int)(r / h);
        float s = 0.f;
        for (int dy = -1; dy <= 2; ++dy) {
            const int RY = 2 * Y + dy;
            const float wy = up2_adj_w(Y, RY, h);
            if (wy == 0.f) continue;
            for (int dx = -1; dx <= 2; ++dx) {
                const int RX = 2 * X + dx;
                const float wx = up2_adj_w(X, RX, w);
                if (wx == 0.f) continue;
                s = fmaf(wy * wx, lda1(gy + (((long)b * 2 * h + RY) * (2 * w) + RX) * C + c), s);
            }
        }
        sta1(gx + i, s);
    }
}

// float4 version (C % 4 == 0): one thread per (low-res pixel, channel quad)
template <typename T>
__global__ __launch_bounds__(256) void up2_adjoint_vec_kernel(const T* __restrict__ gy, T* __restrict__ gx, int B, int h,
                                                              int w, int C) {
    const int Q = C >> 2;
    const long total = (long)B * h * w * Q;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % Q);
        long r = i / Q;
        const int X = (int)(r % w); r /= w;
        const int Y = (int)(r % h);
        const int b = (int)(r / h);
        const T* base = gy + (long)b * 4 * h * w * C + c4 * 4;
        float4 s = f4zero();
#pragma unroll
        for (int dy = -1; dy <= 2; ++dy) {
            const int RY = 2 * Y + dy;
            const float wy = up2_adj_w(Y, RY, h);
            if (wy == 0.f) continue;
            float4 rowsum = f4zero();
#pragma unroll
            for (int dx = -1; dx <= 2; ++dx) {
                const int RX = 2 * X + dx;
                const float wx = up2_adj_w(X, RX, w);
                if (wx != 0.f) rowsum = f4fma(lda4(base + ((long)RY * (2 * w) + RX) * C), wx, rowsum);
            }
            s = f4fma(rowsum, wy, s);
        }
        sta4(gx + i * 4, s);
    }
}

// Adjoint of the bilinear x2 on feature maps, separable and branch-free.  A thread owns one (low-res column, channel quad) and
// walks a strip of low-res rows: per output it loads two NEW high-res rows (4 taps each, horizontally combined on the fly) and
// reuses the two it combined for the previous output -- 8 sixteen-byte loads per output instead of 16, no divergent control
// flow, so all of them are in flight together.  PNBWD: the result is the gradient w.r.t. the output y of a LeakyReLU -> PixelNorm;
// apply that operator's backward in the same pass (gc = m * (g - y*mean_c(g*y)) / r), saving a full read + write of the tensor.
template <typename T, int Q, int PNBWD>
__global__ __launch_bounds__(256) void up2_adjoint_strip_kernel(const T* __restrict__ gy, T* __restrict__ gx,
                                                                const T* __restrict__ yprev, const float* __restrict__ rn,
                                                                int h, int w, float slope, int YT) {
    constexpr int C = 4 * Q;
    const int tid = threadIdx.x, c4 = tid % Q;
    const int X = blockIdx.x * (256 / Q) + tid / Q;
    const bool xok = X < w;
    const int Xc = xok ? X : w - 1;
    const int b = blockIdx.z;
    const int W2 = 2 * w, H2 = 2 * h;
    float wx[4];
    int rx[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int RX = 2 * Xc + k - 1;
        wx[k] = up2_adj_w(Xc, RX, w);
        rx[k] = min(max(RX, 0), W2 - 1) * C + c4 * 4;
    }
    const T* base = gy + (long)b * H2 * W2 * C;
    auto hrow = [&](int RY) {
        const T* r = base + (long)min(max(RY, 0), H2 - 1) * W2 * C;
        float4 a = f4scale(lda4(r + rx[0]), wx[0]);
        a = f4fma(lda4(r + rx[1]), wx[1], a);
        a = f4fma(lda4(r + rx[2]), wx[2], a);
        return f4fma(lda4(r + rx[3]), wx[3], a);
    };
    const int Ys = blockIdx.y * YT, Ye = min(Ys + YT, h);
    float4 r_m1 = hrow(2 * Ys - 1), r_0 = hrow(2 * Ys);
    for (int Y = Ys; Y < Ye; ++Y) {
        const float4 r_1 = hrow(2 * Y + 1), r_2 = hrow(2 * Y + 2);
        float4 sacc = f4scale(r_m1, up2_adj_w(Y, 2 * Y - 1, h));
        sacc = f4fma(r_0, up2_adj_w(Y, 2 * Y, h), sacc);
        sacc = f4fma(r_1, up2_adj_w(Y, 2 * Y + 1, h), sacc);
        sacc = f4fma(r_2, up2_adj_w(Y, 2 * Y + 2, h), sacc);
        const long pix = ((long)b * h + Y) * w + Xc;
        if (PNBWD) {
            const float4 yy = lda4(yprev + pix * C + c4 * 4);
            const float dot = group_sum<Q>(f4dot(sacc, yy)) * (1.0f / (float)C);
            sacc = pn_bwd4_pw(sacc, yy, dot, 1.0f / rn[pix], slope);
        }
        if (xok) sta4(gx + pix * C + c4 * 4, sacc);
        r_m1 = r_1; r_0 = r_2;
    }
}

template <typename T>
__global__ void pool2_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int B, int h, int w, int C) {
    const long total = (long)B * h * w * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        long r = i / C;
        const int X = (int)(r % w); r /= w;
        const int Y = (int)(r % h);
        const int b = (int)(r / h);
        const long W2 = 2L * w;
        const T* p = x + (((long)b * 2 * h + 2 * Y) * W2 + 2 * X) * C + c;
        sta1(y + i, 0.25f * ((lda1(p) + lda1(p + C)) + (lda1(p + W2 * C) + lda1(p + W2 * C + C))));
    }
}

// the same for C a multiple of 4: a thread takes 4 channels of an output pixel (4 x 16-byte loads, one 16-byte store); 32-bit
// index arithmetic (the host checks the element count).  Same association as the scalar kernel: bit-identical.
template <typename T>
__global__ __launch_bounds__(256) void pool2_fwd_v4_kernel(const T* __restrict__ x, T* __restrict__ y, int total4, int h, int w, int C4) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= total4) return;
    const int c4 = i % C4;
    int r = i / C4;
    const int X = r % w; r /= w;                    // r = b * h + Y from here on
    const int C = 4 * C4;
    const long row = (long)2 * w * C;
    const T* p = x + ((long)2 * r * 2 * w + 2 * X) * C + 4 * c4;
    const float4 a = lda4(p), b = lda4(p + C), c = lda4(p + row), d = lda4(p + row + C);
    sta4(y + (long)i * 4, make_float4(0.25f * ((a.x + b.x) + (c.x + d.x)), 0.25f * ((a.y + b.y) + (c.y + d.y)),
                                     0.25f * ((a.z + b.z) + (c.z + d.z)), 0.25f * ((a.w + b.w) + (c.w + d.w))));
}

template <typename T>
__global__ void pool2_adjoint_kernel(const T* __restrict__ gy, T* __restrict__ gx, int B, int h, int w, int C) {
    const long total = (long)B * 4 * h * w * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        long r = i / C;
        const int X = (int)(r % (2 * w)); r /= (2 * w);
        const int Y = (int)(r % (2 * h));
        const int b = (int)(r / (2 * h));
        sta1(gx + i, 0.25f * lda1(gy + (((long)b * h + (Y >> 1)) * w + (X >> 1)) * C + c));
    }
}

// ------------------------------------------------------------------------------------------------------------
// elementwise arithmetic
// ------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ void lerp_kernel(const T* __restrict__ a, const T* __restrict__ b, const float* __restrict__ alpha,
                            T* __restrict__ out, long n) {
    const float al = alpha[0];
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float av = lda1(a + i);
        sta1(out + i, fmaf(al, lda1(b + i) - av, av));
    }
}

__global__ void axpby_kernel(const float* __restrict__ a, const float* __restrict__ b, float ca, float cb,
                             float* __restrict__ out, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        out[i] = b ? fmaf(cb, b[i], ca * a[i]) : ca * a[i];
}

template <typename T>
__global__ void fade_bwd_kernel(const T* __restrict__ g, const float* __restrict__ alpha, T* __restrict__ ga,
                                T* __restrict__ gb, long n) {
    const float al = alpha[0];
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float v = lda1(g + i);
        sta1(ga + i, (1.0f - al) * v);
        sta1(gb + i, al * v);
    }
}

__global__ void xhat_kernel(const float* __restrict__ real, const float* __restrict__ fake, const float* __restrict__ eps,
                            float* __restrict__ out, long n) {
    const int b = blockIdx.y;
    const float e = eps[b];
    const long base = (long)b * n;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        out[base + i] = e * real[base + i] + (1.0f - e) * fake[base + i];
}

__global__ void scale_rows_kernel(const float* __restrict__ g, const float* __restrict__ coef, float* __restrict__ out, long n) {
    const int b = blockIdx.y;
    const float c = coef[b];
    const long base = (long)b * n;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        out[base + i] = c * g[base + i];
}

// per-sample L2 norm in two fixed-order stages: NCHUNK blocks per sample write partial sums of squares (16-byte loads, 4
// independent accumulators per thread), one wave per sample adds them and takes the root
constexpr int L2_CHUNKS = 64;

__global__ __launch_bounds__(256) void sample_sumsq_kernel(const float* __restrict__ g, float* __restrict__ partial, long n) {
    __shared__ float red[4];
    const int b = blockIdx.y, tid = threadIdx.x;
    const float* p = g + (long)b * n;
    const long n4 = n >> 2;
    float4 a = f4zero();
    for (long i = (long)blockIdx.x * 256 + tid; i < n4; i += (long)gridDim.x * 256) {
        const float4 v = ld4(p + i * 4);
        a.x = fmaf(v.x, v.x, a.x); a.y = fmaf(v.y, v.y, a.y); a.z = fmaf(v.z, v.z, a.z); a.w = fmaf(v.w, v.w, a.w);
    }
    float s = (a.x + a.y) + (a.z + a.w);
    if (blockIdx.x == 0)
        for (long i = n4 * 4 + tid; i < n; i += 256) s = fmaf(p[i], p[i], s);      // tail when n is not a multiple of 4
    s = group_sum<64>(s);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) partial[(long)b * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(64) void sample_l2norm_finish_kernel(const float* __restrict__ partial, float* __restrict__ norms, int nchunk) {
    const int b = blockIdx.x, tid = threadIdx.x;
    float s = 0.f;
    for (int i = tid; i < nchunk; i += 64) s += partial[(long)b * nchunk + i];
    s = group_sum<64>(s);
    if (tid == 0) norms[b] = sqrtf(s);
}

int ew_blocks(long n) {
    long b = (n + 255) / 256;
    return (int)(b < 4096 ? (b < 1 ? 1 : b) : 4096);
}

#define Q_DISPATCH(CALL)                       \
    switch (C / 4) {                           \
        case 1: CALL(1); break;                \
        case 2: CALL(2); break;                \
        case 4: CALL(4); break;                \
        case 8: CALL(8); break;                \
        case 16: CALL(16); break;              \
        case 32: CALL(32); break;              \
        default: CALL(64); break;              \
    }


// ---------------------------------------------------------------------------------------------------------
// Scalar heads of the losses and the latent projection: each replaces a chain of 4 - 10 ATen elementwise / reduction launches
// of a few microseconds (88 such launches were 5 % of an exact-fp32 iteration, 7 % of a split-bf16 one).
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float block_sum_256(float v, float* red) {
    v = group_sum<64>(v);
    const int tid = threadIdx.x;
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// z[r, :] = clamp(z[r, :], -c, c) / ||clamp(z[r, :], -c, c)||_2   (reference utils.py:77-78), in place; one wave per row
__global__ __launch_bounds__(64) void latent_normalize_kernel(float* __restrict__ z, int dim, float c) {
    float* row = z + (long)blockIdx.x * dim;
    float ss = 0.f;
    for (int i = threadIdx.x; i < dim; i += 64) {
        const float v = fminf(fmaxf(row[i], -c), c);
        ss += v * v;
    }
    ss = group_sum<64>(ss);
    const float inv = 1.0f / sqrtf(ss);
    for (int i = threadIdx.x; i < dim; i += 64) row[i] = fminf(fmaxf(row[i], -c), c) * inv;
}

// out = {loss, mean real score, mean fake score}; loss = -mean(real) + mean(fake) + drift * mean(real^2)  (loss_functions.py:21-45;
// with n_fake = 0: -mean(real), the generator loss, loss_functions.py:67)
__global__ __launch_bounds__(256) void wloss_head_kernel(const float* __restrict__ scores, int n_real, int n_fake, float drift,
                                                         float* __restrict__ o_loss, float* __restrict__ o_real, float* __restrict__ o_fake) {
    __shared__ float red[4];
    float sr = 0.f, sq = 0.f, sf = 0.f;
    for (int i = threadIdx.x; i < n_real; i += 256) { const float v = scores[i]; sr += v; sq += v * v; }
    for (int i = threadIdx.x; i < n_fake; i += 256) sf += scores[n_real + i];
    sr = block_sum_256(sr, red); sq = block_sum_256(sq, red); sf = block_sum_256(sf, red);
    if (threadIdx.x == 0) {
        const float mr = sr / (float)n_real, mf = n_fake ? sf / (float)n_fake : 0.f;
        o_loss[0] = -mr + mf + (drift > 0.f ? drift * sq / (float)n_real : 0.f);
        o_real[0] = mr; o_fake[0] = mf;
    }
}

__global__ __launch_bounds__(256) void wloss_head_bwd_kernel(const float* __restrict__ scores, int n_real, int n_fake, float drift,
                                                             const float* __restrict__ g_loss, const float* __restrict__ g_real,
                                                             const float* __restrict__ g_fake, float* __restrict__ gs) {
    const float gl = g_loss ? g_loss[0] : 0.f, gr = g_real ? g_real[0] : 0.f, gf = g_fake ? g_fake[0] : 0.f;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n_real) gs[i] = (gl * (-1.0f + 2.0f * drift * scores[i]) + gr) / (float)n_real;
    else if (i < n_real + n_fake) gs[i] = (gl + gf) / (float)n_fake;
}

// penalty = lambda * mean((norms - 1)^2)  (loss_functions.py:176);  coef[b] = g_out * 2 lambda (norms[b] - 1) / (B norms[b])
__global__ __launch_bounds__(256) void gp_head_kernel(const float* __restrict__ norms, int B, float lambda, float* __restrict__ out) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < B; i += 256) { const float d = norms[i] - 1.0f; s += d * d; }
    s = block_sum_256(s, red);
    if (threadIdx.x == 0) out[0] = lambda * s / (float)B;
}

__global__ __launch_bounds__(256) void gp_coef_kernel(const float* __restrict__ norms, int B, float lambda, const float* __restrict__ g_out,
                                                      float* __restrict__ coef) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < B) coef[i] = g_out[0] * 2.0f * lambda * (norms[i] - 1.0f) / ((float)B * norms[i]);
}

}  // namespace

extern "C" int ngan_latent_normalize(float* z, int rows, int dim, float clamp, void* stream) {
    NGAN_REQUIRE(z && rows > 0 && dim > 0 && clamp > 0.f, NGAN_ERR_ARG, "latent_normalize: bad argument");
    hipLaunchKernelGGL(latent_normalize_kernel, dim3(rows), dim3(64), 0, (hipStream_t)stream, z, dim, clamp);
    return ngan::launch_status("ngan_latent_normalize");
}

extern "C" int ngan_wloss_head(const float* scores, int n_real, int n_fake, float drift, float* loss, float* mean_real, float* mean_fake,
                               void* stream) {
    NGAN_REQUIRE(scores && loss && mean_real && mean_fake && n_real > 0 && n_fake >= 0, NGAN_ERR_ARG, "wloss_head: bad argument");
    hipLaunchKernelGGL(wloss_head_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, scores, n_real, n_fake, drift, loss, mean_real, mean_fake);
    return ngan::launch_status("ngan_wloss_head");
}

extern "C" int ngan_wloss_head_bwd(const float* scores, int n_real, int n_fake, float drift, const float* g_loss, const float* g_real,
                                   const float* g_fake, float* g_scores, void* stream) {
    NGAN_REQUIRE(scores && g_scores && n_real > 0 && n_fake >= 0, NGAN_ERR_ARG, "wloss_head_bwd: bad argument");
    hipLaunchKernelGGL(wloss_head_bwd_kernel, dim3(ngan::ceil_div(n_real + n_fake, 256)), dim3(256), 0, (hipStream_t)stream, scores, n_real,
                       n_fake, drift, g_loss, g_real, g_fake, g_scores);
    return ngan::launch_status("ngan_wloss_head_bwd");
}

extern "C" int ngan_gp_head(const float* norms, int B, float lambda, float* out1, void* stream) {
    NGAN_REQUIRE(norms && out1 && B > 0, NGAN_ERR_ARG, "gp_head: bad argument");
    hipLaunchKernelGGL(gp_head_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, norms, B, lambda, out1);
    return ngan::launch_status("ngan_gp_head");
}

extern "C" int ngan_gp_coef(const float* norms, int B, float lambda, const float* g_out, float* coef, void* stream) {
    NGAN_REQUIRE(norms && g_out && coef && B > 0, NGAN_ERR_ARG, "gp_coef: bad argument");
    hipLaunchKernelGGL(gp_coef_kernel, dim3(ngan::ceil_div(B, 256)), dim3(256), 0, (hipStream_t)stream, norms, B, lambda, g_out, coef);
    return ngan::launch_status("ngan_gp_coef");
}


// ---- entry points over the activation type: T = float is the fp32 contract of include/ngan.h (csrc/wide.hip behind it for channel counts
// outside the lane-group kernels' range), T = __bf16 the "bf16 activation storage" section (no wide path: NGAN_ERR_SHAPE instead)
#define BF(p) reinterpret_cast<const __bf16*>(p)
#define BFM(p) reinterpret_cast<__bf16*>(p)
template <typename T> constexpr bool is_f32() { return sizeof(T) == 4; }

template <typename T>
static int channel_sum_impl(const T* g, float* out, float* workspace, long npix, int C, float scale, int accumulate, void* stream) {
    NGAN_REQUIRE(g && out && workspace, NGAN_ERR_ARG, "channel_sum: null pointer");
    if constexpr (is_f32<T>())
        if (npix > 0 && C > 0 && !pow2_quads(C)) {                                                                                // wide.hip
            NGAN_REQUIRE(!accumulate, NGAN_ERR_SHAPE, "channel_sum: accumulate is not available for C=%d", C);
            return ngan::wide_channel_sum(g, out, npix, C, scale, (hipStream_t)stream);
        }
    NGAN_REQUIRE(npix > 0 && pow2_quads(C), NGAN_ERR_SHAPE, "channel_sum: npix=%ld C=%d unsupported", npix, C);
    hipStream_t s = (hipStream_t)stream;
    const int nblk = stream_blocks(npix, C / 4);
#define CALL(QV) hipLaunchKernelGGL((channel_sum_kernel<T, QV>), dim3(nblk), dim3(256), 0, s, g, workspace, npix, C)
    Q_DISPATCH(CALL)
#undef CALL
    int st = ngan::launch_status("ngan_channel_sum");
    if (st) return st;
    return ngan::reduce_partials_acc(workspace, nblk, C, C, out, C, nullptr, scale, accumulate ? 1 : 0, s);
}
extern "C" int ngan_channel_sum_acc(const float* g, float* out, float* workspace, long npix, int C, float scale, int accumulate, void* stream) {
    return channel_sum_impl<float>(g, out, workspace, npix, C, scale, accumulate, stream);
}
extern "C" int ngan_channel_sum(const float* g, float* out, float* workspace, long npix, int C, float scale, void* stream) {
    return channel_sum_impl<float>(g, out, workspace, npix, C, scale, 0, stream);
}
extern "C" int ngan_bf16_channel_sum_acc(const ngan_bf16* g, float* out, float* workspace, long npix, int C, float scale, int accumulate, void* stream) {
    return channel_sum_impl<__bf16>(BF(g), out, workspace, npix, C, scale, accumulate, stream);
}
extern "C" int ngan_bf16_channel_sum(const ngan_bf16* g, float* out, float* workspace, long npix, int C, float scale, void* stream) {
    return channel_sum_impl<__bf16>(BF(g), out, workspace, npix, C, scale, 0, stream);
}

template <typename T>
static int from_image_fwd_impl(const float* x, const float* w, const float* b, T* y, int B, int H, int W, int Ncol, int C, int pool, void* stream) {
    NGAN_REQUIRE(x && w && y, NGAN_ERR_ARG, "from_image_fwd: null pointer");
    NGAN_REQUIRE(B > 0 && H > 0 && W > 0 && Ncol >= 1 && Ncol <= 4 && C > 0 && C % 4 == 0, NGAN_ERR_SHAPE,
                 "from_image_fwd: B=%d H=%d W=%d Ncol=%d C=%d unsupported", B, H, W, Ncol, C);
    NGAN_REQUIRE((long)B * H * W * (C / 4) < (1L << 31), NGAN_ERR_SHAPE, "from_image_fwd: B*H*W*C/4 must be below 2^31");
    hipStream_t s = (hipStream_t)stream;
    NGAN_REQUIRE((long)B * H < 65536, NGAN_ERR_SHAPE, "from_image_fwd: B*H must be below 65536");
    bool wide_access = false;
    if constexpr (!is_f32<T>()) wide_access = C % 8 == 0;              // bf16 storage: 8 channels (16 bytes) per thread
    if (wide_access) {
        const dim3 grid(ceil_div((long)W * (C / 8), 256), B * H);
        if (pool) hipLaunchKernelGGL((from_image_fwd_kernel<T, 1, 2>), grid, dim3(256), 0, s, x, w, b, y, B, H, W, Ncol, C);
        else hipLaunchKernelGGL((from_image_fwd_kernel<T, 0, 2>), grid, dim3(256), 0, s, x, w, b, y, B, H, W, Ncol, C);
    } else {
        const dim3 grid(ceil_div((long)W * (C / 4), 256), B * H);
        if (pool) hipLaunchKernelGGL((from_image_fwd_kernel<T, 1, 1>), grid, dim3(256), 0, s, x, w, b, y, B, H, W, Ncol, C);
        else hipLaunchKernelGGL((from_image_fwd_kernel<T, 0, 1>), grid, dim3(256), 0, s, x, w, b, y, B, H, W, Ncol, C);
    }
    return ngan::launch_status("ngan_from_image_fwd");
}
extern "C" int ngan_from_image_fwd(const float* x, const float* w, const float* b, float* y, int B, int H, int W, int Ncol,
                                   int C, int pool, void* stream) {
    return from_image_fwd_impl<float>(x, w, b, y, B, H, W, Ncol, C, pool, stream);
}
extern "C" int ngan_bf16_from_image_fwd(const float* x, const float* w, const float* b, ngan_bf16* y, int B, int H, int W, int Ncol,
                                        int C, int pool, void* stream) {
    return from_image_fwd_impl<__bf16>(x, w, b, BFM(y), B, H, W, Ncol, C, pool, stream);
}

template <typename T>
static int from_image_dx_impl(const T* g, const float* w, float* gx, int B, int H, int W, int Ncol, int C, int pool, void* stream) {
    NGAN_REQUIRE(g && w && gx, NGAN_ERR_ARG, "from_image_dx: null pointer");
    if constexpr (is_f32<T>())
        if (B > 0 && H > 0 && W > 0 && Ncol >= 1 && Ncol <= 4 && C > 0 && !pow2_quads(C))
            return ngan::wide_from_image_dx(g, w, gx, B, H, W, Ncol, C, pool, (hipStream_t)stream);                             // wide.hip
    NGAN_REQUIRE(B > 0 && H > 0 && W > 0 && Ncol >= 1 && Ncol <= 4 && pow2_quads(C), NGAN_ERR_SHAPE,
                 "from_image_dx: B=%d H=%d W=%d Ncol=%d C=%d unsupported", B, H, W, Ncol, C);
    NGAN_REQUIRE((long)B * H * W * (C / 4) < (1L << 31), NGAN_ERR_SHAPE, "from_image_dx: B*H*W*C/4 must be below 2^31");
    hipStream_t s = (hipStream_t)stream;
    const int nblk = ceil_div((long)B * H * W * (C / 4), 256);
#define CALL(QV)                                                                                                     \
    if (pool) hipLaunchKernelGGL((from_image_dx_kernel<T, QV, 1>), dim3(nblk), dim3(256), 0, s, g, w, gx, B, H, W, Ncol, C); \
    else hipLaunchKernelGGL((from_image_dx_kernel<T, QV, 0>), dim3(nblk), dim3(256), 0, s, g, w, gx, B, H, W, Ncol, C)
    Q_DISPATCH(CALL)
#undef CALL
    return ngan::launch_status("ngan_from_image_dx");
}
extern "C" int ngan_from_image_dx(const float* g, const float* w, float* gx, int B, int H, int W, int Ncol, int C, int pool,
                                  void* stream) {
    return from_image_dx_impl<float>(g, w, gx, B, H, W, Ncol, C, pool, stream);
}
extern "C" int ngan_bf16_from_image_dx(const ngan_bf16* g, const float* w, float* gx, int B, int H, int W, int Ncol, int C, int pool,
                                       void* stream) {
    return from_image_dx_impl<__bf16>(BF(g), w, gx, B, H, W, Ncol, C, pool, stream);
}

// accumulate: bit 0 gw += , bit 1 gb +=
template <typename T>
static int from_image_dw_impl(const float* x, const T* g, float* gw, float* gb, float* workspace, int B, int H, int W,
                              int Ncol, int C, int pool, int accumulate, void* stream) {
    NGAN_REQUIRE(x && g && gw && workspace, NGAN_ERR_ARG, "from_image_dw: null pointer");
    if constexpr (is_f32<T>())
        if (B > 0 && H > 0 && W > 0 && Ncol >= 1 && Ncol <= 4 && C > 0 && !pow2_quads(C)) {
            NGAN_REQUIRE(!accumulate, NGAN_ERR_SHAPE, "from_image_dw: accumulate is not available for C=%d", C);
            return ngan::wide_from_image_dw(x, g, gw, gb, B, H, W, Ncol, C, pool, (hipStream_t)stream);                         // wide.hip
        }
    NGAN_REQUIRE(B > 0 && H > 0 && W > 0 && Ncol >= 1 && Ncol <= 4 && pow2_quads(C), NGAN_ERR_SHAPE,
                 "from_image_dw: B=%d H=%d W=%d Ncol=%d C=%d unsupported", B, H, W, Ncol, C);
    NGAN_REQUIRE((long)B * H * W * (C / 4) < (1L << 31), NGAN_ERR_SHAPE, "from_image_dw: B*H*W*C/4 must be below 2^31");
    hipStream_t s = (hipStream_t)stream;
    const int rows = B * H;
    const int rpb_min = ceil_div(rows, MAX_PARTS), rpb_pref = rows >= 2048 ? 4 : 1;
    const int rpb = rpb_min > rpb_pref ? rpb_min : rpb_pref;
    const int nblk = ceil_div(rows, rpb);                 // <= MAX_PARTS slabs (the callers' workspace holds 1024)
#define CALL(QV)                                                                                                                      \
    if (pool) hipLaunchKernelGGL((from_image_dw_kernel<T, QV, 1>), dim3(nblk), dim3(256), 0, s, x, g, workspace, B, H, W, Ncol, C, rpb); \
    else hipLaunchKernelGGL((from_image_dw_kernel<T, QV, 0>), dim3(nblk), dim3(256), 0, s, x, g, workspace, B, H, W, Ncol, C, rpb)
    Q_DISPATCH(CALL)
#undef CALL
    int st = ngan::launch_status("ngan_from_image_dw");
    if (st) return st;
    const long stride = (long)C * (Ncol + 1);
    if (!gb) return ngan::reduce_partials_acc(workspace, nblk, C * Ncol, stride, gw, C * Ncol, nullptr, 1.0f, accumulate & 1, s);
    return ngan::reduce_partials_acc(workspace, nblk, C * (Ncol + 1), stride, gw, C * Ncol, gb, 1.0f, accumulate, s);   // weight and bias sums: one launch
}
extern "C" int ngan_from_image_dw_acc(const float* x, const float* g, float* gw, float* gb, float* workspace, int B, int H, int W,
                                      int Ncol, int C, int pool, int accumulate, void* stream) {
    return from_image_dw_impl<float>(x, g, gw, gb, workspace, B, H, W, Ncol, C, pool, accumulate, stream);
}
extern "C" int ngan_from_image_dw(const float* x, const float* g, float* gw, float* gb, float* workspace, int B, int H, int W,
                                  int Ncol, int C, int pool, void* stream) {
    return from_image_dw_impl<float>(x, g, gw, gb, workspace, B, H, W, Ncol, C, pool, 0, stream);
}
extern "C" int ngan_bf16_from_image_dw_acc(const float* x, const ngan_bf16* g, float* gw, float* gb, float* workspace, int B, int H, int W,
                                           int Ncol, int C, int pool, int accumulate, void* stream) {
    return from_image_dw_impl<__bf16>(x, BF(g), gw, gb, workspace, B, H, W, Ncol, C, pool, accumulate, stream);
}
extern "C" int ngan_bf16_from_image_dw(const float* x, const ngan_bf16* g, float* gw, float* gb, float* workspace, int B, int H, int W,
                                       int Ncol, int C, int pool, void* stream) {
    return from_image_dw_impl<__bf16>(x, BF(g), gw, gb, workspace, B, H, W, Ncol, C, pool, 0, stream);
}

template <typename T>
static int to_image_fwd_impl(const T* x, const float* w, float* t, long npix, int C, int Ncol, void* stream) {
    NGAN_REQUIRE(x && w && t, NGAN_ERR_ARG, "to_image_fwd: null pointer");
    if constexpr (is_f32<T>())
        if (npix > 0 && Ncol >= 1 && Ncol <= 4 && C > 0 && C % 4 == 0 && !pow2_quads(C))
            return ngan::wide_to_image_fwd(x, w, t, npix, C, Ncol, (hipStream_t)stream);                                        // wide.hip
    NGAN_REQUIRE(npix > 0 && Ncol >= 1 && Ncol <= 4 && pow2_quads(C), NGAN_ERR_SHAPE, "to_image_fwd: npix=%ld C=%d Ncol=%d unsupported",
                 npix, C, Ncol);
    hipStream_t s = (hipStream_t)stream;
    const int nblk = ceil_div(npix * (C / 4), 256);
#define CALL(QV) hipLaunchKernelGGL((to_image_fwd_kernel<T, QV>), dim3(nblk), dim3(256), 0, s, x, w, t, npix, C, Ncol)
    Q_DISPATCH(CALL)
#undef CALL
    return ngan::launch_status("ngan_to_image_fwd");
}
extern "C" int ngan_to_image_fwd(const float* x, const float* w, float* t, long npix, int C, int Ncol, void* stream) {
    return to_image_fwd_impl<float>(x, w, t, npix, C, Ncol, stream);
}
extern "C" int ngan_bf16_to_image_fwd(const ngan_bf16* x, const float* w, float* t, long npix, int C, int Ncol, void* stream) {
    return to_image_fwd_impl<__bf16>(BF(x), w, t, npix, C, Ncol, stream);
}

template <typename T>
static int to_image_bwd_impl(const float* g, const float* t, const T* x, const float* w, T* gx, float* gw,
                             float* workspace, long npix, int C, int Ncol, const float* rn, float slope, int accumulate, void* stream) {
    NGAN_REQUIRE(g && t && x && w && gx && gw && workspace, NGAN_ERR_ARG, "to_image_bwd: null pointer");
    if constexpr (is_f32<T>())
        if (npix > 0 && Ncol >= 1 && Ncol <= 4 && C > 0 && C % 4 == 0 && !pow2_quads(C)) {
            NGAN_REQUIRE(!accumulate, NGAN_ERR_SHAPE, "to_image_bwd: accumulate is not available for C=%d", C);
            return ngan::wide_to_image_bwd(g, t, x, w, gx, gw, npix, C, Ncol, rn, slope, (hipStream_t)stream);                   // wide.hip
        }
    NGAN_REQUIRE(npix > 0 && Ncol >= 1 && Ncol <= 4 && pow2_quads(C), NGAN_ERR_SHAPE, "to_image_bwd: npix=%ld C=%d Ncol=%d unsupported",
                 npix, C, Ncol);
    hipStream_t s = (hipStream_t)stream;
    const int nblk = stream_blocks(npix, C / 4);
    bool wide_access = false;
    if constexpr (!is_f32<T>()) wide_access = C % 8 == 0;              // bf16 storage: 8 channels (16 bytes) per lane
    if (wide_access) {
        switch (C / 8) {
#define CALL(QV) case QV: if (Ncol == 1) hipLaunchKernelGGL((to_image_bwd_kernel<T, QV, 2, 1>), dim3(nblk), dim3(256), 0, s, g, t, x, w, gx, workspace, npix, C, Ncol, rn, slope); \
                          else hipLaunchKernelGGL((to_image_bwd_kernel<T, QV, 2, 0>), dim3(nblk), dim3(256), 0, s, g, t, x, w, gx, workspace, npix, C, Ncol, rn, slope); break;
            CALL(1) CALL(2) CALL(4) CALL(8) CALL(16) CALL(32)
#undef CALL
            default: NGAN_REQUIRE(false, NGAN_ERR_SHAPE, "to_image_bwd: C=%d unsupported", C);
        }
    } else {
#define CALL(QV) do { if (Ncol == 1) hipLaunchKernelGGL((to_image_bwd_kernel<T, QV, 1, 1>), dim3(nblk), dim3(256), 0, s, g, t, x, w, gx, workspace, npix, C, Ncol, rn, slope); \
                      else hipLaunchKernelGGL((to_image_bwd_kernel<T, QV, 1, 0>), dim3(nblk), dim3(256), 0, s, g, t, x, w, gx, workspace, npix, C, Ncol, rn, slope); } while (0)
        Q_DISPATCH(CALL)
#undef CALL
    }
    int st = ngan::launch_status("ngan_to_image_bwd");
    if (st) return st;
    return ngan::reduce_partials_acc(workspace, nblk, C * Ncol, C * Ncol, gw, C * Ncol, nullptr, 1.0f, accumulate ? 1 : 0, s);
}
extern "C" int ngan_to_image_bwd(const float* g, const float* t, const float* x, const float* w, float* gx, float* gw,
                                 float* workspace, long npix, int C, int Ncol, void* stream) {
    return to_image_bwd_impl<float>(g, t, x, w, gx, gw, workspace, npix, C, Ncol, nullptr, 0.f, 0, stream);
}
extern "C" int ngan_to_image_bwd_pnbwd(const float* g, const float* t, const float* y, const float* rnorm, const float* w, float* gc,
                                       float* gw, float* workspace, long npix, int C, int Ncol, float slope, void* stream) {
    NGAN_REQUIRE(rnorm, NGAN_ERR_ARG, "to_image_bwd_pnbwd: null pointer");
    return to_image_bwd_impl<float>(g, t, y, w, gc, gw, workspace, npix, C, Ncol, rnorm, slope, 0, stream);
}
// the same with gw += (accumulate != 0): the colour weights' gradient added straight into an existing gradient buffer
extern "C" int ngan_to_image_bwd_pnbwd_acc(const float* g, const float* t, const float* y, const float* rnorm, const float* w, float* gc,
                                           float* gw, float* workspace, long npix, int C, int Ncol, float slope, int accumulate, void* stream) {
    NGAN_REQUIRE(rnorm, NGAN_ERR_ARG, "to_image_bwd_pnbwd: null pointer");
    return to_image_bwd_impl<float>(g, t, y, w, gc, gw, workspace, npix, C, Ncol, rnorm, slope, accumulate, stream);
}
extern "C" int ngan_bf16_to_image_bwd(const float* g, const float* t, const ngan_bf16* x, const float* w, ngan_bf16* gx, float* gw,
                                      float* workspace, long npix, int C, int Ncol, void* stream) {
    return to_image_bwd_impl<__bf16>(g, t, BF(x), w, BFM(gx), gw, workspace, npix, C, Ncol, nullptr, 0.f, 0, stream);
}
extern "C" int ngan_bf16_to_image_bwd_pnbwd(const float* g, const float* t, const ngan_bf16* y, const float* rnorm, const float* w, ngan_bf16* gc,
                                            float* gw, float* workspace, long npix, int C, int Ncol, float slope, void* stream) {
    NGAN_REQUIRE(rnorm, NGAN_ERR_ARG, "to_image_bwd_pnbwd: null pointer");
    return to_image_bwd_impl<__bf16>(g, t, BF(y), w, BFM(gc), gw, workspace, npix, C, Ncol, rnorm, slope, 0, stream);
}
extern "C" int ngan_bf16_to_image_bwd_pnbwd_acc(const float* g, const float* t, const ngan_bf16* y, const float* rnorm, const float* w,
                                                ngan_bf16* gc, float* gw, float* workspace, long npix, int C, int Ncol, float slope, int accumulate,
                                                void* stream) {
    NGAN_REQUIRE(rnorm, NGAN_ERR_ARG, "to_image_bwd_pnbwd: null pointer");
    return to_image_bwd_impl<__bf16>(g, t, BF(y), w, BFM(gc), gw, workspace, npix, C, Ncol, rnorm, slope, accumulate, stream);
}

template <typename T>
static int up2_fwd_impl(const T* a, T* o, int B, int h, int w, int C, void* stream) {
    NGAN_REQUIRE(a && o, NGAN_ERR_ARG, "ngan_up2_fwd: null pointer");
    NGAN_REQUIRE(B > 0 && h > 0 && w > 0 && C > 0, NGAN_ERR_SHAPE, "ngan_up2_fwd: bad dims %d %d %d %d", B, h, w, C);
    hipLaunchKernelGGL(up2_fwd_kernel<T>, dim3(ew_blocks((long)B * 4 * h * w * C)), dim3(256), 0, (hipStream_t)stream, a, o, B, h, w, C);
    return ngan::launch_status("ngan_up2_fwd");
}
extern "C" int ngan_up2_fwd(const float* a, float* o, int B, int h, int w, int C, void* stream) { return up2_fwd_impl<float>(a, o, B, h, w, C, stream); }
extern "C" int ngan_bf16_up2_fwd(const ngan_bf16* a, ngan_bf16* o, int B, int h, int w, int C, void* stream) {
    return up2_fwd_impl<__bf16>(BF(a), BFM(o), B, h, w, C, stream);
}

template <typename T>
static int launch_up2_adjoint_strip(const T* g, const T* yprev, const float* rn, T* o, int B, int h, int w, int C,
                                    float slope, hipStream_t s) {
    const int Q = C / 4, YT = h < 16 ? h : 16;
    const dim3 grid(ceil_div(w, 256 / Q), ceil_div(h, YT), B), block(256);
#define CALL(QV)                                                                                                                   \
    if (yprev) hipLaunchKernelGGL((up2_adjoint_strip_kernel<T, QV, 1>), grid, block, 0, s, g, o, yprev, rn, h, w, slope, YT);        \
    else hipLaunchKernelGGL((up2_adjoint_strip_kernel<T, QV, 0>), grid, block, 0, s, g, o, yprev, rn, h, w, slope, YT)
    Q_DISPATCH(CALL)
#undef CALL
    return ngan::launch_status("ngan_up2_adjoint(strip)");
}

template <typename T>
static int up2_adjoint_impl(const T* a, T* o, int B, int h, int w, int C, void* stream) {
    NGAN_REQUIRE(a && o, NGAN_ERR_ARG, "ngan_up2_adjoint: null pointer");
    NGAN_REQUIRE(B > 0 && h > 0 && w > 0 && C > 0, NGAN_ERR_SHAPE, "ngan_up2_adjoint: bad dims %d %d %d %d", B, h, w, C);
    const long total = (long)B * h * w * C;
    if (pow2_quads(C) && B < 65536) return launch_up2_adjoint_strip<T>(a, nullptr, nullptr, o, B, h, w, C, 0.f, (hipStream_t)stream);
    if (C % 4 == 0) {
        long nb = (total / 4 + 255) / 256;
        hipLaunchKernelGGL(up2_adjoint_vec_kernel<T>, dim3((int)(nb < 8192 ? nb : 8192)), dim3(256), 0, (hipStream_t)stream, a, o, B, h, w, C);
    } else {
        hipLaunchKernelGGL(up2_adjoint_kernel<T>, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, a, o, B, h, w, C);
    }
    return ngan::launch_status("ngan_up2_adjoint");
}
extern "C" int ngan_up2_adjoint(const float* a, float* o, int B, int h, int w, int C, void* stream) { return up2_adjoint_impl<float>(a, o, B, h, w, C, stream); }
extern "C" int ngan_bf16_up2_adjoint(const ngan_bf16* a, ngan_bf16* o, int B, int h, int w, int C, void* stream) {
    return up2_adjoint_impl<__bf16>(BF(a), BFM(o), B, h, w, C, stream);
}

extern "C" int ngan_up2_adjoint_pnbwd(const float* g, const float* yprev, const float* rnorm, float* o, int B, int h, int w, int C,
                                      float slope, void* stream) {
    NGAN_REQUIRE(g && yprev && rnorm && o, NGAN_ERR_ARG, "ngan_up2_adjoint_pnbwd: null pointer");
    if (B > 0 && h > 0 && w > 0 && C > 0 && C % 4 == 0 && !pow2_quads(C)) {      // wide layers: the two operators one after the other (wide.hip)
        const int st = ngan_up2_adjoint(g, o, B, h, w, C, stream);
        return st ? st : ngan::wide_pn_bwd(o, nullptr, nullptr, yprev, rnorm, o, (long)B * h * w, C, slope, (hipStream_t)stream);
    }
    NGAN_REQUIRE(B > 0 && B < 65536 && h > 0 && w > 0 && pow2_quads(C), NGAN_ERR_SHAPE, "ngan_up2_adjoint_pnbwd: bad dims %d %d %d %d", B, h, w, C);
    return launch_up2_adjoint_strip<float>(g, yprev, rnorm, o, B, h, w, C, slope, (hipStream_t)stream);
}
extern "C" int ngan_bf16_up2_adjoint_pnbwd(const ngan_bf16* g, const ngan_bf16* yprev, const float* rnorm, ngan_bf16* o, int B, int h, int w,
                                           int C, float slope, void* stream) {
    NGAN_REQUIRE(g && yprev && rnorm && o, NGAN_ERR_ARG, "ngan_bf16_up2_adjoint_pnbwd: null pointer");
    NGAN_REQUIRE(B > 0 && B < 65536 && h > 0 && w > 0 && pow2_quads(C), NGAN_ERR_SHAPE, "ngan_bf16_up2_adjoint_pnbwd: bad dims %d %d %d %d", B, h, w, C);
    return launch_up2_adjoint_strip<__bf16>(BF(g), BF(yprev), rnorm, BFM(o), B, h, w, C, slope, (hipStream_t)stream);
}

template <typename T>
static int pool2_fwd_impl(const T* a, T* o, int B, int h, int w, int C, void* stream) {
    NGAN_REQUIRE(a && o, NGAN_ERR_ARG, "ngan_pool2_fwd: null pointer");
    NGAN_REQUIRE(B > 0 && h > 0 && w > 0 && C > 0, NGAN_ERR_SHAPE, "ngan_pool2_fwd: bad dims %d %d %d %d", B, h, w, C);
    const long total = (long)B * h * w * C;
    if (C % 4 == 0 && total / 4 < (1L << 31) - 256)
        hipLaunchKernelGGL(pool2_fwd_v4_kernel<T>, dim3(ceil_div(total / 4, 256)), dim3(256), 0, (hipStream_t)stream, a, o, (int)(total / 4), h, w, C / 4);
    else
        hipLaunchKernelGGL(pool2_fwd_kernel<T>, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, a, o, B, h, w, C);
    return ngan::launch_status("ngan_pool2_fwd");
}
extern "C" int ngan_pool2_fwd(const float* a, float* o, int B, int h, int w, int C, void* stream) { return pool2_fwd_impl<float>(a, o, B, h, w, C, stream); }
extern "C" int ngan_bf16_pool2_fwd(const ngan_bf16* a, ngan_bf16* o, int B, int h, int w, int C, void* stream) {
    return pool2_fwd_impl<__bf16>(BF(a), BFM(o), B, h, w, C, stream);
}

template <typename T>
static int pool2_adjoint_impl(const T* a, T* o, int B, int h, int w, int C, void* stream) {
    NGAN_REQUIRE(a && o, NGAN_ERR_ARG, "ngan_pool2_adjoint: null pointer");
    NGAN_REQUIRE(B > 0 && h > 0 && w > 0 && C > 0, NGAN_ERR_SHAPE, "ngan_pool2_adjoint: bad dims %d %d %d %d", B, h, w, C);
    hipLaunchKernelGGL(pool2_adjoint_kernel<T>, dim3(ew_blocks((long)B * 4 * h * w * C)), dim3(256), 0, (hipStream_t)stream, a, o, B, h, w, C);
    return ngan::launch_status("ngan_pool2_adjoint");
}
extern "C" int ngan_pool2_adjoint(const float* a, float* o, int B, int h, int w, int C, void* stream) {
    return pool2_adjoint_impl<float>(a, o, B, h, w, C, stream);
}
extern "C" int ngan_bf16_pool2_adjoint(const ngan_bf16* a, ngan_bf16* o, int B, int h, int w, int C, void* stream) {
    return pool2_adjoint_impl<__bf16>(BF(a), BFM(o), B, h, w, C, stream);
}

extern "C" int ngan_lerp(const float* a, const float* b, const float* alpha, float* out, long n, void* stream) {
    NGAN_REQUIRE(a && b && alpha && out && n > 0, NGAN_ERR_ARG, "lerp: bad argument");
    hipLaunchKernelGGL(lerp_kernel<float>, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, a, b, alpha, out, n);
    return ngan::launch_status("ngan_lerp");
}
extern "C" int ngan_bf16_lerp(const ngan_bf16* a, const ngan_bf16* b, const float* alpha, ngan_bf16* out, long n, void* stream) {
    NGAN_REQUIRE(a && b && alpha && out && n > 0, NGAN_ERR_ARG, "bf16_lerp: bad argument");
    hipLaunchKernelGGL(lerp_kernel<__bf16>, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, BF(a), BF(b), alpha, BFM(out), n);
    return ngan::launch_status("ngan_bf16_lerp");
}

extern "C" int ngan_axpby(const float* a, const float* b, float ca, float cb, float* out, long n, void* stream) {
    NGAN_REQUIRE(a && out && n > 0, NGAN_ERR_ARG, "axpby: bad argument");
    hipLaunchKernelGGL(axpby_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, a, b, ca, cb, out, n);
    return ngan::launch_status("ngan_axpby");
}

extern "C" int ngan_fade_bwd(const float* g, const float* alpha, float* ga, float* gb, long n, void* stream) {
    NGAN_REQUIRE(g && alpha && ga && gb && n > 0, NGAN_ERR_ARG, "fade_bwd: bad argument");
    hipLaunchKernelGGL(fade_bwd_kernel<float>, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, g, alpha, ga, gb, n);
    return ngan::launch_status("ngan_fade_bwd");
}
extern "C" int ngan_bf16_fade_bwd(const ngan_bf16* g, const float* alpha, ngan_bf16* ga, ngan_bf16* gb, long n, void* stream) {
    NGAN_REQUIRE(g && alpha && ga && gb && n > 0, NGAN_ERR_ARG, "bf16_fade_bwd: bad argument");
    hipLaunchKernelGGL(fade_bwd_kernel<__bf16>, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, BF(g), alpha, BFM(ga), BFM(gb), n);
    return ngan::launch_status("ngan_bf16_fade_bwd");
}

extern "C" int ngan_xhat(const float* real, const float* fake, const float* eps, float* out, int B, long n, void* stream) {
    NGAN_REQUIRE(real && fake && eps && out && B > 0 && n > 0, NGAN_ERR_ARG, "xhat: bad argument");
    int bx = ew_blocks(n);
    if (bx > 256) bx = 256;
    hipLaunchKernelGGL(xhat_kernel, dim3(bx, B), dim3(256), 0, (hipStream_t)stream, real, fake, eps, out, n);
    return ngan::launch_status("ngan_xhat");
}

extern "C" int ngan_scale_rows(const float* g, const float* coef, float* out, int B, long n, void* stream) {
    NGAN_REQUIRE(g && coef && out && B > 0 && n > 0, NGAN_ERR_ARG, "scale_rows: bad argument");
    int bx = ew_blocks(n);
    if (bx > 256) bx = 256;
    hipLaunchKernelGGL(scale_rows_kernel, dim3(bx, B), dim3(256), 0, (hipStream_t)stream, g, coef, out, n);
    return ngan::launch_status("ngan_scale_rows");
}

extern "C" int ngan_sample_l2norm(const float* g, float* norms, float* workspace, int B, long n, void* stream) {
    NGAN_REQUIRE(g && norms && workspace && B > 0 && n > 0, NGAN_ERR_ARG, "sample_l2norm: bad argument");
    NGAN_REQUIRE(((size_t)g & 15) == 0 && (n % 4 == 0 || B == 1), NGAN_ERR_SHAPE, "sample_l2norm: rows must be 16-byte aligned (n=%ld)", n);
    long want = (n / 4 + 255) / 256;
    const int nchunk = (int)(want < 1 ? 1 : (want > L2_CHUNKS ? L2_CHUNKS : want));
    hipLaunchKernelGGL(sample_sumsq_kernel, dim3(nchunk, B), dim3(256), 0, (hipStream_t)stream, g, workspace, n);
    hipLaunchKernelGGL(sample_l2norm_finish_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, workspace, norms, nchunk);
    return ngan::launch_status("ngan_sample_l2norm");
}
