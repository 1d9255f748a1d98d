// The critic's first layer pair as one operator:  FromImage (1x1 conv from ONE colour channel, + bias; models.py:161-165) followed
// by the block's first 3x3 convolution, LeakyReLU and PixelNorm (models.py:252-264), for the passes that are differentiated once.
//
// FromImage's output is affine in a single number per pixel, f[c] = wf[c]*p + bf[c], so the 3x3 convolution over its C channels
// collapses algebraically to a 3x3 convolution over ONE channel:
//     pre[n](x) = sum_{taps t inside the image} ( A[n][t] * p(x + t) + Bv[n][t] ),
//     A[n][t] = scale * sum_c W[n][c][t] * wf[c],     Bv[n][t] = scale * sum_c W[n][c][t] * bf[c]
// (the sum skips taps that fall into the conv's zero padding, where f is 0, not bf).  The C-channel tensor f is never written or
// read: 9 FMAs per output channel instead of a K = 16 contraction, and the layer becomes a pure stream of its output.
// Backward needs only two 9 x N tables of sums over pixels,
//     S1[n][t] = sum_x gc[n](x) * p(x + t),     S0[n][t] = sum_x gc[n](x) * [x + t inside the image],
// from which  gW[n][c][t] = scale*(wf[c]*S1[n][t] + bf[c]*S0[n][t]),  gwf[c] = scale*sum_{n,t} W[n][c][t]*S1[n][t],
// gbf[c] = scale*sum_{n,t} W[n][c][t]*S0[n][t];  and, where the image needs a gradient,  gp(x) = sum_{n,t} A[n][t]*gc[n](x - t).
#include "ngan_common.h"

namespace {

constexpr int FB_MAX_C = 64;
constexpr int FB_ROWS = 8;      // image rows per workgroup of the forward / image-gradient kernels (sliding 3-row window)

// tables[0][t][n] = A, tables[1][t][n] = Bv  (t = tap, n = output channel: a float4 of channels is one load).  One block.
__global__ __launch_bounds__(256) void first_block_tables_kernel(const float* __restrict__ W, const float* __restrict__ wf,
                                                                 const float* __restrict__ bf, float* __restrict__ tables, int N,
                                                                 int C, float scale) {
    for (int e = threadIdx.x; e < N * 9; e += blockDim.x) {
        const int t = e / N, n = e - t * N;
        float a = 0.f, b = 0.f;
        for (int c = 0; c < C; ++c) {
            const float w = W[((long)n * C + c) * 9 + t];
            a = fmaf(w, wf[c], a);
            b = fmaf(w, bf ? bf[c] : 0.f, b);
        }
        tables[e] = a * scale;
        tables[9 * N + e] = b * scale;
    }
}

// three horizontally adjacent pixels of image row sy (zeros outside the image); xx is inside the image
__device__ __forceinline__ void fb_load3(const float* __restrict__ img, int sy, int xx, int H, int Wd, float (&o)[3]) {
    const bool iny = sy >= 0 && sy < H;
    const float* r = img + (long)(iny ? sy : 0) * Wd + xx;
    o[0] = (iny && xx > 0) ? r[-1] : 0.f;
    o[1] = iny ? r[0] : 0.f;
    o[2] = (iny && xx < Wd - 1) ? r[1] : 0.f;
}

// Q = N/4 lanes per pixel (a float4 of output channels each).  grid.x: 256/Q pixels of a row; grid.y: (sample, chunk of R rows).
// The thread walks down its column with a 3x3 window of the image in registers: 3 new pixel loads per 16-byte store.
template <int Q>
__global__ __launch_bounds__(256) void first_block_fwd_kernel(const float* __restrict__ p, const float* __restrict__ tables,
                                                              const float* __restrict__ bc, float* __restrict__ y,
                                                              float* __restrict__ rn, int H, int Wd, int R, int nchunk, float slope,
                                                              float eps) {
    constexpr int N = 4 * Q;
    const int it = blockIdx.x * 256 + threadIdx.x;
    const int xx_raw = it / Q, sub = it % Q;
    const bool ok = xx_raw < Wd;
    const int xx = ok ? xx_raw : Wd - 1;
    const int b = blockIdx.y / nchunk, y0 = (blockIdx.y - b * nchunk) * R, y1 = min(y0 + R, H);
    const float* img = p + (long)b * H * Wd;
    float4 A4[9], Bcol[3];
    const bool mx[3] = {xx > 0, true, xx < Wd - 1};
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        Bcol[ky] = f4zero();                         // bias of the taps of kernel row ky that fall inside the image at this column
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            A4[ky * 3 + kx] = ld4(tables + (ky * 3 + kx) * N + sub * 4);
            const float4 bv = ld4(tables + (9 + ky * 3 + kx) * N + sub * 4);
            if (mx[kx]) Bcol[ky] = f4add(Bcol[ky], bv);
        }
    }
    const float4 bias = bc ? ld4(bc + sub * 4) : f4zero();
    float w0[3], w1[3], w2[3];
    fb_load3(img, y0 - 1, xx, H, Wd, w0);
    fb_load3(img, y0, xx, H, Wd, w1);
    for (int yy = y0; yy < y1; ++yy) {
        fb_load3(img, yy + 1, xx, H, Wd, w2);
        float4 acc = f4add(bias, Bcol[1]);
        if (yy > 0) acc = f4add(acc, Bcol[0]);
        if (yy < H - 1) acc = f4add(acc, Bcol[2]);
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            acc = f4fma(A4[kx], w0[kx], acc);
            acc = f4fma(A4[3 + kx], w1[kx], acc);
            acc = f4fma(A4[6 + kx], w2[kx], acc);
        }
        acc.x = acc.x > 0.f ? acc.x : slope * acc.x; acc.y = acc.y > 0.f ? acc.y : slope * acc.y;
        acc.z = acc.z > 0.f ? acc.z : slope * acc.z; acc.w = acc.w > 0.f ? acc.w : slope * acc.w;
        const float ss = group_sum<Q>(f4dot(acc, acc));
        const float r = sqrtf(ss / (float)N + eps);
        if (ok) {
            const long pix = ((long)b * H + yy) * Wd + xx;
            st4(y + pix * N + sub * 4, f4scale(acc, 1.0f / r));
            if (sub == 0) rn[pix] = r;
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) { w0[k] = w1[k]; w1[k] = w2[k]; }
    }
}

// sum over the PW = 64/Q consecutive lanes of a DPP row segment; the LAST lane of each segment holds the total
template <int PW>
__device__ __forceinline__ float row_tail_sum(float v) {
#define NGAN_DPP_ADD(CTRL) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true))
    NGAN_DPP_ADD(0x111);                 // row_shr:1
    NGAN_DPP_ADD(0x112);                 // row_shr:2
    NGAN_DPP_ADD(0x114);                 // row_shr:4
    if (PW == 16) NGAN_DPP_ADD(0x118);   // row_shr:8
#undef NGAN_DPP_ADD
    return v;
}

// partial S1 / S0 tables of one (sample, chunk of rows, span of columns): slab layout [2][9][N] (S1 then S0), summed afterwards.
// Lane layout inside a wave: pixel = lane % PW, channel quad = lane / PW -- the wave still reads one contiguous 1 KB of gc per
// row, and the lanes that own the same channels sit in one DPP row segment, so the block reduction is 4 DPP adds per number.
template <int Q>
__global__ __launch_bounds__(256) void first_block_sums_kernel(const float* __restrict__ p, const float* __restrict__ gc,
                                                               float* __restrict__ partial, int H, int Wd, int R, int nchunk) {
    constexpr int N = 4 * Q, PW = 64 / Q;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int xx_raw = (blockIdx.x * 4 + wave) * PW + lane % PW, sub = lane / PW;
    const bool ok = xx_raw < Wd;
    const int xx = ok ? xx_raw : Wd - 1;
    const int b = blockIdx.y / nchunk, y0 = (blockIdx.y - b * nchunk) * R, y1 = min(y0 + R, H);
    const float* img = p + (long)b * H * Wd;
    float4 s1[9], r0[3];                 // r0[ky]: sum of gc over the rows where kernel row ky is inside (column masks applied at the end)
#pragma unroll
    for (int t = 0; t < 9; ++t) s1[t] = f4zero();
    r0[0] = r0[1] = r0[2] = f4zero();
    float w0[3], w1[3], w2[3];
    fb_load3(img, y0 - 1, xx, H, Wd, w0);
    fb_load3(img, y0, xx, H, Wd, w1);
    for (int yy = y0; yy < y1; ++yy) {
        fb_load3(img, yy + 1, xx, H, Wd, w2);
        const float4 g = ok ? ld4(gc + (((long)b * H + yy) * Wd + xx) * N + sub * 4) : f4zero();
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            s1[kx] = f4fma(g, w0[kx], s1[kx]);
            s1[3 + kx] = f4fma(g, w1[kx], s1[3 + kx]);
            s1[6 + kx] = f4fma(g, w2[kx], s1[6 + kx]);
        }
        r0[1] = f4add(r0[1], g);
        if (yy > 0) r0[0] = f4add(r0[0], g);
        if (yy < H - 1) r0[2] = f4add(r0[2], g);
#pragma unroll
        for (int k = 0; k < 3; ++k) { w0[k] = w1[k]; w1[k] = w2[k]; }
    }
    __shared__ float4 red[4 * 18 * Q];
    const bool mx[3] = {xx > 0, true, xx < Wd - 1};
#pragma unroll
    for (int k = 0; k < 18; ++k) {
        float4 v = k < 9 ? s1[k] : (mx[(k - 9) % 3] ? r0[(k - 9) / 3] : f4zero());
        v.x = row_tail_sum<PW>(v.x); v.y = row_tail_sum<PW>(v.y); v.z = row_tail_sum<PW>(v.z); v.w = row_tail_sum<PW>(v.w);
        if (lane % PW == PW - 1) red[(wave * 18 + k) * Q + sub] = v;
    }
    __syncthreads();
    float* slab = partial + ((long)blockIdx.y * gridDim.x + blockIdx.x) * (2 * 9 * N);
    for (int e = tid; e < 18 * Q; e += 256) {
        const float4 s = f4add(f4add(red[e], red[18 * Q + e]), f4add(red[2 * 18 * Q + e], red[3 * 18 * Q + e]));
        st4(slab + e * 4, s);                      // e = k*Q + quad  ->  slab[k*N + 4*quad ..]
    }
}

// from the summed tables: weight.grad (+)= gW, the conv bias gradient, and the FromImage gradients.  One block.
// accumulate: bit 0 gW += , bit 1 gwf += , bit 2 gbf += , bit 3 gbc +=  (each output written or added into, e.g. straight into a .grad buffer)
__global__ __launch_bounds__(256) void first_block_finish_kernel(const float* __restrict__ S, const float* __restrict__ W,
                                                                 const float* __restrict__ wf, const float* __restrict__ bf,
                                                                 float* __restrict__ gW, float* __restrict__ gwf, float* __restrict__ gbf,
                                                                 float* __restrict__ gbc, int N, int C, float scale, int accumulate) {
    const float* S1 = S;               // [9][N]
    const float* S0 = S + 9 * N;
    for (int e = threadIdx.x; e < N * C * 9; e += blockDim.x) {
        const int t = e % 9, c = (e / 9) % C, n = e / (9 * C);
        const float v = scale * (wf[c] * S1[t * N + n] + (bf ? bf[c] : 0.f) * S0[t * N + n]);
        gW[e] = (accumulate & 1) ? gW[e] + v : v;
    }
    // FromImage gradients: 16 lanes share one channel c (each takes every 16th of the N*9 products), 16 channels per pass
    for (int c0 = 0; c0 < C; c0 += 16) {
        const int c = c0 + (threadIdx.x >> 4), part = threadIdx.x & 15;
        float a = 0.f, b = 0.f;
        if (c < C)
            for (int e = part; e < N * 9; e += 16) {
                const int n = e / 9, t = e - n * 9;
                const float w = W[((long)n * C + c) * 9 + t];
                a = fmaf(w, S1[t * N + n], a);
                b = fmaf(w, S0[t * N + n], b);
            }
        a = group_sum<16>(a);
        b = group_sum<16>(b);
        if (c < C && part == 0) {
            gwf[c] = (accumulate & 2) ? gwf[c] + a * scale : a * scale;
            if (gbf) gbf[c] = (accumulate & 4) ? gbf[c] + b * scale : b * scale;
        }
    }
    if (gbc)                                                    // conv bias: sum over pixels of gc = S0 at the centre tap
        for (int n = threadIdx.x; n < N; n += blockDim.x) gbc[n] = (accumulate & 8) ? gbc[n] + S0[4 * N + n] : S0[4 * N + n];
}

__device__ __forceinline__ void fb_load3x4(const float* __restrict__ g, int sy, int xx, int H, int Wd, int N, float4 (&o)[3]) {
    const bool iny = sy >= 0 && sy < H;
    const float* r = g + ((long)(iny ? sy : 0) * Wd + xx) * N;
    o[0] = (iny && xx > 0) ? ld4(r - N) : f4zero();
    o[1] = iny ? ld4(r) : f4zero();
    o[2] = (iny && xx < Wd - 1) ? ld4(r + N) : f4zero();
}

// gradient w.r.t. the (pooled) image: gp(x) = sum_{n,t} A[n][t] * gc[n](x - t); pool = 1: written to the 2H x 2W image * 0.25
template <int Q>
__global__ __launch_bounds__(256) void first_block_dx_kernel(const float* __restrict__ gc, const float* __restrict__ tables,
                                                             float* __restrict__ gx, int H, int Wd, int R, int nchunk, int pool) {
    constexpr int N = 4 * Q;
    const int it = blockIdx.x * 256 + threadIdx.x;
    const int xx_raw = it / Q, sub = it % Q;
    const bool ok = xx_raw < Wd;
    const int xx = ok ? xx_raw : Wd - 1;
    const int b = blockIdx.y / nchunk, y0 = (blockIdx.y - b * nchunk) * R, y1 = min(y0 + R, H);
    const float* g = gc + (long)b * H * Wd * N + sub * 4;
    float4 A4[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) A4[t] = ld4(tables + t * N + sub * 4);
    float4 w0[3], w1[3], w2[3];                     // gc rows yy-1, yy, yy+1 at columns xx-1 .. xx+1
    fb_load3x4(g, y0 - 1, xx, H, Wd, N, w0);
    fb_load3x4(g, y0, xx, H, Wd, N, w1);
    for (int yy = y0; yy < y1; ++yy) {
        fb_load3x4(g, yy + 1, xx, H, Wd, N, w2);
        float s = 0.f;                              // tap (ky, kx) was read by the output pixel at (yy - ky + 1, xx - kx + 1)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
            s += f4dot(A4[kx], w2[2 - kx]) + f4dot(A4[3 + kx], w1[2 - kx]) + f4dot(A4[6 + kx], w0[2 - kx]);
        s = group_sum<Q>(s);
        if (ok && sub == 0) {
            if (!pool) {
                gx[((long)b * H + yy) * Wd + xx] = s;
            } else {
                const long W2 = 2L * Wd;
                float* o = gx + ((long)b * 2 * H + 2 * yy) * W2 + 2 * xx;
                const float q4 = 0.25f * s;
                o[0] = q4; o[1] = q4; o[W2] = q4; o[W2 + 1] = q4;
            }
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) { w0[k] = w1[k]; w1[k] = w2[k]; }
    }
}

int fb_tables(const float* w_conv, const float* wf, const float* bf, float* tables, int N, int C, float scale, hipStream_t s) {
    hipLaunchKernelGGL(first_block_tables_kernel, dim3(1), dim3(256), 0, s, w_conv, wf, bf, tables, N, C, scale);
    return ngan::launch_status("ngan_first_block(tables)");
}

bool fb_shape_ok(int B, int H, int W, int C, int N) {
    return B > 0 && H > 0 && W > 0 && C > 0 && C <= FB_MAX_C && (N == 16 || N == 32) &&
           (long)B * ngan::ceil_div(H, FB_ROWS) < 65536 && (long)B * H * W * N < (1L << 31);
}

// rows per workgroup of the sums kernel: at most 1024 slabs
int fb_sums_rows(int B, int H, int gx) {
    int R = FB_ROWS;
    while ((long)gx * B * ngan::ceil_div(H, R) > 1024) R *= 2;
    return R;
}

}  // namespace

extern "C" size_t ngan_first_block_table_floats(int N) { return N > 0 ? (size_t)2 * 9 * N : 0; }

extern "C" int ngan_first_block_fwd(const float* p, const float* w_conv, const float* wf, const float* bf, const float* b_conv,
                                    float* y, float* rnorm, float* tables, int B, int H, int W, int C, int N, float scale,
                                    float slope, float eps, void* stream) {
    NGAN_REQUIRE(p && w_conv && wf && y && rnorm && tables, NGAN_ERR_ARG, "first_block_fwd: null pointer");
    NGAN_REQUIRE(fb_shape_ok(B, H, W, C, N), NGAN_ERR_SHAPE,
                 "first_block_fwd: B=%d H=%d W=%d C=%d N=%d unsupported (N 16 or 32, C <= 64, < 2^31 output elements)", B, H, W, C, N);
    hipStream_t s = (hipStream_t)stream;
    int st = fb_tables(w_conv, wf, bf, tables, N, C, scale, s);
    if (st) return st;
    const int nchunk = ngan::ceil_div(H, FB_ROWS);
    const dim3 grid(ngan::ceil_div((long)W * (N / 4), 256), B * nchunk), block(256);
    if (N == 16) hipLaunchKernelGGL((first_block_fwd_kernel<4>), grid, block, 0, s, p, tables, b_conv, y, rnorm, H, W, FB_ROWS, nchunk, slope, eps);
    else hipLaunchKernelGGL((first_block_fwd_kernel<8>), grid, block, 0, s, p, tables, b_conv, y, rnorm, H, W, FB_ROWS, nchunk, slope, eps);
    return ngan::launch_status("ngan_first_block_fwd");
}

extern "C" size_t ngan_first_block_workspace_floats(int B, int H, int N) {
    if (B <= 0 || H <= 0 || N <= 0) return 0;
    return (size_t)1024 * 2 * 9 * N + (size_t)2 * 9 * N;       // up to 1024 slabs + the reduced tables
}

extern "C" int ngan_first_block_bwd(const float* p, const float* gc, const float* w_conv, const float* wf, const float* bf,
                                    float* gw_conv, float* gwf, float* gbf, float* gb_conv, float* workspace, int B, int H, int W,
                                    int C, int N, float scale, int accumulate, void* stream) {
    NGAN_REQUIRE(p && gc && w_conv && wf && gw_conv && gwf && workspace, NGAN_ERR_ARG, "first_block_bwd: null pointer");
    NGAN_REQUIRE(fb_shape_ok(B, H, W, C, N), NGAN_ERR_SHAPE, "first_block_bwd: B=%d H=%d W=%d C=%d N=%d unsupported", B, H, W, C, N);
    const int gx = ngan::ceil_div((long)W * (N / 4), 256);
    const int R = fb_sums_rows(B, H, gx), nchunk = ngan::ceil_div(H, R);
    const int nblk = gx * B * nchunk;
    NGAN_REQUIRE(nblk <= 1024 && (long)B * nchunk < 65536, NGAN_ERR_SHAPE, "first_block_bwd: B=%d W=%d needs %d slabs (max 1024)", B, W, nblk);
    hipStream_t s = (hipStream_t)stream;
    float* tables = workspace + (size_t)1024 * 2 * 9 * N;
    const dim3 grid(gx, B * nchunk);
    if (N == 16) hipLaunchKernelGGL((first_block_sums_kernel<4>), grid, dim3(256), 0, s, p, gc, workspace, H, W, R, nchunk);
    else hipLaunchKernelGGL((first_block_sums_kernel<8>), grid, dim3(256), 0, s, p, gc, workspace, H, W, R, nchunk);
    int st = ngan::launch_status("ngan_first_block_bwd(sums)");
    if (st) return st;
    st = ngan::reduce_partials(workspace, nblk, 2 * 9 * N, tables, 1.0f, s);
    if (st) return st;
    hipLaunchKernelGGL(first_block_finish_kernel, dim3(1), dim3(256), 0, s, tables, w_conv, wf, bf, gw_conv, gwf, gbf, gb_conv, N, C, scale, accumulate);
    return ngan::launch_status("ngan_first_block_bwd(finish)");
}

extern "C" int ngan_first_block_dx(const float* gc, const float* tables, float* gx, int B, int H, int W, int N, int pool, void* stream) {
    NGAN_REQUIRE(gc && tables && gx, NGAN_ERR_ARG, "first_block_dx: null pointer");
    NGAN_REQUIRE(fb_shape_ok(B, H, W, 1, N), NGAN_ERR_SHAPE, "first_block_dx: B=%d H=%d W=%d N=%d unsupported", B, H, W, N);
    const int nchunk = ngan::ceil_div(H, FB_ROWS);
    const dim3 grid(ngan::ceil_div((long)W * (N / 4), 256), B * nchunk), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (N == 16) hipLaunchKernelGGL((first_block_dx_kernel<4>), grid, block, 0, s, gc, tables, gx, H, W, FB_ROWS, nchunk, pool);
    else hipLaunchKernelGGL((first_block_dx_kernel<8>), grid, block, 0, s, gc, tables, gx, H, W, FB_ROWS, nchunk, pool);
    return ngan::launch_status("ngan_first_block_dx");
}
