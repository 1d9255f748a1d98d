#include "conv3x3_internal.h"

namespace {

// ---------------------------------------------------------------------------------------------------------
// conv3x3(bilinear_x2(x)) with the up-sampling folded into four parity weight sets (split-bf16, N = 16 outputs, K = 16 / 32).
// The tile is 8 x 32 OUTPUT pixels = 4 x 16 low-resolution pixels; only their 6 x 18 halo patch is staged (split into bf16 hi/lo
// once per low-res value) -- there is no expansion to the output resolution at all.  Wave w owns parity (py, px) = (w >> 1, w & 1):
// its weight set lives in registers, its 4 pixel groups are the 4 low-res rows, and its B operands are read from the same patch
// cells as the other waves'.  (Measured alternatives for K = 16: 4 workgroups per CU 180-195 us instead of 167; one low-res row per
// wave with all four weight sets in registers -- 4x fewer LDS reads, 2 workgroups per CU -- 190-197 us.)  Output pixels on the image border (whose 3x3 window reaches into the conv's zero padding, which
// the folded weights cannot express) are skipped here and written by conv3x3_up2_border_kernel.
// ---------------------------------------------------------------------------------------------------------
template <int KG, int EPI>
__global__ __launch_bounds__(256, KG == 1 ? 3 : 2) void conv3x3_up2f_kernel(ConvArgs a, int n_tiles) {
    constexpr int K = KG * 16, N = 16, LP = 24, PH = 6, PW = 18, NPP = PH * PW;
    constexpr int NSTEP = KG == 1 ? 5 : 9;
    constexpr int PLANE = PH * LP * 16, TILE_ELEMS = KG * PLANE;
    constexpr int N_SRC = KG * NPP * 4, NST = (N_SRC + 255) / 256;
    __shared__ __attribute__((aligned(16))) float tile[TILE_ELEMS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int p = lane & 15, q = lane >> 4;
    const int py = wave >> 1, px = wave & 1;
    const int h = a.H >> 1, w = a.W >> 1;

    bf16x8 wh[NSTEP], wlo[NSTEP];
    {
        const float* wset = a.wp + (long)wave * (NSTEP * 2 * 256);
#pragma unroll
        for (int st = 0; st < NSTEP; ++st) {
            float4 h4 = ld4(wset + (st * 2 + 0) * 256 + lane * 4), l4 = ld4(wset + (st * 2 + 1) * 256 + lane * 4);
            pin_registers(h4);                 // loaded once: not to be re-loaded per tile
            pin_registers(l4);
            wh[st] = __builtin_bit_cast(bf16x8, h4);
            wlo[st] = __builtin_bit_cast(bf16x8, l4);
        }
    }
    const TileRun run = tile_run(n_tiles);
    int t = run.t;
    const int t_end = run.t_end;

    int s_ty[NST], s_tx[NST], s_ch[NST], s_lds[NST];
#pragma unroll
    for (int i = 0; i < NST; ++i) {
        const int e = tid + i * 256;
        const int c4 = e & 3, pp = (e >> 2) % NPP, g = (e >> 2) / NPP;
        s_ty[i] = pp / PW - 1; s_tx[i] = pp % PW - 1; s_ch[i] = (g * 16 + c4 * 4) * 4;
        s_lds[i] = bf16_slot<KG, PLANE, LP>(g, c4, pp / PW, pp % PW);
    }
    int rs[NSTEP];
#pragma unroll
    for (int st = 0; st < NSTEP; ++st) {
        int tap = KG == 1 ? 2 * st + (q >> 1) : st;
        if (tap > 8) tap = 8;                                  // zero-weight padding tap: any valid address
        const int dy = tap / 3, dx = tap % 3;
        const int slot = (KG == 1 ? (q & 1) : q) ^ ((((p + dx) >> 2) & 1) << 1);
        rs[st] = (dy * LP + p + dx) * 16 + slot * 4;
    }
    auto decode = [&](int tt, int& b, int& y0, int& x0) {
        const int txi = tt % a.tiles_x; tt /= a.tiles_x;
        const int tyi = tt % a.tiles_y;
        b = tt / a.tiles_y;
        y0 = tyi * 8; x0 = txi * 32;
    };
    float4 stg[NST];
    const unsigned src_img_bytes = (unsigned)(h * w * K) * 4u;
    constexpr unsigned OOB = 0xFFFFFFF0u;
    auto issue = [&](int tt) {                 // always called unconditionally (see the persistent kernel)
        int b, y0, x0;
        decode(tt, b, y0, x0);
        const float* base = a.x + (long)b * h * w * K;
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, src_img_bytes, 0x00020000);
        const int ly0 = y0 >> 1, lx0 = x0 >> 1;
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            const int ly = min(max(ly0 + s_ty[i], 0), h - 1), lx = min(max(lx0 + s_tx[i], 0), w - 1);   // the bilinear taps clamp
            const unsigned off = (tid + i * 256 < N_SRC) ? (unsigned)((ly * w + lx) * K * 4 + s_ch[i]) : OOB;
            stg[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0));
        }
    };
    if (t >= t_end) return;                                // (uniform over the workgroup)
    issue(t);
#pragma unroll
    for (int i = 0; i < NST; ++i) pin_registers(stg[i]);   // (no load is pending on entry to the loop: see pin_registers)
    float4 bv = a.bias ? ld4(a.bias + q * 4) : f4zero();
    pin_registers(bv);                                     // (awaited once, here: conv3x3_internal.h)
    const float inv_n = 1.0f / (float)N;

    while (t < t_end) {
        int b, y0, x0;
        decode(t, b, y0, x0);
        __syncthreads();   // the previous tile's MFMAs have finished reading `tile`
#pragma unroll
        for (int i = 0; i < NST; ++i)
            if (tid + i * 256 < N_SRC) st_split<KG, PLANE>(tile, s_lds[i], stg[i]);
        __syncthreads();
        const int tn = t + run.step;
        issue(min(tn, t_end - 1));   // in flight while this tile is computed
        __builtin_amdgcn_sched_barrier(0);   // (the scheduler would otherwise sink the loads to their first use, behind the MFMAs)
        f32x4 acc[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int st = 0; st < NSTEP; ++st) {
            bf16x8 xh[4], xl[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int base = r * LP * 16 + rs[st];
                xh[r] = *reinterpret_cast<const bf16x8*>(&tile[base]);
                xl[r] = *reinterpret_cast<const bf16x8*>(&tile[KG == 1 ? (base ^ 8) : (base + PLANE)]);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wlo[st], xh[r], acc[r], 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[st], xl[r], acc[r], 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[st], xh[r], acc[r], 0, 0, 0);
        }
        {
            float dep = acc[3][3];
#pragma unroll
            for (int i = 0; i < NST; ++i) pin_registers_after(stg[i], dep);   // next tile's loads land before this tile's stores go out
            acc[3][3] = dep;
        }
        const long img = (long)b * a.H * a.W;
        const __amdgpu_buffer_rsrc_t y_rsrc = __builtin_amdgcn_make_buffer_rsrc(a.y + img * N, 0, (unsigned)(a.H * a.W * N) * 4u, 0x00020000);
        __amdgpu_buffer_rsrc_t rn_rsrc;
        if (EPI == EPI_LRELU_PN) rn_rsrc = __builtin_amdgcn_make_buffer_rsrc(a.rn + img, 0, (unsigned)(a.H * a.W) * 4u, 0x00020000);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int gy = y0 + 2 * r + py, gx = x0 + 2 * p + px;
            // interior pixels only: the border ring belongs to conv3x3_up2_border_kernel
            const bool valid = gy > 0 && gy < a.H - 1 && gx > 0 && gx < a.W - 1;
            float4 c = make_float4(acc[r][0] + bv.x, acc[r][1] + bv.y, acc[r][2] + bv.z, acc[r][3] + bv.w);
            if (EPI == EPI_LRELU_PN) {
                c.x = vmax1(c.x, a.slope * c.x); c.y = vmax1(c.y, a.slope * c.y);
                c.z = vmax1(c.z, a.slope * c.z); c.w = vmax1(c.w, a.slope * c.w);
                float ss = f4dot(c, c);
                ss = sum_rows4(ss);
                const float m = ss * inv_n + a.eps;
                const float inv = __builtin_amdgcn_rsqf(m);
                c = f4scale(c, inv);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, m * inv), rn_rsrc, (valid && q == 0) ? (unsigned)((gy * a.W + gx) * 4) : OOB, 0, 0);
            }
            const unsigned off = valid ? (unsigned)(((gy * a.W + gx) * N + q * 4) * 4) : OOB;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, c), y_rsrc, off, 0, 0);
        }
        t = tn;
    }
}

// The border ring of conv3x3(bilinear_x2(x)) (rows 0, H-1, columns 0, W-1) in exact fp32 from the unfolded weights: 16 pixels per
// workgroup, thread = (pixel, output channel); the 3x3 up-sampled window of each pixel is staged in LDS (zeros outside the image).
template <int EPI, int K>
__global__ __launch_bounds__(256) void conv3x3_up2_border_kernel(ConvArgs a, const float* __restrict__ wraw) {
    constexpr int N = 16, KQ = K / 4;
    constexpr int NW = N * K * 9 / 256;                 // weight elements per thread (9 or 18, exact)
    constexpr int NU = (16 * 9 * KQ + 255) / 256;       // window quads per thread (3 or 5)
    __shared__ __attribute__((aligned(16))) float u[16 * 9 * K];   // [16 px][9 taps][K]
    __shared__ float wt[9 * K * 16];                                // [tap][k][n]
    const int tid = threadIdx.x, b = blockIdx.y;
    const int nborder = 2 * a.W + 2 * (a.H - 2);
    auto pixel = [&](int idx, int& gy, int& gx) {
        if (idx < a.W) { gy = 0; gx = idx; }
        else if (idx < 2 * a.W) { gy = a.H - 1; gx = idx - a.W; }
        else { const int j = idx - 2 * a.W; gy = 1 + (j >> 1); gx = (j & 1) ? a.W - 1 : 0; }
    };
    // every global load of the workgroup is issued before the first LDS store: one memory round trip, not one per loop iteration
    float wv[NW];
#pragma unroll
    for (int i = 0; i < NW; ++i) wv[i] = wraw[tid + i * 256];
    float4 uv[NU];
#pragma unroll
    for (int i = 0; i < NU; ++i) {
        const int e = tid + i * 256;
        const int cq = e % KQ, tap = (e / KQ) % 9, pxi = e / (9 * KQ);
        const int idx = blockIdx.x * 16 + pxi;
        uv[i] = f4zero();
        if (e < 16 * 9 * KQ && idx < nborder) {
            int gy, gx;
            pixel(idx, gy, gx);
            uv[i] = load_resampled<NGAN_RESAMPLE_UP2>(a.x, b, gy + tap / 3 - 1, gx + tap % 3 - 1, cq * 4, a.H, a.W, K);
        }
    }
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        const int e = tid + i * 256;
        const int tap = e % 9, k = (e / 9) % K, n = e / (9 * K);
        wt[(tap * K + k) * 16 + n] = wv[i];
    }
#pragma unroll
    for (int i = 0; i < NU; ++i) {
        const int e = tid + i * 256;
        if (e < 16 * 9 * KQ) st4(&u[e * 4], uv[i]);     // e = (pxi*9 + tap)*KQ + cq  ->  u[(pxi*9 + tap)*K + 4*cq]
    }
    __syncthreads();
    const int pxi = tid >> 4, n = tid & 15;
    const int idx = blockIdx.x * 16 + pxi;
    // four independent partial sums, four contraction indices per LDS read of the window
    float c0 = a.bias ? a.bias[n] : 0.f, c1 = 0.f, c2 = 0.f, c3 = 0.f;
    const float4* u4 = reinterpret_cast<const float4*>(u + pxi * 9 * K);
#pragma unroll 6
    for (int t4 = 0; t4 < 9 * K / 4; ++t4) {
        const float4 x4 = u4[t4];
        const float* wp = wt + t4 * 64 + n;
        c0 = fmaf(x4.x, wp[0], c0); c1 = fmaf(x4.y, wp[16], c1);
        c2 = fmaf(x4.z, wp[32], c2); c3 = fmaf(x4.w, wp[48], c3);
    }
    float c = (c0 + c1) + (c2 + c3);
    float r = 1.f;
    if (EPI == EPI_LRELU_PN) {
        c = vmax1(c, a.slope * c);
        const float ss = group_sum<16>(c * c);
        r = sqrtf(ss / (float)N + a.eps);
        c /= r;
    }
    if (idx < nborder) {
        int gy, gx;
        pixel(idx, gy, gx);
        const long pix = ((long)b * a.H + gy) * a.W + gx;
        a.y[pix * N + n] = c;
        if (EPI == EPI_LRELU_PN && n == 0) a.rn[pix] = r;
    }
}

template <int KG, int EPI>
int launch_up2f(ConvArgs a, hipStream_t s) {
    a.tiles_x = ngan::ceil_div(a.W, 32);
    a.tiles_y = ngan::ceil_div(a.H, 8);
    const int n_tiles = a.B * a.tiles_x * a.tiles_y;
    static int per_cu = 0;
    if (per_cu == 0) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, conv3x3_up2f_kernel<KG, EPI>, 256, 0) != hipSuccess || n < 1) n = 1;
        per_cu = n > 4 ? 4 : n;
    }
    const int grid = persistent_grid(n_tiles, 256 * per_cu);
    hipLaunchKernelGGL((conv3x3_up2f_kernel<KG, EPI>), dim3(grid), dim3(256), 0, s, a, n_tiles);
    return ngan::launch_status("ngan_conv3x3_fwd(bilinear folded)");
}

template <int KG, int EPI>
int launch_up2_border(const ConvArgs& a, hipStream_t s) {
    constexpr int NSTEP = KG == 1 ? 5 : 9;
    const float* wraw = a.wp + 4 * (NSTEP * 2 * 256);           // the scaled OIHW weights behind the four folded sets
    const int nborder = 2 * a.W + 2 * (a.H - 2);
    hipLaunchKernelGGL((conv3x3_up2_border_kernel<EPI, KG * 16>), dim3(ngan::ceil_div(nborder, 16), a.B), dim3(256), 0, s, a, wraw);
    return ngan::launch_status("ngan_conv3x3_up2_border");
}

int dispatch_up2_border(const ConvArgs& a, int epilogue, hipStream_t s) {
    if (a.K == 16) return epilogue ? launch_up2_border<1, EPI_LRELU_PN>(a, s) : launch_up2_border<1, EPI_NONE>(a, s);
    return epilogue ? launch_up2_border<2, EPI_LRELU_PN>(a, s) : launch_up2_border<2, EPI_NONE>(a, s);
}

}  // namespace

int ngan::conv3x3_up2f_launch(const ConvArgs& a, int epilogue, hipStream_t s) {
    return a.K == 16 ? (epilogue ? launch_up2f<1, EPI_LRELU_PN>(a, s) : launch_up2f<1, EPI_NONE>(a, s))
                     : (epilogue ? launch_up2f<2, EPI_LRELU_PN>(a, s) : launch_up2f<2, EPI_NONE>(a, s));
}

int ngan::conv3x3_up2_border_launch(const ConvArgs& a, int epilogue, hipStream_t s) { return dispatch_up2_border(a, epilogue, s); }
