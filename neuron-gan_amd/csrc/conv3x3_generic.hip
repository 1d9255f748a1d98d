// 3x3 convolution (pad 1, stride 1) on channels-last fp32 tensors as an implicit GEMM on
// v_mfma_f32_16x16x4_f32 (exact fp32 MFMA, gfx950).  Replaces ATen conv2d / convolution_backward at
// /root/reference/models.py:203-204 and its autograd.
//
// Forward / input-gradient kernel ("D^T" formulation): one MFMA computes a 16(cout) x 16(pixel) tile, the
// contraction runs over (tap, cin).  A = weights (pre-packed in fragment order, pre-scaled), B = pixels read
// from an LDS-staged halo tile with one ds_read_b128 per 4 MFMAs (the k-order inside a 16-channel group is
// permuted identically on both operands, which an MFMA does not care about).  The accumulator then holds, per
// lane, 4 consecutive output channels of one pixel, so the epilogue (bias, LeakyReLU, PixelNorm) reduces over
// channels with two shuffles and stores 16 B per lane, fully coalesced.
// The avg-pool / bilinear-x2 resampling in front of a block's first conv (models.py:254, 257) is applied
// while the halo tile is staged, so the resampled tensor never exists in HBM.
#include "conv3x3_internal.h"

namespace {

// ---------------------------------------------------------------------------------------------------------
// forward / dgrad kernel.  256 threads = 4 waves arranged WP (along pixels) x WN (along output channels).
// A wave owns PGW pixel groups (16 consecutive pixels of one tile row each) x MTW 16-channel tiles; the block
// tile is NPG = WP*PGW pixel groups laid out PCG per row, and all N = 16*MTW*WN output channels.
// Large-spatial layers use WN = 1 and an 8x32 pixel tile; small-spatial / many-channel layers (128 ch at 16x16)
// split the channels over the waves and shrink the pixel tile so that the launch still has >= 256 workgroups;
// PixelNorm's channel reduction then crosses waves through LDS.
// ---------------------------------------------------------------------------------------------------------
template <int MTW, int WN, int PGW, int PCG, int RES, int EPI, int OUTMODE>
__global__ __launch_bounds__(256) void conv3x3_kernel(ConvArgs a) {
    constexpr int WP = 4 / WN, NPG = WP * PGW, TWc = PCG * 16, THc = NPG / PCG;
    constexpr int HW_ = TWc + 2, HH_ = THc + 2, MT = MTW * WN;
    constexpr int TILE_ELEMS = HH_ * HW_ * 16;
    constexpr int SS_ELEMS = (WN > 1 && EPI == 1) ? WN * NPG * 16 : 0;
    __shared__ __attribute__((aligned(16))) float smem[TILE_ELEMS + SS_ELEMS + 4];
    float* tile = smem;
    float* ss_l = smem + TILE_ELEMS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave % WN, wp = wave / WN;
    const int p = lane & 15, q = lane >> 4;
    int t = blockIdx.x;
    const int txi = t % a.tiles_x; t /= a.tiles_x;
    const int tyi = t % a.tiles_y;
    const int b = t / a.tiles_y;
    const int y0 = tyi * THc, x0 = txi * TWc;
    const int G = a.K >> 4;

    f32x4 acc[PGW][MTW];
#pragma unroll
    for (int pg = 0; pg < PGW; ++pg)
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt) acc[pg][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // Staging is split into "issue every global load" / "write LDS" so that the loads of one 16-channel group are
    // all in flight together, and the loads for group g+1 are issued before the MFMAs of group g.
    constexpr int NST = (HH_ * HW_ * 4 + 255) / 256;
    constexpr bool W_ALL_TAPS = MTW <= 2;   // 9*MTW float4 of weights fit in registers: fetch a whole group at once
    float4 stg[NST];
    auto issue_stage = [&](int g) {
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            const int e = tid + i * 256;
            const int pix = e >> 2, c4 = e & 3;
            const int ty = pix / HW_, tx = pix - ty * HW_;
            stg[i] = (e < HH_ * HW_ * 4)
                         ? load_resampled<RES>(a.x, b, y0 + ty - 1, x0 + tx - 1, g * 16 + c4 * 4, a.H, a.W, a.K)
                         : f4zero();
        }
    };
    auto wptr = [&](int g, int tap, int mt) {
        return a.wp + ((((long)tap * G + g) * MT + wn * MTW + mt) * 64 + lane) * 4;
    };
    issue_stage(0);

    for (int g = 0; g < G; ++g) {
        float4 wall[W_ALL_TAPS ? 9 : 1][MTW];
        float4 wpipe[2][MTW];
        if (W_ALL_TAPS) {
#pragma unroll
            for (int tap = 0; tap < 9; ++tap)
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt) wall[tap][mt] = ld4(wptr(g, tap, mt));
        } else {
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt) wpipe[0][mt] = ld4(wptr(g, 0, mt));
        }
        __syncthreads();   // every wave is done reading the previous group's tile
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            const int e = tid + i * 256;
            if (e < HH_ * HW_ * 4) st4(&tile[e * 4], stg[i]);
        }
        __syncthreads();
        if (g + 1 < G) issue_stage(g + 1);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3, dx = tap % 3;
            if (!W_ALL_TAPS && tap + 1 < 9) {
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt) wpipe[(tap + 1) & 1][mt] = ld4(wptr(g, tap + 1, mt));
            }
            float xv[PGW][4];
#pragma unroll
            for (int pg = 0; pg < PGW; ++pg) {
                const int pgi = wp * PGW + pg;
                const int row = pgi / PCG, col = (pgi % PCG) * 16 + p;
                float4 v = ld4(&tile[((row + dy) * HW_ + col + dx) * 16 + q * 4]);
                xv[pg][0] = v.x; xv[pg][1] = v.y; xv[pg][2] = v.z; xv[pg][3] = v.w;
            }
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt) {
                const float4 wv4 = W_ALL_TAPS ? wall[tap][mt] : wpipe[tap & 1][mt];
                const float wv[4] = {wv4.x, wv4.y, wv4.z, wv4.w};
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int pg = 0; pg < PGW; ++pg)
                        acc[pg][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[i], xv[pg][i], acc[pg][mt], 0, 0, 0);
            }
        }
    }

    // ---- epilogue: lane holds channels (wn*MTW + mt)*16 + 4q + {0..3} of pixel (row, col) ----
    float4 bv[MTW];
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt) bv[mt] = a.bias ? ld4(a.bias + (wn * MTW + mt) * 16 + q * 4) : f4zero();
    const float inv_n = 1.0f / (float)a.N;
    float4 v[PGW][MTW];
    float ssum[PGW];
#pragma unroll
    for (int pg = 0; pg < PGW; ++pg) {
        float ss = 0.f;
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt) {
            float4 c = make_float4(acc[pg][mt][0] + bv[mt].x, acc[pg][mt][1] + bv[mt].y,
                                   acc[pg][mt][2] + bv[mt].z, acc[pg][mt][3] + bv[mt].w);
            if (EPI == 1) {
                c.x = c.x > 0.f ? c.x : a.slope * c.x; c.y = c.y > 0.f ? c.y : a.slope * c.y;
                c.z = c.z > 0.f ? c.z : a.slope * c.z; c.w = c.w > 0.f ? c.w : a.slope * c.w;
                ss += f4dot(c, c);
            }
            v[pg][mt] = c;
        }
        if (EPI == 1) {
            ss = sum_rows4(ss);
        }
        ssum[pg] = ss;
    }
    if (EPI == 1 && WN > 1) {
        if (q == 0) {
#pragma unroll
            for (int pg = 0; pg < PGW; ++pg) ss_l[(wn * NPG + wp * PGW + pg) * 16 + p] = ssum[pg];
        }
        __syncthreads();
#pragma unroll
        for (int pg = 0; pg < PGW; ++pg) {
            float ss = 0.f;
#pragma unroll
            for (int w2 = 0; w2 < WN; ++w2) ss += ss_l[(w2 * NPG + wp * PGW + pg) * 16 + p];
            ssum[pg] = ss;
        }
    }
#pragma unroll
    for (int pg = 0; pg < PGW; ++pg) {
        const int pgi = wp * PGW + pg;
        const int row = pgi / PCG, col = (pgi % PCG) * 16 + p;
        const int gy = y0 + row, gx = x0 + col;
        const bool valid = gy < a.H && gx < a.W;
        if (EPI == 1) {
            const float r = sqrtf(ssum[pg] * inv_n + a.eps);
            const float inv = 1.0f / r;
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt) v[pg][mt] = f4scale(v[pg][mt], inv);
            if (valid && q == 0 && wn == 0) a.rn[((long)b * a.H + gy) * a.W + gx] = r;
        }
        if (valid) {
            const int ch0 = wn * MTW * 16 + q * 4;
            if (OUTMODE == 0) {
                float* o = a.y + (((long)b * a.H + gy) * a.W + gx) * a.N + ch0;
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt) st4(o + mt * 16, v[pg][mt]);
            } else {
                const long W2 = 2L * a.W;
                float* o = a.y + (((long)b * 2 * a.H + 2 * gy) * W2 + 2 * gx) * a.N + ch0;
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt) {
                    float4 s = f4scale(v[pg][mt], 0.25f);
                    st4(o + mt * 16, s); st4(o + a.N + mt * 16, s);
                    st4(o + W2 * a.N + mt * 16, s); st4(o + W2 * a.N + a.N + mt * 16, s);
                }
            }
        }
    }
}

template <int MTI, int CI, int RES, int EPI, int OUTMODE>
int launch_conv(ConvArgs a, hipStream_t s) {
    constexpr TileCfg c = kCfg[MTI][CI];
    int th, tw;
    cfg_tile(c, th, tw);
    a.tiles_x = ngan::ceil_div(a.W, tw);
    a.tiles_y = ngan::ceil_div(a.H, th);
    const int grid = a.B * a.tiles_x * a.tiles_y;
    hipLaunchKernelGGL((conv3x3_kernel<c.mtw, c.wn, c.pgw, c.pcg, RES, EPI, OUTMODE>), dim3(grid), dim3(256), 0, s, a);
    return ngan::launch_status("ngan_conv3x3_fwd");
}

template <int MTI, int CI>
int dispatch_conv2(const ConvArgs& a, int res, int epi, int outmode, hipStream_t s) {
    if (outmode == 1) return launch_conv<MTI, CI, 0, 0, 1>(a, s);
    switch (res * 2 + epi) {
        case 0: return launch_conv<MTI, CI, 0, 0, 0>(a, s);
        case 1: return launch_conv<MTI, CI, 0, 1, 0>(a, s);
        case 2: return launch_conv<MTI, CI, 1, 0, 0>(a, s);
        case 3: return launch_conv<MTI, CI, 1, 1, 0>(a, s);
        case 4: return launch_conv<MTI, CI, 2, 0, 0>(a, s);
        default: return launch_conv<MTI, CI, 2, 1, 0>(a, s);
    }
}

template <int MTI>
int dispatch_conv(const ConvArgs& a, int res, int epi, int outmode, hipStream_t s) {
    switch (pick_cfg(MTI, a.B, a.H, a.W)) {
        case 0: return dispatch_conv2<MTI, 0>(a, res, epi, outmode, s);
        case 1: return dispatch_conv2<MTI, 1>(a, res, epi, outmode, s);
        default: return dispatch_conv2<MTI, 2>(a, res, epi, outmode, s);
    }
}


}  // namespace

// generic exact-fp32 kernel: epilogues 0 and 1 (the caller runs a PixelNorm backward as a second launch)
int ngan::conv3x3_generic_launch(const ConvArgs& a, int resample, int epilogue, int out_mode, hipStream_t s) {
    switch (a.N / 16) {
        case 1: return dispatch_conv<0>(a, resample, epilogue, out_mode, s);
        case 2: return dispatch_conv<1>(a, resample, epilogue, out_mode, s);
        case 4: return dispatch_conv<2>(a, resample, epilogue, out_mode, s);
        default: return dispatch_conv<3>(a, resample, epilogue, out_mode, s);
    }
}
