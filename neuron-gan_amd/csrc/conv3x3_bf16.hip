// 3x3 convolution with bf16 ACTIVATION STORAGE and bf16 MFMA operands ("precision code 5", include/ngan.h): forward and -- with the
// flipped packed weights -- input gradient, for BASELINE.json's C2 "bf16" configuration.  The reference has no such mode
// (/root/reference/train.py:136-144 computes in the default dtype); what it replaces is the same Conv2d_normalized + resample + LeakyReLU
// + PixelNorm group as the fp32 kernels (/root/reference/models.py:203-204, 252-268).
//
// Arithmetic: activations are read as bf16, weights are fp32 masters rounded once to bf16 (scale folded in first) by the packing
// kernel, ONE v_mfma_f32_16x16x32_bf16 per product group (not the three of the split-bf16 mode), fp32 accumulation, fp32 epilogue
// (bias, LeakyReLU, PixelNorm statistics, PixelNorm backward, ToImage), round-to-nearest-even on the store.  The per-pixel norm is fp32.
//
// Roofline: at 2.5 PFLOP/s dense bf16 every layer of the default nets is HBM-bound (16 -> 16 at 512x512, batch 16: 8.5 us of matrix pipe
// against 268 MB = 42 us at the achievable 6.3 TB/s), so the kernel is built around its memory traffic, not the MFMA schedule:
//   * one workgroup = one 2PG x 32-pixel output tile (PG = 4, 2, 1: the launcher shrinks the tile until the grid fills the chip); several
//     workgroups per CU overlap each other's load / MFMA / store phases (11 - 22 KB of LDS and ~100 VGPRs for K, N <= 32);
//   * the halo tile is staged once into LDS as it lies in memory -- [pixel][K] bf16, 16-byte chunks, coalesced 16-byte loads -- with the
//     chunk index XOR-swizzled by the pixel column so that the B-operand fragments (lane = pixel p, k-group q: ONE ds_read_b128 of 8
//     consecutive channels) are conflict-free on ds_read_b128's four lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ...
//     (MI355X_MICROARCH.md, LDS): chunk s of column X lives at s ^ 2*((X / (16/P)) & (P/2 - 1)), P = K/8 chunks per pixel;
//   * avg-pool 2x2 and bilinear x2 inputs are resampled while the tile is staged (fp32 blend of the bf16 sources, one rounding);
//   * "D^T" form as in the fp32 kernels: A = weights (16 output channels x 32 k), B = pixels, so a lane ends with 4 consecutive
//     output channels of one pixel: an 8-byte store, 16 lanes x 4 k-groups = one pixel row segment; the PixelNorm sums are
//     v_permlane swaps over the 4 k-group rows (sum_rows4);
//   * K = 16: a contraction step is a PAIR of taps (k-groups 0,1 -> tap 2s, k-groups 2,3 -> tap 2s + 1; the 10th tap has zero
//     weights): 5 MFMAs per 16 pixels x 16 outputs;  K = 32 KS: step = tap * KS + ks covers channels 32 ks .. 32 ks + 31 of one tap;
//   * weights: S * N/16 fragments of 1 KB.  S * N/16 <= 18 (K, N <= 32): held in registers for the whole tile.  Larger: streamed
//     L2 -> registers one step ahead (these layers have a few thousand pixels: launch-latency territory, not bandwidth);
//   * workgroup -> tile: bands per XCD (blockIdx & 7) so that neighbouring tiles' halos meet in one L2.
// Resample / epilogue / store mode are run-time (wave-uniform) branches, not template parameters: 16 (K, N) pairs x 3 tile heights
// are already 48 instances.
#include "conv3x3_internal.h"

namespace ngan {
struct ConvArgsB {
    const __bf16* x; const __bf16* wp; const float* bias; __bf16* y; float* rn;
    int B, H, W, tiles_x, tiles_y, n_tiles, band;
    int resample, epilogue, out_mode;
    float slope, eps;
    const __bf16* ay; const float* wimg; const float* arn; float* aout;
};
}  // namespace ngan
using ngan::ConvArgsB;

namespace {

struct f8 { float v[8]; };
__device__ __forceinline__ f8 unpack8(uint4 u) {
    f8 r;
    r.v[0] = bf16_lo(u.x); r.v[1] = bf16_hi(u.x); r.v[2] = bf16_lo(u.y); r.v[3] = bf16_hi(u.y);
    r.v[4] = bf16_lo(u.z); r.v[5] = bf16_hi(u.z); r.v[6] = bf16_lo(u.w); r.v[7] = bf16_hi(u.w);
    return r;
}
__device__ __forceinline__ uint4 pack8(const f8& f) {
    return make_uint4(pack_bf16(f.v[0], f.v[1]), pack_bf16(f.v[2], f.v[3]), pack_bf16(f.v[4], f.v[5]), pack_bf16(f.v[6], f.v[7]));
}
__device__ __forceinline__ uint4 ld16(const __bf16* p) { return *reinterpret_cast<const uint4*>(p); }

template <int K, int N, int PG>
__global__ __launch_bounds__(256) void conv3x3_bf16_kernel(ConvArgsB a) {
    constexpr int P = K / 8, NT = N / 16, KS = K >= 32 ? K / 32 : 1, S = K == 16 ? 5 : 9 * KS;
    constexpr int TH = 2 * PG, HH = TH + 2, HW = 34, NPIX = HH * HW;
    constexpr int NCHUNK = NPIX * P, NCH = (NCHUNK + 255) / 256;
    constexpr bool WREG = S * NT <= 18;
    __shared__ uint4 tile[NPIX * P];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, p = lane & 15, q = lane >> 4;

    const int t = (blockIdx.x & 7) * a.band + (blockIdx.x >> 3);
    if (t >= a.n_tiles) return;
    const int txi = t % a.tiles_x, tyi = (t / a.tiles_x) % a.tiles_y, b = t / (a.tiles_x * a.tiles_y);
    const int y0 = tyi * TH, x0 = txi * 32;
    const int H = a.H, W = a.W;

    // column swizzle of the 16-byte chunk index (header comment)
    auto swz = [](int X) -> int { return P > 2 ? 2 * ((X / (16 / P)) & (P / 2 - 1)) : 0; };

    // ---- weights that fit the register file: fetched before the tile so that both round trips overlap
    bf16x8 wr[WREG ? S : 1][WREG ? NT : 1];
    if (WREG) {
#pragma unroll
        for (int s = 0; s < S; ++s)
#pragma unroll
            for (int j = 0; j < NT; ++j) wr[WREG ? s : 0][WREG ? j : 0] = *reinterpret_cast<const bf16x8*>(a.wp + ((long)(s * NT + j) * 64 + lane) * 8);
    }

    // ---- stage the halo tile: (2PG + 2) x 34 pixels x K channels, 16 bytes per item, resampled on the way
    constexpr int SB = NCH < 4 ? NCH : 4;          // items per thread whose loads are in flight together
    if (a.resample == NGAN_RESAMPLE_NONE) {
        const __bf16* img = a.x + (long)b * H * W * K;
#pragma unroll 1
        for (int i0 = 0; i0 < NCH; i0 += SB) {
            uint4 v[SB];
#pragma unroll
            for (int u = 0; u < SB; ++u) {
                const int e = tid + (i0 + u) * 256;
                const int pix = e / P, sl = e % P, hy = pix / HW, hx = pix % HW;
                const int gy = y0 + hy - 1, gx = x0 + hx - 1;
                const bool ok = e < NCHUNK && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
                v[u] = ok ? ld16(img + ((long)gy * W + gx) * K + sl * 8) : make_uint4(0u, 0u, 0u, 0u);
            }
#pragma unroll
            for (int u = 0; u < SB; ++u) {
                const int e = tid + (i0 + u) * 256;
                const int pix = e / P, sl = e % P, hx = pix % HW;
                if (e < NCHUNK) tile[pix * P + (sl ^ swz(hx))] = v[u];
            }
        }
    } else if (a.resample == NGAN_RESAMPLE_POOL2) {
        // x is (B, 2H, 2W, K); a staged element is the 2x2 mean, associated like ngan_pool2_fwd: 0.25 * ((a + b) + (c + d))
        const long W2 = 2L * W;
        const __bf16* img = a.x + (long)b * 4 * H * W * K;
#pragma unroll 1
        for (int i = 0; i < NCH; ++i) {
            const int e = tid + i * 256;
            const int pix = e / P, sl = e % P, hy = pix / HW, hx = pix % HW;
            const int gy = y0 + hy - 1, gx = x0 + hx - 1;
            const bool ok = e < NCHUNK && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
            uint4 o = make_uint4(0u, 0u, 0u, 0u);
            if (ok) {
                const __bf16* s0 = img + ((long)(2 * gy) * W2 + 2 * gx) * K + sl * 8;
                const f8 p00 = unpack8(ld16(s0)), p01 = unpack8(ld16(s0 + K)), p10 = unpack8(ld16(s0 + W2 * K)), p11 = unpack8(ld16(s0 + W2 * K + K));
                f8 r;
#pragma unroll
                for (int c = 0; c < 8; ++c) r.v[c] = 0.25f * ((p00.v[c] + p01.v[c]) + (p10.v[c] + p11.v[c]));
                o = pack8(r);
            }
            if (e < NCHUNK) tile[pix * P + (sl ^ swz(hx))] = o;
        }
    } else {
        // x is (B, H/2, W/2, K); bilinear x2, align_corners = False (models.py:87-89): the taps and the association of up2_fwd_kernel
        const int h = H >> 1, w = W >> 1;
        const __bf16* img = a.x + (long)b * h * w * K;
#pragma unroll 1
        for (int i = 0; i < NCH; ++i) {
            const int e = tid + i * 256;
            const int pix = e / P, sl = e % P, hy = pix / HW, hx = pix % HW;
            const int gy = y0 + hy - 1, gx = x0 + hx - 1;
            const bool ok = e < NCHUNK && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
            uint4 o = make_uint4(0u, 0u, 0u, 0u);
            if (ok) {
                int ya, yb, xa, xb; float wya, wyb, wxa, wxb;
                up2_taps(gy, h, ya, yb, wya, wyb);
                up2_taps(gx, w, xa, xb, wxa, wxb);
                const __bf16* r0 = img + (long)ya * w * K + sl * 8;
                const __bf16* r1 = img + (long)yb * w * K + sl * 8;
                const f8 t0 = unpack8(ld16(r0 + (long)xa * K)), t1 = unpack8(ld16(r0 + (long)xb * K));
                const f8 b0 = unpack8(ld16(r1 + (long)xa * K)), b1 = unpack8(ld16(r1 + (long)xb * K));
                f8 r;
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const float top = fmaf(t1.v[c], wxb, t0.v[c] * wxa), bot = fmaf(b1.v[c], wxb, b0.v[c] * wxa);
                    r.v[c] = fmaf(bot, wyb, top * wya);
                }
                o = pack8(r);
            }
            if (e < NCHUNK) tile[pix * P + (sl ^ swz(hx))] = o;
        }
    }
    __syncthreads();

    // ---- contraction.  Pixel group gi = wave * PG + pg: tile row gi >> 1, columns 16 (gi & 1) + p
    f32x4 acc[PG][NT];
#pragma unroll
    for (int pg = 0; pg < PG; ++pg)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[pg][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int prow[PG], pcol[PG];
#pragma unroll
    for (int pg = 0; pg < PG; ++pg) {
        const int gi = wave * PG + pg;
        prow[pg] = gi >> 1;
        pcol[pg] = (gi & 1) * 16 + p;
    }
    auto bfrag = [&](int pg, int dy, int dx, int sl) -> bf16x8 {
        const int X = pcol[pg] + dx;
        return __builtin_bit_cast(bf16x8, tile[((prow[pg] + dy) * HW + X) * P + (sl ^ swz(X))]);
    };
    if constexpr (WREG) {
#pragma unroll
        for (int s = 0; s < S; ++s) {
            int dy, dx, sl;
            if (K == 16) {
                int tap = 2 * s + (q >> 1);
                tap = tap > 8 ? 8 : tap;                  // the zero-weight padding tap: any valid address
                dy = (tap * 11) >> 5; dx = tap - 3 * dy; sl = q & 1;
            } else {
                dy = (s / KS) / 3; dx = (s / KS) % 3; sl = (s % KS) * 4 + q;
            }
            bf16x8 bf[PG];
#pragma unroll
            for (int pg = 0; pg < PG; ++pg) bf[pg] = bfrag(pg, dy, dx, sl);
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int pg = 0; pg < PG; ++pg) acc[pg][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wr[s][j], bf[pg], acc[pg][j], 0, 0, 0);
        }
    } else {
        // weight fragments stream L2 -> registers, one contraction step ahead of their use
        const __bf16* wnext = a.wp + (long)lane * 8;
        bf16x8 cur[NT], nxt[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) cur[j] = *reinterpret_cast<const bf16x8*>(wnext + (long)j * 512);
        // (K = 16: S = 5 tap-pair steps, the tap depends on the lane's k-group; K = 32 KS: step s = tap * KS + ks)
#pragma unroll 1
        for (int s0 = 0; s0 < S; s0 += KS) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int s = s0 + ks;
                int tap = K == 16 ? 2 * s + (q >> 1) : s0 / KS;
                tap = tap > 8 ? 8 : tap;                                    // (K = 16: the zero-weight padding tap)
                const int dy = (tap * 11) >> 5, dx = tap - 3 * dy;
                const int sn = s + 1 < S ? s + 1 : s;                       // (the last step re-reads its own fragments)
#pragma unroll
                for (int j = 0; j < NT; ++j) nxt[j] = *reinterpret_cast<const bf16x8*>(wnext + ((long)sn * NT + j) * 512);
                bf16x8 bf[PG];
#pragma unroll
                for (int pg = 0; pg < PG; ++pg) bf[pg] = bfrag(pg, dy, dx, K == 16 ? (q & 1) : ks * 4 + q);
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int pg = 0; pg < PG; ++pg) acc[pg][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cur[j], bf[pg], acc[pg][j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < NT; ++j) cur[j] = nxt[j];
            }
        }
    }

    // ---- epilogue: lane (p, q) holds channels 16 j + 4 q .. + 3 of pixel (prow, pcol) of each of its groups
    const int epi = a.epilogue;
    const float inv_n = 1.0f / (float)N;
    float4 bv[NT], wimg[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        bv[j] = a.bias ? ld4(a.bias + j * 16 + q * 4) : f4zero();
        wimg[j] = epi == EPI_TO_IMAGE ? ld4(a.wimg + j * 16 + q * 4) : f4zero();
    }
    const long img_pix = (long)b * H * W;
#pragma unroll
    for (int pg = 0; pg < PG; ++pg) {
        const int gy = y0 + prow[pg], gx = x0 + pcol[pg];
        const bool valid = gy < H && gx < W;
        const long pix = img_pix + (long)(valid ? gy : 0) * W + (valid ? gx : 0);
        float4 v[NT];
        float ss = 0.f;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            float4 c = make_float4(acc[pg][j][0] + bv[j].x, acc[pg][j][1] + bv[j].y, acc[pg][j][2] + bv[j].z, acc[pg][j][3] + bv[j].w);
            if (epi == EPI_LRELU_PN || epi == EPI_TO_IMAGE) {
                c.x = vmax1(c.x, a.slope * c.x); c.y = vmax1(c.y, a.slope * c.y);       // LeakyReLU, 0 <= slope <= 1
                c.z = vmax1(c.z, a.slope * c.z); c.w = vmax1(c.w, a.slope * c.w);
                ss += f4dot(c, c);
            }
            v[j] = c;
        }
        if (epi == EPI_LRELU_PN || epi == EPI_TO_IMAGE) {
            ss = sum_rows4(ss);
            const float m = ss * inv_n + a.eps;
            const float inv = __builtin_amdgcn_rsqf(m);
#pragma unroll
            for (int j = 0; j < NT; ++j) v[j] = f4scale(v[j], inv);
            if (a.rn && valid && q == 0) a.rn[pix] = m * inv;
            if (epi == EPI_TO_IMAGE) {
                float d = 0.f;
#pragma unroll
                for (int j = 0; j < NT; ++j) d += f4dot(v[j], wimg[j]);
                d = sum_rows4(d);
                if (valid && q == 0) a.aout[pix] = tanhf(d);
            }
        }
        if (a.out_mode == 0) {
            if (epi == EPI_PN_BWD) {
                // backward of the LeakyReLU -> PixelNorm that produced this layer's input, applied to the gradient just computed
                float4 yy[NT];
#pragma unroll
                for (int j = 0; j < NT; ++j) yy[j] = lda4(a.ay + pix * N + j * 16 + q * 4);
                const float rr = a.arn[pix];
                float s = 0.f;
#pragma unroll
                for (int j = 0; j < NT; ++j) s += f4dot(v[j], yy[j]);
                s = sum_rows4(s) * inv_n;
                const float inv_r = 1.0f / rr;
#pragma unroll
                for (int j = 0; j < NT; ++j) v[j] = pn_bwd4(v[j], yy[j], s, inv_r, a.slope);
            }
            if (a.y && valid) {
#pragma unroll
                for (int j = 0; j < NT; ++j) sta4(a.y + pix * N + j * 16 + q * 4, v[j]);
            }
        } else {
            // avg-pool adjoint store: y is (B, 2H, 2W, N), each value * 0.25 to the four pixels of its window
            const long W2 = 2L * W;
            const long o00 = 4 * img_pix + (long)(2 * (valid ? gy : 0)) * W2 + 2 * (valid ? gx : 0);
#pragma unroll
            for (int sub = 0; sub < 4; ++sub) {
                const long op = o00 + (sub >> 1) * W2 + (sub & 1);
                float4 o4[NT];
#pragma unroll
                for (int j = 0; j < NT; ++j) o4[j] = f4scale(v[j], 0.25f);
                if (epi == EPI_PN_BWD) {
                    float4 yy[NT];
                    float s = 0.f;
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        yy[j] = lda4(a.ay + op * N + j * 16 + q * 4);
                        s += f4dot(o4[j], yy[j]);
                    }
                    s = sum_rows4(s) * inv_n;
                    const float inv_r = 1.0f / a.arn[op];
#pragma unroll
                    for (int j = 0; j < NT; ++j) o4[j] = pn_bwd4(o4[j], yy[j], s, inv_r, a.slope);
                }
                if (valid) {
#pragma unroll
                    for (int j = 0; j < NT; ++j) sta4(a.y + op * N + j * 16 + q * 4, o4[j]);
                }
            }
        }
    }
}

__global__ void pack_weights_bf16_kernel(const float* __restrict__ w, __bf16* __restrict__ packed, int Cout, int Cin, int mode, float scale,
                                         long total) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < total) packed[idx] = bf16_weight(w, Cout, Cin, mode, scale, idx);
}

template <int K, int N, int PG>
int launch_bf16(ConvArgsB a, hipStream_t s) {
    a.tiles_x = ngan::ceil_div(a.W, 32);
    a.tiles_y = ngan::ceil_div(a.H, 2 * PG);
    a.n_tiles = a.B * a.tiles_x * a.tiles_y;
    a.band = ngan::ceil_div(a.n_tiles, 8);
    hipLaunchKernelGGL((conv3x3_bf16_kernel<K, N, PG>), dim3(8 * a.band), dim3(256), 0, s, a);
    return ngan::launch_status("ngan_bf16_conv3x3_fwd");
}

// tile height: 8 rows while that still gives two workgroups per CU, else 4, else 2 (layers with a few thousand pixels)
int pick_pg(int B, int H, int W) {
    const long t8 = (long)B * ngan::ceil_div(H, 8) * ngan::ceil_div(W, 32);
    if (t8 >= 512) return 4;
    const long t4 = (long)B * ngan::ceil_div(H, 4) * ngan::ceil_div(W, 32);
    return t4 >= 384 ? 2 : 1;
}

template <int K, int N>
int dispatch_pg(const ConvArgsB& a, int pg, hipStream_t s) {
    if (pg == 4) return launch_bf16<K, N, 4>(a, s);
    if (pg == 2) return launch_bf16<K, N, 2>(a, s);
    return launch_bf16<K, N, 1>(a, s);
}

template <int K>
int dispatch_n(const ConvArgsB& a, int N, int pg, hipStream_t s) {
    switch (N) {
        case 16: return dispatch_pg<K, 16>(a, pg, s);
        case 32: return dispatch_pg<K, 32>(a, pg, s);
        case 64: return dispatch_pg<K, 64>(a, pg, s);
        default: return dispatch_pg<K, 128>(a, pg, s);
    }
}

bool bf16_channels_ok(int c) { return c == 16 || c == 32 || c == 64 || c == 128; }

}  // namespace

long ngan::conv3x3_bf16_elements(int K, int N) {
    if (!bf16_channels_ok(K) || !bf16_channels_ok(N)) return 0;
    return (long)(K == 16 ? 5 : 9 * (K / 32)) * (N / 16) * 512;
}

int ngan::conv3x3_bf16_pack_launch(const float* w, float* packed, int Cout, int Cin, int mode, float scale, hipStream_t s) {
    const long tot = conv3x3_bf16_elements(mode == 0 ? Cin : Cout, mode == 0 ? Cout : Cin);
    NGAN_REQUIRE(tot > 0, NGAN_ERR_SHAPE, "conv3x3_pack_weights: precision 5 (bf16) takes 16 / 32 / 64 / 128 channels (Cin=%d, Cout=%d)", Cin, Cout);
    hipLaunchKernelGGL(pack_weights_bf16_kernel, dim3(ngan::ceil_div(tot, 256)), dim3(256), 0, s, w, reinterpret_cast<__bf16*>(packed), Cout, Cin,
                       mode, scale, tot);
    return ngan::launch_status("ngan_conv3x3_pack_weights(bf16)");
}

int ngan::conv3x3_bf16_kernel_name(int B, int H, int W, int K, int N, char* buf, int len) {
    snprintf(buf, len, "conv3x3_bf16_kernel<%d, %d, %d>", K, N, pick_pg(B, H, W));
    return NGAN_OK;
}

extern "C" int ngan_bf16_conv3x3_fwd(const ngan_bf16* x, const float* packed, const float* bias, ngan_bf16* y, float* rnorm,
                                     const void* aux_in, const float* aux_rn, float* aux_out,
                                     int B, int H, int W, int K, int N, int resample, int epilogue, int out_mode,
                                     float slope, float eps, void* stream) {
    NGAN_REQUIRE(x && packed && (y || epilogue == EPI_TO_IMAGE), NGAN_ERR_ARG, "bf16_conv3x3_fwd: null pointer");
    NGAN_REQUIRE(B > 0 && H > 0 && W > 0, NGAN_ERR_SHAPE, "bf16_conv3x3_fwd: bad dims B=%d H=%d W=%d", B, H, W);
    NGAN_REQUIRE(bf16_channels_ok(K) && bf16_channels_ok(N), NGAN_ERR_SHAPE, "bf16_conv3x3_fwd: K=%d, N=%d must be 16 / 32 / 64 / 128", K, N);
    NGAN_REQUIRE(resample >= 0 && resample <= 2, NGAN_ERR_ARG, "bf16_conv3x3_fwd: resample %d", resample);
    NGAN_REQUIRE(epilogue >= EPI_NONE && epilogue <= EPI_TO_IMAGE, NGAN_ERR_ARG, "bf16_conv3x3_fwd: epilogue %d", epilogue);
    NGAN_REQUIRE(out_mode == 0 || (out_mode == 1 && (epilogue == EPI_NONE || epilogue == EPI_PN_BWD) && resample == 0), NGAN_ERR_ARG,
                 "bf16_conv3x3_fwd: out_mode %d needs epilogue 0 or 2 and resample 0", out_mode);
    NGAN_REQUIRE(epilogue != EPI_LRELU_PN || rnorm, NGAN_ERR_ARG, "bf16_conv3x3_fwd: epilogue 1 needs rnorm");
    NGAN_REQUIRE(epilogue != EPI_PN_BWD || (aux_in && aux_rn && resample == 0 && !bias), NGAN_ERR_ARG,
                 "bf16_conv3x3_fwd: epilogue 2 needs aux_in / aux_rn, no resampling and no bias");
    NGAN_REQUIRE(epilogue != EPI_TO_IMAGE || (aux_in && aux_out && (!y || rnorm)), NGAN_ERR_ARG,
                 "bf16_conv3x3_fwd: epilogue 3 needs aux_in (the colour weights, fp32), aux_out, and rnorm whenever y is stored");
    NGAN_REQUIRE(resample != NGAN_RESAMPLE_UP2 || (H % 2 == 0 && W % 2 == 0), NGAN_ERR_SHAPE, "bf16_conv3x3_fwd: bilinear x2 needs even H, W");
    ConvArgsB a{reinterpret_cast<const __bf16*>(x), reinterpret_cast<const __bf16*>(packed), bias, reinterpret_cast<__bf16*>(y),
                (epilogue == EPI_LRELU_PN || epilogue == EPI_TO_IMAGE) ? rnorm : nullptr, B, H, W, 0, 0, 0, 0, resample, epilogue, out_mode, slope, eps,
                epilogue == EPI_PN_BWD ? reinterpret_cast<const __bf16*>(aux_in) : nullptr,
                epilogue == EPI_TO_IMAGE ? reinterpret_cast<const float*>(aux_in) : nullptr, aux_rn, aux_out};
    hipStream_t s = (hipStream_t)stream;
    const int pg = pick_pg(B, H, W);
    switch (K) {
        case 16: return dispatch_n<16>(a, N, pg, s);
        case 32: return dispatch_n<32>(a, N, pg, s);
        case 64: return dispatch_n<64>(a, N, pg, s);
        default: return dispatch_n<128>(a, N, pg, s);
    }
}
