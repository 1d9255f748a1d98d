// Split-bf16 ("bf16x3") 3x3 convolution for the many-channel layers on small images (K, N in {32, 64, 96, 128}; the
// 64/128-channel blocks at 16x16 .. 64x64 of /root/reference/models.py:299-329, 469-503), forward and input-gradient.
//
// Why a separate kernel: these layers have only a few thousand pixels, so a launch that fills the chip gives every wave
// one or two 16x16 MFMA tiles and a K = 9*Cin contraction of up to 1152.  In exact fp32 that chain is MFMA-bound at
// ~10 us and, worse, exposes one L2 round trip per 16-channel group (measured 21-46 us per launch).  Here
//   * the whole halo tile (6 x 18 pixels, ALL input channels) is staged into LDS once, already split into bf16 hi/lo;
//   * a wave streams its weight fragments L2 -> registers through a ring of D+1 register buffers, D steps ahead of the MFMAs
//     (weights of one layer are <= 590 KB and are shared by every workgroup: they live in L2);
//   * the 9*K/32 contraction steps are fully unrolled: 3 v_mfma_f32_16x16x32_bf16 per fp32 product group (16 cycles each
//     instead of 8 x 32 for fp32);
//   * when the image has too few 4x16-pixel tiles to fill the chip, the OUTPUT CHANNELS are split over workgroups
//     (blockIdx.y); PixelNorm then needs all channels of a pixel and runs as a second, tiny launch.
// LDS image: pixel-major, per pixel [32-channel group][hi 32 x bf16 | lo 32 x bf16] + 16 B of padding, which makes the pixel
// pitch = 4 (mod 64) dwords: the 16 lanes of a ds_read_b128 phase (16 consecutive pixels, same channel octet) hit 16
// distinct bank quads.
#include <type_traits>
#include "conv3x3_shared.h"

namespace {

__device__ __forceinline__ float4 f4select(bool ok, float4 v) {
    return make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
}

// load_resampled (conv3x3_shared.h) without the padding test: (gy, gx) must already be inside the H x W conv input
template <int RES>
__device__ __forceinline__ float4 load_inside(const float* __restrict__ x, int b, int gy, int gx, int ch, int H, int W, int C) {
    if (RES == NGAN_RESAMPLE_NONE) {
        return ld4(x + (((long)b * H + gy) * W + gx) * C + ch);
    } else if (RES == NGAN_RESAMPLE_POOL2) {
        const long W2 = 2L * W;
        const float* p = x + (((long)b * 2 * H + 2 * gy) * W2 + 2 * gx) * C + ch;
        return f4scale(f4add(f4add(ld4(p), ld4(p + C)), f4add(ld4(p + W2 * C), ld4(p + W2 * C + C))), 0.25f);
    } else {
        const int h = H >> 1, w = W >> 1;
        int y0, y1, x0, x1; float wy0, wy1, wx0, wx1;
        up2_taps(gy, h, y0, y1, wy0, wy1);
        up2_taps(gx, w, x0, x1, wx0, wx1);
        const float* r0 = x + ((long)b * h + y0) * w * C + ch;
        const float* r1 = x + ((long)b * h + y1) * w * C + ch;
        const float4 top = f4fma(ld4(r0 + (long)x1 * C), wx1, f4scale(ld4(r0 + (long)x0 * C), wx0));
        const float4 bot = f4fma(ld4(r1 + (long)x1 * C), wx1, f4scale(ld4(r1 + (long)x0 * C), wx0));
        return f4fma(bot, wy1, f4scale(top, wy0));
    }
}

// PREC = 1: split bf16 as described above.  PREC = 0: the SAME tiling in exact fp32 -- the LDS pixel holds the K floats as they are,
// a contraction step is one tap of a 16-channel group (one ds_read_b128 per pixel group, one 16-byte weight fragment per channel
// tile in the layout of pack_weights_kernel, 4 x v_mfma_f32_16x16x4_f32 per tile), and the weight ring has 9 slots (slot = tap),
// filled 8 steps ahead, so the loop over channel groups needs no unrolling beyond a pair of groups.
template <int KG, int PGW, int MTW, int WN, int PREC = 1>
struct MidCfg {
    static constexpr int WP = 4 / WN;
    static constexpr int K = KG * 32, NS = 16 * MTW * WN;
    static constexpr int PITCH = K * 4 + 16;             // bytes per LDS pixel
    static constexpr int NPIX = 6 * 18;
    static constexpr int NSTEP = PREC ? 9 * KG : 18 * KG;
    static constexpr int D = !PREC ? 8 : (MTW * PGW >= 8) ? 3 : (MTW * PGW >= 4 ? 5 : 8);   // weight prefetch distance, in steps
    static_assert(WP * PGW == 4, "a workgroup covers 4 pixel groups (4 rows x 16 pixels)");
};

// KS = 2: the contraction steps are dealt to two groups of 4 waves (512 threads per workgroup) and the two partial accumulators
// are exchanged through LDS.  With one 4-wave workgroup per CU (all these layers offer) every SIMD hosts ONE wave, whose staging
// VALU, MFMAs and waits simply add up (PMC: 27 % + 24 % + 38 % of the wave's lifetime); two waves per SIMD overlap them and each
// has half the serial chain.
template <int KG, int PGW, int MTW, int WN, int EPI, int OUTMODE, int KS, int PREC>
__global__ __launch_bounds__(256 * KS) void conv3x3_mid_kernel(ConvArgs a, int resample) {
    using C = MidCfg<KG, PGW, MTW, WN, PREC>;
    static_assert(PREC == 1 || KS == 1, "the fp32 variant has no contraction split");
    constexpr int K = C::K, PITCH = C::PITCH, NPIX = C::NPIX, NSTEP = C::NSTEP, D = C::D, NT = 256 * KS;
    static_assert(NSTEP % KS == 0, "the contraction steps must split evenly");
    constexpr int OWN = NSTEP / KS;                                     // steps of one wave group
    constexpr int SS_BYTES = (EPI != EPI_NONE && WN > 1) ? WN * 4 * 16 * 4 : 0;
    constexpr int XCH_BYTES = KS > 1 ? 4 * KS * PGW * MTW * 1024 : 0;   // accumulator exchange (re-uses the tile region)
    constexpr int TILE_BYTES = NPIX * PITCH > XCH_BYTES ? NPIX * PITCH : XCH_BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char smem[TILE_BYTES + SS_BYTES];
    float* ss_l = reinterpret_cast<float*>(smem + TILE_BYTES);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kh = wave >> 2, w4 = wave & 3;                            // wave group (contraction half), wave inside the group
    const int wn = w4 % WN, wp = w4 / WN;
    const int p = lane & 15, q = lane >> 4;
    int t = blockIdx.x;
    const int txi = t % a.tiles_x; t /= a.tiles_x;
    const int tyi = t % a.tiles_y;
    const int b = t / a.tiles_y;
    const int y0 = tyi * 4, x0 = txi * 16;
    const int MT = a.N >> 4;
    const int mt0 = blockIdx.y * (C::NS / 16) + wn * MTW;      // this wave's first 16-channel output tile
    constexpr unsigned OOB = 0xFFFFFFF0u;

    // ---- weight stream: fragment (step, mt, part) is 64 lanes x 16 B at byte ((step*MT + mt)*2 + part)*1024 + lane*16.  Buffer
    // loads: the lane part is a constant voffset, the (step, mt, part) part a scalar offset -- no per-load vector address math ----
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wp), 0, (unsigned)(9 * K * a.N) * 4u, 0x00020000);   // K, not a.K: the packed layout
    const unsigned w_voff = (unsigned)(mt0 * (PREC ? 2048 : 1024) + lane * 16);
    bf16x8 wr[PREC ? D + 1 : 1][MTW][2];
    auto wload = [&](int slot, int own_step) {
        const int step = own_step * KS + kh;                            // wave group kh takes steps kh, kh + KS, ...
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt) {
            const int soff = (step * MT + mt) * 2048;
            wr[slot][mt][0] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, w_voff, soff, 0));
            wr[slot][mt][1] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, w_voff, soff + 1024, 0));
        }
    };
    // fp32: fragment (tap, g, mt) of pack_weights_kernel is 64 lanes x 16 B at byte ((tap*G + g)*MT + mt)*1024 + lane*16; a fragment
    // index past the last channel group lands on another valid fragment or beyond the descriptor (zeros) and is never consumed
    constexpr int G = 2 * KG;
    f32x4 wf[PREC ? 1 : 9][MTW];
    auto wload_f32 = [&](int tap, int g) {
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt)
            wf[tap][mt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, w_voff, ((tap * G + g) * MT + mt) * 1024, 0));
    };
    if (PREC) {
#pragma unroll
        for (int s = 0; s < D && s < OWN; ++s) wload(s, s);
    } else {
#pragma unroll
        for (int tap = 0; tap < 8; ++tap) wload_f32(tap, 0);
    }

    // ---- stage the halo tile, split into bf16 hi / lo.  Branch-free: plain input through a buffer descriptor of the image (an
    // out-of-range offset = conv padding / unused slot reads zeros); resampled input from a clamped address, zeroed afterwards ----
    constexpr int CQ = K / 4, NITEM = NPIX * CQ, NST = (NITEM + NT - 1) / NT;
    auto to_lds = [&](int e, float4 v) {
        const int pix = e / CQ, c4 = e % CQ;
        if (!PREC) {
            *reinterpret_cast<float4*>(smem + pix * PITCH + c4 * 16) = v;
            return;
        }
        bf16x4 hi, lo;
        hi[0] = (__bf16)v.x; hi[1] = (__bf16)v.y; hi[2] = (__bf16)v.z; hi[3] = (__bf16)v.w;
        lo[0] = (__bf16)(v.x - (float)hi[0]); lo[1] = (__bf16)(v.y - (float)hi[1]);
        lo[2] = (__bf16)(v.z - (float)hi[2]); lo[3] = (__bf16)(v.w - (float)hi[3]);
        unsigned char* dst = smem + pix * PITCH + (c4 >> 3) * 128 + (c4 & 7) * 8;
        *reinterpret_cast<bf16x4*>(dst) = hi;
        *reinterpret_cast<bf16x4*>(dst + 64) = lo;
    };
    // Plain input: every item of the thread is requested before the first one is written (NST x 16 B in flight per lane).  Resampled
    // input has four sources per item; requested all at once they were the kernel's register maximum (K = 128: 14 x 4 x 16 B per lane =
    // 224 VGPRs + the weight ring: 256 with spills, one workgroup per CU whatever the main loop needs) -- they go in rounds of CH items
    auto stage = [&](auto res_tag) {
        constexpr int RES = decltype(res_tag)::value;
        constexpr int CH = RES == NGAN_RESAMPLE_NONE ? NST : (NST < 4 ? NST : 4);
        // a.K is the tensor's channel count; it is smaller than the kernel's K only for a 16-channel input padded to 32 (zeros)
        const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x + (long)b * a.H * a.W * a.K), 0,
                                                                                 (unsigned)(a.H * a.W * a.K) * 4u, 0x00020000);
#pragma unroll
        for (int i0 = 0; i0 < NST; i0 += CH) {
            float4 stg[CH];
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                const int e = tid + (i0 + i) * NT;
                const int pix = e / CQ, c4 = e % CQ;
                const int ty = pix / 18, tx = pix - ty * 18;
                const int gy = y0 + ty - 1, gx = x0 + tx - 1;
                const bool ok = i0 + i < NST && e < NITEM && c4 * 4 < a.K && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
                if (RES == NGAN_RESAMPLE_NONE) {
                    const unsigned off = ok ? (unsigned)(((gy * a.W + gx) * a.K + c4 * 4) * 4) : OOB;
                    stg[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, off, 0, 0));
                } else {
                    const int cy = min(max(gy, 0), a.H - 1), cx = min(max(gx, 0), a.W - 1);
                    stg[i] = f4select(ok, load_inside<RES>(a.x, b, cy, cx, min(c4 * 4, a.K - 4), a.H, a.W, a.K));
                }
            }
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                const int e = tid + (i0 + i) * NT;
                if (i0 + i < NST && e < NITEM) to_lds(e, stg[i]);
            }
            if (RES != NGAN_RESAMPLE_NONE) __builtin_amdgcn_sched_barrier(0);      // keep the rounds apart: the scheduler would hoist every load again
        }
    };
    if (resample == NGAN_RESAMPLE_NONE) stage(std::integral_constant<int, NGAN_RESAMPLE_NONE>());
    else if (resample == NGAN_RESAMPLE_POOL2) stage(std::integral_constant<int, NGAN_RESAMPLE_POOL2>());
    else stage(std::integral_constant<int, NGAN_RESAMPLE_UP2>());
    __syncthreads();

    f32x4 acc[PGW][MTW];
#pragma unroll
    for (int pg = 0; pg < PGW; ++pg)
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt) acc[pg][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // this lane's B-operand base: pixel (wp*PGW, p), channel octet q
    const unsigned char* xb = smem + ((wp * PGW) * 18 + p) * PITCH + q * 16;
    if (!PREC) {
        // lane (p, q) reads channels 16 g + 4 q + {0..3} of its pixels: component j is the B operand of MFMA j of the step, matching
        // weight component j (ci = 16 g + 4 q + j) -- the four MFMAs cover the 16 channels in a permuted order, which a sum does not see
        f32x4 xv[2][PGW];
        auto xload_f32 = [&](int slot, int g, int tap) {
            const int dy = tap / 3, dx = tap % 3;
#pragma unroll
            for (int pg = 0; pg < PGW; ++pg) xv[slot][pg] = *reinterpret_cast<const f32x4*>(xb + ((pg + dy) * 18 + dx) * PITCH + g * 64);
        };
        xload_f32(0, 0, 0);
#pragma unroll 1
        for (int g2 = 0; g2 < KG; ++g2) {
#pragma unroll
            for (int u = 0; u < 18; ++u) {
                const int g = 2 * g2 + u / 9, tap = u % 9;
                wload_f32((tap + 8) % 9, tap == 0 ? g : g + 1);                 // step + 8 into the slot consumed one step ago
                if (u < 17) xload_f32((u + 1) & 1, 2 * g2 + (u + 1) / 9, (u + 1) % 9);
                else if (g2 + 1 < KG) xload_f32(0, 2 * g2 + 2, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
                        for (int pg = 0; pg < PGW; ++pg)
                            acc[pg][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[tap][mt][j], xv[u & 1][pg][j], acc[pg][mt], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    bf16x8 xh[2][PGW], xl[2][PGW];      // B operands, read from LDS one step ahead of their MFMAs
    auto xload = [&](int slot, int own_step) {
        const int step = own_step * KS + kh;
        const int kg = step / 9, tap = step % 9, dy = tap / 3, dx = tap % 3;
#pragma unroll
        for (int pg = 0; pg < PGW; ++pg) {
            const unsigned char* src = xb + ((pg + dy) * 18 + dx) * PITCH + kg * 128;
            xh[slot][pg] = *reinterpret_cast<const bf16x8*>(src);
            xl[slot][pg] = *reinterpret_cast<const bf16x8*>(src + 64);
        }
    };
    if (PREC) xload(0, 0);
#pragma unroll
    for (int s = 0; s < (PREC ? OWN : 0); ++s) {
        if (s + D < OWN) wload((s + D) % (D + 1), s + D);
        if (s + 1 < OWN) xload((s + 1) & 1, s + 1);
        // keep the prefetches HERE: left alone, the machine scheduler sinks every weight load to just above its first use
        // (shorter live ranges) and the loop then pays one full L2 round trip per fragment (measured: 30 us instead of 8)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt) {
            const bf16x8 wh = wr[s % (D + 1)][mt][0], wl = wr[s % (D + 1)][mt][1];
#pragma unroll
            for (int pg = 0; pg < PGW; ++pg) acc[pg][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, xh[s & 1][pg], acc[pg][mt], 0, 0, 0);
#pragma unroll
            for (int pg = 0; pg < PGW; ++pg) acc[pg][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xl[s & 1][pg], acc[pg][mt], 0, 0, 0);
#pragma unroll
            for (int pg = 0; pg < PGW; ++pg) acc[pg][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xh[s & 1][pg], acc[pg][mt], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    if (KS > 1) {
        // both wave groups end with the full sum (group 0 + group 1, the same order in both), so the epilogue below -- including
        // its workgroup barriers -- runs unchanged on all waves; only group 0 stores
        __syncthreads();                                               // every wave is done reading the tile
        f32x4* xch = reinterpret_cast<f32x4*>(smem);
#pragma unroll
        for (int pg = 0; pg < PGW; ++pg)
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt) xch[((wave * PGW + pg) * MTW + mt) * 64 + lane] = acc[pg][mt];
        __syncthreads();
#pragma unroll
        for (int pg = 0; pg < PGW; ++pg)
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt) {
                const f32x4 a0 = xch[((w4 * PGW + pg) * MTW + mt) * 64 + lane], a1 = xch[(((4 + w4) * PGW + pg) * MTW + mt) * 64 + lane];
                acc[pg][mt] = a0 + a1;
            }
    }
    const bool storing = kh == 0;

    // ---- epilogue: lane holds channels (mt0 + mt)*16 + 4q + {0..3} of pixel (y0 + wp*PGW + pg, x0 + p) ----
    float4 bv[MTW];
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt) bv[mt] = a.bias ? ld4(a.bias + (mt0 + mt) * 16 + q * 4) : f4zero();
    float4 v[PGW][MTW];
#pragma unroll
    for (int pg = 0; pg < PGW; ++pg)
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt)
            v[pg][mt] = make_float4(acc[pg][mt][0] + bv[mt].x, acc[pg][mt][1] + bv[mt].y, acc[pg][mt][2] + bv[mt].z, acc[pg][mt][3] + bv[mt].w);
    // sum over ALL channels of a pixel of a per-lane partial: 4 q-lanes by shuffles, the WN waves through LDS (every thread calls it)
    auto pixel_sum = [&](float (&part)[PGW]) {
#pragma unroll
        for (int pg = 0; pg < PGW; ++pg) {
            part[pg] = sum_rows4(part[pg]);
        }
        if (WN > 1) {
            __syncthreads();           // ss_l may still be read from a previous call
            if (q == 0 && storing) {
#pragma unroll
                for (int pg = 0; pg < PGW; ++pg) ss_l[(wn * 4 + wp * PGW + pg) * 16 + p] = part[pg];
            }
            __syncthreads();
#pragma unroll
            for (int pg = 0; pg < PGW; ++pg) {
                float t = 0.f;
#pragma unroll
                for (int w2 = 0; w2 < WN; ++w2) t += ss_l[(w2 * 4 + wp * PGW + pg) * 16 + p];
                part[pg] = t;
            }
        }
    };
    const float inv_n = 1.0f / (float)a.N;
    if (EPI == EPI_LRELU_PN) {
        float ssum[PGW];
#pragma unroll
        for (int pg = 0; pg < PGW; ++pg) {
            float ss = 0.f;
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt) {
                float4 c = v[pg][mt];
                c.x = c.x > 0.f ? c.x : a.slope * c.x; c.y = c.y > 0.f ? c.y : a.slope * c.y;
                c.z = c.z > 0.f ? c.z : a.slope * c.z; c.w = c.w > 0.f ? c.w : a.slope * c.w;
                ss += f4dot(c, c);
                v[pg][mt] = c;
            }
            ssum[pg] = ss;
        }
        pixel_sum(ssum);
#pragma unroll
        for (int pg = 0; pg < PGW; ++pg) {
            const int gy = y0 + wp * PGW + pg, gx = x0 + p;
            const float r = sqrtf(ssum[pg] * inv_n + a.eps);
            const float inv = 1.0f / r;
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt) v[pg][mt] = f4scale(v[pg][mt], inv);
            if (storing && gy < a.H && gx < a.W && q == 0 && wn == 0) a.rn[((long)b * a.H + gy) * a.W + gx] = r;
        }
    }
    const int ch0 = mt0 * 16 + q * 4;
    if (OUTMODE == 0) {
        long pix[PGW];
        bool valid[PGW];
#pragma unroll
        for (int pg = 0; pg < PGW; ++pg) {
            const int gy = y0 + wp * PGW + pg, gx = x0 + p;
            valid[pg] = gy < a.H && gx < a.W;
            pix[pg] = ((long)b * a.H + (valid[pg] ? gy : 0)) * a.W + (valid[pg] ? gx : 0);
        }
        if (EPI == EPI_PN_BWD) {
            float4 yy[PGW][MTW];
            float s[PGW];
#pragma unroll
            for (int pg = 0; pg < PGW; ++pg) {
                s[pg] = 0.f;
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt) {
                    yy[pg][mt] = ld4(a.ay + pix[pg] * a.N + ch0 + mt * 16);
                    s[pg] += f4dot(v[pg][mt], yy[pg][mt]);
                }
            }
            pixel_sum(s);
#pragma unroll
            for (int pg = 0; pg < PGW; ++pg) {
                const float inv_r = 1.0f / a.arn[pix[pg]];
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt) v[pg][mt] = pn_bwd4(v[pg][mt], yy[pg][mt], s[pg] * inv_n, inv_r, a.slope);
            }
        }
#pragma unroll
        for (int pg = 0; pg < PGW; ++pg)
            if (storing && valid[pg]) {
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt) st4(a.y + pix[pg] * a.N + ch0 + mt * 16, v[pg][mt]);
            }
    } else {
        const long W2 = 2L * a.W;
#pragma unroll
        for (int sub = 0; sub < 4; ++sub) {
            long pix[PGW];
            bool valid[PGW];
            float4 o4[PGW][MTW];
#pragma unroll
            for (int pg = 0; pg < PGW; ++pg) {
                const int gy = y0 + wp * PGW + pg, gx = x0 + p;
                valid[pg] = gy < a.H && gx < a.W;
                pix[pg] = ((long)b * 2 * a.H + 2 * (valid[pg] ? gy : 0) + (sub >> 1)) * W2 + 2 * (valid[pg] ? gx : 0) + (sub & 1);
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt) o4[pg][mt] = f4scale(v[pg][mt], 0.25f);
            }
            if (EPI == EPI_PN_BWD) {
                float4 yy[PGW][MTW];
                float s[PGW];
#pragma unroll
                for (int pg = 0; pg < PGW; ++pg) {
                    s[pg] = 0.f;
#pragma unroll
                    for (int mt = 0; mt < MTW; ++mt) {
                        yy[pg][mt] = ld4(a.ay + pix[pg] * a.N + ch0 + mt * 16);
                        s[pg] += f4dot(o4[pg][mt], yy[pg][mt]);
                    }
                }
                pixel_sum(s);
#pragma unroll
                for (int pg = 0; pg < PGW; ++pg) {
                    const float inv_r = 1.0f / a.arn[pix[pg]];
#pragma unroll
                    for (int mt = 0; mt < MTW; ++mt) o4[pg][mt] = pn_bwd4(o4[pg][mt], yy[pg][mt], s[pg] * inv_n, inv_r, a.slope);
                }
            }
#pragma unroll
            for (int pg = 0; pg < PGW; ++pg)
                if (storing && valid[pg]) {
#pragma unroll
                    for (int mt = 0; mt < MTW; ++mt) st4(a.y + pix[pg] * a.N + ch0 + mt * 16, o4[pg][mt]);
                }
        }
    }
}

// output-channel slice per workgroup: the largest of {N, 64, 32} (a divisor of N) that still gives >= 256 workgroups
int mid_slice(int n_tiles, int N) {
    const int cand[3] = {N, 64, 32};
    for (int i = 0; i < 3; ++i) {
        const int ns = cand[i];
        if (ns > N || N % ns || !(ns == 32 || ns == 64 || ns == 128)) continue;
        if ((long)n_tiles * (N / ns) >= 256) return ns;
    }
    return 32;
}

template <int KG, int PGW, int MTW, int WN, int PREC>
int mid_launch_cfg(ConvArgs a, int n_tiles, int n_slices, int resample, int epi, int outmode, hipStream_t s) {
    constexpr int KS = 1;   // KS = 2 (instantiable where KG is even and PGW*MTW <= 4) measured equal within noise in graph replay and
                            // 0-40 % slower per launch under rocprofv3.  Inside a replayed graph a 128->128 launch at 16x16 takes 6.6 us, 6.3 us
                            // with the weight stream switched off (zero-record descriptor): neither the stream nor the wave chain is the limit;
                            // what is left is one memory round trip of staging, ~2 us of MFMAs and the launch itself (tools/micro_graph.py)
    const dim3 grid(n_tiles, n_slices), block(256 * KS);
    if (epi == EPI_PN_BWD && outmode) hipLaunchKernelGGL((conv3x3_mid_kernel<KG, PGW, MTW, WN, EPI_PN_BWD, 1, KS, PREC>), grid, block, 0, s, a, resample);
    else if (epi == EPI_PN_BWD) hipLaunchKernelGGL((conv3x3_mid_kernel<KG, PGW, MTW, WN, EPI_PN_BWD, 0, KS, PREC>), grid, block, 0, s, a, resample);
    else if (outmode) hipLaunchKernelGGL((conv3x3_mid_kernel<KG, PGW, MTW, WN, 0, 1, KS, PREC>), grid, block, 0, s, a, resample);
    else if (epi) hipLaunchKernelGGL((conv3x3_mid_kernel<KG, PGW, MTW, WN, 1, 0, KS, PREC>), grid, block, 0, s, a, resample);
    else hipLaunchKernelGGL((conv3x3_mid_kernel<KG, PGW, MTW, WN, 0, 0, KS, PREC>), grid, block, 0, s, a, resample);
    return ngan::launch_status("ngan_conv3x3_fwd(mid)");
}

template <int KG, int PREC>
int mid_launch_kg(const ConvArgs& a, int n_tiles, int ns, int resample, int epi, int outmode, hipStream_t s) {
    const int n_slices = a.N / ns;
    if (ns == 32) return mid_launch_cfg<KG, 2, 1, 2, PREC>(a, n_tiles, n_slices, resample, epi, outmode, s);
    if (ns == 64) return mid_launch_cfg<KG, 2, 2, 2, PREC>(a, n_tiles, n_slices, resample, epi, outmode, s);
    return mid_launch_cfg<KG, 4, 2, 4, PREC>(a, n_tiles, n_slices, resample, epi, outmode, s);
}

}  // namespace

namespace ngan {

bool conv3x3_mid_eligible(int B, int H, int W, int K, int N) {
    return B > 0 && H > 0 && W > 0 && (K == 32 || K == 64 || K == 128) && (N == 32 || N == 64 || N == 128);
}

bool conv3x3_mid_fuses_epilogue(int B, int H, int W, int K, int N) {
    return conv3x3_mid_eligible(B, H, W, K, N) && mid_slice(B * ceil_div(W, 16) * ceil_div(H, 4), N) == N;
}

int conv3x3_mid_launch(const float* x, const float* packed, const float* bias, float* y, float* rnorm, const float* aux_in,
                       const float* aux_rn, int B, int H, int W, int K, int N, int resample, int epilogue, int out_mode,
                       float slope, float eps, int precision, hipStream_t s) {
    ConvArgs a{x, packed, bias, y, rnorm, B, H, W, K, N, ceil_div(W, 16), ceil_div(H, 4), slope, eps, aux_in, aux_rn, nullptr};
    const int n_tiles = B * a.tiles_x * a.tiles_y;
    const int ns = mid_slice(n_tiles, N);
    const bool fused = epilogue == EPI_NONE || ns == N;      // the channel-reducing epilogues need all N channels in one workgroup
    const int epi = fused ? epilogue : EPI_NONE;
    int st;
    if (precision == 0) {
        if (K == 32) st = mid_launch_kg<1, 0>(a, n_tiles, ns, resample, epi, out_mode, s);
        else if (K == 64) st = mid_launch_kg<2, 0>(a, n_tiles, ns, resample, epi, out_mode, s);
        else st = mid_launch_kg<4, 0>(a, n_tiles, ns, resample, epi, out_mode, s);
    }
    else if (K == 32 || K == 16) st = mid_launch_kg<1, 1>(a, n_tiles, ns, resample, epi, out_mode, s);   // K = 16: padded weights, a.K = 16
    else if (K == 64) st = mid_launch_kg<2, 1>(a, n_tiles, ns, resample, epi, out_mode, s);
    else st = mid_launch_kg<4, 1>(a, n_tiles, ns, resample, epi, out_mode, s);
    if (st || fused) return st;
    // channels were split over workgroups: the epilogue runs as a second, tiny launch over all N channels of a pixel, in place
    const long npix = (long)B * H * W * (out_mode ? 4 : 1);
    if (epilogue == EPI_LRELU_PN) return ngan_lrelu_pixelnorm_fwd(y, nullptr, y, rnorm, npix, N, slope, eps, (void*)s);   // bias already added
    return ngan_lrelu_pixelnorm_bwd(y, nullptr, aux_in, aux_rn, y, npix, N, slope, (void*)s);
}

int conv3x3_mid_kernel_name(int B, int H, int W, int K, int N, int resample, int epilogue, int out_mode, int precision, char* buf, int len) {
    const int n_tiles = B * ceil_div(W, 16) * ceil_div(H, 4);
    const int ns = mid_slice(n_tiles, N);
    const int epi = ns == N ? epilogue : 0;
    const int pgw = ns == 128 ? 4 : 2, mtw = ns == 32 ? 1 : 2, wnn = ns == 128 ? 4 : 2;
    const int ks = 1;
    snprintf(buf, len, "conv3x3_mid_kernel<%d, %d, %d, %d, %d, %d, %d, %d>", K / 32, pgw, mtw, wnn, epi, out_mode ? 1 : 0, ks, precision ? 1 : 0);
    return NGAN_OK;
}

}  // namespace ngan
