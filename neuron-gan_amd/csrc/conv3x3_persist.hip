#include "conv3x3_internal.h"

namespace {


// ---------------------------------------------------------------------------------------------------------
// Persistent, software-pipelined variant for the layers that carry most of the work: K, N in {16, 32} on large
// images (8x32-pixel tiles).  A workgroup keeps the whole packed weight tensor in LDS, walks a band of tiles
// (bands are assigned per XCD so that neighbouring tiles' halos hit the same L2), and issues the global loads of
// tile t+1 before the MFMAs of tile t, so every CU always has a tile's worth of loads in flight.
// Bilinear x2 input: the low-resolution source patch (6x18 pixels) is staged once and expanded LDS -> LDS.
// ---------------------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------------------
// Tile schedule of the persistent kernels: as many workgroups as are resident, an equal share of tiles each, strided inside the
// band of the workgroup's XCD (blockIdx & 7), so that neighbouring tiles meet in one L2 at about the same time.
// Measured and rejected (round 2's in-kernel clock probe, fp32 16 -> 16 at 512x512, 177 us): the workgroups of a CU finish up to 75 us apart
// (the waves sharing a SIMD are served oldest first), but neither an atomic ticket queue (220 us: the compiler guards the tile
// loop's register hazards with s_waitcnt vmcnt(0..1), so every tile waited for the in-flight atomic) nor an over-decomposed grid
// that the dispatcher back-fills (2 / 4 / 8 tiles per workgroup: 184 / 181 / 180 us) is faster: the tail is not where the time goes.
// What the fp32 instances are short of is VALU issue: v_mfma_f32_16x16x4_f32 runs at the vector-FMA rate and does not overlap
// with other waves' VALU work -- with loads, stores, LDS staging and the epilogue compiled out one by one the kernel loses exactly
// the issue time of the instructions removed (161 / 143 / 132 us; the MFMAs alone need 123).
// ---------------------------------------------------------------------------------------------------------
template <int MTW, int KG, int RES, int EPI, int OUTMODE, int PREC>
__global__ __launch_bounds__(256, PREC == 2 ? NGAN_WINO16_WPE : (MTW * KG == 1) ? (PREC ? 3 : 4) : 2) void conv3x3_persist_kernel(ConvArgs a, int n_tiles) {
    // PREC: 0 exact fp32 (direct), 1 split bf16, 2 exact fp32 by Winograd F(2x2, 3x3) (16 -> 16; the MFMA section of conv3x3_tile_kernel)
    constexpr bool BF = PREC == 1, WINO = PREC == 2;
    static_assert(!WINO || (MTW == 1 && KG == 1), "the Winograd form is built for the 16 -> 16 layers");
    // tile: 8 x 32 pixels, 4 pixel groups of 16 per wave; the 32 -> 32 instances use 4 x 32 (2 groups per wave): their 8-row tile
    // needs 88 KB of LDS and ~260 registers, i.e. ONE workgroup per CU with nothing to overlap its load / barrier / MFMA phases
    constexpr int THc = persist_tile_h(MTW, KG, RES), PGW = THc / 2, RPW = THc / 4;
    constexpr int TWc = 32, HH_ = THc + 2, HW_ = TWc + 2, NPIX = HH_ * HW_, LP = 40;
    constexpr int PH = THc / 2 + 2, PW = TWc / 2 + 2, NPP = PH * PW;
    constexpr int NSTEP = KG == 1 ? 5 : 9;   // bf16x3: K = 32 contraction steps per tile
    constexpr int W_ELEMS = BF ? NSTEP * MTW * 2 * 256 : (WINO ? 16 * 256 : 9 * KG * MTW * 256), PLANE = HH_ * LP * 16, TILE_ELEMS = KG * PLANE;
    constexpr int PATCH_ELEMS = RES == NGAN_RESAMPLE_UP2 ? KG * NPP * 16 : 0;
    constexpr int N_SRC = RES == NGAN_RESAMPLE_UP2 ? KG * NPP * 4 : KG * NPIX * 4;   // float4 loads per tile
    constexpr int NST = (N_SRC + 255) / 256;
    constexpr int NEX = (KG * NPIX * 4 + 255) / 256;                                 // expansion items (bilinear)
    __shared__ __attribute__((aligned(16))) float smem[W_ELEMS + TILE_ELEMS + PATCH_ELEMS];
    float* wl = smem;
    float* tile = smem + W_ELEMS;
    float* patch = tile + TILE_ELEMS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int p = lane & 15, q = lane >> 4;
    constexpr int K = KG * 16, N = MTW * 16;


    const TileRun run = tile_run(n_tiles);
    int t = run.t;
    const int t_end = run.t_end;
    const int h = a.H >> 1, w = a.W >> 1;

    // ---- tile-invariant per-thread staging descriptors (all index arithmetic happens once, here) ----
    // source pixel offset from the tile origin (dy in the high half, dx in the low half of one register: the kernel runs at its
    // register cap, and the ticket of the dynamic tile schedule must stay in a register for a whole tile), channel, LDS float index
    auto f32_or_bf16_slot = [&](int g, int c4, int ty, int tx) {
        if (WINO) {      // even / odd columns in separate halves of a row: conv3x3_tile_kernel
            const int pos = (tx >> 1) + (tx & 1) * (LP / 2);
            return (ty * LP + pos) * 16 + ((c4 ^ (((pos >> 2) & 1) << 1)) << 2);
        }
        return BF ? bf16_slot<KG, PLANE, LP>(g, c4, ty, tx) : g * PLANE + (ty * LP + tx) * 16 + ((c4 ^ (((tx >> 2) & 1) << 1)) << 2);
    };
    // pixel group pg of a wave: 16 consecutive pixels of a row (direct forms) / pixel (pg >> 1, pg & 1) of this lane's 2x2 tile (Winograd)
    auto pg_row = [&](int pg) { return WINO ? 2 * wave + (pg >> 1) : wave * RPW + (pg >> 1); };
    auto pg_col = [&](int pg) { return WINO ? 2 * p + (pg & 1) : (pg & 1) * 16 + p; };
    int wrd[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const int pos = p + (b >> 1) + (b & 1) * (LP / 2);
        wrd[b] = pos * 16 + ((q ^ (((pos >> 2) & 1) << 1)) << 2);
    }
    int s_dyx[NST], s_ch[NST], s_lds[NST];
    auto dy_of = [](int v) { return v >> 16; };
    auto dx_of = [](int v) { return (int)(short)(v & 0xffff); };
#pragma unroll
    for (int i = 0; i < NST; ++i) {
        const int e = tid + i * 256;
        const int c4 = e & 3;
        if (RES == NGAN_RESAMPLE_UP2) {
            const int pp = (e >> 2) % NPP, g = (e >> 2) / NPP;
            s_dyx[i] = ((pp / PW - 1) << 16) | ((pp % PW - 1) & 0xffff); s_ch[i] = g * 16 + c4 * 4;
            s_lds[i] = e * 4;                                   // patch is plain [g][py][px][16]
        } else {
            const int pix = (e >> 2) % NPIX, g = (e >> 2) / NPIX;
            const int ty = pix / HW_, tx = pix % HW_;
            s_dyx[i] = ((ty - 1) << 16) | ((tx - 1) & 0xffff); s_ch[i] = g * 16 + c4 * 4;
            s_lds[i] = f32_or_bf16_slot(g, c4, ty, tx);
        }
    }
    // expansion descriptors (bilinear): destination LDS index, the 4 patch taps and whether the item exists
    int x_dst[RES == NGAN_RESAMPLE_UP2 ? NEX : 1], x_src[RES == NGAN_RESAMPLE_UP2 ? NEX : 1];
    int x_ty[RES == NGAN_RESAMPLE_UP2 ? NEX : 1], x_tx[RES == NGAN_RESAMPLE_UP2 ? NEX : 1];
    if (RES == NGAN_RESAMPLE_UP2) {
#pragma unroll
        for (int i = 0; i < NEX; ++i) {
            const int e = tid + i * 256;
            const int c4 = e & 3, pix = (e >> 2) % NPIX, g = (e >> 2) / NPIX;
            const int ty = pix / HW_, tx = pix % HW_;
            x_ty[i] = e < KG * NPIX * 4 ? ty : -100; x_tx[i] = tx;
            x_dst[i] = f32_or_bf16_slot(g, c4, ty, tx);
            // high-res (ty-1, tx-1) relative to an even tile origin: odd offsets are "even" output rows (2i): taps (i-1, i)
            // patch row index = low-res row - (y0/2 - 1); for offset d = ty-1: even d -> rows d/2, d/2+1 ; odd d -> (d+1)/2, (d+1)/2+1 ... see below
            const int dy = ty - 1, dx = tx - 1;   // in [-1, 8] / [-1, 32]
            const int ry = (dy + 1) >> 1, rx = (dx + 1) >> 1;   // first tap's patch row / col (second tap is +1)
            x_src[i] = ((g * PH + ry) * PW + rx) * 16 + c4 * 4;
        }
    }
    // MFMA B-operand read addresses: one per dx (rotation depends on the pixel column only)
    int rd[3];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) rd[dx] = (p + dx) * 16 + ((q ^ ((((p + dx) >> 2) & 1) << 1)) << 2);
    // bf16x3: per contraction step, this lane's offset of the hi fragment (lo = same ^ 8 floats for K = 16, + PLANE for K = 32)
    int rs[BF ? NSTEP : 1];
    if (BF) {
#pragma unroll
        for (int st = 0; st < NSTEP; ++st) {
            int tap = KG == 1 ? 2 * st + (q >> 1) : st;
            if (tap > 8) tap = 8;                                  // zero-weight padding tap: any valid address
            const int dy = tap / 3, dx = tap % 3;
            const int slot = (KG == 1 ? (q & 1) : q) ^ ((((p + dx) >> 2) & 1) << 1);
            rs[st] = (dy * LP + p + dx) * 16 + slot * 4;
        }
    }

    auto decode = [&](int tt, int& b, int& y0, int& x0) {
        const int txi = tt % a.tiles_x; tt /= a.tiles_x;
        const int tyi = tt % a.tiles_y;
        b = tt / a.tiles_y;
        y0 = tyi * THc; x0 = txi * TWc;
    };
    float4 stg[NST];
    // Tile loads go through a buffer descriptor of the tile's image: a 32-bit byte offset per lane (no 64-bit address arithmetic)
    // and the hardware range check turns an out-of-range offset into zeros -- the conv padding and the unused staging slots cost
    // one select on the offset instead of a branch around the load plus four zeroed registers.
    const unsigned src_img_bytes = (unsigned)((RES == NGAN_RESAMPLE_UP2 ? h * w : a.H * a.W) * K) * 4u;   // host: < 2^32
    constexpr unsigned OOB = 0xFFFFFFF0u;
    int s_off[NST];      // tile-invariant part of the byte offset (plain input); bilinear input: channel byte offset
#pragma unroll
    for (int i = 0; i < NST; ++i)
        s_off[i] = RES == NGAN_RESAMPLE_UP2 ? s_ch[i] * 4 : ((dy_of(s_dyx[i]) * a.W + dx_of(s_dyx[i])) * K + s_ch[i]) * 4;
    auto issue = [&](int tt) {
        int b, y0, x0;
        decode(tt, b, y0, x0);
        const float* base = a.x + (long)b * (RES == NGAN_RESAMPLE_UP2 ? h * w : a.H * a.W) * K;
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, src_img_bytes, 0x00020000);
        if (RES == NGAN_RESAMPLE_UP2) {
            const int ly0 = y0 >> 1, lx0 = x0 >> 1;
#pragma unroll
            for (int i = 0; i < NST; ++i) {
                const int ly = min(max(ly0 + dy_of(s_dyx[i]), 0), h - 1), lx = min(max(lx0 + dx_of(s_dyx[i]), 0), w - 1);
                const unsigned off = (tid + i * 256 < N_SRC) ? (unsigned)((ly * w + lx) * K * 4 + s_off[i]) : OOB;
                stg[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0));
            }
        } else {
            const int tile_off = (y0 * a.W + x0) * K * 4;
#pragma unroll
            for (int i = 0; i < NST; ++i) {
                const bool ok = (tid + i * 256 < N_SRC) && (unsigned)(y0 + dy_of(s_dyx[i])) < (unsigned)a.H && (unsigned)(x0 + dx_of(s_dyx[i])) < (unsigned)a.W;
                const unsigned off = ok ? (unsigned)(tile_off + s_off[i]) : OOB;
                stg[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0));
            }
        }
    };
    float4 bv[MTW], wimg[MTW];
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt) {
        bv[mt] = a.bias ? ld4(a.bias + mt * 16 + q * 4) : f4zero();
        wimg[mt] = EPI == EPI_TO_IMAGE ? ld4(a.ay + mt * 16 + q * 4) : f4zero();
    }
    if (t < t_end) issue(t);
    {   // the packed weights -> LDS, requested behind the first tile's loads and all at once (conv3x3_tile_kernel)
        constexpr int NWL = (W_ELEMS / 4 + 255) / 256;
        float4 wtmp[NWL];
#pragma unroll
        for (int i = 0; i < NWL; ++i) wtmp[i] = (tid + i * 256 < W_ELEMS / 4) ? ld4(a.wp + (long)(tid + i * 256) * 4) : f4zero();
#pragma unroll
        for (int i = 0; i < NWL; ++i)
            if (tid + i * 256 < W_ELEMS / 4) st4(wl + (tid + i * 256) * 4, wtmp[i]);
    }
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt) { pin_registers(bv[mt]); pin_registers(wimg[mt]); }     // (awaited once, here: conv3x3_internal.h)
    const float inv_n = 1.0f / (float)N;

    while (t < t_end) {
        int b, y0, x0;
        decode(t, b, y0, x0);
        __syncthreads();   // previous tile's MFMAs have finished reading `tile`
        if (RES == NGAN_RESAMPLE_UP2) {
#pragma unroll
            for (int i = 0; i < NST; ++i)
                if (tid + i * 256 < N_SRC) st4(&patch[s_lds[i]], stg[i]);
            __syncthreads();
#pragma unroll
            for (int i = 0; i < NEX; ++i) {
                if (x_ty[i] < 0) continue;
                const int Y = y0 + x_ty[i] - 1, X = x0 + x_tx[i] - 1;
                float4 v = f4zero();
                if (Y >= 0 && Y < a.H && X >= 0 && X < a.W) {
                    // Y odd -> taps (i, i+1) weights (.75, .25); Y even -> taps (i-1, i) weights (.25, .75); the patch was
                    // loaded with clamped coordinates, so border clamping needs no special case here
                    const float wy0 = (Y & 1) ? 0.75f : 0.25f, wx0 = (X & 1) ? 0.75f : 0.25f;
                    const float* r0 = patch + x_src[i];
                    float4 top = f4fma(ld4(r0 + 16), 1.0f - wx0, f4scale(ld4(r0), wx0));
                    float4 bot = f4fma(ld4(r0 + PW * 16 + 16), 1.0f - wx0, f4scale(ld4(r0 + PW * 16), wx0));
                    v = f4fma(bot, 1.0f - wy0, f4scale(top, wy0));
                }
                if (BF) st_split<KG, PLANE>(tile, x_dst[i], v);
                else st4(&tile[x_dst[i]], v);
            }
        } else {
#pragma unroll
            for (int i = 0; i < NST; ++i)
                if (tid + i * 256 < N_SRC) {
                    if (BF) st_split<KG, PLANE>(tile, s_lds[i], stg[i]);
                    else st4(&tile[s_lds[i]], stg[i]);
                }
        }
        __syncthreads();
        const int tn = t + run.step;
        if (tn < t_end) issue(tn);   // in flight while this tile is computed
        // PixelNorm-backward epilogue: its operands (this tile's pixels of the producer's output and norm) are requested now,
        // so that they arrive during the MFMAs instead of stalling the epilogue
        constexpr bool PRE = EPI == EPI_PN_BWD && OUTMODE == 0 && MTW * KG > 1;   // (the 16 -> 16 instance has no registers to spare: 1.7x slower with it)
        float4 yy_pre[PRE ? PGW : 1][MTW];
        float rn_pre[PRE ? PGW : 1];
        if (PRE) {
            const long img0 = (long)b * a.H * a.W;
#pragma unroll
            for (int pg = 0; pg < PGW; ++pg) {
                const int gy = y0 + pg_row(pg), gx = x0 + pg_col(pg);
                const bool valid = gy < a.H && gx < a.W;
                const long pix = img0 + (long)(valid ? gy : 0) * a.W + (valid ? gx : 0);
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt) yy_pre[pg][mt] = ld4(a.ay + pix * N + mt * 16 + q * 4);
                rn_pre[pg] = a.arn[pix];
            }
        }

        f32x4 acc[PGW][MTW];
#pragma unroll
        for (int pg = 0; pg < PGW; ++pg)
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt) acc[pg][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (WINO) {
            // Winograd F(2x2, 3x3): the section of conv3x3_tile_kernel (documented there); the bias is added by this kernel's epilogue
            const f32x2 m1 = opaque_minus_one();
            f32p bd[4][4];
            {
                const float* trow = tile + (2 * wave) * (LP * 16);
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const f32p d0 = pk2(*reinterpret_cast<const f32x4*>(trow + 0 * LP * 16 + wrd[b])), d1 = pk2(*reinterpret_cast<const f32x4*>(trow + 1 * LP * 16 + wrd[b]));
                    const f32p d2 = pk2(*reinterpret_cast<const f32x4*>(trow + 2 * LP * 16 + wrd[b])), d3 = pk2(*reinterpret_cast<const f32x4*>(trow + 3 * LP * 16 + wrd[b]));
                    bd[0][b] = psub(d0, d2, m1); bd[1][b] = d1 + d2; bd[2][b] = psub(d2, d1, m1); bd[3][b] = psub(d1, d3, m1);
                }
            }
            f32p ta[2][4];
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                f32x4 m[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) m[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
                f32p vv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    vv[u] = v == 0 ? psub(bd[u][0], bd[u][2], m1) : v == 1 ? bd[u][1] + bd[u][2] : v == 2 ? psub(bd[u][2], bd[u][1], m1) : psub(bd[u][1], bd[u][3], m1);
                f32x4 uu[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) uu[u] = *reinterpret_cast<const f32x4*>(&wl[(u * 4 + v) * 256 + lane * 4]);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        m[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(uu[u][i], i < 2 ? vv[u].l[i & 1] : vv[u].h[i & 1], m[u], 0, 0, 0);
                ta[0][v] = pk2(m[0]) + pk2(m[1]) + pk2(m[2]);
                ta[1][v] = psub(psub(pk2(m[1]), pk2(m[2]), m1), pk2(m[3]), m1);
            }
#pragma unroll
            for (int ar = 0; ar < 2; ++ar) {
                acc[ar * 2 + 0][0] = unpk2(ta[ar][0] + ta[ar][1] + ta[ar][2]);
                acc[ar * 2 + 1][0] = unpk2(psub(psub(ta[ar][1], ta[ar][2], m1), ta[ar][3], m1));
            }
        } else if (BF) {
#pragma unroll
            for (int st = 0; st < NSTEP; ++st) {
                bf16x8 xh[PGW], xl[PGW];
#pragma unroll
                for (int pg = 0; pg < PGW; ++pg) {
                    const int row = wave * RPW + (pg >> 1);
                    const int base = (row * LP + (pg & 1) * 16) * 16 + rs[st];
                    xh[pg] = *reinterpret_cast<const bf16x8*>(&tile[base]);
                    xl[pg] = *reinterpret_cast<const bf16x8*>(&tile[KG == 1 ? (base ^ 8) : (base + PLANE)]);
                }
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt) {
                    const bf16x8 wh = *reinterpret_cast<const bf16x8*>(&wl[((st * MTW + mt) * 2 + 0) * 256 + lane * 4]);
                    const bf16x8 wlo = *reinterpret_cast<const bf16x8*>(&wl[((st * MTW + mt) * 2 + 1) * 256 + lane * 4]);
#pragma unroll
                    for (int pg = 0; pg < PGW; ++pg) acc[pg][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wlo, xh[pg], acc[pg][mt], 0, 0, 0);
#pragma unroll
                    for (int pg = 0; pg < PGW; ++pg) acc[pg][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xl[pg], acc[pg][mt], 0, 0, 0);
#pragma unroll
                    for (int pg = 0; pg < PGW; ++pg) acc[pg][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xh[pg], acc[pg][mt], 0, 0, 0);
                }
            }
        } else {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3, dx = tap % 3;
#pragma unroll
            for (int g = 0; g < KG; ++g) {
                float xv[PGW][4];
#pragma unroll
                for (int pg = 0; pg < PGW; ++pg) {
                    const int row = wave * RPW + (pg >> 1);
                    float4 v = ld4(&tile[g * PLANE + ((row + dy) * LP + (pg & 1) * 16) * 16 + rd[dx]]);
                    xv[pg][0] = v.x; xv[pg][1] = v.y; xv[pg][2] = v.z; xv[pg][3] = v.w;
                }
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt) {
                    const float4 wv4 = ld4(&wl[((tap * KG + g) * MTW + mt) * 256 + lane * 4]);
                    const float wv[4] = {wv4.x, wv4.y, wv4.z, wv4.w};
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int pg = 0; pg < PGW; ++pg)
                            acc[pg][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[i], xv[pg][i], acc[pg][mt], 0, 0, 0);
                }
            }
        }
        }
        // ---- epilogue (same math as conv3x3_kernel with WN = 1; reciprocal square root instead of sqrt + divide) ----
        const long img = (long)b * a.H * a.W;
        float timg = 0.f;
        __amdgpu_buffer_rsrc_t y_rsrc, rn_rsrc;
        if (OUTMODE == 0 && (EPI != EPI_TO_IMAGE || a.y)) {
            y_rsrc = __builtin_amdgcn_make_buffer_rsrc(a.y + img * N, 0, (unsigned)(a.H * a.W * N) * 4u, 0x00020000);
            if (EPI == EPI_LRELU_PN || EPI == EPI_TO_IMAGE)
                rn_rsrc = __builtin_amdgcn_make_buffer_rsrc(a.rn + img, 0, (unsigned)(a.H * a.W) * 4u, 0x00020000);
        }
        // PixelNorm-backward operands that were not prefetched: all of the tile's loads before its first store (a load issued behind a
        // store can only be awaited by draining that store, see pin_registers)
        // (operands of the PixelNorm-backward epilogue must have landed before the first store is issued: conv3x3_tile_kernel)
        if (EPI == EPI_PN_BWD && OUTMODE == 0 && PRE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        constexpr bool LATE = EPI == EPI_PN_BWD && OUTMODE == 0 && !PRE;
        float4 yy_epi[LATE ? PGW : 1][MTW];
        float rn_epi[LATE ? PGW : 1];
        if (LATE) {
#pragma unroll
            for (int pg = 0; pg < PGW; ++pg) {
                const int gy = y0 + pg_row(pg), gx = x0 + pg_col(pg);
                const bool valid = gy < a.H && gx < a.W;
                const long pix = img + (long)(valid ? gy : 0) * a.W + (valid ? gx : 0);
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt) yy_epi[pg][mt] = ld4(a.ay + pix * N + mt * 16 + q * 4);
                rn_epi[pg] = a.arn[pix];
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
#pragma unroll
        for (int pg = 0; pg < PGW; ++pg) {
            const int row = pg_row(pg), col = pg_col(pg);
            const int gy = y0 + row, gx = x0 + col;
            const bool valid = gy < a.H && gx < a.W;
            float4 v[MTW];
            float ss = 0.f;
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt) {
                float4 c = make_float4(acc[pg][mt][0] + bv[mt].x, acc[pg][mt][1] + bv[mt].y,
                                       acc[pg][mt][2] + bv[mt].z, acc[pg][mt][3] + bv[mt].w);
                if (EPI == EPI_LRELU_PN || EPI == EPI_TO_IMAGE) {
                    c.x = vmax1(c.x, a.slope * c.x); c.y = vmax1(c.y, a.slope * c.y);   // LeakyReLU, 0 <= slope <= 1
                    c.z = vmax1(c.z, a.slope * c.z); c.w = vmax1(c.w, a.slope * c.w);
                    ss += f4dot(c, c);
                }
                v[mt] = c;
            }
            if (EPI == EPI_LRELU_PN || EPI == EPI_TO_IMAGE) {
                ss = sum_rows4(ss);
                const float m = ss * inv_n + a.eps;
                const float inv = __builtin_amdgcn_rsqf(m);
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt) v[mt] = f4scale(v[mt], inv);
                if (EPI == EPI_LRELU_PN || a.y)
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, m * inv), rn_rsrc, (valid && q == 0) ? (unsigned)((gy * a.W + gx) * 4) : OOB, 0, 0);
            }
            if (EPI == EPI_TO_IMAGE) {
                float d = 0.f;
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt) d += f4dot(v[mt], wimg[mt]);
                d = sum_rows4(d);
                if (q == pg) timg = d;          // all four q-lanes hold pixel group pg's sum; lane group q keeps the one it will finish
            }
            if (EPI == EPI_PN_BWD && OUTMODE == 0) {
                // backward of the LeakyReLU -> PixelNorm that produced this layer's input, applied to the gradient just computed
                float4 yy[MTW];
                float rr;
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt) yy[mt] = PRE ? yy_pre[PRE ? pg : 0][mt] : yy_epi[PRE ? 0 : pg][mt];
                rr = PRE ? rn_pre[PRE ? pg : 0] : rn_epi[PRE ? 0 : pg];
                float s = 0.f;
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt) s += f4dot(v[mt], yy[mt]);
                s = sum_rows4(s);
                s *= inv_n;
                const float inv_r = 1.0f / rr;
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt) v[mt] = pn_bwd4(v[mt], yy[mt], s, inv_r, a.slope);
            }
            if (OUTMODE == 0) {
                if (EPI != EPI_TO_IMAGE || a.y) {
                    // stores through the output image's descriptor: an invalid (off-image) pixel gets an out-of-range offset
                    const unsigned off = valid ? (unsigned)(((gy * a.W + gx) * N + q * 4) * 4) : OOB;
#pragma unroll
                    for (int mt = 0; mt < MTW; ++mt)
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v[mt]), y_rsrc, off + (valid ? mt * 64 : 0), 0, 0);
                }
            } else {
                const long W2 = 2L * a.W;
                const long o00 = 4 * img + (long)(2 * (valid ? gy : 0)) * W2 + 2 * (valid ? gx : 0);
                // the four pooled-over pixels' PixelNorm-backward operands: all loads before the first of the four stores (a load
                // issued behind a store is awaited by draining that store, see pin_registers)
                float4 yy4[EPI == EPI_PN_BWD ? 4 : 1][MTW];
                float rr4[EPI == EPI_PN_BWD ? 4 : 1];
                if (EPI == EPI_PN_BWD) {
#pragma unroll
                    for (int sub = 0; sub < 4; ++sub) {
                        const long pix = o00 + (sub >> 1) * W2 + (sub & 1);
#pragma unroll
                        for (int mt = 0; mt < MTW; ++mt) yy4[sub][mt] = ld4(a.ay + pix * N + mt * 16 + q * 4);
                        rr4[sub] = a.arn[pix];
                    }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
#pragma unroll
                for (int sub = 0; sub < 4; ++sub) {
                    const long pix = o00 + (sub >> 1) * W2 + (sub & 1);
                    float4 o4[MTW];
#pragma unroll
                    for (int mt = 0; mt < MTW; ++mt) o4[mt] = f4scale(v[mt], 0.25f);
                    if (EPI == EPI_PN_BWD) {
                        float4 yy[MTW];
                        float s = 0.f;
#pragma unroll
                        for (int mt = 0; mt < MTW; ++mt) {
                            yy[mt] = yy4[EPI == EPI_PN_BWD ? sub : 0][mt];
                            s += f4dot(o4[mt], yy[mt]);
                        }
                        s = sum_rows4(s);
                        s *= inv_n;
                        const float inv_r = 1.0f / rr4[EPI == EPI_PN_BWD ? sub : 0];
#pragma unroll
                        for (int mt = 0; mt < MTW; ++mt) o4[mt] = pn_bwd4(o4[mt], yy[mt], s, inv_r, a.slope);
                    }
                    if (valid) {
#pragma unroll
                        for (int mt = 0; mt < MTW; ++mt) st4(a.y + pix * N + mt * 16 + q * 4, o4[mt]);
                    }
                }
            }
        }
        if (EPI == EPI_TO_IMAGE) {
            // one tanh per lane instead of four: lane group q finishes pixel group q (same tanhf as the standalone ToImage kernel)
            const int row = pg_row(q), col = pg_col(q);
            const int gy = y0 + row, gx = x0 + col;
            const float tv = tanhf(timg);
            if (q < PGW && gy < a.H && gx < a.W) a.aout[img + (long)gy * a.W + gx] = tv;
        }
        t = tn;
    }
}

template <int MTW, int KG, int RES, int EPI, int OUTMODE, int PREC>
int launch_persist(ConvArgs a, hipStream_t s) {
    a.tiles_x = ngan::ceil_div(a.W, 32);
    a.tiles_y = ngan::ceil_div(a.H, persist_tile_h(MTW, KG, RES));
    const int n_tiles = a.B * a.tiles_x * a.tiles_y;
    // persistent grid = what is actually resident (registers and LDS both limit it): an over-subscribed static
    // tile partition would serialise whole workgroups behind each other
    static int per_cu = 0;
    if (per_cu == 0) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, conv3x3_persist_kernel<MTW, KG, RES, EPI, OUTMODE, PREC>, 256, 0) != hipSuccess || n < 1) n = 1;
        per_cu = n > 4 ? 4 : n;
    }
    const int grid = persistent_grid(n_tiles, 256 * per_cu);
    hipLaunchKernelGGL((conv3x3_persist_kernel<MTW, KG, RES, EPI, OUTMODE, PREC>), dim3(grid), dim3(256), 0, s, a, n_tiles);
    return ngan::launch_status("ngan_conv3x3_fwd(persistent)");
}

template <int MTW, int KG, int PREC>
int dispatch_persist2(const ConvArgs& a, int res, int epi, int outmode, hipStream_t s) {
    if (epi == EPI_PN_BWD) return outmode == 1 ? launch_persist<MTW, KG, 0, EPI_PN_BWD, 1, PREC>(a, s) : launch_persist<MTW, KG, 0, EPI_PN_BWD, 0, PREC>(a, s);
    if (epi == EPI_TO_IMAGE) return launch_persist<MTW, KG, 0, EPI_TO_IMAGE, 0, PREC>(a, s);
    if (outmode == 1) return launch_persist<MTW, KG, 0, 0, 1, PREC>(a, s);
    if (res == 0) return epi ? launch_persist<MTW, KG, 0, 1, 0, PREC>(a, s) : launch_persist<MTW, KG, 0, 0, 0, PREC>(a, s);
    return epi ? launch_persist<MTW, KG, 2, 1, 0, PREC>(a, s) : launch_persist<MTW, KG, 2, 0, 0, PREC>(a, s);
}


template <int MTW, int KG>
int dispatch_persist_prec(const ConvArgs& a, int res, int epi, int outmode, int tprec, hipStream_t s) {
    if (tprec == 2) {
        if constexpr (MTW == 1 && KG == 1) return dispatch_persist2<1, 1, 2>(a, res, epi, outmode, s);
        else return NGAN_ERR_ARG;
    }
    return tprec ? dispatch_persist2<MTW, KG, 1>(a, res, epi, outmode, s) : dispatch_persist2<MTW, KG, 0>(a, res, epi, outmode, s);
}

}  // namespace

// tprec: the kernel's PREC template parameter (0 direct fp32, 1 split bf16, 2 Winograd fp32)
int ngan::conv3x3_persist_launch(const ConvArgs& a, int mtw, int kg, int resample, int epilogue, int out_mode, int tprec, hipStream_t s) {
    if (mtw == 1) return kg == 1 ? dispatch_persist_prec<1, 1>(a, resample, epilogue, out_mode, tprec, s)
                                 : dispatch_persist_prec<1, 2>(a, resample, epilogue, out_mode, tprec, s);
    return kg == 1 ? dispatch_persist_prec<2, 1>(a, resample, epilogue, out_mode, tprec, s)
                   : dispatch_persist_prec<2, 2>(a, resample, epilogue, out_mode, tprec, s);
}
