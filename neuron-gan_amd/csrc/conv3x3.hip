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
#include <cstdlib>
#include "conv3x3_shared.h"

namespace {

// ---------------------------------------------------------------------------------------------------------
// weight packing: OIHW -> [tap][k-group g][n-tile mt][lane][4], value * scale.
// lane l of (tap, g, mt) holds n = 16*mt + (l & 15) and k = 16*g + 4*(l >> 4) + i, i = 0..3.
// ---------------------------------------------------------------------------------------------------------
__global__ void pack_weights_kernel(const float* __restrict__ w, float* __restrict__ packed, int Cout, int Cin,
                                    int mode, float scale) {
    const int K = mode == 0 ? Cin : Cout, N = mode == 0 ? Cout : Cin;
    const int G = K / 16, MT = N / 16;
    const long total = 9L * K * N;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int i = idx & 3, lane = (idx >> 2) & 63;
    long r = idx >> 8;
    const int mt = r % MT; r /= MT;
    const int g = r % G;
    const int tap = r / G;
    const int n = mt * 16 + (lane & 15), k = g * 16 + 4 * (lane >> 4) + i;
    float v;
    if (mode == 0) v = w[((long)n * Cin + k) * 9 + tap];          // co = n, ci = k
    else           v = w[((long)k * Cin + n) * 9 + (8 - tap)];    // co = k, ci = n, taps flipped
    packed[idx] = v * scale;
}

// Winograd F(2x2, 3x3) packing ("precision code 4", fp32): U = G g G^T per (cout, cin) pair, G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1],
// times scale.  Layout [position u*4 + v][n-tile mt][k-group g][lane][4] with the lane convention of pack_weights_kernel.
__device__ __forceinline__ float wino_weight(const float* __restrict__ w, int Cout, int Cin, int mode, float scale, long idx) {
    const int K = mode == 0 ? Cin : Cout, N = mode == 0 ? Cout : Cin;
    const int G = K / 16, MT = N / 16;
    const int i = idx & 3, lane = (idx >> 2) & 63;
    long r = idx >> 8;
    const int g = r % G; r /= G;
    const int mt = r % MT;
    const int pos = r / MT, u = pos >> 2, v = pos & 3;
    const int n = mt * 16 + (lane & 15), k = g * 16 + 4 * (lane >> 4) + i;
    const float* src = mode == 0 ? w + ((long)n * Cin + k) * 9 : w + ((long)k * Cin + n) * 9;      // mode 1: taps flipped below
    float gm[3][3];
#pragma unroll
    for (int t = 0; t < 9; ++t) gm[t / 3][t % 3] = src[mode == 0 ? t : 8 - t];
    // row u of G applied to the rows of g, then row v of G to the columns
    float row[3];
#pragma unroll
    for (int c = 0; c < 3; ++c)
        row[c] = u == 0 ? gm[0][c] : u == 3 ? gm[2][c] : 0.5f * (gm[0][c] + (u == 1 ? gm[1][c] : -gm[1][c]) + gm[2][c]);
    const float val = v == 0 ? row[0] : v == 3 ? row[2] : 0.5f * (row[0] + (v == 1 ? row[1] : -row[1]) + row[2]);
    return val * scale;
}

__global__ void pack_weights_wino_kernel(const float* __restrict__ w, float* __restrict__ packed, int Cout, int Cin, int mode, float scale) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < 16L * Cin * Cout) packed[idx] = wino_weight(w, Cout, Cin, mode, scale, idx);
}

// Split-bf16 ("bf16x3") packing for v_mfma_f32_16x16x32_bf16: every weight w*scale is written as hi = bf16(w) and
// lo = bf16(w - hi).  Layout [step][n-tile mt][part hi/lo][lane][8]; lane l holds n = 16*mt + (l & 15) and
// k = 8*(l >> 4) + j.  K = 16: a step is a PAIR of taps (k < 16 -> tap 2*step, k >= 16 -> tap 2*step + 1; the 10th
// tap is zero padding), 5 steps.  K = 32*KG: step = kg*9 + tap covers input channels 32*kg .. 32*kg + 31 of one tap,
// 9*KG steps (K = 32: a step is one tap; K = 64, 128: conv3x3_mid.hip).
__global__ void pack_weights_bf16x3_kernel(const float* __restrict__ w, __bf16* __restrict__ packed, int Cout, int Cin,
                                           int mode, float scale, int pad32) {
    // pad32 (precision 2): a K = 16 contraction laid out as K = 32 with zero weights for channels 16..31 (conv3x3_mid.hip)
    const int Kreal = mode == 0 ? Cin : Cout, N = mode == 0 ? Cout : Cin;
    const int K = pad32 ? 32 : Kreal;
    const int MT = N / 16, nstep = K == 16 ? 5 : 9 * (K / 32);
    const long total = (long)nstep * MT * 2 * 64 * 8;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int j = idx & 7, lane = (idx >> 3) & 63, part = (idx >> 9) & 1;
    long r = idx >> 10;
    const int mt = r % MT;
    const int step = r / MT;
    const int n = mt * 16 + (lane & 15), kk = 8 * (lane >> 4) + j;
    const int tap = K == 16 ? 2 * step + (kk >> 4) : step % 9;
    const int k = K == 16 ? (kk & 15) : (step / 9) * 32 + kk;
    float v = 0.f;
    if (tap < 9 && k < Kreal) {
        if (mode == 0) v = w[((long)n * Cin + k) * 9 + tap];
        else           v = w[((long)k * Cin + n) * 9 + (8 - tap)];
    }
    v *= scale;
    const __bf16 hi = (__bf16)v;
    packed[idx] = part == 0 ? hi : (__bf16)(v - (float)hi);
}

// Bilinear x2 folded into the weights ("precision 3", conv3x3_up2f_kernel below).  An output pixel (2i + py, 2j + px) of
// conv3x3(up2(x)) only sees the 3x3 low-resolution neighbourhood of (i, j): hi-res row 2i + py + ky - 1 is a fixed blend of low-res
// rows i-1, i, i+1, so  W_eff[py][px][dr][dc] = sum_{ky,kx} W[ky][kx] * E[py][ky][dr] * E[px][kx][dc]  with the blend table E
// (align_corners = False taps .25/.75, ATen upsample_bilinear2d).  Four weight sets, one per output parity.
__device__ __forceinline__ float up2_blend(int parity, int k, int d) {       // weight of low-res offset d-1 in hi-res offset k-1
    // parity 0: rows 2i-1, 2i, 2i+1 -> (.75,.25,0) (.25,.75,0) (0,.75,.25);  parity 1: rows 2i, 2i+1, 2i+2 -> (.25,.75,0) (0,.75,.25) (0,.25,.75)
    const int r = parity + k;                                               // 0..3: hi-res offset from row 2i-1
    const float tab[4][3] = {{.75f, .25f, 0.f}, {.25f, .75f, 0.f}, {0.f, .75f, .25f}, {0.f, .25f, .75f}};
    return tab[r][d];
}

__device__ __forceinline__ float up2_folded_weight(const float* __restrict__ w, int Cin, int n, int k, int set, int tap) {
    const int py = set >> 1, px = set & 1, dr = tap / 3, dc = tap % 3;
    const float* wk = w + ((long)n * Cin + k) * 9;
    float v = 0.f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const float ey = up2_blend(py, ky, dr);
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) v = fmaf(wk[ky * 3 + kx], ey * up2_blend(px, kx, dc), v);
    }
    return v;
}

// element idx of a precision-3 packed weight: 4 sets in the split-bf16 layout, then the scaled fp32 OIHW weights (border pixels)
__device__ __forceinline__ void pack_up2f_element(const float* __restrict__ w, float* __restrict__ dst, int Cout, int Cin, float scale,
                                                  long idx) {
    const int K = Cin, N = Cout, MT = N / 16, nstep = K == 16 ? 5 : 9 * (K / 32);
    const long E = (long)nstep * MT * 2 * 64 * 8;
    if (idx >= 4 * E) {
        const long r = idx - 4 * E;
        dst[2 * E + r] = w[r] * scale;
        return;
    }
    const int set = (int)(idx / E);
    const long li = idx - set * E;
    const int j = li & 7, lane = (li >> 3) & 63, part = (li >> 9) & 1;
    const long r = li >> 10;
    const int mt = r % MT, step = r / MT;
    const int n = mt * 16 + (lane & 15), kk = 8 * (lane >> 4) + j;
    const int tap = K == 16 ? 2 * step + (kk >> 4) : step % 9;
    const int k = K == 16 ? (kk & 15) : (step / 9) * 32 + kk;
    float v = tap < 9 ? up2_folded_weight(w, Cin, n, k, set, tap) * scale : 0.f;
    const __bf16 hi = (__bf16)v;
    reinterpret_cast<__bf16*>(dst)[idx] = part == 0 ? hi : (__bf16)(v - (float)hi);
}

__global__ void pack_weights_up2f_kernel(const float* __restrict__ w, float* __restrict__ packed, int Cout, int Cin, float scale,
                                         long total) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < total) pack_up2f_element(w, packed, Cout, Cin, scale, idx);
}

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
// Measured and rejected (tools/clock_probe.py, fp32 16 -> 16 at 512x512, 177 us): the workgroups of a CU finish up to 75 us apart
// (the waves sharing a SIMD are served oldest first), but neither an atomic ticket queue (220 us: the compiler guards the tile
// loop's register hazards with s_waitcnt vmcnt(0..1), so every tile waited for the in-flight atomic) nor an over-decomposed grid
// that the dispatcher back-fills (2 / 4 / 8 tiles per workgroup: 184 / 181 / 180 us) is faster: the tail is not where the time goes.
// What the fp32 instances are short of is VALU issue: v_mfma_f32_16x16x4_f32 runs at the vector-FMA rate and does not overlap
// with other waves' VALU work -- with loads, stores, LDS staging and the epilogue compiled out one by one the kernel loses exactly
// the issue time of the instructions removed (161 / 143 / 132 us; the MFMAs alone need 123).
// ---------------------------------------------------------------------------------------------------------
struct TileRun { int t, t_end, step; };
__device__ __forceinline__ TileRun tile_run(int n_tiles) {      // this workgroup's tiles: t, t + step, ... < t_end
    const int xcd = blockIdx.x & 7, band = (n_tiles + 7) >> 3;
    TileRun r;
    r.step = gridDim.x >> 3;
    r.t = xcd * band + (blockIdx.x >> 3);
    r.t_end = min((xcd + 1) * band, n_tiles);
    return r;
}

inline int persistent_grid(int n_tiles, int resident) {
    static const int cap = [] { const char* e = getenv("NGAN_PERSIST_WG_PER_CU"); const int n = e ? atoi(e) : 0; return n > 0 ? n * 256 : 1 << 30; }();   // (A/B switch)
    if (resident > cap) resident = cap;
    int grid = resident < n_tiles ? resident : n_tiles;
    grid &= ~7;
    return grid < 8 ? 8 : grid;
}

// "The value must be in its registers HERE": an empty asm that reads and writes v (conv3x3_up2f_kernel).  (a) On the prefetched tile
// registers right after the MFMAs, before the epilogue's stores are issued: gfx9 counts loads and stores in ONE counter (vmcnt) and
// they may retire out of order with each other, so once stores are in flight the compiler can only wait for a load with vmcnt(0),
// i.e. by draining every store of the tile just written.  (b) On loop-invariant operands the compiler would otherwise re-load
// inside the loop.  (Measured on the older persistent kernel the same treatment was neutral to slightly negative -- its tile loop
// is bound by VALU issue and LDS, not by the store drain -- so it keeps the compiler's placement.)
__device__ __forceinline__ void pin_registers(float4& v) { asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); }
// ... and not before `dep` has been computed (an accumulator of the last MFMA: the scheduler may not hoist the wait above the MFMAs)
__device__ __forceinline__ void pin_registers_after(float4& v, float& dep) {
    asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w), "+v"(dep));
}

// float index of the hi half of (16-channel group g, channel quad c4) of tile pixel (ty, tx) in the split-bf16 image
template <int KG, int PLANE, int LP>
__device__ __forceinline__ int bf16_slot(int g, int c4, int ty, int tx) {
    const int slot = (KG == 1 ? (c4 >> 1) : (2 * g + (c4 >> 1))) ^ (((tx >> 2) & 1) << 1);
    return (ty * LP + tx) * 16 + slot * 4 + (c4 & 1) * 2;
}

// write 4 fp32 channels as 4 hi + 4 lo bf16 (8 bytes each); idx = bf16_slot(...)
template <int KG, int PLANE>
__device__ __forceinline__ void st_split(float* tile, int idx, float4 v) {
    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
    bf16x4 hi, lo;
    hi[0] = (__bf16)v.x; hi[1] = (__bf16)v.y; hi[2] = (__bf16)v.z; hi[3] = (__bf16)v.w;
    lo[0] = (__bf16)(v.x - (float)hi[0]); lo[1] = (__bf16)(v.y - (float)hi[1]);
    lo[2] = (__bf16)(v.z - (float)hi[2]); lo[3] = (__bf16)(v.w - (float)hi[3]);
    *reinterpret_cast<bf16x4*>(&tile[idx]) = hi;
    *reinterpret_cast<bf16x4*>(&tile[KG == 1 ? (idx ^ 8) : (idx + PLANE)]) = lo;
}

// LDS image of a tile: rows of LP = 40 pixels (>= 34 used), 16 floats per pixel, one plane per 16-channel group.
// The 16-byte quad c of pixel column X is stored at quad (c ^ 2*((X >> 2) & 1)): with that rotation the 16-lane groups of
// a ds_read_b128 (lanes = 16 consecutive pixels x 4 quads) touch 16 distinct 16-byte slots of a 256-byte bank row
// (conflict-free; the plain layout is 2-way).  A row pitch that is a multiple of 8 pixels keeps the rotation a
// function of the column only, so the read address is 3 registers (one per dx) + immediates.
// PREC = 1: split-bf16 arithmetic (3 x v_mfma_f32_16x16x32_bf16 per fp32 product group, fp32 accumulate): the fp32 input
// is split into hi/lo bf16 halves while the tile is staged; LDS image per pixel (K = 16): [hi c0-7][hi c8-15][lo c0-7]
// [lo c8-15] (16 B each, same 64 B and the same rotation as the fp32 image); K = 32: plane 0 = hi, plane 1 = lo.
// (the bilinear 32-channel instances also use 4 rows: with 8 their tile + low-res patch + weights come to 83 KB, one workgroup per CU)
constexpr int persist_tile_h(int MTW, int KG, int RES) { return (MTW * KG == 4 || (KG == 2 && RES == NGAN_RESAMPLE_UP2)) ? 4 : 8; }

#ifdef NGAN_CLOCK_PROBE
// Diagnostic build only (tools/clock_probe.py, `make clockprobe`): every workgroup of the persistent kernel stamps the shader clock
// (s_memtime) and the 100 MHz reference clock (s_memrealtime) on entry and exit; in-kernel clock = d(memtime) / d(realtime) x 100 MHz
// (MI355X_MICROARCH.md, DVFS give-back item 6).  The stamps go to a buffer nothing else reads.
__device__ unsigned long long g_clock_stamps[4 * 8192];
#endif

typedef float f32x2 __attribute__((ext_vector_type(2)));
struct f32p { f32x2 l, h; };                         // four floats as two register pairs (packed fp32 math)
__device__ __forceinline__ f32p operator+(f32p a, f32p b) { return {a.l + b.l, a.h + b.h}; }
// a - b as fma(b, -1, a): exact, and v_pk_fma_f32 exists where a packed subtraction does not (a v2f32 fsub is scalarised)
// (-1 comes from a register the optimiser cannot see through, or it folds the fma back into the subtraction)
__device__ __forceinline__ f32x2 opaque_minus_one() {
    f32x2 m1 = {-1.0f, -1.0f};
    asm("" : "+v"(m1));
    return m1;
}
__device__ __forceinline__ f32p psub(f32p a, f32p b, f32x2 m1) { return {__builtin_elementwise_fma(b.l, m1, a.l), __builtin_elementwise_fma(b.h, m1, a.h)}; }
__device__ __forceinline__ f32p pk2(f32x4 v) { return {(f32x2){v[0], v[1]}, (f32x2){v[2], v[3]}}; }
__device__ __forceinline__ f32x4 unpk2(f32p v) { return (f32x4){v.l[0], v.l[1], v.h[0], v.h[1]}; }

template <int MTW, int KG, int RES, int EPI, int OUTMODE, int PREC>
__global__ __launch_bounds__(256, PREC == 2 ? 2 : (MTW * KG == 1) ? (PREC ? 3 : 4) : 2) void conv3x3_persist_kernel(ConvArgs a, int n_tiles) {
    // PREC: 0 exact fp32 (direct), 1 split bf16, 2 exact fp32 by Winograd F(2x2, 3x3) (16 -> 16; the MFMA section of conv3x3_tile_kernel)
    constexpr bool BF = PREC == 1, WINO = PREC == 2;
    static_assert(!WINO || (MTW == 1 && KG == 1), "the Winograd form is built for the 16 -> 16 layers");
#ifdef NGAN_CLOCK_PROBE
    const unsigned long long clk0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
#endif
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

    for (int e = tid; e < W_ELEMS / 4; e += 256) st4(wl + e * 4, ld4(a.wp + e * 4));

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
#if defined(NGAN_EXP) && (NGAN_EXP & 2)
                stg[i] = make_float4((float)off, 1.f, 2.f, 3.f);          // timing experiment: no global loads
#else
                stg[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0));
#endif
            }
        }
    };
    if (t < t_end) issue(t);

    float4 bv[MTW], wimg[MTW];
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt) {
        bv[mt] = a.bias ? ld4(a.bias + mt * 16 + q * 4) : f4zero();
        wimg[mt] = EPI == EPI_TO_IMAGE ? ld4(a.ay + mt * 16 + q * 4) : f4zero();
    }
    const float inv_n = 1.0f / (float)N;

    while (t < t_end) {
        int b, y0, x0;
        decode(t, b, y0, x0);
#if !(defined(NGAN_EXP) && (NGAN_EXP & 16))
        __syncthreads();   // previous tile's MFMAs have finished reading `tile`
#endif
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
#if defined(NGAN_EXP) && (NGAN_EXP & 8)
                if (tid + i * 256 < N_SRC && stg[i].y == 77.f) {          // timing experiment: the staging stores never execute
#else
                if (tid + i * 256 < N_SRC) {
#endif
                    if (BF) st_split<KG, PLANE>(tile, s_lds[i], stg[i]);
                    else st4(&tile[s_lds[i]], stg[i]);
                }
        }
#if !(defined(NGAN_EXP) && (NGAN_EXP & 16))
        __syncthreads();
#endif
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
#if defined(NGAN_EXP) && (NGAN_EXP & 4)
        {   // timing experiment: no epilogue; the accumulators are consumed by a store that never happens
            float sum = 0.f;
#pragma unroll
            for (int pg = 0; pg < PGW; ++pg)
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt) sum += acc[pg][mt][0] + acc[pg][mt][1] + acc[pg][mt][2] + acc[pg][mt][3];
            if (sum == 123.456f) a.y[tid] = sum;
        }
#else
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
#if defined(NGAN_EXP) && (NGAN_EXP & 1)
                    const unsigned off = (valid && v[0].x == 123.456f) ? (unsigned)(((gy * a.W + gx) * N + q * 4) * 4) : OOB;   // timing experiment: no stores land
#else
                    const unsigned off = valid ? (unsigned)(((gy * a.W + gx) * N + q * 4) * 4) : OOB;
#endif
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
#endif
        t = tn;
    }
#ifdef NGAN_CLOCK_PROBE
    if (threadIdx.x == 0 && blockIdx.x < 8192) {
        g_clock_stamps[blockIdx.x * 4 + 0] = clk0; g_clock_stamps[blockIdx.x * 4 + 1] = rt0;
        g_clock_stamps[blockIdx.x * 4 + 2] = __builtin_amdgcn_s_memtime(); g_clock_stamps[blockIdx.x * 4 + 3] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------
// conv3x3_tile_kernel: the persistent kernel for plain (not resampled) input on images whose width is a multiple of the 32-pixel
// tile, rebuilt around one measurement (tools/clock_probe.py + the NGAN_EXP builds): v_mfma_f32_16x16x4_f32 runs on the vector
// FMA lanes, so every VALU instruction any wave of the SIMD issues is 4 cycles the matrix instructions do not get -- the fp32
// 16 -> 16 layer spent 190 VALU instructions per 144 MFMAs of a wave's tile, 80 of them integer address / bounds arithmetic.
// Here the per-tile arithmetic is scalar:
//   * tile loads: per-lane byte offsets are tile-invariant constants; the tile moves the descriptor's BASE (64-bit SALU add), and
//     conv padding is done by whole load instructions -- the staging order puts the 32 interior columns into LPG loads per channel
//     group and the two halo columns into one extra load, so the top halo row is "load 0 of waves 0-1" (a zero-record descriptor
//     when the tile touches the image top), the bottom is the descriptor's range check, and left / right only touch the halo load
//     (a 3-instruction lane select);
//   * output stores, PixelNorm-backward operand loads, norms: constant per-lane offsets + a scalar tile offset (one v_add each);
//   * the bias is the accumulators' initial value (C operand of the first MFMA), the epilogue arithmetic is written on float2 so
//     that it compiles to packed v_pk_mul / v_pk_fma.
// Same LDS image and MFMA order as conv3x3_persist_kernel; bit-identical to it without a bias (with one, the bias is added first
// instead of last).
// ---------------------------------------------------------------------------------------------------------
#ifndef NGAN_TILE_PRE
#define NGAN_TILE_PRE 1
#endif

template <int MTW, int KG, int EPI, int OUTMODE, int PREC>
__global__ __launch_bounds__(256, PREC == 2 ? 2 : (MTW * KG == 1) ? (PREC ? 3 : 4) : 2) void conv3x3_tile_kernel(ConvArgs a, int n_tiles) {
    // PREC: 0 exact fp32 (direct), 1 split bf16, 2 exact fp32 by Winograd F(2x2, 3x3) (16 -> 16 only; see the MFMA section)
    constexpr bool BF = PREC == 1, WINO = PREC == 2;
    static_assert(!WINO || (MTW == 1 && KG == 1), "the Winograd form is built for the 16 -> 16 layers");
#ifdef NGAN_CLOCK_PROBE
    const unsigned long long clk0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
#endif
    constexpr int THc = persist_tile_h(MTW, KG, 0), PGW = THc / 2, RPW = THc / 4;
    constexpr int HH_ = THc + 2, LP = 40;
    constexpr int NSTEP = KG == 1 ? 5 : 9;
    constexpr int W_ELEMS = BF ? NSTEP * MTW * 2 * 256 : (WINO ? 16 * 256 : 9 * KG * MTW * 256), PLANE = HH_ * LP * 16, TILE_ELEMS = KG * PLANE;
    constexpr int LPG = HH_ / 2;                 // interior loads per 16-channel group: HH_ rows x 32 columns x 4 quads / 256 threads
    constexpr int NL = KG * LPG, NST = NL + 1;   // + one load for the two halo columns
    constexpr int N_HALO = KG * 2 * HH_ * 4;     // its active lanes: (group, side, row, quad)
    constexpr int K = KG * 16, N = MTW * 16;
    constexpr unsigned OOB = 0xFFFFFFF0u;
    static_assert(HH_ % 2 == 0 && N_HALO <= 256, "staging layout");
    __shared__ __attribute__((aligned(16))) float smem[W_ELEMS + TILE_ELEMS];
    float* wl = smem;
    float* tile = smem + W_ELEMS;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int p = lane & 15, q = lane >> 4;

    for (int e = tid; e < W_ELEMS / 4; e += 256) st4(wl + e * 4, ld4(a.wp + e * 4));

    const TileRun run = tile_run(n_tiles);
    int t = run.t;
    const int t_end = run.t_end;

    auto lds_slot = [&](int g, int c4, int ty, int tx) {
        if (WINO) {
            // even and odd columns in separate halves of a row (a lane reads columns 2p + b): position p + const, the same
            // conflict-free pattern as the direct form's p + dx
            const int pos = (tx >> 1) + (tx & 1) * (LP / 2);
            return (ty * LP + pos) * 16 + ((c4 ^ (((pos >> 2) & 1) << 1)) << 2);
        }
        return BF ? bf16_slot<KG, PLANE, LP>(g, c4, ty, tx) : g * PLANE + (ty * LP + tx) * 16 + ((c4 ^ (((tx >> 2) & 1) << 1)) << 2);
    };
    // ---- tile-invariant staging constants: byte offset from the halo origin (y0 - 1, x0 - 1), LDS float index ----
    unsigned s_voff[NST];
    int s_lds[NST];
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        const int g = i / LPG, e = tid + (i % LPG) * 256;
        const int c4 = e & 3, pix = e >> 2, ty = pix >> 5, tx = (pix & 31) + 1;
        s_voff[i] = (unsigned)(((ty * a.W + tx) * K + g * 16 + c4 * 4) * 4);
        s_lds[i] = lds_slot(g, c4, ty, tx);
    }
    int h_bits;                                   // halo load: 1 = left column, 2 = right column, 4 = top row, 8 = unused lane
    {
        const int c4 = tid & 3, r = (tid >> 2) % HH_, sg = (tid >> 2) / HH_, side = sg & 1, g = sg >> 1;
        const bool used = tid < N_HALO;
        const int tx = side ? 33 : 0;
        s_voff[NL] = used ? (unsigned)(((r * a.W + tx) * K + (used ? g : 0) * 16 + c4 * 4) * 4) : OOB;
        s_lds[NL] = lds_slot(used ? g : 0, c4, r, tx);
        h_bits = used ? ((side ? 2 : 1) | (r == 0 ? 4 : 0)) : 8;
    }
    int rd[3];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) rd[dx] = (p + dx) * 16 + ((q ^ ((((p + dx) >> 2) & 1) << 1)) << 2);
    int wrd[4];                                    // Winograd: LDS float index of column 2p + b of a tile row, channel quad q
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const int pos = p + (b >> 1) + (b & 1) * (LP / 2);
        wrd[b] = pos * 16 + ((q ^ (((pos >> 2) & 1) << 1)) << 2);
    }
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
    // ---- tile-invariant epilogue constants: this lane's output byte offsets from the tile origin ----
    constexpr int OS = OUTMODE ? 2 : 1;            // the pool-adjoint store writes a 2x2 block of a (2H, 2W) tensor per computed pixel
    const int Wo = OS * a.W;
    unsigned e_voff[PGW];
#pragma unroll
    for (int pg = 0; pg < PGW; ++pg) {
        // direct forms: pixel group = 16 consecutive pixels of a row; Winograd: pixel (a, b) of this lane's 2x2 output tile
        const int row = WINO ? 2 * wave + (pg >> 1) : wave * RPW + (pg >> 1), col = WINO ? 2 * p + (pg & 1) : (pg & 1) * 16 + p;
        e_voff[pg] = (unsigned)((((OS * row) * Wo + OS * col) * N + q * 4) * 4);
    }
    constexpr int NSHIFT = MTW == 1 ? 4 : 5;       // pixel index * 4 bytes = (e_voff - 16 q) / N
    unsigned t_voff = 0;                           // ToImage: lane group q finishes pixel group q
    if (EPI == EPI_TO_IMAGE) t_voff = WINO ? (unsigned)(((2 * wave + (q >> 1)) * a.W + 2 * p + (q & 1)) * 4)
                                           : (unsigned)(((wave * RPW + (q >> 1)) * a.W + (q & 1) * 16 + p) * 4);

    auto decode = [&](int tt, int& b, int& y0, int& x0) {
        const int txi = tt % a.tiles_x; tt /= a.tiles_x;
        const int tyi = tt % a.tiles_y;
        b = tt / a.tiles_y;
        y0 = tyi * THc; x0 = txi * 32;
    };
    const unsigned img_bytes = (unsigned)(a.H * a.W * K) * 4u;
    float4 stg[NST];
    auto issue = [&](int tt) {
        int b, y0, x0;
        decode(tt, b, y0, x0);
        const int soff = ((y0 - 1) * a.W + (x0 - 1)) * K * 4;                 // negative on the top row / for the first tile
        const char* base = reinterpret_cast<const char*>(a.x + (long)b * a.H * a.W * K) + soff;
        const unsigned nrec = img_bytes - (unsigned)soff;                     // bytes from `base` to the end of the image
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, nrec, 0x00020000);
        // the top halo row is the first interior load of waves 0 and 1 (+ flagged lanes of the halo load): above the image it reads
        // through a descriptor with no records, i.e. zeros
        const __amdgpu_buffer_rsrc_t rsrc_top = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, (y0 == 0 && wave < 2) ? 0u : nrec, 0x00020000);
#pragma unroll
        for (int i = 0; i < NL; ++i) {
#if defined(NGAN_EXP) && (NGAN_EXP & 2)
            stg[i] = make_float4((float)s_voff[i], 1.f, 2.f, 3.f);          // timing experiment: no global loads
#else
            stg[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128((i % LPG == 0) ? rsrc_top : rsrc, s_voff[i], 0, 0));
#endif
        }
        const int bad = (x0 == 0 ? 1 : 0) | (x0 + 32 >= a.W ? 2 : 0) | (y0 == 0 ? 4 : 0) | 8;
        const unsigned hoff = (h_bits & bad) ? OOB : s_voff[NL];
#if defined(NGAN_EXP) && (NGAN_EXP & 2)
        stg[NL] = make_float4((float)hoff, 1.f, 2.f, 3.f);
#else
        stg[NL] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, hoff, 0, 0));
#endif
    };
    if (t < t_end) issue(t);
#if defined(NGAN_EXP) && (NGAN_EXP & 32)
    {   // timing experiment: stagger the workgroups that (probably) share a CU by a quarter of a tile round each
        const int k = (blockIdx.x >> 8) & 3;
        for (int i = 0; i < k * 10; ++i) __builtin_amdgcn_s_sleep(8);        // 10 x 8 x 64 cycles = 5120 cycles per step
    }
#endif

    f32x4 bvec[MTW];
    float4 wimg[MTW];
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt) {
        const float4 b4 = a.bias ? ld4(a.bias + mt * 16 + q * 4) : f4zero();
        bvec[mt] = (f32x4){b4.x, b4.y, b4.z, b4.w};
        wimg[mt] = EPI == EPI_TO_IMAGE ? ld4(a.ay + mt * 16 + q * 4) : f4zero();
    }
    const float inv_n = 1.0f / (float)N;
    const f32x2 slope2 = {a.slope, a.slope};

    while (t < t_end) {
        int b, y0, x0;
        decode(t, b, y0, x0);
#if !(defined(NGAN_EXP) && (NGAN_EXP & 16))
        __syncthreads();   // previous tile's MFMAs have finished reading `tile`
#else
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // timing experiment: no workgroup barrier (wrong results)
#endif
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            if (BF) st_split<KG, PLANE>(tile, s_lds[i], stg[i]);
            else st4(&tile[s_lds[i]], stg[i]);
        }
        if (tid < N_HALO) {
            if (BF) st_split<KG, PLANE>(tile, s_lds[NL], stg[NL]);
            else st4(&tile[s_lds[NL]], stg[NL]);
        }
#if !(defined(NGAN_EXP) && (NGAN_EXP & 16))
        __syncthreads();
#else
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // timing experiment: no workgroup barrier (wrong results)
#endif
        const int tn = t + run.step;
        if (tn < t_end) issue(tn);   // in flight while this tile is computed

        // ---- per-tile scalars of the epilogue.  The tile's byte offset is ADDED to the per-lane constants (one v_add per access)
        // instead of riding in the buffer instructions' soffset field: a buffer_store_dwordx4 with an SGPR soffset reads its data
        // registers late, the compiler (whose hazard table exempts exactly that form) puts no wait state behind it, and the next
        // VALU write into those registers reached memory instead -- single components of the last four lanes of a pixel group,
        // in a fraction of a percent of the tiles (tools/dbg_tile.py; found the same way: bit-comparison with the old kernel) ----
        const long img = (long)b * a.H * a.W;
        const int pix0 = (OS * y0) * Wo + OS * x0;                       // first output pixel of the tile inside its image
        const unsigned y_soff = (unsigned)pix0 * (N * 4), p_soff = (unsigned)(y0 * a.W + x0) * 4u;
        const unsigned out_bytes = (unsigned)(OS * a.H * Wo * N) * 4u, px_bytes = (unsigned)(a.H * a.W) * 4u;
        __amdgpu_buffer_rsrc_t y_rsrc, rn_rsrc, ay_rsrc, arn_rsrc;
        if (EPI != EPI_TO_IMAGE || a.y) y_rsrc = __builtin_amdgcn_make_buffer_rsrc(a.y + img * (OS * OS) * N, 0, out_bytes, 0x00020000);
        // (no stored activation -- the inference form of epilogue 3 -- means no stored norm either: a descriptor without records)
        if (EPI == EPI_LRELU_PN || EPI == EPI_TO_IMAGE)
            rn_rsrc = __builtin_amdgcn_make_buffer_rsrc(a.rn + img, 0, (EPI == EPI_LRELU_PN || a.y) ? px_bytes : 0u, 0x00020000);
        if (EPI == EPI_PN_BWD) {
            ay_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.ay) + img * (OS * OS) * N, 0, out_bytes, 0x00020000);
            arn_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.arn) + img * (OS * OS), 0, (unsigned)(OS * a.H * Wo) * 4u, 0x00020000);
        }
        // PixelNorm-backward operands (same shape as the output): requested before the MFMAs where registers allow, else before the
        // first store of the epilogue (a load issued behind a store can only be awaited by draining that store)
        constexpr bool PNB = EPI == EPI_PN_BWD && OUTMODE == 0;
        constexpr bool PRE = PNB && MTW * KG > 1 && NGAN_TILE_PRE;
        float4 yy[PNB ? PGW : 1][MTW];
        float rr[PNB ? PGW : 1];
        auto load_pn_operands = [&]() {
#pragma unroll
            for (int pg = 0; pg < PGW; ++pg) {
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt)
                    yy[PNB ? pg : 0][mt] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(ay_rsrc, e_voff[pg] + y_soff + mt * 64, 0, 0));
                rr[PNB ? pg : 0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(arn_rsrc, ((e_voff[pg] - q * 16) >> NSHIFT) + p_soff, 0, 0));
            }
        };
        if (PRE) load_pn_operands();

        f32x4 acc[PGW][MTW];
#pragma unroll
        for (int pg = 0; pg < PGW; ++pg)
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt) acc[pg][mt] = bvec[mt];       // the bias: C operand of the first MFMA
        if (WINO) {
            // Winograd F(2x2, 3x3) (Lavin & Gray): Y = A^T [ (G g G^T) . (B^T d B) ] A per 2x2 output tile and channel pair, the
            // element-wise product summed over input channels = 16 small GEMMs (one per position (u, v) of the 4x4 transformed
            // tile), 64 v_mfma_f32_16x16x4_f32 per wave and 64 output pixels instead of 144.  Wave = one row of 16 output tiles;
            // lane (p, q) owns tile p and, as a B operand, input channels 4q..4q+3 (component s feeds MFMA s, as in the direct
            // form), as a D operand output channels 4q..4q+3.  Both transforms are therefore lane-local: B^T d B on the 4x4 input
            // patch it reads itself (16 ds_read_b128), A^T M A on its own accumulators.  G g G^T is done by the packing kernel.
            //   B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]      A^T = [1 1 1 0; 0 1 -1 -1]
            // (the transforms are written on register PAIRS: v_pk_add_f32 does two of the four channels per instruction -- the
            // compiler left the float4 form as 184 scalar adds, and in this kernel a VALU instruction costs matrix time)
            const f32x2 m1 = opaque_minus_one();
            f32p bd[4][4];                           // B^T d: rows transformed, columns still in pixel space
            {
                const float* trow = tile + (2 * wave) * (LP * 16);
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const f32p d0 = pk2(*reinterpret_cast<const f32x4*>(trow + 0 * LP * 16 + wrd[b])), d1 = pk2(*reinterpret_cast<const f32x4*>(trow + 1 * LP * 16 + wrd[b]));
                    const f32p d2 = pk2(*reinterpret_cast<const f32x4*>(trow + 2 * LP * 16 + wrd[b])), d3 = pk2(*reinterpret_cast<const f32x4*>(trow + 3 * LP * 16 + wrd[b]));
                    bd[0][b] = psub(d0, d2, m1); bd[1][b] = d1 + d2; bd[2][b] = psub(d2, d1, m1); bd[3][b] = psub(d1, d3, m1);
                }
            }
            f32p ta[2][4];                           // A^T M: output rows, columns still in transform space
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                f32x4 m[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) m[u] = (u == 1 && v == 1) ? bvec[0] : (f32x4){0.f, 0.f, 0.f, 0.f};   // A^T e11 A = all ones: the bias
                f32p vv[4];                          // (B^T d B)[u][v]
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
#if defined(NGAN_EXP) && (NGAN_EXP & 4)
        {   // timing experiment: no epilogue; the accumulators are consumed by a store that never happens
            float sum = 0.f;
#pragma unroll
            for (int pg = 0; pg < PGW; ++pg)
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt) sum += acc[pg][mt][0] + acc[pg][mt][1] + acc[pg][mt][2] + acc[pg][mt][3];
            if (sum == 123.456f) a.y[tid] = sum;
        }
#else
        // ---- epilogue ----
        if (PNB && !PRE) load_pn_operands();
        // PixelNorm-backward operands: wait for everything in flight (the operand loads and the next tile, issued a tile's worth of
        // MFMAs ago) BEFORE the first store goes out.  Once stores are in flight, loads and stores of gfx9 retire out of order with each other
        // under one counter, and the counted waits the compiler emits for the older loads returned early: wrong last dwords in the
        // last lanes of a pixel group (tools/dbg_epi2.py; without the prefetch the kernel is bit-identical to conv3x3_persist_kernel)
        if (PNB) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        float timg = 0.f;
#pragma unroll
        for (int pg = 0; pg < PGW; ++pg) {
            f32x2 lo[MTW], hi[MTW];                 // channels (4q, 4q+1) and (4q+2, 4q+3) of each 16-channel tile
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt) {
                lo[mt] = (f32x2){acc[pg][mt][0], acc[pg][mt][1]};
                hi[mt] = (f32x2){acc[pg][mt][2], acc[pg][mt][3]};
            }
            if (EPI == EPI_LRELU_PN || EPI == EPI_TO_IMAGE) {
                f32x2 sq = {0.f, 0.f};
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt) {
                    const f32x2 sl = lo[mt] * slope2, sh = hi[mt] * slope2;            // LeakyReLU, 0 <= slope <= 1 (vmax1: one v_max each;
                    lo[mt] = (f32x2){vmax1(lo[mt].x, sl.x), vmax1(lo[mt].y, sl.y)};     //  __builtin_elementwise_max adds a canonicalising v_max)
                    hi[mt] = (f32x2){vmax1(hi[mt].x, sh.x), vmax1(hi[mt].y, sh.y)};
                    sq = mt == 0 ? lo[mt] * lo[mt] : __builtin_elementwise_fma(lo[mt], lo[mt], sq);
                    sq = __builtin_elementwise_fma(hi[mt], hi[mt], sq);
                }
                float ss = sq.x + sq.y;
                ss = sum_rows4(ss);
                const float m = ss * inv_n + a.eps;
                const float inv = __builtin_amdgcn_rsqf(m);
                const f32x2 inv2 = {inv, inv};
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt) { lo[mt] *= inv2; hi[mt] *= inv2; }
                // the norm: one lane per pixel stores, the others' offset is out of range (a branch here would also cut the epilogue
                // into basic blocks and keep the four pixel groups' reduction chains from being scheduled side by side)
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, m * inv), rn_rsrc, q == 0 ? (e_voff[pg] >> NSHIFT) + p_soff : OOB, 0, 0);
            }
            if (EPI == EPI_TO_IMAGE) {
                f32x2 d2 = {0.f, 0.f};
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt) {
                    d2 = __builtin_elementwise_fma(lo[mt], (f32x2){wimg[mt].x, wimg[mt].y}, d2);
                    d2 = __builtin_elementwise_fma(hi[mt], (f32x2){wimg[mt].z, wimg[mt].w}, d2);
                }
                float d = d2.x + d2.y;
                d = sum_rows4(d);
                if (q == pg) timg = d;          // all four q-lanes hold pixel group pg's sum; lane group q keeps the one it will finish
            }
            if (PNB) {
                // backward of the LeakyReLU -> PixelNorm that produced this layer's input, applied to the gradient just computed
                float s = 0.f;
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt) {
                    const float4 y4 = yy[PNB ? pg : 0][mt];
                    s += lo[mt].x * y4.x + lo[mt].y * y4.y + hi[mt].x * y4.z + hi[mt].y * y4.w;
                }
                s = sum_rows4(s);
                s *= inv_n;
                const float inv_r = 1.0f / rr[PNB ? pg : 0];
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt) {
                    const float4 o = pn_bwd4(make_float4(lo[mt].x, lo[mt].y, hi[mt].x, hi[mt].y), yy[PNB ? pg : 0][mt], s, inv_r, a.slope);
                    lo[mt] = (f32x2){o.x, o.y}; hi[mt] = (f32x2){o.z, o.w};
                }
            }
            if (OUTMODE == 0) {
                if (EPI != EPI_TO_IMAGE || a.y) {
#pragma unroll
                    for (int mt = 0; mt < MTW; ++mt) {
                        const u32x4 v = __builtin_bit_cast(u32x4, make_float4(lo[mt].x, lo[mt].y, hi[mt].x, hi[mt].y));
#if defined(NGAN_EXP) && (NGAN_EXP & 1)
                        __builtin_amdgcn_raw_buffer_store_b128(v, y_rsrc, lo[0].x == 123.456f ? e_voff[pg] + y_soff + mt * 64 : OOB, 0, 0);
#else
                        __builtin_amdgcn_raw_buffer_store_b128(v, y_rsrc, e_voff[pg] + y_soff + mt * 64, 0, 0);
#endif
                    }
                }
            } else {
                // pool-adjoint store: the value * 0.25 goes to the 2x2 block (2gy + i, 2gx + j); with the PixelNorm-backward epilogue
                // each of the four pixels has its own operands -- all loads before the first store
                const unsigned row1 = y_soff + (unsigned)(Wo * N * 4);
                const unsigned prow1 = (unsigned)pix0 * 4u + (unsigned)(Wo * 4);
                const f32x2 quarter = {0.25f, 0.25f};
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt) { lo[mt] *= quarter; hi[mt] *= quarter; }
                float4 y4s[EPI == EPI_PN_BWD ? 4 : 1][MTW];
                float r4s[EPI == EPI_PN_BWD ? 4 : 1];
                if (EPI == EPI_PN_BWD) {
#pragma unroll
                    for (int sub = 0; sub < 4; ++sub) {
#pragma unroll
                        for (int mt = 0; mt < MTW; ++mt)
                            y4s[sub][mt] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(ay_rsrc, e_voff[pg] + ((sub >> 1) ? row1 : y_soff) +
                                                                                                         mt * 64 + (sub & 1) * (N * 4), 0, 0));
                        r4s[sub] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(arn_rsrc, ((e_voff[pg] - q * 16) >> NSHIFT) +
                                                                                                ((sub >> 1) ? prow1 : (unsigned)pix0 * 4u) + (sub & 1) * 4, 0, 0));
                    }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (loads vs. younger stores: see the plain store path above)
                }
#pragma unroll
                for (int sub = 0; sub < 4; ++sub) {
                    float4 o4[MTW];
#pragma unroll
                    for (int mt = 0; mt < MTW; ++mt) o4[mt] = make_float4(lo[mt].x, lo[mt].y, hi[mt].x, hi[mt].y);
                    if (EPI == EPI_PN_BWD) {
                        float s = 0.f;
#pragma unroll
                        for (int mt = 0; mt < MTW; ++mt) s += f4dot(o4[mt], y4s[EPI == EPI_PN_BWD ? sub : 0][mt]);
                        s = sum_rows4(s);
                        s *= inv_n;
                        const float inv_r = 1.0f / r4s[EPI == EPI_PN_BWD ? sub : 0];
#pragma unroll
                        for (int mt = 0; mt < MTW; ++mt) o4[mt] = pn_bwd4(o4[mt], y4s[EPI == EPI_PN_BWD ? sub : 0][mt], s, inv_r, a.slope);
                    }
#pragma unroll
                    for (int mt = 0; mt < MTW; ++mt)
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o4[mt]), y_rsrc,
                                                               e_voff[pg] + ((sub >> 1) ? row1 : y_soff) + mt * 64 + (sub & 1) * (N * 4), 0, 0);
                }
            }
        }
        if (EPI == EPI_TO_IMAGE) {
            // one tanh per lane instead of four: lane group q finishes pixel group q (same tanhf as the standalone ToImage kernel)
            const float tv = tanhf(timg);
            const __amdgpu_buffer_rsrc_t t_rsrc = __builtin_amdgcn_make_buffer_rsrc(a.aout + img, 0, px_bytes, 0x00020000);
            if (q < PGW) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, tv), t_rsrc, t_voff + p_soff, 0, 0);
        }
#endif
        t = tn;
    }
#ifdef NGAN_CLOCK_PROBE
    if (threadIdx.x == 0 && blockIdx.x < 8192) {
        g_clock_stamps[blockIdx.x * 4 + 0] = clk0; g_clock_stamps[blockIdx.x * 4 + 1] = rt0;
        g_clock_stamps[blockIdx.x * 4 + 2] = __builtin_amdgcn_s_memtime(); g_clock_stamps[blockIdx.x * 4 + 3] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}

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
    const float4 bv = a.bias ? ld4(a.bias + q * 4) : f4zero();
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

inline bool tile_kernel_ok(const ConvArgs& a) {      // conv3x3_tile_kernel: whole tiles along x
    static const bool off = getenv("NGAN_TILE_KERNEL") && getenv("NGAN_TILE_KERNEL")[0] == '0';     // A/B switch
    return !off && a.W % 32 == 0;
}

template <int MTW, int KG, int EPI, int OUTMODE, int PREC>
int launch_tile(ConvArgs a, hipStream_t s) {
    a.tiles_x = a.W / 32;
    a.tiles_y = ngan::ceil_div(a.H, persist_tile_h(MTW, KG, 0));
    const int n_tiles = a.B * a.tiles_x * a.tiles_y;
    static int per_cu = 0;
    if (per_cu == 0) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, conv3x3_tile_kernel<MTW, KG, EPI, OUTMODE, PREC>, 256, 0) != hipSuccess || n < 1) n = 1;
        per_cu = n > 4 ? 4 : n;
    }
    const int grid = persistent_grid(n_tiles, 256 * per_cu);
    hipLaunchKernelGGL((conv3x3_tile_kernel<MTW, KG, EPI, OUTMODE, PREC>), dim3(grid), dim3(256), 0, s, a, n_tiles);
    return ngan::launch_status("ngan_conv3x3_fwd(tile)");
}

template <int MTW, int KG, int RES, int EPI, int OUTMODE, int PREC>
int launch_persist(ConvArgs a, hipStream_t s) {
    if (RES == 0 && tile_kernel_ok(a)) return launch_tile<MTW, KG, EPI, OUTMODE, PREC>(a, s);
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
int dispatch_persist(const ConvArgs& a, int res, int epi, int outmode, int prec, hipStream_t s) {
    return prec ? dispatch_persist2<MTW, KG, 1>(a, res, epi, outmode, s) : dispatch_persist2<MTW, KG, 0>(a, res, epi, outmode, s);
}


// tile shapes: {MTW, WN, PGW, PCG}.  Per output-channel count, ordered from the largest pixel tile to the smallest.
struct TileCfg { int mtw, wn, pgw, pcg; };
constexpr TileCfg kCfg[4][3] = {
    {{1, 1, 4, 2}, {1, 1, 1, 1}, {1, 1, 1, 1}},   // N = 16 : 8x32 | 4x16
    {{2, 1, 4, 2}, {2, 1, 1, 1}, {1, 2, 1, 1}},   // N = 32 : 8x32 | 4x16 | 2x16
    {{4, 1, 4, 2}, {2, 2, 2, 1}, {1, 4, 1, 1}},   // N = 64 : 8x32 | 4x16 | 1x16
    {{8, 1, 4, 2}, {2, 4, 4, 1}, {2, 4, 1, 1}},   // N = 128: 8x32 | 4x16 | 1x16
};
constexpr int kMinBlocks = 256;  // one workgroup per CU at least

inline void cfg_tile(const TileCfg& c, int& th, int& tw) {
    const int npg = (4 / c.wn) * c.pgw;
    tw = c.pcg * 16;
    th = npg / c.pcg;
}

// pick the largest tile that still gives kMinBlocks workgroups and wastes < 30 % of its pixels
int pick_cfg(int mti, int B, int H, int W) {
    for (int i = 0; i < 3; ++i) {
        int th, tw;
        cfg_tile(kCfg[mti][i], th, tw);
        const long nwg = (long)B * ngan::ceil_div(H, th) * ngan::ceil_div(W, tw);
        const double waste = (double)nwg * th * tw / ((double)B * H * W);
        if (nwg >= kMinBlocks && waste <= 1.3) return i;
    }
    return 2;
}

// the persistent kernel (and with it the split-bf16 arithmetic) applies to few-channel layers on large images
inline bool persist_eligible(int B, int H, int W, int K, int N, int resample) {
    return N <= 32 && K <= 32 && resample != NGAN_RESAMPLE_POOL2 && pick_cfg(N / 16 - 1, B, H, W) == 0;
}

// bilinear x2 folded into the weights (precision code 3): the persistent kernel's shapes with 16 outputs, and an interior to speak of
inline bool up2f_eligible(int B, int H, int W, int K, int N, int resample) {
    static const bool enabled = !(getenv("NGAN_UP2_FOLDED") && getenv("NGAN_UP2_FOLDED")[0] == '0');   // A/B switch for measurements
    return enabled && resample == NGAN_RESAMPLE_UP2 && N == 16 && (K == 16 || K == 32) && H % 2 == 0 && W % 2 == 0 && H >= 16 && W >= 32 &&
           persist_eligible(B, H, W, K, N, resample);
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

// ---------------------------------------------------------------------------------------------------------
// weight gradient.  D[co][ci] (per tap) += sum_pixels g[pix][co] * xin[pix + tap][ci]: the MFMA contraction
// runs over pixels (4 per instruction).  A block owns a (<=32 couts) x (<=32 cins) slice of the output and
// walks pixel tiles grid-stride; the 4 waves split the tile rows, are summed through LDS at the end, and
// each block writes one partial slab; wgrad_reduce_kernel sums the slabs in a fixed order (deterministic).
// ---------------------------------------------------------------------------------------------------------
struct WgradArgs {
    const float* x; const float* g; float* partial;
    int B, H, W, K, N, tiles_x, tiles_y, n_tiles, n_ci_slices;
};

template <int CS>
__device__ __forceinline__ int swz(int pix, int c) { return CS == 32 ? (c ^ ((pix & 1) << 4)) : c; }

template <int COT, int CIT, int RES, int TW>
__global__ __launch_bounds__(256, (COT * CIT == 1) ? 4 : 2) void wgrad_kernel(WgradArgs a) {
    constexpr int TH = 256 / TW, HALO_H = TH + 2, HALO_W = TW + 2;
    constexpr int CO_S = COT * 16, CI_S = CIT * 16;
    constexpr int WO = COT * CIT, WR = 4 / WO, RPW = TH / WR;   // waves along outputs / along rows; rows per wave
    constexpr int G_ELEMS = TH * TW * CO_S, X_ELEMS = HALO_H * HALO_W * CI_S;
    constexpr int RED_ELEMS = 4 * 9 * 64 * 4;
    constexpr int SMEM = (G_ELEMS + X_ELEMS) > RED_ELEMS ? (G_ELEMS + X_ELEMS) : RED_ELEMS;
    constexpr int NG = TH * TW * (CO_S / 4) / 256, NXI = HALO_H * HALO_W * (CI_S / 4), NX = (NXI + 255) / 256;
    __shared__ __attribute__((aligned(16))) float smem[SMEM];
    float* g_lds = smem;
    float* x_lds = smem + G_ELEMS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int p = lane & 15, q = lane >> 4;
    const int wo = wave % WO, wr = wave / WO;
    const int cot = wo / CIT, cit = wo % CIT;        // this wave's 16x16 (cout, cin) sub-slice, all 9 taps
    const int slice = blockIdx.y;
    const int co0 = (slice / a.n_ci_slices) * CO_S, ci0 = (slice % a.n_ci_slices) * CI_S;

    // tile-invariant staging descriptors
    int g_r[NG], g_c[NG], g_ch[NG], g_l[NG];
#pragma unroll
    for (int i = 0; i < NG; ++i) {
        const int e = tid + i * 256;
        const int pix = e / (CO_S / 4), c4 = e % (CO_S / 4);
        g_r[i] = pix / TW; g_c[i] = pix % TW; g_ch[i] = co0 + c4 * 4;
        g_l[i] = pix * CO_S + swz<CO_S>(pix, c4 * 4);
    }
    int x_r[NX], x_c[NX], x_ch[NX], x_l[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        const int e = tid + i * 256;
        const int pix = e / (CI_S / 4), c4 = e % (CI_S / 4);
        x_r[i] = e < NXI ? pix / HALO_W - 1 : -1000; x_c[i] = pix % HALO_W - 1; x_ch[i] = ci0 + c4 * 4;
        x_l[i] = pix * CI_S + swz<CI_S>(pix, c4 * 4);
    }

    f32x4 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    float4 gst[NG], xst[NX];
    auto issue = [&](int tile) {
        int t = tile;
        const int txi = t % a.tiles_x; t /= a.tiles_x;
        const int tyi = t % a.tiles_y;
        const int b = t / a.tiles_y;
        const int y0 = tyi * TH, x0 = txi * TW;
        const float* gb = a.g + (long)b * a.H * a.W * a.N;
#pragma unroll
        for (int i = 0; i < NG; ++i) {
            const int gy = y0 + g_r[i], gx = x0 + g_c[i];
            gst[i] = (gy < a.H && gx < a.W) ? ld4(gb + ((long)gy * a.W + gx) * a.N + g_ch[i]) : f4zero();
        }
#pragma unroll
        for (int i = 0; i < NX; ++i)
            xst[i] = x_r[i] > -1000 ? load_resampled<RES>(a.x, b, y0 + x_r[i], x0 + x_c[i], x_ch[i], a.H, a.W, a.K) : f4zero();
    };

    int tile = blockIdx.x;
    if (tile < a.n_tiles) issue(tile);
    while (tile < a.n_tiles) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NG; ++i) st4(&g_lds[g_l[i]], gst[i]);
#pragma unroll
        for (int i = 0; i < NX; ++i)
            if (x_r[i] > -1000) st4(&x_lds[x_l[i]], xst[i]);
        __syncthreads();
        const int tn = tile + gridDim.x;
        if (tn < a.n_tiles) issue(tn);   // next tile's loads are in flight during the MFMAs
        for (int rr = 0; rr < RPW; ++rr) {
            const int r = wr * RPW + rr;
#pragma unroll 2
            for (int s = 0; s < TW / 4; ++s) {
                const int gp = r * TW + 4 * s + q;
                const float av = g_lds[gp * CO_S + swz<CO_S>(gp, cot * 16 + p)];
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const int dy = tap / 3, dx = tap % 3;
                    const int xp = (r + dy) * HALO_W + 4 * s + q + dx;
                    const float bvv = x_lds[xp * CI_S + swz<CI_S>(xp, cit * 16 + p)];
                    acc[tap] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bvv, acc[tap], 0, 0, 0);
                }
            }
        }
        tile = tn;
    }

    // ---- sum the WR row-waves of each sub-slice through LDS (fixed order), then write this block's slab ----
    __syncthreads();
    float4* red = reinterpret_cast<float4*>(smem);
#pragma unroll
    for (int t = 0; t < 9; ++t)
        red[(wave * 9 + t) * 64 + lane] = make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
    __syncthreads();
    float* slab = a.partial + ((long)blockIdx.x * gridDim.y + slice) * (9 * CO_S * CI_S);
    for (int e = tid; e < WO * 9 * 64; e += 256) {
        const int l = e & 63, t = (e >> 6) % 9, o = (e >> 6) / 9;
        float4 v = red[(o * 9 + t) * 64 + l];               // wave index = wr*WO + wo
#pragma unroll
        for (int k = 1; k < WR; ++k) v = f4add(v, red[((k * WO + o) * 9 + t) * 64 + l]);
        const int ci_l = (o % CIT) * 16 + (l & 15), co_l = (o / CIT) * 16 + 4 * (l >> 4);
        float* op = slab + ((long)t * CO_S + co_l) * CI_S + ci_l;
        op[0] = v.x; op[CI_S] = v.y; op[2 * CI_S] = v.z; op[3 * CI_S] = v.w;
    }
}

// ---------------------------------------------------------------------------------------------------------
// fp32 weight gradient, second version (wgrad_kernel above: one ds_read_b32 + its address arithmetic per MFMA, 50-55 % of the
// fp32 MFMA peak).  The contraction index is the pixel, 4 per v_mfma_f32_16x16x4_f32, so an operand register must hold ONE channel
// of 4 pixels.  The tile is therefore staged channel-major -- gT[co][row][col], xT[ci][row][col + 1 halo] -- with scalar LDS
// stores (global loads stay 16 B per lane along the channels), and k-lane q of MFMA j takes pixel 16 blk + 4 q + j: the A operands
// of j = 0..3 are ONE ds_read_b128 of gT, the B operands of all three dx taps of a row are x columns 4q .. 4q + 5 of that row, i.e.
// one ds_read_b128 + one ds_read_b64, picked by register index j + dx.  36 MFMAs (a 16-pixel block, all 9 taps) need 7 LDS
// reads and no address arithmetic instead of 40 reads.  Plane pitches are = 4 (mod 64) dwords, which spreads the 16 channel
// lanes of a read over the banks (one 2-way slot per b128 group) and makes the scalar stores 2-way at worst (free, LDS section of
// MI355X_MICROARCH.md).  Tile order, per-block slabs and the fixed-order reduction are those of wgrad_kernel: bit-reproducible.
// NW waves: a 16 x 16 (cout, cin) sub-slice per wave group, the tile's rows split over the groups' waves.
// ---------------------------------------------------------------------------------------------------------
constexpr int pad_plane(int n) { return n + ((4 - n % 64) + 64) % 64; }      // smallest m >= n with m = 4 (mod 64)

// WINO = 1 (16 x 16 slices, 8 x 32 tiles): the contraction in Winograd form, dW = G^T [ sum over 2x2 output tiles of (A dY A^T) . (B^T d B) ] G
// -- the backward-filter counterpart of conv3x3_tile_kernel's F(2x2, 3x3).  A wave takes one row of 16 tiles; an MFMA contracts over
// 4 tiles: lane (p, q) holds, for tile 4 q + ks, the transformed 2x2 output-gradient patch of output channel p (A operand) and the
// transformed 4x4 input patch of input channel p (B operand), both computed by itself from its channel plane (2 + 8 ds_read_b64).
// 16 accumulators (one per position of the 4x4 transformed tile) instead of 9 taps; 64 instead of 144 MFMAs per wave and tile.
// The G^T . G back-transform is linear, so every workgroup applies it to its own partial sum before writing the slab: slab
// format, slab reduction and bit-reproducibility are those of the direct form.
template <int COT, int CIT, int RES, int TW, int NW, int XF, int WINO = 0>
__global__ __launch_bounds__(NW * 64, (WINO && !XF && COT * CIT == 1) ? 2 : (COT * CIT == 1) ? 3 : (COT * CIT == 2 ? 2 : 1)) void wgrad_f32_kernel(WgradArgs a) {
    static_assert(!WINO || TW == 32, "Winograd weight gradient: 8 x 32 tiles (a wave takes whole rows of 16 tiles)");
    constexpr int NT = NW * 64;
    constexpr int TH = 256 / TW, HALO_H = TH + 2, NBLK = TW / 16;
    constexpr int CO_S = COT * 16, CI_S = CIT * 16;
    constexpr int WO = COT * CIT, WR = NW / WO, RPW = TH / WR;                   // wave groups over sub-slices / over rows
    constexpr int XP = TW + 4;                                                   // x row pitch (TW + 2 used), a multiple of 4
    constexpr int PLANE_G = pad_plane(TH * TW), PLANE_X = pad_plane(HALO_H * XP);
    constexpr int G_ELEMS = CO_S * PLANE_G, X_ELEMS = CI_S * PLANE_X;
    constexpr int NACC = WINO ? 16 : 9;
    constexpr int RED_ELEMS = NW * (WINO ? 8 : 9) * 64 * 4;      // (Winograd: the 16 positions cross the waves in two halves of 8)
    // bilinear input: the low-resolution source patch of the halo tile is loaded once (fp32, [py][px][CI_S]) and expanded LDS -> LDS,
    // as in wgrad_bf16x3_kernel: 2 global loads per thread instead of 24, and the tap / weight arithmetic is tile-invariant
    constexpr int PH = TH / 2 + 2, PW = TW / 2 + 2, NPP = PH * PW;
    constexpr int PATCH_ELEMS = RES == NGAN_RESAMPLE_UP2 ? NPP * CI_S : 0;
    constexpr int SMEM = (G_ELEMS + X_ELEMS + PATCH_ELEMS) > RED_ELEMS ? (G_ELEMS + X_ELEMS + PATCH_ELEMS) : RED_ELEMS;
    constexpr int NG = TH * TW * (CO_S / 4) / NT, NXI = HALO_H * (TW + 2) * (CI_S / 4), NX = (NXI + NT - 1) / NT;
    static_assert(TH * TW * (CO_S / 4) % NT == 0 && TH % WR == 0 && NW % WO == 0, "tile split");
    // XF: plain input on an image whose width is a multiple of the 32-pixel tile -- the x tile is staged like conv3x3_tile_kernel's
    // (interior columns by whole loads at constant per-lane offsets, the descriptor base moved per tile, the top halo row behind a
    // zero-record descriptor, the two halo columns in one extra load): ~50 fewer VALU instructions per wave and tile
    constexpr int Q = CI_S / 4, NXINT = HALO_H * 32 * Q / NT, NXF = NXINT + 1, N_HALO = 2 * HALO_H * Q;
    constexpr int NPI = NPP * Q, NXP = (NPI + NT - 1) / NT;                      // patch float4s, per thread
    static_assert(!XF || (RES == NGAN_RESAMPLE_NONE && TW == 32 && (HALO_H * 32 * Q) % NT == 0 && N_HALO <= NT), "fast x staging");
    __shared__ __attribute__((aligned(16))) float smem[SMEM];
    float* g_lds = smem;
    float* x_lds = smem + G_ELEMS;
    float* patch = smem + G_ELEMS + X_ELEMS;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int p = lane & 15, q = lane >> 4;
    const int wo = wave % WO, wr = wave / WO;
    const int cot = wo / CIT, cit = wo % CIT;        // this wave's 16x16 (cout, cin) sub-slice, all 9 taps
    const int slice = blockIdx.y;
    const int co0 = (slice / a.n_ci_slices) * CO_S, ci0 = (slice % a.n_ci_slices) * CI_S;

    // ---- tile-invariant staging constants: global byte offset inside the image relative to the tile origin, LDS float index ----
    int g_off[NG], g_l[NG];
#pragma unroll
    for (int i = 0; i < NG; ++i) {
        const int e = tid + i * NT;
        const int pix = e / (CO_S / 4), c4 = e % (CO_S / 4), r = pix / TW, c = pix % TW;
        g_off[i] = ((r * a.W + c) * a.N + co0 + c4 * 4) * 4;
        g_l[i] = (c4 * 4) * PLANE_G + r * TW + c;
    }
    int x_r[NX], x_c[NX], x_l[NX];                   // halo pixel (row, col) relative to the tile origin, LDS float index
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        const int e = tid + i * NT;
        const int pix = e / (CI_S / 4), c4 = e % (CI_S / 4), r = pix / (TW + 2), c = pix % (TW + 2);
        x_r[i] = e < NXI ? r - 1 : -1000; x_c[i] = c - 1;
        x_l[i] = (c4 * 4) * PLANE_X + r * XP + c;
    }
    const int x_ch = ci0 + (tid % (CI_S / 4)) * 4;   // (NT is a multiple of CI_S / 4: the channel quad of a thread is the same in every slot)
    unsigned xf_voff[XF ? NXF : 1];
    int xf_l[XF ? NXF : 1], xf_bits = 8;
    if (XF) {
#pragma unroll
        for (int i = 0; i < NXINT; ++i) {
            const int e = tid + i * NT, c4 = e % Q, pix = e / Q, r = pix >> 5, c = (pix & 31) + 1;
            xf_voff[i] = (unsigned)(((r * a.W + c) * a.K + ci0 + c4 * 4) * 4);
            xf_l[i] = (c4 * 4) * PLANE_X + r * XP + c;
        }
        const int c4 = tid % Q, r = (tid / Q) % HALO_H, side = tid / (Q * HALO_H);
        const bool used = tid < N_HALO;
        const int c = side ? 33 : 0;
        xf_voff[NXINT] = used ? (unsigned)(((r * a.W + c) * a.K + ci0 + c4 * 4) * 4) : 0xFFFFFFF0u;
        xf_l[NXINT] = (c4 * 4) * PLANE_X + (used ? r : 0) * XP + c;
        xf_bits = used ? ((side ? 2 : 1) | (r == 0 ? 4 : 0)) : 8;
    }

    f32x4 acc[NACC];
#pragma unroll
    for (int t = 0; t < NACC; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    float4 gst[NG], xst[XF ? NXF : (RES == NGAN_RESAMPLE_UP2 ? NXP : NX)];
    constexpr unsigned OOB = 0xFFFFFFF0u;
    auto issue = [&](int tile) {
        int t = tile;
        const int txi = t % a.tiles_x; t /= a.tiles_x;
        const int tyi = t % a.tiles_y;
        const int b = t / a.tiles_y;
        const int y0 = tyi * TH, x0 = txi * TW;
        // g: the tile origin moves the descriptor's base; rows below the image fall outside its records (zeros), columns right
        // of the image are masked per lane (only when W is not a multiple of the tile width)
        {
            const int soff = (y0 * a.W + x0) * a.N * 4;
            const char* base = reinterpret_cast<const char*>(a.g + (long)b * a.H * a.W * a.N) + soff;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, (unsigned)(a.H * a.W * a.N) * 4u - (unsigned)soff, 0x00020000);
            const bool ragged = x0 + TW > a.W;
#pragma unroll
            for (int i = 0; i < NG; ++i) {
                unsigned off = (unsigned)g_off[i];
                if (ragged) { const int e = tid + i * NT; if (x0 + (e / (CO_S / 4)) % TW >= a.W) off = OOB; }
                gst[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
            }
        }
        if (XF) {
            const int soff = ((y0 - 1) * a.W + (x0 - 1)) * a.K * 4;                  // negative on the top row / for the first tile
            const char* base = reinterpret_cast<const char*>(a.x + (long)b * a.H * a.W * a.K) + soff;
            const unsigned nrec = (unsigned)(a.H * a.W * a.K) * 4u - (unsigned)soff;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, nrec, 0x00020000);
            // top halo row = the first interior load of the waves holding items e < 32 Q; above the image: no records, zeros
            const __amdgpu_buffer_rsrc_t rs_top = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, (y0 == 0 && wave < (32 * Q) / 64) ? 0u : nrec, 0x00020000);
#pragma unroll
            for (int i = 0; i < NXINT; ++i)
                xst[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(i == 0 ? rs_top : rs, xf_voff[i], 0, 0));
            const int bad = (x0 == 0 ? 1 : 0) | (x0 + 32 >= a.W ? 2 : 0) | (y0 == 0 ? 4 : 0) | 8;
            xst[XF ? NXINT : 0] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, (xf_bits & bad) ? OOB : xf_voff[XF ? NXINT : 0], 0, 0));
        } else if (RES == NGAN_RESAMPLE_NONE) {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x + (long)b * a.H * a.W * a.K), 0,
                                                                                 (unsigned)(a.H * a.W * a.K) * 4u, 0x00020000);
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                const int gy = y0 + x_r[i], gx = x0 + x_c[i];        // x_r = -1000 marks an unused slot: fails the range test
                const bool ok = (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
                const unsigned off = ok ? (unsigned)(((gy * a.W + gx) * a.K + x_ch) * 4) : OOB;
                xst[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
            }
        } else if (RES == NGAN_RESAMPLE_UP2) {
            const int h = a.H >> 1, w = a.W >> 1;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x + (long)b * h * w * a.K), 0,
                                                                                 (unsigned)(h * w * a.K) * 4u, 0x00020000);
            const int ly0 = (y0 >> 1) - 1, lx0 = (x0 >> 1) - 1;      // patch row 0 / column 0; clamped coordinates = the taps' edge rule
#pragma unroll
            for (int i = 0; i < NXP; ++i) {
                const int e = tid + i * NT, pix = e / Q, c4 = e % Q;
                const int ly = min(max(ly0 + pix / PW, 0), h - 1), lx = min(max(lx0 + pix % PW, 0), w - 1);
                const unsigned off = e < NPI ? (unsigned)(((ly * w + lx) * a.K + ci0 + c4 * 4) * 4) : OOB;
                xst[RES == NGAN_RESAMPLE_UP2 ? i : 0] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
            }
        } else {
#pragma unroll
            for (int i = 0; i < NX; ++i)
                xst[i] = x_r[i] > -1000 ? load_resampled<RES>(a.x, b, y0 + x_r[i], x0 + x_c[i], x_ch, a.H, a.W, a.K) : f4zero();
        }
    };

    // this lane's operand addresses inside a (row, 16-pixel block): channel plane p of its sub-slice, pixels 4q ..
    const float* ga = g_lds + (cot * 16 + p) * PLANE_G + 4 * q;
    const float* xa = x_lds + (cit * 16 + p) * PLANE_X + 4 * q;

    int tile = blockIdx.x;
    if (tile < a.n_tiles) issue(tile);
    while (tile < a.n_tiles) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NG; ++i) {
            float* d = g_lds + g_l[i];
            d[0] = gst[i].x; d[PLANE_G] = gst[i].y; d[2 * PLANE_G] = gst[i].z; d[3 * PLANE_G] = gst[i].w;
        }
        if (XF) {
#pragma unroll
            for (int i = 0; i < NXF; ++i)
                if (i < NXINT || tid < N_HALO) {
                    float* d = x_lds + xf_l[XF ? i : 0];
                    const float4 v = xst[XF ? i : 0];
                    d[0] = v.x; d[PLANE_X] = v.y; d[2 * PLANE_X] = v.z; d[3 * PLANE_X] = v.w;
                }
        } else if (RES == NGAN_RESAMPLE_UP2) {
#pragma unroll
            for (int i = 0; i < NXP; ++i)
                if (tid + i * NT < NPI) st4(patch + (tid + i * NT) * 4, xst[RES == NGAN_RESAMPLE_UP2 ? i : 0]);
            __syncthreads();
            int tt = tile;
            const int txi = tt % a.tiles_x; tt /= a.tiles_x;
            const int y0 = (tt % a.tiles_y) * TH, x0 = txi * TW;
#pragma unroll
            for (int i = 0; i < NX; ++i)
                if (x_r[i] > -1000) {
                    // output row Y odd -> taps (i, i + 1) with weights (.75, .25); even -> (i - 1, i) with (.25, .75); the tile origin is
                    // even, so parity, weights and the patch cell ((r + 1) >> 1, (c + 1) >> 1) depend on the slot only
                    const float wy0 = (x_r[i] & 1) ? 0.75f : 0.25f, wx0 = (x_c[i] & 1) ? 0.75f : 0.25f;
                    const float* r0 = patch + (((x_r[i] + 1) >> 1) * PW + ((x_c[i] + 1) >> 1)) * CI_S + (tid % Q) * 4;
                    // (a packed-math (v_pk_fma) version of this blend needs aligned register pairs for 6 - 11 slots' weights, which the
                    // compiler keeps live across the tile loop: 53 spilled registers, 230 -> 393 us.  Scalar fp32 it is.)
                    const float4 top = f4fma(ld4(r0 + CI_S), 1.0f - wx0, f4scale(ld4(r0), wx0));
                    const float4 bot = f4fma(ld4(r0 + PW * CI_S + CI_S), 1.0f - wx0, f4scale(ld4(r0 + PW * CI_S), wx0));
                    float4 v = f4fma(bot, 1.0f - wy0, f4scale(top, wy0));
                    const bool ok = (unsigned)(y0 + x_r[i]) < (unsigned)a.H && (unsigned)(x0 + x_c[i]) < (unsigned)a.W;   // conv padding
                    v = make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
                    float* d = x_lds + x_l[i];
                    d[0] = v.x; d[PLANE_X] = v.y; d[2 * PLANE_X] = v.z; d[3 * PLANE_X] = v.w;
                }
        } else {
#pragma unroll
            for (int i = 0; i < NX; ++i)
                if (x_r[i] > -1000) {
                    float* d = x_lds + x_l[i];
                    d[0] = xst[i].x; d[PLANE_X] = xst[i].y; d[2 * PLANE_X] = xst[i].z; d[3 * PLANE_X] = xst[i].w;
                }
        }
        __syncthreads();
        const int tn = tile + gridDim.x;
        if (tn < a.n_tiles) issue(tn);   // next tile's loads are in flight during the MFMAs
        if (WINO) {
            // this wave's tile row: output rows 2 wr, 2 wr + 1 = halo rows 2 wr .. 2 wr + 3.  Signs: A = [1 0; 1 1; 1 -1; 0 -1] is used
            // without the minus signs of its last row (one negation per element saved); the back-transform flips the sign of every
            // position with u = 3 xor v = 3 instead.
            // lane (p, q) owns the four consecutive tiles 4 q .. 4 q + 3 of the row (K-step ks contracts tiles 4 q + ks over q): two tiles
            // at a time are one 16-byte + one 8-byte read per input row and one 16-byte read per gradient row
#pragma unroll
            for (int tr = 0; tr < RPW / 2; ++tr) {                          // this wave's rows of tiles (RPW output rows)
            const float* gp = ga - 4 * q + (wr * RPW + 2 * tr) * TW + 8 * q;          // ga = plane p + 4 q: back to the plane, then column 8 q
            const float* xp = xa - 4 * q + (wr * RPW + 2 * tr) * XP + 8 * q;
#pragma unroll
            for (int kp = 0; kp < 2; ++kp) {
                float xr[4][6], gr[2][4];
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const float4 lo = ld4(xp + r4 * XP + 4 * kp);
                    const float2 hi = *reinterpret_cast<const float2*>(xp + r4 * XP + 4 * kp + 4);
                    xr[r4][0] = lo.x; xr[r4][1] = lo.y; xr[r4][2] = lo.z; xr[r4][3] = lo.w; xr[r4][4] = hi.x; xr[r4][5] = hi.y;
                }
#pragma unroll
                for (int r2 = 0; r2 < 2; ++r2) {
                    const float4 v = ld4(gp + r2 * TW + 4 * kp);
                    gr[r2][0] = v.x; gr[r2][1] = v.y; gr[r2][2] = v.z; gr[r2][3] = v.w;
                }
#pragma unroll
                for (int t2 = 0; t2 < 2; ++t2) {
                    float t[4][4], V[4][4], sg[4][2], M[4][4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float d0 = xr[0][2 * t2 + c], d1 = xr[1][2 * t2 + c], d2 = xr[2][2 * t2 + c], d3 = xr[3][2 * t2 + c];
                        t[0][c] = d0 - d2; t[1][c] = d1 + d2; t[2][c] = d2 - d1; t[3][c] = d1 - d3;
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) { V[u][0] = t[u][0] - t[u][2]; V[u][1] = t[u][1] + t[u][2]; V[u][2] = t[u][2] - t[u][1]; V[u][3] = t[u][1] - t[u][3]; }
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        const float g0 = gr[0][2 * t2 + c], g1 = gr[1][2 * t2 + c];
                        sg[0][c] = g0; sg[1][c] = g0 + g1; sg[2][c] = g0 - g1; sg[3][c] = g1;
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) { M[u][0] = sg[u][0]; M[u][1] = sg[u][0] + sg[u][1]; M[u][2] = sg[u][0] - sg[u][1]; M[u][3] = sg[u][1]; }
#pragma unroll
                    for (int u = 0; u < 4; ++u)
#pragma unroll
                        for (int v = 0; v < 4; ++v)
                            acc[u * 4 + v] = __builtin_amdgcn_mfma_f32_16x16x4f32(M[u][v], V[u][v], acc[u * 4 + v], 0, 0, 0);
                }
            }
            }
        } else
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) {
            const int r = wr * RPW + rr;
#pragma unroll
            for (int blk = 0; blk < NBLK; ++blk) {
                const float4 av4 = ld4(ga + r * TW + blk * 16);
                const float av[4] = {av4.x, av4.y, av4.z, av4.w};
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const float* xr = xa + (r + dy) * XP + blk * 16;
                    const float4 b0 = ld4(xr);
                    const float2 b1 = *reinterpret_cast<const float2*>(xr + 4);
                    const float bv[6] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y};
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            acc[dy * 3 + dx] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], bv[j + dx], acc[dy * 3 + dx], 0, 0, 0);
                }
            }
        }
        tile = tn;
    }

    // ---- sum the WR row-waves of each sub-slice through LDS (fixed order), then write this block's slab ----
    __syncthreads();
    float4* red = reinterpret_cast<float4*>(smem);
    float* slab = a.partial + ((long)blockIdx.x * gridDim.y + slice) * (9 * CO_S * CI_S);
    if (!WINO) {
#pragma unroll
        for (int t = 0; t < NACC; ++t)
            red[(wave * NACC + t) * 64 + lane] = make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
        __syncthreads();
    }
    if (WINO) {
        // item (l, u, o): row u of the summed 4x4 position tile of lane l of sub-slice o (its WR waves in fixed order), multiplied by G
        // from the right:  Z[u][j] = sum_v s_v dU[u][v] G[v][j],  G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1],  s = (1, 1, 1, -1) (MFMA section)
        constexpr int ITEMS = WO * 4 * 64, NIT = (ITEMS + NT - 1) / NT;
        float4 du[NIT][4];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            if (half) __syncthreads();
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const f32x4 v = acc[half * 8 + t];
                red[(wave * 8 + t) * 64 + lane] = make_float4(v[0], v[1], v[2], v[3]);
            }
            __syncthreads();
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int item = tid + it * NT, l = item & 63, u = (item >> 6) & 3, o = item >> 8;
                if (item < ITEMS && (u >> 1) == half) {
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        float4 sum = red[((0 * WO + o) * 8 + (u & 1) * 4 + v) * 64 + l];        // wave index = wr * WO + wo
#pragma unroll
                        for (int w = 1; w < WR; ++w) sum = f4add(sum, red[((w * WO + o) * 8 + (u & 1) * 4 + v) * 64 + l]);
                        du[it][v] = sum;
                    }
                }
            }
        }
        __syncthreads();                                     // every thread has read its part of `red`
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int item = tid + it * NT, l = item & 63, u = (item >> 6) & 3, o = item >> 8;
            if (item < ITEMS) {
                const float su = u == 3 ? -1.f : 1.f;
                const float4 h12p = f4scale(f4add(du[it][1], du[it][2]), 0.5f), h12m = f4scale(f4add(du[it][1], f4scale(du[it][2], -1.f)), 0.5f);
                red[((o * 4 + u) * 3 + 0) * 64 + l] = f4scale(f4add(du[it][0], h12p), su);
                red[((o * 4 + u) * 3 + 1) * 64 + l] = f4scale(h12m, su);
                red[((o * 4 + u) * 3 + 2) * 64 + l] = f4scale(f4add(h12p, f4scale(du[it][3], -1.f)), su);
            }
        }
        __syncthreads();
        // item (l, i < 3, o): dW[i][j] = sum_u G^T[i][u] Z[u][j]
        for (int item = tid; item < WO * 3 * 64; item += NT) {
            const int l = item & 63, i = (item >> 6) % 3, o = (item >> 6) / 3;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const float4 z_0 = red[((o * 4 + 0) * 3 + j) * 64 + l], z_1 = red[((o * 4 + 1) * 3 + j) * 64 + l];
                const float4 z_2 = red[((o * 4 + 2) * 3 + j) * 64 + l], z_3 = red[((o * 4 + 3) * 3 + j) * 64 + l];
                const float4 p12 = f4scale(f4add(z_1, z_2), 0.5f), m12 = f4scale(f4add(z_1, f4scale(z_2, -1.f)), 0.5f);
                const float4 v = i == 0 ? f4add(z_0, p12) : i == 1 ? m12 : f4add(p12, z_3);
                const int ci_l = (o % CIT) * 16 + (l & 15), co_l = (o / CIT) * 16 + 4 * (l >> 4);
                float* op = slab + ((long)(i * 3 + j) * CO_S + co_l) * CI_S + ci_l;
                op[0] = v.x; op[CI_S] = v.y; op[2 * CI_S] = v.z; op[3 * CI_S] = v.w;
            }
        }
        return;
    }
    for (int e = tid; e < WO * 9 * 64; e += NT) {
        const int l = e & 63, t = (e >> 6) % 9, o = (e >> 6) / 9;
        float4 v = red[(o * 9 + t) * 64 + l];               // wave index = wr*WO + wo
#pragma unroll
        for (int k = 1; k < WR; ++k) v = f4add(v, red[((k * WO + o) * 9 + t) * 64 + l]);
        const int ci_l = (o % CIT) * 16 + (l & 15), co_l = (o / CIT) * 16 + 4 * (l >> 4);
        float* op = slab + ((long)t * CO_S + co_l) * CI_S + ci_l;
        op[0] = v.x; op[CI_S] = v.y; op[2 * CI_S] = v.z; op[3 * CI_S] = v.w;
    }
}

// ---------------------------------------------------------------------------------------------------------
// Split-bf16 weight gradient.  The contraction index is the PIXEL, so an MFMA operand (v_mfma_f32_16x16x32_bf16: 8
// consecutive k per lane) needs 8 pixels of ONE channel per lane -- the transpose of the channels-last image.  The tile
// is therefore staged pixel-major as bf16 hi/lo planes of 16 channels ([part][plane][pixel][16], 32-byte rows) and the
// operands are fetched with ds_read_b64_tr_b16, which hands lane i of a 16-lane group column (channel) i of 4 rows
// (pixels).  One k-step = one tile row of 32 pixels; lane group kq takes pixels 4kq..4kq+3 and 16+4kq..16+4kq+3 (the
// same permutation on both operands), so each 32-lane half of a read touches 256 contiguous bytes: conflict-free,
// and the dx tap shift is just a different row address (no alignment constraint).
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ bf16x8 tr_frag(const __bf16* base_lo16, const __bf16* base_hi16) {
    const bf16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)base_lo16);
    const bf16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)base_hi16);
    bf16x8 r;
    r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; r[3] = a[3]; r[4] = b[0]; r[5] = b[1]; r[6] = b[2]; r[7] = b[3];
    return r;
}

template <int COT, int CIT, int RES, int TW>
__global__ __launch_bounds__(256, 2) void wgrad_bf16x3_kernel(WgradArgs a) {
    // tile = 256 pixels: 8 x 32, or 16 x 16 for images at most 16 wide.  One k-step = 32 pixels = one tile row (TW = 32) or two
    // consecutive rows (TW = 16): the second 16-pixel half of a fragment then sits one halo row further instead of 16 pixels.
    constexpr int TH = 256 / TW, HALO_H = TH + 2, HALO_W = TW + 2, G_PIX = TH * TW, X_PIX = HALO_H * HALO_W;
    constexpr int CO_S = COT * 16, CI_S = CIT * 16;
    constexpr int WO = COT * CIT, WR = 4 / WO, NKS = 8, KPW = NKS / WR;          // k-steps per tile / per wave
    constexpr int X_HALF2 = (TW == 32 ? 16 : HALO_W) * 16;                       // bf16 offset of a fragment's second half in x
    constexpr int X_ROWS_PER_KS = TW == 32 ? 1 : 2;
    constexpr int G_E = 2 * COT * G_PIX * 16, X_E = 2 * CIT * X_PIX * 16;        // bf16 elements
    // bilinear input: the low-resolution source patch is staged once (fp32) and expanded LDS -> LDS, as in conv3x3_persist_kernel
    constexpr int PH = TH / 2 + 2, PW = TW / 2 + 2, NPP = PH * PW;
    constexpr int PATCH_BYTES = RES == NGAN_RESAMPLE_UP2 ? NPP * CI_S * 4 : 0;
    constexpr int RED_BYTES = 4 * 9 * 64 * 16;
    constexpr int IMG_BYTES = (G_E + X_E) * 2 + PATCH_BYTES;
    constexpr int SMEM_BYTES = IMG_BYTES > RED_BYTES ? IMG_BYTES : RED_BYTES;
    constexpr int NG = G_PIX * (CO_S / 4) / 256, NXI = X_PIX * (CI_S / 4), NX = (NXI + 255) / 256;
    constexpr int NPI = NPP * (CI_S / 4), NXL = RES == NGAN_RESAMPLE_UP2 ? (NPI + 255) / 256 : NX;   // global loads per thread for x
    __shared__ __attribute__((aligned(16))) unsigned char smem_raw[SMEM_BYTES];
    __bf16* g_img = reinterpret_cast<__bf16*>(smem_raw);
    __bf16* x_img = g_img + G_E;
    float* patch = reinterpret_cast<float*>(smem_raw + (G_E + X_E) * 2);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kq = lane >> 4, i16 = lane & 15, qq = i16 >> 2, pp = i16 & 3;
    const int wo = wave % WO, wr = wave / WO;
    const int cot = wo / CIT, cit = wo % CIT;
    const int slice = blockIdx.y;
    const int co0 = (slice / a.n_ci_slices) * CO_S, ci0 = (slice % a.n_ci_slices) * CI_S;
    const int h = a.H >> 1, w = a.W >> 1;

    int g_r[NG], g_c[NG], g_ch[NG], g_l[NG];
#pragma unroll
    for (int i = 0; i < NG; ++i) {
        const int e = tid + i * 256;
        const int pix = e / (CO_S / 4), c4 = e % (CO_S / 4);
        g_r[i] = pix / TW; g_c[i] = pix % TW; g_ch[i] = co0 + c4 * 4;
        g_l[i] = ((c4 >> 2) * G_PIX + pix) * 16 + (c4 & 3) * 4;                 // hi part; lo = + COT*G_PIX*16
    }
    // x staging descriptors: halo pixel (row, col) relative to the tile origin, channel, bf16 index of the hi part
    int x_r[NX], x_c[NX], x_ch[NX], x_l[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        const int e = tid + i * 256;
        const int pix = e / (CI_S / 4), c4 = e % (CI_S / 4);
        x_r[i] = e < NXI ? pix / HALO_W - 1 : -1000; x_c[i] = pix % HALO_W - 1; x_ch[i] = ci0 + c4 * 4;
        x_l[i] = ((c4 >> 2) * X_PIX + pix) * 16 + (c4 & 3) * 4;
    }
    // this lane's transposing-read offsets (bf16 elements) inside one plane: pixel 4kq + qq (second read: + 16), channels 4pp..
    const int tr0 = (4 * kq + qq) * 16 + 4 * pp;

    f32x4 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    float4 gst[NG], xst[NXL];
    auto issue = [&](int tile) {
        int t = tile;
        const int txi = t % a.tiles_x; t /= a.tiles_x;
        const int tyi = t % a.tiles_y;
        const int b = t / a.tiles_y;
        const int y0 = tyi * TH, x0 = txi * TW;
        // loads through per-image buffer descriptors: 32-bit offsets, and an out-of-range offset (tile edge, conv padding, unused
        // staging slot) reads zeros -- no branch around the load and no zero-filled registers (see conv3x3_persist_kernel)
        constexpr unsigned OOB = 0xFFFFFFF0u;
        const __amdgpu_buffer_rsrc_t g_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.g + (long)b * a.H * a.W * a.N), 0,
                                                                                 (unsigned)(a.H * a.W * a.N) * 4u, 0x00020000);
#pragma unroll
        for (int i = 0; i < NG; ++i) {
            const int gy = y0 + g_r[i], gx = x0 + g_c[i];
            const unsigned off = (gy < a.H && gx < a.W) ? (unsigned)(((gy * a.W + gx) * a.N + g_ch[i]) * 4) : OOB;
            gst[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(g_rsrc, off, 0, 0));
        }
        if (RES == NGAN_RESAMPLE_UP2) {
            const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x + (long)b * h * w * a.K), 0,
                                                                                     (unsigned)(h * w * a.K) * 4u, 0x00020000);
            const int ly0 = (y0 >> 1) - 1, lx0 = (x0 >> 1) - 1;
#pragma unroll
            for (int i = 0; i < NXL; ++i) {
                const int e = tid + i * 256;
                const int pix = e / (CI_S / 4), c4 = e % (CI_S / 4);
                const int ly = min(max(ly0 + pix / PW, 0), h - 1), lx = min(max(lx0 + pix % PW, 0), w - 1);
                const unsigned off = e < NPI ? (unsigned)(((ly * w + lx) * a.K + ci0 + c4 * 4) * 4) : OOB;
                xst[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, off, 0, 0));
            }
        } else if (RES == NGAN_RESAMPLE_NONE) {
            const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x + (long)b * a.H * a.W * a.K), 0,
                                                                                     (unsigned)(a.H * a.W * a.K) * 4u, 0x00020000);
#pragma unroll
            for (int i = 0; i < NXL; ++i) {
                const int gy = y0 + x_r[i], gx = x0 + x_c[i];        // x_r = -1000 marks an unused slot: fails the range test
                const bool ok = (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
                const unsigned off = ok ? (unsigned)(((gy * a.W + gx) * a.K + x_ch[i]) * 4) : OOB;
                xst[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, off, 0, 0));
            }
        } else {
#pragma unroll
            for (int i = 0; i < NXL; ++i)
                xst[i] = x_r[i] > -1000 ? load_resampled<RES>(a.x, b, y0 + x_r[i], x0 + x_c[i], x_ch[i], a.H, a.W, a.K) : f4zero();
        }
    };
    auto split_store = [&](__bf16* img, int idx, int lo_off, float4 v) {
        bf16x4 hi, lo;
        hi[0] = (__bf16)v.x; hi[1] = (__bf16)v.y; hi[2] = (__bf16)v.z; hi[3] = (__bf16)v.w;
        lo[0] = (__bf16)(v.x - (float)hi[0]); lo[1] = (__bf16)(v.y - (float)hi[1]);
        lo[2] = (__bf16)(v.z - (float)hi[2]); lo[3] = (__bf16)(v.w - (float)hi[3]);
        *reinterpret_cast<bf16x4*>(img + idx) = hi;
        *reinterpret_cast<bf16x4*>(img + idx + lo_off) = lo;
    };

    int tile = blockIdx.x;
    if (tile < a.n_tiles) issue(tile);
    while (tile < a.n_tiles) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NG; ++i) split_store(g_img, g_l[i], COT * G_PIX * 16, gst[i]);
        if (RES == NGAN_RESAMPLE_UP2) {
#pragma unroll
            for (int i = 0; i < NXL; ++i)
                if (tid + i * 256 < NPI) st4(patch + (tid + i * 256) * 4, xst[i]);       // patch is plain [py][px][CI_S]
            __syncthreads();
            int t = tile;
            const int txi = t % a.tiles_x; t /= a.tiles_x;
            const int tyi = t % a.tiles_y;
            const int y0 = tyi * TH, x0 = txi * TW;
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                if (x_r[i] <= -1000) continue;
                const int Y = y0 + x_r[i], X = x0 + x_c[i];
                float4 v = f4zero();
                if (Y >= 0 && Y < a.H && X >= 0 && X < a.W) {
                    // Y odd -> taps (i, i+1), weights (.75, .25); Y even -> taps (i-1, i), weights (.25, .75); patch row 0 is low-res
                    // row y0/2 - 1 (loaded with clamped coordinates, so the image border needs no special case)
                    const float wy0 = (Y & 1) ? 0.75f : 0.25f, wx0 = (X & 1) ? 0.75f : 0.25f;
                    const int ry = (x_r[i] + 1) >> 1, rx = (x_c[i] + 1) >> 1;
                    const float* r0 = patch + (ry * PW + rx) * CI_S + (x_ch[i] - ci0);
                    const float4 top = f4fma(ld4(r0 + CI_S), 1.0f - wx0, f4scale(ld4(r0), wx0));
                    const float4 bot = f4fma(ld4(r0 + PW * CI_S + CI_S), 1.0f - wx0, f4scale(ld4(r0 + PW * CI_S), wx0));
                    v = f4fma(bot, 1.0f - wy0, f4scale(top, wy0));
                }
                split_store(x_img, x_l[i], CIT * X_PIX * 16, v);
            }
        } else {
#pragma unroll
            for (int i = 0; i < NX; ++i)
                if (x_r[i] > -1000) split_store(x_img, x_l[i], CIT * X_PIX * 16, xst[i]);
        }
        __syncthreads();
        const int tn = tile + gridDim.x;
        if (tn < a.n_tiles) issue(tn);
        const __bf16* gh = g_img + cot * G_PIX * 16 + tr0;
        const __bf16* gl = gh + COT * G_PIX * 16;
        const __bf16* xh = x_img + cit * X_PIX * 16 + tr0;
        const __bf16* xl = xh + CIT * X_PIX * 16;
        for (int kk = 0; kk < KPW; ++kk) {
            const int ks = wr * KPW + kk;
            const bf16x8 ah = tr_frag(gh + ks * 32 * 16, gh + (ks * 32 + 16) * 16);
            const bf16x8 al = tr_frag(gl + ks * 32 * 16, gl + (ks * 32 + 16) * 16);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int dy = tap / 3, dx = tap % 3;
                const int xo = ((ks * X_ROWS_PER_KS + dy) * HALO_W + dx) * 16;
                const bf16x8 bh = tr_frag(xh + xo, xh + xo + X_HALF2);
                const bf16x8 bl = tr_frag(xl + xo, xl + xo + X_HALF2);
                acc[tap] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc[tap], 0, 0, 0);
                acc[tap] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc[tap], 0, 0, 0);
                acc[tap] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc[tap], 0, 0, 0);
            }
        }
        tile = tn;
    }

    __syncthreads();
    float4* red = reinterpret_cast<float4*>(smem_raw);
#pragma unroll
    for (int t = 0; t < 9; ++t)
        red[(wave * 9 + t) * 64 + lane] = make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
    __syncthreads();
    float* slab = a.partial + ((long)blockIdx.x * gridDim.y + slice) * (9 * CO_S * CI_S);
    for (int e = tid; e < WO * 9 * 64; e += 256) {
        const int l = e & 63, t = (e >> 6) % 9, o = (e >> 6) / 9;
        float4 v = red[(o * 9 + t) * 64 + l];
#pragma unroll
        for (int k = 1; k < WR; ++k) v = f4add(v, red[((k * WO + o) * 9 + t) * 64 + l]);
        const int ci_l = (o % CIT) * 16 + (l & 15), co_l = (o / CIT) * 16 + 4 * (l >> 4);
        float* op = slab + ((long)t * CO_S + co_l) * CI_S + ci_l;
        op[0] = v.x; op[CI_S] = v.y; op[2 * CI_S] = v.z; op[3 * CI_S] = v.w;
    }
}

// out[(co*K + ci)*9 + tap] = scale * sum_parts slab[part][slice][tap][co_l][ci_l]; 16 outputs x 16 part-lanes per block
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ gw,
                                                           int nparts, int nslices, int n_ci_slices, int CO_S, int CI_S,
                                                           int K, float scale, int accumulate) {
    __shared__ float red[256];
    const int slab = 9 * CO_S * CI_S;
    const long M = (long)nslices * slab;
    const int tid = threadIdx.x;
    const long i = (long)blockIdx.x * 16 + (tid & 15);
    float s = 0.f;
    if (i < M) {
        const float* src = partial + i;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int j = tid >> 4;
        for (; j + 48 < nparts; j += 64) {
            a0 += src[(long)j * M]; a1 += src[(long)(j + 16) * M];
            a2 += src[(long)(j + 32) * M]; a3 += src[(long)(j + 48) * M];
        }
        for (; j < nparts; j += 16) a0 += src[(long)j * M];
        s = (a0 + a1) + (a2 + a3);
    }
    red[tid] = s;
    __syncthreads();
    if (tid < 16 && i < M) {
        s = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) s += red[tid + 16 * j];
        int r = (int)(i % slab);
        const int slice = (int)(i / slab);
        const int ci_l = r % CI_S; r /= CI_S;
        const int co_l = r % CO_S;
        const int tap = r / CO_S;
        const int co = (slice / n_ci_slices) * CO_S + co_l, ci = (slice % n_ci_slices) * CI_S + ci_l;
        float* o = gw + ((long)co * K + ci) * 9 + tap;
        *o = accumulate ? fmaf(s, scale, *o) : s * scale;
    }
}

struct WgradPlan { int co_s, ci_s, nslices, n_ci_slices, tiles_x, tiles_y, n_tiles, nwx, tw; };
constexpr int wgrad_f32_waves(int cot, int cit) { return cot * cit == 4 ? 8 : 4; }    // a 32 x 32 slice: two waves per 16 x 16 sub-slice

WgradPlan plan_wgrad(int B, int H, int W, int Cin, int Cout, int precision = 1) {
    WgradPlan p;
    p.co_s = (Cout % 32 == 0) ? 32 : 16;
    p.ci_s = (Cin % 32 == 0) ? 32 : 16;
    p.n_ci_slices = Cin / p.ci_s;
    p.nslices = (Cout / p.co_s) * p.n_ci_slices;
    p.tw = W <= 16 ? 16 : 32;   // 16x16-pixel tiles for narrow images, 8x32 otherwise
    p.tiles_x = ngan::ceil_div(W, p.tw);
    p.tiles_y = ngan::ceil_div(H, 256 / p.tw);
    p.n_tiles = B * p.tiles_x * p.tiles_y;
    static const int forced = [] { const char* e = getenv("NGAN_WGRAD_SLABS"); const int n = e ? atoi(e) : 0; return n > 0 ? n : 0; }();   // (A/B switch)
    // about two resident workgroups per CU: few slabs to reduce afterwards (256 / 768 measured slower).  The fp32 kernel's 32 x 32
    // slices are 8-wave workgroups with 83 KB of LDS, one per CU: 256 of them (fp32 32 -> 32 at 128x128: 100 vs 106 us, 64 -> 64 at
    // 32x32: 33 vs 39 us)
    const int total = forced ? forced : ((precision == 0 && p.co_s == 32 && p.ci_s == 32) ? 256 : 512);
    int cap = total / p.nslices;
    if (cap < 1) cap = 1;
    p.nwx = p.n_tiles < cap ? p.n_tiles : cap;
    return p;
}

template <int COT, int CIT>
int launch_wgrad(const WgradArgs& a, const WgradPlan& p, int res, int precision, hipStream_t s) {
    dim3 grid(p.nwx, p.nslices);
    if (precision == 1 && p.tw == 32) {
        if (res == 0) hipLaunchKernelGGL((wgrad_bf16x3_kernel<COT, CIT, 0, 32>), grid, dim3(256), 0, s, a);
        else if (res == 1) hipLaunchKernelGGL((wgrad_bf16x3_kernel<COT, CIT, 1, 32>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((wgrad_bf16x3_kernel<COT, CIT, 2, 32>), grid, dim3(256), 0, s, a);
        return ngan::launch_status("ngan_conv3x3_wgrad(bf16x3)");
    }
    if (precision == 1) {
        if (res == 0) hipLaunchKernelGGL((wgrad_bf16x3_kernel<COT, CIT, 0, 16>), grid, dim3(256), 0, s, a);
        else if (res == 1) hipLaunchKernelGGL((wgrad_bf16x3_kernel<COT, CIT, 1, 16>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((wgrad_bf16x3_kernel<COT, CIT, 2, 16>), grid, dim3(256), 0, s, a);
        return ngan::launch_status("ngan_conv3x3_wgrad(bf16x3, 16x16 tiles)");
    }
    static const bool v1 = getenv("NGAN_WGRAD_V1") && getenv("NGAN_WGRAD_V1")[0] == '1';     // A/B switch: the first fp32 kernel
    if (!v1) {
        constexpr int NW = wgrad_f32_waves(COT, CIT);
        static const bool wino = !(getenv("NGAN_WINOGRAD_WGRAD") && getenv("NGAN_WINOGRAD_WGRAD")[0] == '0');    // A/B switch
        if (wino && p.tw == 32) {
            if (res == 0 && a.W % 32 == 0) hipLaunchKernelGGL((wgrad_f32_kernel<COT, CIT, 0, 32, NW, 1, 1>), grid, dim3(NW * 64), 0, s, a);
            else if (res == 0) hipLaunchKernelGGL((wgrad_f32_kernel<COT, CIT, 0, 32, NW, 0, 1>), grid, dim3(NW * 64), 0, s, a);
            else if (res == 1) hipLaunchKernelGGL((wgrad_f32_kernel<COT, CIT, 1, 32, NW, 0, 1>), grid, dim3(NW * 64), 0, s, a);
            else hipLaunchKernelGGL((wgrad_f32_kernel<COT, CIT, 2, 32, NW, 0, 1>), grid, dim3(NW * 64), 0, s, a);
            return ngan::launch_status("ngan_conv3x3_wgrad(f32, winograd)");
        }
        if (p.tw == 32) {
            if (res == 0 && a.W % 32 == 0) hipLaunchKernelGGL((wgrad_f32_kernel<COT, CIT, 0, 32, NW, 1>), grid, dim3(NW * 64), 0, s, a);
            else if (res == 0) hipLaunchKernelGGL((wgrad_f32_kernel<COT, CIT, 0, 32, NW, 0>), grid, dim3(NW * 64), 0, s, a);
            else if (res == 1) hipLaunchKernelGGL((wgrad_f32_kernel<COT, CIT, 1, 32, NW, 0>), grid, dim3(NW * 64), 0, s, a);
            else hipLaunchKernelGGL((wgrad_f32_kernel<COT, CIT, 2, 32, NW, 0>), grid, dim3(NW * 64), 0, s, a);
        } else {
            if (res == 0) hipLaunchKernelGGL((wgrad_f32_kernel<COT, CIT, 0, 16, NW, 0>), grid, dim3(NW * 64), 0, s, a);
            else if (res == 1) hipLaunchKernelGGL((wgrad_f32_kernel<COT, CIT, 1, 16, NW, 0>), grid, dim3(NW * 64), 0, s, a);
            else hipLaunchKernelGGL((wgrad_f32_kernel<COT, CIT, 2, 16, NW, 0>), grid, dim3(NW * 64), 0, s, a);
        }
        return ngan::launch_status("ngan_conv3x3_wgrad(f32)");
    }
    if (p.tw == 32) {
        if (res == 0) hipLaunchKernelGGL((wgrad_kernel<COT, CIT, 0, 32>), grid, dim3(256), 0, s, a);
        else if (res == 1) hipLaunchKernelGGL((wgrad_kernel<COT, CIT, 1, 32>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((wgrad_kernel<COT, CIT, 2, 32>), grid, dim3(256), 0, s, a);
    } else {
        if (res == 0) hipLaunchKernelGGL((wgrad_kernel<COT, CIT, 0, 16>), grid, dim3(256), 0, s, a);
        else if (res == 1) hipLaunchKernelGGL((wgrad_kernel<COT, CIT, 1, 16>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((wgrad_kernel<COT, CIT, 2, 16>), grid, dim3(256), 0, s, a);
    }
    return ngan::launch_status("ngan_conv3x3_wgrad");
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------
// One launch that (re)packs many weights: entry e = {src, dst, Cout, Cin, mode, precision, scale, first output index}.
// Used after every optimiser step instead of one small launch per (weight, mode, precision).
// ---------------------------------------------------------------------------------------------------------
struct PackEntry { const float* src; float* dst; int cout, cin, mode, precision; float scale; int pad; long first; };

__global__ __launch_bounds__(256) void pack_many_kernel(const PackEntry* __restrict__ table, int n_entries, long total) {
    const long gidx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gidx >= total) return;
    int lo = 0, hi = n_entries - 1;                       // last entry with first <= gidx
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (table[mid].first <= gidx) lo = mid; else hi = mid - 1;
    }
    const PackEntry e = table[lo];
    const long idx = gidx - e.first;
    const int Cout = e.cout, Cin = e.cin, mode = e.mode;
    if (e.precision == 3) { pack_up2f_element(e.src, e.dst, Cout, Cin, e.scale, idx); return; }
    if (e.precision == 4) { e.dst[idx] = wino_weight(e.src, Cout, Cin, mode, e.scale, idx); return; }
    const int Kreal = mode == 0 ? Cin : Cout, N = mode == 0 ? Cout : Cin;
    const int K = e.precision == 2 ? 32 : Kreal;
    if (e.precision == 0) {
        const int G = K / 16, MT = N / 16;
        const int i = idx & 3, lane = (idx >> 2) & 63;
        long r = idx >> 8;
        const int mt = r % MT; r /= MT;
        const int g = r % G;
        const int tap = r / G;
        const int n = mt * 16 + (lane & 15), k = g * 16 + 4 * (lane >> 4) + i;
        const float v = mode == 0 ? e.src[((long)n * Cin + k) * 9 + tap] : e.src[((long)k * Cin + n) * 9 + (8 - tap)];
        e.dst[idx] = v * e.scale;
    } else {
        const int MT = N / 16;
        const int j = idx & 7, lane = (idx >> 3) & 63, part = (idx >> 9) & 1;
        long r = idx >> 10;
        const int mt = r % MT;
        const int step = r / MT;
        const int n = mt * 16 + (lane & 15), kk = 8 * (lane >> 4) + j;
        const int tap = K == 16 ? 2 * step + (kk >> 4) : step % 9;
        const int k = K == 16 ? (kk & 15) : (step / 9) * 32 + kk;
        float v = 0.f;
        if (tap < 9 && k < Kreal) v = mode == 0 ? e.src[((long)n * Cin + k) * 9 + tap] : e.src[((long)k * Cin + n) * 9 + (8 - tap)];
        v *= e.scale;
        const __bf16 hi16 = (__bf16)v;
        reinterpret_cast<__bf16*>(e.dst)[idx] = part == 0 ? hi16 : (__bf16)(v - (float)hi16);
    }
}

// bf16 elements of a split-bf16 packed weight for contraction K, outputs N (0: no split-bf16 kernel takes this shape)
static long bf16x3_elements(int K, int N) {
    if (K <= 0 || N <= 0 || N % 16 || (K != 16 && K % 32)) return 0;
    return (long)(K == 16 ? 5 : 9 * (K / 32)) * (N / 16) * 2 * 64 * 8;
}

extern "C" long ngan_conv3x3_pack_elements(int Cout, int Cin, int mode, int precision) {
    if (Cout <= 0 || Cin <= 0 || Cout % 16 || Cin % 16) return 0;
    if (precision == 0) return 9L * Cin * Cout;
    if (precision == 4) return 16L * Cin * Cout;
    const int K = mode == 0 ? Cin : Cout, N = mode == 0 ? Cout : Cin;
    if (precision == 2) return K == 16 ? bf16x3_elements(32, N) : 0;      // K = 16 padded to 32 (mid kernel)
    if (precision == 3) return mode == 0 && bf16x3_elements(K, N) ? 4 * bf16x3_elements(K, N) + 9L * K * N : 0;   // folded bilinear
    return bf16x3_elements(K, N);
}

extern "C" int ngan_conv3x3_pack_many(const void* table, int n_entries, long total_elements, void* stream) {
    NGAN_REQUIRE(table && n_entries > 0 && total_elements > 0, NGAN_ERR_ARG, "conv3x3_pack_many: bad argument");
    hipLaunchKernelGGL(pack_many_kernel, dim3(ngan::ceil_div(total_elements, 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const PackEntry*>(table), n_entries, total_elements);
    return ngan::launch_status("ngan_conv3x3_pack_many");
}

extern "C" long ngan_conv3x3_packed_floats(int Cout, int Cin, int precision) {
    if (Cout <= 0 || Cin <= 0 || Cout % 16 || Cin % 16) return 0;
    if (precision == 0) return 9L * Cin * Cout;
    if (precision == 4) return 16L * Cin * Cout;
    if (precision == 2) {
        const long p0 = Cin == 16 ? bf16x3_elements(32, Cout) : 0, p1 = Cout == 16 ? bf16x3_elements(32, Cin) : 0;
        return ((p0 > p1 ? p0 : p1) + 1) / 2;
    }
    if (precision == 3) return bf16x3_elements(Cin, Cout) ? 2 * bf16x3_elements(Cin, Cout) + 9L * Cin * Cout : 0;
    const long e0 = bf16x3_elements(Cin, Cout), e1 = bf16x3_elements(Cout, Cin);   // forward / flipped orientation
    return ((e0 > e1 ? e0 : e1) + 1) / 2;
}

// exact-fp32 16 -> 16 layers on large images (plain or bilinear input): Winograd F(2x2, 3x3) form of conv3x3_tile_kernel /
// conv3x3_persist_kernel (NGAN_WINOGRAD=0: direct form, A/B switch)
static bool wino_eligible(int B, int H, int W, int K, int N, int resample) {
    static const bool on = [] { const char* e = getenv("NGAN_WINOGRAD"); return !(e && e[0] == '0'); }();
    static const bool tile_off = getenv("NGAN_TILE_KERNEL") && getenv("NGAN_TILE_KERNEL")[0] == '0';
    return on && !tile_off && K == 16 && N == 16 && resample != NGAN_RESAMPLE_POOL2 && persist_eligible(B, H, W, K, N, resample);
}

extern "C" int ngan_conv3x3_uses_bf16x3(int B, int H, int W, int K, int N, int resample, int precision) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    if (precision == 0) return wino_eligible(B, H, W, K, N, resample) ? 4 : 0;
    if (precision != 1) return 0;
    if (up2f_eligible(B, H, W, K, N, resample)) return 3;
    if (persist_eligible(B, H, W, K, N, resample)) return 1;
    if (ngan::conv3x3_mid_eligible(B, H, W, K, N)) return 1;
    // K = 16 into 32..128 channels where the persistent kernel does not apply (pooled input, small images): the mid kernel with the
    // contraction padded to 32 channels (zero weights) -- its own packed layout, hence its own precision code
    return (K == 16 && ngan::conv3x3_mid_eligible(B, H, W, 32, N)) ? 2 : 0;
}

extern "C" int ngan_conv3x3_pack_weights(const float* w_oihw, float* packed, int Cout, int Cin, int mode, float scale,
                                         int precision, void* stream) {
    NGAN_REQUIRE(w_oihw && packed, NGAN_ERR_ARG, "conv3x3_pack_weights: null pointer");
    NGAN_REQUIRE(Cout > 0 && Cin > 0 && Cout % 16 == 0 && Cin % 16 == 0, NGAN_ERR_SHAPE,
                 "conv3x3_pack_weights: Cin=%d, Cout=%d must be positive multiples of 16", Cin, Cout);
    NGAN_REQUIRE(mode == 0 || mode == 1, NGAN_ERR_ARG, "conv3x3_pack_weights: mode %d", mode);
    NGAN_REQUIRE(precision >= 0 && precision <= 4, NGAN_ERR_ARG, "conv3x3_pack_weights: precision %d", precision);
    if (precision == 4) {
        hipLaunchKernelGGL(pack_weights_wino_kernel, dim3(ngan::ceil_div(16L * Cin * Cout, 256)), dim3(256), 0, (hipStream_t)stream,
                           w_oihw, packed, Cout, Cin, mode, scale);
        return ngan::launch_status("ngan_conv3x3_pack_weights(winograd)");
    }
    if (precision == 3) {
        const long tot = ngan_conv3x3_pack_elements(Cout, Cin, mode, 3);
        NGAN_REQUIRE(tot > 0, NGAN_ERR_SHAPE, "conv3x3_pack_weights: precision 3 (folded bilinear) needs mode 0 and K = 16 or a multiple of 32");
        hipLaunchKernelGGL(pack_weights_up2f_kernel, dim3(ngan::ceil_div(tot, 256)), dim3(256), 0, (hipStream_t)stream, w_oihw, packed,
                           Cout, Cin, scale, tot);
        return ngan::launch_status("ngan_conv3x3_pack_weights(folded bilinear)");
    }
    if (precision >= 1) {
        const int K = mode == 0 ? Cin : Cout, N = mode == 0 ? Cout : Cin;
        NGAN_REQUIRE(precision == 1 || K == 16, NGAN_ERR_SHAPE, "conv3x3_pack_weights: precision 2 is the K = 16 padded layout (K=%d)", K);
        const long tot = precision == 2 ? bf16x3_elements(32, N) : bf16x3_elements(K, N);
        NGAN_REQUIRE(tot > 0, NGAN_ERR_SHAPE, "conv3x3_pack_weights: split-bf16 packing needs K = 16 or a multiple of 32 (K=%d, N=%d)", K, N);
        hipLaunchKernelGGL(pack_weights_bf16x3_kernel, dim3(ngan::ceil_div(tot, 256)), dim3(256), 0, (hipStream_t)stream,
                           w_oihw, reinterpret_cast<__bf16*>(packed), Cout, Cin, mode, scale, precision == 2 ? 1 : 0);
        return ngan::launch_status("ngan_conv3x3_pack_weights(bf16x3)");
    }
    const long total = 9L * Cin * Cout;
    hipLaunchKernelGGL(pack_weights_kernel, dim3(ngan::ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       w_oihw, packed, Cout, Cin, mode, scale);
    return ngan::launch_status("ngan_conv3x3_pack_weights");
}

// NGAN_MID_F32=0: exact-fp32 layers with 32..128 channels on small images go back to the generic kernel (A/B switch)
static bool mid_f32_enabled() {
    static const bool on = [] { const char* e = getenv("NGAN_MID_F32"); return !(e && e[0] == '0'); }();
    return on;
}

extern "C" int ngan_conv3x3_epilogue_fused(int B, int H, int W, int K, int N, int resample, int epilogue, int out_mode, int precision) {
    if (epilogue == EPI_NONE || epilogue == EPI_LRELU_PN) return 1;
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    const bool persist = persist_eligible(B, H, W, K, N, resample);
    if (epilogue == EPI_TO_IMAGE) return persist && resample == 0 && out_mode == 0 ? 1 : 0;
    if (epilogue == EPI_PN_BWD)
        return (persist && resample == 0) ||
               ((precision >= 1 || mid_f32_enabled()) && resample == 0 && ngan::conv3x3_mid_fuses_epilogue(B, H, W, precision == 2 ? 32 : K, N)) ? 1 : 0;
    return 0;
}

extern "C" int ngan_conv3x3_fwd_ex(const float* x, const float* packed, const float* bias, float* y, float* rnorm,
                                   const float* aux_in, const float* aux_rn, float* aux_out,
                                   int B, int H, int W, int K, int N, int resample, int epilogue, int out_mode,
                                   float slope, float eps, int precision, int flags, void* stream) {
    NGAN_REQUIRE(x && packed && (y || epilogue == EPI_TO_IMAGE), NGAN_ERR_ARG, "conv3x3_fwd: null pointer");
    NGAN_REQUIRE((flags & ~NGAN_CONV_SKIP_BORDER) == 0, NGAN_ERR_ARG, "conv3x3_fwd: unknown flags 0x%x", flags);
    NGAN_REQUIRE(precision == 0 || (precision == 4 && ngan_conv3x3_uses_bf16x3(B, H, W, K, N, resample, 0) == 4) ||
                 (precision != 4 && precision == ngan_conv3x3_uses_bf16x3(B, H, W, K, N, resample, 1)), NGAN_ERR_ARG,
                 "conv3x3_fwd: precision %d is not available for this shape (ask ngan_conv3x3_uses_bf16x3)", precision);
    NGAN_REQUIRE(B > 0 && H > 0 && W > 0, NGAN_ERR_SHAPE, "conv3x3_fwd: bad dims B=%d H=%d W=%d", B, H, W);
    NGAN_REQUIRE(K > 0 && K % 16 == 0, NGAN_ERR_SHAPE, "conv3x3_fwd: K=%d must be a positive multiple of 16", K);
    NGAN_REQUIRE(N == 16 || N == 32 || N == 64 || N == 128, NGAN_ERR_SHAPE, "conv3x3_fwd: N=%d must be 16/32/64/128", N);
    NGAN_REQUIRE(resample >= 0 && resample <= 2, NGAN_ERR_ARG, "conv3x3_fwd: resample %d", resample);
    NGAN_REQUIRE(epilogue >= EPI_NONE && epilogue <= EPI_TO_IMAGE, NGAN_ERR_ARG, "conv3x3_fwd: epilogue %d", epilogue);
    NGAN_REQUIRE(out_mode == 0 || (out_mode == 1 && (epilogue == EPI_NONE || epilogue == EPI_PN_BWD) && resample == 0), NGAN_ERR_ARG,
                 "conv3x3_fwd: out_mode %d needs epilogue 0 or 2 and resample 0", out_mode);
    NGAN_REQUIRE(epilogue != EPI_LRELU_PN || rnorm, NGAN_ERR_ARG, "conv3x3_fwd: epilogue 1 needs rnorm");
    NGAN_REQUIRE(epilogue != EPI_PN_BWD || (aux_in && aux_rn && resample == 0 && !bias), NGAN_ERR_ARG,
                 "conv3x3_fwd: epilogue 2 needs aux_in / aux_rn, no resampling and no bias");
    NGAN_REQUIRE(epilogue != EPI_TO_IMAGE || (aux_in && aux_out && (!y || rnorm) &&
                                              ngan_conv3x3_epilogue_fused(B, H, W, K, N, resample, epilogue, out_mode, precision)),
                 NGAN_ERR_ARG, "conv3x3_fwd: epilogue 3 needs aux_in (colour weights), aux_out and a shape ngan_conv3x3_epilogue_fused accepts");
    NGAN_REQUIRE(resample != NGAN_RESAMPLE_UP2 || (H % 2 == 0 && W % 2 == 0), NGAN_ERR_SHAPE,
                 "conv3x3_fwd: bilinear x2 needs even H, W");
    ConvArgs a{x, packed, bias, y, rnorm, B, H, W, K, N, 0, 0, slope, eps, aux_in, aux_rn, aux_out};
    hipStream_t s = (hipStream_t)stream;
    if (persist_eligible(B, H, W, K, N, resample)) {
        // large image, few channels: persistent pipelined kernel (32-bit byte offsets inside one image)
        NGAN_REQUIRE((long)H * W * (K > N ? K : N) * 16 < (1L << 32), NGAN_ERR_SHAPE, "conv3x3_fwd: one image must stay below 1 GiB (H=%d W=%d)", H, W);
        if (precision == 3) {
            NGAN_REQUIRE((epilogue == EPI_NONE || epilogue == EPI_LRELU_PN) && out_mode == 0, NGAN_ERR_ARG,
                         "conv3x3_fwd: the folded bilinear kernel has epilogues 0 and 1");
            const int st = K == 16 ? (epilogue ? launch_up2f<1, EPI_LRELU_PN>(a, s) : launch_up2f<1, EPI_NONE>(a, s))
                                   : (epilogue ? launch_up2f<2, EPI_LRELU_PN>(a, s) : launch_up2f<2, EPI_NONE>(a, s));
            if (st || (flags & NGAN_CONV_SKIP_BORDER)) return st;      // the caller launches ngan_conv3x3_up2_border itself
            return dispatch_up2_border(a, epilogue, s);
        }
        if (precision == 4)        // Winograd form (16 -> 16: checked by the precision test above): template precision 2
            return dispatch_persist2<1, 1, 2>(a, resample, epilogue, out_mode, s);
        if (N == 16) return K == 16 ? dispatch_persist<1, 1>(a, resample, epilogue, out_mode, precision, s)
                                    : dispatch_persist<1, 2>(a, resample, epilogue, out_mode, precision, s);
        return K == 16 ? dispatch_persist<2, 1>(a, resample, epilogue, out_mode, precision, s)
                       : dispatch_persist<2, 2>(a, resample, epilogue, out_mode, precision, s);
    }
    // many channels, small image: the kernel of conv3x3_mid.hip, split-bf16 (precision 2: K = 16 padded to 32) or exact fp32
    // (fp32 with a pooled input stays on the generic kernel, which measured 10 % faster there)
    if (precision >= 1 || (mid_f32_enabled() && resample != NGAN_RESAMPLE_POOL2 && ngan::conv3x3_mid_eligible(B, H, W, K, N)))
        return ngan::conv3x3_mid_launch(x, packed, bias, y, rnorm, aux_in, aux_rn, B, H, W, K, N, resample, epilogue, out_mode, slope, eps, precision, s);
    // generic exact-fp32 kernel: it has epilogues 0 and 1; the PixelNorm backward runs as a second launch, in place
    const int epi = epilogue == EPI_PN_BWD ? EPI_NONE : epilogue;
    int st;
    switch (N / 16) {
        case 1: st = dispatch_conv<0>(a, resample, epi, out_mode, s); break;
        case 2: st = dispatch_conv<1>(a, resample, epi, out_mode, s); break;
        case 4: st = dispatch_conv<2>(a, resample, epi, out_mode, s); break;
        default: st = dispatch_conv<3>(a, resample, epi, out_mode, s); break;
    }
    if (st || epilogue != EPI_PN_BWD) return st;
    return ngan_lrelu_pixelnorm_bwd(y, nullptr, aux_in, aux_rn, y, (long)B * H * W * (out_mode ? 4 : 1), N, slope, stream);
}

extern "C" int ngan_conv3x3_up2_border(const float* x, const float* packed, const float* bias, float* y, float* rnorm,
                                       int B, int H, int W, int K, int N, int epilogue, float slope, float eps, void* stream) {
    NGAN_REQUIRE(x && packed && y, NGAN_ERR_ARG, "conv3x3_up2_border: null pointer");
    NGAN_REQUIRE(ngan_conv3x3_uses_bf16x3(B, H, W, K, N, NGAN_RESAMPLE_UP2, 1) == 3, NGAN_ERR_SHAPE,
                 "conv3x3_up2_border: B=%d H=%d W=%d K=%d N=%d is not a folded-bilinear (precision 3) shape", B, H, W, K, N);
    NGAN_REQUIRE(epilogue == EPI_NONE || (epilogue == EPI_LRELU_PN && rnorm), NGAN_ERR_ARG, "conv3x3_up2_border: epilogue %d", epilogue);
    ConvArgs a{x, packed, bias, y, rnorm, B, H, W, K, N, 0, 0, slope, eps, nullptr, nullptr, nullptr};
    return dispatch_up2_border(a, epilogue, (hipStream_t)stream);
}

extern "C" int ngan_conv3x3_fwd(const float* x, const float* packed, const float* bias, float* y, float* rnorm,
                                int B, int H, int W, int K, int N, int resample, int epilogue, int out_mode,
                                float slope, float eps, int precision, int flags, void* stream) {
    NGAN_REQUIRE(epilogue == 0 || epilogue == 1, NGAN_ERR_ARG, "conv3x3_fwd: epilogue %d (2 and 3 need ngan_conv3x3_fwd_ex)", epilogue);
    return ngan_conv3x3_fwd_ex(x, packed, bias, y, rnorm, nullptr, nullptr, nullptr, B, H, W, K, N, resample, epilogue, out_mode,
                               slope, eps, precision, flags, stream);
}

extern "C" int ngan_conv3x3_kernel_name(int B, int H, int W, int K, int N, int resample, int epilogue, int out_mode,
                                        int precision, char* buf, int len) {
    NGAN_REQUIRE(buf && len > 0, NGAN_ERR_ARG, "conv3x3_kernel_name: bad buffer");
    NGAN_REQUIRE(N == 16 || N == 32 || N == 64 || N == 128, NGAN_ERR_SHAPE, "conv3x3_kernel_name: N=%d", N);
    const int mti = N == 16 ? 0 : N == 32 ? 1 : N == 64 ? 2 : 3;
    const int ci = pick_cfg(mti, B, H, W);
    if (precision == 3)
        snprintf(buf, len, "conv3x3_up2f_kernel<%d, %d>", K / 16, epilogue);
    else if (N <= 32 && K <= 32 && resample != NGAN_RESAMPLE_POOL2 && ci == 0) {
        if ((out_mode || resample == 0) && W % 32 == 0)
            snprintf(buf, len, "conv3x3_tile_kernel<%d, %d, %d, %d, %d>", N / 16, K / 16, (out_mode && epilogue != EPI_PN_BWD) ? 0 : epilogue,
                     out_mode, precision == 4 ? 2 : precision);
        else
            snprintf(buf, len, "conv3x3_persist_kernel<%d, %d, %d, %d, %d, %d>", N / 16, K / 16, out_mode ? 0 : resample,
                     (out_mode && epilogue != EPI_PN_BWD) ? 0 : epilogue, out_mode, precision == 4 ? 2 : precision);
    }
    else if ((precision >= 1 || (mid_f32_enabled() && resample != NGAN_RESAMPLE_POOL2)) && ngan::conv3x3_mid_eligible(B, H, W, precision == 2 ? 32 : K, N))
        return ngan::conv3x3_mid_kernel_name(B, H, W, precision == 2 ? 32 : K, N, resample, epilogue, out_mode, precision, buf, len);
    else {
        const TileCfg c = kCfg[mti][ci];
        snprintf(buf, len, "conv3x3_kernel<%d, %d, %d, %d, %d, %d, %d>", c.mtw, c.wn, c.pgw, c.pcg, out_mode ? 0 : resample,
                 out_mode ? 0 : epilogue, out_mode);
    }
    return NGAN_OK;
}

#ifdef NGAN_CLOCK_PROBE
extern "C" int ngan_debug_clock_stamps(unsigned long long* host_out, int n_blocks) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_clock_stamps), sizeof(unsigned long long) * 4 * n_blocks, 0, hipMemcpyDeviceToHost);
}
#endif

extern "C" int ngan_conv3x3_wgrad_kernel_name(int B, int H, int W, int Cin, int Cout, int resample, int precision, char* buf, int len) {
    NGAN_REQUIRE(buf && len > 0, NGAN_ERR_ARG, "conv3x3_wgrad_kernel_name: bad buffer");
    NGAN_REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && Cin % 16 == 0 && Cout % 16 == 0, NGAN_ERR_SHAPE,
                 "conv3x3_wgrad_kernel_name: bad shape");
    const WgradPlan p = plan_wgrad(B, H, W, Cin, Cout, precision);
    if (precision == 1) snprintf(buf, len, "wgrad_bf16x3_kernel<%d, %d, %d, %d>", p.co_s / 16, p.ci_s / 16, resample, p.tw);
    else {
        const bool wino = !(getenv("NGAN_WINOGRAD_WGRAD") && getenv("NGAN_WINOGRAD_WGRAD")[0] == '0') && p.tw == 32;
        snprintf(buf, len, "wgrad_f32_kernel<%d, %d, %d, %d, %d, %d, %d>", p.co_s / 16, p.ci_s / 16, resample, p.tw,
                 wgrad_f32_waves(p.co_s / 16, p.ci_s / 16), (resample == 0 && p.tw == 32 && W % 32 == 0) ? 1 : 0, wino ? 1 : 0);
    }
    return NGAN_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Deferred slab reduction: one launch reduces the slabs of MANY weight-gradient calls (all of a backward pass).
// An entry describes one gradient tensor and up to 4 slab sets that contribute to it (e.g. the real+fake pass and the
// two gradient-penalty terms of a critic weight), summed in a fixed order: bit-reproducible.  Entries travel as kernel
// arguments (no device table), so the launch can be captured into a HIP graph.
// ---------------------------------------------------------------------------------------------------------
struct ReduceEntry {
    const float* partial[4]; float* gw;
    int nparts[4];
    int nsrc, nslices, n_ci_slices, co_s, ci_s, K, accumulate, first_block;
    float scale[4];
};
constexpr int kReduceBatch = 24;
struct ReduceBatch { ReduceEntry e[kReduceBatch]; int n; };

// One workgroup sums kReduceSpan consecutive elements of a gradient over all slabs: a wave reads 256 contiguous bytes of one slab
// per load (whole cache lines; 16-element spans fetched half-used 128-byte lines), the waves take every 16th slab each.
// 16 waves per workgroup: a 16-channel layer has 512 slabs per source and only 36 spans, so the launch is bound by the CHAIN of
// dependent round trips per wave (128 loads, 8 in flight, x 3 sources with 4 waves: ~100 us measured); 16 waves cut the chain by 4.
constexpr int kReduceSpan = 64, kReduceThreads = 1024;
__global__ __launch_bounds__(kReduceThreads) void wgrad_reduce_many_kernel(ReduceBatch b) {
    __shared__ float red[kReduceThreads];
    int ei = 0;
    for (int i = 1; i < b.n; ++i)
        if ((int)blockIdx.x >= b.e[i].first_block) ei = i;
    const ReduceEntry& e = b.e[ei];
    const int slab = 9 * e.co_s * e.ci_s;
    const long M = (long)e.nslices * slab;
    const int tid = threadIdx.x, el = tid & (kReduceSpan - 1), grp = tid / kReduceSpan;
    constexpr int NG = kReduceThreads / kReduceSpan;
    const long i = (long)(blockIdx.x - e.first_block) * kReduceSpan + el;
    float total = 0.f;
    for (int s = 0; s < e.nsrc; ++s) {
        float acc = 0.f;
        if (i < M) {
            // eight independent partial sums: the slab reads of one thread are in flight together instead of one per round trip
            const float* src = e.partial[s] + i;
            const int np = e.nparts[s];
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f, a5 = 0.f, a6 = 0.f, a7 = 0.f;
            int j = grp;
            for (; j + 7 * NG < np; j += 8 * NG) {
                a0 += src[(long)j * M]; a1 += src[(long)(j + NG) * M];
                a2 += src[(long)(j + 2 * NG) * M]; a3 += src[(long)(j + 3 * NG) * M];
                a4 += src[(long)(j + 4 * NG) * M]; a5 += src[(long)(j + 5 * NG) * M];
                a6 += src[(long)(j + 6 * NG) * M]; a7 += src[(long)(j + 7 * NG) * M];
            }
            for (; j < np; j += NG) a0 += src[(long)j * M];
            a0 += a4; a1 += a5; a2 += a6; a3 += a7;
            acc = (a0 + a1) + (a2 + a3);
        }
        __syncthreads();
        red[tid] = acc;
        __syncthreads();
        if (tid < kReduceSpan) {
            float t = 0.f;
#pragma unroll
            for (int j = 0; j < NG; ++j) t += red[tid + kReduceSpan * j];
            total = fmaf(t, e.scale[s], total);
        }
    }
    if (tid < kReduceSpan && i < M) {
        int r = (int)(i % slab);
        const int slice = (int)(i / slab);
        const int ci_l = r % e.ci_s; r /= e.ci_s;
        const int co_l = r % e.co_s;
        const int tap = r / e.co_s;
        const int co = (slice / e.n_ci_slices) * e.co_s + co_l, ci = (slice % e.n_ci_slices) * e.ci_s + ci_l;
        float* o = e.gw + ((long)co * e.K + ci) * 9 + tap;
        *o = e.accumulate ? *o + total : total;
    }
}

extern "C" size_t ngan_conv3x3_wgrad_workspace_bytes(int B, int H, int W, int Cin, int Cout) {
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || Cin % 16 || Cout % 16) return 0;
    WgradPlan p = plan_wgrad(B, H, W, Cin, Cout);
    return (size_t)p.nwx * p.nslices * 9 * p.co_s * p.ci_s * sizeof(float);
}

extern "C" int ngan_conv3x3_wgrad_plan(int B, int H, int W, int Cin, int Cout, int precision, int* out5) {
    NGAN_REQUIRE(out5 && B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && Cin % 16 == 0 && Cout % 16 == 0, NGAN_ERR_ARG,
                 "conv3x3_wgrad_plan: bad argument");
    WgradPlan p = plan_wgrad(B, H, W, Cin, Cout, precision);
    out5[0] = p.nwx; out5[1] = p.nslices; out5[2] = p.n_ci_slices; out5[3] = p.co_s; out5[4] = p.ci_s;
    return NGAN_OK;
}

// entries: host array of n records { const float* partial[4]; float* gw; int nparts[4]; int nsrc, nslices, n_ci_slices, co_s,
// ci_s, K, accumulate, first_block(ignored); float scale[4]; }  (104 bytes each)
extern "C" int ngan_conv3x3_wgrad_reduce_many(const void* entries, int n, void* stream) {
    NGAN_REQUIRE(entries && n > 0, NGAN_ERR_ARG, "conv3x3_wgrad_reduce_many: bad argument");
    const ReduceEntry* src = reinterpret_cast<const ReduceEntry*>(entries);
    for (int base = 0; base < n; base += kReduceBatch) {
        ReduceBatch b;
        b.n = n - base < kReduceBatch ? n - base : kReduceBatch;
        int blocks = 0;
        for (int i = 0; i < b.n; ++i) {
            b.e[i] = src[base + i];
            NGAN_REQUIRE(b.e[i].nsrc >= 1 && b.e[i].nsrc <= 4 && b.e[i].gw, NGAN_ERR_ARG, "conv3x3_wgrad_reduce_many: bad entry %d", base + i);
            b.e[i].first_block = blocks;
            blocks += ngan::ceil_div((long)b.e[i].nslices * 9 * b.e[i].co_s * b.e[i].ci_s, kReduceSpan);
        }
        hipLaunchKernelGGL(wgrad_reduce_many_kernel, dim3(blocks), dim3(kReduceThreads), 0, (hipStream_t)stream, b);
        int st = ngan::launch_status("ngan_conv3x3_wgrad_reduce_many");
        if (st) return st;
    }
    return NGAN_OK;
}

extern "C" int ngan_conv3x3_wgrad(const float* x, const float* g, float* gw, float* workspace,
                                  int B, int H, int W, int Cin, int Cout, int resample, float scale, int accumulate,
                                  int precision, void* stream) {
    // accumulate == 2: write the slabs only; the caller reduces them later with ngan_conv3x3_wgrad_reduce_many
    NGAN_REQUIRE(x && g && gw && workspace, NGAN_ERR_ARG, "conv3x3_wgrad: null pointer");
    NGAN_REQUIRE(B > 0 && H > 0 && W > 0, NGAN_ERR_SHAPE, "conv3x3_wgrad: bad dims B=%d H=%d W=%d", B, H, W);
    NGAN_REQUIRE(Cin > 0 && Cout > 0 && Cin % 16 == 0 && Cout % 16 == 0, NGAN_ERR_SHAPE,
                 "conv3x3_wgrad: Cin=%d, Cout=%d must be positive multiples of 16", Cin, Cout);
    NGAN_REQUIRE(resample >= 0 && resample <= 2, NGAN_ERR_ARG, "conv3x3_wgrad: resample %d", resample);
    NGAN_REQUIRE(resample != NGAN_RESAMPLE_UP2 || (H % 2 == 0 && W % 2 == 0), NGAN_ERR_SHAPE,
                 "conv3x3_wgrad: bilinear x2 needs even H, W");
    WgradPlan p = plan_wgrad(B, H, W, Cin, Cout, precision);
    WgradArgs a{x, g, workspace, B, H, W, Cin, Cout, p.tiles_x, p.tiles_y, p.n_tiles, p.n_ci_slices};
    hipStream_t s = (hipStream_t)stream;
    int st;
    NGAN_REQUIRE(precision == 0 || precision == 1, NGAN_ERR_ARG, "conv3x3_wgrad: precision %d", precision);
    NGAN_REQUIRE(precision == 0 || (long)H * W * (Cin > Cout ? Cin : Cout) * 16 < (1L << 32), NGAN_ERR_SHAPE,
                 "conv3x3_wgrad: one image must stay below 1 GiB (H=%d W=%d): the split-bf16 kernel uses 32-bit byte offsets", H, W);
    if (p.co_s == 32 && p.ci_s == 32) st = launch_wgrad<2, 2>(a, p, resample, precision, s);
    else if (p.co_s == 32) st = launch_wgrad<2, 1>(a, p, resample, precision, s);
    else if (p.ci_s == 32) st = launch_wgrad<1, 2>(a, p, resample, precision, s);
    else st = launch_wgrad<1, 1>(a, p, resample, precision, s);
    if (st || accumulate == 2) return st;
    const long M = (long)p.nslices * 9 * p.co_s * p.ci_s;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(ngan::ceil_div(M, 16)), dim3(256), 0, s, workspace, gw, p.nwx,
                       p.nslices, p.n_ci_slices, p.co_s, p.ci_s, Cin, scale, accumulate);
    return ngan::launch_status("ngan_conv3x3_wgrad(reduce)");
}
