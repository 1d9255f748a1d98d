// Pieces shared by the translation units of the 3x3 convolution family (one kernel family per file, so that an experiment on one
// of them rebuilds that file only):
//   conv3x3_pack.hip     weight packing                          conv3x3_generic.hip  generic exact-fp32 kernel (any tile shape)
//   conv3x3_persist.hip  persistent kernel (bilinear / ragged)   conv3x3_tile.hip     tile kernel (plain input, W % 32 == 0), Winograd forms
//   conv3x3_up2f.hip     bilinear folded into the weights        conv3x3_mid.hip      many channels on small images
//   conv3x3_wino.hip     Winograd form of the 32-channel layers  conv3x3_wgrad.hip    weight gradients + slab reduction
//   conv3x3_api.hip      the C ABI's dispatch (include/ngan.h)
#pragma once
#include <cstdlib>
#include "conv3x3_shared.h"

// A/B switches for measurements exist in the DIAGNOSTIC build only (`make diag`: -DNGAN_DIAG, written to build/diag/, never into
// the package directory).  In the product library every switch is the compile-time constant of its default: no environment
// variable changes which kernel a call runs.
#ifdef NGAN_DIAG
namespace ngan {
inline bool diag_flag(const char* name, bool dflt) { const char* e = getenv(name); return e ? e[0] != '0' : dflt; }
inline int diag_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
}
#define NGAN_DIAG_FLAG(name, dflt) ([] { static const bool v = ::ngan::diag_flag(name, dflt); return v; }())
#define NGAN_DIAG_INT(name, dflt) ([] { static const int v = ::ngan::diag_int(name, dflt); return v; }())
#else
#define NGAN_DIAG_FLAG(name, dflt) (dflt)
#define NGAN_DIAG_INT(name, dflt) (dflt)
#endif

// workgroups of the 16 -> 16 Winograd instances the register allocation makes room for per CU (waves per SIMD; build-time knob)
#ifndef NGAN_WINO16_WPE
#define NGAN_WINO16_WPE 2
#endif

namespace ngan {
// launchers exported by the kernel files (template instance chosen at run time from the arguments)
int conv3x3_tile_launch(const ConvArgs& a, int mtw, int kg, int epilogue, int out_mode, int tprec, hipStream_t s);                  // conv3x3_tile.hip
int conv3x3_persist_launch(const ConvArgs& a, int mtw, int kg, int resample, int epilogue, int out_mode, int tprec, hipStream_t s); // conv3x3_persist.hip
int conv3x3_wino_launch(const ConvArgs& a, int mtw, int kg, int resample, int epilogue, int out_mode, hipStream_t s);                              // conv3x3_wino.hip
int conv3x3_wino_tile_rows(int mtw, int kg);
int conv3x3_up2f_launch(const ConvArgs& a, int epilogue, hipStream_t s);                                                            // conv3x3_up2f.hip
int conv3x3_up2_border_launch(const ConvArgs& a, int epilogue, hipStream_t s);
int conv3x3_generic_launch(const ConvArgs& a, int resample, int epilogue, int out_mode, hipStream_t s);                             // conv3x3_generic.hip
long conv3x3_bf16x3_elements(int K, int N);                                                                                          // conv3x3_pack.hip
// bf16 activation storage (precision code 5): conv3x3_bf16.hip
long conv3x3_bf16_elements(int K, int N);                   // bf16 elements of a packed weight (0: channel counts the kernel does not take)
int conv3x3_bf16_pack_launch(const float* w, float* packed, int Cout, int Cin, int mode, float scale, hipStream_t s);
int conv3x3_bf16_kernel_name(int B, int H, int W, int K, int N, char* buf, int len);
}  // namespace ngan

namespace {

struct TileRun { int t, t_end, step; };
__device__ __forceinline__ TileRun tile_run(int n_tiles) {      // this workgroup's tiles: t, t + step, ... < t_end
    const int xcd = blockIdx.x & 7, band = (n_tiles + 7) >> 3;
    TileRun r;
    r.step = gridDim.x >> 3;
    r.t = xcd * band + (blockIdx.x >> 3);
    r.t_end = min((xcd + 1) * band, n_tiles);
    return r;
}

// Tile coordinates of a persistent kernel's walk t, t + step, ...: decoded ONCE (the two integer divisions are ~40 scalar instructions each on
// this ISA, and they used to run twice per tile -- once for the tile being computed, once inside the prefetch of the next one -- in every
// wave, between the barrier and the issue of the next tile's loads), then advanced by the decomposed step with two carries.
struct TileCursor { int b, ty, tx; };
struct TileWalk {
    int tiles_x, tiles_y, db, dy, dx;
    __device__ __forceinline__ TileWalk(int tiles_x_, int tiles_y_, int step) : tiles_x(tiles_x_), tiles_y(tiles_y_) {
        dx = step % tiles_x;
        const int r = step / tiles_x;
        dy = r % tiles_y;
        db = r / tiles_y;
    }
    __device__ __forceinline__ TileCursor at(int t) const {
        TileCursor c;
        c.tx = t % tiles_x; t /= tiles_x;
        c.ty = t % tiles_y;
        c.b = t / tiles_y;
        return c;
    }
    __device__ __forceinline__ TileCursor next(TileCursor c) const {      // the tile `step` further on (dx < tiles_x, dy < tiles_y: one carry each)
        c.tx += dx;
        if (c.tx >= tiles_x) { c.tx -= tiles_x; c.ty += 1; }
        c.ty += dy;
        if (c.ty >= tiles_y) { c.ty -= tiles_y; c.b += 1; }
        c.b += db;
        return c;
    }
};

inline int persistent_grid(int n_tiles, int resident) {
    const int per_cu = NGAN_DIAG_INT("NGAN_PERSIST_WG_PER_CU", 0);
    const int cap = per_cu > 0 ? per_cu * 256 : 1 << 30;
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
__device__ __forceinline__ void pin_registers(f32x4& v) { asm volatile("" : "+v"(v)); }
// Loop-invariant operands fetched in a persistent kernel's prologue (bias, colour weights) are pinned ONCE before the tile loop:
// the compiler's wait-count pass is not path sensitive, so a load that may still be pending on the loop's entry edge turns the
// first in-loop use of its registers into an `s_waitcnt vmcnt(0)` on EVERY iteration -- in the middle of the MFMA section, where
// it also drains the next tile's prefetch issued just before (found in the ISA of the 16 -> 16 Winograd instances, round 3).
// ... and not before `dep` has been computed (an accumulator of the last MFMA: the scheduler may not hoist the wait above the MFMAs)
__device__ __forceinline__ void pin_registers_after(float4& v, float& dep) {
    asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w), "+v"(dep));
}

// plain-bf16 packing: fp32 OIHW master weights * scale -> bf16, layout [step][n-tile j][lane][8]; lane l holds output channel
// n = 16 j + (l & 15) and contraction index kk = 8 (l >> 4) + e.  K = 16: step = tap pair (kk < 16 -> tap 2 step, else 2 step + 1; tap 9 is
// zero padding), 5 steps; K = 32 KS: step = tap * KS + ks, channel 32 ks + kk.  mode 1 (input gradient): contraction over the
// OUTPUT channels of the layer, taps flipped.
__device__ __forceinline__ __bf16 bf16_weight(const float* __restrict__ w, int Cout, int Cin, int mode, float scale, long idx) {
    const int K = mode == 0 ? Cin : Cout, N = mode == 0 ? Cout : Cin;
    const int NT = N / 16, KS = K / 32;
    const int e = idx & 7, lane = (idx >> 3) & 63;
    const long r = idx >> 9;
    const int j = r % NT, step = r / NT;
    const int n = j * 16 + (lane & 15), kk = 8 * (lane >> 4) + e;
    const int tap = K == 16 ? 2 * step + (kk >> 4) : step / KS;
    const int k = K == 16 ? (kk & 15) : (step % KS) * 32 + kk;
    float v = 0.f;
    if (tap < 9) v = mode == 0 ? w[((long)n * Cin + k) * 9 + tap] : w[((long)k * Cin + n) * 9 + (8 - tap)];
    return (__bf16)(v * scale);
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

// three-way split (conv3x3_tile_kernel PREC = 3, diagnostic build): hi and mid where the two-way split keeps hi and lo, lo in the same
// slot of a second plane
template <int PLANE>
__device__ __forceinline__ void st_split3(float* tile, int idx, float4 v) {
    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
    bf16x4 hi, mid, lo;
    const float x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        hi[i] = (__bf16)x[i];
        const float r1 = x[i] - (float)hi[i];
        mid[i] = (__bf16)r1;
        lo[i] = (__bf16)(r1 - (float)mid[i]);
    }
    *reinterpret_cast<bf16x4*>(&tile[idx]) = hi;
    *reinterpret_cast<bf16x4*>(&tile[idx ^ 8]) = mid;
    *reinterpret_cast<bf16x4*>(&tile[idx + PLANE]) = lo;
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
inline int pick_cfg(int mti, int B, int H, int W) {
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
    return NGAN_DIAG_FLAG("NGAN_UP2_FOLDED", true) && resample == NGAN_RESAMPLE_UP2 && N == 16 && (K == 16 || K == 32) && H % 2 == 0 && W % 2 == 0 && H >= 16 && W >= 32 &&
           persist_eligible(B, H, W, K, N, resample);
}

}  // namespace
