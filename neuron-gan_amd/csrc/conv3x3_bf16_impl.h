// Kernel template of the bf16-storage 3x3 convolution (precision code 5); included by conv3x3_bf16_k{16,32,64,128}.hip, one translation
// unit per contraction width so that the 64 instances build in parallel.  Design notes: conv3x3_bf16.hip.
#pragma once
#include "conv3x3_internal.h"

namespace ngan {
struct ConvArgsB {
    const __bf16* x; const __bf16* wp; const float* bias; __bf16* y; float* rn;
    int B, H, W, tiles_x, tiles_y, n_tiles, band;
    int resample, epilogue, out_mode;
    float slope, eps;
    const __bf16* ay; const float* wimg; const float* arn; float* aout;
};
// one launcher per contraction width (conv3x3_bf16_k*.hip): picks the instance for (N, tile shape) and launches it
int conv3x3_bf16_launch_k16(ConvArgsB a, int N, int pgt, bool narrow, hipStream_t s);
int conv3x3_bf16_launch_k32(ConvArgsB a, int N, int pgt, bool narrow, hipStream_t s);
int conv3x3_bf16_launch_k64(ConvArgsB a, int N, int pgt, bool narrow, hipStream_t s);
int conv3x3_bf16_launch_k128(ConvArgsB a, int N, int pgt, bool narrow, hipStream_t s);
}  // namespace ngan
using ngan::ConvArgsB;

namespace {

typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
constexpr unsigned BF16_OOB = 0xFFFFFFF0u;          // a buffer offset beyond every image: the load returns zeros, the store is dropped

struct f8 { float v[8]; };
__device__ __forceinline__ f8 unpack8(u32x4 u) {
    f8 r;
    r.v[0] = bf16_lo(u[0]); r.v[1] = bf16_hi(u[0]); r.v[2] = bf16_lo(u[1]); r.v[3] = bf16_hi(u[1]);
    r.v[4] = bf16_lo(u[2]); r.v[5] = bf16_hi(u[2]); r.v[6] = bf16_lo(u[3]); r.v[7] = bf16_hi(u[3]);
    return r;
}
__device__ __forceinline__ u32x4 pack8(const f8& f) {
    return (u32x4){pack_bf16(f.v[0], f.v[1]), pack_bf16(f.v[2], f.v[3]), pack_bf16(f.v[4], f.v[5]), pack_bf16(f.v[6], f.v[7])};
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(base)), 0, bytes, 0x00020000);
}
__device__ __forceinline__ float4 ld4bf(__amdgpu_buffer_rsrc_t rs, unsigned off) {          // 4 bf16 channels at byte offset off
    const u32x2_t u = __builtin_bit_cast(u32x2_t, __builtin_amdgcn_raw_buffer_load_b64(rs, off, 0, 0));
    return make_float4(bf16_lo(u[0]), bf16_hi(u[0]), bf16_lo(u[1]), bf16_hi(u[1]));
}
__device__ __forceinline__ void st4bf(__amdgpu_buffer_rsrc_t rs, unsigned off, float4 v) {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, (u32x2_t){pack_bf16(v.x, v.y), pack_bf16(v.z, v.w)}), rs, off, 0, 0);
}

// PGT pixel groups of 16 per workgroup tile.  NS = false: the 4 waves split the PIXEL groups (PGT / 4 each) and hold all N / 16 output
// tiles (N = 16, 32: the large-image layers).  NS = true (N = 64, 128): the waves split the OUTPUT tiles (N / 64 each), every wave
// contracts all PGT pixel groups -- a weight fragment is fetched by exactly one wave of the workgroup, and the per-pixel sums of the
// PixelNorm epilogues cross the waves through 4 KB of LDS.  NARROW: 16-pixel-wide tiles for images at most 16 wide.
// Tried and rejected (round 4): a PERSISTENT form of the pixel-split instances (a band of tiles per workgroup, tile t + 1's staging loads
// requested before tile t's MFMAs, weights fetched once).  Inside a tile loop the compiler hoists every tile-invariant index -- staging
// slots, LDS addresses, tap offsets -- into loop-carried registers: 133 instead of 61 for 16 -> 16 (three workgroups per CU instead of
// eight); capping the registers (launch bounds, lane coordinates behind an opaque barrier) made it spill 44 - 125 registers to scratch
// instead of recomputing, and the 16 -> 16 forward went from 63 to 152 us.  The one-tile-per-workgroup form below stays: what hides its
// staging and epilogue latency is occupancy (64 registers, 14 KB of LDS: eight workgroups per CU).
template <int K, int N, int PGT, bool NS, bool NARROW>
// (the 16 -> 16 instances fit 64 registers: eight workgroups per CU instead of four.  303 VALU + 190 SALU + 20 MFMA instructions per wave and
// tile: 47 % of every SIMD's cycles go into vector-instruction issue and some instruction is active 91 % of the time -- the kernel is bound
// by the instructions of its staging and epilogue, and the extra waves give the issue logic something to pick from while others wait;
// PMC record and its corrected reading: profiles/r04_pmc_bf16_1616.txt)
__global__ __launch_bounds__(256, (!NS && K == 16 && N == 16) ? 8 : 1) void conv3x3_bf16_kernel(ConvArgsB a) {
    constexpr int P = K / 8, NT = N / 16, KS = K >= 32 ? K / 32 : 1, S = K == 16 ? 5 : 9 * KS;
    constexpr int TW = NARROW ? 16 : 32, GPR = TW / 16, TH = PGT / GPR, HH = TH + 2, HW = TW + 2, NPIX = HH * HW;
    constexpr int NCHUNK = NPIX * P, NCH = (NCHUNK + 255) / 256;
    constexpr int NTW = NS ? NT / 4 : NT, PGW = NS ? PGT : PGT / 4;
    constexpr bool WREG = S * NTW <= 18;                     // the wave's weight fragments fit its registers for the whole tile
    constexpr int RD = S < 8 ? S : 8;                        // otherwise: a register ring, RD contraction steps ahead of their use
    static_assert(PGT % GPR == 0 && (NS || PGT % 4 == 0) && (!NS || NT % 4 == 0), "tile split");
    constexpr int PH = (TH + 1) / 2 + 2, PW = TW / 2 + 2, NPCH = PH * PW * P, NPC = (NPCH + 255) / 256;   // low-resolution patch of a bilinear input
    __shared__ u32x4 tile[NPIX * P];
    __shared__ u32x4 patch[NPCH];
    __shared__ float xs[NS ? 4 * 4 * PGT * 16 : 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), p = lane & 15, q = lane >> 4;

    const int t = (blockIdx.x & 7) * a.band + (blockIdx.x >> 3);
    if (t >= a.n_tiles) return;
    const int txi = t % a.tiles_x, tyi = (t / a.tiles_x) % a.tiles_y, b = t / (a.tiles_x * a.tiles_y);
    const int y0 = tyi * TH, x0 = txi * TW;
    const int H = a.H, W = a.W;
    const int j0 = NS ? wave * NTW : 0;                      // this wave's first output tile

    // column swizzle of the 16-byte chunk index (conv3x3_bf16.hip)
    auto swz = [](int X) -> int { return P > 2 ? 2 * ((X / (16 / P)) & (P / 2 - 1)) : 0; };
    const __bf16* wlane = a.wp + (long)lane * 8;
    auto wfrag = [&](int s, int j) -> bf16x8 { return *reinterpret_cast<const bf16x8*>(wlane + ((long)s * NT + j0 + j) * 512); };

    // ---- weights: all of them (WREG) or the first RD steps of the ring, requested before the tile so that the round trips overlap
    bf16x8 wr[WREG ? S : RD][NTW];
#pragma unroll
    for (int s = 0; s < (WREG ? S : RD); ++s)
#pragma unroll
        for (int j = 0; j < NTW; ++j) wr[s][j] = wfrag(s, j);

    // ---- stage the halo tile: HH x HW pixels x K channels, 16 bytes per item, resampled on the way.  Loads go through a buffer
    // descriptor of the tile's image: a row above / below the image is an out-of-range offset by itself (zeros: the conv padding),
    // a column outside costs one select; no branch around any load.
    if (a.resample == NGAN_RESAMPLE_NONE) {
        const __amdgpu_buffer_rsrc_t rs = rsrc_of(a.x + (long)b * H * W * K, (unsigned)(H * W * K) * 2u);
        const int tile_off = ((y0 - 1) * W + x0 - 1) * K * 2;
        u32x4 v[NCH];
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int e = tid + i * 256;
            const int pix = e / P, sl = e % P, hy = pix / HW, hx = pix % HW;
            const bool ok = e < NCHUNK && (unsigned)(x0 + hx - 1) < (unsigned)W && (unsigned)(y0 + hy - 1) < (unsigned)H;
            v[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? (unsigned)(tile_off + ((hy * W + hx) * K + sl * 8) * 2) : BF16_OOB, 0, 0));
        }
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int e = tid + i * 256;
            const int pix = e / P, sl = e % P, hx = pix % HW;
            if (e < NCHUNK) tile[pix * P + (sl ^ swz(hx))] = v[i];
        }
    } else if (a.resample == NGAN_RESAMPLE_POOL2) {
        // x is (B, 2H, 2W, K); a staged element is the 2x2 mean, associated like ngan_pool2_fwd: 0.25 * ((a + b) + (c + d))
        const int W2 = 2 * W;
        const __amdgpu_buffer_rsrc_t rs = rsrc_of(a.x + (long)b * 4 * H * W * K, (unsigned)(4 * H * W * K) * 2u);
#pragma unroll 2
        for (int i = 0; i < NCH; ++i) {
            const int e = tid + i * 256;
            const int pix = e / P, sl = e % P, hy = pix / HW, hx = pix % HW;
            const int gy = y0 + hy - 1, gx = x0 + hx - 1;
            const bool ok = e < NCHUNK && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
            const unsigned o = ok ? (unsigned)(((2 * gy * W2 + 2 * gx) * K + sl * 8) * 2) : BF16_OOB;
            const unsigned o1 = ok ? o + K * 2 : BF16_OOB, o2 = ok ? o + W2 * K * 2 : BF16_OOB, o3 = ok ? o + (W2 + 1) * K * 2 : BF16_OOB;
            const f8 p00 = unpack8(__builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, o, 0, 0)));
            const f8 p01 = unpack8(__builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, o1, 0, 0)));
            const f8 p10 = unpack8(__builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, o2, 0, 0)));
            const f8 p11 = unpack8(__builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, o3, 0, 0)));
            f8 r;
#pragma unroll
            for (int c = 0; c < 8; ++c) r.v[c] = 0.25f * ((p00.v[c] + p01.v[c]) + (p10.v[c] + p11.v[c]));
            if (e < NCHUNK) tile[pix * P + (sl ^ swz(hx))] = pack8(r);
        }
    } else {
        // x is (B, H/2, W/2, K); bilinear x2, align_corners = False (models.py:87-89), the taps and the association of up2_fwd_kernel.
        // The LOW-resolution patch under the halo tile ((TH + 1) / 2 + 2 rows x TW / 2 + 2 columns, clamped coordinates = the taps'
        // edge rule) is staged once -- one or two 16-byte loads per thread instead of four dependent loads per staged item -- and
        // expanded LDS -> LDS.  Hi-res index Y blends low-res rows i0 = (Y - 1) >> 1 and i0 + 1 with weights (.75, .25) for odd Y and
        // (.25, .75) for even Y; rows / columns outside the hi-res image are the conv's zero padding, not a blend.
        const int h = H >> 1, w = W >> 1;
        const __amdgpu_buffer_rsrc_t rs = rsrc_of(a.x + (long)b * h * w * K, (unsigned)(h * w * K) * 2u);
        const int ly0 = (y0 - 2) >> 1, lx0 = (x0 - 2) >> 1;
        u32x4 pv[NPC];
#pragma unroll
        for (int i = 0; i < NPC; ++i) {
            const int e = tid + i * 256;
            const int pp = e / P, sl = e % P, pr = pp / PW, pc = pp % PW;
            const int ly = min(max(ly0 + pr, 0), h - 1), lx = min(max(lx0 + pc, 0), w - 1);
            pv[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, e < NPCH ? (unsigned)(((ly * w + lx) * K + sl * 8) * 2) : BF16_OOB, 0, 0));
        }
#pragma unroll
        for (int i = 0; i < NPC; ++i)
            if (tid + i * 256 < NPCH) patch[tid + i * 256] = pv[i];
        __syncthreads();
#pragma unroll 2
        for (int i = 0; i < NCH; ++i) {
            const int e = tid + i * 256;
            const int pix = e / P, sl = e % P, hy = pix / HW, hx = pix % HW;
            const int Y = y0 + hy - 1, X = x0 + hx - 1;
            const bool ok = (unsigned)Y < (unsigned)H && (unsigned)X < (unsigned)W;
            const int r0 = ((Y - 1) >> 1) - ly0, c0 = ((X - 1) >> 1) - lx0;
            const float wya = (Y & 1) ? 0.75f : 0.25f, wxa = (X & 1) ? 0.75f : 0.25f, wyb = 1.0f - wya, wxb = 1.0f - wxa;
            u32x4 o = (u32x4){0u, 0u, 0u, 0u};
            if (e < NCHUNK && ok) {
                const u32x4* pr0 = patch + (r0 * PW + c0) * P + sl;
                const f8 t0 = unpack8(pr0[0]), t1 = unpack8(pr0[P]), b0 = unpack8(pr0[PW * P]), b1 = unpack8(pr0[PW * P + P]);
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

    // ---- contraction.  Pixel group gi of this wave: tile row gi / GPR, columns 16 (gi % GPR) + p.  The B-operand fragment of
    // (group, step) is ONE ds_read_b128 at  pbase[group] + toff[step]  (toff: tap offset, 32-channel sub-block and the column
    // swizzle, which does not depend on the group because a group starts at a multiple of 16 columns)
    int prow[PGW], pcol[PGW], pbase[PGW];
#pragma unroll
    for (int pg = 0; pg < PGW; ++pg) {
        const int gi = NS ? pg : wave * PGW + pg;
        prow[pg] = gi / GPR;
        pcol[pg] = (gi % GPR) * 16 + p;
        pbase[pg] = (prow[pg] * HW + pcol[pg]) * P;
    }
    auto toff = [&](int s) -> int {
        int tap, sl;
        if (K == 16) {
            tap = 2 * s + (q >> 1);
            tap = tap > 8 ? 8 : tap;                  // the zero-weight padding tap: any valid address
            sl = q & 1;
        } else {
            tap = s / KS;
            sl = (s % KS) * 4 + q;
        }
        const int dy = (tap * 11) >> 5, dx = tap - 3 * dy;
        return (dy * HW + dx) * P + (sl ^ swz(p + dx));
    };
    f32x4 acc[PGW][NTW];
#pragma unroll
    for (int pg = 0; pg < PGW; ++pg)
#pragma unroll
        for (int j = 0; j < NTW; ++j) acc[pg][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const int to = toff(s);
        bf16x8 bf[PGW];
#pragma unroll
        for (int pg = 0; pg < PGW; ++pg) bf[pg] = __builtin_bit_cast(bf16x8, tile[pbase[pg] + to]);
#pragma unroll
        for (int j = 0; j < NTW; ++j)
#pragma unroll
            for (int pg = 0; pg < PGW; ++pg)
                acc[pg][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wr[WREG ? s : s % RD][j], bf[pg], acc[pg][j], 0, 0, 0);
        if (!WREG && s + RD < S) {
#pragma unroll
            for (int j = 0; j < NTW; ++j) wr[s % RD][j] = wfrag(s + RD, j);
        }
    }

    // ---- epilogue: lane (p, q) holds channels 16 (j0 + j) + 4 q .. + 3 of pixel (prow, pcol) of each of its groups.  A per-pixel sum
    // over the N channels is sum_rows4 over the 4 k-group rows of the wave and, with NS, a fixed-order sum over the 4 waves through LDS.
    const int epi = a.epilogue;
    const float inv_n = 1.0f / (float)N;
    auto xsum = [&](float* v, int n) {              // v[i] (all q rows hold it) -> the sum over the 4 waves, identically on every wave
        if constexpr (NS) {
            for (int i = 0; i < n; ++i)
                if (q == 0) xs[(wave * 4 * PGT + i) * 16 + p] = v[i];
            __syncthreads();
            for (int i = 0; i < n; ++i)
                v[i] = (xs[(0 * 4 * PGT + i) * 16 + p] + xs[(1 * 4 * PGT + i) * 16 + p]) + (xs[(2 * 4 * PGT + i) * 16 + p] + xs[(3 * 4 * PGT + i) * 16 + p]);
            __syncthreads();
        }
    };
    float4 bv[NTW], wimg[NTW];
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
        bv[j] = a.bias ? ld4(a.bias + (j0 + j) * 16 + q * 4) : f4zero();
        wimg[j] = epi == EPI_TO_IMAGE ? ld4(a.wimg + (j0 + j) * 16 + q * 4) : f4zero();
    }
    bool valid[PGW];
    unsigned poff[PGW];                                // pixel index inside the image, or OOB
    float4 v[PGW][NTW];
#pragma unroll
    for (int pg = 0; pg < PGW; ++pg) {
        const int gy = y0 + prow[pg], gx = x0 + pcol[pg];
        valid[pg] = gy < H && gx < W;
        poff[pg] = valid[pg] ? (unsigned)(gy * W + gx) : 0u;
#pragma unroll
        for (int j = 0; j < NTW; ++j)
            v[pg][j] = make_float4(acc[pg][j][0] + bv[j].x, acc[pg][j][1] + bv[j].y, acc[pg][j][2] + bv[j].z, acc[pg][j][3] + bv[j].w);
    }
    const long img_pix = (long)b * H * W;
    if (epi == EPI_LRELU_PN || epi == EPI_TO_IMAGE) {
        float ss[PGW];
#pragma unroll
        for (int pg = 0; pg < PGW; ++pg) {
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < NTW; ++j) {
                float4 c = v[pg][j];
                c.x = vmax1(c.x, a.slope * c.x); c.y = vmax1(c.y, a.slope * c.y);       // LeakyReLU, 0 <= slope <= 1
                c.z = vmax1(c.z, a.slope * c.z); c.w = vmax1(c.w, a.slope * c.w);
                s += f4dot(c, c);
                v[pg][j] = c;
            }
            ss[pg] = sum_rows4(s);
        }
        xsum(ss, PGW);
        // the norms of a wave's pixel groups leave in ONE store instruction: every k-group row q holds all the sums, row q stores group q
        // (one 64-byte run per group; four separate instructions with three quarters of their lanes switched off cost 4x the issue slots)
        const __amdgpu_buffer_rsrc_t rn_rs = rsrc_of(a.rn ? a.rn + img_pix : nullptr, a.rn ? (unsigned)(H * W) * 4u : 0u);
        float rn_val = 0.f;
        unsigned rn_off = BF16_OOB;
        static_assert(PGW <= 4, "one k-group row per pixel group");
#pragma unroll
        for (int pg = 0; pg < PGW; ++pg) {
            const float m = ss[pg] * inv_n + a.eps;
            const float inv = __builtin_amdgcn_rsqf(m);
#pragma unroll
            for (int j = 0; j < NTW; ++j) v[pg][j] = f4scale(v[pg][j], inv);
            if (q == pg) {
                rn_val = m * inv;
                rn_off = valid[pg] ? poff[pg] * 4u : BF16_OOB;
            }
        }
        if (!NS || wave == 0) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, rn_val), rn_rs, rn_off, 0, 0);
        if (epi == EPI_TO_IMAGE) {                     // (N = 16 / 32 instances only: the launcher refuses it for NS)
#pragma unroll
            for (int pg = 0; pg < PGW; ++pg) {
                float d = 0.f;
#pragma unroll
                for (int j = 0; j < NTW; ++j) d += f4dot(v[pg][j], wimg[j]);
                d = sum_rows4(d);
                if (valid[pg] && q == 0) a.aout[img_pix + poff[pg]] = tanhf(d);
            }
        }
    }
    // (Pixel split: every per-pixel sum is complete inside the wave, so the groups / pooled-over pixels are finished one after the other
    // and few registers are live at a time.  Output-tile split: the sums cross the waves, so all of them are formed first and
    // exchanged in ONE pair of barriers.)
    if (a.out_mode == 0) {
        const __amdgpu_buffer_rsrc_t y_rs = rsrc_of(a.y ? a.y + img_pix * N : nullptr, a.y ? (unsigned)(H * W * N) * 2u : 0u);
        if (epi == EPI_PN_BWD) {
            // backward of the LeakyReLU -> PixelNorm that produced this layer's input, applied to the gradient just computed
            const __amdgpu_buffer_rsrc_t ay_rs = rsrc_of(a.ay + img_pix * N, (unsigned)(H * W * N) * 2u);
            if constexpr (NS) {
                float4 yy[PGW][NTW];
                float sd[PGW], rr[PGW];
#pragma unroll
                for (int pg = 0; pg < PGW; ++pg) {
                    float s = 0.f;
#pragma unroll
                    for (int j = 0; j < NTW; ++j) {
                        yy[pg][j] = ld4bf(ay_rs, (poff[pg] * N + (j0 + j) * 16 + q * 4) * 2u);
                        s += f4dot(v[pg][j], yy[pg][j]);
                    }
                    rr[pg] = a.arn[img_pix + poff[pg]];
                    sd[pg] = sum_rows4(s);
                }
                xsum(sd, PGW);
#pragma unroll
                for (int pg = 0; pg < PGW; ++pg) {
                    const float inv_r = __builtin_amdgcn_rcpf(rr[pg]);
#pragma unroll
                    for (int j = 0; j < NTW; ++j) v[pg][j] = pn_bwd4(v[pg][j], yy[pg][j], sd[pg] * inv_n, inv_r, a.slope);
                }
            } else {
#pragma unroll
                for (int pg = 0; pg < PGW; ++pg) {
                    float4 yy[NTW];
                    float s = 0.f;
#pragma unroll
                    for (int j = 0; j < NTW; ++j) {
                        yy[j] = ld4bf(ay_rs, (poff[pg] * N + j * 16 + q * 4) * 2u);
                        s += f4dot(v[pg][j], yy[j]);
                    }
                    const float inv_r = __builtin_amdgcn_rcpf(a.arn[img_pix + poff[pg]]);
                    s = sum_rows4(s) * inv_n;
#pragma unroll
                    for (int j = 0; j < NTW; ++j) st4bf(y_rs, valid[pg] ? (poff[pg] * N + j * 16 + q * 4) * 2u : BF16_OOB, pn_bwd4(v[pg][j], yy[j], s, inv_r, a.slope));
                }
                return;
            }
        }
#pragma unroll
        for (int pg = 0; pg < PGW; ++pg)
#pragma unroll
            for (int j = 0; j < NTW; ++j) st4bf(y_rs, valid[pg] ? (poff[pg] * N + (j0 + j) * 16 + q * 4) * 2u : BF16_OOB, v[pg][j]);
    } else {
        // avg-pool adjoint store: y is (B, 2H, 2W, N), each value * 0.25 to the four pixels of its window
        const int W2 = 2 * W;
        const __amdgpu_buffer_rsrc_t y_rs = rsrc_of(a.y + 4 * img_pix * N, (unsigned)(4 * H * W * N) * 2u);
        const __amdgpu_buffer_rsrc_t ay_rs = rsrc_of(epi == EPI_PN_BWD ? a.ay + 4 * img_pix * N : nullptr, epi == EPI_PN_BWD ? (unsigned)(4 * H * W * N) * 2u : 0u);
#pragma unroll
        for (int pg = 0; pg < PGW; ++pg) {
            const int gy = y0 + prow[pg], gx = x0 + pcol[pg];
            const unsigned o00 = valid[pg] ? (unsigned)(2 * gy * W2 + 2 * gx) : 0u;
#pragma unroll
            for (int j = 0; j < NTW; ++j) v[pg][j] = f4scale(v[pg][j], 0.25f);
            if constexpr (NS) {
                float4 yy[4][NTW];
                float sd[4], rr[4];
                if (epi == EPI_PN_BWD) {
#pragma unroll
                    for (int sub = 0; sub < 4; ++sub) {
                        const unsigned op = o00 + (sub >> 1) * W2 + (sub & 1);
                        float s = 0.f;
#pragma unroll
                        for (int j = 0; j < NTW; ++j) {
                            yy[sub][j] = ld4bf(ay_rs, (op * N + (j0 + j) * 16 + q * 4) * 2u);
                            s += f4dot(v[pg][j], yy[sub][j]);
                        }
                        rr[sub] = a.arn[4 * img_pix + op];
                        sd[sub] = sum_rows4(s);
                    }
                    xsum(sd, 4);
                }
#pragma unroll
                for (int sub = 0; sub < 4; ++sub) {
                    const unsigned op = o00 + (sub >> 1) * W2 + (sub & 1);
                    const float inv_r = epi == EPI_PN_BWD ? __builtin_amdgcn_rcpf(rr[sub]) : 0.f;
#pragma unroll
                    for (int j = 0; j < NTW; ++j)
                        st4bf(y_rs, valid[pg] ? (op * N + (j0 + j) * 16 + q * 4) * 2u : BF16_OOB,
                              epi == EPI_PN_BWD ? pn_bwd4(v[pg][j], yy[sub][j], sd[sub] * inv_n, inv_r, a.slope) : v[pg][j]);
                }
            } else {
#pragma unroll
                for (int sub = 0; sub < 4; ++sub) {
                    const unsigned op = o00 + (sub >> 1) * W2 + (sub & 1);
                    if (epi == EPI_PN_BWD) {
                        float4 yy[NTW];
                        float s = 0.f;
#pragma unroll
                        for (int j = 0; j < NTW; ++j) {
                            yy[j] = ld4bf(ay_rs, (op * N + j * 16 + q * 4) * 2u);
                            s += f4dot(v[pg][j], yy[j]);
                        }
                        const float inv_r = __builtin_amdgcn_rcpf(a.arn[4 * img_pix + op]);
                        s = sum_rows4(s) * inv_n;
#pragma unroll
                        for (int j = 0; j < NTW; ++j) st4bf(y_rs, valid[pg] ? (op * N + j * 16 + q * 4) * 2u : BF16_OOB, pn_bwd4(v[pg][j], yy[j], s, inv_r, a.slope));
                    } else {
#pragma unroll
                        for (int j = 0; j < NTW; ++j) st4bf(y_rs, valid[pg] ? (op * N + j * 16 + q * 4) * 2u : BF16_OOB, v[pg][j]);
                    }
                }
            }
        }
    }
}

template <int K, int N, int PGT, bool NS, bool NARROW>
int launch_bf16(ConvArgsB a, hipStream_t s) {
    constexpr int TW = NARROW ? 16 : 32, TH = PGT / (TW / 16);
    a.tiles_x = ngan::ceil_div(a.W, TW);
    a.tiles_y = ngan::ceil_div(a.H, TH);
    a.n_tiles = a.B * a.tiles_x * a.tiles_y;
    a.band = ngan::ceil_div(a.n_tiles, 8);
    hipLaunchKernelGGL((conv3x3_bf16_kernel<K, N, PGT, NS, NARROW>), dim3(8 * a.band), dim3(256), 0, s, a);
    return ngan::launch_status("ngan_bf16_conv3x3_fwd");
}

// N = 16 / 32: pixel split, PGT = 16 / 8 / 4 (8- / 4- / 2-row tiles of 32 columns) and the narrow 4 x 16 tile;
// N = 64 / 128: output-tile split, PGT = 4 / 2, wide or narrow.  (PGT = 8 -- 128 pixels per workgroup, half the L2 weight traffic of a launch
// with >= 256 such tiles -- was built and measured on BASELINE.json's C2, whose 128 -> 128 layers see 128 images of 16 x 16: iteration
// 2.39 -> 2.47 ms.  Fewer, longer MFMA chains cost more than the weight stream they save; not kept.)
template <int K, int N>
int dispatch_tile(const ConvArgsB& a, int pgt, bool narrow, hipStream_t s) {
    if constexpr (N <= 32) {
        if (narrow) return launch_bf16<K, N, 4, false, true>(a, s);
        if (pgt == 16) return launch_bf16<K, N, 16, false, false>(a, s);
        if (pgt == 8) return launch_bf16<K, N, 8, false, false>(a, s);
        return launch_bf16<K, N, 4, false, false>(a, s);
    } else {
        if (narrow) return pgt == 4 ? launch_bf16<K, N, 4, true, true>(a, s) : launch_bf16<K, N, 2, true, true>(a, s);
        return pgt == 4 ? launch_bf16<K, N, 4, true, false>(a, s) : launch_bf16<K, N, 2, true, false>(a, s);
    }
}

template <int K>
int dispatch_n(const ConvArgsB& a, int N, int pgt, bool narrow, hipStream_t s) {
    switch (N) {
        case 16: return dispatch_tile<K, 16>(a, pgt, narrow, s);
        case 32: return dispatch_tile<K, 32>(a, pgt, narrow, s);
        case 64: return dispatch_tile<K, 64>(a, pgt, narrow, s);
        default: return dispatch_tile<K, 128>(a, pgt, narrow, s);
    }
}

}  // namespace
