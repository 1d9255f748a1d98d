// Winograd F(2x2, 3x3) forward / input-gradient kernel for the 32-channel layers (exact fp32 arithmetic, precision code 4):
// K -> N in {16 -> 32, 32 -> 16, 32 -> 32} on large images with plain input (width a multiple of the 32-pixel tile).
// Replaces conv2d + LeakyReLU + PixelNorm of /root/reference/models.py:252-268 for those layers.
//
// Why a kernel of its own (the 16 -> 16 form lives in conv3x3_tile.hip): v_mfma_f32_16x16x4_f32 runs on the vector-FMA lanes, so
// past the direct form's ~100 TF loop skeleton the only lever is to issue fewer MFMAs, and Y = A^T [ (G g G^T) . (B^T d B) ] A
// needs 16 instead of 36 products per 2x2 output tile and channel pair.  The transformed weights are 16 * K * N floats: 32 KB for
// the mixed shapes, 64 KB for 32 -> 32 -- next to a 25 / 51 KB halo tile that is more than half a CU's LDS, so two workgroups per
// CU cannot each hold a copy.  Instead ONE workgroup of 8 waves owns the CU (two waves per SIMD, the same occupancy as the 16 -> 16
// kernel's two 4-wave workgroups) and the waves divide the work so that nothing but per-pixel scalars crosses between them:
//
//   shape     tile      waves  a wave owns                                     LDS (weights + tile)
//   32 -> 32  8 x 32    8      one row of 16 Winograd tiles x ONE 16-channel   64 + 51 KB
//                              half of the outputs (both input groups)
//   32 -> 16  16 x 32   8      one row of 16 Winograd tiles, all 16 outputs    32 + 92 KB
//   16 -> 32  8 x 32    4      one row of 16 Winograd tiles, both output       32 + 26 KB (two workgroups per CU)
//                              halves
//
// Lane (p, q) of a wave owns Winograd tile p of its row: as the MFMA's B operand it supplies input channels 16 g + 4 q .. + 3 of
// that tile (component s feeds MFMA s), as the D operand it receives output channels 16 mt + 4 q .. + 3 -- B^T d B on the 4x4 input
// patch it reads itself and A^T M A on its own accumulators are lane-local, exactly as in the 16 -> 16 form.  In the 32 -> 32 layout the
// two waves of a tile row repeat the input transform (64 packed additions per 16-channel group against 128 MFMAs) and exchange
// only the per-pixel channel sums PixelNorm / its backward / ToImage need (one float per pixel and wave through LDS).
// Staging, descriptors, padding by whole load instructions and the epilogue arithmetic are those of conv3x3_tile_kernel.
#include "conv3x3_internal.h"

#ifndef NGAN_WINO_STAGGER
#define NGAN_WINO_STAGGER 0
#endif
#ifndef NGAN_WINO_PRIO
#define NGAN_WINO_PRIO 0
#endif

namespace {

template <int KG, int MT, int ROWS, int NWAVES, int RES, int EPI, int OUTMODE>
__global__ __launch_bounds__(NWAVES * 64, 2) void conv3x3_wino_kernel(ConvArgs a, int n_tiles) {
    // RES = 2: the conv input is bilinear_x2(x) (align_corners = False), x at half the resolution.  Nothing is expanded: a 2x2 output
    // tile at even coordinates (Y0, X0) sees hi-res rows Y0-1 .. Y0+2, which are fixed blends of the THREE low-res rows i-1, i, i+1
    // (i = Y0 / 2):  d = E L E^T,  E = [.75 .25 0; .25 .75 0; 0 .75 .25; 0 .25 .75],  so  B^T d B = (B^T E) L (B^T E)^T  is computed
    // straight from the lane's 3x3 low-res patch (9 LDS reads per channel group instead of 16, a 6 x 18 patch in LDS instead of a
    // 10 x 34 tile, no expansion pass and no barrier for it).  The conv's zero padding lives in OUTPUT space: hi-res row -1 / H
    // (column -1 / W) is zero, not a blend -- the rows of E that produce it are switched off by a factor z in {0, 1} folded into the
    // blend coefficients (rows: per wave; columns: per lane).  The bilinear taps' own edge rule (clamped low-res index) is applied
    // by the patch loads.
    static_assert(RES == 0 || (RES == 2 && (EPI == EPI_NONE || EPI == EPI_LRELU_PN) && OUTMODE == 0), "bilinear input: forward convs only");
    constexpr bool UP = RES == 2;
    constexpr int NT = NWAVES * 64, TR = ROWS / 2, NS = NWAVES / TR, MTW = MT / NS;   // tile rows; waves per tile row; n-tiles per wave
    static_assert(NWAVES % TR == 0 && (NS == 1 || NS == 2) && MT % NS == 0, "wave split");
    constexpr int HH_ = ROWS + 2, LP = 40;
    constexpr int PH = ROWS / 2 + 2, PW = 18, PLP = 24;           // low-res patch: rows, columns used, row pitch (a multiple of 8)
    constexpr int PLANE = UP ? PH * PLP * 16 : HH_ * LP * 16, TILE_ELEMS = KG * PLANE;
    constexpr int W_ELEMS = 16 * KG * MT * 256;
    constexpr int K = KG * 16, N = MT * 16, PGW = 4;
    constexpr int N_SRC = KG * PH * PW * 4;                       // patch float4s (bilinear)
    constexpr int NL = UP ? (N_SRC + NT - 1) / NT : KG * HH_ * 128 / NT;   // loads per thread (plain: interior loads, + one for the two halo columns)
    constexpr int NST = NL + 1;
    constexpr int N_HALO = KG * 2 * HH_ * 4;
    static_assert(UP || ((KG * HH_ * 128) % NT == 0 && N_HALO <= NT), "staging layout");
    constexpr unsigned OOB = 0xFFFFFFF0u;
    constexpr bool PNB = EPI == EPI_PN_BWD;
    constexpr int NSUB = (PNB && OUTMODE) ? 4 : 1;                // per-pixel sums a wave contributes per pixel group
    constexpr int XCH_ELEMS = NS > 1 ? 2 * NWAVES * PGW * NSUB * 16 : 4;
    __shared__ __attribute__((aligned(16))) float smem[W_ELEMS + TILE_ELEMS + XCH_ELEMS];
    float* wl = smem;
    float* tile = smem + W_ELEMS;
    float* xch = tile + TILE_ELEMS;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tr = wave % TR, nh = wave / TR;                     // this wave's row of Winograd tiles / its share of the output channels
    const int p = lane & 15, q = lane >> 4;
    const int cb = nh * MTW * 16;                                 // first output channel of this wave

#if NGAN_WINO_PRIO
    if (NWAVES == 8 && wave >= 4) __builtin_amdgcn_s_setprio(1);      // experiment: static priority for the later-dispatched half
#endif

    const TileRun run = tile_run(n_tiles);
    int t = run.t;
    const int t_end = run.t_end;

    // LDS image (plain input): one plane per 16-channel group; in a row, even and odd columns sit in separate halves (a lane reads
    // columns 2p + b: position p + const), quads rotated by the column position as in the direct form -- conflict-free ds_read_b128.
    // Bilinear input: the low-res patch, [group][row][column] with the same quad rotation (a lane reads columns p + j)
    auto lds_slot = [&](int g, int c4, int ty, int tx) {
        const int pos = UP ? tx : (tx >> 1) + (tx & 1) * (LP / 2);
        return g * PLANE + (ty * (UP ? PLP : LP) + pos) * 16 + ((c4 ^ (((pos >> 2) & 1) << 1)) << 2);
    };
    // ---- tile-invariant staging constants: byte offset from the halo origin (y0 - 1, x0 - 1), LDS float index ----
    unsigned s_voff[NST];
    int s_lds[NST];
    int h_bits = 8;                               // halo load: 1 = left column, 2 = right column, 4 = top row, 8 = unused lane
    int s_dyx[UP ? NL : 1];                       // bilinear: patch cell (row in the high half, column in the low half) of load i
    if (UP) {
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int e = tid + i * NT;
            const int c4 = e & 3, pp = (e >> 2) % (PH * PW), g = (e >> 2) / (PH * PW);
            s_dyx[UP ? i : 0] = ((pp / PW) << 16) | (pp % PW);
            s_voff[i] = (unsigned)((g * 16 + c4 * 4) * 4);
            s_lds[i] = lds_slot(g < KG ? g : 0, c4, pp / PW, pp % PW);
        }
        s_voff[NL] = 0; s_lds[NL] = 0;
    } else {
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int e = tid + i * NT;
            const int g = e / (HH_ * 128), r = e % (HH_ * 128);
            const int c4 = r & 3, pix = r >> 2, ty = pix >> 5, tx = (pix & 31) + 1;
            s_voff[i] = (unsigned)(((ty * a.W + tx) * K + g * 16 + c4 * 4) * 4);
            s_lds[i] = lds_slot(g, c4, ty, tx);
        }
        const int c4 = tid & 3, r = (tid >> 2) % HH_, sg = (tid >> 2) / HH_, side = sg & 1, g = sg >> 1;
        const bool used = tid < N_HALO;
        const int tx = side ? 33 : 0;
        s_voff[NL] = used ? (unsigned)(((r * a.W + tx) * K + g * 16 + c4 * 4) * 4) : OOB;
        s_lds[NL] = lds_slot(used ? g : 0, c4, used ? r : 0, tx);
        h_bits = used ? ((side ? 2 : 1) | (r == 0 ? 4 : 0)) : 8;
    }
    int wrd[4];                                    // LDS float index of column 2p + b of a tile row (bilinear: patch column p + b), channel quad q
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const int pos = UP ? p + b : p + (b >> 1) + (b & 1) * (LP / 2);
        wrd[b] = pos * 16 + ((q ^ (((pos >> 2) & 1) << 1)) << 2);
    }
    // ---- tile-invariant epilogue constants: this lane's output byte offsets from the tile origin; pixel group pg = pixel
    // (pg >> 1, pg & 1) of the lane's 2x2 output tile ----
    constexpr int OS = OUTMODE ? 2 : 1;            // the pool-adjoint store writes a 2x2 block of a (2H, 2W) tensor per computed pixel
    const int Wo = OS * a.W;
    unsigned e_voff[PGW], p_voff[PGW];             // byte offset of this lane's first channel / of the pixel in a 1-channel tensor
#pragma unroll
    for (int pg = 0; pg < PGW; ++pg) {
        const int row = 2 * tr + (pg >> 1), col = 2 * p + (pg & 1);
        e_voff[pg] = (unsigned)((((OS * row) * Wo + OS * col) * N + cb + q * 4) * 4);
        p_voff[pg] = (unsigned)(((OS * row) * Wo + OS * col) * 4);
    }
    unsigned t_voff = 0;                           // ToImage: lane group q finishes pixel group q
    if (EPI == EPI_TO_IMAGE) t_voff = (unsigned)(((2 * tr + (q >> 1)) * a.W + 2 * p + (q & 1)) * 4);

    const TileWalk walk(a.tiles_x, a.tiles_y, run.step);
    TileCursor cur_tile = walk.at(t), next_tile = walk.next(cur_tile);      // tile t and tile t + step
    const int lh = a.H >> 1, lw = a.W >> 1;          // low-res extent (bilinear)
    const unsigned img_bytes = (unsigned)((UP ? lh * lw : a.H * a.W) * K) * 4u;
    float4 stg[NST];
    auto issue = [&](const TileCursor& tc) {
        const int b = tc.b, y0 = tc.ty * ROWS, x0 = tc.tx * 32;
        if (UP) {
            // the 6 x 18 low-res patch of the tile, loaded with CLAMPED coordinates: the bilinear taps' edge rule
            const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x + (long)b * lh * lw * K), 0, img_bytes, 0x00020000);
            const int ly0 = (y0 >> 1) - 1, lx0 = (x0 >> 1) - 1;
#pragma unroll
            for (int i = 0; i < NL; ++i) {
                const int ly = min(max(ly0 + (s_dyx[UP ? i : 0] >> 16), 0), lh - 1), lx = min(max(lx0 + (s_dyx[UP ? i : 0] & 0xffff), 0), lw - 1);
                const unsigned off = (tid + i * NT < N_SRC) ? (unsigned)((ly * lw + lx) * K * 4) + s_voff[i] : OOB;
                stg[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0));
            }
            return;
        }
        const int soff = ((y0 - 1) * a.W + (x0 - 1)) * K * 4;                 // negative on the top row / for the first tile
        const char* base = reinterpret_cast<const char*>(a.x + (long)b * a.H * a.W * K) + soff;
        const unsigned nrec = img_bytes - (unsigned)soff;                     // bytes from `base` to the end of the image
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, nrec, 0x00020000);
        // a wave's load i covers 64 consecutive items = half a tile row of one channel group: where that is the TOP halo row of a
        // tile on the image's first row it reads through a descriptor without records, i.e. zeros (a whole-instruction decision)
        // (the choice is made on the descriptor's scalar record count, not between two descriptors: a select of whole descriptors
        // came out of the compiler as a per-lane waterfall loop around every load)
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int hrow = (i * NT + wave * 64) >> 7;              // halo row index over all channel groups: top rows are 0 and HH_
            const bool top = y0 == 0 && (hrow == 0 || (KG == 2 && hrow == HH_));      // (scalar compares: a `% HH_` went through the VALU)
            const __amdgpu_buffer_rsrc_t rsrc_i = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, top ? 0u : nrec, 0x00020000);
            stg[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_i, s_voff[i], 0, 0));
        }
        const int bad = (x0 == 0 ? 1 : 0) | (x0 + 32 >= a.W ? 2 : 0) | (y0 == 0 ? 4 : 0) | 8;
        const unsigned hoff = (h_bits & bad) ? OOB : s_voff[NL];
        stg[NL] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, hoff, 0, 0));
    };
    f32x4 bvec[MTW];
    float4 wimg[MTW];
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt) {
        const float4 b4 = a.bias ? ld4(a.bias + cb + mt * 16 + q * 4) : f4zero();
        bvec[mt] = (f32x4){b4.x, b4.y, b4.z, b4.w};
        wimg[mt] = EPI == EPI_TO_IMAGE ? ld4(a.ay + cb + mt * 16 + q * 4) : f4zero();
    }
    if (t < t_end) issue(cur_tile);
    {   // the packed weights -> LDS, requested BEHIND the first tile's loads and all at once: one memory round trip for both (the copy used to run
        // first, in its own one or two round trips, before the first tile was even requested: ~2 us of every launch)
        constexpr int NWL = (W_ELEMS / 4 + NT - 1) / NT;
        float4 wtmp[NWL];
#pragma unroll
        for (int i = 0; i < NWL; ++i) wtmp[i] = (tid + i * NT < W_ELEMS / 4) ? ld4(a.wp + (long)(tid + i * NT) * 4) : f4zero();
#pragma unroll
        for (int i = 0; i < NWL; ++i)
            if (tid + i * NT < W_ELEMS / 4) st4(wl + (tid + i * NT) * 4, wtmp[i]);
    }
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt) { pin_registers(bvec[mt]); pin_registers(wimg[mt]); }     // (awaited once, here: conv3x3_internal.h)
    const float inv_n = 1.0f / (float)N;
    const f32x2 slope2 = {a.slope, a.slope};
    // partial per-pixel sums of the NS waves that share a tile row: every wave leaves its own in LDS, a workgroup barrier, then adds
    // its partner's (a + b in one wave, b + a in the other: the same float).  `slot` separates the exchanges of one epilogue.
    const int partner = nh ? wave - TR : wave + TR;
    auto exchange = [&](float* vals, int nvals, int slot) {
        if (NS == 1) return;
        float* mine = xch + ((slot * NWAVES + wave) * PGW * NSUB) * 16;
        const float* theirs = xch + ((slot * NWAVES + partner) * PGW * NSUB) * 16;
        if (q == 0)
            for (int j = 0; j < nvals; ++j) mine[j * 16 + p] = vals[j];
        __syncthreads();
        for (int j = 0; j < nvals; ++j) vals[j] += theirs[j * 16 + p];
    };

    while (t < t_end) {
        const int b = cur_tile.b, y0 = cur_tile.ty * ROWS, x0 = cur_tile.tx * 32;
        __syncthreads();   // previous tile's MFMAs have finished reading `tile` (and its exchange buffers have been read)
        if (UP) {
#pragma unroll
            for (int i = 0; i < NL; ++i) {
                pin_registers(stg[i]);          // (awaited by every lane, also those past the end of the patch)
                if (tid + i * NT < N_SRC) st4(&tile[s_lds[i]], stg[i]);
            }
        } else {
#pragma unroll
            for (int i = 0; i < NL; ++i) st4(&tile[s_lds[i]], stg[i]);
            pin_registers(stg[NL]);     // every wave awaits the halo load here (the waves that store nothing would carry it, un-awaited, into the next issue)
            if (tid < N_HALO) st4(&tile[s_lds[NL]], stg[NL]);
        }
        __syncthreads();
#if NGAN_WINO_STAGGER
        if (NS > 1 && nh) __builtin_amdgcn_s_sleep(NGAN_WINO_STAGGER);   // experiment: offset the two waves that share a SIMD and a tile row
#endif
        const int tn = t + run.step;
        if (tn < t_end) issue(next_tile);   // in flight while this tile is computed

        // ---- per-tile scalars of the epilogue (the tile's byte offset is ADDED to the per-lane constants, one v_add per access:
        // a buffer store with an SGPR soffset reads its data registers late, conv3x3_tile_kernel) ----
        const long img = (long)b * a.H * a.W;
        const int pix0 = (OS * y0) * Wo + OS * x0;                       // first output pixel of the tile inside its image
        const unsigned y_soff = (unsigned)pix0 * (N * 4), p_soff = (unsigned)pix0 * 4u;
        const unsigned out_bytes = (unsigned)(OS * a.H * Wo * N) * 4u, px_bytes = (unsigned)(OS * a.H * Wo) * 4u;
        __amdgpu_buffer_rsrc_t y_rsrc, rn_rsrc, ay_rsrc, arn_rsrc;
        if (EPI != EPI_TO_IMAGE || a.y) y_rsrc = __builtin_amdgcn_make_buffer_rsrc(a.y + img * (OS * OS) * N, 0, out_bytes, 0x00020000);
        // (no stored activation -- the inference form of epilogue 3 -- means no stored norm either: a descriptor without records)
        if (EPI == EPI_LRELU_PN || EPI == EPI_TO_IMAGE)
            rn_rsrc = __builtin_amdgcn_make_buffer_rsrc(a.rn + img, 0, (EPI == EPI_LRELU_PN || a.y) ? px_bytes : 0u, 0x00020000);
        if (PNB) {
            ay_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.ay) + img * (OS * OS) * N, 0, out_bytes, 0x00020000);
            arn_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.arn) + img * (OS * OS), 0, px_bytes, 0x00020000);
        }

        // ---- Winograd section: B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1], A^T = [1 1 1 0; 0 1 -1 -1]; transforms on register
        // pairs (v_pk_add_f32 / v_pk_fma_f32), G g G^T done by the packing kernel (layout [position][n-tile][k-group][lane][4]) ----
        const f32x2 m1 = opaque_minus_one();
        f32p bd[KG][4][UP ? 3 : 4];                  // B^T d: rows transformed, columns still in pixel space (bilinear: in low-res space)
        // bilinear: blend coefficients of the row / column transforms  t0 = a0 x0 + b0 x1 - x2/4,  t1 = (x0 + x2)/4 + 1.5 x1,
        // t2 = (x2 - x0)/4,  t3 = x0/4 + b3 x1 + c3 x2  (= B^T E with hi-res row -1 switched off by zt, row H by zb; see the top)
        f32x2 ra0, rb0, rb3, rc3, ca0, cb0, cb3, cc3;
        const f32x2 quarter2 = {0.25f, 0.25f}, c15 = {1.5f, 1.5f};
        if (UP) {
            const float zt = (y0 + 2 * tr > 0) ? 1.f : 0.f, zb = (y0 + 2 * tr + 2 < a.H) ? 1.f : 0.f;
            const float zl = (x0 + 2 * p > 0) ? 1.f : 0.f, zr = (x0 + 2 * p + 2 < a.W) ? 1.f : 0.f;
            const float a0 = 0.75f * zt, b0 = 0.25f * zt - 0.75f, b3 = 0.75f - 0.25f * zb, c3 = -0.75f * zb;
            const float a0c = 0.75f * zl, b0c = 0.25f * zl - 0.75f, b3c = 0.75f - 0.25f * zr, c3c = -0.75f * zr;
            ra0 = (f32x2){a0, a0}; rb0 = (f32x2){b0, b0}; rb3 = (f32x2){b3, b3}; rc3 = (f32x2){c3, c3};
            ca0 = (f32x2){a0c, a0c}; cb0 = (f32x2){b0c, b0c}; cb3 = (f32x2){b3c, b3c}; cc3 = (f32x2){c3c, c3c};
        }
        auto pmul = [](f32p x, f32x2 c) { return f32p{x.l * c, x.h * c}; };
        auto pfma = [](f32p x, f32x2 c, f32p y) { return f32p{__builtin_elementwise_fma(x.l, c, y.l), __builtin_elementwise_fma(x.h, c, y.h)}; };
        // the four transformed values of three samples x0, x1, x2 (one dimension of (B^T E) L (B^T E)^T)
        auto up_t0 = [&](f32p x0, f32p x1, f32p x2, f32x2 ca, f32x2 cb) { return pfma(x0, ca, pfma(x1, cb, pmul(x2, -quarter2))); };
        auto up_t1 = [&](f32p x0, f32p x1, f32p x2) { return pfma(x1, c15, pmul(x0 + x2, quarter2)); };
        auto up_t2 = [&](f32p x0, f32p x2) { return pmul(psub(x2, x0, m1), quarter2); };
        auto up_t3 = [&](f32p x0, f32p x1, f32p x2, f32x2 cb, f32x2 cc) { return pfma(x0, quarter2, pfma(x1, cb, pmul(x2, cc))); };
#pragma unroll
        for (int g = 0; g < KG; ++g) {
            if (UP) {
                const float* prow = tile + g * PLANE + tr * (PLP * 16);        // patch rows tr, tr + 1, tr + 2 = low-res rows i - 1, i, i + 1
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const f32p l0 = pk2(*reinterpret_cast<const f32x4*>(prow + 0 * PLP * 16 + wrd[j])), l1 = pk2(*reinterpret_cast<const f32x4*>(prow + 1 * PLP * 16 + wrd[j]));
                    const f32p l2 = pk2(*reinterpret_cast<const f32x4*>(prow + 2 * PLP * 16 + wrd[j]));
                    bd[g][0][UP ? j : 0] = up_t0(l0, l1, l2, ra0, rb0); bd[g][1][UP ? j : 0] = up_t1(l0, l1, l2);
                    bd[g][2][UP ? j : 0] = up_t2(l0, l2); bd[g][3][UP ? j : 0] = up_t3(l0, l1, l2, rb3, rc3);
                }
                continue;
            }
            const float* trow = tile + g * PLANE + (2 * tr) * (LP * 16);
#pragma unroll
            for (int bb = 0; bb < 4; ++bb) {
                const f32p d0 = pk2(*reinterpret_cast<const f32x4*>(trow + 0 * LP * 16 + wrd[bb])), d1 = pk2(*reinterpret_cast<const f32x4*>(trow + 1 * LP * 16 + wrd[bb]));
                const f32p d2 = pk2(*reinterpret_cast<const f32x4*>(trow + 2 * LP * 16 + wrd[bb])), d3 = pk2(*reinterpret_cast<const f32x4*>(trow + 3 * LP * 16 + wrd[bb]));
                bd[g][0][UP ? 0 : bb] = psub(d0, d2, m1); bd[g][1][UP ? 0 : bb] = d1 + d2; bd[g][2][UP ? 0 : bb] = psub(d2, d1, m1); bd[g][3][UP ? 0 : bb] = psub(d1, d3, m1);
            }
        }
        f32p accp[PGW][MTW];                         // outputs: pixel group (row a, column b) of the 2x2 tile = a * 2 + b
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            f32x4 m[MTW][4];
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
                for (int u = 0; u < 4; ++u) m[mt][u] = (u == 1 && v == 1) ? bvec[mt] : (f32x4){0.f, 0.f, 0.f, 0.f};   // A^T e11 A = all ones: the bias
#pragma unroll
            for (int g = 0; g < KG; ++g) {
                f32p vv[4];                          // (B^T d B)[u][v]
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (UP)
                        vv[u] = v == 0 ? up_t0(bd[g][u][0], bd[g][u][1], bd[g][u][2], ca0, cb0) : v == 1 ? up_t1(bd[g][u][0], bd[g][u][1], bd[g][u][2])
                              : v == 2 ? up_t2(bd[g][u][0], bd[g][u][2]) : up_t3(bd[g][u][0], bd[g][u][1], bd[g][u][2], cb3, cc3);
                    else
                        vv[u] = v == 0 ? psub(bd[g][u][0], bd[g][u][2], m1) : v == 1 ? bd[g][u][1] + bd[g][u][2]
                              : v == 2 ? psub(bd[g][u][2], bd[g][u][1], m1) : psub(bd[g][u][1], bd[g][u][UP ? 0 : 3], m1);
                }
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt) {
                    f32x4 uu[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        uu[u] = *reinterpret_cast<const f32x4*>(&wl[(((u * 4 + v) * MT + nh * MTW + mt) * KG + g) * 256 + lane * 4]);
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            m[mt][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(uu[u][i], i < 2 ? vv[u].l[i & 1] : vv[u].h[i & 1], m[mt][u], 0, 0, 0);
                }
            }
            // A^T M (rows), then the column transform accumulated as v advances: column 0 = t0 + t1 + t2, column 1 = t1 - t2 - t3
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt) {
                const f32p ta0 = pk2(m[mt][0]) + pk2(m[mt][1]) + pk2(m[mt][2]);
                const f32p ta1 = psub(psub(pk2(m[mt][1]), pk2(m[mt][2]), m1), pk2(m[mt][3]), m1);
                if (v == 0) { accp[0][mt] = ta0; accp[2][mt] = ta1; }
                else if (v == 1) { accp[0][mt] = accp[0][mt] + ta0; accp[2][mt] = accp[2][mt] + ta1; accp[1][mt] = ta0; accp[3][mt] = ta1; }
                else if (v == 2) { accp[0][mt] = accp[0][mt] + ta0; accp[2][mt] = accp[2][mt] + ta1; accp[1][mt] = psub(accp[1][mt], ta0, m1); accp[3][mt] = psub(accp[3][mt], ta1, m1); }
                else { accp[1][mt] = psub(accp[1][mt], ta0, m1); accp[3][mt] = psub(accp[3][mt], ta1, m1); }
            }
        }

        // ---- epilogue: lane holds channels cb + 16 mt + 4q .. + 3 of its four pixels ----
        f32x2 lo[PGW][MTW], hi[PGW][MTW];
#pragma unroll
        for (int pg = 0; pg < PGW; ++pg)
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt) { lo[pg][mt] = accp[pg][mt].l; hi[pg][mt] = accp[pg][mt].h; }

        // PixelNorm-backward operands (same shape as the output): every load of the tile before its first store, awaited with
        // vmcnt(0) (loads and stores retire out of order with each other under one counter: conv3x3_tile_kernel)
        float4 yy[PNB ? PGW * NSUB : 1][MTW];
        float rr[PNB ? PGW * NSUB : 1];
        if (PNB) {
            const unsigned row1 = (unsigned)(Wo * N * 4), prow1 = (unsigned)(Wo * 4);
#pragma unroll
            for (int pg = 0; pg < PGW; ++pg)
#pragma unroll
                for (int sub = 0; sub < NSUB; ++sub) {
                    const unsigned eo = e_voff[pg] + y_soff + ((sub >> 1) ? row1 : 0u) + (sub & 1) * (N * 4);
                    const unsigned po = p_voff[pg] + p_soff + ((sub >> 1) ? prow1 : 0u) + (sub & 1) * 4;
#pragma unroll
                    for (int mt = 0; mt < MTW; ++mt)
                        yy[PNB ? pg * NSUB + sub : 0][mt] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(ay_rsrc, eo + mt * 64, 0, 0));
                    rr[PNB ? pg * NSUB + sub : 0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(arn_rsrc, po, 0, 0));
                }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        float timg = 0.f;
        if (EPI == EPI_LRELU_PN || EPI == EPI_TO_IMAGE) {
            float ss[PGW];
#pragma unroll
            for (int pg = 0; pg < PGW; ++pg) {
                f32x2 sq = {0.f, 0.f};
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt) {
                    const f32x2 sl = lo[pg][mt] * slope2, sh = hi[pg][mt] * slope2;      // LeakyReLU, 0 <= slope <= 1: one v_max each
                    lo[pg][mt] = (f32x2){vmax1(lo[pg][mt].x, sl.x), vmax1(lo[pg][mt].y, sl.y)};
                    hi[pg][mt] = (f32x2){vmax1(hi[pg][mt].x, sh.x), vmax1(hi[pg][mt].y, sh.y)};
                    sq = mt == 0 ? lo[pg][mt] * lo[pg][mt] : __builtin_elementwise_fma(lo[pg][mt], lo[pg][mt], sq);
                    sq = __builtin_elementwise_fma(hi[pg][mt], hi[pg][mt], sq);
                }
                ss[pg] = sum_rows4(sq.x + sq.y);
            }
            exchange(ss, PGW, 0);
            float dd[PGW];
#pragma unroll
            for (int pg = 0; pg < PGW; ++pg) {
                const float m = ss[pg] * inv_n + a.eps;
                const float inv = __builtin_amdgcn_rsqf(m);
                const f32x2 inv2 = {inv, inv};
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt) { lo[pg][mt] *= inv2; hi[pg][mt] *= inv2; }
                // the norm: one lane per pixel stores (of one wave), the others' offset is out of range
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, m * inv), rn_rsrc, (q == 0 && nh == 0) ? p_voff[pg] + p_soff : OOB, 0, 0);
                if (EPI == EPI_TO_IMAGE) {
                    f32x2 d2 = {0.f, 0.f};
#pragma unroll
                    for (int mt = 0; mt < MTW; ++mt) {
                        d2 = __builtin_elementwise_fma(lo[pg][mt], (f32x2){wimg[mt].x, wimg[mt].y}, d2);
                        d2 = __builtin_elementwise_fma(hi[pg][mt], (f32x2){wimg[mt].z, wimg[mt].w}, d2);
                    }
                    dd[pg] = sum_rows4(d2.x + d2.y);
                }
            }
            if (EPI == EPI_LRELU_PN && a.aout) {
                // pooled side output: y averaged over the lane's 2x2 block -> (B, H/2, W/2, N), what the next block's avg-pooled conv
                // would otherwise make with a pass of its own (ops._pool_first); (a + b) + (c + d) as in ngan_pool2_fwd: same bits
                const int hw2 = (a.H >> 1) * (a.W >> 1);
                const __amdgpu_buffer_rsrc_t p_rsrc = __builtin_amdgcn_make_buffer_rsrc(a.aout + (long)b * hw2 * N, 0, (unsigned)(hw2 * N) * 4u, 0x00020000);
                const unsigned poff = (unsigned)(((((y0 >> 1) + tr) * (a.W >> 1) + (x0 >> 1) + p) * N + cb + q * 4) * 4);
                const f32x2 quarter = {0.25f, 0.25f};
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt) {
                    const f32x2 l = ((lo[0][mt] + lo[1][mt]) + (lo[2][mt] + lo[3][mt])) * quarter, h = ((hi[0][mt] + hi[1][mt]) + (hi[2][mt] + hi[3][mt])) * quarter;
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, make_float4(l.x, l.y, h.x, h.y)), p_rsrc, poff + mt * 64, 0, 0);
                }
            }
            if (EPI == EPI_TO_IMAGE) {
                exchange(dd, PGW, 1);
#pragma unroll
                for (int pg = 0; pg < PGW; ++pg)
                    if (q == pg) timg = dd[pg];     // all four q-lanes hold pixel group pg's sum; lane group q keeps the one it will finish
            }
        }
        if (PNB) {
            // backward of the LeakyReLU -> PixelNorm that produced this layer's input, applied to the gradient just computed; with the
            // pool-adjoint store (OUTMODE 1) the value * 0.25 goes to four pixels, each with its own operands
            if (OUTMODE) {
                const f32x2 quarter = {0.25f, 0.25f};
#pragma unroll
                for (int pg = 0; pg < PGW; ++pg)
#pragma unroll
                    for (int mt = 0; mt < MTW; ++mt) { lo[pg][mt] *= quarter; hi[pg][mt] *= quarter; }
            }
            float s[PGW * NSUB];
#pragma unroll
            for (int pg = 0; pg < PGW; ++pg)
#pragma unroll
                for (int sub = 0; sub < NSUB; ++sub) {
                    float acc = 0.f;
#pragma unroll
                    for (int mt = 0; mt < MTW; ++mt) {
                        const float4 y4 = yy[PNB ? pg * NSUB + sub : 0][mt];
                        acc += lo[pg][mt].x * y4.x + lo[pg][mt].y * y4.y + hi[pg][mt].x * y4.z + hi[pg][mt].y * y4.w;
                    }
                    s[pg * NSUB + sub] = sum_rows4(acc);
                }
            exchange(s, PGW * NSUB, 0);
            const unsigned row1 = (unsigned)(Wo * N * 4);
#pragma unroll
            for (int pg = 0; pg < PGW; ++pg)
#pragma unroll
                for (int sub = 0; sub < NSUB; ++sub) {
                    const float sm = s[pg * NSUB + sub] * inv_n;
                    const float inv_r = 1.0f / rr[PNB ? pg * NSUB + sub : 0];
                    const unsigned eo = e_voff[pg] + y_soff + ((sub >> 1) ? row1 : 0u) + (sub & 1) * (N * 4);
#pragma unroll
                    for (int mt = 0; mt < MTW; ++mt) {
                        const float4 o = pn_bwd4(make_float4(lo[pg][mt].x, lo[pg][mt].y, hi[pg][mt].x, hi[pg][mt].y), yy[PNB ? pg * NSUB + sub : 0][mt], sm, inv_r, a.slope);
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), y_rsrc, eo + mt * 64, 0, 0);
                    }
                }
        } else if (OUTMODE == 0) {
            if (EPI != EPI_TO_IMAGE || a.y) {
#pragma unroll
                for (int pg = 0; pg < PGW; ++pg)
#pragma unroll
                    for (int mt = 0; mt < MTW; ++mt) {
                        const u32x4 v = __builtin_bit_cast(u32x4, make_float4(lo[pg][mt].x, lo[pg][mt].y, hi[pg][mt].x, hi[pg][mt].y));
                        __builtin_amdgcn_raw_buffer_store_b128(v, y_rsrc, e_voff[pg] + y_soff + mt * 64, 0, 0);
                    }
            }
        } else {
            // pool-adjoint store without an epilogue: the value * 0.25 to the 2x2 block (2gy + i, 2gx + j)
            const unsigned row1 = (unsigned)(Wo * N * 4);
            const f32x2 quarter = {0.25f, 0.25f};
#pragma unroll
            for (int pg = 0; pg < PGW; ++pg)
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt) {
                    const f32x2 l = lo[pg][mt] * quarter, h = hi[pg][mt] * quarter;
                    const u32x4 v = __builtin_bit_cast(u32x4, make_float4(l.x, l.y, h.x, h.y));
#pragma unroll
                    for (int sub = 0; sub < 4; ++sub)
                        __builtin_amdgcn_raw_buffer_store_b128(v, y_rsrc, e_voff[pg] + y_soff + ((sub >> 1) ? row1 : 0u) + (sub & 1) * (N * 4) + mt * 64, 0, 0);
                }
        }
        if (EPI == EPI_TO_IMAGE) {
            // one tanh per lane instead of four: lane group q finishes pixel group q (same tanhf as the standalone ToImage kernel)
            const float tv = tanhf(timg);
            const __amdgpu_buffer_rsrc_t t_rsrc = __builtin_amdgcn_make_buffer_rsrc(a.aout + img, 0, px_bytes, 0x00020000);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, tv), t_rsrc, nh == 0 ? t_voff + p_soff : OOB, 0, 0);
        }
        t = tn;
        cur_tile = next_tile;
        next_tile = walk.next(next_tile);
    }
}

template <int KG, int MT, int ROWS, int NWAVES, int RES, int EPI, int OUTMODE>
int launch_wino(ConvArgs a, hipStream_t s) {
    a.tiles_x = a.W / 32;
    a.tiles_y = ngan::ceil_div(a.H, ROWS);
    const int n_tiles = a.B * a.tiles_x * a.tiles_y;
    static int per_cu = 0;
    if (per_cu == 0) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, conv3x3_wino_kernel<KG, MT, ROWS, NWAVES, RES, EPI, OUTMODE>, NWAVES * 64, 0) != hipSuccess || n < 1) n = 1;
        per_cu = n > 4 ? 4 : n;
    }
    const int grid = persistent_grid(n_tiles, 256 * per_cu);
    hipLaunchKernelGGL((conv3x3_wino_kernel<KG, MT, ROWS, NWAVES, RES, EPI, OUTMODE>), dim3(grid), dim3(NWAVES * 64), 0, s, a, n_tiles);
    return ngan::launch_status("ngan_conv3x3_fwd(winograd)");
}

template <int KG, int MT, int ROWS, int NWAVES>
int dispatch_wino(const ConvArgs& a, int res, int epi, int outmode, hipStream_t s) {
    if (res == NGAN_RESAMPLE_UP2) return epi ? launch_wino<KG, MT, ROWS, NWAVES, 2, EPI_LRELU_PN, 0>(a, s) : launch_wino<KG, MT, ROWS, NWAVES, 2, EPI_NONE, 0>(a, s);
    if constexpr (KG * MT > 1) {      // (the plain 16 -> 16 form is conv3x3_tile_kernel's)
        if (epi == EPI_PN_BWD) return outmode == 1 ? launch_wino<KG, MT, ROWS, NWAVES, 0, EPI_PN_BWD, 1>(a, s) : launch_wino<KG, MT, ROWS, NWAVES, 0, EPI_PN_BWD, 0>(a, s);
        if (epi == EPI_TO_IMAGE) return launch_wino<KG, MT, ROWS, NWAVES, 0, EPI_TO_IMAGE, 0>(a, s);
        if (outmode == 1) return launch_wino<KG, MT, ROWS, NWAVES, 0, EPI_NONE, 1>(a, s);
        return epi ? launch_wino<KG, MT, ROWS, NWAVES, 0, EPI_LRELU_PN, 0>(a, s) : launch_wino<KG, MT, ROWS, NWAVES, 0, EPI_NONE, 0>(a, s);
    }
    return NGAN_ERR_ARG;
}

}  // namespace

int ngan::conv3x3_wino_tile_rows(int mtw, int kg) { return (kg == 2 && mtw == 1) ? 16 : 8; }

// a.W % 32 == 0; plain input: (mtw, kg) in {(2, 2), (1, 2), (2, 1)}; bilinear input (epilogues 0 / 1): also (1, 1)  (the caller checks)
int ngan::conv3x3_wino_launch(const ConvArgs& a, int mtw, int kg, int resample, int epilogue, int out_mode, hipStream_t s) {
    if (kg == 2 && mtw == 2) return dispatch_wino<2, 2, 8, 8>(a, resample, epilogue, out_mode, s);
    if (kg == 2) return dispatch_wino<2, 1, 16, 8>(a, resample, epilogue, out_mode, s);
    if (mtw == 2) return dispatch_wino<1, 2, 8, 4>(a, resample, epilogue, out_mode, s);
    return dispatch_wino<1, 1, 8, 4>(a, resample, epilogue, out_mode, s);
}
