#include "conv3x3_internal.h"

namespace {

// ---------------------------------------------------------------------------------------------------------
// conv3x3_tile_kernel: the persistent kernel for plain (not resampled) input on images whose width is a multiple of the 32-pixel
// tile, rebuilt around one measurement (round 2's in-kernel clock probe and one-phase-compiled-out builds, profiles/r02_clock_probe_and_phase_experiments.txt): v_mfma_f32_16x16x4_f32 runs on the vector
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
#ifndef NGAN_TILE_PRE_WINO
#define NGAN_TILE_PRE_WINO 1
#endif
#ifndef NGAN_TILE_DOUBLE_BUFFER
#define NGAN_TILE_DOUBLE_BUFFER 0      // measured, round 3: slower (below)
#endif

// Phase-timer build only (`make phases`; tools/wgrad_phases.py --op fwd): shader-clock stamps, summed over the sampled waves (one workgroup in
// eight).  0 = the tile's first barrier, 1 = waiting for the tile's loads + LDS writes, 2 = second barrier, 3 = issuing the next tile's loads +
// the epilogue's scalars / operand requests, 4 = transforms + MFMAs, 5 = epilogue arithmetic + stores, 6 = number of waves
#ifdef NGAN_DIAG_PHASES
__device__ unsigned long long tile_phase_ctr[11];      // 7 = earliest loop entry, 8 = latest exit, 9 = sum of entries, 10 = sum of exits (s_memrealtime, 100 MHz)
#define TPH_INIT unsigned long long ph_[6] = {0, 0, 0, 0, 0, 0}, last_ = __builtin_readcyclecounter(), t0_ = __builtin_amdgcn_s_memrealtime()
#define TPH(i) { const unsigned long long now_ = __builtin_readcyclecounter(); ph_[i] += now_ - last_; last_ = now_; }
#define TPH_FLUSH if (lane == 0 && (blockIdx.x & 7) == 0) { const unsigned long long t1_ = __builtin_amdgcn_s_memrealtime(); \
    for (int i_ = 0; i_ < 6; ++i_) atomicAdd(&tile_phase_ctr[i_], ph_[i_]); atomicAdd(&tile_phase_ctr[6], 1ull); \
    atomicMin(&tile_phase_ctr[7], t0_); atomicMax(&tile_phase_ctr[8], t1_); atomicAdd(&tile_phase_ctr[9], t0_); atomicAdd(&tile_phase_ctr[10], t1_); }
#else
#define TPH_INIT
#define TPH(i)
#define TPH_FLUSH
#endif

template <int MTW, int KG, int EPI, int OUTMODE, int PREC>
__global__ __launch_bounds__(256, PREC == 2 ? NGAN_WINO16_WPE : (MTW * KG == 1) ? (PREC ? 3 : 4) : 2) void conv3x3_tile_kernel(ConvArgs a, int n_tiles) {
    // PREC: 0 exact fp32 (direct), 1 split bf16, 2 exact fp32 by Winograd F(2x2, 3x3) (16 -> 16 only; see the MFMA section),
    // 3 (diagnostic build only: ngan_diag_conv3x3_bf16x6 below) THREE-way split bf16: x = hi + mid + lo covers fp32's 24 significand bits, six of
    // the nine cross products per fp32 product (hh, hm, mh, hl, lh, mm; the dropped three are <= 2^-24 relative), fp32 accumulation
    constexpr bool BF = PREC == 1 || PREC == 3, BF6 = PREC == 3, WINO = PREC == 2;
    constexpr int NPART = BF6 ? 3 : 2;
    static_assert(!WINO || (MTW == 1 && KG == 1), "the Winograd form is built for the 16 -> 16 layers");
    static_assert(!BF6 || (MTW == 1 && KG == 1), "the three-way split is an experiment on the 16 -> 16 layers");
    constexpr int THc = persist_tile_h(MTW, KG, 0), PGW = THc / 2, RPW = THc / 4;
    constexpr int HH_ = THc + 2, LP = 40;
    constexpr int NSTEP = KG == 1 ? 5 : 9;
    constexpr int W_ELEMS = BF ? NSTEP * MTW * NPART * 256 : (WINO ? 16 * 256 : 9 * KG * MTW * 256), PLANE = HH_ * LP * 16, TILE_ELEMS = (BF6 ? 2 : KG) * PLANE;
    constexpr int LPG = HH_ / 2;                 // interior loads per 16-channel group: HH_ rows x 32 columns x 4 quads / 256 threads
    constexpr int NL = KG * LPG, NST = NL + 1;   // + one load for the two halo columns
    constexpr int N_HALO = KG * 2 * HH_ * 4;     // its active lanes: (group, side, row, quad)
    constexpr int K = KG * 16, N = MTW * 16;
    constexpr unsigned OOB = 0xFFFFFFF0u;
    static_assert(HH_ % 2 == 0 && N_HALO <= 256, "staging layout");
    // Build-time option for the Winograd instances: TWO tile buffers.  The next tile is staged into the other buffer while this one
    // is being read, so a tile costs one workgroup barrier instead of two and no wave waits for the others' staging before its MFMAs
    // (16 + 2 x 25.6 KB: still two workgroups per CU).  Correct (the whole op suite passes with it) and NOT faster: 16 -> 16 at
    // 512 x 512, batch 16: 145 -> 154 us plain, 165 -> 167 us with LeakyReLU -> PixelNorm, 75.9 -> 80.4 us with the pool-adjoint store;
    // iteration 7.17 -> 7.22 ms.  The second barrier was not what the waves wait for; off.
    constexpr bool DB = WINO && NGAN_TILE_DOUBLE_BUFFER;
    __shared__ __attribute__((aligned(16))) float smem[W_ELEMS + (DB ? 2 : 1) * TILE_ELEMS];
    float* wl = smem;
    float* tile = smem + W_ELEMS;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int p = lane & 15, q = lane >> 4;


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

    const TileWalk walk(a.tiles_x, a.tiles_y, run.step);
    TileCursor cur_tile = walk.at(t), next_tile = walk.next(cur_tile);      // tile t and tile t + step
    const unsigned img_bytes = (unsigned)(a.H * a.W * K) * 4u;
    float4 stg[NST];
    auto issue = [&](const TileCursor& tc) {
        const int b = tc.b, y0 = tc.ty * THc, x0 = tc.tx * 32;
        const int soff = ((y0 - 1) * a.W + (x0 - 1)) * K * 4;                 // negative on the top row / for the first tile
        const char* base = reinterpret_cast<const char*>(a.x + (long)b * a.H * a.W * K) + soff;
        const unsigned nrec = img_bytes - (unsigned)soff;                     // bytes from `base` to the end of the image
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, nrec, 0x00020000);
        // the top halo row is the first interior load of waves 0 and 1 (+ flagged lanes of the halo load): above the image it reads
        // through a descriptor with no records, i.e. zeros
        const __amdgpu_buffer_rsrc_t rsrc_top = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, (y0 == 0 && wave < 2) ? 0u : nrec, 0x00020000);
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            stg[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128((i % LPG == 0) ? rsrc_top : rsrc, s_voff[i], 0, 0));
        }
        const int bad = (x0 == 0 ? 1 : 0) | (x0 + 32 >= a.W ? 2 : 0) | (y0 == 0 ? 4 : 0) | 8;
        const unsigned hoff = (h_bits & bad) ? OOB : s_voff[NL];
        stg[NL] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, hoff, 0, 0));
    };
    f32x4 bvec[MTW];
    float4 wimg[MTW];
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt) {
        const float4 b4 = a.bias ? ld4(a.bias + mt * 16 + q * 4) : f4zero();
        bvec[mt] = (f32x4){b4.x, b4.y, b4.z, b4.w};
        wimg[mt] = EPI == EPI_TO_IMAGE ? ld4(a.ay + mt * 16 + q * 4) : f4zero();
    }
    if (t < t_end) issue(cur_tile);
    {   // the packed weights -> LDS, requested BEHIND the first tile's loads and all at once: one memory round trip for both (the copy used to run
        // first, in its own one or two round trips, before the first tile was even requested: ~2 us of every launch)
        constexpr int NWL = (W_ELEMS / 4 + 256 - 1) / 256;
        float4 wtmp[NWL];
#pragma unroll
        for (int i = 0; i < NWL; ++i) wtmp[i] = (tid + i * 256 < W_ELEMS / 4) ? ld4(a.wp + (long)(tid + i * 256) * 4) : f4zero();
#pragma unroll
        for (int i = 0; i < NWL; ++i)
            if (tid + i * 256 < W_ELEMS / 4) st4(wl + (tid + i * 256) * 4, wtmp[i]);
    }
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt) { pin_registers(bvec[mt]); pin_registers(wimg[mt]); }     // (awaited once, here: conv3x3_internal.h)
    const float inv_n = 1.0f / (float)N;
    const f32x2 slope2 = {a.slope, a.slope};

    auto stage = [&](float* buf) {                 // the loaded tile (stg) -> LDS image
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            if (BF6) st_split3<PLANE>(buf, s_lds[i], stg[i]);
            else if (BF) st_split<KG, PLANE>(buf, s_lds[i], stg[i]);
            else st4(&buf[s_lds[i]], stg[i]);
        }
        pin_registers(stg[NL]);     // every wave awaits the halo load here (the waves that store nothing would carry it, un-awaited, into the next issue)
        if (tid < N_HALO) {
            if (BF6) st_split3<PLANE>(buf, s_lds[NL], stg[NL]);
            else if (BF) st_split<KG, PLANE>(buf, s_lds[NL], stg[NL]);
            else st4(&buf[s_lds[NL]], stg[NL]);
        }
    };
    if (DB && t < t_end) {                         // double-buffered: the first tile is staged here, the second one's loads go out
        stage(tile);
        __syncthreads();
        if (t + run.step < t_end) issue(next_tile);
    }
    float* const tile0 = tile;
    int cur = 0;
    TPH_INIT;
    while (t < t_end) {
        const int b = cur_tile.b, y0 = cur_tile.ty * THc, x0 = cur_tile.tx * 32;
        const int tn = t + run.step;
        if (DB) {
            // stg holds tile tn (loaded during the previous tile): into the buffer nobody reads now; then the loads of the tile after it
            tile = tile0 + cur * TILE_ELEMS;
            if (tn < t_end) stage(tile0 + (cur ^ 1) * TILE_ELEMS);
            if (tn + run.step < t_end) issue(walk.next(next_tile));
        } else {
            __syncthreads();   // previous tile's MFMAs have finished reading `tile`
            TPH(0);
            stage(tile);
            TPH(1);
            __syncthreads();
            TPH(2);
            if (tn < t_end) issue(next_tile);   // in flight while this tile is computed
        }

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
        // (the 16 -> 16 direct / split-bf16 instances run 3 - 4 workgroups per CU and have no registers for it; the Winograd instance
        // has 2 per CU and ~170 of 256 registers in use)
        constexpr bool PRE = PNB && (MTW * KG > 1 || (WINO && NGAN_TILE_PRE_WINO)) && NGAN_TILE_PRE;
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

        TPH(3);
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
        } else if (BF6) {
            // weights [step][part hi / mid / lo][lane][8]; tile: hi and mid share the pixel's 64-byte slot (the two-way split's layout), lo
            // is the same slot of a second plane.  Smallest products first.
#pragma unroll
            for (int st = 0; st < NSTEP; ++st) {
                bf16x8 xh[PGW], xm[PGW], xl[PGW];
#pragma unroll
                for (int pg = 0; pg < PGW; ++pg) {
                    const int row = wave * RPW + (pg >> 1);
                    const int base = (row * LP + (pg & 1) * 16) * 16 + rs[st];
                    xh[pg] = *reinterpret_cast<const bf16x8*>(&tile[base]);
                    xm[pg] = *reinterpret_cast<const bf16x8*>(&tile[base ^ 8]);
                    xl[pg] = *reinterpret_cast<const bf16x8*>(&tile[base + PLANE]);
                }
                const bf16x8 wh = *reinterpret_cast<const bf16x8*>(&wl[(st * 3 + 0) * 256 + lane * 4]);
                const bf16x8 wm = *reinterpret_cast<const bf16x8*>(&wl[(st * 3 + 1) * 256 + lane * 4]);
                const bf16x8 wlo = *reinterpret_cast<const bf16x8*>(&wl[(st * 3 + 2) * 256 + lane * 4]);
#pragma unroll
                for (int pg = 0; pg < PGW; ++pg) acc[pg][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wlo, xh[pg], acc[pg][0], 0, 0, 0);
#pragma unroll
                for (int pg = 0; pg < PGW; ++pg) acc[pg][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xl[pg], acc[pg][0], 0, 0, 0);
#pragma unroll
                for (int pg = 0; pg < PGW; ++pg) acc[pg][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, xm[pg], acc[pg][0], 0, 0, 0);
#pragma unroll
                for (int pg = 0; pg < PGW; ++pg) acc[pg][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, xh[pg], acc[pg][0], 0, 0, 0);
#pragma unroll
                for (int pg = 0; pg < PGW; ++pg) acc[pg][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xm[pg], acc[pg][0], 0, 0, 0);
#pragma unroll
                for (int pg = 0; pg < PGW; ++pg) acc[pg][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xh[pg], acc[pg][0], 0, 0, 0);
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
        TPH(4);
        // ---- epilogue ----
        if (PNB && !PRE) load_pn_operands();
        // PixelNorm-backward operands: wait for everything in flight (the operand loads and the next tile, issued a tile's worth of
        // MFMAs ago) BEFORE the first store goes out.  Once stores are in flight, loads and stores of gfx9 retire out of order with each other
        // under one counter, and the counted waits the compiler emits for the older loads returned early: wrong last dwords in the
        // last lanes of a pixel group (tools/dbg_epi2.py; without the prefetch the kernel is bit-identical to conv3x3_persist_kernel)
        if (PNB) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        float timg = 0.f;
        // pooled side output (Winograd form, epilogue 1, a.aout given): the lane's four pixel groups ARE one 2x2 pooling window
        constexpr bool POOL_OUT = WINO && EPI == EPI_LRELU_PN && OUTMODE == 0;
        f32x2 plo[POOL_OUT ? 2 : 1][MTW], phi[POOL_OUT ? 2 : 1][MTW];
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
            if (POOL_OUT) {
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt) {         // (a + b) + (c + d), the association of ngan_pool2_fwd
                    plo[POOL_OUT ? pg >> 1 : 0][mt] = (pg & 1) ? plo[POOL_OUT ? pg >> 1 : 0][mt] + lo[mt] : lo[mt];
                    phi[POOL_OUT ? pg >> 1 : 0][mt] = (pg & 1) ? phi[POOL_OUT ? pg >> 1 : 0][mt] + hi[mt] : hi[mt];
                }
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
#ifdef NGAN_DIAG
                // Timing experiment (tools/gp_fusion_probe.py, diagnostic build only): what a create_graph pass would need from a fused
                // input-gradient + PixelNorm-backward kernel -- the gradient BEFORE the PixelNorm backward as a second output (aux_out)
                if (OUTMODE == 0 && a.aout) {
                    const __amdgpu_buffer_rsrc_t pre_rsrc = __builtin_amdgcn_make_buffer_rsrc(a.aout + img * N, 0, out_bytes, 0x00020000);
#pragma unroll
                    for (int mt = 0; mt < MTW; ++mt)
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, make_float4(lo[mt].x, lo[mt].y, hi[mt].x, hi[mt].y)), pre_rsrc,
                                                               e_voff[pg] + y_soff + mt * 64, 0, 0);
                }
#endif
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
                        __builtin_amdgcn_raw_buffer_store_b128(v, y_rsrc, e_voff[pg] + y_soff + mt * 64, 0, 0);
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
        if (POOL_OUT && a.aout) {
            // y averaged over the lane's 2x2 block -> (B, H/2, W/2, N): what the next block's avg-pooled conv would otherwise make with
            // a pass of its own (ops._pool_first); same association as ngan_pool2_fwd, so the bits are the same
            const int hw2 = (a.H >> 1) * (a.W >> 1);
            const __amdgpu_buffer_rsrc_t p_rsrc = __builtin_amdgcn_make_buffer_rsrc(a.aout + (long)b * hw2 * N, 0, (unsigned)(hw2 * N) * 4u, 0x00020000);
            const unsigned poff = (unsigned)(((((y0 >> 1) + wave) * (a.W >> 1) + (x0 >> 1) + p) * N + q * 4) * 4);
            const f32x2 quarter = {0.25f, 0.25f};
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt) {
                const f32x2 l = (plo[0][mt] + plo[POOL_OUT ? 1 : 0][mt]) * quarter, h = (phi[0][mt] + phi[POOL_OUT ? 1 : 0][mt]) * quarter;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, make_float4(l.x, l.y, h.x, h.y)), p_rsrc, poff + mt * 64, 0, 0);
            }
        }
        if (EPI == EPI_TO_IMAGE) {
            // one tanh per lane instead of four: lane group q finishes pixel group q (same tanhf as the standalone ToImage kernel)
            const float tv = tanhf(timg);
            const __amdgpu_buffer_rsrc_t t_rsrc = __builtin_amdgcn_make_buffer_rsrc(a.aout + img, 0, px_bytes, 0x00020000);
            if (q < PGW) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, tv), t_rsrc, t_voff + p_soff, 0, 0);
        }
        if (DB) {
            __syncthreads();   // every wave has finished reading this tile's buffer and writing the next tile's
            cur ^= 1;
        }
        TPH(5);
        t = tn;
        cur_tile = next_tile;
        next_tile = walk.next(next_tile);
    }
    TPH_FLUSH;
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

template <int MTW, int KG, int PREC>
int dispatch_tile2(const ConvArgs& a, int epi, int outmode, hipStream_t s) {
    if (epi == EPI_PN_BWD) return outmode == 1 ? launch_tile<MTW, KG, EPI_PN_BWD, 1, PREC>(a, s) : launch_tile<MTW, KG, EPI_PN_BWD, 0, PREC>(a, s);
    if (epi == EPI_TO_IMAGE) return launch_tile<MTW, KG, EPI_TO_IMAGE, 0, PREC>(a, s);
    if (outmode == 1) return launch_tile<MTW, KG, 0, 1, PREC>(a, s);
    return epi ? launch_tile<MTW, KG, 1, 0, PREC>(a, s) : launch_tile<MTW, KG, 0, 0, PREC>(a, s);
}

template <int MTW, int KG>
int dispatch_tile_prec(const ConvArgs& a, int epi, int outmode, int tprec, hipStream_t s) {
    if (tprec == 2) {
        if constexpr (MTW == 1 && KG == 1) return dispatch_tile2<1, 1, 2>(a, epi, outmode, s);
        else return NGAN_ERR_ARG;
    }
    return tprec ? dispatch_tile2<MTW, KG, 1>(a, epi, outmode, s) : dispatch_tile2<MTW, KG, 0>(a, epi, outmode, s);
}

}  // namespace

// plain input, a.W % 32 == 0 (the caller checks); tprec: the kernel's PREC template parameter
int ngan::conv3x3_tile_launch(const ConvArgs& a, int mtw, int kg, int epilogue, int out_mode, int tprec, hipStream_t s) {
    if (mtw == 1) return kg == 1 ? dispatch_tile_prec<1, 1>(a, epilogue, out_mode, tprec, s) : dispatch_tile_prec<1, 2>(a, epilogue, out_mode, tprec, s);
    return kg == 1 ? dispatch_tile_prec<2, 1>(a, epilogue, out_mode, tprec, s) : dispatch_tile_prec<2, 2>(a, epilogue, out_mode, tprec, s);
}

#ifdef NGAN_DIAG
// Diagnostic build only (not in include/ngan.h; tools/bf16x6_probe.py): the three-way split-bf16 form of the 16 -> 16 layer on plain input,
// whole 32-pixel tiles -- the experiment behind the round-2 review's ruling on a "bf16x6" mode.  Packs w (OIHW fp32, 16 x 16 x 3 x 3) * scale
// into `packed` (5 steps x 3 parts x 512 bf16 = 15 KB) and runs conv3x3_tile_kernel<1, 1, epilogue, 0, 3>; epilogue 0 or 1.
namespace {
__global__ void pack_weights_bf16x6_kernel(const float* __restrict__ w, __bf16* __restrict__ packed, float scale) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;                 // [step 5][part 3][lane 64][8]
    if (idx >= 5 * 3 * 512) return;
    const int j = idx & 7, lane = (idx >> 3) & 63, part = (idx >> 9) % 3, step = idx / (3 * 512);
    const int n = lane & 15, kk = 8 * (lane >> 4) + j, tap = 2 * step + (kk >> 4), k = kk & 15;
    float v = tap < 9 ? w[((long)n * 16 + k) * 9 + tap] * scale : 0.f;
    const __bf16 hi = (__bf16)v;
    const float r1 = v - (float)hi;
    const __bf16 mid = (__bf16)r1;
    packed[idx] = part == 0 ? hi : part == 1 ? mid : (__bf16)(r1 - (float)mid);
}
}  // namespace
extern "C" int ngan_diag_conv3x3_bf16x6(const float* x, const float* w, const float* bias, float* y, float* rnorm, float* packed,
                                        int B, int H, int W, float scale, int epilogue, float slope, float eps, void* stream) {
    if (!x || !w || !y || !packed || B <= 0 || H <= 0 || W <= 0 || W % 32 || (epilogue != 0 && epilogue != 1) || (epilogue == 1 && !rnorm)) return NGAN_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(pack_weights_bf16x6_kernel, dim3(30), dim3(256), 0, s, w, reinterpret_cast<__bf16*>(packed), scale);
    ConvArgs a{x, packed, bias, y, rnorm, B, H, W, 16, 16, 0, 0, slope, eps, nullptr, nullptr, nullptr};
    return epilogue ? launch_tile<1, 1, 1, 0, 3>(a, s) : launch_tile<1, 1, 0, 0, 3>(a, s);
}
#endif

#ifdef NGAN_DIAG_PHASES
// phase-timer build only (not declared in include/ngan.h): copies the phase counters of conv3x3_tile_kernel out and optionally zeroes them
extern "C" int ngan_diag_tile_phases(unsigned long long* out11, int reset) {
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    if (hipMemcpyFromSymbol(out11, HIP_SYMBOL(tile_phase_ctr), 11 * sizeof(unsigned long long)) != hipSuccess) return 1;
    if (reset) {
        const unsigned long long z[11] = {0, 0, 0, 0, 0, 0, 0, ~0ull, 0, 0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(tile_phase_ctr), z, sizeof(z)) != hipSuccess) return 1;
    }
    return 0;
}
#endif
