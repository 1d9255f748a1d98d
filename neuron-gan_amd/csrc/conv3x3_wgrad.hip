// Weight gradients of the 3x3 convolution (ATen convolution_backward's weight half, /root/reference/models.py:203-204 under
// autograd) and the fixed-order slab reduction behind them.
#include "conv3x3_internal.h"

namespace {

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
#ifndef NGAN_WGRAD_W22
#define NGAN_WGRAD_W22 8
#endif
#ifndef NGAN_WGRAD_WINO16
#define NGAN_WGRAD_WINO16 1
#endif
#ifndef NGAN_WGRAD_SMALL
#define NGAN_WGRAD_SMALL (1 << 30)
#endif
// smallest m >= n with m = 4 (mod 64).  (Round 3 tried the smaller "any pitch whose quarter is odd" -- 364 instead of 388 dwords for
// the 10 x 36 halo plane: the Winograd form's reads add 8 q to the lane address, and with a pitch of 44 (mod 64) six of the sixteen
// lanes of a ds_read_b128 group collide instead of two: the 16 -> 16 kernel went from 51 to 106 us per launch.  Reverted.)
constexpr int pad_plane(int n) { return n + ((4 - n % 64) + 64) % 64; }

// WINO = 1 (16 x 16 slices, 8 x 32 tiles): the contraction in Winograd form, dW = G^T [ sum over 2x2 output tiles of (A dY A^T) . (B^T d B) ] G
// -- the backward-filter counterpart of conv3x3_tile_kernel's F(2x2, 3x3).  A wave takes one row of 16 tiles; an MFMA contracts over
// 4 tiles: lane (p, q) holds, for tile 4 q + ks, the transformed 2x2 output-gradient patch of output channel p (A operand) and the
// transformed 4x4 input patch of input channel p (B operand), both computed by itself from its channel plane (2 + 8 ds_read_b64).
// 16 accumulators (one per position of the 4x4 transformed tile) instead of 9 taps; 64 instead of 144 MFMAs per wave and tile.
// The G^T . G back-transform is linear, so every workgroup applies it to its own partial sum before writing the slab: slab
// format, slab reduction and bit-reproducibility are those of the direct form.
// Phase-timer build only (`make phases`: -DNGAN_DIAG -DNGAN_DIAG_PHASES, build/phases/): where a wave's time goes, phase by phase
// (shader clock, summed over the waves of a launch; read back through ngan_diag_wgrad_phases -- tools/wgrad_phases.py).  0 = waiting at the
// tile's first barrier (the other waves still computing), 1 = waiting for this tile's global loads, 2 = LDS writes, 3 = second barrier,
// 4 = issuing the next tile's loads, 5 = operand reads + transforms + MFMAs, 6 = the tail's first barrier (the waves' skew at the end of the
// loop), 7 = the wave's own back-transform + its nine taps written to LDS + barrier (first output group), 8 = fixed-order sum over the row-group
// waves + slab store (and the further output groups' rounds), 9 = unused, 10 = number of waves sampled (one workgroup in eight reports: with
// every wave reporting, the atomics of the early finishers stood in the way of the others' tails and tripled the tail's apparent cost).
#ifdef NGAN_DIAG_PHASES
__device__ unsigned long long wgrad_phase_ctr[15];      // 11 = earliest loop entry, 12 = latest exit, 13 / 14 = sums of entries / exits (s_memrealtime, 100 MHz)
#define PHASE_INIT unsigned long long ph_[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, last_ = __builtin_readcyclecounter(), t0_ = __builtin_amdgcn_s_memrealtime()
#define PHASE_STAMP(i) { const unsigned long long now_ = __builtin_readcyclecounter(); ph_[i] += now_ - last_; last_ = now_; }
#define PHASE_WAIT_LOADS asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define PHASE_FLUSH if (lane == 0 && (blockIdx.x & 7) == 0) { const unsigned long long t1_ = __builtin_amdgcn_s_memrealtime(); \
    for (int i_ = 0; i_ < 10; ++i_) atomicAdd(&wgrad_phase_ctr[i_], ph_[i_]); atomicAdd(&wgrad_phase_ctr[10], 1ull); \
    atomicMin(&wgrad_phase_ctr[11], t0_); atomicMax(&wgrad_phase_ctr[12], t1_); atomicAdd(&wgrad_phase_ctr[13], t0_); atomicAdd(&wgrad_phase_ctr[14], t1_); }
#else
#define PHASE_INIT
#define PHASE_STAMP(i)
#define PHASE_WAIT_LOADS
#define PHASE_FLUSH
#endif

// Round 4, measured and NOT kept (profiles/r04_wgrad_phases.txt, r04_micro_mfma_valu.txt, experiments/r04_wgrad_wave_private.diff):
//  * the lane's ten input columns read once per halo row (two 16-byte + one 8-byte read instead of 16 + 8 per tile pair): 190 VGPRs, two
//    instead of three workgroups per CU, 34 -> 41 us;
//  * WAVE-PRIVATE staging for the 16 x 16-slice instance (every wave loads its own two gradient rows and four halo rows into its own LDS
//    region: no barrier in the tile loop, 67 KB of LDS, 206 VGPRs): the phase timers' 18 % "first barrier" share disappears, the wave's
//    lifetime falls by 6.6 %, the launch by 0 - 3.5 % in isolation (42.4 -> 40.9 us at 256 x 256 incl. the slab reduction, 157.4 -> 158.7 at
//    512 x 512) and by nothing inside the iteration (34.0 -> 33.8 us per launch, 7.05 -> 7.07 ms): the time the waves no longer spend at the
//    barrier they spend queueing for the SIMD's one vector pipe, which the MFMAs (2 048 cycles per wave and tile) and the 176 transform
//    instructions (3 - 5 cycles each beside an fp32 MFMA, measured) of TWO waves keep ~75 % busy;
//  * the transforms on register pairs (22 v_pk_* instead of 44 scalar instructions per 16 MFMAs): the compiler's pre-emit peephole splits
//    packed fp32 instructions in an MFMA's shadow again, and rightly -- beside v_mfma_f32_16x16x4_f32 at two waves per SIMD a packed
//    instruction costs 5 - 9 cycles of the slot against 3 - 5 for a scalar one (tools/micro/mfma_valu.hip);
//  * 768 / 1024 instead of 512 workgroups (three per CU): 42.1 -> 43.0 / 46.7 us.
//  * pairs of adjacent pixels per thread so that the transposing store of g is one ds_write_b64 per channel instead of two ds_write_b32
//    (8 instead of 16 LDS instructions per thread and tile; conflict-free): 707 -> 713 us per iteration for the dominant instance: nothing.
template <int COT, int CIT, int RES, int TW, int NW, int XF, int WINO = 0>
__global__ __launch_bounds__(NW * 64, (WINO && !XF && COT * CIT == 1) ? 2 : (COT * CIT == 1) ? 3 : (COT * CIT == 2 || NW == 4 ? 2 : 1)) void wgrad_f32_kernel(WgradArgs a) {
    // Winograd form: a wave step covers 16 Winograd tiles -- one row of an 8 x 32 tile, or (TW = 16: images at most 16 pixels wide)
    // two rows of 8 of a 16 x 16 tile, lane quarters q = 0, 1 on the upper and q = 2, 3 on the lower row
    constexpr int WSTEP_ROWS = TW == 32 ? 2 : 4;                 // output rows a wave step covers
    constexpr int NT = NW * 64;
    constexpr int TH = 256 / TW, HALO_H = TH + 2, NBLK = TW / 16;
    constexpr int CO_S = COT * 16, CI_S = CIT * 16;
    constexpr int WO = COT * CIT, WR = NW / WO, RPW = TH / WR;                   // wave groups over sub-slices / over rows (direct form)
    constexpr int XP = TW + 4;                                                   // x row pitch (TW + 2 used), a multiple of 4
    constexpr int PLANE_G = pad_plane(TH * TW), PLANE_X = pad_plane(HALO_H * XP);
    constexpr int G_ELEMS = CO_S * PLANE_G, X_ELEMS = CI_S * PLANE_X;
    constexpr int NACC = WINO ? 16 : 9;
    // Winograd form: a wave owns ONE 16-channel input group and ALL the slice's output groups (COT accumulator sets): the 4x4 input
    // patch transform B^T d B -- two thirds of the form's VALU work -- is then done once per (tile, input group), not once per
    // (cout, cin) wave pair as in the direct form's mapping (round 3: 2.75 -> 1.75 VALU instructions per MFMA on 32 x 32 slices).
    // (where the registers allow: the 8-wave 32 x 32 slices with plain / pooled input; 4-wave workgroups stage twice the tile per
    // thread and spilled 70 - 160 registers with two accumulator sets, the bilinear instance 56)
    constexpr bool SHARE = WINO && NW == 8 && RES != NGAN_RESAMPLE_UP2;
    constexpr int CW = SHARE ? COT : 1;                          // output groups per wave
    constexpr int WQ = (COT / CW) * CIT;                         // wave groups over (output group sets, input groups)
    constexpr int WRW = NW / WQ, RPWW = TH / WRW;                // Winograd: row groups, rows per wave
    static_assert(!WINO || (NW % WQ == 0 && TH % WRW == 0 && RPWW % WSTEP_ROWS == 0), "Winograd wave split");
    constexpr int RED_WINO = NW * 9 * 64 * 4;                    // every wave's 9 back-transformed taps of one output group
    constexpr int RED_ELEMS = WINO ? RED_WINO : NW * 9 * 64 * 4;
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

    f32x4 acc[WINO ? 1 : NACC];                      // direct form: one (cout, cin) sub-slice per wave
    f32x4 accw[WINO ? CW : 1][16];                   // Winograd form: CW output groups of the slice, one input group
#pragma unroll
    for (int t = 0; t < (WINO ? 1 : NACC); ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < (WINO ? CW : 1); ++c)
#pragma unroll
        for (int t = 0; t < 16; ++t) accw[c][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int wqw = wave % WQ, wrw = wave / WQ;      // Winograd: this wave's (output group set, input group) and row group
    const int citw = wqw % CIT, cot0w = (wqw / CIT) * CW;

    float4 gst[NG], xst[XF ? NXF : (RES == NGAN_RESAMPLE_UP2 ? NXP : NX)];
    constexpr unsigned OOB = 0xFFFFFFF0u;
    auto issue = [&](const TileCursor& tc) {
        const int b = tc.b, y0 = tc.ty * TH, x0 = tc.tx * TW;
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
    const TileWalk walk(a.tiles_x, a.tiles_y, gridDim.x);       // tiles blockIdx.x, + gridDim.x, ...: decoded once, then advanced (conv3x3_internal.h)
    TileCursor cur_tile = walk.at(tile), next_tile = walk.next(cur_tile);
    if (tile < a.n_tiles) issue(cur_tile);
    PHASE_INIT;
    while (tile < a.n_tiles) {
        __syncthreads();
        PHASE_STAMP(0);
        PHASE_WAIT_LOADS;
        PHASE_STAMP(1);
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
            const int y0 = cur_tile.ty * TH, x0 = cur_tile.tx * TW;
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
        PHASE_STAMP(2);
        __syncthreads();
        PHASE_STAMP(3);
        const int tn = tile + gridDim.x;
        if (tn < a.n_tiles) issue(next_tile);   // next tile's loads are in flight during the MFMAs
        PHASE_STAMP(4);
        if (WINO) {
            // this wave's tile rows: output rows row0, row0 + 1 = halo rows row0 .. row0 + 3.  Signs: A = [1 0; 1 1; 1 -1; 0 -1] is used
            // without the minus signs of its last row (one negation per element saved); the back-transform flips the sign of every
            // position with u = 3 xor v = 3 instead.
            // lane (p, q) owns the four consecutive tiles 4 q .. 4 q + 3 of the row (K-step ks contracts tiles 4 q + ks over q): two tiles
            // at a time are one 16-byte + one 8-byte read per input row and one 16-byte read per gradient row and output group
#pragma unroll
            for (int tr = 0; tr < RPWW / WSTEP_ROWS; ++tr) {
            const int row0 = wrw * RPWW + WSTEP_ROWS * tr + (TW == 32 ? 0 : 2 * (q >> 1)), col0 = 8 * (TW == 32 ? q : (q & 1));
            const float* gp = g_lds + (cot0w * 16 + p) * PLANE_G + row0 * TW + col0;       // + c * 16 * PLANE_G per output group
            const float* xp = x_lds + (citw * 16 + p) * PLANE_X + row0 * XP + col0;
#pragma unroll
            for (int kp = 0; kp < 2; ++kp) {
                float xr[4][6], gr[CW][2][4];
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const float4 lo = ld4(xp + r4 * XP + 4 * kp);
                    const float2 hi = *reinterpret_cast<const float2*>(xp + r4 * XP + 4 * kp + 4);
                    xr[r4][0] = lo.x; xr[r4][1] = lo.y; xr[r4][2] = lo.z; xr[r4][3] = lo.w; xr[r4][4] = hi.x; xr[r4][5] = hi.y;
                }
#pragma unroll
                for (int c = 0; c < CW; ++c)
#pragma unroll
                    for (int r2 = 0; r2 < 2; ++r2) {
                        const float4 v = ld4(gp + c * 16 * PLANE_G + r2 * TW + 4 * kp);
                        gr[c][r2][0] = v.x; gr[c][r2][1] = v.y; gr[c][r2][2] = v.z; gr[c][r2][3] = v.w;
                    }
#pragma unroll
                for (int t2 = 0; t2 < 2; ++t2) {
                    float t[4][4], V[4][4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float d0 = xr[0][2 * t2 + c], d1 = xr[1][2 * t2 + c], d2 = xr[2][2 * t2 + c], d3 = xr[3][2 * t2 + c];
                        t[0][c] = d0 - d2; t[1][c] = d1 + d2; t[2][c] = d2 - d1; t[3][c] = d1 - d3;
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) { V[u][0] = t[u][0] - t[u][2]; V[u][1] = t[u][1] + t[u][2]; V[u][2] = t[u][2] - t[u][1]; V[u][3] = t[u][1] - t[u][3]; }
#pragma unroll
                    for (int co = 0; co < CW; ++co) {
                        float sg[4][2], M[4][4];
#pragma unroll
                        for (int c = 0; c < 2; ++c) {
                            const float g0 = gr[co][0][2 * t2 + c], g1 = gr[co][1][2 * t2 + c];
                            sg[0][c] = g0; sg[1][c] = g0 + g1; sg[2][c] = g0 - g1; sg[3][c] = g1;
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) { M[u][0] = sg[u][0]; M[u][1] = sg[u][0] + sg[u][1]; M[u][2] = sg[u][0] - sg[u][1]; M[u][3] = sg[u][1]; }
#pragma unroll
                        for (int u = 0; u < 4; ++u)
#pragma unroll
                            for (int v = 0; v < 4; ++v)
                                accw[WINO ? co : 0][u * 4 + v] = __builtin_amdgcn_mfma_f32_16x16x4f32(M[u][v], V[u][v], accw[WINO ? co : 0][u * 4 + v], 0, 0, 0);
                    }
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
        PHASE_STAMP(5);
        tile = tn;
        cur_tile = next_tile;
        next_tile = walk.next(next_tile);
    }

    // ---- sum the WR row-waves of each sub-slice through LDS (fixed order), then write this block's slab ----
    __syncthreads();
    PHASE_STAMP(6);
    float4* red = reinterpret_cast<float4*>(smem);
    float* slab = a.partial + ((long)blockIdx.x * gridDim.y + slice) * (9 * CO_S * CI_S);
    if (!WINO) {
#pragma unroll
        for (int t = 0; t < NACC; ++t)
            red[(wave * NACC + t) * 64 + lane] = make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
        __syncthreads();
    }
    if (WINO) {
        // The back-transform  dW = G^T [ s . dU ] G  (G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]; s = -1 where u = 3 xor v = 3, MFMA section) is linear
        // and lane-local -- a lane holds all 16 positions of its (cout quad, cin) entries -- so every wave applies it to its OWN partial sums
        // first (once per launch: ~35 float4 operations per output group) and the waves then exchange 9 taps instead of 16 positions, in ONE
        // round per output group: two barriers per group instead of eleven in all.  (Round 4: the phase timer showed the old tail -- four
        // position passes, row transform and column transform each through LDS -- at 30 % of a wave's life on the 64 -> 64 layers, two
        // tiles per workgroup.)  The summation order changes (transform, then the fixed-order sum over the row-group waves), the bits of a
        // run stay reproducible.
#pragma unroll
        for (int c = 0; c < CW; ++c) {
            if (c) __syncthreads();                              // the previous group's readers are done with `red`
            // one column j of the 3 x 3 result at a time (Z[.][j] needs the whole row of positions, dW[.][j] the four Z[.][j]): seven live
            // float4s beside the accumulators, written to LDS as they are finished -- all nine at once cost 45 spilled registers in the
            // dominant instance
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                f32x4 Zj[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const f32x4 d1 = accw[WINO ? c : 0][u * 4 + 1], d2 = accw[WINO ? c : 0][u * 4 + 2];
                    const float su = u == 3 ? -1.f : 1.f;
                    Zj[u] = j == 0 ? (accw[WINO ? c : 0][u * 4 + 0] + (d1 + d2) * 0.5f) * su
                          : j == 1 ? ((d1 - d2) * 0.5f) * su
                                   : ((d1 + d2) * 0.5f - accw[WINO ? c : 0][u * 4 + 3]) * su;
                }
                const f32x4 p12 = (Zj[1] + Zj[2]) * 0.5f, m12 = (Zj[1] - Zj[2]) * 0.5f;
                const f32x4 w0 = Zj[0] + p12, w2 = p12 + Zj[3];
                red[(wave * 9 + 0 * 3 + j) * 64 + lane] = make_float4(w0[0], w0[1], w0[2], w0[3]);      // wave = wrw * WQ + wqw
                red[(wave * 9 + 1 * 3 + j) * 64 + lane] = make_float4(m12[0], m12[1], m12[2], m12[3]);
                red[(wave * 9 + 2 * 3 + j) * 64 + lane] = make_float4(w2[0], w2[1], w2[2], w2[3]);
            }
            __syncthreads();
            if (c == 0) PHASE_STAMP(7);
            // item (l, tap t, oi): wave group oi = (output group set, input group) of this pass, summed over its WRW row-group waves in fixed order
            for (int item = tid; item < WQ * 9 * 64; item += NT) {
                const int l = item & 63, t = (item >> 6) % 9, oi = (item >> 6) / 9;
                float4 v = red[(oi * 9 + t) * 64 + l];
#pragma unroll
                for (int w = 1; w < WRW; ++w) v = f4add(v, red[((w * WQ + oi) * 9 + t) * 64 + l]);
                const int ci_l = (oi % CIT) * 16 + (l & 15), co_l = ((oi / CIT) * CW + c) * 16 + 4 * (l >> 4);
                float* op = slab + ((long)t * CO_S + co_l) * CI_S + ci_l;
                op[0] = v.x; op[CI_S] = v.y; op[2 * CI_S] = v.z; op[3 * CI_S] = v.w;
            }
        }
        PHASE_STAMP(8);
        PHASE_STAMP(9);
        PHASE_FLUSH;
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

// PLAIN = true: bf16 activation storage (precision code 5, include/ngan.h "bf16 activation storage"): x and g ARE bf16 tensors, so there
// is one plane per operand instead of hi + lo, the staging loads are 8 bytes per 4 channels, and a product group is ONE MFMA.
// Resampled input (avg-pool / bilinear) is blended in fp32 from the bf16 sources and rounded once, like the forward kernel's staging.
// XF (plain input, 8 x 32 tiles, image width a multiple of 32): the staging of wgrad_f32_kernel's XF form -- per-lane byte offsets are
// tile-invariant constants, the tile moves the descriptors' bases with scalar instructions, padding is done by whole loads (top halo row
// behind a zero-record descriptor, bottom by the range check, the two halo columns in one extra load).  The generic form computes and
// range-checks an address per load: 118 vector + 106 scalar instructions per tile and wave in front of 18 MFMAs (round-4 ISA count).
template <int COT, int CIT, int RES, int TW, bool PLAIN = false, bool XF = false>
__global__ __launch_bounds__(256, 2) void wgrad_bf16x3_kernel(WgradArgs a) {
    static_assert(!XF || (RES == NGAN_RESAMPLE_NONE && TW == 32), "fast staging: plain input, 8 x 32 tiles");
    // tile = 256 pixels: 8 x 32, or 16 x 16 for images at most 16 wide.  One k-step = 32 pixels = one tile row (TW = 32) or two
    // consecutive rows (TW = 16): the second 16-pixel half of a fragment then sits one halo row further instead of 16 pixels.
    constexpr int TH = 256 / TW, HALO_H = TH + 2, HALO_W = TW + 2, G_PIX = TH * TW, X_PIX = HALO_H * HALO_W;
    constexpr int CO_S = COT * 16, CI_S = CIT * 16;
    constexpr int WO = COT * CIT, WR = 4 / WO, NKS = 8, KPW = NKS / WR;          // k-steps per tile / per wave
    constexpr int X_HALF2 = (TW == 32 ? 16 : HALO_W) * 16;                       // bf16 offset of a fragment's second half in x
    constexpr int X_ROWS_PER_KS = TW == 32 ? 1 : 2;
    constexpr int NPL = PLAIN ? 1 : 2;                                           // planes per operand: hi (+ lo)
    constexpr int ESZ = PLAIN ? 2 : 4;                                           // bytes per stored activation element
    constexpr int G_E = NPL * COT * G_PIX * 16, X_E = NPL * CIT * X_PIX * 16;    // bf16 elements
    // bilinear input: the low-resolution source patch is staged once (fp32) and expanded LDS -> LDS, as in conv3x3_persist_kernel
    constexpr int PH = TH / 2 + 2, PW = TW / 2 + 2, NPP = PH * PW;
    constexpr int PATCH_BYTES = RES == NGAN_RESAMPLE_UP2 ? NPP * CI_S * 4 : 0;
    constexpr int RED_BYTES = 4 * 9 * 64 * 16;
    constexpr int IMG_BYTES = (G_E + X_E) * 2 + PATCH_BYTES;
    constexpr int SMEM_BYTES = IMG_BYTES > RED_BYTES ? IMG_BYTES : RED_BYTES;
    constexpr int NG = G_PIX * (CO_S / 4) / 256, NXI = X_PIX * (CI_S / 4), NX = (NXI + 255) / 256;
    constexpr int Q = CI_S / 4, QG = CO_S / 4;
    constexpr int NXINT = HALO_H * 32 * Q / 256, NXF = NXINT + 1, N_HALO = 2 * HALO_H * Q;             // XF: interior loads per thread, + the halo-column load
    static_assert(!XF || ((HALO_H * 32 * Q) % 256 == 0 && N_HALO <= 256), "fast staging layout");
    constexpr int NPI = NPP * (CI_S / 4), NXL = XF ? NXF : RES == NGAN_RESAMPLE_UP2 ? (NPI + 255) / 256 : NX;   // global loads per thread for x
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
    // XF: byte offsets from the tile origin (g) / the halo origin (y0 - 1, x0 - 1) (x), bf16 index of the hi part in LDS
    unsigned gf_off[XF ? NG : 1], xf_off[XF ? NXF : 1];
    int xf_l[XF ? NXF : 1], xf_bits = 8;          // halo load: 1 = left column, 2 = right column, 4 = top row, 8 = unused lane
    if constexpr (XF) {
#pragma unroll
        for (int i = 0; i < NG; ++i) {
            const int e = tid + i * 256, pix = e / QG, c4 = e % QG;
            gf_off[i] = (unsigned)((((pix >> 5) * a.W + (pix & 31)) * a.N + co0 + c4 * 4) * ESZ);
        }
#pragma unroll
        for (int i = 0; i < NXINT; ++i) {
            const int e = tid + i * 256, c4 = e % Q, pix = e / Q, r = pix >> 5, c = (pix & 31) + 1;
            xf_off[i] = (unsigned)(((r * a.W + c) * a.K + ci0 + c4 * 4) * ESZ);
            xf_l[i] = ((c4 >> 2) * X_PIX + r * HALO_W + c) * 16 + (c4 & 3) * 4;
        }
        const int c4 = tid % Q, r = (tid / Q) % HALO_H, side = tid / (Q * HALO_H), c = side ? 33 : 0;
        const bool used = tid < N_HALO;
        xf_off[NXINT] = used ? (unsigned)(((r * a.W + c) * a.K + ci0 + c4 * 4) * ESZ) : 0xFFFFFFF0u;
        xf_l[NXINT] = ((c4 >> 2) * X_PIX + (used ? r : 0) * HALO_W + c) * 16 + (c4 & 3) * 4;
        xf_bits = used ? ((side ? 2 : 1) | (r == 0 ? 4 : 0)) : 8;
    }
    // this lane's transposing-read offsets (bf16 elements) inside one plane: pixel 4kq + qq (second read: + 16), channels 4pp..
    const int tr0 = (4 * kq + qq) * 16 + 4 * pp;

    f32x4 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    float4 gst[NG], xst[NXL];
    uint2 graw[XF && PLAIN ? NG : 1], xraw[XF && PLAIN ? NXF : 1];     // XF on bf16 storage: the tile is COPIED (8 bytes per lane and load), not converted
    auto issue = [&](const TileCursor& tc) {
        const int b = tc.b, y0 = tc.ty * TH, x0 = tc.tx * TW;
        // loads through per-image buffer descriptors: 32-bit offsets, and an out-of-range offset (tile edge, conv padding, unused
        // staging slot) reads zeros -- no branch around the load and no zero-filled registers (see conv3x3_persist_kernel)
        constexpr unsigned OOB = 0xFFFFFFF0u;
        // (byte addresses through char*: with PLAIN the tensors behind a.g / a.x hold 2-byte elements)
        const __amdgpu_buffer_rsrc_t g_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char*>(reinterpret_cast<const char*>(a.g) + (long)b * a.H * a.W * a.N * ESZ), 0, (unsigned)(a.H * a.W * a.N) * ESZ, 0x00020000);
        auto load4b = [&](const __amdgpu_buffer_rsrc_t& rs, unsigned bo) -> float4 {     // 4 consecutive channels at byte offset bo
            if constexpr (PLAIN) {
                const uint2 u = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(rs, bo, 0, 0));
                return make_float4(bf16_lo(u.x), bf16_hi(u.x), bf16_lo(u.y), bf16_hi(u.y));
            } else {
                return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, bo, 0, 0));
            }
        };
        if constexpr (XF) {
            const char* gb = reinterpret_cast<const char*>(a.g) + (long)b * a.H * a.W * a.N * ESZ;
            const char* xb = reinterpret_cast<const char*>(a.x) + (long)b * a.H * a.W * a.K * ESZ;
            const int gs = (y0 * a.W + x0) * a.N * ESZ;                          // rows below the image lie past the records: zeros
            const __amdgpu_buffer_rsrc_t g_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(gb + gs), 0, (unsigned)(a.H * a.W * a.N * ESZ) - (unsigned)gs, 0x00020000);
#pragma unroll
            for (int i = 0; i < NG; ++i) {
                if constexpr (PLAIN) graw[XF && PLAIN ? i : 0] = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(g_rs, gf_off[XF ? i : 0], 0, 0));
                else gst[i] = load4b(g_rs, gf_off[XF ? i : 0]);
            }
            const int xs = ((y0 - 1) * a.W + (x0 - 1)) * a.K * ESZ;              // negative on the top row / for the first tile
            const unsigned nrec = (unsigned)(a.H * a.W * a.K * ESZ) - (unsigned)xs;
            const __amdgpu_buffer_rsrc_t x_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(xb + xs), 0, nrec, 0x00020000);
            const __amdgpu_buffer_rsrc_t x_none = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(xb + xs), 0, 0u, 0x00020000);
#pragma unroll
            for (int i = 0; i < NXINT; ++i) {
                const bool top = i * 256 + wave * 64 < 32 * Q;                   // a wave's load lies in one halo row; row 0 is above the image when y0 = 0
                if constexpr (PLAIN) xraw[XF && PLAIN ? i : 0] = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64((top && y0 == 0) ? x_none : x_rs, xf_off[XF ? i : 0], 0, 0));
                else xst[XF ? i : 0] = load4b((top && y0 == 0) ? x_none : x_rs, xf_off[XF ? i : 0]);
            }
            const int bad = (x0 == 0 ? 1 : 0) | (x0 + 32 >= a.W ? 2 : 0) | (y0 == 0 ? 4 : 0) | 8;
            const unsigned hoff = (xf_bits & bad) ? OOB : xf_off[XF ? NXINT : 0];
            if constexpr (PLAIN) xraw[XF && PLAIN ? NXINT : 0] = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(x_rs, hoff, 0, 0));
            else xst[XF ? NXINT : 0] = load4b(x_rs, hoff);
            return;
        }
        auto load4 = [&](const __amdgpu_buffer_rsrc_t& rs, unsigned off) -> float4 {     // 4 consecutive channels at element-offset off
            const unsigned bo = off == OOB ? OOB : off * (unsigned)ESZ;          // out of range: the descriptor's range check returns zeros
            if constexpr (PLAIN) {
                const uint2 u = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(rs, bo, 0, 0));
                return make_float4(bf16_lo(u.x), bf16_hi(u.x), bf16_lo(u.y), bf16_hi(u.y));
            } else {
                return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, bo, 0, 0));
            }
        };
#pragma unroll
        for (int i = 0; i < NG; ++i) {
            const int gy = y0 + g_r[i], gx = x0 + g_c[i];
            const unsigned off = (gy < a.H && gx < a.W) ? (unsigned)((gy * a.W + gx) * a.N + g_ch[i]) : OOB;
            gst[i] = load4(g_rsrc, off);
        }
        if (RES == NGAN_RESAMPLE_UP2) {
            const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<char*>(reinterpret_cast<const char*>(a.x) + (long)b * h * w * a.K * ESZ), 0, (unsigned)(h * w * a.K) * ESZ, 0x00020000);
            const int ly0 = (y0 >> 1) - 1, lx0 = (x0 >> 1) - 1;
#pragma unroll
            for (int i = 0; i < NXL; ++i) {
                const int e = tid + i * 256;
                const int pix = e / (CI_S / 4), c4 = e % (CI_S / 4);
                const int ly = min(max(ly0 + pix / PW, 0), h - 1), lx = min(max(lx0 + pix % PW, 0), w - 1);
                const unsigned off = e < NPI ? (unsigned)((ly * w + lx) * a.K + ci0 + c4 * 4) : OOB;
                xst[i] = load4(x_rsrc, off);
            }
        } else if (RES == NGAN_RESAMPLE_NONE) {
            const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<char*>(reinterpret_cast<const char*>(a.x) + (long)b * a.H * a.W * a.K * ESZ), 0, (unsigned)(a.H * a.W * a.K) * ESZ, 0x00020000);
#pragma unroll
            for (int i = 0; i < NXL; ++i) {
                const int gy = y0 + x_r[i], gx = x0 + x_c[i];        // x_r = -1000 marks an unused slot: fails the range test
                const bool ok = (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
                const unsigned off = ok ? (unsigned)((gy * a.W + gx) * a.K + x_ch[i]) : OOB;
                xst[i] = load4(x_rsrc, off);
            }
        } else {
#pragma unroll
            for (int i = 0; i < NXL; ++i) {
                if constexpr (PLAIN) {     // avg-pooled input: the 2x2 mean of the bf16 source, associated like ngan_pool2_fwd
                    const int gy = y0 + x_r[i], gx = x0 + x_c[i];
                    float4 v = f4zero();
                    if (x_r[i] > -1000 && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W) {
                        const long W2 = 2L * a.W;
                        const __bf16* sp = reinterpret_cast<const __bf16*>(a.x) + (((long)b * 2 * a.H + 2 * gy) * W2 + 2 * gx) * a.K + x_ch[i];
                        const float4 p00 = lda4(sp), p01 = lda4(sp + a.K), p10 = lda4(sp + W2 * a.K), p11 = lda4(sp + W2 * a.K + a.K);
                        v = f4scale(f4add(f4add(p00, p01), f4add(p10, p11)), 0.25f);
                    }
                    xst[i] = v;
                } else {
                    xst[i] = x_r[i] > -1000 ? load_resampled<RES>(a.x, b, y0 + x_r[i], x0 + x_c[i], x_ch[i], a.H, a.W, a.K) : f4zero();
                }
            }
        }
    };
    auto split_store = [&](__bf16* img, int idx, int lo_off, float4 v) {
        bf16x4 hi, lo;
        hi[0] = (__bf16)v.x; hi[1] = (__bf16)v.y; hi[2] = (__bf16)v.z; hi[3] = (__bf16)v.w;
        *reinterpret_cast<bf16x4*>(img + idx) = hi;
        if constexpr (!PLAIN) {
            lo[0] = (__bf16)(v.x - (float)hi[0]); lo[1] = (__bf16)(v.y - (float)hi[1]);
            lo[2] = (__bf16)(v.z - (float)hi[2]); lo[3] = (__bf16)(v.w - (float)hi[3]);
            *reinterpret_cast<bf16x4*>(img + idx + lo_off) = lo;
        }
    };

    int tile = blockIdx.x;
    const TileWalk walk(a.tiles_x, a.tiles_y, gridDim.x);
    TileCursor cur_tile = walk.at(tile), next_tile = walk.next(cur_tile);
    if (tile < a.n_tiles) issue(cur_tile);
    while (tile < a.n_tiles) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NG; ++i) {
            if constexpr (XF && PLAIN) *reinterpret_cast<uint2*>(g_img + g_l[i]) = graw[XF && PLAIN ? i : 0];
            else split_store(g_img, g_l[i], COT * G_PIX * 16, gst[i]);
        }
        if (RES == NGAN_RESAMPLE_UP2) {
#pragma unroll
            for (int i = 0; i < NXL; ++i)
                if (tid + i * 256 < NPI) st4(patch + (tid + i * 256) * 4, xst[i]);       // patch is plain [py][px][CI_S]
            __syncthreads();
            const int y0 = cur_tile.ty * TH, x0 = cur_tile.tx * TW;
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
        } else if constexpr (XF) {
#pragma unroll
            for (int i = 0; i < NXF; ++i)
                if (i < NXINT || tid < N_HALO) {
                    if constexpr (PLAIN) *reinterpret_cast<uint2*>(x_img + xf_l[XF ? i : 0]) = xraw[XF && PLAIN ? i : 0];
                    else split_store(x_img, xf_l[XF ? i : 0], CIT * X_PIX * 16, xst[XF ? i : 0]);
                }
        } else {
#pragma unroll
            for (int i = 0; i < NX; ++i)
                if (x_r[i] > -1000) split_store(x_img, x_l[i], CIT * X_PIX * 16, xst[i]);
        }
        __syncthreads();
        const int tn = tile + gridDim.x;
        if (tn < a.n_tiles) issue(next_tile);
        const __bf16* gh = g_img + cot * G_PIX * 16 + tr0;
        const __bf16* gl = gh + COT * G_PIX * 16;
        const __bf16* xh = x_img + cit * X_PIX * 16 + tr0;
        const __bf16* xl = xh + CIT * X_PIX * 16;
        for (int kk = 0; kk < KPW; ++kk) {
            const int ks = wr * KPW + kk;
            const bf16x8 ah = tr_frag(gh + ks * 32 * 16, gh + (ks * 32 + 16) * 16);
            bf16x8 al;
            if constexpr (!PLAIN) al = tr_frag(gl + ks * 32 * 16, gl + (ks * 32 + 16) * 16);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int dy = tap / 3, dx = tap % 3;
                const int xo = ((ks * X_ROWS_PER_KS + dy) * HALO_W + dx) * 16;
                const bf16x8 bh = tr_frag(xh + xo, xh + xo + X_HALF2);
                if constexpr (!PLAIN) {
                    const bf16x8 bl = tr_frag(xl + xo, xl + xo + X_HALF2);
                    acc[tap] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc[tap], 0, 0, 0);
                    acc[tap] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc[tap], 0, 0, 0);
                }
                acc[tap] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc[tap], 0, 0, 0);
            }
        }
        tile = tn;
        cur_tile = next_tile;
        next_tile = walk.next(next_tile);
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
constexpr int wgrad_f32_waves(int cot, int cit) { return cot * cit == 4 ? NGAN_WGRAD_W22 : 4; }    // a 32 x 32 slice: one wave (round 2: two) per 16 x 16 sub-slice

WgradPlan plan_wgrad(int B, int H, int W, int Cin, int Cout, int precision = 1) {
    WgradPlan p;
    p.tw = W <= 16 ? 16 : 32;   // 16x16-pixel tiles for narrow images, 8x32 otherwise
    p.tiles_x = ngan::ceil_div(W, p.tw);
    p.tiles_y = ngan::ceil_div(H, 256 / p.tw);
    p.n_tiles = B * p.tiles_x * p.tiles_y;
    // fp32 layers with at most 32 channels on either side: 16 x 16 slices -- four times the workgroups of a 32 x 32 slicing, three
    // resident per CU instead of one 8-wave workgroup.  Measured per tile count (NGAN_WGRAD_SMALL = threshold, round 3): 32 -> 32 at
    // 64x64, batch 16 / 32: 23.6 -> 18.5 / 29.3 -> 25.6 us; at 128x128, batch 16: 43.6 -> 41.0; batch 32: 71.0 -> 71.4; 32 -> 16 at 256x256:
    // 77.5 -> 74.1; whole iteration 7.29 -> 7.25 ms with no threshold at all, which is what stays.
    const bool small = precision == 0 && Cin <= 32 && Cout <= 32 && p.tw == 32 && p.n_tiles <= NGAN_WGRAD_SMALL;
    p.co_s = (Cout % 32 == 0 && !small) ? 32 : 16;
    p.ci_s = (Cin % 32 == 0 && !small) ? 32 : 16;
    p.n_ci_slices = Cin / p.ci_s;
    p.nslices = (Cout / p.co_s) * p.n_ci_slices;
    const int forced = NGAN_DIAG_INT("NGAN_WGRAD_SLABS", 0);
    // about two resident workgroups per CU: few slabs to reduce afterwards (256 / 768 measured slower).  The fp32 kernel's 32 x 32
    // slices are 8-wave workgroups with 83 KB of LDS, one per CU: 256 of them (fp32 32 -> 32 at 128x128: 100 vs 106 us, 64 -> 64 at
    // 32x32: 33 vs 39 us)
    const int total = forced > 0 ? forced : ((precision == 0 && p.co_s == 32 && p.ci_s == 32 && NGAN_WGRAD_W22 == 8) ? 256 : 512);
    int cap = total / p.nslices;
    if (cap < 1) cap = 1;
    p.nwx = p.n_tiles < cap ? p.n_tiles : cap;
    return p;
}

inline bool wgrad_wino_on() { return NGAN_DIAG_FLAG("NGAN_WINOGRAD_WGRAD", true); }   // (one latch for the launch and its label)

template <int COT, int CIT>
int launch_wgrad(const WgradArgs& a, const WgradPlan& p, int res, int precision, hipStream_t s) {
    dim3 grid(p.nwx, p.nslices);
    if (precision == 5) {       // bf16 activation storage: x and g are bf16 tensors (ngan_bf16_conv3x3_wgrad)
        if (p.tw == 32) {
            if (res == 0 && a.W % 32 == 0) hipLaunchKernelGGL((wgrad_bf16x3_kernel<COT, CIT, 0, 32, true, true>), grid, dim3(256), 0, s, a);
            else if (res == 0) hipLaunchKernelGGL((wgrad_bf16x3_kernel<COT, CIT, 0, 32, true>), grid, dim3(256), 0, s, a);
            else if (res == 1) hipLaunchKernelGGL((wgrad_bf16x3_kernel<COT, CIT, 1, 32, true>), grid, dim3(256), 0, s, a);
            else hipLaunchKernelGGL((wgrad_bf16x3_kernel<COT, CIT, 2, 32, true>), grid, dim3(256), 0, s, a);
        } else {
            if (res == 0) hipLaunchKernelGGL((wgrad_bf16x3_kernel<COT, CIT, 0, 16, true>), grid, dim3(256), 0, s, a);
            else if (res == 1) hipLaunchKernelGGL((wgrad_bf16x3_kernel<COT, CIT, 1, 16, true>), grid, dim3(256), 0, s, a);
            else hipLaunchKernelGGL((wgrad_bf16x3_kernel<COT, CIT, 2, 16, true>), grid, dim3(256), 0, s, a);
        }
        return ngan::launch_status("ngan_bf16_conv3x3_wgrad");
    }
    if (precision == 1 && p.tw == 32) {
        if (res == 0 && a.W % 32 == 0) hipLaunchKernelGGL((wgrad_bf16x3_kernel<COT, CIT, 0, 32, false, true>), grid, dim3(256), 0, s, a);
        else if (res == 0) hipLaunchKernelGGL((wgrad_bf16x3_kernel<COT, CIT, 0, 32>), grid, dim3(256), 0, s, a);
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
    constexpr int NW = wgrad_f32_waves(COT, CIT);
    if (wgrad_wino_on() && p.tw == 16 && NGAN_WGRAD_WINO16) {
        if (res == 0) hipLaunchKernelGGL((wgrad_f32_kernel<COT, CIT, 0, 16, NW, 0, 1>), grid, dim3(NW * 64), 0, s, a);
        else if (res == 1) hipLaunchKernelGGL((wgrad_f32_kernel<COT, CIT, 1, 16, NW, 0, 1>), grid, dim3(NW * 64), 0, s, a);
        else hipLaunchKernelGGL((wgrad_f32_kernel<COT, CIT, 2, 16, NW, 0, 1>), grid, dim3(NW * 64), 0, s, a);
        return ngan::launch_status("ngan_conv3x3_wgrad(f32, winograd, 16 x 16 tiles)");
    }
    if (wgrad_wino_on() && p.tw == 32) {
        if (res == 0 && a.W % 32 == 0) hipLaunchKernelGGL((wgrad_f32_kernel<COT, CIT, 0, 32, NW, 1, 1>), grid, dim3(NW * 64), 0, s, a);
        else if (res == 0) hipLaunchKernelGGL((wgrad_f32_kernel<COT, CIT, 0, 32, NW, 0, 1>), grid, dim3(NW * 64), 0, s, a);
        else if (res == 1) hipLaunchKernelGGL((wgrad_f32_kernel<COT, CIT, 1, 32, NW, 0, 1>), grid, dim3(NW * 64), 0, s, a);
        else hipLaunchKernelGGL((wgrad_f32_kernel<COT, CIT, 2, 32, NW, 0, 1>), grid, dim3(NW * 64), 0, s, a);
        return ngan::launch_status("ngan_conv3x3_wgrad(f32, winograd)");
    }
    if (p.tw == 32) {
#ifdef NGAN_DIAG                                   // the direct form on 8 x 32 tiles: reachable with NGAN_WINOGRAD_WGRAD=0 only
        if (res == 0 && a.W % 32 == 0) hipLaunchKernelGGL((wgrad_f32_kernel<COT, CIT, 0, 32, NW, 1>), grid, dim3(NW * 64), 0, s, a);
        else if (res == 0) hipLaunchKernelGGL((wgrad_f32_kernel<COT, CIT, 0, 32, NW, 0>), grid, dim3(NW * 64), 0, s, a);
        else if (res == 1) hipLaunchKernelGGL((wgrad_f32_kernel<COT, CIT, 1, 32, NW, 0>), grid, dim3(NW * 64), 0, s, a);
        else hipLaunchKernelGGL((wgrad_f32_kernel<COT, CIT, 2, 32, NW, 0>), grid, dim3(NW * 64), 0, s, a);
#endif
    } else {
        if (res == 0) hipLaunchKernelGGL((wgrad_f32_kernel<COT, CIT, 0, 16, NW, 0>), grid, dim3(NW * 64), 0, s, a);
        else if (res == 1) hipLaunchKernelGGL((wgrad_f32_kernel<COT, CIT, 1, 16, NW, 0>), grid, dim3(NW * 64), 0, s, a);
        else hipLaunchKernelGGL((wgrad_f32_kernel<COT, CIT, 2, 16, NW, 0>), grid, dim3(NW * 64), 0, s, a);
    }
    return ngan::launch_status("ngan_conv3x3_wgrad(f32)");
}

}  // namespace

extern "C" int ngan_conv3x3_wgrad_kernel_name(int B, int H, int W, int Cin, int Cout, int resample, int precision, char* buf, int len) {
    NGAN_REQUIRE(buf && len > 0, NGAN_ERR_ARG, "conv3x3_wgrad_kernel_name: bad buffer");
    NGAN_REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && Cin % 16 == 0 && Cout % 16 == 0, NGAN_ERR_SHAPE,
                 "conv3x3_wgrad_kernel_name: bad shape");
    const WgradPlan p = plan_wgrad(B, H, W, Cin, Cout, precision);
    const char* xf = (resample == 0 && p.tw == 32 && W % 32 == 0) ? "true" : "false";
    if (precision == 1) snprintf(buf, len, "wgrad_bf16x3_kernel<%d, %d, %d, %d, false, %s>", p.co_s / 16, p.ci_s / 16, resample, p.tw, xf);
    else if (precision == 5) snprintf(buf, len, "wgrad_bf16x3_kernel<%d, %d, %d, %d, true, %s>", p.co_s / 16, p.ci_s / 16, resample, p.tw, xf);
    else {
        const bool wino = wgrad_wino_on() && (p.tw == 32 || NGAN_WGRAD_WINO16);
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

// One workgroup of 1024 threads sums a SPAN of consecutive elements of a gradient over all slabs, split into NG slab groups:
// thread (group g, element e) adds slabs g, g + NG, ... with eight independent partial sums (eight loads in flight), the groups are
// summed through LDS in a fixed order.  NG is chosen per entry so that a thread has about eight slabs per source: a 16-channel layer
// (512 slabs, 2 304 elements) takes NG = 16 groups of 64-element spans -- the launch is then bound by the chain of dependent round
// trips per wave, which 16 groups cut by 16 -- while a 128-channel layer (32 slabs, 147 456 elements) takes NG = 4 groups of 256
// elements: round 2's fixed NG = 16 gave it 2 304 workgroups of 1 024 threads with TWO loads per thread each and made those layers
// three quarters of the launch's time.  A wave always reads whole 256-byte runs of one slab.
constexpr int kReduceThreads = 1024;
__host__ __device__ inline int reduce_groups(const ReduceEntry& e) {
    int np = 0;
    for (int s = 0; s < e.nsrc; ++s) np = e.nparts[s] > np ? e.nparts[s] : np;
    return np >= 128 ? 16 : np >= 64 ? 8 : np >= 32 ? 4 : np >= 16 ? 2 : 1;
}
__global__ __launch_bounds__(kReduceThreads) void wgrad_reduce_many_kernel(ReduceBatch b) {
    // a thread owns FOUR consecutive elements (one 16-byte load per slab: a wave reads 1 KB runs, four times the bytes in flight of
    // the dword version, which ran at 2.4 TB/s); a slab is 9 * co_s * ci_s floats, a multiple of 4, so a quad never straddles M
    __shared__ float4 red[kReduceThreads];
    int ei = 0;
    for (int i = 1; i < b.n; ++i)
        if ((int)blockIdx.x >= b.e[i].first_block) ei = i;
    const ReduceEntry& e = b.e[ei];
    const int slab = 9 * e.co_s * e.ci_s;
    const long M = (long)e.nslices * slab;
    const int NG = reduce_groups(e), span = kReduceThreads / NG;
    const int tid = threadIdx.x, el = tid & (span - 1), grp = tid / span;
    const long i = ((long)(blockIdx.x - e.first_block) * span + el) * 4;
    float4 total = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s = 0; s < e.nsrc; ++s) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < M) {
            // eight independent partial sums: the slab reads of one thread are in flight together instead of one per round trip
            const float* src = e.partial[s] + i;
            const int np = e.nparts[s];
            float4 a[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) a[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            int j = grp;
            for (; j + 7 * NG < np; j += 8 * NG) {
#pragma unroll
                for (int k = 0; k < 8; ++k) a[k] = f4add(a[k], ld4(src + (long)(j + k * NG) * M));
            }
            for (; j < np; j += NG) a[0] = f4add(a[0], ld4(src + (long)j * M));
#pragma unroll
            for (int k = 0; k < 4; ++k) a[k] = f4add(a[k], a[k + 4]);
            acc = f4add(f4add(a[0], a[1]), f4add(a[2], a[3]));
        }
        __syncthreads();
        red[tid] = acc;
        __syncthreads();
        if (tid < span) {
            float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int j = 0; j < NG; ++j) t = f4add(t, red[tid + span * j]);
            total = f4fma(t, e.scale[s], total);
        }
    }
    if (tid < span && i < M) {
        const float tv[4] = {total.x, total.y, total.z, total.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            int r = (int)((i + k) % slab);
            const int slice = (int)((i + k) / slab);
            const int ci_l = r % e.ci_s; r /= e.ci_s;
            const int co_l = r % e.co_s;
            const int tap = r / e.co_s;
            const int co = (slice / e.n_ci_slices) * e.co_s + co_l, ci = (slice % e.n_ci_slices) * e.ci_s + ci_l;
            float* o = e.gw + ((long)co * e.K + ci) * 9 + tap;
            *o = e.accumulate ? *o + tv[k] : tv[k];
        }
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
            blocks += ngan::ceil_div((long)b.e[i].nslices * 9 * b.e[i].co_s * b.e[i].ci_s, 4 * (kReduceThreads / reduce_groups(b.e[i])));
        }
        hipLaunchKernelGGL(wgrad_reduce_many_kernel, dim3(blocks), dim3(kReduceThreads), 0, (hipStream_t)stream, b);
        int st = ngan::launch_status("ngan_conv3x3_wgrad_reduce_many");
        if (st) return st;
    }
    return NGAN_OK;
}

static int wgrad_entry(const float* x, const float* g, float* gw, float* workspace,
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
    NGAN_REQUIRE(precision == 0 || precision == 1 || precision == 5, NGAN_ERR_ARG, "conv3x3_wgrad: precision %d", precision);
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


extern "C" int ngan_conv3x3_wgrad(const float* x, const float* g, float* gw, float* workspace,
                                  int B, int H, int W, int Cin, int Cout, int resample, float scale, int accumulate,
                                  int precision, void* stream) {
    NGAN_REQUIRE(precision != 5, NGAN_ERR_ARG, "conv3x3_wgrad: precision 5 (bf16 activation storage) has its own entry point, ngan_bf16_conv3x3_wgrad");
    return wgrad_entry(x, g, gw, workspace, B, H, W, Cin, Cout, resample, scale, accumulate, precision, stream);
}

// x and g are bf16 tensors; slabs, gw and the reduction are fp32 (include/ngan.h, "bf16 activation storage")
extern "C" int ngan_bf16_conv3x3_wgrad(const ngan_bf16* x, const ngan_bf16* g, float* gw, float* workspace,
                                       int B, int H, int W, int Cin, int Cout, int resample, float scale, int accumulate, void* stream) {
    return wgrad_entry(reinterpret_cast<const float*>(x), reinterpret_cast<const float*>(g), gw, workspace, B, H, W, Cin, Cout, resample, scale,
                       accumulate, 5, stream);
}

#ifdef NGAN_DIAG_PHASES
// phase-timer build only (not declared in include/ngan.h): copies the phase counters of wgrad_f32_kernel out and optionally zeroes them
extern "C" int ngan_diag_wgrad_phases(unsigned long long* out11, int reset) {
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    if (hipMemcpyFromSymbol(out11, HIP_SYMBOL(wgrad_phase_ctr), 15 * sizeof(unsigned long long)) != hipSuccess) return 1;
    if (reset) {
        const unsigned long long z[15] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, ~0ull, 0, 0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(wgrad_phase_ctr), z, sizeof(z)) != hipSuccess) return 1;
    }
    return 0;
}
#endif
