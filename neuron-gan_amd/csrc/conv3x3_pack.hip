// Weight packing for the 3x3 convolution kernels (include/ngan.h: ngan_conv3x3_pack_weights, _pack_many, _packed_floats,
// _pack_elements): OIHW fp32 parameters -> MFMA-fragment order, pre-multiplied by the equalised-LR constant
// (/root/reference/models.py:195-204), in the layout of each kernel family ("precision code" 0..5, ngan_conv3x3_algorithm).
#include "conv3x3_internal.h"

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
    if (e.precision == 5) { reinterpret_cast<__bf16*>(e.dst)[idx] = bf16_weight(e.src, Cout, Cin, mode, e.scale, idx); return; }
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
long ngan::conv3x3_bf16x3_elements(int K, int N) {
    if (K <= 0 || N <= 0 || N % 16 || (K != 16 && K % 32)) return 0;
    return (long)(K == 16 ? 5 : 9 * (K / 32)) * (N / 16) * 2 * 64 * 8;
}

extern "C" long ngan_conv3x3_pack_elements(int Cout, int Cin, int mode, int precision) {
    if (Cout <= 0 || Cin <= 0 || Cout % 16 || Cin % 16) return 0;
    if (precision == 0) return 9L * Cin * Cout;
    if (precision == 4) return 16L * Cin * Cout;
    const int K = mode == 0 ? Cin : Cout, N = mode == 0 ? Cout : Cin;
    if (precision == 5) return ngan::conv3x3_bf16_elements(K, N);
    if (precision == 2) return K == 16 ? ngan::conv3x3_bf16x3_elements(32, N) : 0;      // K = 16 padded to 32 (mid kernel)
    if (precision == 3) return mode == 0 && ngan::conv3x3_bf16x3_elements(K, N) ? 4 * ngan::conv3x3_bf16x3_elements(K, N) + 9L * K * N : 0;   // folded bilinear
    return ngan::conv3x3_bf16x3_elements(K, N);
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
    if (precision == 5) {
        const long e0 = ngan::conv3x3_bf16_elements(Cin, Cout), e1 = ngan::conv3x3_bf16_elements(Cout, Cin);      // forward / flipped orientation
        return ((e0 > e1 ? e0 : e1) + 1) / 2;
    }
    if (precision == 2) {
        const long p0 = Cin == 16 ? ngan::conv3x3_bf16x3_elements(32, Cout) : 0, p1 = Cout == 16 ? ngan::conv3x3_bf16x3_elements(32, Cin) : 0;
        return ((p0 > p1 ? p0 : p1) + 1) / 2;
    }
    if (precision == 3) return ngan::conv3x3_bf16x3_elements(Cin, Cout) ? 2 * ngan::conv3x3_bf16x3_elements(Cin, Cout) + 9L * Cin * Cout : 0;
    const long e0 = ngan::conv3x3_bf16x3_elements(Cin, Cout), e1 = ngan::conv3x3_bf16x3_elements(Cout, Cin);   // forward / flipped orientation
    return ((e0 > e1 ? e0 : e1) + 1) / 2;
}

extern "C" int ngan_conv3x3_pack_weights(const float* w_oihw, float* packed, int Cout, int Cin, int mode, float scale,
                                         int precision, void* stream) {
    NGAN_REQUIRE(w_oihw && packed, NGAN_ERR_ARG, "conv3x3_pack_weights: null pointer");
    NGAN_REQUIRE(Cout > 0 && Cin > 0 && Cout % 16 == 0 && Cin % 16 == 0, NGAN_ERR_SHAPE,
                 "conv3x3_pack_weights: Cin=%d, Cout=%d must be positive multiples of 16", Cin, Cout);
    NGAN_REQUIRE(mode == 0 || mode == 1, NGAN_ERR_ARG, "conv3x3_pack_weights: mode %d", mode);
    NGAN_REQUIRE(precision >= 0 && precision <= 5, NGAN_ERR_ARG, "conv3x3_pack_weights: precision %d", precision);
    if (precision == 5) return ngan::conv3x3_bf16_pack_launch(w_oihw, packed, Cout, Cin, mode, scale, (hipStream_t)stream);
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
        const long tot = precision == 2 ? ngan::conv3x3_bf16x3_elements(32, N) : ngan::conv3x3_bf16x3_elements(K, N);
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
