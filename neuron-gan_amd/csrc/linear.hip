// Generator stem (Linear_normalized -> Unflatten -> LeakyReLU -> PixelNorm, /root/reference/models.py:299-311)
// and critic head (Conv2d_normalized(C, 1, (S,S), padding 0) -> Flatten, models.py:485-490).
// Both are small-M contractions: the work is reading the weight once (67 MB for the default stem), so the
// kernels stream weight rows with 16-byte loads straight into registers (no LDS round trip for the operand
// that is read once) and keep the tiny activation operand in LDS.
#include "ngan_common.h"

namespace {

constexpr int BCH = 16;  // batch rows handled per block

// One block per output pixel p and batch chunk of 16.  out[b][c] = sum_k z[b][k] * W[c*S + p][k] is a 16 x C x K GEMM:
// v_mfma_f32_16x16x4_f32 with A = z (16 samples x 4 k, from LDS) and B = W^T (4 k x 16 weight rows, streamed from HBM
// straight into registers, 16 B per lane; the k-order inside a 16-wide group is permuted identically on both operands).
// The accumulator holds sample 4q+r of weight row (channel) l & 15; LeakyReLU + PixelNorm over the C channels of the
// pixel then go through LDS.
template <typename T>
__global__ __launch_bounds__(256) void linear_fwd_kernel(const float* __restrict__ z, const float* __restrict__ Wt,
                                                         T* __restrict__ y, float* __restrict__ rn, int B, int K,
                                                         int S, int C, float scale, float slope, float eps) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    // rows of K + 4 floats: the 16 lanes of a ds_read_b128 phase read 16 different samples at the same k, and a pitch of
    // K floats would put all of them in the same banks (16-way conflict)
    const int ZP = K + 4, CP = C + 4;
    float* z_l = sm;                 // BCH * ZP   (row b, k contiguous)
    float* out_l = sm + BCH * ZP;    // BCH * CP
    float* r_l = out_l + BCH * CP;   // BCH
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, q = lane >> 4;
    const int p = blockIdx.x, b0 = blockIdx.y * BCH;
    const int nb = min(BCH, B - b0);
    for (int bb = tid >> 4; bb < BCH; bb += 16)
        for (int k = (tid & 15) * 4; k < K; k += 64) {
            const float4 v = bb < nb ? ld4(z + (long)(b0 + bb) * K + k) : f4zero();
            st4(z_l + bb * ZP + k, f4scale(v, scale));                            // weight_scale * x, models.py:241
        }
    __syncthreads();
    const int ntile = (C + 15) >> 4, ksteps = K >> 4;
    for (int ct = wave; ct < ntile; ct += 4) {
        const int c = ct * 16 + j;
        const float* wrow = Wt + ((long)min(c, C - 1) * S + p) * K + q * 4;
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
        // the weight row is read once from HBM: keep UNR 16-byte loads per lane in flight (one load per MFMA group left the
        // wave waiting a full memory round trip every 4 MFMAs: 45 us for the 67 MB default stem instead of ~15)
        constexpr int UNR = 16;
        int s = 0;
        for (; s + UNR <= ksteps; s += UNR) {
            float4 wv[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) wv[u] = ld4(wrow + (s + u) * 16);     // W[row c][16(s+u) + 4q .. +3]
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const float4 zv = ld4(z_l + j * ZP + (s + u) * 16 + q * 4);       // z[sample j][16(s+u) + 4q .. +3]
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(zv.x, wv[u].x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(zv.y, wv[u].y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(zv.z, wv[u].z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(zv.w, wv[u].w, acc, 0, 0, 0);
            }
        }
        for (; s < ksteps; ++s) {
            const float4 wv = ld4(wrow + s * 16);
            const float4 zv = ld4(z_l + j * ZP + s * 16 + q * 4);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(zv.x, wv.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(zv.y, wv.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(zv.z, wv.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(zv.w, wv.w, acc, 0, 0, 0);
        }
        if (c < C) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = acc[r];
                out_l[(4 * q + r) * CP + c] = v > 0.f ? v : slope * v;
            }
        }
    }
    __syncthreads();
    {   // PixelNorm over the C channels of each sample: 16 lanes per sample
        const int bb = tid >> 4, part = tid & 15;
        float ss = 0.f;
        for (int c = part; c < C; c += 16) ss = fmaf(out_l[bb * CP + c], out_l[bb * CP + c], ss);
        ss = group_sum<16>(ss);
        if (part == 0 && bb < nb) {
            const float r = sqrtf(ss / (float)C + eps);
            r_l[bb] = r;
            rn[(long)(b0 + bb) * S + p] = r;
        }
    }
    __syncthreads();
    for (int bb = tid >> 4; bb < nb; bb += 16) {
        const float inv = 1.0f / r_l[bb];
        for (int c = tid & 15; c < C; c += 16) sta1(y + ((long)(b0 + bb) * S + p) * C + c, out_l[bb * CP + c] * inv);
    }
}

// gW[c*S+p][k] = scale * sum_b gc[b][p][c] * z[b][k].  The output (67 MB for the default stem) is written once; a thread owns one
// k-quad and keeps z[b][k-quad] for a chunk of WB samples in registers, a wave walks weight rows, so gc[b][row] is the same
// for all of its lanes (one broadcast load) and a row costs WB x 4 FMAs per lane and one 16-byte store.
constexpr int WB = 16;

template <typename T>
__global__ __launch_bounds__(256) void linear_wgrad_kernel(const float* __restrict__ z, const T* __restrict__ gc,
                                                           float* __restrict__ gW, int B, int K, int S, int C, float scale,
                                                           int rows_per_block) {
    const int K4 = K >> 2;
    const int tid = threadIdx.x;
    const int wpr = (K4 + 63) >> 6;                     // waves per weight row
    const int wave = tid >> 6, lane = tid & 63;
    const int k4 = (wave % wpr) * 64 + lane;            // this lane's k-quad
    const int rsub = wave / wpr, rstep = 4 / wpr;       // rows are dealt to the block's wave groups
    const bool act = k4 < K4 && rsub < rstep;
    const long rows = (long)C * S;
    const long r0 = (long)blockIdx.x * rows_per_block;
    const long r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
    for (int b0 = 0; b0 < B; b0 += WB) {
        float4 zr[WB];
#pragma unroll
        for (int i = 0; i < WB; ++i) zr[i] = (act && b0 + i < B) ? ld4(z + (long)(b0 + i) * K + k4 * 4) : f4zero();
        for (long j = r0 + rsub; j < r1; j += rstep) {
            const int p = (int)(j % S), c = (int)(j / S);
            float4 acc = f4zero();
#pragma unroll
            for (int i = 0; i < WB; ++i) {
                const float gv = (b0 + i < B) ? lda1(gc + ((long)(b0 + i) * S + p) * C + c) : 0.f;     // wave-uniform address
                acc = f4fma(zr[i], gv, acc);
            }
            if (act) {
                float* o = gW + j * K + k4 * 4;
                if (b0 == 0) st4(o, f4scale(acc, scale));
                else st4(o, f4fma(acc, scale, ld4(o)));
            }
        }
    }
}

// MFMA form of the same contraction for K <= 512 (the contraction index is the SAMPLE, any B in one pass -- the data-parallel
// stem exchange forms the full-batch gradient from world*B gathered samples): a wave owns 16 weight rows and all K columns,
// A = z^T (16 k-columns x 4 samples), B = gc (4 samples x 16 rows), so a lane ends with 4 consecutive k of one row: one
// 16-byte store.  z (B x K) and gc are tiny and stay in L1/L2; the 4*C*S*K-byte result is written exactly once.
// ADAM: the gradient is never stored -- the lane applies Adam to its 4 * NT elements of the parameter and its moments (p, m, v are
// the tensor's slices of the flat buffers; step / hyper as in ngan_adam_step).  The stem holds 16.8 M of the generator's 17.1 M
// parameters: the stored form costs a 67 MB zero fill, a 67 MB read-modify-write here and a 67 MB read in the Adam kernel.
struct StemAdam { float* p; float* m; float* v; const float* hyper; const float* step; };
template <typename T, int NT, bool ADAM = false>
__global__ __launch_bounds__(256) void linear_wgrad_mfma_kernel(const float* __restrict__ z, const T* __restrict__ gc,
                                                                float* __restrict__ gW, int B, int K, int S, int C, float scale,
                                                                int accumulate, StemAdam ad = StemAdam{}) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = lane & 15, kq = lane >> 4;
    const long rows = (long)C * S;
    const long row = ((long)blockIdx.x * 4 + wave) * 16 + m;            // this lane's weight row (as B-operand column)
    const bool rok = row < rows;
    const int p = (int)(row % S), c = (int)(row / S);
    const int nt = K >> 4;
    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int b0 = 0; b0 < B; b0 += 4) {
        const int b = b0 + kq;
        const bool bok = b < B;
        const float gv = (rok && bok) ? lda1(gc + ((long)b * S + p) * C + c) : 0.f;
        const float* zr = z + (long)(bok ? b : 0) * K + m;
        float zv[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) zv[t] = (bok && t < nt) ? zr[t * 16] : 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t)
            if (t < nt) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(zv[t], gv, acc[t], 0, 0, 0);
    }
    if (ADAM) {
        if (rok) {
            const AdamCoef k = adam_coef(ad.hyper, ad.step[0]);
            const long o = row * K + kq * 4;
#pragma unroll
            for (int t = 0; t < NT; ++t)
                if (t < nt) {
                    float4 pv = ld4(ad.p + o + t * 16), mv = ld4(ad.m + o + t * 16), vv = ld4(ad.v + o + t * 16);
                    adam_update(k, acc[t][0] * scale, pv.x, mv.x, vv.x);
                    adam_update(k, acc[t][1] * scale, pv.y, mv.y, vv.y);
                    adam_update(k, acc[t][2] * scale, pv.z, mv.z, vv.z);
                    adam_update(k, acc[t][3] * scale, pv.w, mv.w, vv.w);
                    st4(ad.m + o + t * 16, mv);
                    st4(ad.v + o + t * 16, vv);
                    st4(ad.p + o + t * 16, pv);
                }
        }
        return;
    }
    if (rok) {
        float* o = gW + row * K + kq * 4;
#pragma unroll
        for (int t = 0; t < NT; ++t)
            if (t < nt) {
                float4 v = make_float4(acc[t][0] * scale, acc[t][1] * scale, acc[t][2] * scale, acc[t][3] * scale);
                if (accumulate) v = f4add(v, ld4(o + t * 16));      // gW += ...: adds straight into the parameter's gradient buffer
                st4(o + t * 16, v);
            }
    }
}

// gz[b][k] = scale * sum_j gc[b][j'] * W[j][k]; one block per (b, 256-wide k slab)
template <typename T>
__global__ __launch_bounds__(256) void linear_dgrad_kernel(const T* __restrict__ gc, const float* __restrict__ Wt,
                                                           float* __restrict__ gz, int B, int K, int S, int C, float scale) {
    const int b = blockIdx.y;
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= K) return;
    float acc = 0.f;
    for (int c = 0; c < C; ++c)
        for (int p = 0; p < S; ++p) acc = fmaf(lda1(gc + ((long)b * S + p) * C + c), Wt[((long)c * S + p) * K + k], acc);
    gz[(long)b * K + k] = acc * scale;
}

// ---- critic head ------------------------------------------------------------------------------------------
// One block per sample.  y is (S2, C) channels-last, W is (C, S2): the block first copies W into LDS transposed ([p][c], pitch
// C + 1) with coalesced global reads, then both operands of the dot product are contiguous.  (Gathering W with stride S2 per
// lane instead costs one cache line per lane per load: 23 us for a 2 MB input.)
template <typename T>
__global__ __launch_bounds__(1024) void final_dot_fwd_kernel(const T* __restrict__ y, const float* __restrict__ W,
                                                             const float* __restrict__ bias, float* __restrict__ out,
                                                             int S2, int C, float scale, int use_lds, int lg_s2, int lg_c) {
    // lg_s2 / lg_c: log2 of S2 / C when they are powers of two (the usual 16 x 16 x 128 head), else -1.  With 16 - 32 blocks in
    // flight the block's own instruction count is the launch's duration: 40 runtime integer divisions per thread were 8 of its 14 us
    extern __shared__ float wt[];
    __shared__ float red[16];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int n = S2 * C;
    const T* yb = y + (long)b * n;
    float s = 0.f;
    if (use_lds) {
        const int CP = C + 1;
        // this sample's activations first: up to 32 loads per thread in flight while W is fetched and transposed (with the loads
        // behind the transposition the block sat through two dependent rounds of memory latency: 14 us for 32 K elements)
        constexpr int PRE = 32;
        float yv[PRE];
#pragma unroll
        for (int u = 0; u < PRE; ++u) yv[u] = tid + u * 1024 < n ? lda1(yb + tid + u * 1024) : 0.f;
        if ((S2 & 3) == 0) {       // four positions of one channel per 16-byte load
            for (int e = tid * 4; e < n; e += 4096) {
                const int c = lg_s2 >= 0 ? e >> lg_s2 : e / S2, p = e - c * S2;
                const float4 w = ld4(W + e);
                wt[p * CP + c] = w.x; wt[(p + 1) * CP + c] = w.y; wt[(p + 2) * CP + c] = w.z; wt[(p + 3) * CP + c] = w.w;
            }
        } else {
            for (int e = tid; e < n; e += 1024) {
                const int c = e / S2, p = e - c * S2;
                wt[p * CP + c] = W[e];
            }
        }
        __syncthreads();
        float a[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < PRE; ++u) {
            const int i = tid + u * 1024;
            if (i < n) {
                const int p0 = lg_c >= 0 ? i >> lg_c : i / C, c0 = i - p0 * C;
                a[u & 3] = fmaf(yv[u], wt[p0 * CP + c0], a[u & 3]);
            }
        }
        for (int e = tid + PRE * 1024; e < n; e += 1024) {
            const int p0 = e / C, c0 = e - p0 * C;
            a[0] = fmaf(lda1(yb + e), wt[p0 * CP + c0], a[0]);
        }
        s = (a[0] + a[1]) + (a[2] + a[3]);
    } else {
        for (int e = tid; e < n; e += 1024) {
            const int p = e / C, c = e - p * C;
            s = fmaf(lda1(yb + e), W[(long)c * S2 + p], s);
        }
    }
    s = group_sum<64>(s);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) t += red[i];
        out[b] = t * scale + (bias ? bias[0] : 0.f);
    }
}

template <typename T>
__global__ void final_dot_dx_kernel(const float* __restrict__ go, const float* __restrict__ W, T* __restrict__ gy,
                                    int B, int S2, int C, float scale) {
    const long n = (long)S2 * C, total = (long)B * n;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int b = (int)(i / n);
        const long e = i - (long)b * n;
        const int c = (int)(e % C);
        const int p = (int)(e / C);
        sta1(gy + i, scale * go[b] * W[(long)c * S2 + p]);
    }
}

template <typename T>
__global__ void final_dot_dw_kernel(const T* __restrict__ y, const float* __restrict__ go, float* __restrict__ gW,
                                    float* __restrict__ gb, int B, int S2, int C, float scale, int accumulate) {
    const long n = (long)S2 * C;                          // accumulate: bit 0 gW += , bit 1 gb +=
    // a thread per element of y's (p, c) order: the B reads of a wave are contiguous (the gradient's own (c, p) order made every
    // lane fetch a line of its own); the 4-byte stores scatter instead, once
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int p = (int)(i / C);
        float s0 = 0.f, s1 = 0.f;
        int b = 0;
        for (; b + 1 < B; b += 2) {
            s0 = fmaf(go[b], lda1(y + (long)b * n + i), s0);
            s1 = fmaf(go[b + 1], lda1(y + (long)(b + 1) * n + i), s1);
        }
        if (b < B) s0 = fmaf(go[b], lda1(y + (long)b * n + i), s0);
        const float s = s0 + s1;
        float* o = gW + (long)c * S2 + p;
        *o = (accumulate & 1) ? *o + s * scale : s * scale;
    }
    if (gb && blockIdx.x == 0 && threadIdx.x == 0) {
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += go[b];
        gb[0] = (accumulate & 2) ? gb[0] + s : s;
    }
}

int ew_blocks(long n) {
    long b = (n + 255) / 256;
    return (int)(b < 4096 ? (b < 1 ? 1 : b) : 4096);
}

}  // namespace

#define BF(p) reinterpret_cast<const __bf16*>(p)
#define BFM(p) reinterpret_cast<__bf16*>(p)

// T: storage type of the activation operand (y / gc): float, or __bf16 for the "bf16 activation storage" entry points.  z, the
// weight and everything derived from them stay fp32.
template <typename T>
static int linear_fwd_impl(const float* z, const float* Wt, T* y, float* rnorm, int B, int K, int S, int C, float scale, float slope, float eps,
                           void* stream) {
    NGAN_REQUIRE(z && Wt && y && rnorm, NGAN_ERR_ARG, "linear_lrelu_pn_fwd: null pointer");
    NGAN_REQUIRE(B > 0 && K > 0 && K % 16 == 0 && S > 0 && C > 0, NGAN_ERR_SHAPE, "linear_lrelu_pn_fwd: B=%d K=%d (multiple of 16) S=%d C=%d unsupported",
                 B, K, S, C);
    NGAN_REQUIRE(K % 4 == 0, NGAN_ERR_SHAPE, "linear_lrelu_pn_fwd: K=%d must be a multiple of 4", K);
    const size_t lds = (size_t)(BCH * (K + 4) + BCH * (C + 4) + BCH) * sizeof(float);
    NGAN_REQUIRE(lds <= 160 * 1024, NGAN_ERR_SHAPE, "linear_lrelu_pn_fwd: K=%d C=%d need %zu B of LDS", K, C, lds);
    if (lds > 64 * 1024) {      // wide stems (the 1024-channel presets): more than the default dynamic-LDS limit of a launch
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(linear_fwd_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        NGAN_REQUIRE(e == hipSuccess, (int)e, "linear_lrelu_pn_fwd: cannot raise the dynamic LDS limit: %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL(linear_fwd_kernel<T>, dim3(S, ngan::ceil_div(B, BCH)), dim3(256), lds, (hipStream_t)stream, z, Wt, y, rnorm,
                       B, K, S, C, scale, slope, eps);
    return ngan::launch_status("ngan_linear_lrelu_pn_fwd");
}
extern "C" int ngan_linear_lrelu_pn_fwd(const float* z, const float* Wt, float* y, float* rnorm, int B, int K, int S, int C,
                                        float scale, float slope, float eps, void* stream) {
    return linear_fwd_impl<float>(z, Wt, y, rnorm, B, K, S, C, scale, slope, eps, stream);
}
extern "C" int ngan_bf16_linear_lrelu_pn_fwd(const float* z, const float* Wt, ngan_bf16* y, float* rnorm, int B, int K, int S, int C,
                                             float scale, float slope, float eps, void* stream) {
    return linear_fwd_impl<__bf16>(z, Wt, BFM(y), rnorm, B, K, S, C, scale, slope, eps, stream);
}

template <typename T>
static int linear_wgrad_impl(const float* z, const T* gc, float* gW, int B, int K, int S, int C, float scale, int accumulate, void* stream) {
    NGAN_REQUIRE(z && gc && gW, NGAN_ERR_ARG, "linear_wgrad: null pointer");
    NGAN_REQUIRE(!accumulate || (K % 16 == 0 && K <= 512), NGAN_ERR_SHAPE, "linear_wgrad: accumulate needs K <= 512, a multiple of 16 (K=%d)", K);
    NGAN_REQUIRE(B > 0 && K > 0 && K % 4 == 0 && S > 0 && C > 0, NGAN_ERR_SHAPE, "linear_wgrad: B=%d K=%d S=%d C=%d unsupported", B, K, S, C);
    NGAN_REQUIRE(K / 4 <= 256, NGAN_ERR_SHAPE, "linear_wgrad: K=%d must be at most 1024", K);
    const long rows = (long)C * S;
    if (K % 16 == 0 && K <= 512) {
        const dim3 grid(ngan::ceil_div(rows, 64)), block(256);
        if (K <= 128) hipLaunchKernelGGL((linear_wgrad_mfma_kernel<T, 8, false>), grid, block, 0, (hipStream_t)stream, z, gc, gW, B, K, S, C, scale, accumulate, StemAdam{});
        else hipLaunchKernelGGL((linear_wgrad_mfma_kernel<T, 32, false>), grid, block, 0, (hipStream_t)stream, z, gc, gW, B, K, S, C, scale, accumulate, StemAdam{});
        return ngan::launch_status("ngan_linear_wgrad(mfma)");
    }
    const int rpb = rows >= 4096 ? 16 : 4;
    hipLaunchKernelGGL(linear_wgrad_kernel<T>, dim3(ngan::ceil_div(rows, rpb)), dim3(256), 0, (hipStream_t)stream, z, gc, gW,
                       B, K, S, C, scale, rpb);
    return ngan::launch_status("ngan_linear_wgrad");
}
extern "C" int ngan_linear_wgrad_acc(const float* z, const float* gc, float* gW, int B, int K, int S, int C, float scale,
                                     int accumulate, void* stream) {
    return linear_wgrad_impl<float>(z, gc, gW, B, K, S, C, scale, accumulate, stream);
}
extern "C" int ngan_linear_wgrad(const float* z, const float* gc, float* gW, int B, int K, int S, int C, float scale, void* stream) {
    return linear_wgrad_impl<float>(z, gc, gW, B, K, S, C, scale, 0, stream);
}
extern "C" int ngan_bf16_linear_wgrad_acc(const float* z, const ngan_bf16* gc, float* gW, int B, int K, int S, int C, float scale,
                                          int accumulate, void* stream) {
    return linear_wgrad_impl<__bf16>(z, BF(gc), gW, B, K, S, C, scale, accumulate, stream);
}
extern "C" int ngan_bf16_linear_wgrad(const float* z, const ngan_bf16* gc, float* gW, int B, int K, int S, int C, float scale, void* stream) {
    return linear_wgrad_impl<__bf16>(z, BF(gc), gW, B, K, S, C, scale, 0, stream);
}

template <typename T>
static int linear_wgrad_adam_impl(const float* z, const T* gc, float* p, float* m, float* v, const float* seg_step,
                                  const float* hyper, int n_hyper, int B, int K, int S, int C, float scale, void* stream) {
    NGAN_REQUIRE(z && gc && p && m && v && seg_step && hyper, NGAN_ERR_ARG, "linear_wgrad_adam: null pointer");
    NGAN_REQUIRE(n_hyper == NGAN_ADAM_HYPER_FLOATS, NGAN_ERR_ARG, "linear_wgrad_adam: hyper holds %d floats, this library reads %d (include/ngan.h)",
                 n_hyper, NGAN_ADAM_HYPER_FLOATS);
    NGAN_REQUIRE(B > 0 && S > 0 && C > 0 && K > 0 && K % 16 == 0 && K <= 512, NGAN_ERR_SHAPE,
                 "linear_wgrad_adam: B=%d K=%d S=%d C=%d unsupported (K a multiple of 16, at most 512)", B, K, S, C);
    const dim3 grid(ngan::ceil_div((long)C * S, 64)), block(256);
    const StemAdam ad{p, m, v, hyper, seg_step};
    if (K <= 128) hipLaunchKernelGGL((linear_wgrad_mfma_kernel<T, 8, true>), grid, block, 0, (hipStream_t)stream, z, gc, nullptr, B, K, S, C, scale, 0, ad);
    else hipLaunchKernelGGL((linear_wgrad_mfma_kernel<T, 32, true>), grid, block, 0, (hipStream_t)stream, z, gc, nullptr, B, K, S, C, scale, 0, ad);
    return ngan::launch_status("ngan_linear_wgrad_adam");
}
extern "C" int ngan_linear_wgrad_adam(const float* z, const float* gc, float* p, float* m, float* v, const float* seg_step,
                                      const float* hyper, int n_hyper, int B, int K, int S, int C, float scale, void* stream) {
    return linear_wgrad_adam_impl<float>(z, gc, p, m, v, seg_step, hyper, n_hyper, B, K, S, C, scale, stream);
}
extern "C" int ngan_bf16_linear_wgrad_adam(const float* z, const ngan_bf16* gc, float* p, float* m, float* v, const float* seg_step,
                                           const float* hyper, int n_hyper, int B, int K, int S, int C, float scale, void* stream) {
    return linear_wgrad_adam_impl<__bf16>(z, BF(gc), p, m, v, seg_step, hyper, n_hyper, B, K, S, C, scale, stream);
}

template <typename T>
static int linear_dgrad_impl(const T* gc, const float* Wt, float* gz, int B, int K, int S, int C, float scale, void* stream) {
    NGAN_REQUIRE(gc && Wt && gz, NGAN_ERR_ARG, "linear_dgrad: null pointer");
    NGAN_REQUIRE(B > 0 && K > 0 && S > 0 && C > 0, NGAN_ERR_SHAPE, "linear_dgrad: B=%d K=%d S=%d C=%d unsupported", B, K, S, C);
    hipLaunchKernelGGL(linear_dgrad_kernel<T>, dim3(ngan::ceil_div(K, 256), B), dim3(256), 0, (hipStream_t)stream, gc, Wt, gz, B, K, S, C, scale);
    return ngan::launch_status("ngan_linear_dgrad");
}
extern "C" int ngan_linear_dgrad(const float* gc, const float* Wt, float* gz, int B, int K, int S, int C, float scale, void* stream) {
    return linear_dgrad_impl<float>(gc, Wt, gz, B, K, S, C, scale, stream);
}
extern "C" int ngan_bf16_linear_dgrad(const ngan_bf16* gc, const float* Wt, float* gz, int B, int K, int S, int C, float scale, void* stream) {
    return linear_dgrad_impl<__bf16>(BF(gc), Wt, gz, B, K, S, C, scale, stream);
}

template <typename T>
static int final_dot_fwd_impl(const T* y, const float* W, const float* bias, float* out, int B, int S2, int C, float scale, void* stream) {
    NGAN_REQUIRE(y && W && out && B > 0 && S2 > 0 && C > 0, NGAN_ERR_ARG, "final_dot_fwd: bad argument");
    const size_t lds = (size_t)S2 * (C + 1) * sizeof(float);
    const int use_lds = lds <= 150 * 1024;
    static bool attr_set = false;          // (one flag per instantiation, i.e. per kernel)
    if (use_lds && lds > 64 * 1024 && !attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(final_dot_fwd_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        NGAN_REQUIRE(e == hipSuccess, (int)e, "final_dot_fwd: cannot raise the dynamic LDS limit: %s", hipGetErrorString(e));
        attr_set = true;
    }
    auto lg = [](int v) { int l = 0; while ((1 << l) < v) ++l; return (1 << l) == v ? l : -1; };
    hipLaunchKernelGGL(final_dot_fwd_kernel<T>, dim3(B), dim3(1024), use_lds ? lds : 0, (hipStream_t)stream, y, W, bias, out, S2, C, scale, use_lds,
                       lg(S2), lg(C));
    return ngan::launch_status("ngan_final_dot_fwd");
}
extern "C" int ngan_final_dot_fwd(const float* y, const float* W, const float* bias, float* out, int B, int S2, int C, float scale,
                                  void* stream) {
    return final_dot_fwd_impl<float>(y, W, bias, out, B, S2, C, scale, stream);
}
extern "C" int ngan_bf16_final_dot_fwd(const ngan_bf16* y, const float* W, const float* bias, float* out, int B, int S2, int C, float scale,
                                       void* stream) {
    return final_dot_fwd_impl<__bf16>(BF(y), W, bias, out, B, S2, C, scale, stream);
}

template <typename T>
static int final_dot_dx_impl(const float* go, const float* W, T* gy, int B, int S2, int C, float scale, void* stream) {
    NGAN_REQUIRE(go && W && gy && B > 0 && S2 > 0 && C > 0, NGAN_ERR_ARG, "final_dot_dx: bad argument");
    hipLaunchKernelGGL(final_dot_dx_kernel<T>, dim3(ew_blocks((long)B * S2 * C)), dim3(256), 0, (hipStream_t)stream, go, W, gy, B, S2, C, scale);
    return ngan::launch_status("ngan_final_dot_dx");
}
extern "C" int ngan_final_dot_dx(const float* go, const float* W, float* gy, int B, int S2, int C, float scale, void* stream) {
    return final_dot_dx_impl<float>(go, W, gy, B, S2, C, scale, stream);
}
extern "C" int ngan_bf16_final_dot_dx(const float* go, const float* W, ngan_bf16* gy, int B, int S2, int C, float scale, void* stream) {
    return final_dot_dx_impl<__bf16>(go, W, BFM(gy), B, S2, C, scale, stream);
}

template <typename T>
static int final_dot_dw_impl(const T* y, const float* go, float* gW, float* gb, int B, int S2, int C, float scale, int accumulate, void* stream) {
    NGAN_REQUIRE(y && go && gW && B > 0 && S2 > 0 && C > 0, NGAN_ERR_ARG, "final_dot_dw: bad argument");
    hipLaunchKernelGGL(final_dot_dw_kernel<T>, dim3(ew_blocks((long)S2 * C)), dim3(256), 0, (hipStream_t)stream, y, go, gW, gb, B, S2, C, scale, accumulate);
    return ngan::launch_status("ngan_final_dot_dw");
}
extern "C" int ngan_final_dot_dw_acc(const float* y, const float* go, float* gW, float* gb, int B, int S2, int C, float scale, int accumulate,
                                     void* stream) {
    return final_dot_dw_impl<float>(y, go, gW, gb, B, S2, C, scale, accumulate, stream);
}
extern "C" int ngan_final_dot_dw(const float* y, const float* go, float* gW, float* gb, int B, int S2, int C, float scale, void* stream) {
    return final_dot_dw_impl<float>(y, go, gW, gb, B, S2, C, scale, 0, stream);
}
extern "C" int ngan_bf16_final_dot_dw_acc(const ngan_bf16* y, const float* go, float* gW, float* gb, int B, int S2, int C, float scale,
                                          int accumulate, void* stream) {
    return final_dot_dw_impl<__bf16>(BF(y), go, gW, gb, B, S2, C, scale, accumulate, stream);
}
extern "C" int ngan_bf16_final_dot_dw(const ngan_bf16* y, const float* go, float* gW, float* gb, int B, int S2, int C, float scale, void* stream) {
    return final_dot_dw_impl<__bf16>(BF(y), go, gW, gb, B, S2, C, scale, 0, stream);
}
