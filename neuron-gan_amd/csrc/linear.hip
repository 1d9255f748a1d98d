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
__global__ __launch_bounds__(256) void linear_fwd_kernel(const float* __restrict__ z, const float* __restrict__ Wt,
                                                         float* __restrict__ y, float* __restrict__ rn, int B, int K,
                                                         int S, int C, float scale, float slope, float eps) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* z_l = sm;                 // BCH * K   (row b, k contiguous)
    float* out_l = sm + BCH * K;     // BCH * C
    float* r_l = out_l + BCH * C;    // BCH
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, q = lane >> 4;
    const int p = blockIdx.x, b0 = blockIdx.y * BCH;
    const int nb = min(BCH, B - b0);
    for (int e = tid; e < BCH * K; e += 256) {
        const int bb = e / K;
        z_l[e] = bb < nb ? z[(long)(b0 + bb) * K + (e - bb * K)] * scale : 0.f;   // weight_scale * x, models.py:241
    }
    __syncthreads();
    const int ntile = (C + 15) >> 4, ksteps = K >> 4;
    for (int ct = wave; ct < ntile; ct += 4) {
        const int c = ct * 16 + j;
        const float* wrow = Wt + ((long)min(c, C - 1) * S + p) * K + q * 4;
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int s = 0; s < ksteps; ++s) {
            const float4 wv = ld4(wrow + s * 16);                      // W[row c][16s + 4q .. +3]
            const float4 zv = ld4(z_l + j * K + s * 16 + q * 4);       // z[sample j][16s + 4q .. +3]
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(zv.x, wv.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(zv.y, wv.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(zv.z, wv.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(zv.w, wv.w, acc, 0, 0, 0);
        }
        if (c < C) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = acc[r];
                out_l[(4 * q + r) * C + c] = v > 0.f ? v : slope * v;
            }
        }
    }
    __syncthreads();
    if (tid < nb) {
        float ss = 0.f;
        for (int c = 0; c < C; ++c) ss = fmaf(out_l[tid * C + c], out_l[tid * C + c], ss);
        const float r = sqrtf(ss / (float)C + eps);
        r_l[tid] = r;
        rn[(long)(b0 + tid) * S + p] = r;
    }
    __syncthreads();
    for (int e = tid; e < nb * C; e += 256) {
        const int bb = e / C, c = e - bb * C;
        y[((long)(b0 + bb) * S + p) * C + c] = out_l[e] / r_l[bb];
    }
}

// gW[c*S+p][k] = scale * sum_b gc[b][p][c] * z[b][k]; one thread per (row, k-quad)
__global__ __launch_bounds__(256) void linear_wgrad_kernel(const float* __restrict__ z, const float* __restrict__ gc,
                                                           float* __restrict__ gW, int B, int K, int S, int C, float scale) {
    const int K4 = K >> 2;
    const long total = (long)C * S * K4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int k4 = (int)(i % K4);
        const long j = i / K4;
        const int p = (int)(j % S), c = (int)(j / S);
        float4 acc = f4zero();
        for (int b = 0; b < B; ++b) acc = f4fma(ld4(z + (long)b * K + k4 * 4), gc[((long)b * S + p) * C + c], acc);
        st4(gW + j * K + k4 * 4, f4scale(acc, scale));
    }
}

// gz[b][k] = scale * sum_j gc[b][j'] * W[j][k]; one block per (b, 256-wide k slab)
__global__ __launch_bounds__(256) void linear_dgrad_kernel(const float* __restrict__ gc, const float* __restrict__ Wt,
                                                           float* __restrict__ gz, int B, int K, int S, int C, float scale) {
    const int b = blockIdx.y;
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= K) return;
    float acc = 0.f;
    for (int c = 0; c < C; ++c)
        for (int p = 0; p < S; ++p) acc = fmaf(gc[((long)b * S + p) * C + c], Wt[((long)c * S + p) * K + k], acc);
    gz[(long)b * K + k] = acc * scale;
}

// ---- critic head ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void final_dot_fwd_kernel(const float* __restrict__ y, const float* __restrict__ W,
                                                            const float* __restrict__ bias, float* __restrict__ out,
                                                            int S2, int C, float scale) {
    __shared__ float red[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    const long n = (long)S2 * C;
    const float* yb = y + (long)b * n;
    float s = 0.f;
    for (long e = tid; e < n; e += 256) {
        const int c = (int)(e % C);
        const int p = (int)(e / C);
        s = fmaf(yb[e], W[(long)c * S2 + p], s);
    }
    s = group_sum<64>(s);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) out[b] = ((red[0] + red[1]) + (red[2] + red[3])) * scale + (bias ? bias[0] : 0.f);
}

__global__ void final_dot_dx_kernel(const float* __restrict__ go, const float* __restrict__ W, float* __restrict__ gy,
                                    int B, int S2, int C, float scale) {
    const long n = (long)S2 * C, total = (long)B * n;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int b = (int)(i / n);
        const long e = i - (long)b * n;
        const int c = (int)(e % C);
        const int p = (int)(e / C);
        gy[i] = scale * go[b] * W[(long)c * S2 + p];
    }
}

__global__ void final_dot_dw_kernel(const float* __restrict__ y, const float* __restrict__ go, float* __restrict__ gW,
                                    float* __restrict__ gb, int B, int S2, int C, float scale) {
    const long n = (long)S2 * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int p = (int)(i % S2);
        const int c = (int)(i / S2);
        float s = 0.f;
        for (int b = 0; b < B; ++b) s = fmaf(go[b], y[(long)b * n + (long)p * C + c], s);
        gW[i] = s * scale;
    }
    if (gb && blockIdx.x == 0 && threadIdx.x == 0) {
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += go[b];
        gb[0] = s;
    }
}

int ew_blocks(long n) {
    long b = (n + 255) / 256;
    return (int)(b < 4096 ? (b < 1 ? 1 : b) : 4096);
}

}  // namespace

extern "C" int ngan_linear_lrelu_pn_fwd(const float* z, const float* Wt, float* y, float* rnorm, int B, int K, int S, int C,
                                        float scale, float slope, float eps, void* stream) {
    NGAN_REQUIRE(z && Wt && y && rnorm, NGAN_ERR_ARG, "linear_lrelu_pn_fwd: null pointer");
    NGAN_REQUIRE(B > 0 && K > 0 && K % 16 == 0 && S > 0 && C > 0, NGAN_ERR_SHAPE, "linear_lrelu_pn_fwd: B=%d K=%d (multiple of 16) S=%d C=%d unsupported",
                 B, K, S, C);
    const size_t lds = (size_t)(BCH * K + BCH * C + BCH) * sizeof(float);
    NGAN_REQUIRE(lds <= 64 * 1024, NGAN_ERR_SHAPE, "linear_lrelu_pn_fwd: K=%d C=%d need %zu B of LDS", K, C, lds);
    hipLaunchKernelGGL(linear_fwd_kernel, dim3(S, ngan::ceil_div(B, BCH)), dim3(256), lds, (hipStream_t)stream, z, Wt, y, rnorm,
                       B, K, S, C, scale, slope, eps);
    return ngan::launch_status("ngan_linear_lrelu_pn_fwd");
}

extern "C" int ngan_linear_wgrad(const float* z, const float* gc, float* gW, int B, int K, int S, int C, float scale, void* stream) {
    NGAN_REQUIRE(z && gc && gW, NGAN_ERR_ARG, "linear_wgrad: null pointer");
    NGAN_REQUIRE(B > 0 && K > 0 && K % 4 == 0 && S > 0 && C > 0, NGAN_ERR_SHAPE, "linear_wgrad: B=%d K=%d S=%d C=%d unsupported", B, K, S, C);
    hipLaunchKernelGGL(linear_wgrad_kernel, dim3(ew_blocks((long)C * S * (K / 4))), dim3(256), 0, (hipStream_t)stream, z, gc, gW,
                       B, K, S, C, scale);
    return ngan::launch_status("ngan_linear_wgrad");
}

extern "C" int ngan_linear_dgrad(const float* gc, const float* Wt, float* gz, int B, int K, int S, int C, float scale, void* stream) {
    NGAN_REQUIRE(gc && Wt && gz, NGAN_ERR_ARG, "linear_dgrad: null pointer");
    NGAN_REQUIRE(B > 0 && K > 0 && S > 0 && C > 0, NGAN_ERR_SHAPE, "linear_dgrad: B=%d K=%d S=%d C=%d unsupported", B, K, S, C);
    hipLaunchKernelGGL(linear_dgrad_kernel, dim3(ngan::ceil_div(K, 256), B), dim3(256), 0, (hipStream_t)stream, gc, Wt, gz, B, K, S, C, scale);
    return ngan::launch_status("ngan_linear_dgrad");
}

extern "C" int ngan_final_dot_fwd(const float* y, const float* W, const float* bias, float* out, int B, int S2, int C, float scale,
                                  void* stream) {
    NGAN_REQUIRE(y && W && out && B > 0 && S2 > 0 && C > 0, NGAN_ERR_ARG, "final_dot_fwd: bad argument");
    hipLaunchKernelGGL(final_dot_fwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, y, W, bias, out, S2, C, scale);
    return ngan::launch_status("ngan_final_dot_fwd");
}

extern "C" int ngan_final_dot_dx(const float* go, const float* W, float* gy, int B, int S2, int C, float scale, void* stream) {
    NGAN_REQUIRE(go && W && gy && B > 0 && S2 > 0 && C > 0, NGAN_ERR_ARG, "final_dot_dx: bad argument");
    hipLaunchKernelGGL(final_dot_dx_kernel, dim3(ew_blocks((long)B * S2 * C)), dim3(256), 0, (hipStream_t)stream, go, W, gy, B, S2, C, scale);
    return ngan::launch_status("ngan_final_dot_dx");
}

extern "C" int ngan_final_dot_dw(const float* y, const float* go, float* gW, float* gb, int B, int S2, int C, float scale, void* stream) {
    NGAN_REQUIRE(y && go && gW && B > 0 && S2 > 0 && C > 0, NGAN_ERR_ARG, "final_dot_dw: bad argument");
    hipLaunchKernelGGL(final_dot_dw_kernel, dim3(ew_blocks((long)S2 * C)), dim3(256), 0, (hipStream_t)stream, y, go, gW, gb, B, S2, C, scale);
    return ngan::launch_status("ngan_final_dot_dw");
}
