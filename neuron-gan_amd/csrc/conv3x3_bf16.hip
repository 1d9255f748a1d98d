// 3x3 convolution with bf16 ACTIVATION STORAGE and bf16 MFMA operands ("precision code 5", include/ngan.h): forward and -- with the
// flipped packed weights -- input gradient, for BASELINE.json's C2 "bf16" configuration.  The reference has no such mode
// (/root/reference/train.py:136-144 computes in the default dtype); what it replaces is the same Conv2d_normalized + resample + LeakyReLU
// + PixelNorm group as the fp32 kernels (/root/reference/models.py:203-204, 252-268).
//
// Arithmetic: activations are read as bf16, weights are fp32 masters rounded once to bf16 (scale folded in first) by the packing
// kernel, ONE v_mfma_f32_16x16x32_bf16 per product group (not the three of the split-bf16 mode), fp32 accumulation, fp32 epilogue
// (bias, LeakyReLU, PixelNorm statistics, PixelNorm backward, ToImage), round-to-nearest-even on the store.  The per-pixel norm is fp32.
//
// Roofline: at 2.5 PFLOP/s dense bf16 every layer of the default nets is HBM-bound (16 -> 16 at 512x512, batch 16: 8.5 us of matrix pipe
// against 268 MB = 42 us at the achievable 6.3 TB/s), so the kernel is built around its memory traffic, not the MFMA schedule:
//   * one workgroup = one output tile of PGT groups of 16 pixels.  N = 16 / 32 (the large-image layers): 8 x 32, 4 x 32 or 2 x 32 pixels
//     (the launcher shrinks the tile until the grid fills the chip; 4 x 16 for images at most 16 wide), the four waves split the
//     PIXEL groups; several workgroups per CU overlap each other's load / MFMA / store phases (11 - 22 KB of LDS, < 100 VGPRs).
//     N = 64 / 128 (a few thousand pixels, contraction up to 1152): 64- or 32-pixel tiles and the four waves split the OUTPUT
//     channels instead, so that every 1 KB weight fragment is fetched by exactly one wave of the workgroup (with the pixel split each
//     wave streamed all 288 KB of a 128 -> 128 layer: 26 us per launch, L2-bound); the per-pixel PixelNorm sums cross the waves
//     through 4 KB of LDS in a fixed order;
//   * the halo tile is staged once into LDS as it lies in memory -- [pixel][K] bf16, 16-byte chunks, coalesced 16-byte loads -- with the
//     chunk index XOR-swizzled by the pixel column so that the B-operand fragments (lane = pixel p, k-group q: ONE ds_read_b128 of 8
//     consecutive channels) are conflict-free on ds_read_b128's four lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ...
//     (MI355X_MICROARCH.md, LDS): chunk s of column X lives at s ^ 2*((X / (16/P)) & (P/2 - 1)), P = K/8 chunks per pixel;
//   * avg-pool 2x2 and bilinear x2 inputs are resampled while the tile is staged (fp32 blend of the bf16 sources, one rounding);
//   * "D^T" form as in the fp32 kernels: A = weights (16 output channels x 32 k), B = pixels, so a lane ends with 4 consecutive
//     output channels of one pixel: an 8-byte store, 16 lanes x 4 k-groups = one pixel row segment; the PixelNorm sums are
//     v_permlane swaps over the 4 k-group rows (sum_rows4);
//   * K = 16: a contraction step is a PAIR of taps (k-groups 0,1 -> tap 2s, k-groups 2,3 -> tap 2s + 1; the 10th tap has zero
//     weights): 5 MFMAs per 16 pixels x 16 outputs;  K = 32 KS: step = tap * KS + ks covers channels 32 ks .. 32 ks + 31 of one tap;
//   * weights: S * N/16 fragments of 1 KB.  At most 18 per wave: held in registers for the whole tile.  More: streamed L2 ->
//     registers through a ring 8 contraction steps ahead of their use (an L2 round trip is ~5 steps of 8 MFMAs);
//   * every global access goes through a buffer descriptor of the tile's image (32-bit offsets; padding, tile edges and unused
//     staging slots are out-of-range offsets: zeros on load, dropped on store -- no branch around any memory instruction);
//   * workgroup -> tile: bands per XCD (blockIdx & 7) so that neighbouring tiles' halos meet in one L2.
// Resample / epilogue / store mode are run-time (wave-uniform) branches, not template parameters: 16 (K, N) pairs x 4 tile shapes
// are already 64 instances (conv3x3_bf16_impl.h, instantiated per K in conv3x3_bf16_k*.hip).
#include "conv3x3_bf16_impl.h"

namespace {

__global__ void pack_weights_bf16_kernel(const float* __restrict__ w, __bf16* __restrict__ packed, int Cout, int Cin, int mode, float scale,
                                         long total) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < total) packed[idx] = bf16_weight(w, Cout, Cin, mode, scale, idx);
}

bool bf16_channels_ok(int c) { return c == 16 || c == 32 || c == 64 || c == 128; }

// tile shape (pixel groups of 16 per workgroup, narrow = 16-pixel-wide tiles).  N <= 32: 8 rows of 32 while that still gives two
// workgroups per CU, else 4, else 2 rows (4 x 16 for images at most 16 wide).  N >= 64: 64 pixels, 32 when that leaves CUs idle.
void pick_tile(int B, int H, int W, int N, int& pgt, bool& narrow) {
    narrow = W <= 16;
    if (N <= 32) {
        const long t8 = (long)B * ngan::ceil_div(H, 8) * ngan::ceil_div(W, 32), t4 = (long)B * ngan::ceil_div(H, 4) * ngan::ceil_div(W, 32);
        pgt = narrow ? 4 : t8 >= 512 ? 16 : t4 >= 384 ? 8 : 4;
        return;
    }
    const long t64 = narrow ? (long)B * ngan::ceil_div(H, 4) * ngan::ceil_div(W, 16) : (long)B * ngan::ceil_div(H, 2) * ngan::ceil_div(W, 32);
    pgt = t64 >= NGAN_DIAG_INT("NGAN_BF16_T64", 256) ? 4 : 2;      // (threshold: A/B in the diagnostic build)
}

}  // namespace

long ngan::conv3x3_bf16_elements(int K, int N) {
    if (!bf16_channels_ok(K) || !bf16_channels_ok(N)) return 0;
    return (long)(K == 16 ? 5 : 9 * (K / 32)) * (N / 16) * 512;
}

int ngan::conv3x3_bf16_pack_launch(const float* w, float* packed, int Cout, int Cin, int mode, float scale, hipStream_t s) {
    const long tot = conv3x3_bf16_elements(mode == 0 ? Cin : Cout, mode == 0 ? Cout : Cin);
    NGAN_REQUIRE(tot > 0, NGAN_ERR_SHAPE, "conv3x3_pack_weights: precision 5 (bf16) takes 16 / 32 / 64 / 128 channels (Cin=%d, Cout=%d)", Cin, Cout);
    hipLaunchKernelGGL(pack_weights_bf16_kernel, dim3(ngan::ceil_div(tot, 256)), dim3(256), 0, s, w, reinterpret_cast<__bf16*>(packed), Cout, Cin,
                       mode, scale, tot);
    return ngan::launch_status("ngan_conv3x3_pack_weights(bf16)");
}

int ngan::conv3x3_bf16_kernel_name(int B, int H, int W, int K, int N, char* buf, int len) {
    int pgt; bool narrow;
    pick_tile(B, H, W, N, pgt, narrow);
    snprintf(buf, len, "conv3x3_bf16_kernel<%d, %d, %d, %s, %s>", K, N, pgt, N >= 64 ? "true" : "false", narrow ? "true" : "false");
    return NGAN_OK;
}

extern "C" int ngan_bf16_conv3x3_fwd(const ngan_bf16* x, const float* packed, const float* bias, ngan_bf16* y, float* rnorm,
                                     const void* aux_in, const float* aux_rn, float* aux_out,
                                     int B, int H, int W, int K, int N, int resample, int epilogue, int out_mode,
                                     float slope, float eps, void* stream) {
    NGAN_REQUIRE(x && packed && (y || epilogue == EPI_TO_IMAGE), NGAN_ERR_ARG, "bf16_conv3x3_fwd: null pointer");
    NGAN_REQUIRE(B > 0 && H > 0 && W > 0, NGAN_ERR_SHAPE, "bf16_conv3x3_fwd: bad dims B=%d H=%d W=%d", B, H, W);
    NGAN_REQUIRE(bf16_channels_ok(K) && bf16_channels_ok(N), NGAN_ERR_SHAPE, "bf16_conv3x3_fwd: K=%d, N=%d must be 16 / 32 / 64 / 128", K, N);
    NGAN_REQUIRE(resample >= 0 && resample <= 2, NGAN_ERR_ARG, "bf16_conv3x3_fwd: resample %d", resample);
    NGAN_REQUIRE(epilogue >= EPI_NONE && epilogue <= EPI_TO_IMAGE, NGAN_ERR_ARG, "bf16_conv3x3_fwd: epilogue %d", epilogue);
    NGAN_REQUIRE(out_mode == 0 || (out_mode == 1 && (epilogue == EPI_NONE || epilogue == EPI_PN_BWD) && resample == 0), NGAN_ERR_ARG,
                 "bf16_conv3x3_fwd: out_mode %d needs epilogue 0 or 2 and resample 0", out_mode);
    NGAN_REQUIRE(epilogue != EPI_LRELU_PN || rnorm, NGAN_ERR_ARG, "bf16_conv3x3_fwd: epilogue 1 needs rnorm");
    NGAN_REQUIRE(epilogue != EPI_PN_BWD || (aux_in && aux_rn && resample == 0 && !bias), NGAN_ERR_ARG,
                 "bf16_conv3x3_fwd: epilogue 2 needs aux_in / aux_rn, no resampling and no bias");
    NGAN_REQUIRE(epilogue != EPI_TO_IMAGE || (aux_in && aux_out && (!y || rnorm) && N <= 32), NGAN_ERR_ARG,
                 "bf16_conv3x3_fwd: epilogue 3 needs aux_in (the colour weights, fp32), aux_out, rnorm whenever y is stored, and N <= 32");
    NGAN_REQUIRE(resample != NGAN_RESAMPLE_UP2 || (H % 2 == 0 && W % 2 == 0), NGAN_ERR_SHAPE, "bf16_conv3x3_fwd: bilinear x2 needs even H, W");
    // 32-bit byte offsets inside one image (buffer descriptors): the largest tensor of the call, per image, must stay below 2 GiB
    NGAN_REQUIRE((long)H * W * (out_mode || resample == NGAN_RESAMPLE_POOL2 ? 4 : 1) * (K > N ? K : N) * 2 < (1L << 31), NGAN_ERR_SHAPE,
                 "bf16_conv3x3_fwd: one image must stay below 2 GiB (H=%d W=%d)", H, W);
    ConvArgsB a{reinterpret_cast<const __bf16*>(x), reinterpret_cast<const __bf16*>(packed), bias, reinterpret_cast<__bf16*>(y),
                (epilogue == EPI_LRELU_PN || epilogue == EPI_TO_IMAGE) ? rnorm : nullptr, B, H, W, 0, 0, 0, 0, resample, epilogue, out_mode, slope, eps,
                epilogue == EPI_PN_BWD ? reinterpret_cast<const __bf16*>(aux_in) : nullptr,
                epilogue == EPI_TO_IMAGE ? reinterpret_cast<const float*>(aux_in) : nullptr, aux_rn, aux_out};
    hipStream_t s = (hipStream_t)stream;
    int pgt; bool narrow;
    pick_tile(B, H, W, N, pgt, narrow);
    switch (K) {
        case 16: return ngan::conv3x3_bf16_launch_k16(a, N, pgt, narrow, s);
        case 32: return ngan::conv3x3_bf16_launch_k32(a, N, pgt, narrow, s);
        case 64: return ngan::conv3x3_bf16_launch_k64(a, N, pgt, narrow, s);
        default: return ngan::conv3x3_bf16_launch_k128(a, N, pgt, narrow, s);
    }
}
