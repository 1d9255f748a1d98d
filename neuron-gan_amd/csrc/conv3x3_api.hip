// 3x3 convolution (pad 1, stride 1) on channels-last fp32 tensors: the C ABI's forward / input-gradient entry points
// (include/ngan.h) and the choice of kernel family per shape.  Replaces ATen conv2d / convolution_backward at
// /root/reference/models.py:203-204 and the resample / LeakyReLU / PixelNorm modules fused around it (models.py:252-268).
#include "conv3x3_internal.h"

// Exact-fp32 layers on large images in Winograd F(2x2, 3x3) form: 16 -> 16 with plain or bilinear input (conv3x3_tile_kernel /
// conv3x3_persist_kernel), and the shapes with a 32-channel side with plain input on whole 32-pixel tiles (conv3x3_wino.hip)
static bool wino_eligible(int B, int H, int W, int K, int N, int resample) {
    if (!NGAN_DIAG_FLAG("NGAN_WINOGRAD", true) || resample == NGAN_RESAMPLE_POOL2 || !persist_eligible(B, H, W, K, N, resample)) return false;
    if (K == 16 && N == 16) return true;
    return NGAN_DIAG_FLAG("NGAN_WINOGRAD32", true) && W % 32 == 0;
}

extern "C" int ngan_conv3x3_algorithm(int B, int H, int W, int K, int N, int resample, int precision) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    if (precision == 0) return wino_eligible(B, H, W, K, N, resample) ? 4 : 0;
    // bf16 activation storage: one kernel family for every shape it takes (the answer 0 means "no bf16 kernel": the caller must not
    // fall back to an fp32-storage call with bf16 pointers)
    if (precision == 5) return ngan::conv3x3_bf16_elements(K, N) > 0 ? 5 : 0;
    if (precision != 1) return 0;
    if (up2f_eligible(B, H, W, K, N, resample)) return 3;
    if (persist_eligible(B, H, W, K, N, resample)) return 1;
    if (ngan::conv3x3_mid_eligible(B, H, W, K, N)) return 1;
    // K = 16 into 32..128 channels where the persistent kernel does not apply (pooled input, small images): the mid kernel with the
    // contraction padded to 32 channels (zero weights) -- its own packed layout, hence its own precision code
    return (K == 16 && ngan::conv3x3_mid_eligible(B, H, W, 32, N)) ? 2 : 0;
}

// exact-fp32 layers with 32..128 channels on small images run in the fp32 variant of the mid kernel
static bool mid_f32_enabled() { return NGAN_DIAG_FLAG("NGAN_MID_F32", true); }

// epilogue 1 can also write y averaged over 2x2 blocks (aux_out of ngan_conv3x3_fwd_ex): the Winograd kernels on whole tiles, whose
// lanes own exactly one pooling window each
extern "C" int ngan_conv3x3_pooled_output(int B, int H, int W, int K, int N, int resample, int precision) {
    if (precision != 4 || B <= 0 || H <= 0 || W <= 0 || (H & 1) || W % 32) return 0;
    if (resample == NGAN_RESAMPLE_UP2 && !NGAN_DIAG_FLAG("NGAN_WINOGRAD_UP2", true) && K == 16 && N == 16) return 0;   // (conv3x3_persist_kernel)
    return ngan_conv3x3_algorithm(B, H, W, K, N, resample, 0) == 4 ? 1 : 0;
}

extern "C" int ngan_conv3x3_epilogue_fused(int B, int H, int W, int K, int N, int resample, int epilogue, int out_mode, int precision) {
    if (epilogue == EPI_NONE || epilogue == EPI_LRELU_PN) return 1;
    if (precision == 5)      // the bf16 kernel has every epilogue built in (ToImage: plain input, plain store)
        return ngan::conv3x3_bf16_elements(K, N) > 0 && (epilogue == EPI_PN_BWD ? resample == 0 : (resample == 0 && out_mode == 0 && N <= 32)) ? 1 : 0;
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
    NGAN_REQUIRE(precision != 5, NGAN_ERR_ARG, "conv3x3_fwd: precision 5 (bf16 activation storage) has its own entry point, ngan_bf16_conv3x3_fwd");
    NGAN_REQUIRE(precision == 0 || (precision == 4 && ngan_conv3x3_algorithm(B, H, W, K, N, resample, 0) == 4) ||
                 (precision != 4 && precision == ngan_conv3x3_algorithm(B, H, W, K, N, resample, 1)), NGAN_ERR_ARG,
                 "conv3x3_fwd: precision %d is not available for this shape (ask ngan_conv3x3_algorithm)", precision);
    NGAN_REQUIRE(B > 0 && H > 0 && W > 0, NGAN_ERR_SHAPE, "conv3x3_fwd: bad dims B=%d H=%d W=%d", B, H, W);
    NGAN_REQUIRE(K > 0 && K % 16 == 0, NGAN_ERR_SHAPE, "conv3x3_fwd: K=%d must be a positive multiple of 16", K);
    NGAN_REQUIRE(N == 16 || N == 32 || N == 64 || N == 128, NGAN_ERR_SHAPE, "conv3x3_fwd: N=%d must be 16/32/64/128", N);
    NGAN_REQUIRE(resample >= 0 && resample <= 2, NGAN_ERR_ARG, "conv3x3_fwd: resample %d", resample);
    NGAN_REQUIRE(epilogue >= EPI_NONE && epilogue <= EPI_TO_IMAGE, NGAN_ERR_ARG, "conv3x3_fwd: epilogue %d", epilogue);
    NGAN_REQUIRE(out_mode == 0 || (out_mode == 1 && (epilogue == EPI_NONE || epilogue == EPI_PN_BWD) && resample == 0), NGAN_ERR_ARG,
                 "conv3x3_fwd: out_mode %d needs epilogue 0 or 2 and resample 0", out_mode);
    NGAN_REQUIRE(epilogue != EPI_LRELU_PN || rnorm, NGAN_ERR_ARG, "conv3x3_fwd: epilogue 1 needs rnorm");
    NGAN_REQUIRE(epilogue != EPI_LRELU_PN || !aux_out || ngan_conv3x3_pooled_output(B, H, W, K, N, resample, precision), NGAN_ERR_ARG,
                 "conv3x3_fwd: epilogue 1 writes a pooled side output (aux_out) only where ngan_conv3x3_pooled_output(...) returns 1");
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
            const int st = ngan::conv3x3_up2f_launch(a, epilogue, s);
            if (st || (flags & NGAN_CONV_SKIP_BORDER)) return st;      // the caller launches ngan_conv3x3_up2_border itself
            return ngan::conv3x3_up2_border_launch(a, epilogue, s);
        }
        // precision code 4 = the Winograd form (16 -> 16: checked by the precision test above) = the kernels' PREC parameter 2
        const int tprec = precision == 4 ? 2 : precision;
        // Winograd form in conv3x3_wino.hip: the shapes with a 32-channel side, and every bilinear-input shape on whole tiles
        if (tprec == 2 && W % 32 == 0 && (K == 32 || N == 32 || (resample == NGAN_RESAMPLE_UP2 && NGAN_DIAG_FLAG("NGAN_WINOGRAD_UP2", true))))
            return ngan::conv3x3_wino_launch(a, N / 16, K / 16, resample, epilogue, out_mode, s);
        if (resample == 0 && NGAN_DIAG_FLAG("NGAN_TILE_KERNEL", true) && W % 32 == 0)      // plain input, whole tiles along x
            return ngan::conv3x3_tile_launch(a, N / 16, K / 16, epilogue, out_mode, tprec, s);
        return ngan::conv3x3_persist_launch(a, N / 16, K / 16, resample, epilogue, out_mode, tprec, s);
    }
    // many channels, small image: the kernel of conv3x3_mid.hip, split-bf16 (precision 2: K = 16 padded to 32) or exact fp32
    // (fp32 with a pooled input stays on the generic kernel, which measured 10 % faster there)
    if (precision >= 1 || (mid_f32_enabled() && resample != NGAN_RESAMPLE_POOL2 && ngan::conv3x3_mid_eligible(B, H, W, K, N)))
        return ngan::conv3x3_mid_launch(x, packed, bias, y, rnorm, aux_in, aux_rn, B, H, W, K, N, resample, epilogue, out_mode, slope, eps, precision, s);
    // generic exact-fp32 kernel: it has epilogues 0 and 1; the PixelNorm backward runs as a second launch, in place
    const int epi = epilogue == EPI_PN_BWD ? EPI_NONE : epilogue;
    const int st = ngan::conv3x3_generic_launch(a, resample, epi, out_mode, s);
    if (st || epilogue != EPI_PN_BWD) return st;
    return ngan_lrelu_pixelnorm_bwd(y, nullptr, aux_in, aux_rn, y, (long)B * H * W * (out_mode ? 4 : 1), N, slope, stream);
}

extern "C" int ngan_conv3x3_up2_border(const float* x, const float* packed, const float* bias, float* y, float* rnorm,
                                       int B, int H, int W, int K, int N, int epilogue, float slope, float eps, void* stream) {
    NGAN_REQUIRE(x && packed && y, NGAN_ERR_ARG, "conv3x3_up2_border: null pointer");
    NGAN_REQUIRE(ngan_conv3x3_algorithm(B, H, W, K, N, NGAN_RESAMPLE_UP2, 1) == 3, NGAN_ERR_SHAPE,
                 "conv3x3_up2_border: B=%d H=%d W=%d K=%d N=%d is not a folded-bilinear (precision 3) shape", B, H, W, K, N);
    NGAN_REQUIRE(epilogue == EPI_NONE || (epilogue == EPI_LRELU_PN && rnorm), NGAN_ERR_ARG, "conv3x3_up2_border: epilogue %d", epilogue);
    ConvArgs a{x, packed, bias, y, rnorm, B, H, W, K, N, 0, 0, slope, eps, nullptr, nullptr, nullptr};
    return ngan::conv3x3_up2_border_launch(a, epilogue, (hipStream_t)stream);
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
    if (precision == 5) return ngan::conv3x3_bf16_kernel_name(B, H, W, K, N, buf, len);
    if (precision == 3)
        snprintf(buf, len, "conv3x3_up2f_kernel<%d, %d>", K / 16, epilogue);
    else if (precision == 4 && W % 32 == 0 && (K == 32 || N == 32 || (resample == NGAN_RESAMPLE_UP2 && NGAN_DIAG_FLAG("NGAN_WINOGRAD_UP2", true))))
        snprintf(buf, len, "conv3x3_wino_kernel<%d, %d, %d, %d, %d, %d, %d>", K / 16, N / 16, ngan::conv3x3_wino_tile_rows(N / 16, K / 16),
                 (K == 16) ? 4 : 8, resample, (out_mode && epilogue != EPI_PN_BWD) ? 0 : epilogue, out_mode);
    else if (N <= 32 && K <= 32 && resample != NGAN_RESAMPLE_POOL2 && ci == 0) {
        if ((out_mode || resample == 0) && NGAN_DIAG_FLAG("NGAN_TILE_KERNEL", true) && W % 32 == 0)
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

