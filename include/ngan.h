/*
 * ngan.h -- C ABI of libngan_hip.so: hand-written gfx950 (MI355X) kernels for the PGGAN / WGAN-GP
 * training step of oliviertrottier/neuron-gan.
 *
 * The reference has no FFI of its own: its hot path dispatches torch ATen operators from Python
 * (SURVEY.md 2.1).  Each entry point below replaces one of those dispatch sites; the reference
 * call site it stands in for is cited as file:line into /root/reference.  The reference-side
 * binding a maintainer would add is a ctypes stub, shown in INTEGRATION.md.
 *
 * Conventions
 *   - every tensor is fp32, device memory, pixel-major / channels-last: (B, H, W, C) contiguous.
 *     Images with C == 1 have the same bytes as the reference's NCHW tensors.  (The last section, "bf16 activation
 *     storage", adds entry points whose activation tensors are bf16; everything before it is the fp32 contract.)
 *   - weights keep the reference's parameter layouts (OIHW for convs, (out, in) for Linear), so
 *     state_dict tensors are passed as they are.
 *   - the caller allocates every output and workspace; the library never allocates, frees or
 *     synchronises.  `stream` is a hipStream_t (PyTorch's current stream), passed as void*.
 *   - return value: 0 ok; < 0 invalid argument / unsupported shape (see ngan_last_error());
 *     > 0 a hipError_t from the launch.
 *   - resample codes:  0 none, 1 avg-pool 2x2 on load, 2 bilinear x2 (align_corners=False) on load.
 */
#ifndef NGAN_H
#define NGAN_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NGAN_OK 0
#define NGAN_ERR_ARG (-1)
#define NGAN_ERR_SHAPE (-2)

#define NGAN_RESAMPLE_NONE 0
#define NGAN_RESAMPLE_POOL2 1
#define NGAN_RESAMPLE_UP2 2
/* flags of ngan_conv3x3_fwd / ngan_conv3x3_fwd_ex (per call; the library keeps no mutable global state) */
#define NGAN_CONV_SKIP_BORDER 1   /* precision 3 only: do not launch the border-ring kernel, the caller follows up with ngan_conv3x3_up2_border */

const char* ngan_version(void);
const char* ngan_last_error(void);

/* ---- 3x3 convolution, pad 1, stride 1: Conv2d_normalized.forward, models.py:203-204 (ATen conv2d) ---------
 * The kernels are an implicit GEMM on v_mfma_f32_16x16x4_f32 (exact fp32).  Weights are first re-ordered
 * into MFMA fragment order and pre-multiplied by the equalised-LR constant `scale` (models.py:201).
 *   mode 0 (forward):  k = Cin, n = Cout.
 *   mode 1 (dgrad):    k = Cout, n = Cin, taps flipped: the same kernel then computes the input gradient.
 * `packed` holds ngan_conv3x3_packed_floats(...) floats.  Cin and Cout must be multiples of 16. */
int ngan_conv3x3_pack_weights(const float* w_oihw, float* packed, int Cout, int Cin, int mode, float scale, int precision,
                              void* stream);
/* precision 0: exact fp32 MFMA (v_mfma_f32_16x16x4_f32).  precision 1: "bf16x3" -- every fp32 operand is split into
 * hi = bf16(v), lo = bf16(v - hi) and a product is hi*hi + hi*lo + lo*hi on v_mfma_f32_16x16x32_bf16 with fp32
 * accumulation (relative error ~1e-5 per product instead of ~1e-7; ~5x the fp32 MFMA rate, which makes these layers
 * HBM-bound).  Ask before packing / calling -- the answer is the precision CODE to pack with and to pass to the conv call:
 *   0 exact fp32;  1 split-bf16;  2 split-bf16 with a K = 16 contraction zero-padded to 32 (its own packed layout);
 *   3 split-bf16 with the bilinear x2 of `resample` 2 folded into the weights (N = 16, K in {16, 32}, large images): `packed` then
 *     holds four 3x3 weight sets over the LOW-resolution input, one per output parity (py, px),
 *     W_eff[dr][dc] = sum_{ky,kx} W[ky][kx] * E[py][ky][dr] * E[px][kx][dc]  (E: the .25/.75 blend rows of upsample_bilinear2d,
 *     align_corners = False), followed by the scaled fp32 weights for the one-pixel border ring (mode 0 only);
 *   4 exact fp32 by Winograd F(2x2, 3x3) (answered for a REQUESTED precision 0 on large images: the 16 -> 16 layers with plain or
 *     bilinear input, and -- on widths that are multiples of 32 -- every shape with K, N in {16, 32}, plain or bilinear input):
 *     `packed` holds the 16 transformed weight sets G g G^T (16 * Cin * Cout floats), the kernel transforms 4x4 input patches
 *     (B^T d B; bilinear input: (B^T E) L (B^T E)^T straight from the 3x3 low-resolution patch) and 2x2 output tiles (A^T M A) per
 *     lane and spends 16 instead of 36 v_mfma_f32_16x16x4_f32 per 16 pixels and 16 x 16 channel pair.  fp32 arithmetic throughout;
 *     its error against an fp64 evaluation of the fused operator is below 1e-6 relative L2 (tests/test_gpu_ops.py::
 *     test_winograd_kernels_against_fp64; the direct form: 2.6e-7 against 1.5e-7 on the same operands).
 * The answer depends on the shape only: the library reads no environment variable (the measurement switches that select the
 * direct forms exist in the diagnostic build, `make diag`, only). */
int ngan_conv3x3_algorithm(int B, int H, int W, int K, int N, int resample, int precision);
long ngan_conv3x3_packed_floats(int Cout, int Cin, int precision);   /* size of `packed` in floats */

/* Re-pack many weights with ONE launch (after an optimiser step).  `table` is a device array of n_entries records
 *   { const float* src; float* dst; int Cout, Cin, mode, precision; float scale; int pad; long first; }          (48 bytes)
 * where `first` is the running sum of ngan_conv3x3_pack_elements(...) over the preceding entries (the unit is one packed
 * element: a float for precision 0 / 4, a bf16 for precision 1 / 2, precision 3: bf16 for the four sets, then one per raw fp32
 * weight) and total_elements is the sum over all entries. */
long ngan_conv3x3_pack_elements(int Cout, int Cin, int mode, int precision);
int ngan_conv3x3_pack_many(const void* table, int n_entries, long total_elements, void* stream);

/* y = epilogue(conv3x3(resample(x), packed) + bias)      (models.py:252-268 fused: resample, conv, LReLU, PixelNorm)
 *   x        resample 0: (B,H,W,K)   1: (B,2H,2W,K)   2: (B,H/2,W/2,K)       (H, W: conv/output resolution)
 *   bias     N floats or NULL
 *   epilogue 0: y = conv (+bias)            1: y = PixelNorm(LeakyReLU(conv + bias)), rnorm (B,H,W) = sqrt(mean_c a^2 + eps)
 *   out_mode 0: y is (B,H,W,N)              1: avg-pool adjoint store: y is (B,2H,2W,N), each value * 0.25 to its 2x2 block
 * K = contraction channels (any multiple of 16), N = output channels: 16, 32, 64 or 128 per call (the Python layer runs wider
 * layers as output-channel chunks of these sizes, ops._n_chunks). */
int ngan_conv3x3_fwd(const float* x, const float* packed, const float* bias, float* y, float* rnorm,
                     int B, int H, int W, int K, int N, int resample, int epilogue, int out_mode,
                     float slope, float eps, int precision, int flags, void* stream);

/* The same kernels with two more fused epilogues (what the hand-scheduled first-order passes of train.py:365, 384 use):
 *   epilogue 2: the call computes an input gradient g (packed = flipped weights) and applies the backward of the
 *               LeakyReLU -> PixelNorm that produced the layer's input: y = m*(g - aux_in*mean_c(g*aux_in))/aux_rn with
 *               m = aux_in > 0 ? 1 : slope.  aux_in (same shape as y, also with out_mode 1) is that input, aux_rn its norms.
 *               No resampling, no bias.  Always available: shapes without a fused kernel run the PixelNorm backward in place
 *               as a second launch (ngan_conv3x3_epilogue_fused tells which).
 *   epilogue 3: epilogue 1 followed by ToImage (models.py:141-146, one colour): aux_out (B,H,W) = tanh(sum_c aux_in[c]*y[c]);
 *               aux_in = the N colour weights.  y / rnorm may be NULL (inference: the activation is never written).
 *               Only where ngan_conv3x3_epilogue_fused(...) returns 1. */
/* Precision 3 (bilinear x2 folded into the weights): the one-pixel border ring of the output is written by a second, small kernel.
 * By default ngan_conv3x3_fwd / _fwd_ex launch it themselves.  With flags & NGAN_CONV_SKIP_BORDER a call launches the main kernel
 * only and the caller follows it with ngan_conv3x3_up2_border on the same stream (the Python layer does, so that a per-call timer
 * around ngan_conv3x3_fwd brackets exactly one kernel).  The choice is per call: nothing is remembered between calls. */
int ngan_conv3x3_up2_border(const float* x, const float* packed, const float* bias, float* y, float* rnorm,
                            int B, int H, int W, int K, int N, int epilogue, float slope, float eps, void* stream);
int ngan_conv3x3_epilogue_fused(int B, int H, int W, int K, int N, int resample, int epilogue, int out_mode, int precision);
/* Pooled side output of epilogue 1: where this returns 1 (the Winograd kernels, precision code 4, on whole 32-pixel tiles and even
 * H), ngan_conv3x3_fwd_ex with epilogue 1 and aux_out != NULL also writes aux_out (B, H/2, W/2, N) = the 2x2 average of y -- the
 * input of the next block's AvgPool2d(2) + conv (models.py:252-254) -- with the association of ngan_pool2_fwd, i.e. the same
 * bits a separate pooling pass over y would give.  Elsewhere aux_out must be NULL for epilogue 1. */
int ngan_conv3x3_pooled_output(int B, int H, int W, int K, int N, int resample, int precision);
int ngan_conv3x3_fwd_ex(const float* x, const float* packed, const float* bias, float* y, float* rnorm,
                        const float* aux_in, const float* aux_rn, float* aux_out,
                        int B, int H, int W, int K, int N, int resample, int epilogue, int out_mode,
                        float slope, float eps, int precision, int flags, void* stream);

/* name of the kernel template instance ngan_conv3x3_fwd dispatches to for these arguments, as rocprofv3 prints it
 * (profiling aid: lets bench.py label its HIP-event timings with the same names as the kernel trace) */
int ngan_conv3x3_kernel_name(int B, int H, int W, int K, int N, int resample, int epilogue, int out_mode, int precision,
                             char* buf, int len);
/* the same for ngan_conv3x3_wgrad (its main kernel; the slab reduction is a second kernel) */
int ngan_conv3x3_wgrad_kernel_name(int B, int H, int W, int Cin, int Cout, int resample, int precision, char* buf, int len);

/* weight gradient (ATen convolution_backward, weight part):
 *   gw[co][ci][ky][kx] = scale * sum_{b,y,x} g[b,y,x,co] * resample(x)[b,y+ky-1,x+kx-1,ci]      gw is OIHW
 * precision 1 requests the split-bf16 kernel (used when the image is at least 32 pixels wide, else exact fp32).
 * accumulate != 0: gw += ... (adds into an existing gradient buffer).  workspace: ngan_conv3x3_wgrad_workspace_bytes(...) bytes.
 * precision 0 on images wider than 16 pixels contracts in Winograd form, dW = G^T [ sum over 2x2 output tiles (A dY A^T) . (B^T d B) ] G
 * (fp32 arithmetic, 16 position accumulators instead of 9 taps, the back-transform applied to every partial sum before it is written:
 * same slabs, same fixed-order reduction). */
size_t ngan_conv3x3_wgrad_workspace_bytes(int B, int H, int W, int Cin, int Cout);
int ngan_conv3x3_wgrad(const float* x, const float* g, float* gw, float* workspace,
                       int B, int H, int W, int Cin, int Cout, int resample, float scale, int accumulate, int precision,
                       void* stream);

/* Deferred reduction.  ngan_conv3x3_wgrad(..., accumulate = 2, ...) writes the slabs only; later ONE call reduces the slabs of many
 * such calls (a whole backward pass) and merges up to 4 contributions to the same gradient tensor in a fixed order.
 * `entries`: host array of n records
 *   { const float* partial[4]; float* gw; int nparts[4]; int nsrc, nslices, n_ci_slices, co_s, ci_s, K, accumulate, reserved;
 *     float scale[4]; }                                                                                     (104 bytes)
 * with (nparts, nslices, n_ci_slices, co_s, ci_s) from ngan_conv3x3_wgrad_plan (out5).  accumulate != 0: gw += sum. */
int ngan_conv3x3_wgrad_plan(int B, int H, int W, int Cin, int Cout, int precision, int* out5);
int ngan_conv3x3_wgrad_reduce_many(const void* entries, int n, void* stream);

/* ---- LeakyReLU -> PixelNorm: models.py:263-264, 118-126 (ATen leaky_relu, pow, mean, sqrt, div) ---------------
 * fwd:    a = lrelu(c + bias); r = sqrt(mean_c(a^2) + eps); y = a / r
 * bwd:    gc = m * ((gy - y*mean_c(gy*y)) / r + gr*y/C),  m = (y > 0 ? 1 : slope); gr (npix) may be NULL
 * bwdbwd: given h = dL/d(gc) of a bwd call made with gr == NULL:
 *           ggy = (h' - y*t)/r,  gy_out = -(s*h' + t*gy)/r,  gr_out = -C*(u - s*t)/r^2
 *           with h' = m*h, s = mean_c(gy*y), t = mean_c(h'*y), u = mean_c(h'*gy)
 * C must be a multiple of 4 with C/4 a power of two <= 64. */
int ngan_lrelu_pixelnorm_fwd(const float* c, const float* bias, float* y, float* rnorm, long npix, int C,
                             float slope, float eps, void* stream);
int ngan_lrelu_pixelnorm_bwd(const float* gy, const float* gr, const float* y, const float* rnorm, float* gc,
                             long npix, int C, float slope, void* stream);
/* bwd with the incoming gradient given as a sum of two tensors (gy + gy2; gy2 may be NULL): in the gradient-penalty pass the
 * gradient w.r.t. a layer output has two contributions, and summing them here saves a separate elementwise pass */
int ngan_lrelu_pixelnorm_bwd2(const float* gy, const float* gy2, const float* gr, const float* y, const float* rnorm, float* gc,
                              long npix, int C, float slope, void* stream);
int ngan_lrelu_pixelnorm_bwdbwd(const float* h, const float* gy, const float* y, const float* rnorm,
                                float* ggy, float* gy_out, float* gr_out, long npix, int C, float slope, void* stream);

/* column sums over pixels: out[c] = scale * sum_p g[p][c]   (bias gradient of conv2d).  workspace: 1024*C floats */
int ngan_channel_sum(const float* g, float* out, float* workspace, long npix, int C, float scale, void* stream);
/* Small parameter gradients added straight into an existing gradient buffer (the step driver's flat .grad views) instead of being
 * returned and added by a separate elementwise launch: the `_acc` forms take `accumulate`, a bit mask over the function's outputs
 * (bit i set: output i is added into, clear: written).  channel_sum: out (bit 0). */
int ngan_channel_sum_acc(const float* g, float* out, float* workspace, long npix, int C, float scale, int accumulate, void* stream);

/* ---- FromImage 1x1 conv + bias: models.py:161-165 ---------------------------------------------------------------
 * fwd: y[p][c] = sum_k w[c][k]*x[p][k] + b[c];  pool=1: x is (B,2H,2W,Ncol) and is 2x2-averaged on load (models.py:519)
 * dx:  gx[p][k] = sum_c w[c][k]*g[p][c];        pool=1: gx is (B,2H,2W,Ncol), avg-pool adjoint store
 * dw:  gw[c][k] = sum_p g[p][c]*x[p][k], gb[c] = sum_p g[p][c] (gb may be NULL).  workspace: 1024*C*(Ncol+1) floats */
int ngan_from_image_fwd(const float* x, const float* w, const float* b, float* y, int B, int H, int W, int Ncol, int C,
                        int pool, void* stream);
int ngan_from_image_dx(const float* g, const float* w, float* gx, int B, int H, int W, int Ncol, int C, int pool, void* stream);
int ngan_from_image_dw(const float* x, const float* g, float* gw, float* gb, float* workspace,
                       int B, int H, int W, int Ncol, int C, int pool, void* stream);
int ngan_from_image_dw_acc(const float* x, const float* g, float* gw, float* gb, float* workspace,
                           int B, int H, int W, int Ncol, int C, int pool, int accumulate, void* stream);   /* accumulate: bit 0 gw, bit 1 gb */

/* ---- ToImage 1x1 conv + tanh: models.py:141-149 ------------------------------------------------------------------
 * fwd: t[p][k] = tanh(sum_c w[k][c]*x[p][c])
 * bwd: q = g*(1-t^2); gx[p][c] = sum_k q[p][k]*w[k][c]; gw[k][c] = sum_p q[p][k]*x[p][c].  workspace: 1024*C*Ncol floats */
int ngan_to_image_fwd(const float* x, const float* w, float* t, long npix, int C, int Ncol, void* stream);
int ngan_to_image_bwd(const float* g, const float* t, const float* x, const float* w, float* gx, float* gw,
                      float* workspace, long npix, int C, int Ncol, void* stream);
/* to_image_bwd where the ToImage input is the output y of a LeakyReLU -> PixelNorm (norms rnorm): gc receives the gradient w.r.t.
 * that operator's input, i.e. the PixelNorm/LeakyReLU backward is applied before the store (one pass over the activation) */
int ngan_to_image_bwd_pnbwd(const float* g, const float* t, const float* y, const float* rnorm, const float* w, float* gc,
                            float* gw, float* workspace, long npix, int C, int Ncol, float slope, void* stream);
int ngan_to_image_bwd_pnbwd_acc(const float* g, const float* t, const float* y, const float* rnorm, const float* w, float* gc,
                                float* gw, float* workspace, long npix, int C, int Ncol, float slope, int accumulate, void* stream);   /* accumulate != 0: gw += */

/* ---- resampling: models.py:87-89 (F.interpolate bilinear, align_corners=None) and models.py:254 (AvgPool2d(2)) --
 * (h, w) is always the LOW resolution; adjoint = transpose of the linear map. */
int ngan_up2_fwd(const float* x, float* y, int B, int h, int w, int C, void* stream);
int ngan_up2_adjoint(const float* gy, float* gx, int B, int h, int w, int C, void* stream);
/* up2_adjoint followed by the backward of the LeakyReLU -> PixelNorm that produced the low-resolution tensor `yprev` (B,h,w,C) with
 * norms `rnorm`: out = m*(g' - yprev*mean_c(g'*yprev))/rnorm, g' = up2_adjoint(g).  C/4 a power of two <= 64. */
int ngan_up2_adjoint_pnbwd(const float* g, const float* yprev, const float* rnorm, float* out, int B, int h, int w, int C,
                           float slope, void* stream);
int ngan_pool2_fwd(const float* x, float* y, int B, int h, int w, int C, void* stream);
int ngan_pool2_adjoint(const float* gy, float* gx, int B, int h, int w, int C, void* stream);

/* ---- fade-in and interpolation arithmetic: models.py:350, 521; loss_functions.py:171 -----------------------------
 * lerp:   out = a + alpha*(b - a), alpha read from device memory (so a captured graph follows the transition)
 * axpby:  out = ca*a + cb*b with host scalars (b may be NULL -> out = ca*a)
 * fade_bwd: ga = (1-alpha)*g, gb = alpha*g
 * xhat:   out[b,:] = eps[b]*real[b,:] + (1-eps[b])*fake[b,:] */
int ngan_lerp(const float* a, const float* b, const float* alpha, float* out, long n, void* stream);
int ngan_axpby(const float* a, const float* b, float ca, float cb, float* out, long n, void* stream);
int ngan_fade_bwd(const float* g, const float* alpha, float* ga, float* gb, long n, void* stream);
int ngan_xhat(const float* real, const float* fake, const float* eps, float* out, int B, long n, void* stream);

/* ---- gradient penalty pieces: loss_functions.py:176 (ATen linalg_vector_norm) ----------------------------------
 * norms[b] = ||g[b,:]||_2 (two fixed-order stages; rows 16-byte aligned, n a multiple of 4);   scale_rows: out[b,:] = coef[b] * g[b,:] */
int ngan_sample_l2norm(const float* g, float* norms, float* workspace /* 64*B floats */, int B, long n, void* stream);
int ngan_scale_rows(const float* g, const float* coef, float* out, int B, long n, void* stream);
/* the penalty's scalar head and its adjoint: out1 = lambda * mean((norms - 1)^2);  coef[b] = g_out * 2 lambda (norms[b] - 1) / (B norms[b])
 * (feed coef to ngan_scale_rows to get the gradient w.r.t. g) */
int ngan_gp_head(const float* norms, int B, float lambda, float* out1, void* stream);
int ngan_gp_coef(const float* norms, int B, float lambda, const float* g_out /* 1 float */, float* coef, void* stream);

/* ---- scalar heads of the Wasserstein losses: loss_functions.py:21-45, 67 (ATen mean / neg / add / square chains) -----------
 * scores holds n_real real scores followed by n_fake fake scores.  loss = -mean(real) + mean(fake) + drift * mean(real^2), plus the two means
 * (three separate 1-float outputs);  n_fake = 0 gives -mean(scores), the generator loss.  bwd: gradient w.r.t. scores from the three
 * output gradients (device scalars, NULL = 0). */
int ngan_wloss_head(const float* scores, int n_real, int n_fake, float drift, float* loss, float* mean_real, float* mean_fake,
                    void* stream);
int ngan_wloss_head_bwd(const float* scores, int n_real, int n_fake, float drift, const float* g_loss, const float* g_real,
                        const float* g_fake, float* g_scores, void* stream);

/* ---- latent projection: utils.py:77-78 (clamp(-c, c), L2-normalise each row), in place on (rows, dim) normal draws ---------- */
int ngan_latent_normalize(float* z, int rows, int dim, float clamp, void* stream);

/* ---- generator stem: Linear_normalized -> Unflatten -> LeakyReLU -> PixelNorm, models.py:299-311 (ATen mm) --------
 * fwd:   y[b][p][c] = PN(LReLU(scale * sum_k z[b][k] * Wt[c*S + p][k])),  y (B,S,C), rnorm (B,S); W is (C*S, K)
 * wgrad: gW[c*S+p][k] = scale * sum_b gc[b][p][c] * z[b][k]
 * dgrad: gz[b][k] = scale * sum_{p,c} gc[b][p][c] * W[c*S+p][k] */
int ngan_linear_lrelu_pn_fwd(const float* z, const float* Wt, float* y, float* rnorm, int B, int K, int S, int C,
                             float scale, float slope, float eps, void* stream);
int ngan_linear_wgrad(const float* z, const float* gc, float* gW, int B, int K, int S, int C, float scale, void* stream);
/* accumulate != 0: gW += ... (K <= 512, a multiple of 16): adds straight into the parameter's gradient buffer */
int ngan_linear_wgrad_acc(const float* z, const float* gc, float* gW, int B, int K, int S, int C, float scale, int accumulate,
                          void* stream);
/* the same contraction with Adam applied in its epilogue instead of a stored gradient (K <= 512, a multiple of 16; any B -- the
 * data-parallel ranks pass the gathered factors): p, m, v are the stem weight's slices of the flat parameter / moment buffers,
 * seg_step points at its (already advanced) step count, hyper as in ngan_adam_step.  Replaces, for this tensor, the store in
 * ngan_linear_wgrad plus its chunks of ngan_adam_step: same arithmetic, same bits. */
int ngan_linear_wgrad_adam(const float* z, const float* gc, float* p, float* m, float* v, const float* seg_step,
                           const float* hyper, int n_hyper, int B, int K, int S, int C, float scale, void* stream);
int ngan_linear_dgrad(const float* gc, const float* Wt, float* gz, int B, int K, int S, int C, float scale, void* stream);

/* ---- critic head: Conv2d_normalized(C, 1, (S,S), padding 0) + Flatten, models.py:485-490 ------------------------
 * fwd: out[b] = scale * sum_{p,c} y[b][p][c]*W[c*S2+p] + bias[0]
 * dx:  gy[b][p][c] = scale * go[b] * W[c*S2+p]
 * dw:  gW[c*S2+p] = scale * sum_b go[b]*y[b][p][c];  gb[0] = sum_b go[b] */
int ngan_final_dot_fwd(const float* y, const float* W, const float* bias, float* out, int B, int S2, int C, float scale, void* stream);
int ngan_final_dot_dx(const float* go, const float* W, float* gy, int B, int S2, int C, float scale, void* stream);
int ngan_final_dot_dw(const float* y, const float* go, float* gW, float* gb, int B, int S2, int C, float scale, void* stream);
int ngan_final_dot_dw_acc(const float* y, const float* go, float* gW, float* gb, int B, int S2, int C, float scale, int accumulate,
                          void* stream);                                   /* accumulate: bit 0 gW, bit 1 gb */

/* ---- Adam: optim.Adam.step, train.py:224-225, 366, 385 (betas (beta1, 0.999), eps 1e-8, no weight decay) ---------
 * One launch updates every active segment of a flat parameter buffer.
 *   seg_off[i], seg_len[i]  element offset / length of tensor i inside p, g, m, v   (device, int64)
 *   seg_active[i]           1 if tensor i received a gradient this step (inactive tensors keep step and state)
 *   seg_step[i]             per-tensor step count (device float, incremented here for active tensors)
 *   hyper                   9 device floats {lr, beta1, beta2, eps, grad_scale, 1 - beta1, 1 - beta2, ln beta1, ln beta2} (the last
 *                           four rounded from the host's double values: torch forms `1 - beta` and its bias corrections in
 *                           double; here 1 - beta^t = -expm1(t ln beta)); the gradient is multiplied by grad_scale
 *                           (1/world_size after a SUM all-reduce across data-parallel ranks, otherwise 1)
 *   n_hyper                 the number of floats the caller put into `hyper`: must equal NGAN_ADAM_HYPER_FLOATS.  (`hyper` grew from 5
 *                           to 9 floats in round 3; the count is checked so that a binding written against the older layout gets
 *                           NGAN_ERR_ARG instead of a kernel that reads past its buffer.  ngan_linear_wgrad_adam takes the same pair.)
 * chunk_seg / chunk_off (device int32 / int64): work list, one entry per 4096-element chunk. */
#define NGAN_ADAM_HYPER_FLOATS 9
int ngan_adam_step(float* p, const float* g, float* m, float* v, const long* seg_off, const long* seg_len,
                   const int* seg_active, float* seg_step, int n_seg, const int* chunk_seg, const long* chunk_off,
                   int n_chunks, const float* hyper, int n_hyper, void* stream);

/* ---- the critic's first layer pair as one operator (first-order passes): FromImage (ONE colour channel, models.py:161-165) folded
 * into the block's first 3x3 conv + LeakyReLU + PixelNorm (models.py:252-264).  f[c] = wf[c]*p + bf[c] is affine in one number per
 * pixel, so the conv over its C channels is a 3x3 conv over ONE channel with A[n][t] = scale*sum_c W[n][c][t]*wf[c] and a
 * border-aware bias sum_t Bv[n][t] (taps inside the image only); the C-channel tensor is never written.
 *   p (B,H,W) image (already pooled);  w_conv (N,C,3,3), b_conv (N) or NULL;  wf, bf (C);  y (B,H,W,N), rnorm (B,H,W);
 *   N in {16, 32}, C <= 64
 *   bwd: gw_conv (+)= dL/dW, gwf, gbf, gb_conv (or NULL) from gc = dL/d(pre-activation);  workspace: ngan_first_block_workspace_floats floats
 *   dx:  gx = dL/dp (pool = 0) or its avg-pool adjoint on the (B,2H,2W) image (pool = 1) */
size_t ngan_first_block_table_floats(int N);      /* = 2*9*N: the folded tables A, Bv (written by fwd, read by dx) */
int ngan_first_block_fwd(const float* p, const float* w_conv, const float* wf, const float* bf, const float* b_conv, float* y,
                         float* rnorm, float* tables, int B, int H, int W, int C, int N, float scale, float slope, float eps,
                         void* stream);
size_t ngan_first_block_workspace_floats(int B, int H, int N);
int ngan_first_block_bwd(const float* p, const float* gc, const float* w_conv, const float* wf, const float* bf,
                         float* gw_conv, float* gwf, float* gbf, float* gb_conv, float* workspace, int B, int H, int W, int C,
                         int N, float scale, int accumulate, void* stream);   /* accumulate: bit 0 gw_conv, 1 gwf, 2 gbf, 3 gb_conv */
int ngan_first_block_dx(const float* gc, const float* tables, float* gx, int B, int H, int W, int N, int pool, void* stream);

/* ---- on-device input pipeline: data/NeuronDataset.py:112-126, 149-164 (torchvision RandomAffine / RandomVerticalFlip /
 * ColorJitter / CenterCrop / Renormalize / Resize(antialias) per image) as two launches per batch, one colour channel.
 *   src     (N, P, P) padded images in [0, 1];  idx (B) int32: which image each sample uses
 *   params  B records { float cos, sin, tx, ty, brightness, contrast; int flip, contrast_first; }   (32 bytes)
 *           source pixel = nearest([cos, sin; -sin, cos] * (dst - (tx, ty))) in centred pixel coordinates, 0 outside
 *   out     (B, S, S) in [-1, 1]: centre crop R x R of the P x P canvas, renormalised, down-sampled to S x S (S divides R) with the
 *           antialiased bilinear (triangle) filter;  workspace: ngan_augment_workspace_bytes(B, P) bytes */
size_t ngan_augment_workspace_bytes(int B, int P);
int ngan_augment_batch(const float* src, const int* idx, const void* params, float* workspace, float* out,
                       int N, int B, int P, int R, int S, void* stream);

/* ==== bf16 activation storage ("bf16" mode, precision code 5): BASELINE.json's C2 configuration ==================================
 * The reference computes in the default dtype (/root/reference/train.py:136-144: fp32); this mode is an addition with its OWN,
 * stated tolerance (~1.5e-2 on |grad D|, 1e-1 on gradients against the fp32 path) -- never the headline.
 *
 * What changes: every ACTIVATION tensor -- the (B,H,W,C) outputs of conv / stem / FromImage layers, LeakyReLU -> PixelNorm outputs, and
 * the gradients w.r.t. them -- is stored as bf16 (ngan_bf16 = the raw 16 bits, round-to-nearest-even on store), and the 3x3
 * convolutions multiply bf16 operands with ONE v_mfma_f32_16x16x32_bf16 per product group (the fp32 master weights are rounded to
 * bf16 by the packing kernel, scale folded in first).  What does not: accumulation, PixelNorm statistics and norms (rnorm), biases,
 * LeakyReLU / tanh, the scalar loss heads, images (C = colours: x, x_hat, G(z), dD/dx), latents, every parameter, every parameter
 * gradient and the Adam state are fp32, and those entry points are the ones above.
 *
 * Each ngan_bf16_<op> below has the arguments and semantics of ngan_<op> above; the pointers typed ngan_bf16 are the activation
 * tensors.  Channel counts: the 3x3 conv takes K, N in {16, 32, 64, 128}; the per-pixel operators take C with C/4 a power of two
 * <= 64 (there is no wide.hip path behind them: other counts return NGAN_ERR_SHAPE).
 *
 * 3x3 convolution: ngan_conv3x3_algorithm(..., precision 5) answers 5 for the shapes the bf16 kernel takes (else 0: there is no
 * fallback), ngan_conv3x3_pack_weights / _pack_many / _packed_floats / _pack_elements take precision 5 (packed = bf16 MFMA
 * fragments), ngan_conv3x3_epilogue_fused(..., 5) answers for epilogues 2 and 3.  Resampling (avg-pool 2x2, bilinear x2) happens
 * while the input tile is staged: fp32 blend of the bf16 sources, one rounding.  aux_in: epilogue 2 -> the producer's output (bf16);
 * epilogue 3 -> the N colour weights (fp32).  There is no pooled side output in this mode (the consumer pools on load). */
typedef unsigned short ngan_bf16;
int ngan_bf16_conv3x3_fwd(const ngan_bf16* x, const float* packed, const float* bias, ngan_bf16* y, float* rnorm,
                          const void* aux_in, const float* aux_rn, float* aux_out,
                          int B, int H, int W, int K, int N, int resample, int epilogue, int out_mode,
                          float slope, float eps, void* stream);
/* weight gradient from bf16 x and g (fp32 slabs, fp32 gw; same workspace size, plan (precision 5) and deferred reduction as the
 * fp32 entry point: ngan_conv3x3_wgrad_workspace_bytes, ngan_conv3x3_wgrad_plan(..., 5, out5), ngan_conv3x3_wgrad_reduce_many) */
int ngan_bf16_conv3x3_wgrad(const ngan_bf16* x, const ngan_bf16* g, float* gw, float* workspace,
                            int B, int H, int W, int Cin, int Cout, int resample, float scale, int accumulate, void* stream);

int ngan_bf16_lrelu_pixelnorm_fwd(const ngan_bf16* c, const float* bias, ngan_bf16* y, float* rnorm, long npix, int C,
                                  float slope, float eps, void* stream);
int ngan_bf16_lrelu_pixelnorm_bwd(const ngan_bf16* gy, const float* gr, const ngan_bf16* y, const float* rnorm, ngan_bf16* gc,
                                  long npix, int C, float slope, void* stream);
int ngan_bf16_lrelu_pixelnorm_bwd2(const ngan_bf16* gy, const ngan_bf16* gy2, const float* gr, const ngan_bf16* y, const float* rnorm,
                                   ngan_bf16* gc, long npix, int C, float slope, void* stream);
int ngan_bf16_lrelu_pixelnorm_bwdbwd(const ngan_bf16* h, const ngan_bf16* gy, const ngan_bf16* y, const float* rnorm,
                                     ngan_bf16* ggy, ngan_bf16* gy_out, float* gr_out, long npix, int C, float slope, void* stream);
int ngan_bf16_channel_sum(const ngan_bf16* g, float* out, float* workspace, long npix, int C, float scale, void* stream);
int ngan_bf16_channel_sum_acc(const ngan_bf16* g, float* out, float* workspace, long npix, int C, float scale, int accumulate, void* stream);

int ngan_bf16_from_image_fwd(const float* x, const float* w, const float* b, ngan_bf16* y, int B, int H, int W, int Ncol, int C,
                             int pool, void* stream);
int ngan_bf16_from_image_dx(const ngan_bf16* g, const float* w, float* gx, int B, int H, int W, int Ncol, int C, int pool, void* stream);
int ngan_bf16_from_image_dw(const float* x, const ngan_bf16* g, float* gw, float* gb, float* workspace,
                            int B, int H, int W, int Ncol, int C, int pool, void* stream);
int ngan_bf16_from_image_dw_acc(const float* x, const ngan_bf16* g, float* gw, float* gb, float* workspace,
                                int B, int H, int W, int Ncol, int C, int pool, int accumulate, void* stream);
int ngan_bf16_to_image_fwd(const ngan_bf16* x, const float* w, float* t, long npix, int C, int Ncol, void* stream);
int ngan_bf16_to_image_bwd(const float* g, const float* t, const ngan_bf16* x, const float* w, ngan_bf16* gx, float* gw,
                           float* workspace, long npix, int C, int Ncol, void* stream);
int ngan_bf16_to_image_bwd_pnbwd(const float* g, const float* t, const ngan_bf16* y, const float* rnorm, const float* w, ngan_bf16* gc,
                                 float* gw, float* workspace, long npix, int C, int Ncol, float slope, void* stream);
int ngan_bf16_to_image_bwd_pnbwd_acc(const float* g, const float* t, const ngan_bf16* y, const float* rnorm, const float* w, ngan_bf16* gc,
                                     float* gw, float* workspace, long npix, int C, int Ncol, float slope, int accumulate, void* stream);

int ngan_bf16_up2_fwd(const ngan_bf16* x, ngan_bf16* y, int B, int h, int w, int C, void* stream);
int ngan_bf16_up2_adjoint(const ngan_bf16* gy, ngan_bf16* gx, int B, int h, int w, int C, void* stream);
int ngan_bf16_up2_adjoint_pnbwd(const ngan_bf16* g, const ngan_bf16* yprev, const float* rnorm, ngan_bf16* out, int B, int h, int w, int C,
                                float slope, void* stream);
int ngan_bf16_pool2_fwd(const ngan_bf16* x, ngan_bf16* y, int B, int h, int w, int C, void* stream);
int ngan_bf16_pool2_adjoint(const ngan_bf16* gy, ngan_bf16* gx, int B, int h, int w, int C, void* stream);
/* the critic's fade-in mixes two FEATURE tensors (models.py:521); the generator's mixes images (models.py:350: the fp32 ngan_lerp) */
int ngan_bf16_lerp(const ngan_bf16* a, const ngan_bf16* b, const float* alpha, ngan_bf16* out, long n, void* stream);
int ngan_bf16_fade_bwd(const ngan_bf16* g, const float* alpha, ngan_bf16* ga, ngan_bf16* gb, long n, void* stream);

/* stem and head: the contraction reads the fp32 master weight and fp32 latents (33 MFLOP per image: nothing to gain from a bf16
 * copy of a 67 MB weight that is read once); y / gc are bf16 */
int ngan_bf16_linear_lrelu_pn_fwd(const float* z, const float* Wt, ngan_bf16* y, float* rnorm, int B, int K, int S, int C,
                                  float scale, float slope, float eps, void* stream);
int ngan_bf16_linear_wgrad(const float* z, const ngan_bf16* gc, float* gW, int B, int K, int S, int C, float scale, void* stream);
int ngan_bf16_linear_wgrad_acc(const float* z, const ngan_bf16* gc, float* gW, int B, int K, int S, int C, float scale, int accumulate,
                               void* stream);
int ngan_bf16_linear_wgrad_adam(const float* z, const ngan_bf16* gc, float* p, float* m, float* v, const float* seg_step,
                                const float* hyper, int n_hyper, int B, int K, int S, int C, float scale, void* stream);
int ngan_bf16_linear_dgrad(const ngan_bf16* gc, const float* Wt, float* gz, int B, int K, int S, int C, float scale, void* stream);
int ngan_bf16_final_dot_fwd(const ngan_bf16* y, const float* W, const float* bias, float* out, int B, int S2, int C, float scale, void* stream);
int ngan_bf16_final_dot_dx(const float* go, const float* W, ngan_bf16* gy, int B, int S2, int C, float scale, void* stream);
int ngan_bf16_final_dot_dw(const ngan_bf16* y, const float* go, float* gW, float* gb, int B, int S2, int C, float scale, void* stream);
int ngan_bf16_final_dot_dw_acc(const ngan_bf16* y, const float* go, float* gW, float* gb, int B, int S2, int C, float scale, int accumulate,
                               void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NGAN_H */
