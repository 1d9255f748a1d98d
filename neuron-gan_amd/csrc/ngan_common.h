// Shared helpers for the gfx950 kernels behind include/ngan.h.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdarg>
#include "../../include/ngan.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace ngan {

void set_error(const char* fmt, ...);

// status of the launch just issued on this thread: 0 or the hipError_t
inline int launch_status(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return (int)e;
    }
    return NGAN_OK;
}

#define NGAN_REQUIRE(cond, code, ...)          \
    do {                                       \
        if (!(cond)) {                         \
            ::ngan::set_error(__VA_ARGS__);    \
            return (code);                     \
        }                                      \
    } while (0)

inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

// stage-2 of every two-stage reduction: out[i] = scale * sum_j partials[j*M + i]
int reduce_partials(const float* partials, int nparts, int M, float* out, float scale, hipStream_t s);
int reduce_partials_acc(const float* partials, int nparts, int M, long stride, float* out, int M1, float* out2, float scale, int accumulate,
                        hipStream_t s);          // accumulate: bit 0 out += , bit 1 out2 +=
int reduce_partials_split(const float* partials, int nparts, int M, long stride, float* out, int M1, float* out2, float scale,
                          hipStream_t s);

// wide.hip: the same operators for channel counts outside the lane-group kernels' range (C % 4 == 0, any size)
int wide_pn_fwd(const float* c, const float* bias, float* y, float* rn, long npix, int C, float slope, float eps, hipStream_t s);
int wide_pn_bwd(const float* gy, const float* gy2, const float* gr, const float* y, const float* rn, float* gc, long npix, int C, float slope, hipStream_t s);
int wide_pn_bwdbwd(const float* h, const float* gy, const float* y, const float* rn, float* ggy, float* gy_out, float* gr_out, long npix, int C,
                   float slope, hipStream_t s);
int wide_channel_sum(const float* g, float* out, long npix, int C, float scale, hipStream_t s);
int wide_to_image_fwd(const float* x, const float* w, float* t, long npix, int C, int Ncol, hipStream_t s);
int wide_to_image_bwd(const float* g, const float* t, const float* x, const float* w, float* gx, float* gw, long npix, int C, int Ncol,
                      const float* rn, float slope, hipStream_t s);
int wide_from_image_dx(const float* g, const float* w, float* gx, int B, int H, int W, int Ncol, int C, int pool, hipStream_t s);
int wide_from_image_dw(const float* x, const float* g, float* gw, float* gb, int B, int H, int W, int Ncol, int C, int pool, hipStream_t s);

}  // namespace ngan

// ---- device helpers ---------------------------------------------------------------------------------------
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
// One element of optim.Adam.step (train.py:224-225: betas (beta1, 0.999), eps 1e-8, no weight decay, no amsgrad) -- shared by the flat
// multi-tensor kernel (adam.hip) and the generator stem's gradient-free update (linear.hip), so both produce the same bits.
// hyper = {lr, beta1, beta2, eps, grad_scale, 1 - beta1, 1 - beta2, ln beta1, ln beta2}: the last four are rounded from the host's
// double values, as torch rounds the Python-side `1 - beta` it passes to lerp_ / addcmul_ and forms its bias corrections in double
// (1.0f - 0.999f is 1.3e-5 off 0.001f: every exp_avg_sq, and the first steps' 1 - beta2^t, would carry that factor).  The step count
// lives on the device (graph replay), so the bias corrections are formed here: 1 - beta^t = -expm1(t ln beta), good to ~1e-7.
struct AdamCoef { float b2, omb1, omb2, eps, gscale, step_size, inv_sqrt_bc2; };
__device__ __forceinline__ AdamCoef adam_coef(const float* __restrict__ hyper, float t) {
    const float lr = hyper[0], b2 = hyper[2];
    const float bc1 = -expm1f(t * hyper[7]), bc2 = -expm1f(t * hyper[8]);
    return AdamCoef{b2, hyper[5], hyper[6], hyper[3], hyper[4], lr / bc1, 1.0f / sqrtf(bc2)};
}
__device__ __forceinline__ void adam_update(const AdamCoef& k, float g, float& p, float& m, float& v) {
    const float gv = g * k.gscale;                                 // 1/world_size after a SUM exchange, else 1
    const float mv = fmaf(k.omb1, gv - m, m);                      // m.lerp_(g, 1 - beta1)
    const float vv = fmaf(k.b2, v, k.omb2 * gv * gv);             // v.mul_(beta2).addcmul_(g, g, 1 - beta2)
    m = mv;
    v = vv;
    p -= k.step_size * (mv / (sqrtf(vv) * k.inv_sqrt_bc2 + k.eps));   // p.addcdiv_(m, sqrt(v)/sqrt(bc2) + eps, -lr/bc1)
}

// ---- activation storage type (round 4: "bf16" mode, precision code 5 of include/ngan.h) -------------------------------------------
// Every kernel that reads or writes an ACTIVATION tensor (a conv / stem / FromImage output, or the gradient w.r.t. one) is a template
// over its storage type T: float (the reference's arithmetic, the product default) or __bf16 (2-byte storage, round-to-nearest-even
// on store).  Arithmetic between a load and a store is fp32 in both; byte offsets follow from sizeof(T) through the pointer type.
// Images, norms, scalars, parameters, parameter gradients and Adam state are fp32 in both modes.
//   lda4(p) : 4 consecutive channels at p -> float4          sta4(p, v) : float4 -> 4 consecutive channels at p
//   lda1 / sta1 : one element
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float bf16_lo(unsigned u) { return __uint_as_float(u << 16); }            // element 0 of a packed pair
__device__ __forceinline__ float bf16_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }    // element 1
__device__ __forceinline__ unsigned pack_bf16(float a, float b) {                                     // {a, b} -> packed pair, RNE
    const bf16x2_t r = __builtin_convertvector((f32x2_t){a, b}, bf16x2_t);
    return __builtin_bit_cast(unsigned, r);
}
template <typename T> __device__ __forceinline__ float4 lda4(const T* p);
template <> __device__ __forceinline__ float4 lda4<float>(const float* p) { return *reinterpret_cast<const float4*>(p); }
template <> __device__ __forceinline__ float4 lda4<__bf16>(const __bf16* p) {
    const uint2 u = *reinterpret_cast<const uint2*>(p);
    return make_float4(bf16_lo(u.x), bf16_hi(u.x), bf16_lo(u.y), bf16_hi(u.y));
}
template <typename T> __device__ __forceinline__ void sta4(T* p, float4 v);
template <> __device__ __forceinline__ void sta4<float>(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
template <> __device__ __forceinline__ void sta4<__bf16>(__bf16* p, float4 v) {
    *reinterpret_cast<uint2*>(p) = make_uint2(pack_bf16(v.x, v.y), pack_bf16(v.z, v.w));
}
// V float4s = 4 V consecutive channels per lane.  V = 2 exists for bf16 storage: one 16-byte access per lane instead of 8 bytes -- the
// per-pixel kernels are pure streams, and with 2-byte elements a 4-channel access puts half as many bytes in flight per wave
template <typename T, int V> __device__ __forceinline__ void ldav(const T* p, float4 (&out)[V]) {
    if constexpr (V == 2 && sizeof(T) == 2) {
        const uint4 u = *reinterpret_cast<const uint4*>(p);
        out[0] = make_float4(bf16_lo(u.x), bf16_hi(u.x), bf16_lo(u.y), bf16_hi(u.y));
        out[1] = make_float4(bf16_lo(u.z), bf16_hi(u.z), bf16_lo(u.w), bf16_hi(u.w));
    } else {
#pragma unroll
        for (int i = 0; i < V; ++i) out[i] = lda4(p + 4 * i);
    }
}
template <typename T, int V> __device__ __forceinline__ void stav(T* p, const float4 (&v)[V]) {
    if constexpr (V == 2 && sizeof(T) == 2) {
        *reinterpret_cast<uint4*>(p) = make_uint4(pack_bf16(v[0].x, v[0].y), pack_bf16(v[0].z, v[0].w), pack_bf16(v[1].x, v[1].y), pack_bf16(v[1].z, v[1].w));
    } else {
#pragma unroll
        for (int i = 0; i < V; ++i) sta4(p + 4 * i, v[i]);
    }
}
template <typename T> __device__ __forceinline__ float lda1(const T* p) { return (float)*p; }
template <typename T> __device__ __forceinline__ void sta1(T* p, float v) { *p = (T)v; }

__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 f4scale(float4 a, float s) { return make_float4(a.x * s, a.y * s, a.z * s, a.w * s); }
__device__ __forceinline__ float4 f4fma(float4 a, float s, float4 c) {
    return make_float4(fmaf(a.x, s, c.x), fmaf(a.y, s, c.y), fmaf(a.z, s, c.z), fmaf(a.w, s, c.w));
}
__device__ __forceinline__ float f4dot(float4 a, float4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }

// butterfly sum over groups of `width` consecutive lanes (width a power of two <= 64)
template <int WIDTH>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int o = WIDTH / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// max(a, b) as ONE v_max_f32.  fmaxf() first canonicalises each operand (a `v_max_f32 x, x, x` apiece: 4 extra VALU instructions
// per LeakyReLU of a float4); the values fed here are MFMA accumulators and products, canonical already.  NaN in -> NaN out as before.
__device__ __forceinline__ float vmax1(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// Sum over the four 16-lane rows of a wave (lanes l, l ^ 16, l ^ 32, l ^ 48); every lane ends with the total.  The same two
// additions as `v += __shfl_xor(v, 16); v += __shfl_xor(v, 32)` -- bit-identical -- but on gfx950's row / half swaps
// (v_permlane16_swap: odd rows of the first register <-> even rows of the second; v_permlane32_swap: upper half <-> lower half),
// which are VALU instructions: the shuffles compile to ds_bpermute_b32, an LDS round trip with a full `s_waitcnt lgkmcnt(0)` each,
// i.e. two serial LDS latencies per pixel group in every PixelNorm epilogue.  (Inline asm: the builtin's second result came back
// aliased to the first in this compiler.)  s_nop 1 in front: the swap reads its operands as a VALU-written pair.  s_nop 1 behind:
// the hazard recogniser does not look inside an asm statement, so the wait states between the swap and whatever consumes its
// result are spelled out here instead of being left to the schedule (the compiler happened to put one s_nop 0 there; after an
// inlining change it might not -- round-2 advisor note; the same blindness lost MFMA wait states once).
__device__ __forceinline__ float sum_rows4(float v) {
    float w = v;
    asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(v), "+v"(w));
    v += w;
    w = v;
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(v), "+v"(w));
    return v + w;
}

// Bilinear x2 (align_corners=False) source taps for output index d (0 <= d < 2n):
// out[d] = w0*in[i0] + w1*in[i1]     (models.py:87-89 -> ATen upsample_bilinear2d)
__device__ __forceinline__ void up2_taps(int d, int n, int& i0, int& i1, float& w0, float& w1) {
    int i = d >> 1;
    if (d & 1) { i0 = i; i1 = min(i + 1, n - 1); w0 = 0.75f; w1 = 0.25f; }
    else       { i0 = max(i - 1, 0); i1 = i; w0 = 0.25f; w1 = 0.75f; }
}
