// Fused multi-tensor Adam over a flat parameter buffer.  Replaces optim.Adam.step
// (/root/reference/train.py:224-225, 366, 385: betas (beta1, 0.999), eps 1e-8, no weight decay, no amsgrad),
// i.e. the ATen lerp_/addcmul_/sqrt/addcdiv_ chain per parameter tensor.
// Per-tensor step counts are kept because a progressively grown net activates tensors at different times:
// torch skips parameters whose .grad is None, so their bias correction starts when they first receive one.
// Hyper-parameters and step counts live in device memory so a captured graph replays with fresh values.
#include "ngan_common.h"

namespace {

__global__ void adam_advance_kernel(const int* __restrict__ active, float* __restrict__ step, int n_seg) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_seg && active[i]) step[i] += 1.0f;
}

constexpr int CHUNK = 4096;

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, const long* __restrict__ seg_off,
                                                   const long* __restrict__ seg_len, const int* __restrict__ seg_active,
                                                   const float* __restrict__ seg_step, const int* __restrict__ chunk_seg,
                                                   const long* __restrict__ chunk_off, const float* __restrict__ hyper) {
    const int seg = chunk_seg[blockIdx.x];
    if (!seg_active[seg]) return;
    const AdamCoef k = adam_coef(hyper, seg_step[seg]);
    const long off = chunk_off[blockIdx.x];
    const long base = seg_off[seg] + off;
    const long n = min((long)CHUNK, seg_len[seg] - off);
    for (long i = threadIdx.x; i < n; i += 256) {
        const long j = base + i;
        float pv = p[j], mv = m[j], vv = v[j];
        adam_update(k, g[j], pv, mv, vv);
        m[j] = mv;
        v[j] = vv;
        p[j] = pv;
    }
}

}  // namespace

extern "C" int ngan_adam_step(float* p, const float* g, float* m, float* v, const long* seg_off, const long* seg_len,
                              const int* seg_active, float* seg_step, int n_seg, const int* chunk_seg, const long* chunk_off,
                              int n_chunks, const float* hyper, int n_hyper, void* stream) {
    NGAN_REQUIRE(p && g && m && v && seg_off && seg_len && seg_active && seg_step && chunk_seg && chunk_off && hyper,
                 NGAN_ERR_ARG, "adam_step: null pointer");
    NGAN_REQUIRE(n_hyper == NGAN_ADAM_HYPER_FLOATS, NGAN_ERR_ARG, "adam_step: hyper holds %d floats, this library reads %d (include/ngan.h)",
                 n_hyper, NGAN_ADAM_HYPER_FLOATS);
    NGAN_REQUIRE(n_seg > 0 && n_chunks > 0, NGAN_ERR_SHAPE, "adam_step: n_seg=%d n_chunks=%d", n_seg, n_chunks);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(adam_advance_kernel, dim3(ngan::ceil_div(n_seg, 256)), dim3(256), 0, s, seg_active, seg_step, n_seg);
    int st = ngan::launch_status("ngan_adam_step(advance)");
    if (st) return st;
    hipLaunchKernelGGL(adam_kernel, dim3(n_chunks), dim3(256), 0, s, p, g, m, v, seg_off, seg_len, seg_active, seg_step,
                       chunk_seg, chunk_off, hyper);
    return ngan::launch_status("ngan_adam_step");
}
