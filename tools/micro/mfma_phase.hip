// Which non-MFMA phases of a tile loop does the fp32 matrix pipe of a SIMD fail to hide behind the other resident waves?
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_phase.hip -o tools/micro/mfma_phase && tools/micro/mfma_phase
// Workgroups of 256 threads (4 waves, one per SIMD), WG workgroups per CU resident; every wave runs `tiles` x { phase; 144 MFMAs }.
// phase kinds: 0 none | 1 barrier only | 2 six ds_write_b128 + barrier + five ds_read_b128 (the conv kernel's staging) |
// 3 six global_load_dwordx4 issued before the MFMAs and awaited after them (prefetch, as the kernels do) | 4 like 3 but awaited
// right away (exposed latency) | 5 four global_store_dwordx4 | 6 = 2 + 3 + 5 + 40 VALU (the whole tile loop's skeleton)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int KIND>
__global__ __launch_bounds__(256) void k(const float4* __restrict__ src, float4* __restrict__ dst, float* out, int tiles, long stride) {
    __shared__ float4 lds[6 * 256 + 64];
    f32x4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int tid = threadIdx.x;
    float a = tid * 0.001f + 1.0f, b = 0.5f;
    const float4* sp = src + (long)blockIdx.x * stride + tid;
    float4* dp = dst + (long)blockIdx.x * stride + tid;
    float4 st[6];
    for (int i = 0; i < 6; ++i) st[i] = make_float4(a, b, a, b);
    if (KIND == 3 || KIND == 6) for (int i = 0; i < 6; ++i) st[i] = sp[i * 256];
    for (int t = 0; t < tiles; ++t) {
        if (KIND == 1) __syncthreads();
        if (KIND == 2 || KIND == 6) {
            __syncthreads();
            for (int i = 0; i < 6; ++i) lds[i * 256 + tid] = st[i];
            __syncthreads();
            float4 r = lds[(tid * 5) & 1023];
            for (int i = 1; i < 5; ++i) { float4 r2 = lds[(tid * 5 + i * 257) & 1023]; r.x += r2.x; r.y += r2.y; }
            a += r.x * 1e-20f; b += r.y * 1e-20f;
        }
        if (KIND == 3 || KIND == 6) { sp += 6 * 256; for (int i = 0; i < 6; ++i) st[i] = sp[i * 256]; }     // next tile's loads: in flight during the MFMAs
        if (KIND == 4) { sp += 6 * 256; for (int i = 0; i < 6; ++i) st[i] = sp[i * 256]; a += st[0].x * 1e-20f + st[5].y * 1e-20f; }
#pragma unroll
        for (int i = 0; i < 36; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j], 0, 0, 0);
        }
        if (KIND == 6) {
            float v = acc[0][0];
#pragma unroll
            for (int i = 0; i < 40; ++i) v = __builtin_fmaf(v, 0.999f, acc[i & 3][i & 3]);
            acc[0][0] = v;
        }
        if (KIND == 5 || KIND == 6) {
            for (int i = 0; i < 4; ++i) dp[i * 256] = make_float4(acc[i][0], acc[i][1], acc[i][2], acc[i][3]);
            dp += 4 * 256;
        }
        if (KIND == 3) a += st[0].x * 1e-20f;
    }
    float s = a + b + st[3].x;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 123.456f) out[tid] = s;
}

template <int KIND>
double run(const float4* src, float4* dst, float* out, int wg_per_cu, int tiles, long stride) {
    const int grid = 256 * wg_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(256), 0, 0, src, dst, out, tiles, stride);
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(256), 0, 0, src, dst, out, tiles, stride);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 200.0;   // us per launch
}

int main() {
    const int tiles = 32;
    const long stride = (long)(tiles + 2) * 6 * 256;      // float4 elements per workgroup
    float4 *src, *dst; float* out;
    hipMalloc(&src, stride * 1024 * 16); hipMalloc(&dst, stride * 1024 * 16); hipMalloc(&out, 4096);
    hipMemset(src, 0, stride * 1024 * 16);
    const char* names[] = {"none", "barrier", "lds stage", "prefetch ld", "exposed ld", "stores", "all"};
    printf("%-12s %-6s %10s %10s\n", "phase", "WG/CU", "us", "TFLOP/s");
    for (int kind = 0; kind < 7; ++kind)
        for (int w : {1, 2, 4}) {
            double us = kind == 0 ? run<0>(src, dst, out, w, tiles, stride) : kind == 1 ? run<1>(src, dst, out, w, tiles, stride)
                      : kind == 2 ? run<2>(src, dst, out, w, tiles, stride) : kind == 3 ? run<3>(src, dst, out, w, tiles, stride)
                      : kind == 4 ? run<4>(src, dst, out, w, tiles, stride) : kind == 5 ? run<5>(src, dst, out, w, tiles, stride)
                      : run<6>(src, dst, out, w, tiles, stride);
            double flops = 2048.0 * 144 * tiles * 256 * 4 * w;
            printf("%-12s %-6d %10.1f %10.1f\n", names[kind], w, us, flops / us / 1e6);
        }
    return 0;
}
