// Does the fp32 matrix pipe of a SIMD stay busy when its waves alternate MFMA bursts with non-MFMA gaps?
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_gap.hip -o tools/micro/mfma_gap && tools/micro/mfma_gap
// Each wave: `tiles` x { 144 x v_mfma_f32_16x16x4_f32 (4 accumulator chains); gap }.  gap kinds: 0 none, 1 s_sleep (the wave parks, no
// instruction issue), 2 a dependent VALU chain (v_fma: 4-cycle issue each), 3 a dependent chain of LDS round trips (ds_bpermute).
// Workgroups of 64 threads (one wave), W waves per SIMD resident = 4 W waves per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int KIND>
__global__ __launch_bounds__(64) void k(float* out, int tiles, int gap) {
    f32x4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 0.001f + 1.0f, b = 0.5f + blockIdx.x * 1e-6f, chain = a;
    for (int t = 0; t < tiles; ++t) {
#pragma unroll
        for (int i = 0; i < 36; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j], 0, 0, 0);
        }
        if (KIND == 1) { for (int g = 0; g < gap; g += 64) __builtin_amdgcn_s_sleep(1); }
        if (KIND == 2) { for (int g = 0; g < gap; g += 4) chain = __builtin_fmaf(chain, 0.999f, 0.001f); }
        if (KIND == 3) { for (int g = 0; g < gap; g += 128) chain = __shfl_xor(chain, 16, 64) + 1.0f; }
        a += chain * 1e-9f;
    }
    float s = chain;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 123.456f) out[threadIdx.x] = s;
}

template <int KIND>
double run(float* out, int waves_per_simd, int tiles, int gap) {
    const int grid = 256 * 4 * waves_per_simd;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(64), 0, 0, out, tiles, gap);
    hipEventRecord(e0);
    for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(64), 0, 0, out, tiles, gap);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 100.0;   // us per launch
}

int main() {
    float* out; hipMalloc(&out, 4096);
    const int tiles = 64;
    printf("%-10s %-6s %-6s %10s %10s\n", "gap kind", "gap", "w/simd", "us", "TFLOP/s");
    for (int kind = 0; kind < 4; ++kind)
        for (int gap : {0, 700, 1400, 2800})
            for (int w : {1, 2, 4}) {
                if (kind == 0 && gap) continue;
                if (kind != 0 && gap == 0) continue;
                double us = kind == 0 ? run<0>(out, w, tiles, gap) : kind == 1 ? run<1>(out, w, tiles, gap) : kind == 2 ? run<2>(out, w, tiles, gap) : run<3>(out, w, tiles, gap);
                double flops = 2048.0 * 144 * tiles * 256 * 4 * w;
                printf("%-10s %-6d %-6d %10.1f %10.1f\n", kind == 0 ? "none" : kind == 1 ? "s_sleep" : kind == 2 ? "valu" : "bpermute", gap, w, us, flops / us / 1e6);
            }
    return 0;
}
