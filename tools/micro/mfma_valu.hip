// What does a VALU instruction cost beside v_mfma_f32_16x16x4_f32 (the fp32 MFMA, which runs on the vector-FMA lanes)?
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_valu.hip -o tools/micro/mfma_valu && tools/micro/mfma_valu
// Each wave: `iters` x { 16 MFMAs on 16 accumulators, each followed by N filler instructions }.  The fillers are INDEPENDENT of the MFMAs
// and of each other (8 rotating registers), fully unrolled, written in inline asm so that the compiler can neither pack nor unpack nor
// move them: kind 0 = none, 1 = v_fma_f32, 2 = v_pk_fma_f32 (two fp32 FMAs per lane), 3 = v_add_f32, 4 = v_pk_add_f32, 5 = s_nop 0.
// Output: cycles per MFMA slot (one MFMA + its N fillers) per SIMD = wall cycles / (MFMAs issued on the SIMD); 32 = the MFMA alone.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int KIND, int N>
__global__ __launch_bounds__(64) void k(float* out, int iters) {
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();      // shader clock / 100 MHz reference
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float a = threadIdx.x * 0.001f + 1.0f, b = 0.5f + blockIdx.x * 1e-6f;
    f32x2 r[8];
    for (int i = 0; i < 8; ++i) r[i] = (f32x2){a + i, b - i};
    const f32x2 c = {0.999f, 1.001f};
    for (int t = 0; t < iters; ++t) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < N; ++j) {
                f32x2& x = r[(i * N + j) & 7];
                if (KIND == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[0]) : "v"(c[0]), "v"(c[1]));
                if (KIND == 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(c));
                if (KIND == 3) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[0]) : "v"(c[0]));
                if (KIND == 4) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x) : "v"(c));
                if (KIND == 5) asm volatile("s_nop 0");
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += r[i][0] + r[i][1];
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 123.456f) out[threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) {      // the clock this wave ran at: shader cycles per 10 ns of the constant reference counter
        const unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
        out[512] = (float)(c1 - c0);
        out[513] = (float)(r1 - r0);
    }
}

template <int KIND, int N>
void run(float* out, int wps) {
    const int iters = 2000, grid = 256 * 4 * wps;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k<KIND, N>), dim3(grid), dim3(64), 0, 0, out, iters);
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k<KIND, N>), dim3(grid), dim3(64), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / 5;
    const double mfma_per_simd = 16.0 * iters * wps;
    static const char* names[] = {"none", "v_fma_f32", "v_pk_fma_f32", "v_add_f32", "v_pk_add_f32", "s_nop 0"};
    float clk[2];
    hipMemcpy(clk, out + 512, sizeof(clk), hipMemcpyDeviceToHost);
    const double ghz = clk[1] > 0 ? clk[0] / (clk[1] * 10.0) : 0.0;       // shader cycles per ns
    printf("%-14s N=%d  waves/SIMD %d  %9.1f us  %6.1f cycles per MFMA slot at 2.4 GHz  (%5.1f TFLOP/s of MFMA)  in-kernel clock %.2f GHz (s_memtime / s_memrealtime) -> %5.1f of ITS cycles per slot\n",
           names[KIND], N, wps, us, us * 2400.0 / mfma_per_simd, 2048.0 * 16 * iters * grid / us / 1e6, ghz, us * 1e3 * ghz / mfma_per_simd);
}

template <int KIND>
void sweep(float* out) {
    for (int wps : {1, 2}) {
        if (KIND == 0) { run<0, 0>(out, wps); continue; }
        run<KIND, 1>(out, wps); run<KIND, 2>(out, wps); run<KIND, 3>(out, wps); run<KIND, 4>(out, wps); run<KIND, 6>(out, wps);
    }
}

int main() {
    float* out; hipMalloc(&out, 4096);      // (floats 512, 513: the clock probe)
    sweep<0>(out); sweep<1>(out); sweep<2>(out); sweep<3>(out); sweep<4>(out); sweep<5>(out);
    return 0;
}
