// Does s_memtime (the shader clock counter) keep counting while a CU has nothing to issue?  Calibration of the in-kernel clock probe
// (s_memtime / s_memrealtime) of tools/wgrad_phases.py:   hipcc --offload-arch=gfx950 -O3 tools/micro/clock_gate.hip -o tools/micro/clock_gate && tools/micro/clock_gate
// One wave per CU (256 blocks of 64 threads): kind 0 = a dependent pointer chase through 64 MB (the wave waits for memory almost all the
// time, the CU is idle), 1 = s_sleep loops (parked), 2 = a dependent v_fma chain (always issuing).  Prints shader cycles per ns.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int KIND>
__global__ __launch_bounds__(64) void k(const unsigned* chain, float* out, int iters) {
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    unsigned idx = blockIdx.x * 4099u + threadIdx.x;
    float f = threadIdx.x * 0.5f;
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) idx = chain[idx & ((1u << 24) - 1)];
        if (KIND == 1) __builtin_amdgcn_s_sleep(127);
        if (KIND == 2) f = __builtin_fmaf(f, 0.999f, 0.001f);
    }
    if (idx == 0xdeadbeefu || f == 123.456f) out[0] = f;
    if (threadIdx.x == 0) {
        const unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
        out[2 + 2 * blockIdx.x] = (float)(c1 - c0);
        out[3 + 2 * blockIdx.x] = (float)(r1 - r0);
    }
}

template <int KIND>
void run(const unsigned* chain, float* out, int iters, const char* name) {
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(64), 0, 0, chain, out, iters);
    hipDeviceSynchronize();
    std::vector<float> h(2 + 512);
    hipMemcpy(h.data(), out, h.size() * sizeof(float), hipMemcpyDeviceToHost);
    double c = 0, r = 0;
    for (int b = 0; b < 256; ++b) { c += h[2 + 2 * b]; r += h[3 + 2 * b]; }
    printf("%-46s %8.1f us per wave, %.2f shader cycles per ns (s_memtime / s_memrealtime)\n", name, r / 256 / 100.0, c / (r * 10.0));
}

int main() {
    const size_t n = 1u << 24;
    std::vector<unsigned> h(n);
    unsigned x = 12345;
    for (size_t i = 0; i < n; ++i) { x = x * 1664525u + 1013904223u; h[i] = x & (n - 1); }
    unsigned* chain; float* out;
    hipMalloc(&chain, n * 4); hipMalloc(&out, 4096);
    hipMemcpy(chain, h.data(), n * 4, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {
        run<0>(chain, out, 200, "pointer chase (CU idle, waiting for memory)");
        run<1>(chain, out, 60, "s_sleep (wave parked)");
        run<2>(chain, out, 40000, "dependent v_fma chain (always issuing)");
    }
    return 0;
}
