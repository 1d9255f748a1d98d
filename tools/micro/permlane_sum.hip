// Is sum_rows4 (v_permlane16_swap / v_permlane32_swap, csrc/ngan_common.h) bit-identical to the shuffle butterfly it replaces?
//   hipcc --offload-arch=gfx950 -O3 -I neuron-gan_amd/csrc -I include tools/micro/permlane_sum.hip -o tools/micro/permlane_sum && tools/micro/permlane_sum
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "ngan_common.h"

__global__ void k(const float* in, float* a, float* b) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float v = in[i];
    float s = v;
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    a[i] = s;
    b[i] = sum_rows4(v);
}

int main() {
    const int n = 64 * 1024;
    std::vector<float> h(n), ha(n), hb(n);
    srand(3);
    for (auto& x : h) x = (float)rand() / RAND_MAX * 2.f - 1.f;
    float *d, *da, *db;
    hipMalloc(&d, n * 4); hipMalloc(&da, n * 4); hipMalloc(&db, n * 4);
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, d, da, db);
    hipMemcpy(ha.data(), da, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(hb.data(), db, n * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n; ++i) bad += memcmp(&ha[i], &hb[i], 4) != 0;
    double ref = 0; for (int l = 0; l < 4; ++l) ref += h[5 + 16 * l];
    printf("permlane_sum: %d of %d lanes differ from the shuffle butterfly; lane 5 = %.7f (fp64 sum of its four rows %.7f)\n", bad, n, hb[5], ref);
    return bad != 0;
}
