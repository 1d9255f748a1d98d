// Semantics check of ds_read_b64_tr_b16 (gfx950): lane i of 16-lane group g must receive column i of rows 4g..4g+3.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
__global__ void k(float* y) {
    __shared__ __attribute__((aligned(16))) __bf16 img[16 * 16];
    for (int i = threadIdx.x; i < 256; i += 64) img[i] = (__bf16)(float)i;
    __syncthreads();
    const int l = threadIdx.x, g = l >> 4, i = l & 15, q = i >> 2, p = i & 3;
    auto ptr = (__attribute__((address_space(3))) bf16x4*)(&img[(4 * g + q) * 16 + 4 * p]);
    bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(ptr);
    for (int j = 0; j < 4; ++j) y[l * 4 + j] = (float)v[j];
}
int main() {
    float* d; float h[256];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) {
        const int g = l >> 4, i = l & 15;
        if (h[l * 4 + j] != (float)((4 * g + j) * 16 + i)) ++bad;
    }
    printf("tr_read: %d mismatches; lane 5 -> %g %g %g %g (expect 5 21 37 53)\n", bad, h[20], h[21], h[22], h[23]);
    return bad != 0;
}
