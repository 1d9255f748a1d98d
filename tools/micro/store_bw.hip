// Store-bandwidth probe: how fast can a kernel WRITE a 512 MB buffer with different store shapes?
//   hipcc --offload-arch=gfx950 -O3 -o store_bw tools/micro/store_bw.hip && ./store_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ __launch_bounds__(256) void fill(float* __restrict__ out, long n4) {   // n4 = number of float4
    const long tid = (long)blockIdx.x * 256 + threadIdx.x, nthreads = (long)gridDim.x * 256;
    const float4 v = make_float4(1.f, 2.f, 3.f, (float)threadIdx.x);
    if (MODE == 0) {            // one dword per lane, 256 contiguous bytes per wave-instruction
        for (long i = tid; i < n4 * 4; i += nthreads) out[i] = v.w;
    } else if (MODE == 1) {     // 16 bytes per lane, 1 KB contiguous per wave-instruction
        for (long i = tid; i < n4; i += nthreads) reinterpret_cast<float4*>(out)[i] = v;
    } else if (MODE == 2) {     // same, nontemporal
        typedef float f4v __attribute__((ext_vector_type(4)));
        const f4v vv = {1.f, 2.f, 3.f, (float)threadIdx.x};
        for (long i = tid; i < n4; i += nthreads) __builtin_nontemporal_store(vv, reinterpret_cast<f4v*>(out) + i);
    } else if (MODE == 3) {     // dword, nontemporal
        for (long i = tid; i < n4 * 4; i += nthreads) __builtin_nontemporal_store(v.w, out + i);
    } else if (MODE == 4) {     // 16 bytes per lane, but every wave-instruction covers 16 segments of 64 B at a 128 B stride (parity-interleaved pixels)
        for (long i = tid; i < n4 / 2; i += nthreads) {                 // idx <= 2*(n4/2 - 64) + 15*8 + 3 + 4 = n4 - 1
            const long base = i & ~63L, l = i & 63;
            const long quad = (l & 3), px = l >> 2;                   // 16 pixels x 4 quads
            const long idx = 2 * base + px * 8 + quad;                  // even pixels of a 2 KB span
            reinterpret_cast<float4*>(out)[idx] = v;
            reinterpret_cast<float4*>(out)[idx + 4] = v;                // odd pixels (second instruction)
        }
    } else if (MODE == 5) {     // 16 B per lane + a 4-byte store by every 4th lane into a second array (the rnorm pattern)
        float* rn = out + n4 * 4;
        for (long i = tid; i < n4; i += nthreads) {
            reinterpret_cast<float4*>(out)[i] = v;
            if ((i & 3) == 0) rn[i >> 2] = v.x;
        }
    }
}

// MODE 0: the persistent conv kernels' store pattern without any compute: 8 x 32-pixel tiles of a (B, H, W, 16) fp32 tensor, wave w
// writes rows 2w, 2w+1 (2 x 16 pixels each), lane (p, q) -> pixel p, channel quad q (1 KB contiguous per instruction), plus the
// 4-byte-per-pixel norm.  MODE 1: the parity-interleaved pattern of the folded-bilinear kernel (wave = parity, 64 B segments at a
// 128 B stride, rows 2r + py).  XCD-banded persistent tile walk as in csrc/conv3x3.hip.
template <int MODE, int WITH_RN>
__global__ __launch_bounds__(256) void tile_store(float* __restrict__ y, float* __restrict__ rn, int B, int H, int W, int n_tiles) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, p = lane & 15, q = lane >> 4;
    const int tiles_x = W / 32, tiles_y = H / 8;
    const int xcd = blockIdx.x & 7, nper = gridDim.x >> 3, band = (n_tiles + 7) >> 3;
    const int t_end = min((xcd + 1) * band, n_tiles);
    const float4 v = make_float4(1.f, 2.f, 3.f, (float)tid);
    for (int t = xcd * band + (blockIdx.x >> 3); t < t_end; t += nper) {
        int tt = t;
        const int txi = tt % tiles_x; tt /= tiles_x;
        const int tyi = tt % tiles_y;
        const int b = tt / tiles_y, y0 = tyi * 8, x0 = txi * 32;
#pragma unroll
        for (int pg = 0; pg < 4; ++pg) {
            int gy, gx;
            if (MODE == 0) { gy = y0 + wave * 2 + (pg >> 1); gx = x0 + (pg & 1) * 16 + p; }
            else { gy = y0 + 2 * pg + (wave >> 1); gx = x0 + 2 * p + (wave & 1); }
            const long pix = ((long)b * H + gy) * W + gx;
            reinterpret_cast<float4*>(y)[pix * 4 + q] = v;
            if (WITH_RN && q == 0) rn[pix] = v.x;
        }
    }
}

template <int MODE, int WITH_RN>
float run_tiles(float* y, float* rn, int B, int H, int W, int blocks) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    const int n_tiles = B * (H / 8) * (W / 32);
    for (int rep = 0; rep < 6; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((tile_store<MODE, WITH_RN>), dim3(blocks), dim3(256), 0, 0, y, rn, B, H, W, n_tiles);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    return best;
}

int main() {
    {
        const int B = 16, H = 512, W = 512;
        float *y, *rn;
        hipMalloc(&y, (size_t)B * H * W * 64);
        hipMalloc(&rn, (size_t)B * H * W * 4);
        for (int blocks : {768, 1024, 2048}) {
            const double by = (double)B * H * W * 64, brn = (double)B * H * W * 4;
            float ms;
            ms = run_tiles<0, 0>(y, rn, B, H, W, blocks); printf("tiles blocks %4d  plain rows            %8.1f us %6.2f TB/s\n", blocks, ms * 1e3, by / (ms * 1e-3) / 1e12);
            ms = run_tiles<0, 1>(y, rn, B, H, W, blocks); printf("tiles blocks %4d  plain rows + norm     %8.1f us %6.2f TB/s\n", blocks, ms * 1e3, (by + brn) / (ms * 1e-3) / 1e12);
            ms = run_tiles<1, 0>(y, rn, B, H, W, blocks); printf("tiles blocks %4d  parity interleaved    %8.1f us %6.2f TB/s\n", blocks, ms * 1e3, by / (ms * 1e-3) / 1e12);
            ms = run_tiles<1, 1>(y, rn, B, H, W, blocks); printf("tiles blocks %4d  parity interl. + norm %8.1f us %6.2f TB/s\n", blocks, ms * 1e3, (by + brn) / (ms * 1e-3) / 1e12);
        }
        hipFree(y); hipFree(rn);
    }

    const long n4 = 32L << 20;   // 512 MB of float4
    float* buf;
    hipMalloc(&buf, n4 * 16 + n4 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[] = {"dword/lane 256B/instr", "b128/lane 1KB/instr", "b128 nontemporal", "dword nontemporal", "b128 64B segments @128B", "b128 + rnorm dword"};
    for (int blocks : {1024, 2048, 4096}) {
        for (int mode = 0; mode < 6; ++mode) {
            float best = 1e9;
            for (int rep = 0; rep < 6; ++rep) {
                hipEventRecord(e0);
                switch (mode) {
                    case 0: hipLaunchKernelGGL(fill<0>, dim3(blocks), dim3(256), 0, 0, buf, n4); break;
                    case 1: hipLaunchKernelGGL(fill<1>, dim3(blocks), dim3(256), 0, 0, buf, n4); break;
                    case 2: hipLaunchKernelGGL(fill<2>, dim3(blocks), dim3(256), 0, 0, buf, n4); break;
                    case 3: hipLaunchKernelGGL(fill<3>, dim3(blocks), dim3(256), 0, 0, buf, n4); break;
                    case 4: hipLaunchKernelGGL(fill<4>, dim3(blocks), dim3(256), 0, 0, buf, n4); break;
                    default: hipLaunchKernelGGL(fill<5>, dim3(blocks), dim3(256), 0, 0, buf, n4); break;
                }
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (rep && ms < best) best = ms;
            }
            const double bytes = (double)n4 * 16 * (mode == 5 ? 1.0625 : 1.0);
            printf("blocks %5d  %-28s %8.1f us  %6.2f TB/s\n", blocks, names[mode], best * 1e3, bytes / (best * 1e-3) / 1e12);
        }
    }
    return 0;
}
