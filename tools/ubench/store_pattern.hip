// Store-pattern microbenchmark (tuning tool): the NHWC epilogue's store shape (a lane owns a pixel: every store instruction writes 32 B of each of
// 32 pixel lines, four instructions complete a 128-B line) against whole-line stores (eight lanes per pixel line: every instruction writes 1 KiB
// contiguous).  usage: hipcc --offload-arch=gfx950 -O3 store_pattern.hip -o store_pattern && ./store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(256) void k(uint4* out, long npix) {  // 128 B per pixel
    const int lane = threadIdx.x & 63;
    const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((long)gridDim.x * blockDim.x) >> 6;
    const uint4 v = make_uint4(lane, 1, 2, 3);
    for (long p0 = wave * 32; p0 < npix; p0 += nwaves * 32) {
        if (MODE == 0) {  // epilogue shape: lane (r, h) -> pixel r, segments 2 j + h
            const int r = lane & 31, h = lane >> 5;
#pragma unroll
            for (int j = 0; j < 4; ++j) out[(p0 + r) * 8 + 2 * j + h] = v;
        } else if (MODE == 1) {  // whole lines: lane -> (pixel l >> 3, segment l & 7)
#pragma unroll
            for (int j = 0; j < 4; ++j) out[(p0 + 8 * j + (lane >> 3)) * 8 + (lane & 7)] = v;
        } else {  // epilogue shape with 64 B per pixel and instruction (lane (r, h): segments 4 (j & 1) ... two instructions per half line)
            const int r = lane & 31, h = lane >> 5;
#pragma unroll
            for (int j = 0; j < 4; ++j) out[(p0 + r) * 8 + 4 * h + j] = v;
        }
    }
}
int main() {
    const long npix = 8L * 256 * 512;  // 134 MB at 128 B per pixel
    uint4* d;
    hipMalloc(&d, npix * 128);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int grid : {1024, 2048, 8192}) {
        for (int mode = 0; mode < 3; ++mode) {
            float best = 1e9;
            for (int rep = 0; rep < 5; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, d, npix);
                else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, d, npix);
                else hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, d, npix);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            printf("grid %5d mode %d: %.1f us  %.2f TB/s\n", grid, mode, best * 1e3, npix * 128.0 / best / 1e9);
        }
    }
    return 0;
}
