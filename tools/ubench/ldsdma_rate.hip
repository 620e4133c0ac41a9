// LDS-DMA fill rate per CU (tuning tool): what a CU's `global_load_lds_dwordx4` pieces (1 KiB per wave instruction) sustain with NOTHING else in the
// kernel -- the ceiling under the conv / weight-gradient kernels' chunk loops -- by source residency (a 2 MiB window: L2; the whole 1 GiB buffer:
// HBM / MALL), gather shape (a piece = 16 pixels x 64 B out of 128-B pixel lines, as the patch pieces; or 1 KiB contiguous, as the weight pieces),
// waves per CU that issue (1 / 4 / 8) and pieces in flight per wave (counted vmcnt).
// usage: hipcc --offload-arch=gfx950 -O3 ldsdma_rate.hip -o /tmp/ldsdma_rate && /tmp/ldsdma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// GATHER: lane l fetches 16 B of pixel (l >> 2) at byte (l & 3) * 16 of a 128-B line (64 B used per pixel); else lane l fetches bytes 16 l .. 16 l + 15
template <bool GATHER, int INFLIGHT>
__global__ __launch_bounds__(512) void fill(const char* src, size_t window, int pieces_per_wave, unsigned long long* cycles) {
    __shared__ __attribute__((aligned(1024))) char lds[8 * INFLIGHT * 1024];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
    const unsigned base = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)(__attribute__((address_space(3))) void*)lds + wave * INFLIGHT * 1024);
    const size_t piece_bytes = GATHER ? 2048 : 1024;
    const size_t lane_off = GATHER ? (size_t)(lane >> 2) * 128 + (lane & 3) * 16 : (size_t)lane * 16;
    // every wave of the chip walks its own stream of pieces, wrapping inside the window
    size_t pos = (((size_t)blockIdx.x * nw + wave) * 7919u * piece_bytes) % window;
    const size_t stride = (size_t)gridDim.x * nw * piece_bytes;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < pieces_per_wave; ++i) {
        glds16(src + pos + lane_off, base + (i % INFLIGHT) * 1024);
        pos += stride;
        if (pos >= window) pos -= window;
        if constexpr (INFLIGHT == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if constexpr (INFLIGHT == 4) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}
int main() {
    const size_t total = 1ull << 30;
    char* d; unsigned long long* cyc;
    hipMalloc(&d, total); hipMemset(d, 1, total); hipMalloc(&cyc, 256 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("%-28s %-10s %-6s %-9s %10s %12s %12s\n", "source", "shape", "waves", "inflight", "us", "GB/s chip", "B/clk/CU");
    for (size_t window : {(size_t)2 << 20, (size_t)64 << 20, total})
        for (int gather = 0; gather < 2; ++gather)
            for (int waves : {1, 4, 8})
                for (int inflight : {1, 4, 8}) {
                    const int ppw = 2048 / waves * 2;  // pieces per wave: 4 MiB per CU per launch
                    float best = 1e9; unsigned long long c0 = 0;
                    for (int rep = 0; rep < 4; ++rep) {
                        hipEventRecord(e0);
#define L(G, F) hipLaunchKernelGGL(HIP_KERNEL_NAME(fill<G, F>), dim3(256), dim3(64 * waves), 0, 0, d, window, ppw, cyc)
                        if (gather) { if (inflight == 1) L(true, 1); else if (inflight == 4) L(true, 4); else L(true, 8); }
                        else { if (inflight == 1) L(false, 1); else if (inflight == 4) L(false, 4); else L(false, 8); }
                        hipEventRecord(e1); hipEventSynchronize(e1);
                        float ms; hipEventElapsedTime(&ms, e0, e1);
                        if (ms < best) { best = ms; hipMemcpy(&c0, cyc, 8, hipMemcpyDeviceToHost); }
                    }
                    const double bytes = 256.0 * waves * ppw * 1024;
                    printf("%-28s %-10s %-6d %-9d %10.1f %12.0f %12.1f\n", window == total ? "1 GiB (HBM)" : (window == ((size_t)2 << 20) ? "2 MiB window (L2)" : "64 MiB window (MALL)"),
                           gather ? "16px x 64B" : "contiguous", waves, inflight, best * 1e3, bytes / (best * 1e-3) / 1e9, (double)waves * ppw * 1024 / (double)c0);
                }
    return 0;
}
