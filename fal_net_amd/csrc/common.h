// Shared helpers for libfalnet_hip.so (gfx950 only; no CUDA / multi-backend paths).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/falnet_hip.h"

void falnet_set_error(const char* fmt, ...);

#define FALNET_CHECK_ARG(cond, ...)                 \
    do {                                            \
        if (!(cond)) {                              \
            falnet_set_error(__VA_ARGS__);          \
            return -1;                              \
        }                                           \
    } while (0)

// kernel launches never synchronise; a launch-configuration error surfaces here
#define FALNET_RETURN_LAUNCH()                                                   \
    do {                                                                         \
        hipError_t e_ = hipGetLastError();                                       \
        if (e_ != hipSuccess) {                                                  \
            falnet_set_error("%s: launch failed: %s", __func__, hipGetErrorString(e_)); \
            return (int)e_;                                                      \
        }                                                                        \
        return 0;                                                                \
    } while (0)

typedef __bf16 bf16_t;
typedef _Float16 f16_t;

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16_t v) { return (float)v; }
__device__ __forceinline__ float to_f32(f16_t v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return (bf16_t)v; }
template <> __device__ __forceinline__ f16_t from_f32<f16_t>(float v) { return (f16_t)v; }

// wave64 sum via DPP-free shuffles, then one LDS hop across the block's waves
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// block-wide sum for blockDim.x <= 1024; result valid in thread 0
__device__ __forceinline__ float block_sum(float v, float* red /* >= 16 floats of LDS */) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) red[w] = v;
    __syncthreads();
    float r = 0.f;
    if (threadIdx.x == 0) {
        const int nw = (blockDim.x + 63) >> 6;
        for (int i = 0; i < nw; ++i) r += red[i];
    }
    return r;
}
