// Shared helpers for libfalnet_hip.so (gfx950 only; no CUDA / multi-backend paths).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/falnet_hip.h"

void falnet_set_error(const char* fmt, ...);
int falnet_deterministic();  // api.cpp: falnet_set_deterministic() state

// Kernel-selection switches are for A/B experiments and ablations only: the product library (default build) reads NO environment
// variable -- every switch sits at its default and the alternative kernels are not reachable.  `python -m fal_net_amd._build --ab`
// (-DFALNET_AB) builds the experiment library in which the FALNET_* variables named in include/falnet_hip.h are honoured.
#ifdef FALNET_AB
#include <stdlib.h>
static inline const char* falnet_ab_env(const char* name) { return getenv(name); }
#else
static inline const char* falnet_ab_env(const char*) { return nullptr; }
#endif

#define FALNET_CHECK_ARG(cond, ...)                 \
    do {                                            \
        if (!(cond)) {                              \
            falnet_set_error(__VA_ARGS__);          \
            return -1;                              \
        }                                           \
    } while (0)

// Entry of every launch function: make the device that OWNS the caller's stream current on this thread (backward
// runs on autograd's worker thread, whose current device need not be the forward thread's; a launch on a stream of another
// device fails with hipErrorInvalidResourceHandle).  The device comes from the stream itself, so no entry point needs a
// device argument; the NULL stream keeps the thread's current device (falnet_set_device sets it explicitly).
int* falnet_replay_depth();  // api.cpp: > 0 while falnet_replay (which has selected the device once) issues commands on this thread
static inline void falnet_enter_stream(void* stream) {
    if (!stream || *falnet_replay_depth() > 0) return;
    hipDevice_t dev;
    int cur = -1;
    if (hipStreamGetDevice((hipStream_t)stream, &dev) == hipSuccess && hipGetDevice(&cur) == hipSuccess && cur != (int)dev) (void)hipSetDevice((int)dev);
}
#define FALNET_ENTER(stream) falnet_enter_stream(stream)

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

// ---- 16-bit float operand types (bf16: 8 significant bits / f32 range; f16: 11 bits / 6e-5..65504) ----------------
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef short s16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
template <typename T> struct H16;
template <> struct H16<bf16_t> {
    static __device__ __forceinline__ f32x4_t mma16(s16x8_t a, s16x8_t b, f32x4_t c) {  // v_mfma_f32_16x16x32_bf16
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ float lo(unsigned w) { return __uint_as_float(w << 16); }          // element 0 of a packed pair
    static __device__ __forceinline__ float hi(unsigned w) { return __uint_as_float(w & 0xffff0000u); }  // element 1
    static __device__ __forceinline__ unsigned short bits(float v) { return __builtin_bit_cast(unsigned short, (bf16_t)v); }
    static __device__ __forceinline__ f32x16_t mma(s16x8_t a, s16x8_t b, f32x16_t c) {  // v_mfma_f32_32x32x16_bf16
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
    }
};
template <> struct H16<f16_t> {
    static __device__ __forceinline__ f32x4_t mma16(s16x8_t a, s16x8_t b, f32x4_t c) {  // v_mfma_f32_16x16x32_f16
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ float lo(unsigned w) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(w & 0xffffu)); }
    static __device__ __forceinline__ float hi(unsigned w) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(w >> 16)); }
    static __device__ __forceinline__ unsigned short bits(float v) { return __builtin_bit_cast(unsigned short, (f16_t)v); }
    static __device__ __forceinline__ f32x16_t mma(s16x8_t a, s16x8_t b, f32x16_t c) {  // v_mfma_f32_32x32x16_f16
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    }
};
template <typename T> __device__ __forceinline__ unsigned pack16x2(float lo, float hi) {
    return (unsigned)H16<T>::bits(lo) | ((unsigned)H16<T>::bits(hi) << 16);
}
// launch dispatch over the three operand types: X(T) is a statement using the type
#define FALNET_DISPATCH_DTYPE(dtype, X)             \
    do {                                            \
        if ((dtype) == FALNET_BF16) { X(bf16_t); }  \
        else if ((dtype) == FALNET_F16) { X(f16_t); } \
        else { X(float); }                          \
    } while (0)
#define FALNET_DISPATCH_16(dtype, X)                \
    do {                                            \
        if ((dtype) == FALNET_F16) { X(f16_t); }    \
        else { X(bf16_t); }                         \
    } while (0)

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
