// Training-data augmentation on the GPU: the reference does this per sample on the host with PIL + numpy
// (data_transforms.py:46-157, Train_Stage1_K.py:116-128, 4 loader workers) -- at >1000 pairs/s per GPU that loader is the
// bottleneck.  Byte / integer work, HBM bound: one thread per output element, coalesced along x.
//   resample_u8_kernel       : one pass of PIL's 8-bit bicubic resampling (third-party Pillow, src/libImaging/Resample.c
//                              ImagingResampleHorizontal_8bpc / Vertical_8bpc): 22-bit fixed-point coefficients, +2^21 rounding,
//                              >> 22, clip to [0,255]; bit-exact with Image.resize(..., BICUBIC).
//   augment_normalize_kernel : crop + horizontal flip + RandomGamma + RandomBrightness + RandomCBrightness + ArrayToTensor +
//                              Normalize(0,255) + Normalize(mean,1) fused: uint8 HWC in, planar f32 out, float64 arithmetic where
//                              the reference's numpy arrays are float64.
#include <stdint.h>
#include "common.h"

#define DATA_THREADS 256
#define DATA_PRECISION_BITS 22

__global__ __launch_bounds__(DATA_THREADS) void resample_u8_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int H, int W, int C,
                                                                   int out_size, int horizontal, const int32_t* __restrict__ bounds,
                                                                   const int32_t* __restrict__ kk, int ksize) {
    const int OW = horizontal ? out_size : W, OH = horizontal ? H : out_size;
    const int64_t total = (int64_t)OH * OW * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C), x = (int)((i / C) % OW), y = (int)(i / ((int64_t)C * OW));
        const int o = horizontal ? x : y;
        const int first = bounds[2 * o], n = bounds[2 * o + 1];
        const int32_t* k = kk + (int64_t)o * ksize;
        int ss = 1 << (DATA_PRECISION_BITS - 1);
        if (horizontal) {
            const uint8_t* p = src + ((int64_t)y * W + first) * C + c;
            for (int t = 0; t < n; ++t) ss += (int)p[(int64_t)t * C] * k[t];
        } else {
            const uint8_t* p = src + ((int64_t)first * W + x) * C + c;
            for (int t = 0; t < n; ++t) ss += (int)p[(int64_t)t * W * C] * k[t];
        }
        ss >>= DATA_PRECISION_BITS;  // arithmetic shift, as clip8() in Resample.c
        dst[i] = (uint8_t)(ss < 0 ? 0 : (ss > 255 ? 255 : ss));
    }
}

struct AugArgs {
    int H, W, x1, y1, th, tw, flip;
    double gamma, bright, cb[3];  // <= 0: transform not applied
    float mean[3];
};

__global__ __launch_bounds__(DATA_THREADS) void augment_normalize_kernel(const uint8_t* __restrict__ src, float* __restrict__ dst, AugArgs a) {
    const int64_t total = (int64_t)3 * a.th * a.tw;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % a.tw), y = (int)((i / a.tw) % a.th), c = (int)(i / ((int64_t)a.tw * a.th));
        const int sx = a.x1 + (a.flip ? a.tw - 1 - x : x), sy = a.y1 + y;  // np.fliplr of the crop
        const uint8_t u = src[((int64_t)sy * a.W + sx) * 3 + c];
        double v = (double)u;
        const bool is_float = a.gamma > 0.0 || a.bright > 0.0;  // the reference's array left uint8 only if neither fired
        if (a.gamma > 0.0) v = 255.0 * pow(v / 255.0, a.gamma);
        if (a.bright > 0.0) {
            v = v * a.bright;
            if (v > 255.0) v = 255.0;
        }
        if (a.cb[0] > 0.0) {
            if (is_float) {
                v = v * a.cb[c];
                if (v > 255.0) v = 255.0;
            } else {
                // assignment of a float64 product into the uint8 array (data_transforms.py:155): C conversion, low 8 bits
                v = (double)(uint8_t)(int)(v * a.cb[c]);
            }
        }
        float t = (float)v;
        t = (t - 0.0f) / 255.0f;
        t = (t - a.mean[c]) / 1.0f;
        dst[i] = t;
    }
}

static inline int data_grid(int64_t n) {
    int64_t g = (n + DATA_THREADS - 1) / DATA_THREADS;
    return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

extern "C" int falnet_resample_u8(const uint8_t* src, uint8_t* dst, int H, int W, int C, int out_size, int horizontal, const int32_t* bounds,
                                  const int32_t* kk, int ksize, void* stream) {
    FALNET_ENTER(stream);
    FALNET_CHECK_ARG(src && dst && bounds && kk && H > 0 && W > 0 && C > 0 && out_size > 0 && ksize > 0, "resample_u8: bad argument");
    const int64_t total = (int64_t)(horizontal ? H : out_size) * (horizontal ? out_size : W) * C;
    hipLaunchKernelGGL(resample_u8_kernel, dim3(data_grid(total)), dim3(DATA_THREADS), 0, (hipStream_t)stream, src, dst, H, W, C, out_size,
                       horizontal ? 1 : 0, bounds, kk, ksize);
    FALNET_RETURN_LAUNCH();
}

extern "C" int falnet_augment_normalize(const uint8_t* src, int H, int W, int x1, int y1, int th, int tw, int flip, double gamma, double bright,
                                        double cb0, double cb1, double cb2, float mean0, float mean1, float mean2, float* dst, void* stream) {
    FALNET_ENTER(stream);
    FALNET_CHECK_ARG(src && dst && th > 0 && tw > 0 && x1 >= 0 && y1 >= 0 && x1 + tw <= W && y1 + th <= H, "augment_normalize: crop outside the image");
    AugArgs a{H, W, x1, y1, th, tw, flip ? 1 : 0, gamma, bright, {cb0, cb1, cb2}, {mean0, mean1, mean2}};
    hipLaunchKernelGGL(augment_normalize_kernel, dim3(data_grid((int64_t)3 * th * tw)), dim3(DATA_THREADS), 0, (hipStream_t)stream, src, dst, a);
    FALNET_RETURN_LAUNCH();
}
