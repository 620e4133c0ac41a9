// HBM-bound loss / optimiser kernels for gfx950 (reference: loss_functions.py:52-101,
// Train_Stage1_K.py:180,248-262, Train_Stage2_K.py:248-253,319-324).  All f32 arithmetic.
// Reductions: grid-stride partial sums in registers -> wave shuffle -> one LDS hop -> ONE float
// atomic per workgroup on the scalar (grids capped at RED_BLOCKS so the single-address atomic
// chain stays in the microsecond range; cdna guide G12).
#include <math.h>
#include <pthread.h>
#include "common.h"

#define RED_THREADS 256
#define RED_BLOCKS 512

static inline int red_grid(int64_t work_items) {
    int64_t g = (work_items + RED_THREADS - 1) / RED_THREADS;
    return (int)(g < 1 ? 1 : (g > RED_BLOCKS ? RED_BLOCKS : g));
}


__device__ __forceinline__ float sgn(float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); }

// Tail of every scalar reduction: thread 0 of each workgroup holds the workgroup's (scaled) sum.
//   default       : one f32 atomic per workgroup on the scalar -- the order of the <= RED_BLOCKS adds varies from run to run;
//   deterministic : falnet_set_deterministic(1) -- every workgroup publishes its sum (agent-scope atomic store), takes a ticket,
//                   and the LAST arriver adds the partials in workgroup order onto the scalar: bit-identical from run to run.
//                   The scratch (partials + ticket) is per STREAM: `det` = 1 + the slot the launcher assigned to its stream (red_det_slot:
//                   up to RED_SLOTS streams per process), so reductions on different streams -- a loss on the auxiliary stream, two models
//                   in one process -- never share a ticket; launches of one stream are ordered by the stream.
#define RED_SLOTS 72  // torch hands out streams from pools of 32 per priority and device: 2 x 32 + the default stream + spares
__device__ float g_red_partial[RED_SLOTS][RED_BLOCKS];
__device__ unsigned g_red_ticket[RED_SLOTS] = {};
__device__ __forceinline__ void red_finish(float* out, float v, int det) {
    if (threadIdx.x != 0) return;
    if (!det) {
        atomicAdd(out, v);
        return;
    }
    float* const partial = g_red_partial[det - 1];
    unsigned* const ticket = &g_red_ticket[det - 1];
    __hip_atomic_store(&partial[blockIdx.x], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence();
    const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (t + 1 == gridDim.x) {
        __threadfence();
        float s = 0.f;
        for (unsigned i = 0; i < gridDim.x; ++i) s += __hip_atomic_load(&partial[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        out[0] += s;  // the only writer of this launch; earlier launches are ordered by the stream
        __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
// 0 outside deterministic mode, else 1 + the scratch slot of `stream`: assigned on first use; when every slot has an owner the LEAST RECENTLY
// USED one is handed on (a long-lived process that rotates through torch's stream pools must not start failing; the scratch and
// ticket are __device__ globals, i.e. one set per device already, and a slot is only shared by two streams of one device that are BOTH in a
// reduction at the same time after more than RED_SLOTS distinct streams have carried one -- far beyond the 65 handles torch can hand out).
static int red_det_slot(void* stream) {
    if (!falnet_deterministic()) return 0;
    static void* owner[RED_SLOTS] = {};
    static unsigned long long last[RED_SLOTS] = {};
    static unsigned long long tick = 0;
    static pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER;
    pthread_mutex_lock(&mu);
    int slot = -1, lru = 0;
    for (int i = 0; i < RED_SLOTS; ++i) {
        if (last[i] && owner[i] == stream) slot = i;
        if (last[i] < last[lru]) lru = i;  // (never-used slots have last = 0: taken first)
    }
    if (slot < 0) {
        slot = lru;
        owner[slot] = stream;
    }
    last[slot] = ++tick;
    pthread_mutex_unlock(&mu);
    return slot + 1;
}
#define FALNET_DET_SLOT(var, stream)                                                                                      \
    const int var = red_det_slot(stream);                                                                                 \
    FALNET_CHECK_ARG(var >= 0, "deterministic reductions: more than %d streams carry scalar reductions in this process", RED_SLOTS)

static inline int zero_scalar_if(float* out, int accumulate, hipStream_t s) {
    if (!accumulate) {
        hipError_t e = hipMemsetAsync(out, 0, sizeof(float), s);
        if (e != hipSuccess) return (int)e;
    }
    return 0;
}

// ------------------------------------------------------------------ L1 (loss_functions.py:53)
__global__ __launch_bounds__(RED_THREADS) void l1_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                             const float* __restrict__ mask, int C, int64_t HW,
                                                             int64_t total, float scale, float* out,
                                                             const float* __restrict__ gscale = nullptr, float* __restrict__ ga = nullptr, int det = 0,
                                                             const float* __restrict__ gadd = nullptr) {
    __shared__ float red[16];
    float acc = 0.f;
    const float gs = ga ? scale * (gscale ? gscale[0] : 1.f) : 0.f;  // ga: the gradient gscale * scale * sign(a - b) (+ gadd) in the same pass (unmasked)
    if (!mask && (total & 3) == 0 &&
        ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(ga) | reinterpret_cast<uintptr_t>(gadd)) & 15) == 0) {  // 16-byte loads
        const float4* a4 = reinterpret_cast<const float4*>(a);
        const float4* b4 = reinterpret_cast<const float4*>(b);
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (total >> 2); i += (int64_t)gridDim.x * blockDim.x) {
            const float4 x = a4[i], y = b4[i];
            const float d0 = x.x - y.x, d1 = x.y - y.y, d2 = x.z - y.z, d3 = x.w - y.w;
            acc += fabsf(d0) + fabsf(d1) + fabsf(d2) + fabsf(d3);
            if (ga) {
                const float4 o = gadd ? reinterpret_cast<const float4*>(gadd)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
                reinterpret_cast<float4*>(ga)[i] = make_float4(sgn(d0) * gs + o.x, sgn(d1) * gs + o.y, sgn(d2) * gs + o.z, sgn(d3) * gs + o.w);
            }
        }
        const float s4 = block_sum(acc, red);
        red_finish(out, s4 * scale, det);
        return;
    }
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        if (ga) ga[i] = sgn(a[i] - b[i]) * gs + (gadd ? gadd[i] : 0.f);
        float d = fabsf(a[i] - b[i]);
        if (mask) {
            const int64_t bi = i / (C * HW), p = i % HW;
            d *= mask[bi * HW + p];
        }
        acc += d;
    }
    const float s = block_sum(acc, red);
    red_finish(out, s * scale, det);
}

__global__ __launch_bounds__(RED_THREADS) void l1_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                             const float* __restrict__ mask, int C, int64_t HW,
                                                             int64_t total, float scale,
                                                             const float* __restrict__ gscale, float* __restrict__ ga,
                                                             int accumulate) {
    const float gs = gscale ? gscale[0] * scale : scale;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const float d = a[i] - b[i];
        float g = d > 0.f ? gs : (d < 0.f ? -gs : 0.f);
        if (mask) {
            const int64_t bi = i / (C * HW), p = i % HW;
            g *= mask[bi * HW + p];
        }
        ga[i] = accumulate ? ga[i] + g : g;
    }
}

// ------------------------------------------------------------------ perceptual MSE on NHWC (loss_functions.py:61-65)
template <typename T>  // 16-bit operand types
__device__ __forceinline__ void diff8(const T* a, const T* b, float (&d)[8]) {
    const uint4 x = *reinterpret_cast<const uint4*>(a), y = *reinterpret_cast<const uint4*>(b);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned u = (&x.x)[i], v = (&y.x)[i];
        d[2 * i] = H16<T>::lo(u) - H16<T>::lo(v);
        d[2 * i + 1] = H16<T>::hi(u) - H16<T>::hi(v);
    }
}
__device__ __forceinline__ void diff8(const float* a, const float* b, float (&d)[8]) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const float4 x = reinterpret_cast<const float4*>(a)[h], y = reinterpret_cast<const float4*>(b)[h];
        d[4 * h] = x.x - y.x;
        d[4 * h + 1] = x.y - y.y;
        d[4 * h + 2] = x.z - y.z;
        d[4 * h + 3] = x.w - y.w;
    }
}
template <typename T>  // 16-bit operand types
__device__ __forceinline__ void put8(T* p, const float (&v)[8]) {
    uint4 q;
#pragma unroll
    for (int i = 0; i < 4; ++i) (&q.x)[i] = pack16x2<T>(v[2 * i], v[2 * i + 1]);
    *reinterpret_cast<uint4*>(p) = q;
}
__device__ __forceinline__ void put8(float* p, const float (&v)[8]) {
    reinterpret_cast<float4*>(p)[0] = make_float4(v[0], v[1], v[2], v[3]);
    reinterpret_cast<float4*>(p)[1] = make_float4(v[4], v[5], v[6], v[7]);
}

// 8 elements (16 B bf16 / 32 B f32) per thread and iteration when `total` and the pointers allow (VEC), else one
// ga != nullptr: the gradient gscale[0] * 2 * gsc * (a - b) is written in the same pass (the fused training step knows its upstream
// scalars before the forward runs: one read of the feature maps instead of two)
template <typename T, bool VEC>
__global__ __launch_bounds__(RED_THREADS) void mse_fwd_kernel(const T* __restrict__ a, const T* __restrict__ b,
                                                              int64_t total, float scale, float* out,
                                                              const float* __restrict__ gscale = nullptr, float gsc = 0.f, T* __restrict__ ga = nullptr, int det = 0) {
    __shared__ float red[16];
    float acc = 0.f;
    const float gs = ga ? 2.f * gsc * (gscale ? gscale[0] : 1.f) : 0.f;
    if constexpr (VEC) {
        // four groups of eight elements per thread and iteration: eight independent 16-B loads in flight before the first use (a grid of RED_BLOCKS
        // workgroups with one group per thread keeps ~4 MB in flight, half of what HBM's latency x bandwidth wants: 30 us for the 100 MB of slice 1)
        const int64_t n8 = total >> 3, stride = (int64_t)gridDim.x * blockDim.x;
        int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
        for (; i + 3 * stride < n8; i += 4 * stride) {
            float d[4][8];
#pragma unroll
            for (int u = 0; u < 4; ++u) diff8(a + 8 * (i + u * stride), b + 8 * (i + u * stride), d[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int j = 0; j < 8; ++j) acc += d[u][j] * d[u][j];
                if (ga) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) d[u][j] *= gs;
                    put8(ga + 8 * (i + u * stride), d[u]);
                }
            }
        }
        for (; i < n8; i += stride) {
            float d[8];
            diff8(a + 8 * i, b + 8 * i, d);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += d[j] * d[j];
            if (ga) {
#pragma unroll
                for (int j = 0; j < 8; ++j) d[j] *= gs;
                put8(ga + 8 * i, d);
            }
        }
    } else {
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
            const float d = to_f32(a[i]) - to_f32(b[i]);
            acc += d * d;
            if (ga) ga[i] = from_f32<T>(gs * d);
        }
    }
    const float s = block_sum(acc, red);
    red_finish(out, s * scale, det);
}

template <typename T, bool VEC>
__global__ __launch_bounds__(RED_THREADS) void mse_bwd_kernel(const T* __restrict__ a, const T* __restrict__ b,
                                                              int64_t total, float scale,
                                                              const float* __restrict__ gscale, T* __restrict__ ga) {
    const float gs = 2.f * scale * (gscale ? gscale[0] : 1.f);
    if constexpr (VEC) {
        const int64_t n8 = total >> 3;
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
            float d[8];
            diff8(a + 8 * i, b + 8 * i, d);
#pragma unroll
            for (int j = 0; j < 8; ++j) d[j] *= gs;
            put8(ga + 8 * i, d);
        }
    } else {
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
            ga[i] = from_f32<T>(gs * (to_f32(a[i]) - to_f32(b[i])));
    }
}

// The perceptual term's three slices in ONE launch (loss_functions.py:61-65: sum over the VGG slices of the mean squared difference, each with its
// own 1 / numel): blocks [begin_k, begin_{k+1}) work on tensor k -- the two small slices are latency-, not bandwidth-bound as launches of their own
// (10-15 us for 25-50 MB).  VEC layout only (16-B aligned, multiples of 8 elements: NHWC feature maps); value AND gradient, as mse_fwd_kernel with ga.
struct Mse3Args {
    const void* a[3];
    const void* b[3];
    void* ga[3];
    int64_t n8[3];
    float scale_out[3], scale_grad[3];
    int begin[4];
};
template <typename T>
__global__ __launch_bounds__(RED_THREADS) void mse3_fwd_bwd_kernel(Mse3Args m, float* out, const float* __restrict__ gscale, int det) {
    __shared__ float red[16];
    const int k = (int)blockIdx.x >= m.begin[2] ? 2 : ((int)blockIdx.x >= m.begin[1] ? 1 : 0);
    const T* a = reinterpret_cast<const T*>(m.a[k]);
    const T* b = reinterpret_cast<const T*>(m.b[k]);
    T* ga = reinterpret_cast<T*>(m.ga[k]);
    const float gs = 2.f * m.scale_grad[k] * (gscale ? gscale[0] : 1.f);
    const int64_t n8 = m.n8[k], stride = (int64_t)(m.begin[k + 1] - m.begin[k]) * blockDim.x;
    int64_t i = (int64_t)((int)blockIdx.x - m.begin[k]) * blockDim.x + threadIdx.x;
    float acc = 0.f;
    for (; i + 3 * stride < n8; i += 4 * stride) {
        float d[4][8];
#pragma unroll
        for (int u = 0; u < 4; ++u) diff8(a + 8 * (i + u * stride), b + 8 * (i + u * stride), d[u]);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += d[u][j] * d[u][j];
#pragma unroll
            for (int j = 0; j < 8; ++j) d[u][j] *= gs;
            put8(ga + 8 * (i + u * stride), d[u]);
        }
    }
    for (; i < n8; i += stride) {
        float d[8];
        diff8(a + 8 * i, b + 8 * i, d);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += d[j] * d[j];
#pragma unroll
        for (int j = 0; j < 8; ++j) d[j] *= gs;
        put8(ga + 8 * i, d);
    }
    const float s = block_sum(acc, red);
    red_finish(out, s * m.scale_out[k], det);
}

// ------------------------------------------------------------------ edge-aware smoothness (loss_functions.py:70-101)
struct SmoothArgs {
    const float* img;
    const float* disp;
    int B, H, W, x0, x1;
    float gamma;
};

__device__ __forceinline__ float gray_at(const SmoothArgs& s, int b, int y, int x) {
    if (y < 0 || y >= s.H || x < s.x0 || x >= s.x1) return 0.f;  // zero padding applied AFTER the crop
    const int64_t HW = (int64_t)s.H * s.W, o = (int64_t)y * s.W + x;
    const float* im = s.img + (int64_t)b * 3 * HW;
    return 0.299f * (im[o] + 0.411f) + 0.587f * (im[HW + o] + 0.432f) + 0.114f * (im[2 * HW + o] + 0.45f);
}
__device__ __forceinline__ float disp_at(const SmoothArgs& s, int b, int y, int x) {
    if (y < 0 || y >= s.H || x < s.x0 || x >= s.x1) return 0.f;
    return s.disp[((int64_t)b * s.H + y) * s.W + x];
}
__device__ __forceinline__ float wx_at(const SmoothArgs& s, int b, int y, int x) {  // exp(-gamma |dx_img|)
    const float g = -gray_at(s, b, y, x - 1) + 2.f * gray_at(s, b, y, x) - gray_at(s, b, y, x + 1);
    return __expf(-s.gamma * fabsf(g));
}
__device__ __forceinline__ float wy_at(const SmoothArgs& s, int b, int y, int x) {
    const float g = -gray_at(s, b, y - 1, x) + 2.f * gray_at(s, b, y, x) - gray_at(s, b, y + 1, x);
    return __expf(-s.gamma * fabsf(g));
}

// Tiled form: a workgroup stages the grayscale and disparity of a 16 x 64 tile (+ halo 2 / 1) in LDS ONCE and every pixel reads its
// neighbours from there -- the first kernels recomputed the gray value of every neighbour from three planar loads with bounds checks
// (23 loads per pixel forward, 59 backward: 25 us each for 13 MB of input).  Sixteen rows per tile since round 6 (four per thread): a 4-row tile
// staged 8 rows of gray for 4 of output behind two barriers (32 us for the fused forward + adjoint at B = 8, 256 x 512).
#define SM_TY 16
#define SM_RPT (SM_TY * SM_TX / RED_THREADS)  // rows per thread
#define SM_TX 64
#define SM_GW (SM_TX + 4)
#define SM_GH (SM_TY + 4)
#define SM_DW (SM_TX + 2)
#define SM_DH (SM_TY + 2)
__device__ __forceinline__ void smooth_stage_tile(const SmoothArgs& s, int b, int ty0, int tx0, float* gt, float* dt) {
    for (int i = threadIdx.x; i < SM_GH * SM_GW; i += blockDim.x) gt[i] = gray_at(s, b, ty0 - 2 + i / SM_GW, tx0 - 2 + i % SM_GW);
    for (int i = threadIdx.x; i < SM_DH * SM_DW; i += blockDim.x) dt[i] = disp_at(s, b, ty0 - 1 + i / SM_DW, tx0 - 1 + i % SM_DW);
}
// weights at tile-local (ly, lx) (may be -1 .. SM_T? : one pixel outside the tile)
__device__ __forceinline__ float tile_wx(const float* gt, float gamma, int ly, int lx) {
    const float* g = gt + (ly + 2) * SM_GW + lx + 2;
    return __expf(-gamma * fabsf(-g[-1] + 2.f * g[0] - g[1]));
}
__device__ __forceinline__ float tile_wy(const float* gt, float gamma, int ly, int lx) {
    const float* g = gt + (ly + 2) * SM_GW + lx + 2;
    return __expf(-gamma * fabsf(-g[-SM_GW] + 2.f * g[0] - g[SM_GW]));
}

__global__ __launch_bounds__(RED_THREADS) void smooth_fwd_kernel(SmoothArgs s, float scale, float* out, int det = 0) {
    __shared__ float red[16];
    __shared__ float gt[SM_GH * SM_GW], dt[SM_DH * SM_DW];
    const int Wc = s.x1 - s.x0;
    const int tiles_x = (Wc + SM_TX - 1) / SM_TX, tiles_y = (s.H + SM_TY - 1) / SM_TY;
    const int ntiles = s.B * tiles_y * tiles_x;
    const int lx = threadIdx.x % SM_TX, ly0 = threadIdx.x / SM_TX;
    float acc = 0.f;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int tx0 = s.x0 + (tile % tiles_x) * SM_TX, ty0 = ((tile / tiles_x) % tiles_y) * SM_TY, b = tile / (tiles_x * tiles_y);
        __syncthreads();  // previous tile consumed
        smooth_stage_tile(s, b, ty0, tx0, gt, dt);
        __syncthreads();
#pragma unroll
        for (int rr = 0; rr < SM_RPT; ++rr) {
            const int ly = ly0 + rr * (RED_THREADS / SM_TX);
            const int x = tx0 + lx, y = ty0 + ly;
            if (x < s.x1 && y < s.H) {
                const float* dp = dt + (ly + 1) * SM_DW + lx + 1;
                const float d = dp[0];
                const float ax = fabsf(d - dp[1]) + fabsf(d - dp[-1]);
                const float ay = fabsf(d - dp[-SM_DW]) + fabsf(d - dp[SM_DW]);
                acc += ax * tile_wx(gt, s.gamma, ly, lx) + ay * tile_wy(gt, s.gamma, ly, lx);
            }
        }
    }
    const float r = block_sum(acc, red);
    red_finish(out, r * scale, det);
}

// gather form of the adjoint: every pixel sums its own four terms and the one term each of its
// four in-window neighbours holds on it.  Tiles cover all W columns (zero outside the window).
// fwd_out != nullptr: the loss itself (smooth_fwd_kernel's sum, times scale) is accumulated from the same tiles
__global__ __launch_bounds__(RED_THREADS) void smooth_bwd_kernel(SmoothArgs s, float scale,
                                                                 const float* __restrict__ gscale,
                                                                 float* __restrict__ gdisp, int accumulate, float* fwd_out = nullptr, int det = 0) {
    __shared__ float gt[SM_GH * SM_GW], dt[SM_DH * SM_DW];
    __shared__ float red[16];
    float facc = 0.f;
    const float gs = scale * (gscale ? gscale[0] : 1.f);
    const int tiles_x = (s.W + SM_TX - 1) / SM_TX, tiles_y = (s.H + SM_TY - 1) / SM_TY;
    const int ntiles = s.B * tiles_y * tiles_x;
    const int lx = threadIdx.x % SM_TX, ly0 = threadIdx.x / SM_TX;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int tx0 = (tile % tiles_x) * SM_TX, ty0 = ((tile / tiles_x) % tiles_y) * SM_TY, b = tile / (tiles_x * tiles_y);
        __syncthreads();
        const bool any = tx0 < s.x1 && tx0 + SM_TX > s.x0;  // block-uniform: tile touches the window
        if (any) smooth_stage_tile(s, b, ty0, tx0, gt, dt);
        __syncthreads();
#pragma unroll
        for (int rr = 0; rr < SM_RPT; ++rr) {
            const int ly = ly0 + rr * (RED_THREADS / SM_TX);
            const int x = tx0 + lx, y = ty0 + ly;
            if (x >= s.W || y >= s.H) continue;
            float g = 0.f;
            if (x >= s.x0 && x < s.x1) {
                const float* dp = dt + (ly + 1) * SM_DW + lx + 1;
                const float d = dp[0], dl = dp[-1], dr = dp[1], du = dp[-SM_DW], dd = dp[SM_DW];
                const float wx = tile_wx(gt, s.gamma, ly, lx), wy = tile_wy(gt, s.gamma, ly, lx);
                g += (sgn(d - dr) + sgn(d - dl)) * wx + (sgn(d - du) + sgn(d - dd)) * wy;
                if (fwd_out) facc += (fabsf(d - dr) + fabsf(d - dl)) * wx + (fabsf(d - du) + fabsf(d - dd)) * wy;
                if (x - 1 >= s.x0) g -= sgn(dl - d) * tile_wx(gt, s.gamma, ly, lx - 1);  // left pixel's dx_d  = d[x-1]-d[x]
                if (x + 1 < s.x1) g -= sgn(dr - d) * tile_wx(gt, s.gamma, ly, lx + 1);   // right pixel's dx1_d = d[x+1]-d[x]
                if (y - 1 >= 0) g -= sgn(du - d) * tile_wy(gt, s.gamma, ly - 1, lx);     // upper pixel's dy1_d = d[y-1]-d[y]
                if (y + 1 < s.H) g -= sgn(dd - d) * tile_wy(gt, s.gamma, ly + 1, lx);    // lower pixel's dy_d  = d[y+1]-d[y]
                g *= gs;
            }
            const int64_t i = ((int64_t)b * s.H + y) * s.W + x;
            gdisp[i] = accumulate ? gdisp[i] + g : g;
        }
    }
    if (fwd_out) {  // block-uniform
        __syncthreads();
        const float r = block_sum(facc, red);
        red_finish(fwd_out, r * scale, det);
    }
}

// ------------------------------------------------------------------ mask mix (loss_functions.py:55)
__global__ __launch_bounds__(RED_THREADS) void mask_mix_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                               const float* __restrict__ m, float* __restrict__ out,
                                                               int C, int64_t HW, int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t bi = i / (C * HW), p = i % HW;
        const float mv = m[bi * HW + p];
        out[i] = mv * a[i] + (1.f - mv) * b[i];
    }
}

// ------------------------------------------------------------------ Stage-2 occlusion masks and mirror-loss weights
// (Train_Stage2_K.py:296-302, :316-324): single-channel planar maps, no gradient.
//   occlusion: O = a * b, forced to 1 in the column window [x0, x1)   (O_L[:, :, :, 0:0.2W] = 1 / O_R[:, :, :, 0.8W:] = 1)
//   weight   : w = (1 - O) / max_b(teacher disparity) inside [x0, x1), 0 outside -- the mirror loss is then the masked L1
//              sum(w |d - d_teacher|) / (B H (x1 - x0)) of falnet_l1_fwd / _bwd
__global__ __launch_bounds__(RED_THREADS) void occlusion_mask_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                                     float* __restrict__ out, int W, int x0, int x1, int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % W);
        out[i] = (x >= x0 && x < x1) ? 1.f : a[i] * b[i];
    }
}
__global__ __launch_bounds__(RED_THREADS) void mirror_weight_kernel(const float* __restrict__ occ, const float* __restrict__ rowmax,
                                                                    float* __restrict__ out, int64_t HW, int W, int x0, int x1, int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % W);
        out[i] = (x >= x0 && x < x1) ? (1.f - occ[i]) * (1.f / rowmax[i / HW]) : 0.f;
    }
}

// ------------------------------------------------------------------ fused flat Adam (torch.optim.Adam, Train_Stage1_K.py:180)
// 7 f32 streams per element (read p,g,m,v; write p,m,v): 28 B/param, pure HBM streaming, float4 lanes.
__global__ __launch_bounds__(RED_THREADS) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                           float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                           float step_size, float b1, float b2, float eps,
                                                           float rsqrt_bc2, float grad_scale) {
    const int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        float4 pp = reinterpret_cast<float4*>(p)[i], gg = reinterpret_cast<const float4*>(g)[i];
        float4 mm = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
        float* P = &pp.x;
        float* G = &gg.x;
        float* M = &mm.x;
        float* V = &vv.x;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float gr = G[j] * grad_scale;
            M[j] = b1 * M[j] + (1.f - b1) * gr;
            V[j] = b2 * V[j] + (1.f - b2) * gr * gr;
            P[j] -= step_size * M[j] / (sqrtf(V[j]) * rsqrt_bc2 + eps);
        }
        reinterpret_cast<float4*>(p)[i] = pp;
        reinterpret_cast<float4*>(m)[i] = mm;
        reinterpret_cast<float4*>(v)[i] = vv;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const int64_t i = (n4 << 2) + threadIdx.x;
        const float gr = g[i] * grad_scale;
        const float mi = b1 * m[i] + (1.f - b1) * gr, vi = b2 * v[i] + (1.f - b2) * gr * gr;
        m[i] = mi;
        v[i] = vi;
        p[i] -= step_size * mi / (sqrtf(vi) * rsqrt_bc2 + eps);
    }
}

// Adam with its hyper-state on the device: state = {lr, t}.  A captured hipGraph replays the same launch
// arguments, so the step count (bias corrections) must advance in device memory, not in a kernel argument.
__global__ __launch_bounds__(RED_THREADS) void adam_dev_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                               float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                               const float* __restrict__ state, float b1, float b2, float eps,
                                                               float grad_scale, const float* __restrict__ scaler) {
    if (scaler != nullptr) {
        if (scaler[2] != 0.f) return;  // non-finite gradient somewhere: skip the whole update (uniform over the grid)
        grad_scale /= scaler[0];
    }
    const float lr = state[0];
    const float t = state[1] + 1.0f;
    const float step_size = lr / (1.0f - powf(b1, t));
    const float rsqrt_bc2 = rsqrtf(1.0f - powf(b2, t));
    const int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        float4 pp = reinterpret_cast<float4*>(p)[i], gg = reinterpret_cast<const float4*>(g)[i];
        float4 mm = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
        float* P = &pp.x;
        float* G = &gg.x;
        float* M = &mm.x;
        float* V = &vv.x;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float gr = G[j] * grad_scale;
            M[j] = b1 * M[j] + (1.f - b1) * gr;
            V[j] = b2 * V[j] + (1.f - b2) * gr * gr;
            P[j] -= step_size * M[j] / (sqrtf(V[j]) * rsqrt_bc2 + eps);
        }
        reinterpret_cast<float4*>(p)[i] = pp;
        reinterpret_cast<float4*>(m)[i] = mm;
        reinterpret_cast<float4*>(v)[i] = vv;
    }
}
// The same update over a list of element ranges of the flat buffer (the parameters falnet_adam_pack_batched does not own: biases, the two
// factors of the composed logits weights): ranges[2 r] = first element, ranges[2 r + 1] = count; blockIdx.y = range.
__global__ __launch_bounds__(RED_THREADS) void adam_ranges_kernel(float* __restrict__ p, int64_t g_off, int64_t m_off, int64_t v_off,
                                                                  const int64_t* __restrict__ ranges, const float* __restrict__ state, float b1, float b2,
                                                                  float eps, float grad_scale, const float* __restrict__ scaler) {
    if (scaler != nullptr) {
        if (scaler[2] != 0.f) return;
        grad_scale /= scaler[0];
    }
    const float t = state[1] + 1.0f;
    const float step_size = state[0] / (1.0f - powf(b1, t));
    const float rsqrt_bc2 = rsqrtf(1.0f - powf(b2, t));
    const int64_t beg = ranges[2 * blockIdx.y], cnt = ranges[2 * blockIdx.y + 1];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += (int64_t)gridDim.x * blockDim.x) {
        float* wp = p + beg + i;
        const float gr = wp[g_off] * grad_scale;
        const float m = b1 * wp[m_off] + (1.f - b1) * gr;
        const float v = b2 * wp[v_off] + (1.f - b2) * gr * gr;
        *wp -= step_size * m / (sqrtf(v) * rsqrt_bc2 + eps);
        wp[m_off] = m;
        wp[v_off] = v;
    }
}
__global__ void adam_tick_kernel(float* state, const float* __restrict__ scaler) {
    if (scaler == nullptr || scaler[2] == 0.f) state[1] += 1.0f;  // a skipped (overflowed) step does not count
}

// ------------------------------------------------------------------ dynamic loss scale of the f16 path (GradScaler semantics, device-resident)
// scaler = {scale, good_steps, overflow_flag, skipped_steps}.  The guard raises the flag when any element of the (all-reduced)
// flat gradient is inf / NaN; the guarded Adam then leaves p, m, v and the step count untouched; the update kernel halves the scale
// (or doubles it after `interval` clean steps) and clears the flag.  No host synchronisation anywhere.
__global__ __launch_bounds__(RED_THREADS) void grad_guard_kernel(const float* __restrict__ g, int64_t n4, float* __restrict__ scaler) {
    bool bad = false;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 v = reinterpret_cast<const float4*>(g)[i];
        // (x - x) is 0 for every finite x and NaN for inf / NaN
        const float t = (v.x - v.x) + (v.y - v.y) + (v.z - v.z) + (v.w - v.w);
        bad |= !(t == 0.f);
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) scaler[2] = 1.f;  // benign race: every writer stores the same value
}
__global__ void loss_seeds_kernel(const float* __restrict__ scaler, const float* __restrict__ coef, float* __restrict__ seeds, int n) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) seeds[i] = scaler[0] * coef[i];
}
__global__ void loss_scale_update_kernel(float* scaler, float growth, float backoff, float interval, float min_scale, float max_scale) {
    if (scaler[2] != 0.f) {
        // scaler[1] < 0 marks an overflow at a scale that was ALREADY at its floor (LossScaler.check: the run is diverging); a back-off that
        // merely arrives at the floor is not one -- no step has been tried there yet
        const bool at_floor = scaler[0] <= min_scale;
        scaler[0] = fmaxf(scaler[0] * backoff, min_scale);
        scaler[1] = at_floor ? -1.f : 0.f;
        scaler[3] += 1.f;
    } else {
        scaler[1] = fmaxf(scaler[1], 0.f) + 1.f;
        if (scaler[1] >= interval) {
            scaler[0] = fminf(scaler[0] * growth, max_scale);
            scaler[1] = 0.f;
        }
    }
    scaler[2] = 0.f;
}

// ------------------------------------------------------------------ flip / per-sample max (Train_Stage2_K.py:248-253,319)
__global__ __launch_bounds__(RED_THREADS) void hflip_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                            int64_t total, int W) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % W);
        dst[i] = src[i - x + (W - 1 - x)];
    }
}

// One 1024-thread block per sample (no scratch, no second launch): 16-byte loads, four in flight per thread -- the first version walked the
// sample with 256 scalar lanes (206 us for 8 x 131 072 floats, 2.5 % of a Stage-2 step; now ~10 us).
__global__ __launch_bounds__(1024) void rowmax_kernel(const float* __restrict__ src, float* __restrict__ out, int64_t n) {
    __shared__ float red[16];
    const float* s = src + (int64_t)blockIdx.x * n;
    float m = -INFINITY;
    if ((n & 3) == 0 && (reinterpret_cast<uintptr_t>(s) & 15) == 0) {
        const float4* s4 = reinterpret_cast<const float4*>(s);
        const int64_t n4 = n >> 2;
        int64_t i = threadIdx.x;
        for (; i + 3 * 1024 < n4; i += 4 * 1024) {
            const float4 a = s4[i], b = s4[i + 1024], c = s4[i + 2048], d = s4[i + 3072];
            m = fmaxf(m, fmaxf(fmaxf(fmaxf(a.x, a.y), fmaxf(a.z, a.w)), fmaxf(fmaxf(b.x, b.y), fmaxf(b.z, b.w))));
            m = fmaxf(m, fmaxf(fmaxf(fmaxf(c.x, c.y), fmaxf(c.z, c.w)), fmaxf(fmaxf(d.x, d.y), fmaxf(d.z, d.w))));
        }
        for (; i < n4; i += 1024) {
            const float4 a = s4[i];
            m = fmaxf(m, fmaxf(fmaxf(a.x, a.y), fmaxf(a.z, a.w)));
        }
    } else {
        for (int64_t i = threadIdx.x; i < n; i += blockDim.x) m = fmaxf(m, s[i]);
    }
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        float r = red[0];
        for (int i = 1; i < (int)(blockDim.x >> 6); ++i) r = fmaxf(r, red[i]);
        out[blockIdx.x] = r;
    }
}

// ------------------------------------------------------------------ C-ABI
extern "C" int falnet_l1_fwd(const float* a, const float* b, const float* mask, int B, int C, int64_t HW, float scale,
                             float* out, int accumulate, void* stream) {
    FALNET_ENTER(stream);
    FALNET_DET_SLOT(det_slot, stream);
    FALNET_CHECK_ARG(a && b && out && B > 0 && C > 0 && HW > 0, "l1_fwd: bad argument");
    if (int r = zero_scalar_if(out, accumulate, (hipStream_t)stream)) return r;
    const int64_t total = (int64_t)B * C * HW;
    hipLaunchKernelGGL(l1_fwd_kernel, dim3(red_grid(total)), dim3(RED_THREADS), 0, (hipStream_t)stream, a, b, mask, C, HW,
                       total, scale, out, (const float*)nullptr, (float*)nullptr, det_slot);
    FALNET_RETURN_LAUNCH();
}

extern "C" int falnet_l1_bwd(const float* a, const float* b, const float* mask, int B, int C, int64_t HW, float scale,
                             const float* gscale, float* ga, int accumulate, void* stream) {
    FALNET_ENTER(stream);
    FALNET_CHECK_ARG(a && b && ga && B > 0 && C > 0 && HW > 0, "l1_bwd: bad argument");
    const int64_t total = (int64_t)B * C * HW;
    hipLaunchKernelGGL(l1_bwd_kernel, dim3(red_grid(total) * 4), dim3(RED_THREADS), 0, (hipStream_t)stream, a, b, mask, C,
                       HW, total, scale, gscale, ga, accumulate);
    FALNET_RETURN_LAUNCH();
}

extern "C" int falnet_mse_fwd(const void* a, const void* b, int64_t npix, int Cpad, float scale, float* out,
                              int accumulate, int dtype, void* stream) {
    FALNET_ENTER(stream);
    FALNET_DET_SLOT(det_slot, stream);
    FALNET_CHECK_ARG(a && b && out && npix > 0 && Cpad > 0, "mse_fwd: bad argument");
    if (int r = zero_scalar_if(out, accumulate, (hipStream_t)stream)) return r;
    const int64_t total = npix * Cpad;
    const bool vec = (total & 7) == 0 && ((((uintptr_t)a | (uintptr_t)b) & 31) == 0);
    const int grid = red_grid(vec ? total / 8 : total);
#define MSE_FWD(T, V) hipLaunchKernelGGL(HIP_KERNEL_NAME(mse_fwd_kernel<T, V>), dim3(grid), dim3(RED_THREADS), 0, (hipStream_t)stream, (const T*)a, (const T*)b, total, scale, out, (const float*)nullptr, 0.f, (T*)nullptr, det_slot)
#define MSE_FWD_T(T) if (vec) MSE_FWD(T, true); else MSE_FWD(T, false)
    FALNET_DISPATCH_DTYPE(dtype, MSE_FWD_T);
#undef MSE_FWD_T
#undef MSE_FWD
    FALNET_RETURN_LAUNCH();
}

extern "C" int falnet_mse_bwd(const void* a, const void* b, int64_t npix, int Cpad, float scale, const float* gscale,
                              void* ga, int dtype, void* stream) {
    FALNET_ENTER(stream);
    FALNET_CHECK_ARG(a && b && ga && npix > 0 && Cpad > 0, "mse_bwd: bad argument");
    const int64_t total = npix * Cpad;
    const bool vec = (total & 7) == 0 && ((((uintptr_t)a | (uintptr_t)b | (uintptr_t)ga) & 31) == 0);
    const int grid = red_grid(vec ? total / 8 : total) * 4;
#define MSE_BWD(T, V) hipLaunchKernelGGL(HIP_KERNEL_NAME(mse_bwd_kernel<T, V>), dim3(grid), dim3(RED_THREADS), 0, (hipStream_t)stream, (const T*)a, (const T*)b, total, scale, gscale, (T*)ga)
#define MSE_BWD_T(T) if (vec) MSE_BWD(T, true); else MSE_BWD(T, false)
    FALNET_DISPATCH_DTYPE(dtype, MSE_BWD_T);
#undef MSE_BWD_T
#undef MSE_BWD
    FALNET_RETURN_LAUNCH();
}

extern "C" int falnet_smooth_fwd(const float* img, const float* disp, int B, int H, int W, int x0, int x1, float gamma,
                                 float scale, float* out, int accumulate, void* stream) {
    FALNET_ENTER(stream);
    FALNET_DET_SLOT(det_slot, stream);
    FALNET_CHECK_ARG(img && disp && out && B > 0 && H > 0 && 0 <= x0 && x0 < x1 && x1 <= W, "smooth_fwd: bad argument");
    if (int r = zero_scalar_if(out, accumulate, (hipStream_t)stream)) return r;
    SmoothArgs s{img, disp, B, H, W, x0, x1, gamma};
    const int64_t ftiles = (int64_t)B * ((H + SM_TY - 1) / SM_TY) * ((x1 - x0 + SM_TX - 1) / SM_TX);
    hipLaunchKernelGGL(smooth_fwd_kernel, dim3((unsigned)(ftiles < RED_BLOCKS ? ftiles : RED_BLOCKS)), dim3(RED_THREADS), 0,
                       (hipStream_t)stream, s, scale, out, det_slot);
    FALNET_RETURN_LAUNCH();
}

extern "C" int falnet_smooth_bwd(const float* img, const float* disp, int B, int H, int W, int x0, int x1, float gamma,
                                 float scale, const float* gscale, float* gdisp, int accumulate, void* stream) {
    FALNET_ENTER(stream);
    FALNET_CHECK_ARG(img && disp && gdisp && B > 0 && H > 0 && 0 <= x0 && x0 < x1 && x1 <= W, "smooth_bwd: bad argument");
    SmoothArgs s{img, disp, B, H, W, x0, x1, gamma};
    const int64_t btiles = (int64_t)B * ((H + SM_TY - 1) / SM_TY) * ((W + SM_TX - 1) / SM_TX);
    hipLaunchKernelGGL(smooth_bwd_kernel, dim3((unsigned)(btiles < 4 * RED_BLOCKS ? btiles : 4 * RED_BLOCKS)), dim3(RED_THREADS), 0,
                       (hipStream_t)stream, s, scale, gscale, gdisp, accumulate);
    FALNET_RETURN_LAUNCH();
}

extern "C" int falnet_mask_mix(const float* a, const float* b, const float* m, float* out, int B, int C, int64_t HW,
                               void* stream) {
    FALNET_ENTER(stream);
    FALNET_CHECK_ARG(a && b && m && out && B > 0 && C > 0 && HW > 0, "mask_mix: bad argument");
    const int64_t total = (int64_t)B * C * HW;
    hipLaunchKernelGGL(mask_mix_kernel, dim3(red_grid(total) * 4), dim3(RED_THREADS), 0, (hipStream_t)stream, a, b, m, out,
                       C, HW, total);
    FALNET_RETURN_LAUNCH();
}

extern "C" int falnet_occlusion_mask(const float* a, const float* b, float* out, int B, int H, int W, int x0, int x1, void* stream) {
    FALNET_ENTER(stream);
    FALNET_CHECK_ARG(a && b && out && B > 0 && H > 0 && W > 0 && 0 <= x0 && x0 <= x1 && x1 <= W, "occlusion_mask: bad argument");
    const int64_t total = (int64_t)B * H * W;
    hipLaunchKernelGGL(occlusion_mask_kernel, dim3(red_grid(total) * 4), dim3(RED_THREADS), 0, (hipStream_t)stream, a, b, out, W, x0, x1, total);
    FALNET_RETURN_LAUNCH();
}

extern "C" int falnet_mirror_weight(const float* occ, const float* rowmax, float* out, int B, int H, int W, int x0, int x1, void* stream) {
    FALNET_ENTER(stream);
    FALNET_CHECK_ARG(occ && rowmax && out && B > 0 && H > 0 && W > 0 && 0 <= x0 && x0 <= x1 && x1 <= W, "mirror_weight: bad argument");
    const int64_t total = (int64_t)B * H * W;
    hipLaunchKernelGGL(mirror_weight_kernel, dim3(red_grid(total) * 4), dim3(RED_THREADS), 0, (hipStream_t)stream, occ, rowmax, out, (int64_t)H * W, W,
                       x0, x1, total);
    FALNET_RETURN_LAUNCH();
}

extern "C" int falnet_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2,
                                float eps, int step, float grad_scale, void* stream) {
    FALNET_ENTER(stream);
    FALNET_CHECK_ARG(p && g && m && v && n > 0 && step >= 1, "adam_step: bad argument");
    FALNET_CHECK_ARG((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0, "adam_step: buffers must be 16-B aligned");
    const double bc1 = 1.0 - pow((double)b1, step), bc2 = 1.0 - pow((double)b2, step);
    const float step_size = (float)(lr / bc1), rsqrt_bc2 = (float)(1.0 / sqrt(bc2));
    hipLaunchKernelGGL(adam_kernel, dim3(2048), dim3(RED_THREADS), 0, (hipStream_t)stream, p, g, m, v, n, step_size, b1,
                       b2, eps, rsqrt_bc2, grad_scale);
    FALNET_RETURN_LAUNCH();
}

static int adam_dev_launch(float* p, const float* g, float* m, float* v, int64_t n, float* state, float b1, float b2, float eps, float grad_scale,
                           const float* scaler, void* stream) {
    FALNET_ENTER(stream);
    FALNET_CHECK_ARG(p && g && m && v && state && n > 0 && (n & 3) == 0, "adam_step_dev: bad argument (n must be a multiple of 4)");
    FALNET_CHECK_ARG((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0, "adam_step_dev: buffers must be 16-B aligned");
    hipLaunchKernelGGL(adam_dev_kernel, dim3(2048), dim3(RED_THREADS), 0, (hipStream_t)stream, p, g, m, v, n, state, b1, b2, eps, grad_scale, scaler);
    hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, state, scaler);
    FALNET_RETURN_LAUNCH();
}
extern "C" int falnet_adam_ranges(float* p, int64_t g_off, int64_t m_off, int64_t v_off, const int64_t* ranges_dev, int n_ranges, const float* state,
                                  float b1, float b2, float eps, float grad_scale, const float* scaler, void* stream) {
    FALNET_ENTER(stream);
    FALNET_CHECK_ARG(p && ranges_dev && n_ranges > 0 && state && g_off != 0 && m_off != 0 && v_off != 0, "adam_ranges: bad argument");
    hipLaunchKernelGGL(adam_ranges_kernel, dim3(48, n_ranges), dim3(RED_THREADS), 0, (hipStream_t)stream, p, g_off, m_off, v_off, ranges_dev, state, b1, b2, eps,
                       grad_scale, scaler);
    FALNET_RETURN_LAUNCH();
}
extern "C" int falnet_adam_tick(float* state, const float* scaler, void* stream) {
    FALNET_ENTER(stream);
    FALNET_CHECK_ARG(state, "adam_tick: bad argument");
    hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, state, scaler);
    FALNET_RETURN_LAUNCH();
}
extern "C" int falnet_adam_step_dev(float* p, const float* g, float* m, float* v, int64_t n, float* state, float b1, float b2,
                                    float eps, float grad_scale, void* stream) {
    return adam_dev_launch(p, g, m, v, n, state, b1, b2, eps, grad_scale, nullptr, stream);
}
extern "C" int falnet_adam_step_guarded(float* p, const float* g, float* m, float* v, int64_t n, float* state, float b1, float b2,
                                        float eps, float grad_scale, const float* scaler, void* stream) {
    FALNET_CHECK_ARG(scaler, "adam_step_guarded: scaler is NULL");
    return adam_dev_launch(p, g, m, v, n, state, b1, b2, eps, grad_scale, scaler, stream);
}
// Tail of a fused training step: out = {S0 + a * S1, S0, S1}, then S0 = S1 = 0 for the next step (one launch instead of a clone, an
// axpy and a fill).
__global__ void step_scalars_kernel(float* S, float a, float* out) {
    const float s0 = S[0], s1 = S[1];
    out[0] = s0 + a * s1;
    out[1] = s0;
    out[2] = s1;
    S[0] = 0.f;
    S[1] = 0.f;
}
extern "C" int falnet_step_scalars(float* S, float a, float* out, void* stream) {
    FALNET_ENTER(stream);
    FALNET_CHECK_ARG(S && out, "step_scalars: bad argument");
    hipLaunchKernelGGL(step_scalars_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, S, a, out);
    FALNET_RETURN_LAUNCH();
}

extern "C" int falnet_grad_guard(const float* g, int64_t n, float* scaler, void* stream) {
    FALNET_ENTER(stream);
    FALNET_CHECK_ARG(g && scaler && n > 0 && (n & 3) == 0 && ((uintptr_t)g & 15) == 0, "grad_guard: bad argument (n % 4 == 0, 16-B aligned)");
    hipLaunchKernelGGL(grad_guard_kernel, dim3(1024), dim3(RED_THREADS), 0, (hipStream_t)stream, g, n >> 2, scaler);
    FALNET_RETURN_LAUNCH();
}
extern "C" int falnet_loss_seeds(const float* scaler, const float* coef, float* seeds, int n, void* stream) {
    FALNET_ENTER(stream);
    FALNET_CHECK_ARG(scaler && coef && seeds && n > 0, "loss_seeds: bad argument");
    hipLaunchKernelGGL(loss_seeds_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, scaler, coef, seeds, n);
    FALNET_RETURN_LAUNCH();
}
extern "C" int falnet_loss_scale_update(float* scaler, float growth, float backoff, int interval, float min_scale, float max_scale, void* stream) {
    FALNET_ENTER(stream);
    FALNET_CHECK_ARG(scaler && growth >= 1.f && backoff > 0.f && backoff <= 1.f && interval >= 1 && min_scale > 0.f && max_scale >= min_scale,
                     "loss_scale_update: bad argument");
    hipLaunchKernelGGL(loss_scale_update_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, scaler, growth, backoff, (float)interval, min_scale, max_scale);
    FALNET_RETURN_LAUNCH();
}

extern "C" int falnet_hflip(const float* src, float* dst, int64_t n_rows, int W, void* stream) {
    FALNET_ENTER(stream);
    FALNET_CHECK_ARG(src && dst && src != dst && n_rows > 0 && W > 0, "hflip: bad argument (in-place not supported)");
    const int64_t total = n_rows * W;
    hipLaunchKernelGGL(hflip_kernel, dim3(red_grid(total) * 4), dim3(RED_THREADS), 0, (hipStream_t)stream, src, dst, total, W);
    FALNET_RETURN_LAUNCH();
}

extern "C" int falnet_rowmax(const float* src, float* out, int B, int64_t n, void* stream) {
    FALNET_ENTER(stream);
    FALNET_CHECK_ARG(src && out && B > 0 && n > 0, "rowmax: bad argument");
    hipLaunchKernelGGL(rowmax_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, src, out, n);
    FALNET_RETURN_LAUNCH();
}

// ---- loss and gradient in ONE pass (the fused training step: upstream scalars are known before the forward runs) ----------------
extern "C" int falnet_l1_fwd_bwd(const float* a, const float* b, int B, int C, int64_t HW, float scale, float* out, const float* gscale,
                                 float* ga, void* stream) {
    FALNET_ENTER(stream);
    FALNET_DET_SLOT(det_slot, stream);
    FALNET_CHECK_ARG(a && b && out && ga && B > 0 && C > 0 && HW > 0, "l1_fwd_bwd: bad argument");
    const int64_t total = (int64_t)B * C * HW;
    hipLaunchKernelGGL(l1_fwd_kernel, dim3(red_grid(total)), dim3(RED_THREADS), 0, (hipStream_t)stream, a, b, (const float*)nullptr, C, HW, total, scale,
                       out, gscale, ga, det_slot, (const float*)nullptr);
    FALNET_RETURN_LAUNCH();
}

extern "C" int falnet_l1_fwd_bwd_add(const float* a, const float* b, int B, int C, int64_t HW, float scale, float* out, const float* gscale,
                                     const float* gadd, float* ga, void* stream) {
    FALNET_ENTER(stream);
    FALNET_DET_SLOT(det_slot, stream);
    FALNET_CHECK_ARG(a && b && out && ga && gadd && B > 0 && C > 0 && HW > 0, "l1_fwd_bwd_add: bad argument");
    const int64_t total = (int64_t)B * C * HW;
    hipLaunchKernelGGL(l1_fwd_kernel, dim3(red_grid(total)), dim3(RED_THREADS), 0, (hipStream_t)stream, a, b, (const float*)nullptr, C, HW, total, scale,
                       out, gscale, ga, det_slot, gadd);
    FALNET_RETURN_LAUNCH();
}

extern "C" int falnet_mse_fwd_bwd(const void* a, const void* b, int64_t npix, int Cpad, float scale_out, float* out, float scale_grad,
                                  const float* gscale, void* ga, int dtype, void* stream) {
    FALNET_ENTER(stream);
    FALNET_DET_SLOT(det_slot, stream);
    FALNET_CHECK_ARG(a && b && out && ga && npix > 0 && Cpad > 0, "mse_fwd_bwd: bad argument");
    const int64_t total = npix * Cpad;
    const bool vec = (total & 7) == 0 && ((((uintptr_t)a | (uintptr_t)b | (uintptr_t)ga) & 31) == 0);
    const int grid = red_grid(vec ? total / 8 : total);
#define MSE_FB(T, V) hipLaunchKernelGGL(HIP_KERNEL_NAME(mse_fwd_kernel<T, V>), dim3(grid), dim3(RED_THREADS), 0, (hipStream_t)stream, (const T*)a, (const T*)b, total, scale_out, out, gscale, scale_grad, (T*)ga, det_slot)
#define MSE_FB_T(T) if (vec) MSE_FB(T, true); else MSE_FB(T, false)
    FALNET_DISPATCH_DTYPE(dtype, MSE_FB_T);
#undef MSE_FB_T
#undef MSE_FB
    FALNET_RETURN_LAUNCH();
}

extern "C" int falnet_mse3_fwd_bwd(const void* const* a, const void* const* b, const int64_t* numel, const float* scale_out, float* out,
                                   const float* scale_grad, const float* gscale, void* const* ga, int dtype, void* stream) {
    FALNET_ENTER(stream);
    FALNET_DET_SLOT(det_slot, stream);
    FALNET_CHECK_ARG(a && b && numel && scale_out && out && scale_grad && ga, "mse3_fwd_bwd: bad argument");
    Mse3Args m;
    int64_t total = 0;
    for (int k = 0; k < 3; ++k) {
        FALNET_CHECK_ARG(a[k] && b[k] && ga[k] && numel[k] > 0 && (numel[k] & 7) == 0 && ((((uintptr_t)a[k] | (uintptr_t)b[k] | (uintptr_t)ga[k]) & 31) == 0),
                         "mse3_fwd_bwd: tensor %d must be 32-B aligned with a multiple of 8 elements", k);
        m.a[k] = a[k]; m.b[k] = b[k]; m.ga[k] = ga[k];
        m.n8[k] = numel[k] >> 3;
        m.scale_out[k] = scale_out[k]; m.scale_grad[k] = scale_grad[k];
        total += numel[k];
    }
    // RED_BLOCKS workgroups shared out by size, at least 16 each (one atomic / one deterministic slot entry per workgroup as in the single form)
    int used = 0;
    for (int k = 0; k < 3; ++k) {
        m.begin[k] = used;
        int nb = k == 2 ? RED_BLOCKS - used : (int)((double)numel[k] / (double)total * RED_BLOCKS);
        if (nb < 16) nb = 16;
        if (used + nb > RED_BLOCKS - 16 * (2 - k)) nb = RED_BLOCKS - 16 * (2 - k) - used;
        used += nb;
    }
    m.begin[3] = used;
#define MSE3_L(T) hipLaunchKernelGGL(HIP_KERNEL_NAME(mse3_fwd_bwd_kernel<T>), dim3(used), dim3(RED_THREADS), 0, (hipStream_t)stream, m, out, gscale, det_slot)
    FALNET_DISPATCH_DTYPE(dtype, MSE3_L);
#undef MSE3_L
    FALNET_RETURN_LAUNCH();
}

extern "C" int falnet_smooth_fwd_bwd(const float* img, const float* disp, int B, int H, int W, int x0, int x1, float gamma, float scale,
                                     float* out, const float* gscale, float* gdisp, void* stream) {
    FALNET_ENTER(stream);
    FALNET_DET_SLOT(det_slot, stream);
    FALNET_CHECK_ARG(img && disp && out && gdisp && B > 0 && H > 0 && 0 <= x0 && x0 < x1 && x1 <= W, "smooth_fwd_bwd: bad argument");
    SmoothArgs s{img, disp, B, H, W, x0, x1, gamma};
    const int64_t btiles = (int64_t)B * ((H + SM_TY - 1) / SM_TY) * ((W + SM_TX - 1) / SM_TX);
    hipLaunchKernelGGL(smooth_bwd_kernel, dim3((unsigned)(btiles < RED_BLOCKS ? btiles : RED_BLOCKS)), dim3(RED_THREADS), 0, (hipStream_t)stream, s,
                       scale, gscale, gdisp, 0, out, det_slot);
    FALNET_RETURN_LAUNCH();
}
