// Fused MED head for gfx950: softmax over the N disparity planes, expectation, plane-sweep warp,
// second softmax and view blend in ONE pass over the logits (reference: models/FAL_netB.py:216-282,
// ~1500 aten launches and an O(N^2) torch.cat there).
//
// Layout: everything planar f32 (the warp is a shift along the contiguous W axis).  One workgroup
// owns one image row (b, y): the row of the left image (and, in backward, the per-pixel softmax
// statistics and upstream gradients) is staged in LDS once and reused by all N planes; each lane owns
// pixels x and walks the planes with a chunked online softmax (one rescale per 8 planes), so no
// cross-lane reduction is needed and the only HBM traffic is the N logit rows, read once.
// HBM-bound: algorithmic bytes (N+7)*HW*4 forward, (2N+7)*HW*4 backward per sample.
#include <math.h>
#include "common.h"

#define HEAD_THREADS 256
#define HEAD_MAXN 128
#define CH 8  // planes per online-softmax chunk

struct PlaneTab {  // per-sample plane table in LDS
    float d[HEAD_MAXN];  // disparity of plane n in pixels (FAL_netB.py:223-225)
    float a[HEAD_MAXN];  // fractional part of the shift s_n = d_n (W-1)/W
    int k[HEAD_MAXN];    // integer part
};

__device__ __forceinline__ void build_plane_tab(PlaneTab& t, float mn, float mx, int N, int W) {
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
        const float c = (float)n / (float)(N - 1);
        const float d = mx * expf(logf(mx / mn) * (c - 1.0f));
        const float s = d * (float)(W - 1) / (float)W;  // 2d/W normalised * (W-1)/2 (align_corners=True)
        const float kf = floorf(s);
        t.d[n] = d;
        t.k[n] = (int)kf;
        t.a[n] = s - kf;
    }
}

// ---------------------------------------------------------------------------------------- forward
// LDS: PlaneTab | left[3][W+2] (two trailing zeros: out-of-range taps read 0 without a branch)
__global__ __launch_bounds__(HEAD_THREADS) void med_head_fwd_kernel(
    const float* __restrict__ dlog0, const float* __restrict__ left, const float* __restrict__ min_disp,
    const float* __restrict__ max_disp, float* __restrict__ disp, float* __restrict__ p_im0,
    float* __restrict__ stats, int N, int H, int W) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    PlaneTab& tab = *reinterpret_cast<PlaneTab*>(smem);
    float* lrow = reinterpret_cast<float*>(smem + sizeof(PlaneTab));
    const int b = blockIdx.x / H, y = blockIdx.x % H;
    const int64_t HW = (int64_t)H * W;
    const int WP = W + 2;
    build_plane_tab(tab, min_disp[b], max_disp[b], N, W);
    const bool want_pan = p_im0 != nullptr;
    if (want_pan) {
        for (int i = threadIdx.x; i < 3 * WP; i += blockDim.x) {
            const int c = i / WP, x = i % WP;
            lrow[i] = x < W ? left[((int64_t)b * 3 + c) * HW + (int64_t)y * W + x] : 0.f;
        }
    }
    __syncthreads();
    const float* Lrow = dlog0 + (int64_t)b * N * HW + (int64_t)y * W;

    for (int x = threadIdx.x; x < W; x += blockDim.x) {
        float m0 = -INFINITY, z0 = 0.f, dacc = 0.f;
        float mw = -INFINITY, zw = 0.f, p0 = 0.f, p1 = 0.f, p2 = 0.f;
        for (int n0 = 0; n0 < N; n0 += CH) {
            float l0[CH], lw[CH];
            float cm0 = -INFINITY, cmw = -INFINITY;
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                const int n = n0 + j;
                if (n < N) {
                    const float* Ln = Lrow + (int64_t)n * HW;
                    l0[j] = Ln[x];
                    const int i0 = x + tab.k[n];
                    const float a = tab.a[n];
                    const float t0 = i0 < W ? Ln[i0] : 0.f;           // zero padding: OOB logit is 0, not -inf
                    const float t1 = i0 + 1 < W ? Ln[i0 + 1] : 0.f;   // (grid_sample padding_mode='zeros')
                    lw[j] = (1.f - a) * t0 + a * t1;
                } else {
                    l0[j] = -INFINITY;
                    lw[j] = -INFINITY;
                }
                cm0 = fmaxf(cm0, l0[j]);
                cmw = fmaxf(cmw, lw[j]);
            }
            if (cm0 > m0) {
                const float s = __expf(m0 - cm0);
                z0 *= s;
                dacc *= s;
                m0 = cm0;
            }
            if (cmw > mw) {
                const float s = __expf(mw - cmw);
                zw *= s;
                p0 *= s;
                p1 *= s;
                p2 *= s;
                mw = cmw;
            }
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                const int n = n0 + j;
                if (n < N) {
                    const float e = __expf(l0[j] - m0);
                    z0 += e;
                    dacc += tab.d[n] * e;
                    const float ew = __expf(lw[j] - mw);
                    zw += ew;
                    if (want_pan) {
                        const int i0 = min(x + tab.k[n], W);
                        const float a = tab.a[n];
                        p0 += ew * ((1.f - a) * lrow[i0] + a * lrow[i0 + 1]);
                        p1 += ew * ((1.f - a) * lrow[WP + i0] + a * lrow[WP + i0 + 1]);
                        p2 += ew * ((1.f - a) * lrow[2 * WP + i0] + a * lrow[2 * WP + i0 + 1]);
                    }
                }
            }
        }
        const int64_t pix = (int64_t)y * W + x;
        if (disp) disp[(int64_t)b * HW + pix] = dacc / z0;
        if (want_pan) {
            const float r = 1.f / zw;
            p_im0[((int64_t)b * 3 + 0) * HW + pix] = p0 * r;
            p_im0[((int64_t)b * 3 + 1) * HW + pix] = p1 * r;
            p_im0[((int64_t)b * 3 + 2) * HW + pix] = p2 * r;
        }
        if (stats) {
            float* st = stats + (int64_t)b * 4 * HW + pix;
            st[0] = m0;
            st[HW] = z0;
            st[2 * HW] = mw;
            st[3 * HW] = zw;
        }
    }
}

// Forward, LDS-staged form (used when the row fits: W <= 2048).  The first version read every logit three times
// per pixel with 4-byte lane loads (x, x+k, x+k+1): TA-bound at ~1.9 TB/s.  Here the workgroup streams CH plane rows of
// its image row into LDS with coalesced 16-byte loads (each logit leaves HBM/L2 exactly once) and the three taps
// become LDS reads; per-pixel softmax state lives in registers across chunks (PPT pixels per thread).
template <int PPT>
__global__ __launch_bounds__(HEAD_THREADS) void med_head_fwd_lds_kernel(
    const float* __restrict__ dlog0, const float* __restrict__ left, const float* __restrict__ min_disp,
    const float* __restrict__ max_disp, float* __restrict__ disp, float* __restrict__ p_im0,
    float* __restrict__ stats, int N, int H, int W) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    PlaneTab& tab = *reinterpret_cast<PlaneTab*>(smem);
    const int WP = (W + 7) & ~3;                                      // row pitch in floats: >= W+4, multiple of 4 (zero tail)
    float* lrow = reinterpret_cast<float*>(smem + sizeof(PlaneTab));  // [3][WP]
    float* prow = lrow + 3 * WP;                                      // [CH][WP]
    const int b = blockIdx.x / H, y = blockIdx.x % H;
    const int64_t HW = (int64_t)H * W;
    build_plane_tab(tab, min_disp[b], max_disp[b], N, W);
    const bool want_pan = p_im0 != nullptr;
    for (int i = threadIdx.x; i < 3 * WP; i += blockDim.x) {
        const int c = i / WP, x = i % WP;
        lrow[i] = (want_pan && x < W) ? left[((int64_t)b * 3 + c) * HW + (int64_t)y * W + x] : 0.f;
    }
    for (int i = threadIdx.x; i < CH * WP; i += blockDim.x)
        if (i % WP >= W) prow[i] = 0.f;                               // zero tail of every plane row (out-of-range taps read 0)
    const float* Lrow = dlog0 + (int64_t)b * N * HW + (int64_t)y * W;
    const bool vec4 = (W & 3) == 0 && ((reinterpret_cast<uintptr_t>(Lrow) & 15) == 0) && ((HW & 3) == 0);

    float m0[PPT], z0[PPT], dacc[PPT], mw[PPT], zw[PPT], p0[PPT], p1[PPT], p2[PPT];
#pragma unroll
    for (int q = 0; q < PPT; ++q) {
        m0[q] = mw[q] = -INFINITY;
        z0[q] = dacc[q] = zw[q] = p0[q] = p1[q] = p2[q] = 0.f;
    }
    // register prefetch of the next chunk's plane rows (vec4 path): the global round trip of chunk n0+CH overlaps the
    // softmax arithmetic of chunk n0 instead of sitting between two barriers
    constexpr int PF = (CH * PPT + 3) / 4;  // float4 per thread: CH rows of W <= 256*PPT floats over 256 threads
    float4 pf[PF];
    const int w4 = W >> 2;
    auto fetch = [&](int n0) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int i = threadIdx.x + u * HEAD_THREADS;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < CH * w4) {
                const int j = i / w4, q = i % w4;
                if (n0 + j < N) v = reinterpret_cast<const float4*>(Lrow + (int64_t)(n0 + j) * HW)[q];
            }
            pf[u] = v;
        }
    };
    if (vec4) fetch(0);
    for (int n0 = 0; n0 < N; n0 += CH) {
        __syncthreads();                                              // previous chunk fully consumed (and tab/lrow ready)
        if (vec4) {
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const int i = threadIdx.x + u * HEAD_THREADS;
                if (i < CH * w4) *reinterpret_cast<float4*>(prow + (i / w4) * WP + 4 * (i % w4)) = pf[u];
            }
        } else {
            for (int i = threadIdx.x; i < CH * W; i += blockDim.x) {
                const int j = i / W, x = i % W;
                prow[j * WP + x] = n0 + j < N ? Lrow[(int64_t)(n0 + j) * HW + x] : 0.f;
            }
        }
        __syncthreads();
        if (vec4 && n0 + CH < N) fetch(n0 + CH);
#pragma unroll
        for (int q = 0; q < PPT; ++q) {
            const int x = threadIdx.x + q * HEAD_THREADS;
            if (x >= W) continue;
            float l0[CH], lw[CH];
            float cm0 = -INFINITY, cmw = -INFINITY;
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                const int n = n0 + j;
                if (n < N) {
                    const float* pr = prow + j * WP;
                    l0[j] = pr[x];
                    const int i0 = min(x + tab.k[n], W);
                    const float a = tab.a[n];
                    lw[j] = (1.f - a) * pr[i0] + a * pr[i0 + 1];      // zero tail: OOB logit is 0, not -inf
                } else {
                    l0[j] = -INFINITY;
                    lw[j] = -INFINITY;
                }
                cm0 = fmaxf(cm0, l0[j]);
                cmw = fmaxf(cmw, lw[j]);
            }
            if (cm0 > m0[q]) {
                const float s = __expf(m0[q] - cm0);
                z0[q] *= s;
                dacc[q] *= s;
                m0[q] = cm0;
            }
            if (cmw > mw[q]) {
                const float s = __expf(mw[q] - cmw);
                zw[q] *= s;
                p0[q] *= s;
                p1[q] *= s;
                p2[q] *= s;
                mw[q] = cmw;
            }
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                const int n = n0 + j;
                if (n < N) {
                    const float e = __expf(l0[j] - m0[q]);
                    z0[q] += e;
                    dacc[q] += tab.d[n] * e;
                    const float ew = __expf(lw[j] - mw[q]);
                    zw[q] += ew;
                    if (want_pan) {
                        const int i0 = min(x + tab.k[n], W);
                        const float a = tab.a[n];
                        p0[q] += ew * ((1.f - a) * lrow[i0] + a * lrow[i0 + 1]);
                        p1[q] += ew * ((1.f - a) * lrow[WP + i0] + a * lrow[WP + i0 + 1]);
                        p2[q] += ew * ((1.f - a) * lrow[2 * WP + i0] + a * lrow[2 * WP + i0 + 1]);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int q = 0; q < PPT; ++q) {
        const int x = threadIdx.x + q * HEAD_THREADS;
        if (x >= W) continue;
        const int64_t pix = (int64_t)y * W + x;
        if (disp) disp[(int64_t)b * HW + pix] = dacc[q] / z0[q];
        if (want_pan) {
            const float r = 1.f / zw[q];
            p_im0[((int64_t)b * 3 + 0) * HW + pix] = p0[q] * r;
            p_im0[((int64_t)b * 3 + 1) * HW + pix] = p1[q] * r;
            p_im0[((int64_t)b * 3 + 2) * HW + pix] = p2[q] * r;
        }
        if (stats) {
            float* st = stats + (int64_t)b * 4 * HW + pix;
            st[0] = m0[q];
            st[HW] = z0[q];
            st[2 * HW] = mw[q];
            st[3 * HW] = zw[q];
        }
    }
}

// ---------------------------------------------------------------------------------------- backward
__device__ __forceinline__ void store8(float* p, const float (&v)[8]) {
    reinterpret_cast<float4*>(p)[0] = make_float4(v[0], v[1], v[2], v[3]);
    reinterpret_cast<float4*>(p)[1] = make_float4(v[4], v[5], v[6], v[7]);
}
template <typename T>  // 16-bit operand types
__device__ __forceinline__ void store8(T* p, const float (&v)[8]) {
    uint4 q;
#pragma unroll
    for (int i = 0; i < 4; ++i) (&q.x)[i] = pack16x2<T>(v[2 * i], v[2 * i + 1]);
    *reinterpret_cast<uint4*>(p) = q;
}
// grad_dlog0[n, x'] = (1-a) gwl_n(x'-k) + a gwl_n(x'-k-1) + gd(x') sm_n(x') (d_n - disp(x'))
//   gwl_n(x) = Dprob_n(x) (sum_c gp_c(x) S_{c,n}(x) - q(x)),  q = sum_c gp_c p_c
// Each (n, x') is owned by exactly one lane (gather form of the transposed warp): no atomics.
// LDS rows (index x+1, one zero column in front and two behind):
//   rowU[3] = gp_c / Zw, rowV = q / Zw, rowM = Mw   (at the *source* pixel x)
// OUT = float: planar f32 [B][N][H][W] (the reference layout).  OUT = bf16_t / NHWC = true: pixel-major
// [B][H][W][cpad] in the conv stack's compute dtype, channels >= N zero -- what the 1x1 logits conv's data / weight
// gradient launches read, so no planar round trip + layout conversion launch in between.
template <typename OUT, bool NHWC>
__global__ __launch_bounds__(HEAD_THREADS) void med_head_bwd_kernel(
    const float* __restrict__ dlog0, const float* __restrict__ left, const float* __restrict__ min_disp,
    const float* __restrict__ max_disp, const float* __restrict__ disp, const float* __restrict__ p_im0,
    const float* __restrict__ stats, const float* __restrict__ gdisp, const float* __restrict__ gpan,
    OUT* __restrict__ gdlog0, int N, int H, int W, int cpad) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    PlaneTab& tab = *reinterpret_cast<PlaneTab*>(smem);
    const int WP = W + 3;
    float* rowU = reinterpret_cast<float*>(smem + sizeof(PlaneTab));  // [3][WP]
    float* rowV = rowU + 3 * WP;
    float* rowM = rowV + WP;
    const int b = blockIdx.x / H, y = blockIdx.x % H;
    const int64_t HW = (int64_t)H * W;
    const int64_t rowoff = (int64_t)y * W;
    build_plane_tab(tab, min_disp[b], max_disp[b], N, W);
    const bool has_pan = gpan != nullptr, has_disp = gdisp != nullptr;
    const float* st = stats + (int64_t)b * 4 * HW + rowoff;
    if (has_pan) {
        for (int i = threadIdx.x; i < WP; i += blockDim.x) {
            const int x = i - 1;
            float u0 = 0.f, u1 = 0.f, u2 = 0.f, v = 0.f, m = 0.f;
            if (x >= 0 && x < W) {
                const float rz = 1.f / st[3 * HW + x];
                const float g0 = gpan[((int64_t)b * 3 + 0) * HW + rowoff + x];
                const float g1 = gpan[((int64_t)b * 3 + 1) * HW + rowoff + x];
                const float g2 = gpan[((int64_t)b * 3 + 2) * HW + rowoff + x];
                const float q = g0 * p_im0[((int64_t)b * 3 + 0) * HW + rowoff + x] +
                                g1 * p_im0[((int64_t)b * 3 + 1) * HW + rowoff + x] +
                                g2 * p_im0[((int64_t)b * 3 + 2) * HW + rowoff + x];
                u0 = g0 * rz;
                u1 = g1 * rz;
                u2 = g2 * rz;
                v = q * rz;
                m = st[2 * HW + x];
            }
            rowU[i] = u0;
            rowU[WP + i] = u1;
            rowU[2 * WP + i] = u2;
            rowV[i] = v;
            rowM[i] = m;
        }
    }
    __syncthreads();
    const float* Lrow = dlog0 + (int64_t)b * N * HW + rowoff;
    OUT* Grow = NHWC ? gdlog0 + ((int64_t)b * HW + rowoff) * cpad : gdlog0 + (int64_t)b * N * HW + rowoff;

    for (int x = threadIdx.x; x < W; x += blockDim.x) {
        // per-pixel constants across planes
        float lm[3], lc[3], lp[3];
        if (has_pan) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float* lr = left + ((int64_t)b * 3 + c) * HW + rowoff;
                lm[c] = x > 0 ? lr[x - 1] : 0.f;
                lc[c] = lr[x];
                lp[c] = x + 1 < W ? lr[x + 1] : 0.f;
            }
        }
        float m0 = 0.f, rz0 = 0.f, gd = 0.f, dsp = 0.f;
        if (has_disp) {
            m0 = st[x];
            rz0 = 1.f / st[HW + x];
            gd = gdisp[(int64_t)b * HW + rowoff + x];
            dsp = disp[(int64_t)b * HW + rowoff + x];
        }
        float gv[8];
        const int nend = NHWC ? cpad : N;
        for (int n = 0; n < nend; ++n) {
            if (NHWC && n >= N) {  // zero padding channels
                gv[n & 7] = 0.f;
                if ((n & 7) == 7) store8(Grow + (int64_t)x * cpad + (n - 7), gv);
                continue;
            }
            const float* Ln = Lrow + (int64_t)n * HW;
            const float Lc = Ln[x];
            float g = 0.f;
            if (has_pan) {
                const int k = tab.k[n];
                const float a = tab.a[n];
                const float Lm = x > 0 ? Ln[x - 1] : 0.f;
                const float Lp = x + 1 < W ? Ln[x + 1] : 0.f;
                const int xa = x - k;  // source pixel whose first tap is x
                if (xa >= 0) {         // xa <= W-1 always (k >= 0)
                    const float wl = (1.f - a) * Lc + a * Lp;
                    const float G = rowU[xa + 1] * ((1.f - a) * lc[0] + a * lp[0]) +
                                    rowU[WP + xa + 1] * ((1.f - a) * lc[1] + a * lp[1]) +
                                    rowU[2 * WP + xa + 1] * ((1.f - a) * lc[2] + a * lp[2]);
                    g += (1.f - a) * __expf(wl - rowM[xa + 1]) * (G - rowV[xa + 1]);
                }
                const int xb = xa - 1;  // source pixel whose second tap is x
                if (xb >= 0) {
                    const float wl = (1.f - a) * Lm + a * Lc;
                    const float G = rowU[xb + 1] * ((1.f - a) * lm[0] + a * lc[0]) +
                                    rowU[WP + xb + 1] * ((1.f - a) * lm[1] + a * lc[1]) +
                                    rowU[2 * WP + xb + 1] * ((1.f - a) * lm[2] + a * lc[2]);
                    g += a * __expf(wl - rowM[xb + 1]) * (G - rowV[xb + 1]);
                }
            }
            if (has_disp) g += gd * __expf(Lc - m0) * rz0 * (tab.d[n] - dsp);
            if constexpr (NHWC) {
                gv[n & 7] = g;
                if ((n & 7) == 7) store8(Grow + (int64_t)x * cpad + (n - 7), gv);
            } else {
                Grow[(int64_t)n * HW + x] = (OUT)g;
            }
        }
    }
}

// Backward, LDS-staged form (NHWC output; W <= 256*PPT).  The kernel above reads every logit three times per (n, x) with
// 4-byte lane loads (x-1, x, x+1) and stores 16 B per lane at a cpad stride: ~2.6x its HBM time.  Here the workgroup streams
// CH plane rows of its image row into LDS with coalesced 16-byte loads (each logit leaves L2 once), the three taps become LDS
// reads, and every thread keeps its pixels' constants in registers across the chunks; a chunk = 8 planes = one 16-B bf16
// (32-B f32) store per pixel.
template <typename OUT, int PPT>
__global__ __launch_bounds__(HEAD_THREADS) void med_head_bwd_lds_kernel(
    const float* __restrict__ dlog0, const float* __restrict__ left, const float* __restrict__ min_disp,
    const float* __restrict__ max_disp, const float* __restrict__ disp, const float* __restrict__ p_im0,
    const float* __restrict__ stats, const float* __restrict__ gdisp, const float* __restrict__ gpan,
    OUT* __restrict__ gdlog0, int N, int H, int W, int cpad, int pair) {
    static_assert(CH == 8, "one chunk = one 8-channel store");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    PlaneTab& tab = *reinterpret_cast<PlaneTab*>(smem);
    const int WP = W + 3;
    float* rowU = reinterpret_cast<float*>(smem + sizeof(PlaneTab));  // [3][WP]
    float* rowV = rowU + 3 * WP;
    float* rowM = rowV + WP;
    const int PP = (W + 11) & ~3;  // plane-row pitch in floats: x lives at index x+4 (16-B aligned fills), zeros at 3 and >= W+4
    float* prow = reinterpret_cast<float*>(smem + (sizeof(PlaneTab) + (size_t)5 * WP * sizeof(float) + 15) / 16 * 16);  // [CH][PP]
    const int b = blockIdx.x / H, y = blockIdx.x % H;
    const int64_t HW = (int64_t)H * W;
    const int64_t rowoff = (int64_t)y * W;
    build_plane_tab(tab, min_disp[b], max_disp[b], N, W);
    const bool has_pan = gpan != nullptr, has_disp = gdisp != nullptr;
    const float* st = stats + (int64_t)b * 4 * HW + rowoff;
    if (has_pan) {
        for (int i = threadIdx.x; i < WP; i += blockDim.x) {
            const int x = i - 1;
            float u0 = 0.f, u1 = 0.f, u2 = 0.f, v = 0.f, m = 0.f;
            if (x >= 0 && x < W) {
                const float rz = 1.f / st[3 * HW + x];
                const float g0 = gpan[((int64_t)b * 3 + 0) * HW + rowoff + x];
                const float g1 = gpan[((int64_t)b * 3 + 1) * HW + rowoff + x];
                const float g2 = gpan[((int64_t)b * 3 + 2) * HW + rowoff + x];
                const float q = g0 * p_im0[((int64_t)b * 3 + 0) * HW + rowoff + x] +
                                g1 * p_im0[((int64_t)b * 3 + 1) * HW + rowoff + x] +
                                g2 * p_im0[((int64_t)b * 3 + 2) * HW + rowoff + x];
                u0 = g0 * rz;
                u1 = g1 * rz;
                u2 = g2 * rz;
                v = q * rz;
                m = st[2 * HW + x];
            }
            rowU[i] = u0;
            rowU[WP + i] = u1;
            rowU[2 * WP + i] = u2;
            rowV[i] = v;
            rowM[i] = m;
        }
    }
    for (int i = threadIdx.x; i < CH * PP; i += blockDim.x) {
        const int c = i % PP;
        if (c < 4 || c >= W + 4) prow[i] = 0.f;  // out-of-row taps read 0 (the logit of a pixel outside the image)
    }
    const float* Lrow = dlog0 + (int64_t)b * N * HW + rowoff;
    OUT* Grow = gdlog0 + ((int64_t)b * HW + rowoff) * cpad;
    const bool vec4 = (W & 3) == 0 && ((reinterpret_cast<uintptr_t>(Lrow) & 15) == 0) && ((HW & 3) == 0);

    // per-pixel constants across planes
    float lm[PPT][3], lc[PPT][3], lp[PPT][3], m0[PPT], rz0[PPT], gd[PPT], dsp[PPT];
#pragma unroll
    for (int q = 0; q < PPT; ++q) {
        const int x = threadIdx.x + q * HEAD_THREADS;
        m0[q] = rz0[q] = gd[q] = dsp[q] = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) lm[q][c] = lc[q][c] = lp[q][c] = 0.f;
        if (x < W) {
            if (has_pan) {
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float* lr = left + ((int64_t)b * 3 + c) * HW + rowoff;
                    lm[q][c] = x > 0 ? lr[x - 1] : 0.f;
                    lc[q][c] = lr[x];
                    lp[q][c] = x + 1 < W ? lr[x + 1] : 0.f;
                }
            }
            if (has_disp) {
                m0[q] = st[x];
                rz0[q] = 1.f / st[HW + x];
                gd[q] = gdisp[(int64_t)b * HW + rowoff + x];
                dsp[q] = disp[(int64_t)b * HW + rowoff + x];
            }
        }
    }
    // register prefetch of the next chunk's plane rows (vec4 path): its global round trip overlaps this chunk's arithmetic
    constexpr int PF = (CH * PPT + 3) / 4;
    float4 pf[PF];
    const int w4 = W >> 2;
    auto fetch = [&](int n0) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int i = threadIdx.x + u * HEAD_THREADS;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < CH * w4) {
                const int j = i / w4, q4 = i % w4;
                if (n0 + j < N) v = reinterpret_cast<const float4*>(Lrow + (int64_t)(n0 + j) * HW)[q4];
            }
            pf[u] = v;
        }
    };
    if (vec4) fetch(0);
    uint4 held[PPT];
    for (int n0 = 0; n0 < cpad; n0 += CH) {
        __syncthreads();  // previous chunk fully consumed (and tab / rows ready)
        if (n0 < N) {
            if (vec4) {
#pragma unroll
                for (int u = 0; u < PF; ++u) {
                    const int i = threadIdx.x + u * HEAD_THREADS;
                    if (i < CH * w4) *reinterpret_cast<float4*>(prow + (i / w4) * PP + 4 + 4 * (i % w4)) = pf[u];
                }
            } else {
                for (int i = threadIdx.x; i < CH * W; i += blockDim.x) {
                    const int j = i / W, x = i % W;
                    prow[j * PP + 4 + x] = n0 + j < N ? Lrow[(int64_t)(n0 + j) * HW + x] : 0.f;
                }
            }
        }
        __syncthreads();
        if (vec4 && n0 + CH < N) fetch(n0 + CH);
#pragma unroll
        for (int q = 0; q < PPT; ++q) {
            const int x = threadIdx.x + q * HEAD_THREADS;
            if (x >= W) continue;
            float gv[8];
            // 16-bit output: a chunk is 16 B per pixel -- half a 32-B sector.  Stored chunk by chunk, the 8 pieces of a pixel's 128-B
            // line reach L2 microseconds apart and many are evicted half-written (WRITE_SIZE was 2x the tensor).  Even chunks are
            // therefore held (packed, 4 registers per pixel) and written together with the odd chunk that completes their sector.
            constexpr bool PAIR = sizeof(OUT) == 2;
            const bool second = ((n0 / CH) & 1) != 0;
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                const int n = n0 + j;
                float g = 0.f;
                if (n < N) {
                    const float* pr = prow + j * PP + 4 + x;
                    const float Lc = pr[0];
                    if (has_pan) {
                        const int k = tab.k[n];
                        const float a = tab.a[n];
                        const float Lm = pr[-1], Lp = pr[1];
                        const int xa = x - k;  // source pixel whose first tap is x
                        if (xa >= 0) {
                            const float wl = (1.f - a) * Lc + a * Lp;
                            const float G = rowU[xa + 1] * ((1.f - a) * lc[q][0] + a * lp[q][0]) +
                                            rowU[WP + xa + 1] * ((1.f - a) * lc[q][1] + a * lp[q][1]) +
                                            rowU[2 * WP + xa + 1] * ((1.f - a) * lc[q][2] + a * lp[q][2]);
                            g += (1.f - a) * __expf(wl - rowM[xa + 1]) * (G - rowV[xa + 1]);
                        }
                        const int xb = xa - 1;  // source pixel whose second tap is x
                        if (xb >= 0) {
                            const float wl = (1.f - a) * Lm + a * Lc;
                            const float G = rowU[xb + 1] * ((1.f - a) * lm[q][0] + a * lc[q][0]) +
                                            rowU[WP + xb + 1] * ((1.f - a) * lm[q][1] + a * lc[q][1]) +
                                            rowU[2 * WP + xb + 1] * ((1.f - a) * lm[q][2] + a * lc[q][2]);
                            g += a * __expf(wl - rowM[xb + 1]) * (G - rowV[xb + 1]);
                        }
                    }
                    if (has_disp) g += gd[q] * __expf(Lc - m0[q]) * rz0[q] * (tab.d[n] - dsp[q]);
                }
                gv[j] = g;
            }
            if constexpr (PAIR) {
                uint4 pk;
#pragma unroll
                for (int i = 0; i < 4; ++i) (&pk.x)[i] = pack16x2<OUT>(gv[2 * i], gv[2 * i + 1]);
                if (!second && n0 + CH < cpad && pair) held[q] = pk;
                else if (!second || !pair) *reinterpret_cast<uint4*>(Grow + (int64_t)x * cpad + n0) = pk;  // odd number of chunks: last one alone
                else {
                    uint4* dst = reinterpret_cast<uint4*>(Grow + (int64_t)x * cpad + n0 - CH);
                    dst[0] = held[q];
                    dst[1] = pk;
                }
            } else {
                store8(Grow + (int64_t)x * cpad + n0, gv);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------- masks
// maskR(x) = min(1, sum_n (1-a) sm_n(x+k) + a sm_n(x+k+1)),   sm = softmax(dlog0)        (FAL_netB.py:266-267)
// maskL(x) = min(1, sum_n a Dprob_n(x-k-1) + (1-a) Dprob_n(x-k))  (shift by -s_n)         (FAL_netB.py:270-273)
// Dprob_n(x) is rebuilt from its two logit taps and the saved (maxW, sumW): no N*HW workspace.
__global__ __launch_bounds__(HEAD_THREADS) void med_masks_kernel(
    const float* __restrict__ dlog0, const float* __restrict__ min_disp, const float* __restrict__ max_disp,
    const float* __restrict__ stats, float* __restrict__ maskL, float* __restrict__ maskR, int N, int H, int W) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    PlaneTab& tab = *reinterpret_cast<PlaneTab*>(smem);
    const int WP = W + 3;
    float* rM0 = reinterpret_cast<float*>(smem + sizeof(PlaneTab));  // index x+1
    float* rZ0 = rM0 + WP;   // 1/sum0, 0 outside the row
    float* rMw = rZ0 + WP;
    float* rZw = rMw + WP;   // 1/sumW, 0 outside
    const int b = blockIdx.x / H, y = blockIdx.x % H;
    const int64_t HW = (int64_t)H * W;
    const int64_t rowoff = (int64_t)y * W;
    build_plane_tab(tab, min_disp[b], max_disp[b], N, W);
    const float* st = stats + (int64_t)b * 4 * HW + rowoff;
    for (int i = threadIdx.x; i < WP; i += blockDim.x) {
        const int x = i - 1;
        const bool in = x >= 0 && x < W;
        rM0[i] = in ? st[x] : 0.f;
        rZ0[i] = in ? 1.f / st[HW + x] : 0.f;
        rMw[i] = in ? st[2 * HW + x] : 0.f;
        rZw[i] = in ? 1.f / st[3 * HW + x] : 0.f;
    }
    __syncthreads();
    const float* Lrow = dlog0 + (int64_t)b * N * HW + rowoff;
    for (int x = threadIdx.x; x < W; x += blockDim.x) {
        float mr = 0.f, ml = 0.f;
        for (int n = 0; n < N; ++n) {
            const float* Ln = Lrow + (int64_t)n * HW;
            const int k = tab.k[n];
            const float a = tab.a[n];
            // right mask: samples of softmax(dlog0)_n at x+k, x+k+1 (zero outside the image)
            const int i0 = min(x + k, W), i1 = min(x + k + 1, W);
            const float t0 = i0 < W ? Ln[i0] : 0.f, t1 = i1 < W ? Ln[i1] : 0.f;
            mr += (1.f - a) * __expf(t0 - rM0[i0 + 1]) * rZ0[i0 + 1] + a * __expf(t1 - rM0[i1 + 1]) * rZ0[i1 + 1];
            // left mask: samples of Dprob_n at x-k-1 and x-k
            const float Lc = Ln[x];
            const float Lm = x > 0 ? Ln[x - 1] : 0.f;
            const float Lp = x + 1 < W ? Ln[x + 1] : 0.f;
            const int xa = x - k, xb = xa - 1;
            if (xa >= 0) ml += (1.f - a) * __expf((1.f - a) * Lc + a * Lp - rMw[xa + 1]) * rZw[xa + 1];
            if (xb >= 0) ml += a * __expf((1.f - a) * Lm + a * Lc - rMw[xb + 1]) * rZw[xb + 1];
        }
        maskR[(int64_t)b * HW + rowoff + x] = fminf(mr, 1.f);
        maskL[(int64_t)b * HW + rowoff + x] = fminf(ml, 1.f);
    }
}

// FAL_netA's right mask (FAL_netA.py:264): softmax(dlog0)_n sampled by grid_sample with its DEFAULT align_corners=False on a grid
// that was built for align_corners=True (:231,:241-242).  Pixel (x, y) therefore reads the bilinear sample (zero padding) at
//   ix = ((xn + 1) W - 1) / 2,  xn = 2x/(W-1) - 1 + 2 d_n / W      iy = ((yn + 1) H - 1) / 2,  yn = 2y/(H-1) - 1
// i.e. a slightly magnified, half-pixel-shifted image: two rows and two columns per plane.  maskR = min(1, sum_n sample_n).
__global__ __launch_bounds__(HEAD_THREADS) void med_maskr_acfalse_kernel(
    const float* __restrict__ dlog0, const float* __restrict__ min_disp, const float* __restrict__ max_disp,
    const float* __restrict__ stats, float* __restrict__ maskR, int N, int H, int W) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    PlaneTab& tab = *reinterpret_cast<PlaneTab*>(smem);
    const int b = blockIdx.x / H, y = blockIdx.x % H;
    const int64_t HW = (int64_t)H * W;
    build_plane_tab(tab, min_disp[b], max_disp[b], N, W);
    __syncthreads();
    const float yn = H > 1 ? 2.f * (float)y / (float)(H - 1) - 1.f : 0.f;
    const float iy = ((yn + 1.f) * (float)H - 1.f) * 0.5f;
    const float y0f = floorf(iy);
    const int y0 = (int)y0f;
    const float wy1 = iy - y0f, wy0 = 1.f - wy1;
    const bool vy0 = y0 >= 0 && y0 < H, vy1 = y0 + 1 >= 0 && y0 + 1 < H;
    const float* st = stats + (int64_t)b * 4 * HW;
    const float* Lb = dlog0 + (int64_t)b * N * HW;
    for (int x = threadIdx.x; x < W; x += blockDim.x) {
        const float xb = W > 1 ? 2.f * (float)x / (float)(W - 1) - 1.f : 0.f;
        float mr = 0.f;
        for (int n = 0; n < N; ++n) {
            const float xn = xb + 2.f * tab.d[n] / (float)W;
            const float ix = ((xn + 1.f) * (float)W - 1.f) * 0.5f;
            const float x0f = floorf(ix);
            if (x0f >= (float)W) continue;  // both columns right of the image
            const int x0 = (int)x0f;
            const float wx1 = ix - x0f, wx0 = 1.f - wx1;
            const float* Ln = Lb + (int64_t)n * HW;
            auto tap = [&](int yy, int xx) -> float {
                if (xx < 0 || xx >= W) return 0.f;
                const int64_t o = (int64_t)yy * W + xx;
                return __expf(Ln[o] - st[o]) / st[HW + o];
            };
            if (vy0) mr += wy0 * (wx0 * tap(y0, x0) + wx1 * tap(y0, x0 + 1));
            if (vy1) mr += wy1 * (wx0 * tap(y0 + 1, x0) + wx1 * tap(y0 + 1, x0 + 1));
        }
        maskR[(int64_t)b * HW + (int64_t)y * W + x] = fminf(mr, 1.f);
    }
}

// ---------------------------------------------------------------------------------------- C-ABI
#include <stdlib.h>
// FALNET_HEAD_V1=1: first forward kernel (per-lane global taps) instead of the LDS-staged one (A/B, tests)
static const bool g_head_v1 = [] { const char* e = falnet_ab_env("FALNET_HEAD_V1"); return e && e[0] == '1'; }();

// wave-neighbour kernels (med_head2.hip)
bool falnet_head_wave_applicable(int W);
bool falnet_head_fwd_lds2_applicable(int N, int W);
bool falnet_head_bwd_lds2_applicable(int N, int W, int dtype);
bool falnet_head_fwd_lds2_launch(const float* dlog0, const float* left, const float* min_disp, const float* max_disp, float* disp,
                                 float* p_im0, float* stats, int B, int N, int H, int W, hipStream_t stream);
bool falnet_head_bwd_lds2_launch(const float* dlog0, const float* left, const float* min_disp, const float* max_disp, const float* disp,
                                 const float* p_im0, const float* stats, const float* gdisp, const float* gpan, void* gdlog0, int cpad,
                                 int dtype, int B, int N, int H, int W, hipStream_t stream);
void falnet_head_bwd_wave_launch(const float* dlog0, const float* left, const float* min_disp, const float* max_disp, const float* disp,
                               const float* p_im0, const float* stats, const float* gdisp, const float* gpan, void* gdlog0, int cpad,
                               int dtype, int B, int N, int H, int W, hipStream_t stream);

static int check_head(int B, int N, int H, int W) {
    FALNET_CHECK_ARG(B > 0 && H > 0 && W > 0, "med_head: empty shape B=%d H=%d W=%d", B, H, W);
    FALNET_CHECK_ARG(N >= 2 && N <= HEAD_MAXN, "med_head: N=%d outside [2,%d]", N, HEAD_MAXN);
    FALNET_CHECK_ARG((size_t)(W + 3) * 5 * 4 + sizeof(PlaneTab) <= 160 * 1024, "med_head: W=%d too wide for LDS", W);
    return 0;
}

extern "C" int falnet_med_head_fwd(const float* dlog0, const float* left, const float* min_disp,
                                   const float* max_disp, float* disp, float* p_im0, float* stats, int B, int N,
                                   int H, int W, void* stream) {
    FALNET_ENTER(stream);
    if (int r = check_head(B, N, H, W)) return r;
    FALNET_CHECK_ARG(dlog0 && min_disp && max_disp, "med_head_fwd: null input");
    FALNET_CHECK_ARG(!p_im0 || left, "med_head_fwd: p_im0 requested without left image");
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    if (!g_head_v1 && al16(dlog0) && al16(left) &&
        falnet_head_fwd_lds2_launch(dlog0, left, min_disp, max_disp, disp, p_im0, stats, B, N, H, W, (hipStream_t)stream))
        FALNET_RETURN_LAUNCH();
    const int ppt = (W + HEAD_THREADS - 1) / HEAD_THREADS;
    const size_t WP = (size_t)((W + 7) & ~3);
    const size_t lds2 = sizeof(PlaneTab) + (3 + CH) * WP * sizeof(float);
    if (ppt <= 8 && lds2 <= 64 * 1024 && !g_head_v1) {
#define LAUNCH_HEAD(P)                                                                                                   \
    hipLaunchKernelGGL(med_head_fwd_lds_kernel<P>, dim3(B * H), dim3(HEAD_THREADS), lds2, (hipStream_t)stream, dlog0, left, \
                       min_disp, max_disp, disp, p_im0, stats, N, H, W)
        switch (ppt) {
            case 1: LAUNCH_HEAD(1); break;
            case 2: LAUNCH_HEAD(2); break;
            case 3: LAUNCH_HEAD(3); break;
            case 4: LAUNCH_HEAD(4); break;
            case 5: LAUNCH_HEAD(5); break;
            case 6: LAUNCH_HEAD(6); break;
            case 7: LAUNCH_HEAD(7); break;
            default: LAUNCH_HEAD(8); break;
        }
#undef LAUNCH_HEAD
        FALNET_RETURN_LAUNCH();
    }
    const size_t lds = sizeof(PlaneTab) + (size_t)3 * (W + 2) * sizeof(float);
    hipLaunchKernelGGL(med_head_fwd_kernel, dim3(B * H), dim3(HEAD_THREADS), lds, (hipStream_t)stream, dlog0, left,
                       min_disp, max_disp, disp, p_im0, stats, N, H, W);
    FALNET_RETURN_LAUNCH();
}

extern "C" int falnet_med_head_bwd(const float* dlog0, const float* left, const float* min_disp,
                                   const float* max_disp, const float* disp, const float* p_im0,
                                   const float* stats, const float* grad_disp, const float* grad_p_im0,
                                   float* grad_dlog0, int B, int N, int H, int W, void* stream) {
    FALNET_ENTER(stream);
    if (int r = check_head(B, N, H, W)) return r;
    FALNET_CHECK_ARG(dlog0 && min_disp && max_disp && stats && grad_dlog0, "med_head_bwd: null input");
    FALNET_CHECK_ARG(!grad_p_im0 || (left && p_im0), "med_head_bwd: grad_p_im0 needs left and p_im0");
    FALNET_CHECK_ARG(!grad_disp || disp, "med_head_bwd: grad_disp needs disp");
    const size_t lds = sizeof(PlaneTab) + (size_t)5 * (W + 3) * sizeof(float);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(med_head_bwd_kernel<float, false>), dim3(B * H), dim3(HEAD_THREADS), lds, (hipStream_t)stream, dlog0, left,
                       min_disp, max_disp, disp, p_im0, stats, grad_disp, grad_p_im0, grad_dlog0, N, H, W, 0);
    FALNET_RETURN_LAUNCH();
}

extern "C" int falnet_med_head_bwd_nhwc(const float* dlog0, const float* left, const float* min_disp,
                                        const float* max_disp, const float* disp, const float* p_im0,
                                        const float* stats, const float* grad_disp, const float* grad_p_im0,
                                        void* grad_dlog0_nhwc, int cpad, int dtype, int B, int N, int H, int W, void* stream) {
    FALNET_ENTER(stream);
    if (int r = check_head(B, N, H, W)) return r;
    FALNET_CHECK_ARG(dlog0 && min_disp && max_disp && stats && grad_dlog0_nhwc, "med_head_bwd_nhwc: null input");
    FALNET_CHECK_ARG(!grad_p_im0 || (left && p_im0), "med_head_bwd_nhwc: grad_p_im0 needs left and p_im0");
    FALNET_CHECK_ARG(!grad_disp || disp, "med_head_bwd_nhwc: grad_disp needs disp");
    FALNET_CHECK_ARG(cpad >= N && cpad % 8 == 0, "med_head_bwd_nhwc: cpad=%d must be a multiple of 8 >= N", cpad);
    FALNET_CHECK_ARG(dtype == FALNET_F32 || dtype == FALNET_BF16 || dtype == FALNET_F16, "med_head_bwd_nhwc: bad dtype %d", dtype);
    const size_t lds = sizeof(PlaneTab) + (size_t)5 * (W + 3) * sizeof(float);
    static const bool v1 = [] { const char* e = falnet_ab_env("FALNET_HEAD_BWD_V1"); return e && e[0] == '1'; }();
    static const int pair = [] { const char* e = falnet_ab_env("FALNET_HEAD_PAIR"); return (e && e[0] == '0') ? 0 : 1; }();
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    if (!v1 && al16(dlog0) && al16(grad_dlog0_nhwc) &&
        falnet_head_bwd_lds2_launch(dlog0, left, min_disp, max_disp, disp, p_im0, stats, grad_disp, grad_p_im0, grad_dlog0_nhwc, cpad, dtype, B, N, H,
                                    W, (hipStream_t)stream))
        FALNET_RETURN_LAUNCH();
    if (!v1 && W > 1024 && falnet_head_wave_applicable(W) && al16(dlog0) && al16(left) && al16(disp) && al16(p_im0) && al16(stats) && al16(grad_disp) &&
        al16(grad_p_im0) && al16(grad_dlog0_nhwc)) {
        falnet_head_bwd_wave_launch(dlog0, left, min_disp, max_disp, disp, p_im0, stats, grad_disp, grad_p_im0, grad_dlog0_nhwc, cpad, dtype, B, N,
                                  H, W, (hipStream_t)stream);
        FALNET_RETURN_LAUNCH();
    }
    if (!v1 && W <= 4 * HEAD_THREADS) {  // LDS-staged plane rows
        const size_t lds2 = (lds + 15) / 16 * 16 + (size_t)CH * ((W + 11) & ~3) * sizeof(float);
#define LAUNCH_BWD_LDS(T, P)                                                                                                   \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(med_head_bwd_lds_kernel<T, P>), dim3(B * H), dim3(HEAD_THREADS), lds2, (hipStream_t)stream, dlog0, left, \
                       min_disp, max_disp, disp, p_im0, stats, grad_disp, grad_p_im0, (T*)grad_dlog0_nhwc, N, H, W, cpad, pair)
        const int ppt = (W + HEAD_THREADS - 1) / HEAD_THREADS;
#define BWD_LDS_T(T) if (ppt <= 1) LAUNCH_BWD_LDS(T, 1); else if (ppt == 2) LAUNCH_BWD_LDS(T, 2); else LAUNCH_BWD_LDS(T, 4)
        FALNET_DISPATCH_DTYPE(dtype, BWD_LDS_T);
#undef BWD_LDS_T
#undef LAUNCH_BWD_LDS
        FALNET_RETURN_LAUNCH();
    }
#define BWD_V1_T(T) hipLaunchKernelGGL(HIP_KERNEL_NAME(med_head_bwd_kernel<T, true>), dim3(B * H), dim3(HEAD_THREADS), lds, (hipStream_t)stream, dlog0, left, \
                                       min_disp, max_disp, disp, p_im0, stats, grad_disp, grad_p_im0, (T*)grad_dlog0_nhwc, N, H, W, cpad)
    FALNET_DISPATCH_DTYPE(dtype, BWD_V1_T);
#undef BWD_V1_T
    FALNET_RETURN_LAUNCH();
}

// Which kernel the three entry points above dispatch to for a shape (16-byte aligned operands assumed: torch allocations are) -- the same
// predicates in the same order.  pass: 0 = falnet_med_head_fwd, 1 = falnet_med_head_bwd (planar f32), 2 = falnet_med_head_bwd_nhwc.
extern "C" int falnet_med_head_kernel_name(int pass, int dtype, int N, int W, char* buf, int len) {
    if (int r = check_head(1, N, 1, W)) return r;
    FALNET_CHECK_ARG(buf && len > 0 && pass >= 0 && pass <= 2, "med_head_kernel_name: bad argument");
    const char* name;
    if (pass == 0) {
        const int ppt = (W + HEAD_THREADS - 1) / HEAD_THREADS;
        const size_t lds2 = sizeof(PlaneTab) + (3 + CH) * (size_t)((W + 7) & ~3) * sizeof(float);
        name = (!g_head_v1 && falnet_head_fwd_lds2_applicable(N, W)) ? (W > 1024 ? "med_head_fwd_lds2_kernel<512 threads>" : "med_head_fwd_lds2_kernel")
               : (!g_head_v1 && ppt <= 8 && lds2 <= 64 * 1024)     ? "med_head_fwd_lds_kernel"
                                                                    : "med_head_fwd_kernel";
    } else if (pass == 1) {
        name = "med_head_bwd_kernel<planar>";
    } else {
        static const bool v1 = [] { const char* e = falnet_ab_env("FALNET_HEAD_BWD_V1"); return e && e[0] == '1'; }();
        name = (!v1 && falnet_head_bwd_lds2_applicable(N, W, dtype)) ? (W > 1024 ? "med_head_bwd_lds2_kernel<512 threads>" : "med_head_bwd_lds2_kernel")
               : (!v1 && W > 1024 && falnet_head_wave_applicable(W)) ? "med_head_bwd_wave_kernel"
               : (!v1 && W <= 4 * HEAD_THREADS)                      ? "med_head_bwd_lds_kernel"
                                                                      : "med_head_bwd_kernel<nhwc>";
    }
    snprintf(buf, (size_t)len, "%s", name);
    return 0;
}

extern "C" int falnet_med_masks_fwd(const float* dlog0, const float* min_disp, const float* max_disp,
                                    const float* stats, float* maskL, float* maskR, int B, int N, int H, int W,
                                    void* stream) {
    FALNET_ENTER(stream);
    if (int r = check_head(B, N, H, W)) return r;
    FALNET_CHECK_ARG(dlog0 && min_disp && max_disp && stats && maskL && maskR, "med_masks_fwd: null input");
    const size_t lds = sizeof(PlaneTab) + (size_t)4 * (W + 3) * sizeof(float);
    hipLaunchKernelGGL(med_masks_kernel, dim3(B * H), dim3(HEAD_THREADS), lds, (hipStream_t)stream, dlog0, min_disp,
                       max_disp, stats, maskL, maskR, N, H, W);
    FALNET_RETURN_LAUNCH();
}

extern "C" int falnet_med_maskr_acfalse_fwd(const float* dlog0, const float* min_disp, const float* max_disp,
                                            const float* stats, float* maskR, int B, int N, int H, int W, void* stream) {
    FALNET_ENTER(stream);
    if (int r = check_head(B, N, H, W)) return r;
    FALNET_CHECK_ARG(dlog0 && min_disp && max_disp && stats && maskR, "med_maskr_acfalse_fwd: null input");
    hipLaunchKernelGGL(med_maskr_acfalse_kernel, dim3(B * H), dim3(HEAD_THREADS), sizeof(PlaneTab), (hipStream_t)stream, dlog0,
                       min_disp, max_disp, stats, maskR, N, H, W);
    FALNET_RETURN_LAUNCH();
}
