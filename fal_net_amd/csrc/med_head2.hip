// MED head, second generation (models/FAL_netB.py:216-282 and its adjoint); med_head.hip keeps the C-ABI, the first kernels (any W,
// unaligned rows), the masks.  PMC on those kernels (tools/pmc_cmd.sh, B=8 256x512 N=49): forward 57 / backward 79 VALU instructions
// per (pixel, plane) with the VALU busy 97 % / 75 % of the launch, LDS 12 / 16 dwords per (pixel, plane) -- instruction bound at
// 2.4 TB/s, not HBM bound.  Two rewrites live here:
//
//  (1) strided form, W <= 2048 (512 threads above 1024), W % 4 == 0 (`*_lds2_kernel`: the Stage-1/2 shapes; at 384x1280 the forward): the same thread -> pixel map (t, t+256, ..) on a
//      diet: plane table in VGPRs read by v_readlane (no LDS table, no dependent round trip per plane), logits pre-scaled by log2(e)
//      while staged (bare v_exp_f32), mixes as t0 + a (t1 - t0), tap index once per (pixel, plane), no per-plane branches, compile-time
//      pitch for W = 512; backward: the five per-source values interleaved ([s+1][5]: both sources of a pixel behind ONE address,
//      conflict free), selects instead of divergent branches, 16-bit output in 32-B sectors.  46 / 61 VALU per (pixel, plane):
//      forward 101 -> 86 us, backward 200 -> 160 us.
//  (2) wave-neighbour form, 1024 < W <= 1984 (`med_head_bwd_wave_kernel`: the 16-bit backward of the 384x1280 high-resolution shape; its
//      forward twin lost to (1)'s 512-thread form, 1097 vs 652 us at 8 x 384 x 1280, N = 96, and was removed in round 4):
//      the plane shift k_n is the same for every pixel, so the second tap of pixel x is the FIRST tap of pixel x+1 -- one lane over.
//      A wave owns 64 consecutive columns, every lane reads ONE tap from LDS and takes its neighbour's with a DPP wavefront shift; the
//      last lane has no neighbour, so waves overlap by one column (63 outputs per wave forward).  Backward: the adjoint of the two-tap
//      warp needs per SOURCE pixel s  T_n(s) = exp(W_n(s) - Mw(s)) (sum_c U_c(s) I_{c,n}(s) - V(s)),
//      grad(x) = (1-a) T_n(x-k) + a T_n(x-k-1) + (disparity term): lane l computes T for ITS source only (one exponential instead of
//      two) and takes the other from lane l+1: 62 outputs per wave.  LDS dwords per (pixel, plane) 12 -> 5 / 16 -> 6.  Every DPP runs
//      with all 64 lanes active: out-of-row lanes clamp their addresses into the zero pads and only skip the stores.
//      At 256x512 this form measured SLOWER than (1) (142 / 176 us: it needs 9 waves per row and the kernels are latency-, not
//      LDS-bound there), at 2x384x1280 162 -> 133 us forward and 306 -> 151 us backward against the first kernels.
//      (Also measured and dropped: four ADJACENT pixels per thread sharing taps -- stride-4 scalar LDS reads are 4-way bank conflicts.)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <math.h>

#include "common.h"

bool falnet_head_wave_applicable(int W);
#define HW_MAXT 1024  // one wave per unit of 62 / 63 output columns: 64 * ceil(units / R) threads, R rounds only for rows wider than 16 units
#define HW_CH 8  // plane rows staged per chunk = planes per online-softmax rescale = channels per gradient store

struct WavePlanes {  // plane n <-> lane n & 63 of register n >> 6
    float d[2], a[2];
    int k[2];
};

__device__ __forceinline__ WavePlanes wave_build_planes(float mn, float mx, int N, int W) {
    WavePlanes t;
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int n = lane + 64 * h;
        const float c = (float)n / (float)(N - 1);
        const float d = mx * expf(logf(mx / mn) * (c - 1.0f));  // FAL_netB.py:223-225
        const float s = d * (float)(W - 1) / (float)W;           // 2d/W normalised * (W-1)/2 (align_corners=True)
        const float kf = floorf(s);
        t.d[h] = d;
        t.k[h] = (int)kf;
        t.a[h] = s - kf;
    }
    return t;
}

__device__ __forceinline__ void wave_plane(const WavePlanes& t, int n, float& d, float& a, int& k) {  // n wave-uniform
    const int h = n >> 6, l = n & 63;
    const int di = h ? __builtin_bit_cast(int, t.d[1]) : __builtin_bit_cast(int, t.d[0]);
    const int ai = h ? __builtin_bit_cast(int, t.a[1]) : __builtin_bit_cast(int, t.a[0]);
    const int ki = h ? t.k[1] : t.k[0];
    d = __builtin_bit_cast(float, __builtin_amdgcn_readlane(di, l));
    a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(ai, l));
    k = __builtin_amdgcn_readlane(ki, l);
}

// value of lane + 1 (lane 63: undefined -- callers never use it); DPP wave_shl:1, all lanes must be active
__device__ __forceinline__ float lane_next(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xF, 0xF, false));
}

// Coalesced 16-byte staging of HW_CH plane rows through registers (the global round trip of chunk n0 + HW_CH overlaps the arithmetic
// of chunk n0): float4 index i = thread + u * blockDim over [HW_CH][W / 4].
template <int PF>
struct RowFetch {
    float4 v[PF];
    __device__ __forceinline__ void load(const float* Lrow, int64_t HW, int n0, int N, int w4) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int i = threadIdx.x + u * blockDim.x;
            float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < HW_CH * w4) {
                const int j = i / w4, q = i - j * w4;
                if (n0 + j < N) x = reinterpret_cast<const float4*>(Lrow + (int64_t)(n0 + j) * HW)[q];
            }
            v[u] = x;
        }
    }
    __device__ __forceinline__ void store(float* prow, int pitch, int x_at, int w4) const {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int i = threadIdx.x + u * blockDim.x;
            if (i < HW_CH * w4) {
                const int j = i / w4, q = i - j * w4;
                *reinterpret_cast<float4*>(prow + j * pitch + x_at + 4 * q) = v[u];
            }
        }
    }
};

// ---------------------------------------------------------------------------------------- backward (NHWC gradient)
// Lane l of unit u sits on column x' = 62u - 1 + l: it holds L_n(x'), gets L_n(x'+1) from lane l+1, and computes T_n for the source
// pixel s = x' - k whose first tap is x'.  Output column x = 62u + j (j = lane, 0..61) = (1-a) T[lane j+1] + a T[lane j] + disparity
// term with L_n(x) = the neighbour's logit.
// LDS: source rows U[3], V, M [WS] (s at index s + 1, index 0 = the zero column of every source left of the image) |
//      plane rows [HW_CH][PP], x at index x + 4 (zeros at 0..3 and W+4..W+7).
template <typename OUT, int R>
__global__ __launch_bounds__(HW_MAXT) void med_head_bwd_wave_kernel(
    const float* __restrict__ dlog0, const float* __restrict__ left, const float* __restrict__ min_disp,
    const float* __restrict__ max_disp, const float* __restrict__ disp, const float* __restrict__ p_im0,
    const float* __restrict__ stats, const float* __restrict__ gdisp, const float* __restrict__ gpan,
    OUT* __restrict__ gdlog0, int N, int H, int W, int cpad) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int WS = W + 4, PP = W + 8;
    float* rowU = reinterpret_cast<float*>(smem);  // [3][WS]
    float* rowV = rowU + 3 * WS;
    float* rowM = rowV + WS;
    float* prow = rowM + WS;  // [HW_CH][PP]   (5 * WS * 4 bytes is a multiple of 16)
    const int b = blockIdx.x / H, y = blockIdx.x % H;
    const int64_t HW = (int64_t)H * W;
    const int64_t rowoff = (int64_t)y * W;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int w4 = W >> 2;
    const WavePlanes tab = wave_build_planes(min_disp[b], max_disp[b], N, W);
    const bool has_pan = gpan != nullptr, has_disp = gdisp != nullptr;
    const float* st = stats + (int64_t)b * 4 * HW + rowoff;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = threadIdx.x; i < WS; i += blockDim.x) {
        const int s = i - 1;
        float u0 = 0.f, u1 = 0.f, u2 = 0.f, v = 0.f, m = 0.f;
        if (has_pan && s >= 0 && s < W) {
            const float rz = 1.f / st[3 * HW + s];
            const float g0 = gpan[((int64_t)b * 3 + 0) * HW + rowoff + s];
            const float g1 = gpan[((int64_t)b * 3 + 1) * HW + rowoff + s];
            const float g2 = gpan[((int64_t)b * 3 + 2) * HW + rowoff + s];
            const float q = g0 * p_im0[((int64_t)b * 3 + 0) * HW + rowoff + s] + g1 * p_im0[((int64_t)b * 3 + 1) * HW + rowoff + s] +
                            g2 * p_im0[((int64_t)b * 3 + 2) * HW + rowoff + s];
            u0 = g0 * rz;
            u1 = g1 * rz;
            u2 = g2 * rz;
            v = q * rz;
            m = st[2 * HW + s];
        }
        rowU[i] = u0;
        rowU[WS + i] = u1;
        rowU[2 * WS + i] = u2;
        rowV[i] = v;
        rowM[i] = m;
    }
    for (int i = threadIdx.x; i < 2 * HW_CH; i += blockDim.x)  // zero head / tail of every plane row
        *reinterpret_cast<float4*>(prow + (i >> 1) * PP + ((i & 1) ? W + 4 : 0)) = zero4;
    const float* Lrow = dlog0 + (int64_t)b * N * HW + rowoff;

    int xq[R];     // x' of this lane per round
    bool unit[R];  // wave-uniform
    float lc0[R][3], lc1[R][3];  // left image at x', x'+1 (zero outside the row)
    float m0[R], rz0[R], gd[R], dsp[R];  // disparity-term factors of the OUTPUT column x = x' + 1
#pragma unroll
    for (int q = 0; q < R; ++q) {
        const int X0 = 62 * (wave + nw * q);
        unit[q] = X0 < W;
        const int xp = X0 - 1 + lane;
        xq[q] = xp;
        m0[q] = rz0[q] = gd[q] = dsp[q] = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float v = (has_pan && unit[q] && xp >= 0 && xp < W) ? left[((int64_t)b * 3 + c) * HW + rowoff + xp] : 0.f;
            lc0[q][c] = v;
        }
        const int x = xp + 1;
        if (has_disp && unit[q] && x < W) {
            m0[q] = st[x];
            rz0[q] = 1.f / st[HW + x];
            gd[q] = gdisp[(int64_t)b * HW + rowoff + x];
            dsp[q] = disp[(int64_t)b * HW + rowoff + x];
        }
    }
#pragma unroll
    for (int q = 0; q < R; ++q)  // (all lanes active again)
#pragma unroll
        for (int c = 0; c < 3; ++c) lc1[q][c] = lane_next(lc0[q][c]);
    constexpr int PF = 2 * R;
    RowFetch<PF> pf;
    pf.load(Lrow, HW, 0, N, w4);
    constexpr bool PAIR = sizeof(OUT) == 2;  // 16-bit output: hold even chunks, write whole 32-B sectors (see med_head.hip)
    uint4 held[R];
    for (int n0 = 0; n0 < cpad; n0 += HW_CH) {
        __syncthreads();  // previous chunk fully consumed (and the source rows / zero columns written)
        if (n0 < N) pf.store(prow, PP, 4, w4);
        __syncthreads();
        if (n0 + HW_CH < N) pf.load(Lrow, HW, n0 + HW_CH, N, w4);
        const bool second = ((n0 / HW_CH) & 1) != 0;
#pragma unroll
        for (int q = 0; q < R; ++q) {
            if (!unit[q]) continue;  // wave-uniform
            const int xp = xq[q];
            const int pi = min(xp, W + 3) + 4;  // index of x' in a plane row (x' = -1 -> zero head, beyond the row -> zero tail)
            float gv[HW_CH];
            if (n0 < N) {  // wave-uniform; no per-plane branches inside (planes >= N of the last chunk: zero rows, result discarded)
                float P[HW_CH], pd[HW_CH], pa[HW_CH];
                int pk[HW_CH];
#pragma unroll
                for (int j = 0; j < HW_CH; ++j) {
                    wave_plane(tab, n0 + j, pd[j], pa[j], pk[j]);
                    P[j] = prow[j * PP + pi];
                }
#pragma unroll
                for (int j = 0; j < HW_CH; ++j) {
                    const float a = pa[j];
                    const int k = pk[j];
                    const float Pn = lane_next(P[j]);  // L_n(x' + 1) = L_n(x) of this lane's output column
                    float g = 0.f;
                    if (has_pan) {
                        const int si = min(max(xp - k, -1), W + 2) + 1;  // sources left of the image: the zero column (U = V = 0 -> T = 0)
                        const float wl = (1.f - a) * P[j] + a * Pn;
                        const float G = rowU[si] * ((1.f - a) * lc0[q][0] + a * lc1[q][0]) + rowU[WS + si] * ((1.f - a) * lc0[q][1] + a * lc1[q][1]) +
                                        rowU[2 * WS + si] * ((1.f - a) * lc0[q][2] + a * lc1[q][2]);
                        float T = __expf(wl - rowM[si]) * (G - rowV[si]);
                        T = xp - k >= 0 ? T : 0.f;  // (exp of an un-normalised logit may be inf: select, do not rely on 0 * x)
                        g = (1.f - a) * lane_next(T) + a * T;
                    }
                    if (has_disp) g += gd[q] * __expf(Pn - m0[q]) * rz0[q] * (pd[j] - dsp[q]);
                    gv[j] = n0 + j < N ? g : 0.f;
                }
            } else {
#pragma unroll
                for (int j = 0; j < HW_CH; ++j) gv[j] = 0.f;
            }
            const int x = xp + 1;
            const bool ok = lane < 62 && x < W;
            OUT* dst = gdlog0 + ((int64_t)b * HW + rowoff + min(x, W - 1)) * cpad + n0;
            if constexpr (PAIR) {
                uint4 pk;
#pragma unroll
                for (int u = 0; u < 4; ++u) (&pk.x)[u] = pack16x2<OUT>(gv[2 * u], gv[2 * u + 1]);
                if (!second && n0 + HW_CH < cpad) held[q] = pk;
                else if (ok) {
                    if (!second) *reinterpret_cast<uint4*>(dst) = pk;  // odd number of chunks: the last one alone
                    else {
                        reinterpret_cast<uint4*>(dst - HW_CH)[0] = held[q];
                        reinterpret_cast<uint4*>(dst - HW_CH)[1] = pk;
                    }
                }
            } else if (ok) {
                reinterpret_cast<float4*>(dst)[0] = make_float4(gv[0], gv[1], gv[2], gv[3]);
                reinterpret_cast<float4*>(dst)[1] = make_float4(gv[4], gv[5], gv[6], gv[7]);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------- forward, strided form (W <= 1024)
// The LDS-staged forward of med_head.hip (thread t owns pixels t, t+256, ...) re-issued with the instruction count cut from ~57 to
// ~25 VALU per (pixel, plane) -- PMC: that kernel's VALU was busy 97 % of its 96 us, i.e. it was instruction bound, not HBM bound:
//   * plane table in VGPRs + v_readlane (SGPR operands) instead of two dependent LDS round trips per plane;
//   * logits are pre-scaled by log2(e) while they are staged, so every exponential is a bare v_exp_f32 (the statistics are scaled
//     back when they are stored);  * the bilinear mixes as t0 + a (t1 - t0);  * tap index computed once per (pixel, plane);
//   * compile-time row pitch (WPC) for the common widths: every LDS offset is an immediate;  * no per-plane branches (planes >= N of
//     the last chunk are masked once), so a chunk's LDS reads issue back to back.
template <int PPT, int WPC, int NT>
__global__ __launch_bounds__(NT) void med_head_fwd_lds2_kernel(
    const float* __restrict__ dlog0, const float* __restrict__ left, const float* __restrict__ min_disp,
    const float* __restrict__ max_disp, float* __restrict__ disp, float* __restrict__ p_im0,
    float* __restrict__ stats, int N, int H, int W) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int WP = WPC ? WPC : ((W + 7) & ~3);      // row pitch in floats: >= W + 4, multiple of 4 (zero tail)
    float* lrow = reinterpret_cast<float*>(smem);   // [3][WP]
    float* prow = lrow + 3 * WP;                    // [HW_CH][WP]
    const int b = blockIdx.x / H, y = blockIdx.x % H;
    const int64_t HW = (int64_t)H * W;
    const int w4 = W >> 2;
    const WavePlanes tab = wave_build_planes(min_disp[b], max_disp[b], N, W);
    const bool want_pan = p_im0 != nullptr;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
    for (int i = threadIdx.x; i < 3 * w4; i += NT) {
        const int c = i / w4, q = i - c * w4;
        *reinterpret_cast<float4*>(lrow + c * WP + 4 * q) =
            want_pan ? reinterpret_cast<const float4*>(left + ((int64_t)b * 3 + c) * HW + (int64_t)y * W)[q] : zero4;
    }
    for (int i = threadIdx.x; i < (3 + HW_CH) * ((WP - W) >> 2); i += NT) {  // zero tail of every LDS row
        const int per = (WP - W) >> 2;
        *reinterpret_cast<float4*>(lrow + (i / per) * WP + W + 4 * (i % per)) = zero4;
    }
    const float* Lrow = dlog0 + (int64_t)b * N * HW + (int64_t)y * W;

    float m0[PPT], z0[PPT], dacc[PPT], mw[PPT], zw[PPT], p0[PPT], p1[PPT], p2[PPT];
#pragma unroll
    for (int q = 0; q < PPT; ++q) {
        m0[q] = mw[q] = -INFINITY;
        z0[q] = dacc[q] = zw[q] = p0[q] = p1[q] = p2[q] = 0.f;
    }
    constexpr int PF = (HW_CH * PPT + 3) / 4;  // float4 per thread: HW_CH rows of W <= NT PPT floats over NT threads
    float4 pf[PF];
    auto fetch = [&](int n0) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int i = threadIdx.x + u * NT;
            float4 v = zero4;
            if (i < HW_CH * w4) {
                const int j = i / w4, q = i - j * w4;
                if (n0 + j < N) v = reinterpret_cast<const float4*>(Lrow + (int64_t)(n0 + j) * HW)[q];
            }
            pf[u] = v;
        }
    };
    fetch(0);
    for (int n0 = 0; n0 < N; n0 += HW_CH) {
        __syncthreads();  // previous chunk fully consumed (and the left row / tails written)
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int i = threadIdx.x + u * NT;
            if (i < HW_CH * w4) {
                const int j = i / w4, q = i - j * w4;
                *reinterpret_cast<float4*>(prow + j * WP + 4 * q) =
                    make_float4(pf[u].x * LOG2E, pf[u].y * LOG2E, pf[u].z * LOG2E, pf[u].w * LOG2E);
            }
        }
        __syncthreads();
        if (n0 + HW_CH < N) fetch(n0 + HW_CH);
        float pd[HW_CH], pa[HW_CH];  // wave-uniform (SGPR) plane constants of this chunk
        int pk[HW_CH];
#pragma unroll
        for (int j = 0; j < HW_CH; ++j) wave_plane(tab, n0 + j, pd[j], pa[j], pk[j]);  // n0 + j < 64 + HW_CH: inside the 128-entry table
        const bool partial = n0 + HW_CH > N;
#pragma unroll
        for (int q = 0; q < PPT; ++q) {
            const int x = threadIdx.x + q * NT;
            if (x >= W) continue;
            float l0[HW_CH], lw[HW_CH];
            int i0[HW_CH];
#pragma unroll
            for (int j = 0; j < HW_CH; ++j) {
                const float* pr = prow + j * WP;
                l0[j] = pr[x];
                i0[j] = min(x + pk[j], W);
                const float t0 = pr[i0[j]], t1 = pr[i0[j] + 1];  // zero tail: an out-of-image logit is 0, not -inf
                lw[j] = t0 + pa[j] * (t1 - t0);
            }
            if (partial) {
#pragma unroll
                for (int j = 0; j < HW_CH; ++j)
                    if (n0 + j >= N) l0[j] = lw[j] = -INFINITY;  // weight exp2(-inf) = 0: the plane contributes nothing below
            }
            float cm0 = l0[0], cmw = lw[0];
#pragma unroll
            for (int j = 1; j < HW_CH; ++j) {
                cm0 = fmaxf(cm0, l0[j]);
                cmw = fmaxf(cmw, lw[j]);
            }
            if (cm0 > m0[q]) {
                const float s = __builtin_amdgcn_exp2f(m0[q] - cm0);
                z0[q] *= s;
                dacc[q] *= s;
                m0[q] = cm0;
            }
            if (cmw > mw[q]) {
                const float s = __builtin_amdgcn_exp2f(mw[q] - cmw);
                zw[q] *= s;
                p0[q] *= s;
                p1[q] *= s;
                p2[q] *= s;
                mw[q] = cmw;
            }
#pragma unroll
            for (int j = 0; j < HW_CH; ++j) {
                const float e = __builtin_amdgcn_exp2f(l0[j] - m0[q]);
                z0[q] += e;
                dacc[q] += pd[j] * e;
                const float ew = __builtin_amdgcn_exp2f(lw[j] - mw[q]);
                zw[q] += ew;
                if (want_pan) {
                    const float wa = ew * pa[j], wb = ew - wa;
                    const float* lr = lrow + i0[j];
                    p0[q] += wb * lr[0] + wa * lr[1];
                    p1[q] += wb * lr[WP] + wa * lr[WP + 1];
                    p2[q] += wb * lr[2 * WP] + wa * lr[2 * WP + 1];
                }
            }
        }
    }
#pragma unroll
    for (int q = 0; q < PPT; ++q) {
        const int x = threadIdx.x + q * NT;
        if (x >= W) continue;
        const int64_t pix = (int64_t)y * W + x;
        if (disp) disp[(int64_t)b * HW + pix] = dacc[q] / z0[q];
        if (want_pan) {
            const float r = 1.f / zw[q];
            p_im0[((int64_t)b * 3 + 0) * HW + pix] = p0[q] * r;
            p_im0[((int64_t)b * 3 + 1) * HW + pix] = p1[q] * r;
            p_im0[((int64_t)b * 3 + 2) * HW + pix] = p2[q] * r;
        }
        if (stats) {  // (maxima back in natural-log units: what the backward / mask kernels expect)
            float* st = stats + (int64_t)b * 4 * HW + pix;
            st[0] = m0[q] * LN2;
            st[HW] = z0[q];
            st[2 * HW] = mw[q] * LN2;
            st[3 * HW] = zw[q];
        }
    }
}

bool falnet_head_fwd_lds2_applicable(int N, int W) {
    static const bool off = [] { const char* e = falnet_ab_env("FALNET_HEAD_FWD2"); return e && e[0] == '0'; }();
    return !(off || (W & 3) || W < 4 || W > 2048 || N + HW_CH > 128);  // (plane table: 128 entries, read up to N - 1 + HW_CH - 1)
}

bool falnet_head_fwd_lds2_launch(const float* dlog0, const float* left, const float* min_disp, const float* max_disp, float* disp,
                                 float* p_im0, float* stats, int B, int N, int H, int W, hipStream_t stream) {
    if (!falnet_head_fwd_lds2_applicable(N, W)) return false;
    const int wp = (W + 7) & ~3;
    const size_t lds = (size_t)(3 + HW_CH) * wp * sizeof(float);
#define HW_L(P, C, NT) hipLaunchKernelGGL(HIP_KERNEL_NAME(med_head_fwd_lds2_kernel<P, C, NT>), dim3(B * H), dim3(NT), lds, stream, dlog0, left, min_disp, max_disp, disp, p_im0, stats, N, H, W)
    if (W > 1024) {  // 512 threads: the 384 x 1280 shape (3 pixels per thread)
        const int ppt = (W + 511) / 512;
        if (ppt <= 3) HW_L(3, 0, 512); else HW_L(4, 0, 512);
        return true;
    }
    const int ppt = (W + 255) / 256;
    if (W == 512) HW_L(2, 516, 256);
    else if (ppt == 1) HW_L(1, 0, 256);
    else if (ppt == 2) HW_L(2, 0, 256);
    else if (ppt == 3) HW_L(3, 0, 256);
    else HW_L(4, 0, 256);
#undef HW_L
    return true;
}

// ---------------------------------------------------------------------------------------- backward, strided form (W <= 1024)
// med_head.hip's LDS-staged backward with the same diet as the forward above (PMC: 79 VALU instructions per (pixel, plane), VALU
// busy 75 % of 180 us): plane table by v_readlane, logits pre-scaled by log2(e) (bare v_exp_f32), mixes as t0 + a (t1 - t0) with the
// image differences precomputed per pixel, selects instead of divergent branches, and the five per-source-pixel values
// (U0, U1, U2, V, Mw) interleaved in LDS ([s + 1][5]; entry 0 = the zero source left of the image) so that the two sources of an
// output pixel are TEN consecutive dwords behind one address (lane stride 5 dwords: conflict free).
// (Measured and dropped: one pixel per thread with the pixel's whole 128-B output line held in registers and written at the end --
// no faster than the paired 32-B stores, 167 vs 169 us.)
template <typename OUT, int PPT, int NT>
__global__ __launch_bounds__(NT, NT == 256 ? 4 : 2) void med_head_bwd_lds2_kernel(
    const float* __restrict__ dlog0, const float* __restrict__ left, const float* __restrict__ min_disp,
    const float* __restrict__ max_disp, const float* __restrict__ disp, const float* __restrict__ p_im0,
    const float* __restrict__ stats, const float* __restrict__ gdisp, const float* __restrict__ gpan,
    OUT* __restrict__ gdlog0, int N, int H, int W, int cpad) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int PP = (W + 11) & ~3;  // plane-row pitch in floats: x lives at index x + 4 (16-B aligned fills), zeros at 3 and >= W + 4
    float* prow = reinterpret_cast<float*>(smem);  // [HW_CH][PP]
    float* src5 = prow + HW_CH * PP;               // [W + 1][5]
    const int b = blockIdx.x / H, y = blockIdx.x % H;
    const int64_t HW = (int64_t)H * W;
    const int64_t rowoff = (int64_t)y * W;
    const int w4 = W >> 2;
    const WavePlanes tab = wave_build_planes(min_disp[b], max_disp[b], N, W);
    const bool has_pan = gpan != nullptr, has_disp = gdisp != nullptr;
    const float* st = stats + (int64_t)b * 4 * HW + rowoff;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    constexpr float LOG2E = 1.4426950408889634f;
    for (int i = threadIdx.x; i < W + 1; i += NT) {
        const int sx = i - 1;
        float u0 = 0.f, u1 = 0.f, u2 = 0.f, v = 0.f, m = 0.f;
        if (has_pan && sx >= 0) {
            const float rz = 1.f / st[3 * HW + sx];
            const float g0 = gpan[((int64_t)b * 3 + 0) * HW + rowoff + sx];
            const float g1 = gpan[((int64_t)b * 3 + 1) * HW + rowoff + sx];
            const float g2 = gpan[((int64_t)b * 3 + 2) * HW + rowoff + sx];
            const float q = g0 * p_im0[((int64_t)b * 3 + 0) * HW + rowoff + sx] + g1 * p_im0[((int64_t)b * 3 + 1) * HW + rowoff + sx] +
                            g2 * p_im0[((int64_t)b * 3 + 2) * HW + rowoff + sx];
            u0 = g0 * rz;
            u1 = g1 * rz;
            u2 = g2 * rz;
            v = q * rz;
            m = st[2 * HW + sx] * LOG2E;
        }
        float* e = src5 + i * 5;
        e[0] = u0, e[1] = u1, e[2] = u2, e[3] = v, e[4] = m;
    }
    for (int i = threadIdx.x; i < 2 * HW_CH; i += NT)  // zero head / tail of every plane row
        *reinterpret_cast<float4*>(prow + (i >> 1) * PP + ((i & 1) ? W + 4 : 0)) = zero4;
    const float* Lrow = dlog0 + (int64_t)b * N * HW + rowoff;
    OUT* Grow = gdlog0 + ((int64_t)b * HW + rowoff) * cpad;

    // per-pixel constants: left image at x-1, x and the differences to the next column; disparity-term factors
    float lm[PPT][3], lc[PPT][3], dm[PPT][3], dc[PPT][3], m0[PPT], gr[PPT], dsp[PPT];
#pragma unroll
    for (int q = 0; q < PPT; ++q) {
        const int x = threadIdx.x + q * NT;
        m0[q] = gr[q] = dsp[q] = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) lm[q][c] = lc[q][c] = dm[q][c] = dc[q][c] = 0.f;
        if (x < W) {
            if (has_pan) {
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float* lr = left + ((int64_t)b * 3 + c) * HW + rowoff;
                    const float a0 = x > 0 ? lr[x - 1] : 0.f, a1 = lr[x], a2 = x + 1 < W ? lr[x + 1] : 0.f;
                    lm[q][c] = a0, lc[q][c] = a1;
                    dm[q][c] = a1 - a0, dc[q][c] = a2 - a1;
                }
            }
            if (has_disp) {
                m0[q] = st[x] * LOG2E;
                gr[q] = gdisp[(int64_t)b * HW + rowoff + x] / st[HW + x];
                dsp[q] = disp[(int64_t)b * HW + rowoff + x];
            }
        }
    }
    constexpr int PF = (HW_CH * PPT + 3) / 4;
    float4 pf[PF];
    auto fetch = [&](int n0) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int i = threadIdx.x + u * NT;
            float4 v = zero4;
            if (i < HW_CH * w4) {
                const int j = i / w4, q4 = i - j * w4;
                if (n0 + j < N) v = reinterpret_cast<const float4*>(Lrow + (int64_t)(n0 + j) * HW)[q4];
            }
            pf[u] = v;
        }
    };
    fetch(0);
    constexpr bool PAIR = sizeof(OUT) == 2;  // 16-bit output: hold even chunks, write whole 32-B sectors (see med_head.hip)
    uint4 held[PPT];
    for (int n0 = 0; n0 < cpad; n0 += HW_CH) {
        __syncthreads();  // previous chunk fully consumed (and the source entries / zero columns written)
        if (n0 < N) {
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const int i = threadIdx.x + u * NT;
                if (i < HW_CH * w4) {
                    const int j = i / w4, q4 = i - j * w4;
                    *reinterpret_cast<float4*>(prow + j * PP + 4 + 4 * q4) =
                        make_float4(pf[u].x * LOG2E, pf[u].y * LOG2E, pf[u].z * LOG2E, pf[u].w * LOG2E);
                }
            }
        }
        __syncthreads();
        if (n0 + HW_CH < N) fetch(n0 + HW_CH);
        float pd[HW_CH], pa[HW_CH];
        int pk[HW_CH];
#pragma unroll
        for (int j = 0; j < HW_CH; ++j) wave_plane(tab, n0 + j, pd[j], pa[j], pk[j]);
        const bool second = ((n0 / HW_CH) & 1) != 0;
#pragma unroll
        for (int q = 0; q < PPT; ++q) {
            const int x = threadIdx.x + q * NT;
            if (x >= W) continue;
            float gv[HW_CH];
            if (n0 < N) {  // block-uniform
#pragma unroll
                for (int j = 0; j < HW_CH; ++j) {
                    const float a = pa[j];
                    const float* pr = prow + j * PP + 4 + x;
                    const float Lm = pr[-1], Lc = pr[0], Lp = pr[1];
                    float g = 0.f;
                    if (has_pan) {
                        const int xa = x - pk[j];  // source pixel whose first tap is x (xa - 1: the one whose second tap is x)
                        const float* e = src5 + max(xa, 0) * 5;  // entries of sources xa - 1, xa (index s + 1)
                        const float wb = Lm + a * (Lc - Lm), wa = Lc + a * (Lp - Lc);
                        const float Gb = e[0] * (lm[q][0] + a * dm[q][0]) + e[1] * (lm[q][1] + a * dm[q][1]) + e[2] * (lm[q][2] + a * dm[q][2]);
                        const float Ga = e[5] * (lc[q][0] + a * dc[q][0]) + e[6] * (lc[q][1] + a * dc[q][1]) + e[7] * (lc[q][2] + a * dc[q][2]);
                        const float Tb = __builtin_amdgcn_exp2f(wb - e[4]) * (Gb - e[3]);
                        const float Ta = __builtin_amdgcn_exp2f(wa - e[9]) * (Ga - e[8]);
                        // (select, do not multiply: exp2 of an un-normalised logit may be inf where there is no source)
                        const float ta = xa >= 0 ? Ta : 0.f, tb = xa >= 1 ? Tb : 0.f;
                        g = ta + a * (tb - ta);
                    }
                    if (has_disp) g += gr[q] * __builtin_amdgcn_exp2f(Lc - m0[q]) * (pd[j] - dsp[q]);
                    gv[j] = n0 + j < N ? g : 0.f;
                }
            } else {
#pragma unroll
                for (int j = 0; j < HW_CH; ++j) gv[j] = 0.f;
            }
            OUT* dst = Grow + (int64_t)x * cpad + n0;
            if constexpr (PAIR) {
                uint4 pk4;
#pragma unroll
                for (int u = 0; u < 4; ++u) (&pk4.x)[u] = pack16x2<OUT>(gv[2 * u], gv[2 * u + 1]);
                if (!second && n0 + HW_CH < cpad) held[q] = pk4;
                else if (!second) *reinterpret_cast<uint4*>(dst) = pk4;  // odd number of chunks: the last one alone
                else {
                    reinterpret_cast<uint4*>(dst - HW_CH)[0] = held[q];
                    reinterpret_cast<uint4*>(dst - HW_CH)[1] = pk4;
                }
            } else {
                reinterpret_cast<float4*>(dst)[0] = make_float4(gv[0], gv[1], gv[2], gv[3]);
                reinterpret_cast<float4*>(dst)[1] = make_float4(gv[4], gv[5], gv[6], gv[7]);
            }
        }
    }
}

template <typename T>
static void bwd_lds2_launch_t(const float* dlog0, const float* left, const float* min_disp, const float* max_disp, const float* disp,
                              const float* p_im0, const float* stats, const float* gdisp, const float* gpan, T* gdlog0, int cpad, int B, int N,
                              int H, int W, hipStream_t stream) {
    const size_t lds = (size_t)HW_CH * ((W + 11) & ~3) * sizeof(float) + (size_t)(W + 1) * 5 * sizeof(float);
#define HW_L(P, NT) hipLaunchKernelGGL(HIP_KERNEL_NAME(med_head_bwd_lds2_kernel<T, P, NT>), dim3(B * H), dim3(NT), lds, stream, dlog0, left, min_disp, max_disp, disp, p_im0, stats, gdisp, gpan, gdlog0, N, H, W, cpad)
    if (W > 1024) {
        const int ppt = (W + 511) / 512;
        if (ppt <= 3) HW_L(3, 512); else HW_L(4, 512);
        return;
    }
    const int ppt = (W + 255) / 256;
    if (ppt == 1) HW_L(1, 256);
    else if (ppt == 2) HW_L(2, 256);
    else if (ppt == 3) HW_L(3, 256);
    else HW_L(4, 256);
#undef HW_L
}

bool falnet_head_bwd_lds2_applicable(int N, int W, int dtype) {
    static const bool off = [] { const char* e = falnet_ab_env("FALNET_HEAD_BWD2"); return e && e[0] == '0'; }();
    if (off || (W & 3) || W < 4 || W > 2048 || N + HW_CH > 128) return false;
    // 8 x 384 x 1280, N = 96: 16-bit gradient 991 us here vs 955 us on the wave-neighbour kernel (f32: 1277 vs 1357)
    if (W > 1024 && dtype != FALNET_F32 && falnet_head_wave_applicable(W)) return false;
    return true;
}

bool falnet_head_bwd_lds2_launch(const float* dlog0, const float* left, const float* min_disp, const float* max_disp, const float* disp,
                                 const float* p_im0, const float* stats, const float* gdisp, const float* gpan, void* gdlog0, int cpad,
                                 int dtype, int B, int N, int H, int W, hipStream_t stream) {
    if (!falnet_head_bwd_lds2_applicable(N, W, dtype)) return false;
#define HW_D(T) bwd_lds2_launch_t<T>(dlog0, left, min_disp, max_disp, disp, p_im0, stats, gdisp, gpan, (T*)gdlog0, cpad, B, N, H, W, stream)
    FALNET_DISPATCH_DTYPE(dtype, HW_D);
#undef HW_D
    return true;
}

// ---------------------------------------------------------------------------------------- launch helpers (called from med_head.hip's C-ABI)
bool falnet_head_wave_applicable(int W) {
    static const bool off = [] { const char* e = falnet_ab_env("FALNET_HEAD_WAVE"); return e && e[0] == '0'; }();
    return !off && (W & 3) == 0 && W >= 64 && W <= 62 * 16 * 2;
}

static void wave_geometry(int W, int& rounds, int& threads) {
    const int units = (W + 61) / 62;  // 62 outputs per wave are enough for both kernels (63 / 62)
    rounds = (units + 15) / 16;
    threads = 64 * ((units + rounds - 1) / rounds);
}

template <typename T>
static void bwd_wave_launch_t(const float* dlog0, const float* left, const float* min_disp, const float* max_disp, const float* disp,
                              const float* p_im0, const float* stats, const float* gdisp, const float* gpan, T* gdlog0, int cpad, int B, int N,
                              int H, int W, hipStream_t stream) {
    const size_t lds = (size_t)5 * (W + 4) * sizeof(float) + (size_t)HW_CH * (W + 8) * sizeof(float);
    int r, nt;
    wave_geometry(W, r, nt);
#define HW_B(RR) hipLaunchKernelGGL(HIP_KERNEL_NAME(med_head_bwd_wave_kernel<T, RR>), dim3(B * H), dim3(nt), lds, stream, dlog0, left, min_disp, max_disp, disp, p_im0, stats, gdisp, gpan, gdlog0, N, H, W, cpad)
    if (r == 1) HW_B(1); else HW_B(2);
#undef HW_B
}

void falnet_head_bwd_wave_launch(const float* dlog0, const float* left, const float* min_disp, const float* max_disp, const float* disp,
                                 const float* p_im0, const float* stats, const float* gdisp, const float* gpan, void* gdlog0, int cpad,
                                 int dtype, int B, int N, int H, int W, hipStream_t stream) {
#define HW_D(T) bwd_wave_launch_t<T>(dlog0, left, min_disp, max_disp, disp, p_im0, stats, gdisp, gpan, (T*)gdlog0, cpad, B, N, H, W, stream)
    FALNET_DISPATCH_DTYPE(dtype, HW_D);
#undef HW_D
}
