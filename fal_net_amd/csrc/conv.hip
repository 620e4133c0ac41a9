// Implicit-GEMM convolution on the gfx950 matrix cores (MFMA), forward / data-gradient / weight-gradient.
//
// Replaces every cuDNN conv2d call site of the hot path (models/FAL_netB.py:38,45,55,73,75,127,190;
// loss_functions.py:21-29) and their autograd.  One gather kernel serves forward and dgrad: the GEMM is
//   D[m, n] = sum_k A[m, k] * Bw[n, k],  m = output position, n = output channel, k = (tap, source, channel)
// with A gathered on the fly from one or two NHWC sources (fused channel concat), optionally through a
// nearest-neighbour upsample (fused F.interpolate), zero outside the image (padding=1), and a per-launch
// tap table (stride, dgrad parity classes).  Both operands are staged K-contiguous in LDS in 64-byte rows
// (32 bf16 / 16 f32 of K) padded to 80 B so that the 16-lane ds_read_b128 groups hit 16 distinct 16-B
// slots; global->register->LDS double buffering, one barrier per K step.
//   bf16: v_mfma_f32_32x32x16_bf16, f32 accumulate.
//   f32 : v_mfma_f32_32x32x2_f32 (exact f32, the parity path); K is walked in a lane-half-permuted order
//         (half h owns k in [8h, 8h+8)) so that each lane's operands are two ds_read_b128.
// Wave tiling: 4 waves, 128 positions x {32,64,128} channels per workgroup.
#include <limits.h>
#include <stdlib.h>
#include <type_traits>
#include "common.h"
#include "conv_epilogue.h"

#define CONV_THREADS 256
#define CONV_BM 128
#define ROWB 64   // K bytes per LDS row
#define ROWP 80   // padded row pitch

template <typename T> struct KC;
template <> struct KC<float> { static constexpr int value = 16; };
template <> struct KC<bf16_t> { static constexpr int value = 32; };
template <> struct KC<f16_t> { static constexpr int value = 32; };

// ---- vectorised epilogue ------------------------------------------------------------------------------
// The MFMA C/D layout puts one channel on each lane, so a direct store is 2 B (bf16) per lane: store-issue
// bound.  Instead every wave stages one 32-row accumulator slab at a time through its private LDS area
// (f32, pitch NT*32+4 floats) and then owns (row, 8-channel segment) pieces: bias / residual / activation /
// activation-gradient are applied on 8 values and written with 16-B stores; addend / actout come in 16-B loads.
template <typename T> struct Vec8 {  // 16-bit operand types (bf16 / f16)
    uint4 v;
    __device__ __forceinline__ void load(const T* p) { v = *reinterpret_cast<const uint4*>(p); }
    __device__ __forceinline__ void store(T* p) const { *reinterpret_cast<uint4*>(p) = v; }
    __device__ __forceinline__ float get(int i) const {
        const unsigned w = (&v.x)[i >> 1];
        return (i & 1) ? H16<T>::hi(w) : H16<T>::lo(w);
    }
    __device__ __forceinline__ void set8(const float (&f)[8]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) (&v.x)[i] = pack16x2<T>(f[2 * i], f[2 * i + 1]);
    }
};
template <> struct Vec8<float> {
    float4 a, b;
    __device__ __forceinline__ void load(const float* p) { a = reinterpret_cast<const float4*>(p)[0]; b = reinterpret_cast<const float4*>(p)[1]; }
    __device__ __forceinline__ void store(float* p) const { reinterpret_cast<float4*>(p)[0] = a; reinterpret_cast<float4*>(p)[1] = b; }
    __device__ __forceinline__ float get(int i) const { return i < 4 ? (&a.x)[i] : (&b.x)[i - 4]; }
    __device__ __forceinline__ void set8(const float (&f)[8]) {
        a = make_float4(f[0], f[1], f[2], f[3]);
        b = make_float4(f[4], f[5], f[6], f[7]);
    }
};

// RowOff: functor row(0..31 of slab mt) -> element offset of the output pixel's channel 0, or -1
// PoolOff (optional, MT even): slabs are consecutive image rows and slab rows consecutive columns; called for odd mt /
// even row, returns the element offset of the 2x2-pooled pixel in p.pool_out or -1.  The horizontal neighbour lives in
// lane^CS (one shuffle), the vertical one in the previous slab (kept in registers).
template <typename T, int MT, int NT, typename RowOff, typename PoolOff = NoPool>
__device__ __forceinline__ void epilogue_nhwc(const falnet_conv_t& p, f32x16 (&acc)[MT][NT], float* stage, int nbase, int lane,
                                              RowOff rowoff, PoolOff pooloff = PoolOff()) {
    constexpr bool POOL = !std::is_same<PoolOff, NoPool>::value;
    constexpr int PITCHF = NT * 32 + 4;
    constexpr int CS = NT * 4;          // 8-channel segments per row
    constexpr int RPP = 64 / CS;        // rows per pass
    constexpr int PASSES = 32 / RPP;
    const int r = lane & 31, h = lane >> 5;
    const T* addend = reinterpret_cast<const T*>(p.addend);
    const T* actout = reinterpret_cast<const T*>(p.actout);
    T* out = reinterpret_cast<T*>(p.out);
    const int cs = lane % CS, rsub = lane / CS;
    const int n = nbase + cs * 8;
    const bool nok = n < p.Cout;
    float bias[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) bias[i] = (p.bias && nok) ? p.bias[n + i] : 0.f;
    // residual / activation-output vectors of a whole slab are requested BEFORE that slab is staged (one slab ahead
    // for bf16), so their memory latency overlaps the LDS staging instead of serialising every 8-channel piece
    constexpr int NBUF = sizeof(T) == 2 ? 2 : 1;
    Vec8<T> addv[NBUF][PASSES], actv[NBUF][PASSES];
    auto prefetch = [&](int mt, int buf) {
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) {
            const int64_t o = rowoff(mt, ps * RPP + rsub);
            if (o >= 0 && nok) {
                if (addend) addv[buf][ps].load(addend + o + n);
                if (actout) actv[buf][ps].load(actout + o + n);
            }
        }
    };
    if (addend || actout) prefetch(0, 0);
    T* pool_out = reinterpret_cast<T*>(p.pool_out);
    const bool pooling = POOL && pool_out != nullptr;
    Vec8<T> hprev[POOL ? PASSES : 1];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int buf = NBUF == 2 ? (mt & 1) : 0;
        if (NBUF == 2 && mt + 1 < MT && (addend || actout)) prefetch(mt + 1, (mt + 1) & 1);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int j = 0; j < 16; ++j) stage[((j & 3) + 8 * (j >> 2) + 4 * h) * PITCHF + nt * 32 + r] = acc[mt][nt][j];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) {
            const int row = ps * RPP + rsub;
            const int64_t o = rowoff(mt, row);
            Vec8<T> t;
            if (o >= 0 && nok) {
                const float4 x0 = *reinterpret_cast<const float4*>(stage + row * PITCHF + cs * 8);
                const float4 x1 = *reinterpret_cast<const float4*>(stage + row * PITCHF + cs * 8 + 4);
                float v[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
                if (addend) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] += addv[buf][ps].get(i);
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = apply_act(v[i] + bias[i], p.act);
                if (actout) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] *= act_grad_from_out(actv[buf][ps].get(i), p.actout_kind);
                }
                t.set8(v);
                if (!POOL || out) t.store(out + o + n);
            } else if (POOL) {
                const float z[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                t.set8(z);  // never reaches a valid pooled pixel (floor semantics); defined value for the shuffle
            }
            if constexpr (POOL) {
                if (pooling) {  // workgroup-uniform
                    const bool sum = p.pool_mode == 1;
                    float m[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float a = t.get(i);
                        const float nb = __shfl_xor(a, CS);  // column neighbour: row ^ 1
                        m[i] = sum ? a + nb : fmaxf(a, nb);
                    }
                    if ((mt & 1) == 0) {
                        hprev[ps].set8(m);  // max: exact (the values are already rounded to T)
                    } else {
#pragma unroll
                        for (int i = 0; i < 8; ++i) m[i] = sum ? m[i] + hprev[ps].get(i) : fmaxf(m[i], hprev[ps].get(i));
                        const int64_t po = (row & 1) ? (int64_t)-1 : pooloff(mt, row);
                        if (po >= 0 && nok) {
                            if (p.pool_actout) {
                                Vec8<T> av;
                                av.load(reinterpret_cast<const T*>(p.pool_actout) + po + n);
#pragma unroll
                                for (int i = 0; i < 8; ++i) m[i] *= act_grad_from_out(av.get(i), p.pool_actout_kind);
                            }
                            Vec8<T> q;
                            q.set8(m);
                            q.store(pool_out + po + n);
                        }
                    }
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (NBUF == 1 && mt + 1 < MT && (addend || actout)) prefetch(mt + 1, 0);
    }
}

template <typename T, int MT, int NT, bool SWAP>
__device__ __forceinline__ void mma_tile(const char* __restrict__ As, const char* __restrict__ Bs, int arow0, int brow0,
                                         int lane, f32x16 (&acc)[MT][NT]) {
    const int r = lane & 31, h = lane >> 5;
    if constexpr (sizeof(T) == 2) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            s16x8_t a[MT], b[NT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                a[mt] = *reinterpret_cast<const s16x8_t*>(As + (arow0 + mt * 32 + r) * ROWP + (ks * 2 + h) * 16);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                b[nt] = *reinterpret_cast<const s16x8_t*>(Bs + (brow0 + nt * 32 + r) * ROWP + (ks * 2 + h) * 16);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[mt][nt] = SWAP ? H16<T>::mma(b[nt], a[mt], acc[mt][nt]) : H16<T>::mma(a[mt], b[nt], acc[mt][nt]);
        }
    } else {
        float a[MT][8], b[NT][8];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const float4* p = reinterpret_cast<const float4*>(As + (arow0 + mt * 32 + r) * ROWP + h * 32);
            const float4 x = p[0], y = p[1];
            a[mt][0] = x.x; a[mt][1] = x.y; a[mt][2] = x.z; a[mt][3] = x.w;
            a[mt][4] = y.x; a[mt][5] = y.y; a[mt][6] = y.z; a[mt][7] = y.w;
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const float4* p = reinterpret_cast<const float4*>(Bs + (brow0 + nt * 32 + r) * ROWP + h * 32);
            const float4 x = p[0], y = p[1];
            b[nt][0] = x.x; b[nt][1] = x.y; b[nt][2] = x.z; b[nt][3] = x.w;
            b[nt][4] = y.x; b[nt][5] = y.y; b[nt][6] = y.z; b[nt][7] = y.w;
        }
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[mt][nt] = SWAP ? __builtin_amdgcn_mfma_f32_32x32x2f32(b[nt][s], a[mt][s], acc[mt][nt], 0, 0, 0)
                                       : __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt][s], b[nt][s], acc[mt][nt], 0, 0, 0);
    }
}

// ------------------------------------------------------------------------------------------ fwd / dgrad
template <typename T, int BN, bool SWAP>
__device__ __forceinline__ void conv_igemm_body(const falnet_conv_t& p, const int zi, const int nz) {
    constexpr int BM = CONV_BM;
    constexpr int WAVES_N = BN >= 64 ? 2 : 1;
    constexpr int WAVES_M = 4 / WAVES_N;
    constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
    constexpr int MT = WTM / 32, NT = WTN / 32;
    constexpr int A_LOADS = BM * 4 / CONV_THREADS;                        // 16-B segments per thread
    constexpr int B_LOADS = (BN * 4 + CONV_THREADS - 1) / CONV_THREADS;
    constexpr int KCV = KC<T>::value;
    constexpr int EPS = 16 / sizeof(T);                                   // elements per 16-B segment

    __shared__ __attribute__((aligned(16))) char lds[2 * (BM + BN) * ROWP + BM * 4];
    auto Abuf = [&](int b) -> char* { return lds + b * (BM + BN) * ROWP; };
    auto Bbuf = [&](int b) -> char* { return lds + b * (BM + BN) * ROWP + BM * ROWP; };
    int* outpix = reinterpret_cast<int*>(lds + 2 * (BM + BN) * ROWP);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int64_t M = (int64_t)p.B * p.TH * p.TW;
    const int64_t m0 = (int64_t)blockIdx.x * BM;
    if (m0 >= M) return;  // uniform per block (multi-descriptor launches are sized for the largest member)
    const int n0 = blockIdx.y * BN;
    const int w_rows = p.w_rows;

    // per-row output addresses (shared) and per-thread gather coordinates
    if (tid < BM) {
        const int64_t m = m0 + tid;
        int o = -1;
        if (m < M) {
            const int tx = (int)(m % p.TW), ty = (int)((m / p.TW) % p.TH), b = (int)(m / ((int64_t)p.TW * p.TH));
            const int oy = ty * p.osy + p.ooy, ox = tx * p.osx + p.oox;
            if (oy < p.OH && ox < p.OW)
                o = p.out_layout == FALNET_OUT_PLANAR_F32 ? (b * p.Cout * p.OH + oy) * p.OW + ox : (b * p.OH + oy) * p.OW + ox;
        }
        outpix[tid] = o;
    }
    int a_b[A_LOADS], a_y[A_LOADS], a_x[A_LOADS];
#pragma unroll
    for (int i = 0; i < A_LOADS; ++i) {
        const int row = (tid + i * CONV_THREADS) >> 2;
        const int64_t m = m0 + row;
        if (m < M) {
            a_x[i] = (int)(m % p.TW) * p.isx;
            a_y[i] = (int)((m / p.TW) % p.TH) * p.isy;
            a_b[i] = (int)(m / ((int64_t)p.TW * p.TH));
        } else {
            a_b[i] = -1;
            a_x[i] = a_y[i] = 0;
        }
    }

    // source descriptors and the current tap in registers: a runtime-indexed p.src[s] / p.tap_dy[t] inside the K loop is a scalar
    // LOAD from the kernel-argument segment in front of every gather -- on the small deep layers (K loops of 5-20 dependent
    // global -> LDS -> MFMA rounds per workgroup) that latency is on the critical path of every round
    struct SrcRegs { const T* ptr; int64_t sb, sy, sx; int H, W, C; };
    const SrcRegs S0 = {reinterpret_cast<const T*>(p.src[0].ptr), p.src[0].sb, p.src[0].sy, p.src[0].sx, p.src[0].H, p.src[0].W, p.src[0].C};
    const SrcRegs S1 = p.nsrc > 1 ? SrcRegs{reinterpret_cast<const T*>(p.src[1].ptr), p.src[1].sb, p.src[1].sy, p.src[1].sx, p.src[1].H, p.src[1].W, p.src[1].C} : S0;
    const int ntaps = p.ntaps, IH = p.IH, IW = p.IW, cin_total = p.cin_total;
    const int ch0 = S0.C / KCV, ch1 = p.nsrc > 1 ? S1.C / KCV : 0;   // K chunks per tap of each source
    const int niter_all = ntaps * (ch0 + ch1);
    // split-K: blockIdx.z owns K iterations [it0, it1); partial sums go to an f32 workspace with atomics
    const int it0 = (int)((int64_t)niter_all * zi / nz), it1 = (int)((int64_t)niter_all * (zi + 1) / nz);
    const int niter = it1 - it0;

    // K-walk state: source s, tap t, channel offset c0; srcoff = packed-weight offset of source s.  Closed form of the walk
    // (source-major, then tap, then chunk) at iteration it0 -- the last split of a 16-way launch would otherwise step there
    int s_ = 0, t_ = 0, c0_ = 0, srcoff_ = 0;
    {
        int r = it0;
        if (r >= ntaps * ch0 && ch1 > 0) {
            r -= ntaps * ch0;
            s_ = 1;
            srcoff_ = S0.C;
        }
        const int chs = s_ ? ch1 : ch0;
        if (chs > 0) {
            t_ = r / chs;
            c0_ = (r % chs) * KCV;
        }
    }
    int cur_dy = p.tap_dy[t_ < ntaps ? t_ : 0], cur_dx = p.tap_dx[t_ < ntaps ? t_ : 0], cur_w = p.tap_w[t_ < ntaps ? t_ : 0];
    uint4 areg[A_LOADS], breg[B_LOADS];

    auto gload = [&]() {
        const SrcRegs S = s_ == 0 ? S0 : S1;
        const int dy = cur_dy, dx = cur_dx;
        const bool ups = (S.H != IH) || (S.W != IW);
#pragma unroll
        for (int i = 0; i < A_LOADS; ++i) {
            const int seg = (tid + i * CONV_THREADS) & 3;
            int vy = a_y[i] + dy, vx = a_x[i] + dx;
            const bool ok = a_b[i] >= 0 && vy >= 0 && vy < IH && vx >= 0 && vx < IW;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (ok) {
                if (ups) {
                    vy = (2 * S.H == IH) ? (vy >> 1) : (int)(((int64_t)vy * S.H) / IH);
                    vx = (2 * S.W == IW) ? (vx >> 1) : (int)(((int64_t)vx * S.W) / IW);
                }
                const T* src = S.ptr + (int64_t)a_b[i] * S.sb + (int64_t)vy * S.sy + (int64_t)vx * S.sx + c0_ + seg * EPS;
                v = *reinterpret_cast<const uint4*>(src);
            }
            areg[i] = v;
        }
        const int64_t wk = (int64_t)cur_w * cin_total + srcoff_ + c0_;
#pragma unroll
        for (int i = 0; i < B_LOADS; ++i) {
            const int idx = tid + i * CONV_THREADS;
            const int row = idx >> 2, seg = idx & 3;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (row < BN && n0 + row < w_rows) {
                const T* w = reinterpret_cast<const T*>(p.weight) + (int64_t)(n0 + row) * p.w_taps * cin_total + wk + seg * EPS;
                v = *reinterpret_cast<const uint4*>(w);
            }
            breg[i] = v;
        }
        // advance the K walk (the next tap's offsets are requested here, a whole round before the gather that uses them)
        c0_ += KCV;
        if (c0_ >= S.C) {
            c0_ = 0;
            if (++t_ >= ntaps) {
                t_ = 0;
                srcoff_ += S.C;
                ++s_;
            }
            cur_dy = p.tap_dy[t_];
            cur_dx = p.tap_dx[t_];
            cur_w = p.tap_w[t_];
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_LOADS; ++i) {
            const int idx = tid + i * CONV_THREADS;
            *reinterpret_cast<uint4*>(Abuf(buf) + (idx >> 2) * ROWP + (idx & 3) * 16) = areg[i];
        }
#pragma unroll
        for (int i = 0; i < B_LOADS; ++i) {
            const int idx = tid + i * CONV_THREADS;
            if ((idx >> 2) < BN) *reinterpret_cast<uint4*>(Bbuf(buf) + (idx >> 2) * ROWP + (idx & 3) * 16) = breg[i];
        }
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[mt][nt][j] = 0.f;

    if (niter > 0) {
        gload();
        lstore(0);
    }
    __syncthreads();
    for (int it = 0; it < niter; ++it) {
        const int cur = it & 1;
        if (it + 1 < niter) gload();
        mma_tile<T, MT, NT, SWAP>(Abuf(cur), Bbuf(cur), wm * WTM, wn * WTN, lane, acc);
        if (it + 1 < niter) lstore(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue: v = act(acc + bias + addend) * act'(actout) ----
    const int r = lane & 31, h = lane >> 5;
    if (nz > 1) {
        // split-K partial: raw accumulators into ws[m][n] (f32 atomics: 32 lanes = 128 contiguous bytes per row)
        if constexpr (!SWAP) {
            float* ws = p.splitk_ws;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int n = n0 + wn * WTN + nt * 32 + r;
                if (n >= p.w_rows) continue;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        const int64_t m = m0 + wm * WTM + mt * 32 + (j & 3) + 8 * (j >> 2) + 4 * h;
                        if (m < M) atomicAdd(ws + m * p.w_rows + n, acc[mt][nt][j]);
                    }
            }
        }
        return;
    }
    if constexpr (!SWAP) {
        // outpix lives behind the A/B buffers; the dead A/B area becomes the per-wave staging slabs
        float* stage = reinterpret_cast<float*>(lds) + wave * (32 * (NT * 32 + 4));
        static_assert(4 * 32 * (NT * 32 + 4) * 4 <= 2 * (BM + BN) * ROWP, "staging must fit in the A/B buffers");
        const int cstride = p.out_cstride;
        epilogue_nhwc<T, MT, NT>(p, acc, stage, n0 + wn * WTN, lane, [&](int mt, int row) -> int64_t {
            const int o = outpix[wm * WTM + mt * 32 + row];
            return o < 0 ? (int64_t)-1 : (int64_t)o * cstride;
        });
    } else {
        // swapped operands: D rows = channels, D columns (lanes) = positions -> planar f32 stores of
        // 32 consecutive pixels per channel
        float* out = reinterpret_cast<float*>(p.out);
        const int64_t plane = (int64_t)p.OH * p.OW;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int o = outpix[wm * WTM + mt * 32 + r];
            if (o < 0) continue;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const int n = n0 + wn * WTN + nt * 32 + (j & 3) + 8 * (j >> 2) + 4 * h;
                    if (n >= p.Cout) continue;
                    float v = acc[mt][nt][j] + (p.bias ? p.bias[n] : 0.f);
                    v = apply_act(v, p.act);
                    out[(int64_t)o + (int64_t)n * plane] = v;
                }
        }
    }
}

template <typename T, int BN, bool SWAP>
__global__ __launch_bounds__(CONV_THREADS) void conv_igemm_kernel(const falnet_conv_t p) {
    conv_igemm_body<T, BN, SWAP>(p, blockIdx.z, gridDim.z);
}

// Up to four independent launches of the same shape family in ONE grid (blockIdx.z = member): the four output-parity
// classes of a stride-2 data gradient, which are otherwise four small latency-bound launches.
struct falnet_conv4_t { falnet_conv_t c[4]; };
template <typename T, int BN>
__global__ __launch_bounds__(CONV_THREADS) void conv_igemm_multi_kernel(const falnet_conv4_t pp, int ksplit) {
    // blockIdx.z = member * ksplit + K slice (ksplit > 1: f32 atomics into the MEMBER's own splitk_ws region, then
    // splitk_epilogue_multi_kernel)
    conv_igemm_body<T, BN, false>(pp.c[blockIdx.z / ksplit], blockIdx.z % ksplit, ksplit);
}

// split-K epilogue: ws[m][n] (f32 sums) -> bias / residual / activation / activation-gradient -> NHWC output
template <typename T>
__device__ __forceinline__ void splitk_epilogue_body(const falnet_conv_t& p) {
    // Reads the f32 sums the split-K workgroups accumulated with atomics, applies the epilogue, and ZEROES what it read:
    // the workspace is all-zero between launches (contract of falnet_conv_t::splitk_ws), so no memset dispatch is needed
    // in front of the next split-K launch (~40 per training step, each a ~5 us hole on the critical stream).
    const int segs = p.w_rows / 8;  // every column a workgroup may have touched, not only the Cout that are stored
    const int64_t M = (int64_t)p.B * p.TH * p.TW;
    const int64_t total = M * segs;
    const T* addend = reinterpret_cast<const T*>(p.addend);
    const T* actout = reinterpret_cast<const T*>(p.actout);
    T* out = reinterpret_cast<T*>(p.out);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int seg = (int)(i % segs);
        const int64_t m = i / segs;
        const int n = seg * 8;
        float* wsp = p.splitk_ws + m * p.w_rows + n;
        const float4 x0 = *reinterpret_cast<const float4*>(wsp);
        const float4 x1 = *reinterpret_cast<const float4*>(wsp + 4);
        *reinterpret_cast<float4*>(wsp) = make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(wsp + 4) = make_float4(0.f, 0.f, 0.f, 0.f);
        const int tx = (int)(m % p.TW), ty = (int)((m / p.TW) % p.TH), b = (int)(m / ((int64_t)p.TW * p.TH));
        const int oy = ty * p.osy + p.ooy, ox = tx * p.osx + p.oox;
        if (oy >= p.OH || ox >= p.OW || n >= p.Cout) continue;
        float v[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
        const int64_t off = (((int64_t)b * p.OH + oy) * p.OW + ox) * p.out_cstride + n;
        Vec8<T> t;
        if (addend) {
            t.load(addend + off);
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] += t.get(k);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = apply_act(v[k] + (p.bias ? p.bias[n + k] : 0.f), p.act);
        if (actout) {
            t.load(actout + off);
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] *= act_grad_from_out(t.get(k), p.actout_kind);
        }
        t.set8(v);
        t.store(out + off);
    }
}
template <typename T>
__global__ __launch_bounds__(256) void splitk_epilogue_kernel(const falnet_conv_t p) {
    splitk_epilogue_body<T>(p);
}
template <typename T>
__global__ __launch_bounds__(256) void splitk_epilogue_multi_kernel(const falnet_conv4_t pp) {  // blockIdx.y = member
    splitk_epilogue_body<T>(pp.c[blockIdx.y]);
}

// ------------------------------------------------------------------------------------------ 3x3 halo-patch kernel
// Dense 3x3 stride-1 convolutions (forward, stride-1 dgrad, fused upsample / concat included) on large images:
// a workgroup owns an 8x32 block of output positions (M = 256) x BN channels.  For every K chunk (KCB bytes of
// input channels of one source) the (8+2)x(32+2) input halo patch is staged in LDS ONCE and re-used by all 9
// taps (the generic gather kernel re-fetches it per tap); the packed weights stream through LDS per tap group.
//   mode P (TPS=1): one tap per barrier, double-buffered weight tile, next chunk's patch prefetched in 9 slices
//                   (one per tap) behind the MFMAs; used when there is enough K per tap (KCB = 128 B).
//   mode S (TPS=9): small-channel layers (C = 32): patch + all 9 taps' weights in one stage, several
//                   workgroups per CU overlap each other.
// One 32-position MFMA row tile = one image row of the block, so A-fragment rows are consecutive patch pixels
// (pitch KCB+16 B: conflict-free ds_read_b128, as in the gather kernel).
// MFMA over a compile-time sequence of NSTEP (tap, k-segment) steps with the fragment reads of step i+2 issued before
// the MFMAs of step i (three fragment sets in flight): the ~128-cycle ds_read latency is covered by two steps of
// MFMAs instead of being exposed in front of every small MFMA group, so one or two waves per SIMD keep the matrix
// pipe busy.  a_of(step, mt) / b_of(step, nt) return the 16-B fragment address of this lane.
struct NoHook { __device__ __forceinline__ void operator()(int) const {} };
#ifndef FALNET_PIN_S
#define FALNET_PIN_S 0        // nine-tap stages of the halo-patch kernel: 1 = pinned fragment prefetch
#define FALNET_PIN_SLOTS_S 3  // ... 2 = one step ahead
#define FALNET_PIN_WS 0       // weight-stationary kernel: 1 = pinned, two steps ahead
#endif

// Hook(step) runs after the MFMAs of every step, pinned in program order: work that must ISSUE while the matrix pipe is
// busy (global loads of the next block, the previous block's stores) instead of in front of / behind the whole sequence.
template <typename T, int MT, int NT, int NSTEP, bool SWAPAB = false, int SLOTS = 3, bool PIN = false, typename AOf, typename BOf, typename Hook = NoHook>
__device__ __forceinline__ void mma_steps(AOf a_of, BOf b_of, f32x16 (&acc)[MT][NT], Hook hook = Hook()) {
    // SLOTS fragment sets in flight (3: reads two steps ahead; 2: one step ahead, a third fewer registers).  PIN: keep that issue
    // order with sched_barriers -- hipcc otherwise sinks the reads back next to their MFMAs.  Same-box A/B: launched alone with
    // warm caches, pinned is 4-7 % faster in the nine-tap stages and in the weight-stationary kernel and 2-7 % SLOWER in the
    // one-tap-per-stage loop; inside the training step (pinned only where it won) the step time and the serial kernel sum do
    // not move (1124 vs 1124 pairs/s, two alternating runs) -- the chip holds ~1.57 GHz under this load and returns issue
    // savings as clock -- so the default stays unpinned.
    uint4 fa[SLOTS][MT], fb[SLOTS][NT];
    auto load = [&](int step, int slot) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) fa[slot][mt] = *reinterpret_cast<const uint4*>(a_of(step, mt));
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) fb[slot][nt] = *reinterpret_cast<const uint4*>(b_of(step, nt));
    };
    load(0, 0);
    if (SLOTS > 2 && NSTEP > 1) load(1, 1);
    if constexpr (PIN) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int st = 0; st < NSTEP; ++st) {
        if (st + SLOTS - 1 < NSTEP) load(st + SLOTS - 1, (st + SLOTS - 1) % SLOTS);
        if constexpr (PIN) __builtin_amdgcn_sched_barrier(0);
        const int sl = st % SLOTS;
        if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[mt][nt] = SWAPAB  // exchanged operands: pixels on lanes, channels in the accumulators (epilogue_direct)
                        ? H16<T>::mma(__builtin_bit_cast(s16x8_t, fb[sl][nt]), __builtin_bit_cast(s16x8_t, fa[sl][mt]), acc[mt][nt])
                        : H16<T>::mma(__builtin_bit_cast(s16x8_t, fa[sl][mt]), __builtin_bit_cast(s16x8_t, fb[sl][nt]), acc[mt][nt]);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[mt][nt] = SWAPAB
                            ? __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float((&fb[sl][nt].x)[e]), __uint_as_float((&fa[sl][mt].x)[e]), acc[mt][nt], 0, 0, 0)
                            : __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float((&fa[sl][mt].x)[e]), __uint_as_float((&fb[sl][nt].x)[e]), acc[mt][nt], 0, 0, 0);
        }
        if constexpr (!std::is_same<Hook, NoHook>::value) {
            __builtin_amdgcn_sched_barrier(0);
            hook(st);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// The same walk on v_mfma_f32_16x16x32 (16-bit operands, exchanged operands only: positions on lanes): one step = one (chunk, tap) with
// K = 32 (a whole 64-B chunk row), a 32x32 tile = 2 x 2 MFMAs (conv_epilogue.h: Acc16).  px_of(step, mt, pt) / w_of(step, nt, ct) return
// this lane's 16-B fragment address: position 16 pt + (lane & 15) / weight row m16_row_channel(lane & 15) of half ct, K block lane >> 4.
// CTN = 1: only the first 16-channel half of every 32-channel tile is computed (the other half stays zero) -- launches with at most 16 real output
// channels, e.g. the VGG data gradient into the 3-channel image, whose 32-wide tile is 29 channels of padding.
template <typename T, int MT, int NT, int NSTEP, int CTN = 2, typename PxOf, typename WOf, typename Hook = NoHook>
__device__ __forceinline__ void mma_steps16(PxOf px_of, WOf w_of, Acc16 (&acc)[MT][NT], Hook hook = Hook()) {
    constexpr int SLOTS = 3;
    uint4 fp[SLOTS][MT][2], fw[SLOTS][NT][CTN];
    auto load = [&](int step, int slot) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int pt = 0; pt < 2; ++pt) fp[slot][mt][pt] = *reinterpret_cast<const uint4*>(px_of(step, mt, pt));
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int ct = 0; ct < CTN; ++ct) fw[slot][nt][ct] = *reinterpret_cast<const uint4*>(w_of(step, nt, ct));
    };
    load(0, 0);
    if (NSTEP > 1) load(1, 1);
#pragma unroll
    for (int st = 0; st < NSTEP; ++st) {
        if (st + SLOTS - 1 < NSTEP) load(st + SLOTS - 1, (st + SLOTS - 1) % SLOTS);
        const int sl = st % SLOTS;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int ct = 0; ct < CTN; ++ct)
#pragma unroll
                    for (int pt = 0; pt < 2; ++pt)
                        acc[mt][nt].t[ct][pt] = H16<T>::mma16(__builtin_bit_cast(s16x8_t, fw[sl][nt][ct]), __builtin_bit_cast(s16x8_t, fp[sl][mt][pt]),
                                                              acc[mt][nt].t[ct][pt]);
        if constexpr (!std::is_same<Hook, NoHook>::value) {
            __builtin_amdgcn_sched_barrier(0);
            hook(st);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

#define PT_TH 8
#define PT_TW 32
#define PT_PW (PT_TW + 2)
#define PT_NPIX ((PT_TH + 2) * PT_PW)

template <typename T, int BN, int KCB, int TPS, bool ADB, int TH, int NWAVES>
__global__ __launch_bounds__(NWAVES * 64) void conv3x3_patch_kernel(const falnet_conv_t p, int tiles_x, int tiles_y, int flip) {
    // Taps are walked in spatial order t = (dy+1)*3 + (dx+1) with COMPILE-TIME offsets (the 9-tap loops are fully
    // unrolled, so every LDS address below is lane base + immediate); the packed-weight tap of spatial tap t is t
    // for a forward launch and 8-t for a stride-1 dgrad (flip) -- checked by falnet_conv2d.
    // TH x 32 output positions per workgroup (TH = 8: M = 256, 4 waves; TH = 16: M = 512, 8 waves = 2 per SIMD).
    // The bigger block halves the weight bytes loaded per MAC: these kernels sit on the CU load path (~8-10 B/clk/CU).
    constexpr int NTHR = NWAVES * 64;
    constexpr int NPIX = (TH + 2) * PT_PW;
    constexpr int PITCH = KCB + 16;
    constexpr int SEGS = KCB / 16;
    constexpr int KCV = KCB / (int)sizeof(T);
    constexpr int EPS = 16 / (int)sizeof(T);
    constexpr int WAVES_N = BN >= 128 ? 2 : 1, WAVES_M = NWAVES / WAVES_N;
    constexpr int MT = TH / WAVES_M;
    constexpr int WTN = BN / WAVES_N, NT = WTN / 32;
    constexpr bool PIPE = TPS < 9;  // mode P
    constexpr int A_BYTES = NPIX * PITCH, B_BYTES = TPS * BN * PITCH;
    constexpr int ROWL = PT_PW * SEGS;                                   // 16-B loads per patch row
    constexpr int A_SLOTS = (ROWL + NTHR - 1) / NTHR;    // per thread per patch row
    constexpr int BL = BN * SEGS;                                        // 16-B loads per weight tap tile
    constexpr int B_SLOTS = (BL + NTHR - 1) / NTHR;
    constexpr bool DBS = !PIPE && ADB;  // mode D: single-stage chunks, double-buffered (next chunk prefetched behind the MFMAs)
    __shared__ __attribute__((aligned(16))) char lds[(ADB ? 2 : 1) * A_BYTES + ((PIPE || DBS) ? 2 : 1) * B_BYTES];
    auto Abuf = [&](int b) -> char* { return lds + (ADB ? b : 0) * A_BYTES; };
    auto Bbuf = [&](int b) -> char* { return lds + (ADB ? 2 : 1) * A_BYTES + ((PIPE || DBS) ? b : 0) * B_BYTES; };

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int r = lane & 31, h = lane >> 5;
    int bid = blockIdx.x;
    const int tix = bid % tiles_x;
    bid /= tiles_x;
    const int tiy = bid % tiles_y;
    const int b = bid / tiles_y;
    const int ty0 = tiy * TH, tx0 = tix * PT_TW;
    const int n0 = blockIdx.y * BN;

    // source descriptors in registers: a runtime-indexed p.src[s] inside the K loop is a scalar LOAD from the kernel-argument
    // segment (plus its s_waitcnt) in front of every patch row -- in-kernel stamps: ~40 % of a tap was spent there
    struct SrcRegs { const T* ptrb; int64_t sy; int H, C; };
    const SrcRegs S0 = {reinterpret_cast<const T*>(p.src[0].ptr) + (int64_t)b * p.src[0].sb, p.src[0].sy, p.src[0].H, p.src[0].C};
    const SrcRegs S1 = p.nsrc > 1 ? SrcRegs{reinterpret_cast<const T*>(p.src[1].ptr) + (int64_t)b * p.src[1].sb, p.src[1].sy, p.src[1].H, p.src[1].C} : S0;
    const int nsrc = p.nsrc, IH = p.IH;
    int nchunks = S0.C / KCV + (nsrc > 1 ? S1.C / KCV : 0);

    // ---- loop-invariant per-thread load descriptors (no divisions / 64-bit multiplies in the main loop) ----
    // patch row slots: this thread's (column, segment) inside ANY patch row, per source (upsampling differs)
    int a_lds[A_SLOTS];            // byte offset inside a patch row in LDS, -1 = no slot
    int a_goff[A_SLOTS][2];        // element offset (px*sx + seg*EPS) per source, -1 = outside the image (zero fill)
#pragma unroll
    for (int u = 0; u < A_SLOTS; ++u) {
        const int idx = tid + u * NTHR;
        a_lds[u] = -1;
        a_goff[u][0] = a_goff[u][1] = -1;
        if (idx < ROWL) {
            const int pc = idx / SEGS, seg = idx % SEGS;
            a_lds[u] = pc * PITCH + seg * 16;
            const int vx = tx0 - 1 + pc;
            if (vx >= 0 && vx < p.IW) {
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    if (s < p.nsrc) {
                        const falnet_src_t& S = p.src[s];
                        int px = vx;
                        if (S.W != p.IW) px = (2 * S.W == p.IW) ? (vx >> 1) : (int)(((int64_t)vx * S.W) / p.IW);
                        a_goff[u][s] = px * (int)S.sx + seg * EPS;
                    }
                }
            }
        }
    }
    // weight tile slots: (row, segment) of one tap tile
    int b_lds[B_SLOTS];
    int64_t b_goff[B_SLOTS];
#pragma unroll
    for (int u = 0; u < B_SLOTS; ++u) {
        const int idx = tid + u * NTHR;
        b_lds[u] = -1;
        b_goff[u] = 0;
        if (idx < BL) {
            const int row = idx / SEGS, seg = idx % SEGS;
            if (n0 + row < p.w_rows) {
                b_lds[u] = row * PITCH + seg * 16;
                b_goff[u] = (int64_t)(n0 + row) * p.w_taps * p.cin_total + seg * EPS;
            } else {
                b_lds[u] = -2 - (row * PITCH + seg * 16);  // zero-fill slot
            }
        }
    }

    // one patch row `pr` of source s / channel offset c0 -> registers
    auto patch_row_load = [&](int pr, int s, int c0, uint4 (&regs)[A_SLOTS]) {
        const T* sptr = s == 0 ? S0.ptrb : S1.ptrb;
        const int64_t ssy = s == 0 ? S0.sy : S1.sy;
        const int sH = s == 0 ? S0.H : S1.H;
        int vy = ty0 - 1 + pr;
        const bool rowok = pr < TH + 2 && vy >= 0 && vy < IH;
        if (sH != IH) vy = (2 * sH == IH) ? (vy >> 1) : (int)(((int64_t)vy * sH) / IH);
        const T* base = sptr + (int64_t)vy * ssy + c0;
#pragma unroll
        for (int u = 0; u < A_SLOTS; ++u) {
            uint4 v = make_uint4(0, 0, 0, 0);
            const int go = a_goff[u][s];
            if (rowok && go >= 0) v = *reinterpret_cast<const uint4*>(base + go);
            regs[u] = v;
        }
    };
    auto patch_row_store = [&](char* A, int pr, const uint4 (&regs)[A_SLOTS]) {
        if (pr < TH + 2) {
#pragma unroll
            for (int u = 0; u < A_SLOTS; ++u)
                if (a_lds[u] >= 0) *reinterpret_cast<uint4*>(A + pr * (PT_PW * PITCH) + a_lds[u]) = regs[u];
        }
    };
    auto w_tile_load = [&](int tap, int kofs, uint4 (&regs)[B_SLOTS]) {
        const T* base = reinterpret_cast<const T*>(p.weight) + (flip ? 8 - tap : tap) * p.cin_total + kofs;
#pragma unroll
        for (int u = 0; u < B_SLOTS; ++u) {
            uint4 v = make_uint4(0, 0, 0, 0);
            if (b_lds[u] >= 0) v = *reinterpret_cast<const uint4*>(base + b_goff[u]);
            regs[u] = v;
        }
    };
    auto w_tile_store = [&](char* B, const uint4 (&regs)[B_SLOTS]) {
#pragma unroll
        for (int u = 0; u < B_SLOTS; ++u) {
            if (b_lds[u] >= 0) *reinterpret_cast<uint4*>(B + b_lds[u]) = regs[u];
            else if (b_lds[u] < -1) *reinterpret_cast<uint4*>(B + (-2 - b_lds[u])) = make_uint4(0, 0, 0, 0);
        }
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[mt][nt][j] = 0.f;

    // per-lane fragment bases; every other address component below is a compile-time constant after unrolling
    //   bf16: k-step ks of a tap reads the 16-B segment (2*ks + h);  f32: lane half h owns bytes [h*KCB/2, (h+1)*KCB/2)
    constexpr int KSEG = sizeof(T) == 2 ? KCB / 32 : KCB / 32;   // 16-B fragment segments per lane per tap (both dtypes: KCB/32)
    const int lane_k = sizeof(T) == 2 ? h * 16 : h * (KCB / 2);
    constexpr int KSTRIDE = sizeof(T) == 2 ? 32 : 16;            // byte distance between consecutive segments of one lane
    const int a_lane = ((wm * MT) * PT_PW + r) * PITCH + lane_k;  // patch pixel (first row of this wave, lane column) at tap (-1,-1)
    const int b_lane = (wn * WTN + r) * PITCH + lane_k;
    // NTAPS consecutive taps starting at compile-time spatial tap t0; weight tiles are laid out [tap][BN][PITCH] from Btile
    auto compute_taps = [&](const char* A, const char* Btile, auto t0c, auto ntapsc) {
        constexpr int t0 = decltype(t0c)::value, NTAPS = decltype(ntapsc)::value;
        const char* ab = A + a_lane;
        const char* bb = Btile + b_lane;
        mma_steps<T, MT, NT, NTAPS * KSEG, true, (NTAPS > 1 ? FALNET_PIN_SLOTS_S : 3), (NTAPS > 1 && FALNET_PIN_S)>(
            [&](int st, int mt) { const int t = t0 + st / KSEG, ks = st % KSEG;
                                  return ab + (mt * PT_PW * PITCH + ((t / 3) * PT_PW + (t % 3)) * PITCH + ks * KSTRIDE); },
            [&](int st, int nt) { const int tt = st / KSEG, ks = st % KSEG;
                                  return bb + (tt * BN * PITCH + nt * 32 * PITCH + ks * KSTRIDE); },
            acc);
    };

    // K walk over chunks: (source, channel offset, packed-weight offset)
    int s_ = 0, c0_ = 0, kofs_ = 0;
    auto advance = [&](int& s, int& c0, int& kofs) {
        c0 += KCV;
        kofs += KCV;
        if (s < 2 && c0 >= (s == 0 ? S0.C : S1.C)) {
            c0 = 0;
            s = s + 1 < nsrc ? s + 1 : s;
        }
    };
    // all (8+2) rows are requested before the first one is stored: ONE memory round trip, not ten
    auto load_whole_patch = [&](char* A, int s, int c0) {
        uint4 regs[TH + 2][A_SLOTS];
#pragma unroll
        for (int pr = 0; pr < TH + 2; ++pr) patch_row_load(pr, s, c0, regs[pr]);
#pragma unroll
        for (int pr = 0; pr < TH + 2; ++pr) patch_row_store(A, pr, regs[pr]);
    };

    if constexpr (DBS) {
        // mode D: chunk c is computed from LDS buffer c&1 while ALL of chunk c+1 (patch rows + nine weight tiles) is in
        // flight into registers; one barrier per chunk
        {
            uint4 wregs[9][B_SLOTS];
#pragma unroll
            for (int tt = 0; tt < 9; ++tt) w_tile_load(tt, kofs_, wregs[tt]);
            load_whole_patch(Abuf(0), s_, c0_);
#pragma unroll
            for (int tt = 0; tt < 9; ++tt) w_tile_store(Bbuf(0) + tt * BN * PITCH, wregs[tt]);
        }
        __syncthreads();
        for (int c = 0; c < nchunks; ++c) {
            const bool next_chunk = c + 1 < nchunks;
            int sn = s_, c0n = c0_, kofsn = kofs_;
            advance(sn, c0n, kofsn);
            uint4 wregs[9][B_SLOTS], pregs[TH + 2][A_SLOTS];
            if (next_chunk) {
#pragma unroll
                for (int tt = 0; tt < 9; ++tt) w_tile_load(tt, kofsn, wregs[tt]);
#pragma unroll
                for (int pr = 0; pr < TH + 2; ++pr) patch_row_load(pr, sn, c0n, pregs[pr]);
            }
            compute_taps(Abuf(c & 1), Bbuf(c & 1), std::integral_constant<int, 0>{}, std::integral_constant<int, 9>{});
            if (next_chunk) {
#pragma unroll
                for (int tt = 0; tt < 9; ++tt) w_tile_store(Bbuf((c + 1) & 1) + tt * BN * PITCH, wregs[tt]);
#pragma unroll
                for (int pr = 0; pr < TH + 2; ++pr) patch_row_store(Abuf((c + 1) & 1), pr, pregs[pr]);
            }
            __syncthreads();
            s_ = sn;
            c0_ = c0n;
            kofs_ = kofsn;
        }
    } else if constexpr (!PIPE) {
        for (int c = 0; c < nchunks; ++c) {
            if (c > 0) __syncthreads();
            {
                uint4 wregs[9][B_SLOTS];
#pragma unroll
                for (int tt = 0; tt < 9; ++tt) w_tile_load(tt, kofs_, wregs[tt]);
                load_whole_patch(Abuf(0), s_, c0_);
#pragma unroll
                for (int tt = 0; tt < 9; ++tt) w_tile_store(Bbuf(0) + tt * BN * PITCH, wregs[tt]);
            }
            __syncthreads();
            compute_taps(Abuf(0), Bbuf(0), std::integral_constant<int, 0>{}, std::integral_constant<int, 9>{});
            advance(s_, c0_, kofs_);
        }
    } else {
        // mode P.  In-kernel phase stamps (tools/p_stamps.py, 256->256 @64x128, one workgroup of 8 waves per CU): per tap ~40 % of
        // the time passes while the waves sit in the ISSUE of this tap's 1-3 global loads (12 KB per workgroup through the CU's
        // load path), ~40 % in fragment reads + MFMAs, 2 % waiting for the loads at the store, the rest in LDS stores and the
        // barrier.  Tried and measured slower or equal: requesting tiles two taps ahead (hand-counted vmcnt, +25 % time: the
        // extra loads queue in issue, not in flight), pinning all fragment reads in front of the MFMAs (+2..7 %), hoisting the
        // source descriptors out of the loop (0 %; kept), letting only the second wave of every SIMD load (0 %).  s_memtime ticks
        // over the launch time give an in-kernel clock of ~1.57 GHz: the matrix pipe is busy ~42 % of a tap at THAT clock.
        // prologue: patch of chunk 0 and the weight tile of (chunk 0, tap 0), all loads in flight together
        {
            uint4 regs[B_SLOTS];
            w_tile_load(0, kofs_, regs);
            load_whole_patch(Abuf(0), s_, c0_);
            w_tile_store(Bbuf(0), regs);
        }
        __syncthreads();
        int sn = s_, c0n = c0_, kofsn = kofs_;  // next chunk
        advance(sn, c0n, kofsn);
#ifdef FALNET_P_STAMPS
        // profiling build (tools/p_stamps.py): per-phase s_memtime sums of workgroup 0, every wave -> p.splitk_ws
        unsigned int ph_sum[5] = {0, 0, 0, 0, 0};
        unsigned long long t_prev;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_prev)::"memory");
#define P_STAMP(k)                                                                      \
    do {                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                              \
        unsigned long long t_;                                                          \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");      \
        __builtin_amdgcn_sched_barrier(0);                                              \
        ph_sum[k] += (unsigned int)(t_ - t_prev);                                       \
        t_prev = t_;                                                                    \
    } while (0)
#else
#define P_STAMP(k) do {} while (0)
#endif
        for (int c = 0; c < nchunks; ++c) {
            const bool next_chunk = ADB && (c + 1 < nchunks);
            const char* Acur = Abuf(c & 1);
            char* Anext = Abuf((c + 1) & 1);
            static_for<0, 9>([&](auto gc) {
                constexpr int g = decltype(gc)::value;
                const bool more = g < 8 || c + 1 < nchunks;
                uint4 breg[B_SLOTS], areg0[A_SLOTS], areg1[A_SLOTS];
                if (more) w_tile_load(g == 8 ? 0 : g + 1, g == 8 ? kofsn : kofs_, breg);
                constexpr int RPT = (TH + 2 + 8) / 9;  // patch rows prefetched per tap (2: tap g brings rows g and g+9)
                if (next_chunk) {
                    patch_row_load(g, sn, c0n, areg0);
                    if (RPT > 1 && g + 9 < TH + 2) patch_row_load(g + 9, sn, c0n, areg1);
                }
                P_STAMP(0);  // loads issued
                compute_taps(Acur, Bbuf((c + g) & 1), std::integral_constant<int, g>{}, std::integral_constant<int, 1>{});
                P_STAMP(1);  // fragment reads + MFMAs issued
#ifdef FALNET_P_STAMPS
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                P_STAMP(2);  // this tap's loads landed
#endif
                if (more) w_tile_store(Bbuf((c + g + 1) & 1), breg);
                if (next_chunk) {
                    patch_row_store(Anext, g, areg0);
                    if (RPT > 1 && g + 9 < TH + 2) patch_row_store(Anext, g + 9, areg1);
                }
                P_STAMP(3);  // LDS stores issued
                __syncthreads();
                P_STAMP(4);  // barrier
            });
            s_ = sn;
            c0_ = c0n;
            kofs_ = kofsn;
            advance(sn, c0n, kofsn);
        }
#ifdef FALNET_P_STAMPS
        if (p.splitk_ws && blockIdx.x == 0 && blockIdx.y == 0 && lane == 0) {
            for (int k = 0; k < 5; ++k) p.splitk_ws[wave * 8 + k] = (float)ph_sum[k];
            p.splitk_ws[wave * 8 + 5] = (float)(9 * nchunks);
        }
#endif
#undef P_STAMP
    }

    // ---- epilogue (same contract as the gather kernel), straight from the accumulators: 16-B stores, no LDS ----
    {
        float bias[NT][16];
        load_bias16<NT>(p, n0 + wn * WTN, h, bias);
        const int cstride = p.out_cstride;
        const int x = tx0 + r;
        const bool planar_out = p.out_layout == FALNET_OUT_PLANAR_F32;
        auto pixoff = [&](int mt) -> int64_t {
            const int y = ty0 + wm * MT + mt;
            if (!(y < p.OH && x < p.OW)) return (int64_t)-1;
            return planar_out ? ((int64_t)b * p.Cout * p.OH + y) * p.OW + x : (((int64_t)b * p.OH + y) * p.OW + x) * cstride;
        };
        if constexpr (MT % 2 == 0) {
            // fused 2x2 reduction (p.pool_out): block rows start even and every wave owns an even number of rows
            auto pooloff = [&](int mt) -> int64_t {
                const int py = (ty0 + wm * MT + mt) >> 1, px = x >> 1, PH = p.OH >> 1, PW = p.OW >> 1;
                return (py < PH && px < PW) ? (((int64_t)b * PH + py) * PW + px) * cstride : (int64_t)-1;
            };
            epilogue_direct<T, MT, NT>(p, acc, bias, n0 + wn * WTN, lane, pixoff, pooloff);
        } else {
            epilogue_direct<T, MT, NT>(p, acc, bias, n0 + wn * WTN, lane, pixoff);
        }
    }
}

// ------------------------------------------------------------------------------------------ weight-stationary 3x3
// The halo-patch kernels above re-stage the nine weight tap tiles for every 8x32-position block: with <= 64 input channels
// that is ~2/3 of the bytes a workgroup pulls through its CU's load path (~10 B/clk), which is what bounds them
// (PMC: DESIGN.md).  For these layers (the full-resolution ones: VGG conv1_2/2_1, iconv1, deconv1, the level-0/1 residual
// blocks and their data gradients) ALL weights of a BN-channel slice fit in LDS next to one patch, so a persistent
// workgroup loads them ONCE and then streams blocks: per block only the (8+2)x(32+2) input patch crosses the load path,
// prefetched into registers behind the previous block's MFMAs.  K chunks are 64 B (32 bf16 / 16 f32 channels), NCH <= 2.
//   LDS: W [NCH][9][BN][80 B] + A [NCH][340][80 B]  (BN=64, NCH=2: 92 KB + 54 KB)
//   4 waves, each 2 rows x BN channels of the block (MT = 2, NT = BN/32); epilogue straight from the accumulators.
template <typename T, int BN, int NCH, bool M16 = false>
__global__ __launch_bounds__(512) void conv3x3_ws_kernel(const falnet_conv_t p, int tiles_x, int tiles_y, int flip) {
    // eight waves = two per SIMD: while one wave sits in a load / store issue or in its epilogue arithmetic the other one
    // keeps the matrix pipe fed (in-kernel stamps at one wave per SIMD: MFMA 36 % of a block, load issue 25 %, epilogue 32 %).
    //   BN = 64: 8x32 positions, waves 4 (rows) x 2 (32-channel halves);  BN = 32: 16x32 positions, waves 8 x 1.
    constexpr int NTHR = 512, WAVES_N = BN / 32, WAVES_M = 8 / WAVES_N;
    constexpr int MT = 2, NT = 1, TH = MT * WAVES_M;
    constexpr int KCB = 64, PITCH = KCB + 16, SEGS = KCB / 16;
    constexpr int NPIX = (TH + 2) * PT_PW;
    constexpr int KCV = KCB / (int)sizeof(T), EPS = 16 / (int)sizeof(T);
    constexpr int W_BYTES = NCH * 9 * BN * PITCH, A_BYTES = NCH * NPIX * PITCH;
    constexpr int W_LOADS = NCH * 9 * BN * SEGS, W_SLOTS = (W_LOADS + NTHR - 1) / NTHR;
    static_assert(W_BYTES + A_BYTES <= 160 * 1024, "weights + one patch must fit in LDS");
    __shared__ __attribute__((aligned(16))) char lds[W_BYTES + A_BYTES];
    char* Wl = lds;
    char* Al = lds + W_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int n0 = blockIdx.y * BN;

    // ---- weights: once per workgroup.  LDS tile (chunk c, spatial tap t) <- packed tap (flip ? 8-t : t), channels [c*KCV, +KCV)
    {
        uint4 wreg[W_SLOTS];
#pragma unroll
        for (int u = 0; u < W_SLOTS; ++u) {
            const int idx = tid + u * NTHR;
            wreg[u] = make_uint4(0, 0, 0, 0);
            if (idx < W_LOADS) {
                const int seg = idx % SEGS, row = (idx / SEGS) % BN, t = (idx / (SEGS * BN)) % 9, c = idx / (SEGS * BN * 9);
                if (n0 + row < p.w_rows && c * KCV < p.cin_total)
                    wreg[u] = *reinterpret_cast<const uint4*>(reinterpret_cast<const T*>(p.weight) +
                                                              ((int64_t)(n0 + row) * p.w_taps + (flip ? 8 - t : t)) * p.cin_total + c * KCV + seg * EPS);
            }
        }
#pragma unroll
        for (int u = 0; u < W_SLOTS; ++u) {
            const int idx = tid + u * NTHR;
            if (idx < W_LOADS) {
                const int seg = idx % SEGS, row = (idx / SEGS) % BN, tc = idx / (SEGS * BN);  // tc = c*9 + t
                *reinterpret_cast<uint4*>(Wl + (tc * BN + row) * PITCH + seg * 16) = wreg[u];
            }
        }
    }
    // ---- patch slots of this thread, per chunk: (patch pixel, 16-B segment).  Chunk c = channels [c*KCV, +KCV) of the
    // concatenated sources; its source and everything read from the descriptor are workgroup-uniform scalars hoisted here
    // (a per-lane source index would turn every descriptor field into a vector memory load inside the block loop).
    constexpr int C_LOADS = NPIX * SEGS, C_SLOTS = (C_LOADS + NTHR - 1) / NTHR;
    const T* c_ptr[NCH];
    int64_t c_sb[NCH], c_sy[NCH];
    int c_sx[NCH], c_hs[NCH], c_ws[NCH];
    bool c_ok[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        int sidx = 0, c0 = c * KCV;
        if (c0 >= p.src[0].C) {
            c0 -= p.src[0].C;
            sidx = 1;
        }
        c_ok[c] = sidx < p.nsrc && c0 < (sidx ? p.src[1].C : p.src[0].C);
        c_ptr[c] = reinterpret_cast<const T*>(sidx ? p.src[1].ptr : p.src[0].ptr) + c0;
        c_sb[c] = sidx ? p.src[1].sb : p.src[0].sb;
        c_sy[c] = sidx ? p.src[1].sy : p.src[0].sy;
        c_sx[c] = (int)(sidx ? p.src[1].sx : p.src[0].sx);
        c_hs[c] = (sidx ? p.src[1].H : p.src[0].H) != p.IH ? 1 : 0;  // exact 2x nearest upsampling (checked by the dispatcher)
        c_ws[c] = (sidx ? p.src[1].W : p.src[0].W) != p.IW ? 1 : 0;
    }
    int a_lds[C_SLOTS], a_pp[C_SLOTS];  // LDS byte offset inside a chunk's patch (-1: no slot); (seg*EPS << 16 | pr << 8 | pc)
#pragma unroll
    for (int u = 0; u < C_SLOTS; ++u) {
        const int idx = tid + u * NTHR;
        a_lds[u] = -1;
        a_pp[u] = 0;
        if (idx < C_LOADS) {
            const int seg = idx % SEGS, px = idx / SEGS;
            a_lds[u] = px * PITCH + seg * 16;
            a_pp[u] = ((seg * EPS) << 16) | ((px / PT_PW) << 8) | (px % PT_PW);
        }
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c)
        if (!c_ok[c]) {  // chunk beyond the sources: zero once
#pragma unroll
            for (int u = 0; u < C_SLOTS; ++u)
                if (a_lds[u] >= 0) *reinterpret_cast<uint4*>(Al + c * NPIX * PITCH + a_lds[u]) = make_uint4(0, 0, 0, 0);
        }

    const int ntiles = p.B * tiles_y * tiles_x;
    uint4 areg[NCH][C_SLOTS];
    auto patch_load = [&](int tile) {
        const int tix = tile % tiles_x, tiy = (tile / tiles_x) % tiles_y, b = tile / (tiles_x * tiles_y);
        const int ty0 = tiy * TH, tx0 = tix * PT_TW;
#pragma unroll
        for (int u = 0; u < C_SLOTS; ++u) {
            const int pp = a_pp[u];
            const int vy = ty0 - 1 + ((pp >> 8) & 0xff), vx = tx0 - 1 + (pp & 0xff);
            const bool ok = a_lds[u] >= 0 && vy >= 0 && vy < p.IH && vx >= 0 && vx < p.IW;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                uint4 v = make_uint4(0, 0, 0, 0);
                if (ok && c_ok[c])
                    v = *reinterpret_cast<const uint4*>(c_ptr[c] + (int64_t)b * c_sb[c] + (int64_t)(vy >> c_hs[c]) * c_sy[c] + (vx >> c_ws[c]) * c_sx[c] + (pp >> 16));
                areg[c][u] = v;
            }
        }
    };
    auto patch_store = [&]() {
#pragma unroll
        for (int c = 0; c < NCH; ++c)
            if (c_ok[c]) {
#pragma unroll
                for (int u = 0; u < C_SLOTS; ++u)
                    if (a_lds[u] >= 0) *reinterpret_cast<uint4*>(Al + c * NPIX * PITCH + a_lds[u]) = areg[c][u];
            }
    };

    constexpr int KSEG = KCB / 32;
    const int lane_k = sizeof(T) == 2 ? h * 16 : h * (KCB / 2);
    constexpr int KSTRIDE = sizeof(T) == 2 ? 32 : 16;
    const char* ab = Al + ((wm * MT) * PT_PW + r) * PITCH + lane_k;
    const char* bb = Wl + (wn * 32 + r) * PITCH + lane_k;

    float bias[NT][16];
    load_bias16<NT>(p, n0 + wn * 32, h, bias);
    int tile = blockIdx.x;
    if (tile < ntiles) patch_load(tile);
#ifdef FALNET_WS_STAMPS
    // profiling build (tools/ws_stamps.py): per-phase s_memtime stamps of workgroup 0, every wave -> p.splitk_ws
    unsigned long long* stamp_out = reinterpret_cast<unsigned long long*>(p.splitk_ws);
    int stamp_i = 0;
#define WS_STAMP()                                                                                       \
    do {                                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                               \
        unsigned long long t_;                                                                           \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                        \
        __builtin_amdgcn_sched_barrier(0);                                                               \
        if (stamp_out && blockIdx.x == 0 && blockIdx.y == 0 && lane == 0 && wave < 4 && stamp_i < 256) stamp_out[(wave & 3) * 256 + stamp_i] = t_; \
        ++stamp_i;                                                                                       \
    } while (0)
#else
#define WS_STAMP() do {} while (0)
#endif
    for (; tile < ntiles; tile += gridDim.x) {
        WS_STAMP();  // 0: loop top
        __syncthreads();  // every wave has finished reading the previous patch (first pass: the weight / zero stores are ordered below)
        WS_STAMP();  // 1: after barrier A
        patch_store();
        WS_STAMP();  // 2: after the patch stores (includes the wait for the prefetched loads)
        __syncthreads();
        WS_STAMP();  // 3: after barrier B
        const int next = tile + gridDim.x;
        if (next < ntiles) patch_load(next);  // in flight behind this block's MFMAs and epilogue
        WS_STAMP();  // 4: prefetch issued
        f32x16 acc[MT][NT];
        if constexpr (M16 && sizeof(T) == 2) {
            // v_mfma_f32_16x16x32 form (steps = (chunk, tap), K = 32): same LDS image, lane = (position / weight row lane & 15, K block lane >> 4)
            Acc16 a16[MT][NT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) a16[mt][nt].zero();
            const int lp = lane & 15, lg = lane >> 4;
            const char* pxb = Al + ((wm * MT) * PT_PW + lp) * PITCH + lg * 16;
            const char* wb16 = Wl + (wn * 32 + m16_row_channel(lp)) * PITCH + lg * 16;
            auto px_of = [&](int st, int mt, int pt) { const int c = st / 9, t = st % 9;
                                                       return pxb + ((c * NPIX + mt * PT_PW + 16 * pt + (t / 3) * PT_PW + (t % 3)) * PITCH); };
            auto w_of = [&](int st, int nt, int ct) { return wb16 + ((st * BN + nt * 32 + 16 * ct) * PITCH); };
            bool half_only = false;
            if constexpr (BN == 32) half_only = p.Cout <= 16;  // (workgroup-uniform) the second 16-channel half is padding: half the MFMAs
            if (half_only) mma_steps16<T, MT, NT, NCH * 9, 1>(px_of, w_of, a16);
            else mma_steps16<T, MT, NT, NCH * 9>(px_of, w_of, a16);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) a16[mt][nt].to32(acc[mt][nt]);
        } else {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int j = 0; j < 16; ++j) acc[mt][nt][j] = 0.f;
        // steps: (chunk, tap, k-segment), all operand addresses = lane base + immediate
        mma_steps<T, MT, NT, NCH * 9 * KSEG, true, 3, FALNET_PIN_WS>(
            [&](int st, int mt) { const int c = st / (9 * KSEG), t = (st / KSEG) % 9, ks = st % KSEG;
                                  return ab + ((c * NPIX + mt * PT_PW + (t / 3) * PT_PW + (t % 3)) * PITCH + ks * KSTRIDE); },
            [&](int st, int nt) { const int tc = st / KSEG, ks = st % KSEG;
                                  return bb + ((tc * BN + nt * 32) * PITCH + ks * KSTRIDE); },
            acc);
        }
        WS_STAMP();  // 5: MFMAs done (issued)
        const int tix = tile % tiles_x, tiy = (tile / tiles_x) % tiles_y, b = tile / (tiles_x * tiles_y);
        const int ty0 = tiy * TH, x = tix * PT_TW + r;
        const int cstride = p.out_cstride;
        const bool planar_out = p.out_layout == FALNET_OUT_PLANAR_F32;
        auto pixoff = [&](int mt) -> int64_t {
            const int y = ty0 + wm * MT + mt;
            if (!(y < p.OH && x < p.OW)) return (int64_t)-1;
            return planar_out ? ((int64_t)b * p.Cout * p.OH + y) * p.OW + x : (((int64_t)b * p.OH + y) * p.OW + x) * cstride;
        };
        auto pooloff = [&](int mt) -> int64_t {
            const int py = (ty0 + wm * MT + mt) >> 1, px = x >> 1, PH = p.OH >> 1, PW = p.OW >> 1;
            return (py < PH && px < PW) ? (((int64_t)b * PH + py) * PW + px) * cstride : (int64_t)-1;
        };
        epilogue_direct<T, MT, NT>(p, acc, bias, n0 + wn * 32, lane, pixoff, pooloff);
        WS_STAMP();  // 6: epilogue done
    }
#undef WS_STAMP
}

// ------------------------------------------------------------------------------------------ weight-stationary 3x3, two-phase
// conv3x3_ws_kernel's eight waves walk barrier -> refill -> MFMA -> epilogue in lock step (one patch buffer): while they all sit in the
// epilogue arithmetic (ELU: ~11 % of the launch) and in the load / store issue, the matrix pipe idles (in-kernel stamps).
// Here the eight waves are TWO groups of four (one wave per SIMD each) with a patch buffer per group and HALF-height tiles, and the groups
// run half a period apart: between two workgroup barriers group A issues the MFMAs of its tile while group B -- the OTHER wave of every
// SIMD -- does the epilogue of its previous tile, loads its next patch and stores it to LDS; at the next barrier the roles swap.
//   LDS: W [NCH][9][BN][80 B] + 2 x A [NCH][(TH+2) x 34][80 B]   (BN = 64, NCH = 2: TH = 4, 92 KB + 2 x 32 KB)
// Tiles j = 0, 1, ... of a persistent workgroup go to group j & 1.
template <typename T, int BN, int NCH, bool M16 = false, bool FULL = false>
__global__ __launch_bounds__(512) void conv3x3_ws2_kernel(const falnet_conv_t p, int tiles_x, int tiles_y, int flip) {
    constexpr int NTHR = 512, GTHR = 256, WAVES_N = BN / 32, WAVES_M = 4 / WAVES_N;
    constexpr int MT = 2, NT = 1, TH = MT * WAVES_M;
    constexpr int KCB = 64, PITCH = KCB + 16, SEGS = KCB / 16;
    constexpr int NPIX = (TH + 2) * PT_PW;
    constexpr int KCV = KCB / (int)sizeof(T), EPS = 16 / (int)sizeof(T);
    constexpr int W_BYTES = NCH * 9 * BN * PITCH, A_BYTES = NCH * NPIX * PITCH;
    constexpr int W_LOADS = NCH * 9 * BN * SEGS, W_SLOTS = (W_LOADS + NTHR - 1) / NTHR;
    static_assert(W_BYTES + 2 * A_BYTES <= 160 * 1024, "weights + two patches must fit in LDS");
    __shared__ __attribute__((aligned(16))) char lds[W_BYTES + 2 * A_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int grp = wave >> 2, gtid = tid & (GTHR - 1), gw = wave & 3;  // waves w and w + 4 share a SIMD and belong to different groups
    char* Wl = lds;
    char* Al = lds + W_BYTES + grp * A_BYTES;
    const int r = lane & 31, h = lane >> 5;
    const int wm = gw / WAVES_N, wn = gw % WAVES_N;
    const int n0 = blockIdx.y * BN;

    {  // weights: once per workgroup, all 512 threads (as conv3x3_ws_kernel)
        uint4 wreg[W_SLOTS];
#pragma unroll
        for (int u = 0; u < W_SLOTS; ++u) {
            const int idx = tid + u * NTHR;
            wreg[u] = make_uint4(0, 0, 0, 0);
            if (idx < W_LOADS) {
                const int seg = idx % SEGS, row = (idx / SEGS) % BN, t = (idx / (SEGS * BN)) % 9, c = idx / (SEGS * BN * 9);
                if (n0 + row < p.w_rows && c * KCV < p.cin_total)
                    wreg[u] = *reinterpret_cast<const uint4*>(reinterpret_cast<const T*>(p.weight) +
                                                              ((int64_t)(n0 + row) * p.w_taps + (flip ? 8 - t : t)) * p.cin_total + c * KCV + seg * EPS);
            }
        }
#pragma unroll
        for (int u = 0; u < W_SLOTS; ++u) {
            const int idx = tid + u * NTHR;
            if (idx < W_LOADS) {
                const int seg = idx % SEGS, row = (idx / SEGS) % BN, tc = idx / (SEGS * BN);
                *reinterpret_cast<uint4*>(Wl + (tc * BN + row) * PITCH + seg * 16) = wreg[u];
            }
        }
    }
    // FULL (both chunks from ONE source of exactly 2 x 64 B per pixel): one load stream whose consecutive eight lanes cover a pixel's whole
    // 128-B line, instead of one stream per chunk reading half lines
    constexpr int LS = FULL ? 1 : NCH, LSEGS = FULL ? SEGS * NCH : SEGS;
    static_assert(!FULL || NCH == 2, "FULL: two chunks");
    constexpr int C_LOADS = NPIX * LSEGS, C_SLOTS = (C_LOADS + GTHR - 1) / GTHR;
    const T* c_ptr[NCH];
    int64_t c_sb[NCH], c_sy[NCH];
    int c_sx[NCH], c_hs[NCH], c_ws[NCH];
    bool c_ok[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {  // workgroup-uniform source of chunk c (hoisted scalars, see conv3x3_ws_kernel)
        int sidx = 0, c0 = c * KCV;
        if (c0 >= p.src[0].C) {
            c0 -= p.src[0].C;
            sidx = 1;
        }
        c_ok[c] = sidx < p.nsrc && c0 < (sidx ? p.src[1].C : p.src[0].C);
        c_ptr[c] = reinterpret_cast<const T*>(sidx ? p.src[1].ptr : p.src[0].ptr) + c0;
        c_sb[c] = sidx ? p.src[1].sb : p.src[0].sb;
        c_sy[c] = sidx ? p.src[1].sy : p.src[0].sy;
        c_sx[c] = (int)(sidx ? p.src[1].sx : p.src[0].sx);
        c_hs[c] = (sidx ? p.src[1].H : p.src[0].H) != p.IH ? 1 : 0;
        c_ws[c] = (sidx ? p.src[1].W : p.src[0].W) != p.IW ? 1 : 0;
    }
    int a_lds[C_SLOTS], a_pp[C_SLOTS];
#pragma unroll
    for (int u = 0; u < C_SLOTS; ++u) {
        const int idx = gtid + u * GTHR;
        a_lds[u] = -1;
        a_pp[u] = 0;
        if (idx < C_LOADS) {
            const int seg = idx % LSEGS, px = idx / LSEGS;
            a_lds[u] = (seg / SEGS) * NPIX * PITCH + px * PITCH + (seg % SEGS) * 16;
            a_pp[u] = ((seg * EPS) << 16) | ((px / PT_PW) << 8) | (px % PT_PW);
        }
    }
#pragma unroll
    for (int c = 0; c < LS; ++c)
        if (!FULL && !c_ok[c]) {
#pragma unroll
            for (int u = 0; u < C_SLOTS; ++u)
                if (a_lds[u] >= 0) *reinterpret_cast<uint4*>(Al + c * NPIX * PITCH + a_lds[u]) = make_uint4(0, 0, 0, 0);
        }

    const int ntiles = p.B * tiles_y * tiles_x;
    uint4 areg[LS][C_SLOTS];
    int pl_b = 0, pl_ty0 = 0, pl_tx0 = 0;
    auto patch_target = [&](int tile) {
        const int tix = tile % tiles_x, tiy = (tile / tiles_x) % tiles_y;
        pl_b = tile / (tiles_x * tiles_y);
        pl_ty0 = tiy * TH;
        pl_tx0 = tix * PT_TW;
    };
    auto patch_load_slot = [&](int c, int u) {  // one 16-B load of the tile set by patch_target
        const int pp = a_pp[u];
        const int vy = pl_ty0 - 1 + ((pp >> 8) & 0xff), vx = pl_tx0 - 1 + (pp & 0xff);
        const bool ok = a_lds[u] >= 0 && vy >= 0 && vy < p.IH && vx >= 0 && vx < p.IW;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (ok && c_ok[c])
            v = *reinterpret_cast<const uint4*>(c_ptr[c] + (int64_t)pl_b * c_sb[c] + (int64_t)(vy >> c_hs[c]) * c_sy[c] + (vx >> c_ws[c]) * c_sx[c] + (pp >> 16));
        areg[c][u] = v;
    };
    // the slots of a patch in two halves: the first is loaded by the wave in its MATRIX role (in front of the MFMAs), the second in its
    // store role (in front of the epilogue) -- a wave is held ~290 cycles in the issue of every 1-KB load by the CU's memory path, so the
    // loads are split to balance the two roles (stamps: matrix 3.9 k, epilogue 3.4 k, load issue 2.3 k, LDS stores 0.8 k cycles per tile)
    constexpr int C_HALF = (C_SLOTS + 1) / 2;
    auto patch_load_half = [&](auto half) {
        constexpr int H = decltype(half)::value;
#pragma unroll
        for (int u = H ? C_HALF : 0; u < (H ? C_SLOTS : C_HALF); ++u)
#pragma unroll
            for (int c = 0; c < LS; ++c) patch_load_slot(c, u);
    };
    auto patch_store_half = [&](auto half) {
        constexpr int H = decltype(half)::value;
#pragma unroll
        for (int c = 0; c < LS; ++c)
            if (c_ok[c]) {
#pragma unroll
                for (int u = H ? C_HALF : 0; u < (H ? C_SLOTS : C_HALF); ++u)
                    if (a_lds[u] >= 0) *reinterpret_cast<uint4*>(Al + c * NPIX * PITCH + a_lds[u]) = areg[c][u];
            }
    };
    using Half0 = std::integral_constant<int, 0>;
    using Half1 = std::integral_constant<int, 1>;

    constexpr int KSEG = KCB / 32;
    const int lane_k = sizeof(T) == 2 ? h * 16 : h * (KCB / 2);
    constexpr int KSTRIDE = sizeof(T) == 2 ? 32 : 16;
    const char* ab = Al + ((wm * MT) * PT_PW + r) * PITCH + lane_k;
    const char* bb = Wl + (wn * 32 + r) * PITCH + lane_k;
    float bias[NT][16];
    load_bias16<NT>(p, n0 + wn * 32, h, bias);

    auto tile_xy = [&](int tile, int& b, int& ty0, int& x) {
        const int tix = tile % tiles_x, tiy = (tile / tiles_x) % tiles_y;
        b = tile / (tiles_x * tiles_y);
        ty0 = tiy * TH;
        x = tix * PT_TW + r;
    };
    // tiles of this workgroup: blockIdx.x + j * gridDim.x; group grp takes j = grp, grp + 2, ...
    const int nj = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int n_a = (nj + 1) >> 1, n_b = nj >> 1;
    const int nph = max(2 * n_a + 1, 2 * n_b + 2);  // group 0: refill at even phases, group 1 at odd ones
    const int tstride = 2 * (int)gridDim.x;
    int next = (int)blockIdx.x + grp * (int)gridDim.x;  // next tile to stage
    int ready = -1, done = -1;                           // tile staged in LDS / finished tile whose outputs await their stores
    const bool planar_out = p.out_layout == FALNET_OUT_PLANAR_F32;
    PackedOut<T, MT, NT> fin;                            // the finished tile between the two phases (NHWC outputs)
    f32x16 acc[MT][NT];
#ifdef FALNET_WS_STAMPS
    // profiling build (tools/ws2_stamps.py): four s_memtime stamps per phase of workgroup 0, every wave -> p.splitk_ws
    unsigned long long* stamp_out = reinterpret_cast<unsigned long long*>(p.splitk_ws);
    int stamp_i = 0;
#define WS2_STAMP()                                                                                      \
    do {                                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                               \
        unsigned long long t_;                                                                           \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                        \
        __builtin_amdgcn_sched_barrier(0);                                                               \
        if (stamp_out && blockIdx.x == 0 && blockIdx.y == 0 && lane == 0 && stamp_i < 256) stamp_out[wave * 256 + stamp_i] = t_; \
        ++stamp_i;                                                                                       \
    } while (0)
#else
#define WS2_STAMP() do {} while (0)
#endif
    for (int ph = 0; ph < nph; ++ph) {
        WS2_STAMP();  // 0: phase top
        __syncthreads();
        WS2_STAMP();  // 1: after the barrier
        if (((ph + grp) & 1) == 0) {
            // ---- refill role (this group's MFMAs are done, its patch buffer is free): next patch global -> registers, the STORES of the tile
            // finished in the previous phase (its arithmetic ran in the matrix role, behind the MFMAs), patch registers -> LDS
            const bool more = next < ntiles;
            if (more) {
                patch_target(next);
                patch_load_half(Half0{});
                patch_load_half(Half1{});
            }
            WS2_STAMP();  // 2: loads issued
            if (done >= 0) {
                int b, ty0, x;
                tile_xy(done, b, ty0, x);
                const int cstride = p.out_cstride;
                auto pixoff = [&](int mt) -> int64_t {
                    const int y = ty0 + wm * MT + mt;
                    if (!(y < p.OH && x < p.OW)) return (int64_t)-1;
                    return planar_out ? ((int64_t)b * p.Cout * p.OH + y) * p.OW + x : (((int64_t)b * p.OH + y) * p.OW + x) * cstride;
                };
                auto pooloff = [&](int mt) -> int64_t {
                    const int py = (ty0 + wm * MT + mt) >> 1, px = x >> 1, PH = p.OH >> 1, PW = p.OW >> 1;
                    return (py < PH && px < PW) ? (((int64_t)b * PH + py) * PW + px) * cstride : (int64_t)-1;
                };
                if (planar_out) epilogue_direct<T, MT, NT>(p, acc, bias, n0 + wn * 32, lane, pixoff, pooloff);  // (planar f32 logits: whole epilogue here)
                else epilogue_store_packed<T, MT, NT>(p, fin, n0 + wn * 32, lane, pixoff, pooloff);
                done = -1;
            }
            WS2_STAMP();  // 3: outputs stored
            if (more) {
                patch_store_half(Half0{});
                patch_store_half(Half1{});
                ready = next;
                next += tstride;
            }
        } else if (ready >= 0) {
            // ---- matrix role: MFMAs of the tile staged in the previous phase, then its epilogue ARITHMETIC (bias, residual, activation,
            // activation gradient, pooling, 16-bit packing) into `fin` -- VALU work this wave issues while the other wave of the SIMD sits in load / store issue
            WS2_STAMP();  // 2
            WS2_STAMP();
            if constexpr (M16 && sizeof(T) == 2) {
                Acc16 a16[MT][NT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) a16[mt][nt].zero();
                const int lp = lane & 15, lg = lane >> 4;
                const char* pxb = Al + ((wm * MT) * PT_PW + lp) * PITCH + lg * 16;
                const char* wb16 = Wl + (wn * 32 + m16_row_channel(lp)) * PITCH + lg * 16;
                mma_steps16<T, MT, NT, NCH * 9>(
                    [&](int st, int mt, int pt) { const int c = st / 9, t = st % 9;
                                                  return pxb + ((c * NPIX + mt * PT_PW + 16 * pt + (t / 3) * PT_PW + (t % 3)) * PITCH); },
                    [&](int st, int nt, int ct) { return wb16 + ((st * BN + nt * 32 + 16 * ct) * PITCH); },
                    a16);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) a16[mt][nt].to32(acc[mt][nt]);
            } else {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int j = 0; j < 16; ++j) acc[mt][nt][j] = 0.f;
                mma_steps<T, MT, NT, NCH * 9 * KSEG, true, 3, FALNET_PIN_WS>(
                    [&](int st, int mt) { const int c = st / (9 * KSEG), t = (st / KSEG) % 9, ks = st % KSEG;
                                          return ab + ((c * NPIX + mt * PT_PW + (t / 3) * PT_PW + (t % 3)) * PITCH + ks * KSTRIDE); },
                    [&](int st, int nt) { const int tc = st / KSEG, ks = st % KSEG;
                                          return bb + ((tc * BN + nt * 32) * PITCH + ks * KSTRIDE); },
                    acc);
            }
            if (!planar_out) {
                int b, ty0, x;
                tile_xy(ready, b, ty0, x);
                const int cstride = p.out_cstride;
                auto pixoff = [&](int mt) -> int64_t {
                    const int y = ty0 + wm * MT + mt;
                    if (!(y < p.OH && x < p.OW)) return (int64_t)-1;
                    return (((int64_t)b * p.OH + y) * p.OW + x) * cstride;
                };
                auto pooloff = [&](int mt) -> int64_t {
                    const int py = (ty0 + wm * MT + mt) >> 1, px = x >> 1, PH = p.OH >> 1, PW = p.OW >> 1;
                    return (py < PH && px < PW) ? (((int64_t)b * PH + py) * PW + px) * cstride : (int64_t)-1;
                };
                epilogue_direct<T, MT, NT, decltype(pixoff), decltype(pooloff), 1, true>(p, acc, bias, n0 + wn * 32, lane, pixoff, pooloff, &fin);
            }
            done = ready;
            ready = -1;
        } else {
            WS2_STAMP();
            WS2_STAMP();
        }
    }
#undef WS2_STAMP
}

// ------------------------------------------------------------------------------------------ first layer (Cin = 3)
// conv 3x3 / stride 1 / pad 1 on a 3-channel planar f32 image (FAL_netB.py:99 conv0; VGG19 features[0]): K = 27.  The
// generic kernels pad the 3 channels to 32 (K = 288, ~10x wasted MFMA work and a layout-conversion launch in front).
// Here the (8+2)x(32+2)x3 patch is staged planar in LDS straight from the NCHW f32 input, every lane builds its
// K = 32 (27 + 5 zeros) im2col fragment from it (k = c*9 + tap: the OIHW flattening, so the f32 master weights are used
// as they are, no packing), two bf16 MFMAs (sixteen f32 ones) per 32x32 output tile: purely output-write bound.
#ifndef C3_OCC
#define C3_OCC 2
#endif
#ifndef C3_TPW
#define C3_TPW 8  // 8x32 tiles (consecutive along x) per workgroup: weights fetched once, ONE patch load for the whole 8 x 256 strip (round 4: 8 instead
                  // of 4 -- the per-workgroup prologue, 7.6 us of index arithmetic and uncoalesced weight loads, is paid half as often: 64 channels
                  // 41.1 -> 35.8 us, 32 channels 26.5 -> 25.3 us at B = 8, 256 x 512; 2 tiles: 48.2 / 30.1 us; profiles/r04_c3_strip.txt)
#endif
template <typename T, int NT>
__global__ __launch_bounds__(CONV_THREADS, C3_OCC) void conv3x3_c3_kernel(const float* __restrict__ x, const float* __restrict__ w_oihw,
                                                                  const falnet_conv_t p, int groups_x, int tiles_y) {
    // The (8+2) x (4*32+2) x 3 patch of the workgroup's four tiles is staged at once (16 loads per thread in flight, one barrier): the first
    // version staged a tile at a time -- two barriers and an exposed global round trip per 8 x 32 tile, 3.8 us per tile for 0.5 us of work
    // (2.2-2.9 TB/s of output against 5.2 TB/s for the same store shape alone, tools/ubench/store_pattern.hip).
    constexpr int PH = PT_TH + 2, PW = C3_TPW * PT_TW + 2, NEL = 3 * PH * PW, SLOTS = (NEL + CONV_THREADS - 1) / CONV_THREADS;
    __shared__ __attribute__((aligned(16))) float patch[NEL];                       // [3][PH][PW] f32
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
#ifdef C3_STAMPS
    unsigned long long* const stamp_out = reinterpret_cast<unsigned long long*>(const_cast<void*>(p.pool_actout)) + (size_t)blockIdx.x * 8;
#define C3_STAMP(k) do { unsigned long long t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); if (threadIdx.x == 0) stamp_out[k] = t_; } while (0)
#else
#define C3_STAMP(k) do { } while (0)
#endif
    C3_STAMP(0);
    int bid = blockIdx.x;
    const int gx = bid % groups_x;
    bid /= groups_x;
    const int tiy = bid % tiles_y;
    const int b = bid / tiles_y;
    const int ty0 = tiy * PT_TH;
    const int64_t HW = (int64_t)p.IH * p.IW;
    const float* xb = x + (int64_t)b * 3 * HW;
    const int tix0 = gx * C3_TPW, sx0 = tix0 * PT_TW;
    // every global load of the workgroup is issued before the first is consumed: the strip's patch (16 per thread), then the B fragments
    // straight from the f32 OIHW master weights (k = c*9 + tap is their flattening: lane (r, h) owns cout nt*32 + r and the k values its MFMA
    // operand slot covers; 6.9 KB in all, L2-resident, but 32 uncoalesced loads per lane: 4.7 us when they started only after the patch
    // had been stored), then the bias
    float pv[SLOTS];
#pragma unroll
    for (int u = 0; u < SLOTS; ++u) {
        const int i = tid + u * CONV_THREADS;
        const int c = i / (PH * PW), rem = i % (PH * PW), pr = rem / PW, pc = rem % PW;
        const int vy = ty0 - 1 + pr, vx = sx0 - 1 + pc;
        pv[u] = (i < NEL && vy >= 0 && vy < p.IH && vx >= 0 && vx < p.IW) ? xb[c * HW + (int64_t)vy * p.IW + vx] : 0.f;
    }
    C3_STAMP(1);
    constexpr int KS = sizeof(T) == 2 ? 2 : 16, KJ = sizeof(T) == 2 ? 8 : 1;
    float wv[KS][NT][KJ];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int j = 0; j < KJ; ++j) {
                const int co = nt * 32 + r, k = (ks * 2 + h) * KJ + j;
                wv[ks][nt][j] = (co < p.Cout && k < 27) ? w_oihw[co * 27 + k] : 0.f;
            }
    float bias[NT][16];
    load_bias16<NT>(p, 0, h, bias);
#pragma unroll
    for (int u = 0; u < SLOTS; ++u) {
        const int i = tid + u * CONV_THREADS;
        if (i < NEL) patch[i] = pv[u];
    }
    s16x8_t bfr[sizeof(T) == 2 ? 2 : 1][NT];
    if constexpr (sizeof(T) == 2) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int j = 0; j < 8; ++j) bfr[ks][nt][j] = (short)H16<T>::bits(wv[ks][nt][j]);
    }
    constexpr int MT = PT_TH / 4;  // rows per wave
    // im2col element k of output pixel (row, r): patch[c][row + t/3][r + t%3], k = c*9 + t.  k0 / k1 are the k of lane
    // half 0 / 1 and compile-time after unrolling, so the LDS offset is a select between two immediates
    auto a_elem = [&](const float* base, int k0, int k1) -> float {
        const int o0 = k0 < 27 ? ((k0 / 9) * PH + (k0 % 9) / 3) * PW + (k0 % 9) % 3 : 0;
        const int o1 = k1 < 27 ? ((k1 / 9) * PH + (k1 % 9) / 3) * PW + (k1 % 9) % 3 : 0;
        const float v = base[h ? o1 : o0];
        return ((h ? k1 : k0) < 27) ? v : 0.f;
    };
    const int cstride = p.out_cstride;
    const int tiles_x = (p.OW + PT_TW - 1) / PT_TW;
    C3_STAMP(2);
    __syncthreads();  // the strip's patch is in LDS
    C3_STAMP(3);
    for (int tt = 0; tt < C3_TPW; ++tt) {
        if (tt == 1) C3_STAMP(4);
        const int tix = tix0 + tt;
        if (tix >= tiles_x) break;  // workgroup-uniform
        const int tx0 = tix * PT_TW;
        // one output row (32 positions x 32 NT channels) at a time: accumulators + epilogue operands of ONE row in registers
        // (the whole 2-row tile at once needed 256 VGPRs + spills = 2 waves per SIMD for a kernel that only streams its output)
#pragma unroll 1
        for (int mt = 0; mt < MT; ++mt) {
            f32x16 acc[1][NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int j = 0; j < 16; ++j) acc[0][nt][j] = 0.f;
            const float* base = patch + (wave * MT + mt) * PW + tt * PT_TW + r;
            if constexpr (sizeof(T) == 2) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    s16x8_t afr;
#pragma unroll
                    for (int j = 0; j < 8; ++j) afr[j] = (short)H16<T>::bits(a_elem(base, ks * 16 + j, ks * 16 + 8 + j));
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[0][nt] = H16<T>::mma(bfr[ks][nt], afr, acc[0][nt]);
                }
            } else {
#pragma unroll
                for (int ks = 0; ks < 16; ++ks) {
                    const float a = a_elem(base, ks * 2, ks * 2 + 1);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[0][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[ks][nt][0], a, acc[0][nt], 0, 0, 0);
                }
            }
            auto pixoff = [&](int) -> int64_t {  // operands exchanged above: pixels on lanes
                const int y = ty0 + wave * MT + mt, xx = tx0 + r;
                return (y < p.OH && xx < p.OW) ? (((int64_t)b * p.OH + y) * p.OW + xx) * cstride : (int64_t)-1;
            };
            epilogue_direct<T, 1, NT, decltype(pixoff), NoPool, 1, false, true>(p, acc, bias, 0, lane, pixoff);  // (NHWC only: falnet_conv3x3_c3)
        }
    }
    C3_STAMP(5);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    C3_STAMP(6);
}
#undef C3_STAMP

// ------------------------------------------------------------------------------------------ wgrad
// dW[co, tap, ci] = sum_p G[p, co] * In[nbr(p, tap), ci]: both operands are pixel-major (the contraction
// index is the slow one), so the LDS tiles are [pixel][channel] and the MFMA operands are read transposed:
//   bf16: ds_read_b64_tr_b16 (two per 8-element fragment);  f32: one ds_read_b32 per lane (A[m][k]: m on lanes).
// Workgroup = 64 couts x 64 cins for one (tap, pixel split); 4 waves 2x2, 32x32 each; 64 pixels per K step.
#define WG_BM 64
#define WG_BN 64
#define WG_KP 64

typedef short s16x4 __attribute__((ext_vector_type(4)));

template <typename T>
__global__ __launch_bounds__(CONV_THREADS) void wgrad_kernel(const falnet_wgrad_t p, int w_rows) {
    constexpr int EPS = 16 / sizeof(T);
    constexpr int ROW_ELEMS = 64;                          // channels per tile row
    constexpr int SEGS = ROW_ELEMS / EPS;                  // 16-B segments per row (8 bf16 / 16 f32)
    constexpr int PITCH = ROW_ELEMS * sizeof(T) + 16;      // bytes
    constexpr int LOADS = WG_KP * SEGS / CONV_THREADS;     // per operand per thread (2 bf16 / 4 f32)
    __shared__ __attribute__((aligned(16))) char lds[2 * 2 * WG_KP * PITCH];
    auto Gbuf = [&](int b) -> char* { return lds + b * 2 * WG_KP * PITCH; };
    auto Ibuf = [&](int b) -> char* { return lds + b * 2 * WG_KP * PITCH + WG_KP * PITCH; };

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int ci0 = blockIdx.x * WG_BN;   // packed input-channel offset (over all sources)
    const int co0 = blockIdx.y * WG_BM;
    const int tap = blockIdx.z % p.ntaps, split = blockIdx.z / p.ntaps;
    const int dy = p.tap_dy[tap], dx = p.tap_dx[tap];
    const int c_first = p.src[0].C;  // packed channels [0, c_first) come from source 0, the rest from source 1

    const int64_t M = (int64_t)p.B * p.TH * p.TW;
    const int64_t per = ((M + p.nsplit - 1) / p.nsplit + WG_KP - 1) / WG_KP * WG_KP;
    const int64_t pbeg = (int64_t)split * per, pend = pbeg + per < M ? pbeg + per : M;
    const int niter = pbeg < pend ? (int)((pend - pbeg + WG_KP - 1) / WG_KP) : 0;

    uint4 greg[LOADS], ireg[LOADS];
    int64_t pcur = pbeg;
    auto gload = [&]() {
#pragma unroll
        for (int i = 0; i < LOADS; ++i) {
            const int idx = tid + i * CONV_THREADS;
            const int row = idx / SEGS, seg = idx % SEGS;
            const int64_t m = pcur + row;
            uint4 g = make_uint4(0, 0, 0, 0), v = make_uint4(0, 0, 0, 0);
            if (m < pend) {
                if (co0 + seg * EPS < p.gC)
                    g = *reinterpret_cast<const uint4*>(reinterpret_cast<const T*>(p.gout) + m * p.gC + co0 + seg * EPS);
                const int tx = (int)(m % p.TW), ty = (int)((m / p.TW) % p.TH), b = (int)(m / ((int64_t)p.TW * p.TH));
                int vy = ty * p.isy + dy, vx = tx * p.isx + dx;
                const int cpk = ci0 + seg * EPS;  // packed channel of this 16-B segment
                const falnet_src_t& S = p.src[cpk < c_first ? 0 : 1];
                const int cloc = cpk < c_first ? cpk : cpk - c_first;
                if (vy >= 0 && vy < p.IH && vx >= 0 && vx < p.IW && cpk < p.cin_total) {
                    if ((S.H != p.IH) || (S.W != p.IW)) {
                        vy = (2 * S.H == p.IH) ? (vy >> 1) : (int)(((int64_t)vy * S.H) / p.IH);
                        vx = (2 * S.W == p.IW) ? (vx >> 1) : (int)(((int64_t)vx * S.W) / p.IW);
                    }
                    v = *reinterpret_cast<const uint4*>(reinterpret_cast<const T*>(S.ptr) + (int64_t)b * S.sb +
                                                        (int64_t)vy * S.sy + (int64_t)vx * S.sx + cloc);
                }
            }
            greg[i] = g;
            ireg[i] = v;
        }
        pcur += WG_KP;
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < LOADS; ++i) {
            const int idx = tid + i * CONV_THREADS;
            const int row = idx / SEGS, seg = idx % SEGS;
            *reinterpret_cast<uint4*>(Gbuf(buf) + row * PITCH + seg * 16) = greg[i];
            *reinterpret_cast<uint4*>(Ibuf(buf) + row * PITCH + seg * 16) = ireg[i];
        }
    };

    f32x16 acc;
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = 0.f;

    if (niter > 0) {
        gload();
        lstore(0);
    }
    __syncthreads();
    for (int it = 0; it < niter; ++it) {
        const int cur = it & 1;
        if (it + 1 < niter) gload();
        const char* G = Gbuf(cur);
        const char* I = Ibuf(cur);
        if constexpr (sizeof(T) == 2) {
            // lane group g16 = lane>>4: k-half = g16>>1, 16-channel block = g16&1; lane i=lane&15 supplies row q=i>>2, cols 4*(i&3)
            const int i16 = lane & 15, g16 = lane >> 4;
            const int kh = g16 >> 1, cb = g16 & 1, q = i16 >> 2, pc = i16 & 3;
#pragma unroll
            for (int ks = 0; ks < WG_KP / 16; ++ks) {
                const int krow = ks * 16 + kh * 8 + q;
                const int acol = (wm * 32 + cb * 16 + pc * 4) * 2, bcol = (wn * 32 + cb * 16 + pc * 4) * 2;
                typedef s16x4 __attribute__((address_space(3))) * lds_v4;
                s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(G + krow * PITCH + acol));
                s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(G + (krow + 4) * PITCH + acol));
                s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(I + krow * PITCH + bcol));
                s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(I + (krow + 4) * PITCH + bcol));
                typedef short s16x8 __attribute__((ext_vector_type(8)));
                s16x8 av = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
                s16x8 bv = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
                acc = H16<T>::mma(__builtin_bit_cast(s16x8_t, av), __builtin_bit_cast(s16x8_t, bv), acc);
            }
        } else {
            const int r = lane & 31, h = lane >> 5;
#pragma unroll
            for (int ks = 0; ks < WG_KP / 2; ++ks) {
                const float a = *reinterpret_cast<const float*>(G + (ks * 2 + h) * PITCH + (wm * 32 + r) * 4);
                const float b = *reinterpret_cast<const float*>(I + (ks * 2 + h) * PITCH + (wn * 32 + r) * 4);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
            }
        }
        if (it + 1 < niter) lstore(cur ^ 1);
        __syncthreads();
    }
    // partial[split][tap][co][ci]
    const int r = lane & 31, h = lane >> 5;
    const int ci = ci0 + wn * 32 + r;
    if (ci < p.cin_total) {
        float* dst = p.partial + (((int64_t)split * p.ntaps + tap) * w_rows) * p.cin_total;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int co = co0 + wm * 32 + (j & 3) + 8 * (j >> 2) + 4 * h;
            if (co < w_rows) dst[(int64_t)co * p.cin_total + ci] = acc[j];
        }
    }
}

// ------------------------------------------------------------------------------------------ wgrad, halo-patch form
// Dense 3x3 stride-1 layers (the bulk of the weight-gradient FLOPs): a workgroup owns a 32x32 (cout x cin)
// channel block for ALL nine taps and walks a range of 4x32-position patches.  Per patch the gout rows
// [128 px][32 cout] and the input halo [6x34 px][32 cin] are staged ONCE in LDS (the per-tap kernel re-read both
// nine times); wave w contracts image row w of the patch (K = 32 positions) into its nine 32x32 accumulators,
// operands read transposed (bf16: ds_read_b64_tr_b16) as in wgrad_kernel.  The four waves' accumulators are
// summed through LDS at the end and the workgroup writes ONE f32 slab [9][32][32] into
// partial[split][tap][co][ci].  Next patch is prefetched into registers behind the MFMAs.
#define WP_TH 4
#define WP_TW 32
#define WP_PW (WP_TW + 2)
#define WP_NPIX ((WP_TH + 2) * WP_PW)

#define WP_THREADS 192  // three waves: wave w owns the tap row dy = w-1 (taps 3w..3w+2)

// Bias gradient from the gout tile a weight-gradient workgroup has in LDS ([plane][pixel][32 channels], zero-filled outside the
// image): every thread owns one 16-B channel segment and strides over the tile's pixels, accumulating in registers across
// all patches of the workgroup; bias_grad_flush sums the pixel groups through LDS and issues one atomic per channel.
template <typename T, int COT, int NTHR>
__device__ __forceinline__ void bias_grad_accumulate(const char* G, int tid, float (&bsum)[16 / (int)sizeof(T)]) {
    constexpr int EPS = 16 / (int)sizeof(T), SEGS = 32 / EPS, NSEG = SEGS * COT, PSTEP = NTHR / NSEG;
    static_assert(NTHR % NSEG == 0, "threads map evenly onto channel segments");
    constexpr int NPIXT = WP_TH * WP_TW, G_PLANE = NPIXT * 32 * (int)sizeof(T);
    const int sg = tid % NSEG, pg = tid / NSEG;
    const char* base = G + (sg / SEGS) * G_PLANE + (sg % SEGS) * 16;
    for (int px = pg; px < NPIXT; px += PSTEP) {
        const uint4 v = *reinterpret_cast<const uint4*>(base + px * 32 * (int)sizeof(T));
        if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const unsigned w = (&v.x)[i];
                bsum[2 * i] += H16<T>::lo(w);
                bsum[2 * i + 1] += H16<T>::hi(w);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) bsum[i] += __uint_as_float((&v.x)[i]);
        }
    }
}
template <typename T, int COT, int NTHR>
__device__ __forceinline__ void bias_grad_flush(float* lds_f /* >= NTHR * EPS floats, all waves past their last LDS use */, int tid,
                                                const float (&bsum)[16 / (int)sizeof(T)], float* db, int co0, int gC) {
    constexpr int EPS = 16 / (int)sizeof(T), SEGS = 32 / EPS, NSEG = SEGS * COT, PSTEP = NTHR / NSEG;
#pragma unroll
    for (int i = 0; i < EPS; ++i) lds_f[tid * EPS + i] = bsum[i];
    __syncthreads();
    if (tid < NSEG * EPS) {
        const int sg = tid / EPS, i = tid % EPS;
        float t = 0.f;
        for (int pg = 0; pg < PSTEP; ++pg) t += lds_f[(pg * NSEG + sg) * EPS + i];
        const int co = co0 + (sg / SEGS) * 32 + (sg % SEGS) * EPS + i;
        if (co < gC) atomicAdd(db + co, t);
    }
}

// CIT x COT = 32-channel tiles per workgroup along cin / cout (1x1, or 2x2 for bf16 layers with >= 64 channels on both
// sides): every wave then owns 3 taps x CIT x COT accumulator tiles, and one A (gout) fragment feeds 3*CIT MFMAs, one B
// (input) fragment COT of them -- half the LDS reads and half the global bytes per MFMA of the 1x1 form, whose ~2.7
// transposed reads per MFMA and 21 KB per 72 MFMAs sit on the LDS / CU load path.  Channel planes are stored separately
// ([plane][pixel][32 channels], 64-B rows) so the transposed-read addressing is the same for every plane.
template <typename T, int CIT, int COT>
__global__ __launch_bounds__(WP_THREADS) void wgrad3x3_patch_kernel(const falnet_wgrad_t p, int w_rows, int tiles_x, int tiles_y,
                                                                    int patches_per_split) {
    constexpr int EPS = 16 / (int)sizeof(T);
    constexpr int ROWB_ = 32 * (int)sizeof(T);      // bytes of 32 channels
    constexpr int SEGS = ROWB_ / 16;                // 4 (bf16) / 8 (f32)
    // no row padding: a ds_read_b64_tr_b16 32-lane half reads 4 rows x 64 B = exactly the 64 banks once
    constexpr int PITCH = ROWB_;
    constexpr int G_PLANE = WP_TH * WP_TW * PITCH, I_PLANE = WP_NPIX * PITCH;
    constexpr int G_BYTES = COT * G_PLANE, I_BYTES = CIT * I_PLANE;
    constexpr int G_LOADS = WP_TH * WP_TW * SEGS * COT, I_LOADS = WP_NPIX * SEGS * CIT;
    constexpr int G_SLOTS = (G_LOADS + WP_THREADS - 1) / WP_THREADS;
    constexpr int I_SLOTS = (I_LOADS + WP_THREADS - 1) / WP_THREADS;
    __shared__ __attribute__((aligned(16))) char lds[2 * (G_BYTES + I_BYTES)];
    auto Gbuf = [&](int b) -> char* { return lds + b * (G_BYTES + I_BYTES); };
    auto Ibuf = [&](int b) -> char* { return lds + b * (G_BYTES + I_BYTES) + G_BYTES; };

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ci0 = blockIdx.x * 32 * CIT, co0 = blockIdx.y * 32 * COT, split = blockIdx.z;
    const int c_first = p.src[0].C;
    // per cin tile: source, channel offset inside it, resize flags (workgroup-uniform; a 64-channel block may straddle
    // the two sources of a fused concat)
    const T* t_ptr[CIT];
    int64_t t_sb[CIT], t_sy[CIT], t_sx[CIT];
    int t_H[CIT], t_W[CIT];
    bool t_ok[CIT];
#pragma unroll
    for (int t = 0; t < CIT; ++t) {
        const int c = ci0 + 32 * t;
        const bool second = c >= c_first;
        t_ok[t] = c < p.cin_total;
        t_ptr[t] = reinterpret_cast<const T*>(second ? p.src[1].ptr : p.src[0].ptr) + (second ? c - c_first : c);
        t_sb[t] = second ? p.src[1].sb : p.src[0].sb;
        t_sy[t] = second ? p.src[1].sy : p.src[0].sy;
        t_sx[t] = second ? p.src[1].sx : p.src[0].sx;
        t_H[t] = second ? p.src[1].H : p.src[0].H;
        t_W[t] = second ? p.src[1].W : p.src[0].W;
    }
    const int npatch = p.B * tiles_x * tiles_y;
    const int pbeg = split * patches_per_split, pend = min(pbeg + patches_per_split, npatch);

    // halo slots: (row, col) of the 6x34 patch per slot (division by 34 hoisted out of the patch loop); with CIT = 2 the
    // eight 16-B segments of a pixel are consecutive lanes (one full 128-B line when both tiles share a source)
    short i_row[I_SLOTS], i_col[I_SLOTS];
#pragma unroll
    for (int u = 0; u < I_SLOTS; ++u) {
        const int pix = (tid + u * WP_THREADS) / (SEGS * CIT);
        i_row[u] = (short)(pix / WP_PW);
        i_col[u] = (short)(pix % WP_PW);
    }

    // Interior patches (the vast majority) take a fast path: every slot's element offset relative to the patch origin is
    // loop-invariant (also through an exact 2x nearest upsampling: origins are even), so a load is one 64-bit add -- the
    // general path costs ~25 VALU per load, i.e. ~8 VALU per MFMA of this kernel (PMC), as much issue time as the MFMAs.
    int g_off[G_SLOTS], i_off[I_SLOTS];
    bool fast_ok = true;  // every tile's source is at the launch size or exactly half of it
#pragma unroll
    for (int t = 0; t < CIT; ++t)
        fast_ok = fast_ok && (t_H[t] == p.IH || 2 * t_H[t] == p.IH) && (t_W[t] == p.IW || 2 * t_W[t] == p.IW) &&
                  (int64_t)p.B * t_sb[t] < (1ll << 31);
    fast_ok = fast_ok && (int64_t)p.B * p.TH * p.TW * p.gC < (1ll << 31) && co0 + 32 * COT <= p.gC && ci0 + 32 * CIT <= p.cin_total;
#pragma unroll
    for (int u = 0; u < G_SLOTS; ++u) {
        const int idx = tid + u * WP_THREADS;
        const int seg = idx % (SEGS * COT), pix = idx / (SEGS * COT);
        g_off[u] = ((pix / WP_TW) * p.TW + pix % WP_TW) * p.gC + seg * EPS;
    }
#pragma unroll
    for (int u = 0; u < I_SLOTS; ++u) {
        const int idx = tid + u * WP_THREADS;
        const int seg8 = idx % (SEGS * CIT), seg = seg8 % SEGS;
        const bool t1 = CIT > 1 && seg8 >= SEGS;
        const int hs = (t1 ? t_H[CIT - 1] : t_H[0]) != p.IH ? 1 : 0, ws = (t1 ? t_W[CIT - 1] : t_W[0]) != p.IW ? 1 : 0;
        const int ry = (i_row[u] - 1) >> hs, rx = (i_col[u] - 1) >> ws;  // arithmetic shifts: -1 stays -1
        i_off[u] = (int)(ry * (t1 ? t_sy[CIT - 1] : t_sy[0]) + rx * (t1 ? t_sx[CIT - 1] : t_sx[0])) + seg * EPS;
    }
    struct Regs { uint4 g[G_SLOTS]; uint4 i[I_SLOTS]; };
    auto gload = [&](int patch, Regs& R) {
        int q = patch;
        const int tix = q % tiles_x;
        q /= tiles_x;
        const int tiy = q % tiles_y;
        const int b = q / tiles_y;
        const int y0 = tiy * WP_TH, x0 = tix * WP_TW;
        const T* gbase = reinterpret_cast<const T*>(p.gout) + ((int64_t)b * p.TH * p.TW) * p.gC + co0;
        if (fast_ok && y0 >= 1 && x0 >= 1 && y0 + WP_TH + 1 <= p.IH && x0 + WP_TW + 1 <= p.IW && y0 + WP_TH <= p.TH && x0 + WP_TW <= p.TW) {
            const T* gb = gbase + ((int64_t)y0 * p.TW + x0) * p.gC;
#pragma unroll
            for (int u = 0; u < G_SLOTS; ++u) {
                uint4 v = make_uint4(0, 0, 0, 0);
                if (tid + u * WP_THREADS < G_LOADS) v = *reinterpret_cast<const uint4*>(gb + g_off[u]);
                R.g[u] = v;
            }
            const T* ib[CIT];
#pragma unroll
            for (int t = 0; t < CIT; ++t) {
                const int hs = t_H[t] != p.IH ? 1 : 0, ws = t_W[t] != p.IW ? 1 : 0;
                ib[t] = t_ptr[t] + (int64_t)b * t_sb[t] + (int64_t)(y0 >> hs) * t_sy[t] + (int64_t)(x0 >> ws) * t_sx[t];
            }
#pragma unroll
            for (int u = 0; u < I_SLOTS; ++u) {
                const int idx = tid + u * WP_THREADS;
                const bool t1 = CIT > 1 && idx % (SEGS * CIT) >= SEGS;
                uint4 v = make_uint4(0, 0, 0, 0);
                if (idx < I_LOADS) v = *reinterpret_cast<const uint4*>((t1 ? ib[CIT - 1] : ib[0]) + i_off[u]);
                R.i[u] = v;
            }
            return;
        }
#pragma unroll
        for (int u = 0; u < G_SLOTS; ++u) {
            const int idx = tid + u * WP_THREADS;
            const int seg = idx % (SEGS * COT), pix = idx / (SEGS * COT);
            const int y = y0 + pix / WP_TW, x = x0 + pix % WP_TW;  // WP_TW = 32: shifts
            uint4 v = make_uint4(0, 0, 0, 0);
            if (idx < G_LOADS && y < p.TH && x < p.TW && co0 + seg * EPS < p.gC)
                v = *reinterpret_cast<const uint4*>(gbase + ((int64_t)y * p.TW + x) * p.gC + seg * EPS);
            R.g[u] = v;
        }
#pragma unroll
        for (int u = 0; u < I_SLOTS; ++u) {
            const int idx = tid + u * WP_THREADS;
            const int seg8 = idx % (SEGS * CIT), t = seg8 / SEGS, seg = seg8 % SEGS;
            const bool t1 = CIT > 1 && t == 1;
            uint4 v = make_uint4(0, 0, 0, 0);
            int vy = y0 - 1 + i_row[u], vx = x0 - 1 + i_col[u];
            if (idx < I_LOADS && (t1 ? t_ok[CIT - 1] : t_ok[0]) && vy >= 0 && vy < p.IH && vx >= 0 && vx < p.IW) {
                const int sH = t1 ? t_H[CIT - 1] : t_H[0], sW = t1 ? t_W[CIT - 1] : t_W[0];
                if (sH != p.IH) vy = (2 * sH == p.IH) ? (vy >> 1) : (int)(((int64_t)vy * sH) / p.IH);
                if (sW != p.IW) vx = (2 * sW == p.IW) ? (vx >> 1) : (int)(((int64_t)vx * sW) / p.IW);
                const T* ib = t1 ? t_ptr[CIT - 1] : t_ptr[0];
                v = *reinterpret_cast<const uint4*>(ib + (int64_t)b * (t1 ? t_sb[CIT - 1] : t_sb[0]) + (int64_t)vy * (t1 ? t_sy[CIT - 1] : t_sy[0]) +
                                                    (int64_t)vx * (t1 ? t_sx[CIT - 1] : t_sx[0]) + seg * EPS);
            }
            R.i[u] = v;
        }
    };
    auto lstore = [&](int buf, const Regs& R) {
#pragma unroll
        for (int u = 0; u < G_SLOTS; ++u) {
            const int idx = tid + u * WP_THREADS;
            const int seg8 = idx % (SEGS * COT), pix = idx / (SEGS * COT);
            if (idx < G_LOADS) *reinterpret_cast<uint4*>(Gbuf(buf) + (seg8 / SEGS) * G_PLANE + (pix * SEGS + seg8 % SEGS) * 16) = R.g[u];
        }
#pragma unroll
        for (int u = 0; u < I_SLOTS; ++u) {
            const int idx = tid + u * WP_THREADS;
            const int seg8 = idx % (SEGS * CIT), pix = idx / (SEGS * CIT);
            if (idx < I_LOADS) *reinterpret_cast<uint4*>(Ibuf(buf) + (seg8 / SEGS) * I_PLANE + (pix * SEGS + seg8 % SEGS) * 16) = R.i[u];
        }
    };

    f32x16 acc[3][CIT][COT];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int a = 0; a < CIT; ++a)
#pragma unroll
            for (int c = 0; c < COT; ++c)
#pragma unroll
                for (int j = 0; j < 16; ++j) acc[t][a][c][j] = 0.f;

    auto compute = [&](int cur) {
        const char* G = Gbuf(cur);
        const char* I = Ibuf(cur) + wave * (WP_PW * PITCH);   // tap row dy = wave-1: halo rows shifted by `wave`
        if constexpr (sizeof(T) == 2) {
            const int i16 = lane & 15, g16 = lane >> 4;
            const int kh = g16 >> 1, cb = g16 & 1, q = i16 >> 2, pc = i16 & 3;
            typedef s16x4 __attribute__((address_space(3))) * lds_v4;
            const int lane_off = (kh * 8 + q) * PITCH + (cb * 16 + pc * 4) * 2;
            const char* gl = G + lane_off;
            const char* il = I + lane_off;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {  // K = 128 positions: image row ks>>1 of the patch, 16-position half ks&1
                const int goff = ks * 16 * PITCH;
                s16x8_t av[COT];
#pragma unroll
                for (int c = 0; c < COT; ++c) {
                    s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(gl + c * G_PLANE + goff));
                    s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(gl + c * G_PLANE + goff + 4 * PITCH));
                    av[c] = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
                }
#pragma unroll
                for (int a = 0; a < CIT; ++a)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        const int ioff = a * I_PLANE + ((ks >> 1) * WP_PW + (ks & 1) * 16 + dx) * PITCH;
                        s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(il + ioff));
                        s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(il + ioff + 4 * PITCH));
                        const s16x8_t bv = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
                        for (int c = 0; c < COT; ++c) acc[dx][a][c] = H16<T>::mma(av[c], bv, acc[dx][a][c]);
                    }
            }
        } else {
            const int r = lane & 31, h = lane >> 5;
#pragma unroll 8
            for (int ks = 0; ks < 64; ++ks) {  // 2 positions per MFMA
                const int pos = ks * 2 + h;     // 0..127 inside the patch
                float av[COT];
#pragma unroll
                for (int c = 0; c < COT; ++c) av[c] = *reinterpret_cast<const float*>(G + c * G_PLANE + pos * PITCH + r * 4);
                const int ipix = (pos >> 5) * WP_PW + (pos & 31);
#pragma unroll
                for (int a = 0; a < CIT; ++a)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        const float bb = *reinterpret_cast<const float*>(I + a * I_PLANE + (ipix + dx) * PITCH + r * 4);
#pragma unroll
                        for (int c = 0; c < COT; ++c) acc[dx][a][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c], bb, acc[dx][a][c], 0, 0, 0);
                    }
            }
        }
    };

    // distance-1 prefetch (a second register set for distance 2 costs a wave of occupancy and measured slower)
    Regs R0;
    if (pbeg < pend) {
        gload(pbeg, R0);
        lstore(0, R0);
    }
    __syncthreads();
    const bool do_bias = p.bias_grad != nullptr && blockIdx.x == 0;  // one cin tile per cout slice sums the bias gradient
    float bsum[EPS];
#pragma unroll
    for (int i = 0; i < EPS; ++i) bsum[i] = 0.f;
    for (int patch = pbeg; patch < pend; ++patch) {
        const int cur = (patch - pbeg) & 1;
        if (patch + 1 < pend) gload(patch + 1, R0);
        compute(cur);
        if (do_bias) bias_grad_accumulate<T, COT, WP_THREADS>(Gbuf(cur), tid, bsum);
        if (patch + 1 < pend) lstore(cur ^ 1, R0);
        __syncthreads();
    }
    if (do_bias) bias_grad_flush<T, COT, WP_THREADS>(reinterpret_cast<float*>(lds), tid, bsum, p.bias_grad, co0, p.cout);
    // every wave owns its three taps: no cross-wave reduction
    const int r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int a = 0; a < CIT; ++a) {
        const int ci = ci0 + 32 * a + r;
        if (ci < p.cin_total) {
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                float* dst = p.partial + (((int64_t)split * 9 + wave * 3 + dx) * w_rows) * p.cin_total;
#pragma unroll
                for (int c = 0; c < COT; ++c)
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        const int co = co0 + 32 * c + (j & 3) + 8 * (j >> 2) + 4 * h;
                        if (co < w_rows) dst[(int64_t)co * p.cin_total + ci] = acc[dx][a][c][j];
                    }
            }
        }
    }
}

// Stride-2 3x3 weight gradient (encoder convs conv1..conv6, FAL_netB.py:101-111), bf16: the same wave-per-tap-row scheme
// as the dense kernel on a 4x32 block of OUTPUT positions; the (2*4+1)x(2*32+1) input region is loaded as whole contiguous
// rows and de-interleaved by row / column parity into four LDS planes (even/odd rows x even/odd columns), so that the
// pixels tap (ky,kx) needs for 16 consecutive outputs (2x+kx-1: stride 2 in the image) are 16 CONSECUTIVE rows of one
// plane -- the transposed reads and their conflict-free 64-B pitch are exactly those of the dense kernel.
//   tap ky: rows 2y+ky-1 -> ky=1: odd region rows (index y), ky=0 / 2: even region rows (index y / y+1); columns alike.
// (The per-tap gather kernel it replaces ran these layers at 65-130 TFLOP/s on the longest chain of the backward pass.)
#define WS2_RH (2 * WP_TH + 1)   // region rows
#define WS2_RW (2 * WP_TW + 1)   // region columns
template <typename T, int COT>
__global__ __launch_bounds__(WP_THREADS, 2) void wgrad3x3_s2_kernel(const falnet_wgrad_t p, int w_rows, int tiles_x, int tiles_y,
                                                                 int patches_per_split) {
    constexpr int EPS = 8, SEGS = 4, PITCH = 64;
    constexpr int NE_R = WP_TH + 1, NO_R = WP_TH, NE_C = WP_TW + 1, NO_C = WP_TW;  // even / odd region rows and columns
    // plane (row parity, column parity) -> pixel offset of its first pixel; E = even region index
    constexpr int P_EE = 0, P_EO = P_EE + NE_R * NE_C, P_OE = P_EO + NE_R * NO_C, P_OO = P_OE + NO_R * NE_C, I_PIX = P_OO + NO_R * NO_C;
    static_assert(I_PIX == WS2_RH * WS2_RW, "the four planes tile the region");
    constexpr int G_PLANE = WP_TH * WP_TW * PITCH;
    constexpr int G_BYTES = COT * G_PLANE, I_BYTES = I_PIX * PITCH;
    constexpr int G_LOADS = WP_TH * WP_TW * SEGS * COT, I_LOADS = I_PIX * SEGS;
    constexpr int G_SLOTS = (G_LOADS + WP_THREADS - 1) / WP_THREADS, I_SLOTS = (I_LOADS + WP_THREADS - 1) / WP_THREADS;
    // ONE LDS buffer (54 KB with COT = 2) + register prefetch of the next block: two workgroups per CU overlap each other
    __shared__ __attribute__((aligned(16))) char lds[G_BYTES + I_BYTES];
    auto Gbuf = [&](int) -> char* { return lds; };
    auto Ibuf = [&](int) -> char* { return lds + G_BYTES; };
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ci0 = blockIdx.x * 32, co0 = blockIdx.y * 32 * COT, split = blockIdx.z;
    const int c_first = p.src[0].C;
    const bool second = ci0 >= c_first;
    const T* s_ptr = reinterpret_cast<const T*>(second ? p.src[1].ptr : p.src[0].ptr) + (second ? ci0 - c_first : ci0);
    const int64_t s_sb = second ? p.src[1].sb : p.src[0].sb, s_sy = second ? p.src[1].sy : p.src[0].sy, s_sx = second ? p.src[1].sx : p.src[0].sx;
    const int npatch = p.B * tiles_x * tiles_y;
    const int pbeg = split * patches_per_split, pend = min(pbeg + patches_per_split, npatch);

    // region slots: (row, column, 16-B segment) -> LDS offset inside the parity plane (loop-invariant)
    short i_row[I_SLOTS], i_col[I_SLOTS];
    int i_lds[I_SLOTS];
#pragma unroll
    for (int u = 0; u < I_SLOTS; ++u) {
        const int idx = tid + u * WP_THREADS;
        const int seg = idx % SEGS, pix = idx / SEGS;
        const int r = pix / WS2_RW, c = pix % WS2_RW;
        i_row[u] = (short)r;
        i_col[u] = (short)c;
        const int base = (r & 1) ? ((c & 1) ? P_OO : P_OE) : ((c & 1) ? P_EO : P_EE);
        const int pw = (c & 1) ? NO_C : NE_C;
        i_lds[u] = idx < I_LOADS ? (base + (r >> 1) * pw + (c >> 1)) * PITCH + seg * 16 : -1;
    }
    struct Regs { uint4 g[G_SLOTS]; uint4 i[I_SLOTS]; };
    auto gload = [&](int patch, Regs& R) {
        int q = patch;
        const int tix = q % tiles_x;
        q /= tiles_x;
        const int tiy = q % tiles_y;
        const int b = q / tiles_y;
        const int y0 = tiy * WP_TH, x0 = tix * WP_TW;
        const T* gbase = reinterpret_cast<const T*>(p.gout) + ((int64_t)b * p.TH * p.TW) * p.gC + co0;
#pragma unroll
        for (int u = 0; u < G_SLOTS; ++u) {
            const int idx = tid + u * WP_THREADS;
            const int seg = idx % (SEGS * COT), pix = idx / (SEGS * COT);
            const int y = y0 + pix / WP_TW, x = x0 + pix % WP_TW;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (idx < G_LOADS && y < p.TH && x < p.TW && co0 + seg * EPS < p.gC)
                v = *reinterpret_cast<const uint4*>(gbase + ((int64_t)y * p.TW + x) * p.gC + seg * EPS);
            R.g[u] = v;
        }
        const T* ibase = s_ptr + (int64_t)b * s_sb;
#pragma unroll
        for (int u = 0; u < I_SLOTS; ++u) {
            const int idx = tid + u * WP_THREADS;
            uint4 v = make_uint4(0, 0, 0, 0);
            const int vy = 2 * y0 - 1 + i_row[u], vx = 2 * x0 - 1 + i_col[u];
            if (idx < I_LOADS && vy >= 0 && vy < p.IH && vx >= 0 && vx < p.IW)
                v = *reinterpret_cast<const uint4*>(ibase + (int64_t)vy * s_sy + (int64_t)vx * s_sx + (idx % SEGS) * EPS);
            R.i[u] = v;
        }
    };
    auto lstore = [&](int buf, const Regs& R) {
#pragma unroll
        for (int u = 0; u < G_SLOTS; ++u) {
            const int idx = tid + u * WP_THREADS;
            const int seg8 = idx % (SEGS * COT), pix = idx / (SEGS * COT);
            if (idx < G_LOADS) *reinterpret_cast<uint4*>(Gbuf(buf) + (seg8 / SEGS) * G_PLANE + (pix * SEGS + seg8 % SEGS) * 16) = R.g[u];
        }
#pragma unroll
        for (int u = 0; u < I_SLOTS; ++u)
            if (i_lds[u] >= 0) *reinterpret_cast<uint4*>(Ibuf(buf) + i_lds[u]) = R.i[u];
    };

    f32x16 acc[3][COT];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int c = 0; c < COT; ++c)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[t][c][j] = 0.f;
    const int i16 = lane & 15, g16 = lane >> 4;
    const int kh = g16 >> 1, cb = g16 & 1, q4 = i16 >> 2, pc = i16 & 3;
    typedef s16x4 __attribute__((address_space(3))) * lds_v4;
    const int lane_off = (kh * 8 + q4) * PITCH + (cb * 16 + pc * 4) * 2;
    // wave = ky: region-row parity and row shift inside the plane
    const bool odd_rows = wave == 1;
    const int row_shift = wave == 2 ? 1 : 0;

    auto compute = [&](int cur) {
        const char* gl = Gbuf(cur) + lane_off;
        const char* il = Ibuf(cur) + lane_off;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {  // K = 128 output positions: block row ks>>1, 16-position half ks&1
            const int goff = ks * 16 * PITCH;
            s16x8_t av[COT];
#pragma unroll
            for (int c = 0; c < COT; ++c) {
                s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(gl + c * G_PLANE + goff));
                s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(gl + c * G_PLANE + goff + 4 * PITCH));
                av[c] = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
            }
            const int prow = (ks >> 1) + row_shift;  // row inside the parity plane
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                // column parity odd for kx = 1; even columns start at lx (kx = 0) or lx + 1 (kx = 2)
                const int pw = kx == 1 ? NO_C : NE_C;
                const int pbase = odd_rows ? (kx == 1 ? P_OO : P_OE) : (kx == 1 ? P_EO : P_EE);
                const int ioff = (pbase + prow * pw + (ks & 1) * 16 + (kx == 2 ? 1 : 0)) * PITCH;
                s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(il + ioff));
                s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(il + ioff + 4 * PITCH));
                const s16x8_t bv = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
                for (int c = 0; c < COT; ++c) acc[kx][c] = H16<T>::mma(av[c], bv, acc[kx][c]);
            }
        }
    };

    Regs R0;
    if (pbeg < pend) {
        gload(pbeg, R0);
        lstore(0, R0);
    }
    __syncthreads();
    const bool do_bias = p.bias_grad != nullptr && blockIdx.x == 0;
    float bsum[EPS];
#pragma unroll
    for (int i = 0; i < EPS; ++i) bsum[i] = 0.f;
    for (int patch = pbeg; patch < pend; ++patch) {
        if (patch + 1 < pend) gload(patch + 1, R0);
        compute(0);
        if (do_bias) bias_grad_accumulate<T, COT, WP_THREADS>(Gbuf(0), tid, bsum);
        __syncthreads();  // every wave is done reading this block
        if (patch + 1 < pend) lstore(0, R0);
        __syncthreads();
    }
    if (do_bias) bias_grad_flush<T, COT, WP_THREADS>(reinterpret_cast<float*>(lds), tid, bsum, p.bias_grad, co0, p.cout);
    const int r = lane & 31, h = lane >> 5;
    const int ci = ci0 + r;
    if (ci < p.cin_total) {
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            float* dst = p.partial + (((int64_t)split * 9 + wave * 3 + kx) * w_rows) * p.cin_total;
#pragma unroll
            for (int c = 0; c < COT; ++c)
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const int co = co0 + 32 * c + (j & 3) + 8 * (j >> 2) + 4 * h;
                    if (co < w_rows) dst[(int64_t)co * p.cin_total + ci] = acc[kx][c][j];
                }
        }
    }
}

// First layer (Cin = 3) weight gradient, bf16 gout: dW[co][c][tap] = sum_p gout[p][co] * x[c][p + tap] straight from the PLANAR f32
// image (variant 6) -- the generic kernels need an NHWC copy of the image padded to 32 channels (a 67 MB conversion per step
// for 3 real channels).  GEMM view: D[co 32][k 32] += A[co][p] B[p][k], k = c*9 + tap (27 used): A fragments are the dense
// kernel's transposed gout reads, B fragments eight consecutive image columns (f32 -> bf16) of the lane's (c, tap) row in
// the LDS patch.  Four waves split the eight 16-position K steps of a 4x32 block; partial sums are reduced through LDS and
// written as a standard [tap][co][cin_pad] slab (the batched reduce un-pads it).
#define WC3_THREADS 256
template <typename T>
__global__ __launch_bounds__(WC3_THREADS) void wgrad3x3_c3_kernel(const falnet_wgrad_t p, int w_rows, int tiles_x, int tiles_y,
                                                                  int patches_per_split) {
    constexpr int PITCH = 64, SEGS = 4;
    constexpr int G_BYTES = WP_TH * WP_TW * PITCH, X_FLOATS = 3 * (WP_TH + 2) * WP_PW;
    constexpr int G_LOADS = WP_TH * WP_TW * SEGS, G_SLOTS = (G_LOADS + WC3_THREADS - 1) / WC3_THREADS;
    constexpr int X_SLOTS = (X_FLOATS + WC3_THREADS - 1) / WC3_THREADS;
    constexpr int BUF_BYTES = G_BYTES + ((X_FLOATS * 4 + 15) / 16) * 16;
    __shared__ __attribute__((aligned(16))) char lds[2 * BUF_BYTES > 4 * 32 * 33 * 4 ? 2 * BUF_BYTES : 4 * 32 * 33 * 4];
    auto Gbuf = [&](int b) -> char* { return lds + b * BUF_BYTES; };
    auto Xbuf = [&](int b) -> float* { return reinterpret_cast<float*>(lds + b * BUF_BYTES + G_BYTES); };
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int split = blockIdx.z;
    const float* x = reinterpret_cast<const float*>(p.src[0].ptr);
    const int64_t HW = (int64_t)p.IH * p.IW;
    const int npatch = p.B * tiles_x * tiles_y;
    const int pbeg = split * patches_per_split, pend = min(pbeg + patches_per_split, npatch);

    struct Regs { uint4 g[G_SLOTS]; float xv[X_SLOTS]; };
    auto gload = [&](int patch, Regs& R) {
        int q = patch;
        const int tix = q % tiles_x;
        q /= tiles_x;
        const int tiy = q % tiles_y;
        const int b = q / tiles_y;
        const int y0 = tiy * WP_TH, x0 = tix * WP_TW;
        const T* gbase = reinterpret_cast<const T*>(p.gout) + ((int64_t)b * p.TH * p.TW) * p.gC;
#pragma unroll
        for (int u = 0; u < G_SLOTS; ++u) {
            const int idx = tid + u * WC3_THREADS;
            const int seg = idx % SEGS, pix = idx / SEGS;
            const int y = y0 + pix / WP_TW, xx = x0 + pix % WP_TW;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (idx < G_LOADS && y < p.TH && xx < p.TW) v = *reinterpret_cast<const uint4*>(gbase + ((int64_t)y * p.TW + xx) * p.gC + seg * 8);
            R.g[u] = v;
        }
#pragma unroll
        for (int u = 0; u < X_SLOTS; ++u) {
            const int idx = tid + u * WC3_THREADS;
            const int c = idx / ((WP_TH + 2) * WP_PW), rem = idx % ((WP_TH + 2) * WP_PW);
            const int vy = y0 - 1 + rem / WP_PW, vx = x0 - 1 + rem % WP_PW;
            R.xv[u] = (idx < X_FLOATS && vy >= 0 && vy < p.IH && vx >= 0 && vx < p.IW) ? x[((int64_t)b * 3 + c) * HW + (int64_t)vy * p.IW + vx] : 0.f;
        }
    };
    auto lstore = [&](int buf, const Regs& R) {
#pragma unroll
        for (int u = 0; u < G_SLOTS; ++u) {
            const int idx = tid + u * WC3_THREADS;
            if (idx < G_LOADS) *reinterpret_cast<uint4*>(Gbuf(buf) + idx * 16) = R.g[u];
        }
#pragma unroll
        for (int u = 0; u < X_SLOTS; ++u) {
            const int idx = tid + u * WC3_THREADS;
            if (idx < X_FLOATS) Xbuf(buf)[idx] = R.xv[u];
        }
    };

    f32x16 acc;
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = 0.f;
    const int i16 = lane & 15, g16 = lane >> 4;
    const int kh = g16 >> 1, cb = g16 & 1, q4 = i16 >> 2, pc = i16 & 3;
    typedef s16x4 __attribute__((address_space(3))) * lds_v4;
    const int lane_off = (kh * 8 + q4) * PITCH + (cb * 16 + pc * 4) * 2;
    const int r = lane & 31, h = lane >> 5;
    const bool kvalid = r < 27;
    const int kc = r / 9, kt = r % 9;
    const int koff = (kc * (WP_TH + 2) + kt / 3) * WP_PW + kt % 3;  // patch offset of this lane's (channel, tap)

    const bool do_bias = p.bias_grad != nullptr;
    float bsum[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) bsum[i] = 0.f;
    Regs R0;
    if (pbeg < pend) {
        gload(pbeg, R0);
        lstore(0, R0);
    }
    __syncthreads();
    for (int patch = pbeg; patch < pend; ++patch) {
        const int cur = (patch - pbeg) & 1;
        if (patch + 1 < pend) gload(patch + 1, R0);
        const char* gl = Gbuf(cur) + lane_off;
        const float* X = Xbuf(cur);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int ks = wave * 2 + kk;  // 16-position K step: block row ks>>1, half ks&1
            s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(gl + ks * 16 * PITCH));
            s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(gl + ks * 16 * PITCH + 4 * PITCH));
            const s16x8_t av = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
            s16x8_t bv;
            const float* xr = X + koff + (ks >> 1) * WP_PW + (ks & 1) * 16 + h * 8;
#pragma unroll
            for (int j = 0; j < 8; ++j) bv[j] = (short)H16<T>::bits(kvalid ? xr[j] : 0.f);
            acc = H16<T>::mma(av, bv, acc);
        }
        if (do_bias) bias_grad_accumulate<T, 1, WC3_THREADS>(Gbuf(cur), tid, bsum);
        if (patch + 1 < pend) lstore(cur ^ 1, R0);
        __syncthreads();
    }
    if (do_bias) {
        bias_grad_flush<T, 1, WC3_THREADS>(reinterpret_cast<float*>(lds), tid, bsum, p.bias_grad, 0, p.cout);
        __syncthreads();
    }
    // sum the four waves' partial tiles through LDS ([wave][co][k], pitch 33), then one thread per (co, k)
    float* red = reinterpret_cast<float*>(lds);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int co = (j & 3) + 8 * (j >> 2) + 4 * h;
        red[(wave * 32 + co) * 33 + r] = acc[j];
    }
    __syncthreads();
    for (int e = tid; e < 32 * 27; e += WC3_THREADS) {
        const int co = e / 27, k = e % 27;
        const float v = red[(0 * 32 + co) * 33 + k] + red[(1 * 32 + co) * 33 + k] + red[(2 * 32 + co) * 33 + k] + red[(3 * 32 + co) * 33 + k];
        const int c = k / 9, t = k % 9;
        if (co < w_rows) p.partial[(((int64_t)split * 9 + t) * w_rows + co) * p.cin_total + c] = v;
    }
}

// partial [nsplit][ntaps][w_rows][cin_total] -> OIHW f32, un-padding the (possibly two-group) channel axis
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial, int nsplit, int ntaps,
                                                           int w_rows, int cin_total, float* __restrict__ grad, int cout,
                                                           int cin, int c0_real, int c0_pad, int use_atomics) {
    // block = (output channel co, 64 packed input channels); blockIdx.z = slab group.  Reads are coalesced along
    // the packed channel axis; the [tap][ci] -> [ci][tap] transposition goes through LDS so that the OIHW
    // writes are contiguous runs of ntaps*64 floats.
    __shared__ float tile[64 * 9];
    const int co = blockIdx.x, cp0 = blockIdx.y * 64;
    const int ngroups = gridDim.z, grp = blockIdx.z;
    const int s0 = (int)((int64_t)nsplit * grp / ngroups), s1 = (int)((int64_t)nsplit * (grp + 1) / ngroups);
    const int64_t slab = (int64_t)ntaps * w_rows * cin_total;
    for (int e = threadIdx.x; e < ntaps * 64; e += blockDim.x) {
        const int t = e / 64, cl = e % 64;
        float s = 0.f;
        if (cp0 + cl < cin_total) {
            const float* src = partial + ((int64_t)t * w_rows + co) * cin_total + cp0 + cl;
            for (int k = s0; k < s1; ++k) s += src[k * slab];
        }
        tile[cl * ntaps + t] = s;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < ntaps * 64; e += blockDim.x) {
        const int cl = e / ntaps, t = e % ntaps;
        const int cp = cp0 + cl;
        int ci = -1;
        if (cp < c0_pad) {
            if (cp < c0_real) ci = cp;
        } else if (c0_real + (cp - c0_pad) < cin) {
            ci = c0_real + (cp - c0_pad);
        }
        if (ci >= 0) {
            float* dst = grad + ((int64_t)co * cin + ci) * ntaps + t;
            if (use_atomics) atomicAdd(dst, tile[e]);
            else *dst = tile[e];
        }
    }
}

// db[c] += sum_p g[p, c]: every thread owns one 8-channel segment (16-B bf16 / 32-B f32 loads) and strides over
// pixels; rows of threads are summed through LDS, one atomic per channel per block.
template <typename T>
__global__ __launch_bounds__(256) void bias_grad_kernel(const T* __restrict__ g, int64_t npix, int gC, int cout,
                                                        float* __restrict__ db) {
    __shared__ float red[256 * 8];
    const int segs = gC / 8;                       // gC is a multiple of 32
    const int spb = segs < 256 ? segs : 256;       // segments handled per block pass
    const int rows = 256 / spb;
    const int sl = threadIdx.x % spb, rr = threadIdx.x / spb;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int seg = blockIdx.y * spb + sl; seg < segs; seg += gridDim.y * spb) {
        for (int64_t pix = (int64_t)blockIdx.x * rows + rr; pix < npix; pix += (int64_t)gridDim.x * rows) {
            Vec8<T> v;
            v.load(g + pix * gC + seg * 8);
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] += v.get(i);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) red[threadIdx.x * 8 + i] = acc[i];
        __syncthreads();
        if (rr == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float t = 0.f;
                for (int k = 0; k < rows; ++k) t += red[(k * spb + sl) * 8 + i];
                if (seg * 8 + i < cout) atomicAdd(db + seg * 8 + i, t);
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = 0.f;
    }
}

// entry lookup for the batched kernels: the block_begin column is fetched by n threads in parallel into LDS (a serial
// walk of the global table cost ~0.5 us per entry per block)
template <typename D>
__device__ __forceinline__ int find_entry(const D* __restrict__ descs, int n, int* sh /* >= 64 ints */) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) sh[i] = descs[i].block_begin;
    __syncthreads();
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if ((int)blockIdx.x >= sh[mid]) lo = mid;
        else hi = mid - 1;
    }
    return lo;
}

// ---- batched forms: ONE launch reduces the split-K slabs of every layer / sums every bias gradient ----------
// (a step has ~34 weight tensors and 13 biases; per-layer launches are launch-latency bound)
// Slab reduce, streaming form.  A block owns `cob` consecutive output channels of one layer (cob * cin_total <= 1024 packed
// input channels: for a fixed tap they are ONE contiguous run of the slab) and one group of slabs: thread i sums float4 i of
// that run for all NT taps over the group's slabs -- every wave load is 1 KiB contiguous, NT (x2: two slabs per trip)
// independent 16-B loads in flight per thread -- then the [tap][co][ci] sums are transposed through LDS ([co][ci][tap],
// stride-NT stores: odd stride, conflict-free) and leave as contiguous runs of the OIHW gradient: plain STORES when the
// block is the only writer (groups == 1 and the launch does not accumulate), 256-B-contiguous f32 atomics otherwise.
// (The previous form gave every block ONE output channel and 64 input channels: 256-B runs per load, 64 x 9 scalar atomics
// per block -- 3.5 TB/s on 758 MB of slabs.)
template <int NT>
__device__ __forceinline__ void wgrad_reduce_body(const falnet_reduce_t& d, int rel, float* __restrict__ tile /* 1024 * NT floats */, int accumulate) {
    const int cob = d.cin_total >= 1024 ? 1 : 1024 / d.cin_total;
    const int cblocks = (d.cout + cob - 1) / cob;
    const int grp = rel % d.groups, cb = rel / d.groups;
    if (cb >= cblocks) return;
    const int co0 = cb * cob, nco = min(cob, d.cout - co0);
    const int s0 = (int)((int64_t)d.nsplit * grp / d.groups), s1 = (int)((int64_t)d.nsplit * (grp + 1) / d.groups);
    const int64_t tapstride = (int64_t)d.w_rows * d.cin_total, slab = (int64_t)NT * tapstride;
    const int run = nco * d.cin_total;  // floats per tap of this block (a multiple of 32)
    for (int base = 0; base < run; base += 1024) {  // (one trip unless cin_total > 1024)
        const int i4 = base + threadIdx.x * 4;
        float4 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i4 < run) {
            const float* src = d.partial + (int64_t)co0 * d.cin_total + i4;
            int k = s0;
            for (; k + 2 <= s1; k += 2) {
                float4 v[2][NT];
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int t = 0; t < NT; ++t) v[u][t] = *reinterpret_cast<const float4*>(src + (k + u) * slab + t * tapstride);
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        acc[t].x += v[u][t].x;
                        acc[t].y += v[u][t].y;
                        acc[t].z += v[u][t].z;
                        acc[t].w += v[u][t].w;
                    }
            }
            for (; k < s1; ++k) {
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const float4 v = *reinterpret_cast<const float4*>(src + k * slab + t * tapstride);
                    acc[t].x += v.x;
                    acc[t].y += v.y;
                    acc[t].z += v.z;
                    acc[t].w += v.w;
                }
            }
            const int l = threadIdx.x * 4;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                tile[(l + 0) * NT + t] = acc[t].x;
                tile[(l + 1) * NT + t] = acc[t].y;
                tile[(l + 2) * NT + t] = acc[t].z;
                tile[(l + 3) * NT + t] = acc[t].w;
            }
        }
        __syncthreads();
        const int nloc = min(1024, run - base);  // packed (co, ci) pairs of this trip
        for (int e = threadIdx.x; e < nloc * NT; e += blockDim.x) {
            const int l = e / NT, t = e - l * NT;
            const int g = base + l;
            const int col = g / d.cin_total, cp = g - col * d.cin_total;
            int ci = -1;
            if (cp < d.c0_pad) {
                if (cp < d.c0_real) ci = cp;
            } else if (d.c0_real + (cp - d.c0_pad) < d.cin) {
                ci = d.c0_real + (cp - d.c0_pad);
            }
            if (ci >= 0) {
                float* dst = d.grad + ((int64_t)(co0 + col) * d.cin + ci) * NT + t;
                // no-return atomics pipeline; a read-modify-write would serialise one memory round trip per element
                if (d.groups > 1 || accumulate) atomicAdd(dst, tile[e]);
                else *dst = tile[e];
            }
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void wgrad_reduce_batched_kernel(const falnet_reduce_t* __restrict__ descs, int n, int accumulate) {
    __shared__ float tile[1024 * 9];
    __shared__ int entry_begin[64];
    const int li = find_entry(descs, n, entry_begin);
    const falnet_reduce_t d = descs[li];
    const int rel = blockIdx.x - d.block_begin;
    if (d.ntaps == 9) wgrad_reduce_body<9>(d, rel, tile, accumulate);
    else if (d.ntaps == 3) wgrad_reduce_body<3>(d, rel, tile, accumulate);
    else if (d.ntaps == 1) wgrad_reduce_body<1>(d, rel, tile, accumulate);
}

// Deterministic form: `ws` != nullptr -> every block writes its per-channel sums to ws[blockIdx.x][512] (plain stores) and
// bias_grad_finish_kernel adds them in block order (the atomic form's result depends on the order its blocks arrive in).
__global__ __launch_bounds__(512) void bias_grad_finish_kernel(const falnet_biasgrad_t* __restrict__ descs, const float* __restrict__ ws) {
    const falnet_biasgrad_t d = descs[blockIdx.x];
    const int c = threadIdx.x;
    if (c >= d.cout) return;
    float s = 0.f;
    for (int k = 0; k < d.blocks; ++k) s += ws[(int64_t)(d.block_begin + k) * 512 + c];
    d.db[c] += s;
}

template <typename T>
__global__ __launch_bounds__(256) void bias_grad_batched_kernel(const falnet_biasgrad_t* __restrict__ descs, int n, float* __restrict__ ws = nullptr) {
    __shared__ float red[256 * 8];
    __shared__ int entry_begin[64];
    const int li = find_entry(descs, n, entry_begin);
    const falnet_biasgrad_t d = descs[li];
    const int bx = blockIdx.x - d.block_begin, nbx = d.blocks;
    const T* g = reinterpret_cast<const T*>(d.g);
    const int segs = d.gC / 8;
    const int spb = segs < 256 ? segs : 256;
    const int rows = 256 / spb;
    const int sl = threadIdx.x % spb, rr = threadIdx.x / spb;
    for (int seg = sl; seg < segs; seg += spb) {
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        int64_t pix = (int64_t)bx * rows + rr;
        const int64_t stride = (int64_t)nbx * rows;
        for (; pix + 7 * stride < d.npix; pix += 8 * stride) {  // eight independent 16-B loads in flight
            Vec8<T> v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j].load(g + (pix + j * stride) * d.gC + seg * 8);
#pragma unroll
            for (int i = 0; i < 8; ++i)
                acc[i] += ((v[0].get(i) + v[1].get(i)) + (v[2].get(i) + v[3].get(i))) + ((v[4].get(i) + v[5].get(i)) + (v[6].get(i) + v[7].get(i)));
        }
        for (; pix < d.npix; pix += stride) {
            Vec8<T> v;
            v.load(g + pix * d.gC + seg * 8);
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] += v.get(i);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) red[threadIdx.x * 8 + i] = acc[i];
        __syncthreads();
        if (rr == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float t = 0.f;
                for (int k = 0; k < rows; ++k) t += red[(k * spb + sl) * 8 + i];
                if (seg * 8 + i < d.cout) {
                    if (ws) ws[(int64_t)blockIdx.x * 512 + seg * 8 + i] = t;
                    else atomicAdd(d.db + seg * 8 + i, t);
                }
            }
        }
        __syncthreads();
    }
}

// One block = one 32(cout) x 32(packed cin) tile of one layer, all taps, staged through LDS: the OIHW reads are
// runs of taps*32 contiguous floats, the wf rows ([co][tap][32 cin]) and wd rows ([cin][tap][32 cout]) are written as
// 32 contiguous elements.  (The element-per-thread version gathered with stride `taps` and scattered 2-byte writes.)
// ADAM: the tile's master weights are UPDATED while they are loaded (torch.optim.Adam, Train_Stage1_K.py:177-180: the same arithmetic as
// losses.hip: adam_dev_kernel) -- every real (co, ci, tap) element of a layer belongs to exactly one tile, so the optimiser step of all
// packed layers and their re-pack are ONE pass over the masters (the stand-alone re-pack read the 68 MB Adam had just written again).
struct PackAdam {
    int64_t g_off, m_off, v_off;  // element offsets from a master weight to its gradient / first / second moment (the flat buffers share one layout)
    float b1, b2, eps, grad_scale, step_size, rsqrt_bc2;
};
template <typename T, int TAPS, bool ADAM = false>
__device__ __forceinline__ void pack_tile(const falnet_pack_t& d, int rel, float (&tile)[32][32 * 9 + 1], const PackAdam* ad = nullptr) {
    const int ctiles = d.cin_pad / 32;
    const int co0 = (rel / ctiles) * 32, cp0 = (rel % ctiles) * 32;
    constexpr int rowlen = 32 * TAPS;
    // packed columns cp0..cp0+31 map to a contiguous run of real channels (group boundaries are multiples of 32)
    const int ci0 = cp0 < d.c0_pad ? cp0 : d.c0_real + (cp0 - d.c0_pad);
    const int ci_end = cp0 < d.c0_pad ? d.c0_real : d.cin;   // exclusive bound of valid real channels for this tile
    for (int e = threadIdx.x; e < 32 * rowlen; e += 256) {
        const int r = e / rowlen, k = e % rowlen;            // r: cout row of the tile, k = cil*TAPS + t
        const int co = co0 + r, ci = ci0 + k / TAPS;
        float val = 0.f;
        if (co < d.cout && ci < ci_end) {
            float* wp = const_cast<float*>(d.w) + ((int64_t)co * d.cin + ci0) * TAPS + k;
            val = *wp;
            if constexpr (ADAM) {
                const float gr = wp[ad->g_off] * ad->grad_scale;
                const float m = ad->b1 * wp[ad->m_off] + (1.f - ad->b1) * gr;
                const float v = ad->b2 * wp[ad->v_off] + (1.f - ad->b2) * gr * gr;
                val -= ad->step_size * m / (sqrtf(v) * ad->rsqrt_bc2 + ad->eps);
                *wp = val;
                wp[ad->m_off] = m;
                wp[ad->v_off] = v;
            }
        }
        tile[r][k] = val;
    }
    __syncthreads();
    T* wf = reinterpret_cast<T*>(d.wf);
    T* wd = reinterpret_cast<T*>(d.wd);
    // eight consecutive channels per thread and store (16 B in the 16-bit types): the first version stored one element per lane -- 2-B scalar
    // stores, 128 B per wave instruction -- and ran the 204 MB of the step's re-pack at 2 TB/s
    for (int e = threadIdx.x; e < 32 * TAPS * 4; e += 256) {
        const int g = e & 3, t = (e >> 2) % TAPS, r = e / (4 * TAPS);
        float vf[8], vd[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            vf[j] = tile[r][(8 * g + j) * TAPS + t];   // r = cout row, channels 8 g + j of the cin tile
            vd[j] = tile[8 * g + j][r * TAPS + t];     // r = cin row, channels 8 g + j of the cout tile
        }
        T* pf = wf ? wf + ((int64_t)(co0 + r) * TAPS + t) * d.cin_pad + cp0 + 8 * g : nullptr;
        T* pd = wd ? wd + ((int64_t)(cp0 + r) * TAPS + t) * d.cout_pad + co0 + 8 * g : nullptr;
        if constexpr (sizeof(T) == 2) {
            uint4 of, od;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                (&of.x)[k] = pack16x2<T>(vf[2 * k], vf[2 * k + 1]);
                (&od.x)[k] = pack16x2<T>(vd[2 * k], vd[2 * k + 1]);
            }
            if (pf) *reinterpret_cast<uint4*>(pf) = of;
            if (pd) *reinterpret_cast<uint4*>(pd) = od;
        } else {
            if (pf) {
                reinterpret_cast<float4*>(pf)[0] = make_float4(vf[0], vf[1], vf[2], vf[3]);
                reinterpret_cast<float4*>(pf)[1] = make_float4(vf[4], vf[5], vf[6], vf[7]);
            }
            if (pd) {
                reinterpret_cast<float4*>(pd)[0] = make_float4(vd[0], vd[1], vd[2], vd[3]);
                reinterpret_cast<float4*>(pd)[1] = make_float4(vd[4], vd[5], vd[6], vd[7]);
            }
        }
    }
}

// One block = one 32(cout) x 32(packed cin) tile of one layer, all taps, staged through LDS: the OIHW reads are
// runs of taps*32 contiguous floats, the wf rows ([co][tap][32 cin]) and wd rows ([cin][tap][32 cout]) are written as
// 32 contiguous elements.  taps is 9, 3 or 1 (compile-time divisions).
template <typename T>
__global__ __launch_bounds__(256) void pack_weights_batched_kernel(const falnet_pack_t* __restrict__ descs, int n) {
    __shared__ float tile[32][32 * 9 + 1];
    __shared__ int entry_begin[64];
    const int li = find_entry(descs, n, entry_begin);
    const falnet_pack_t d = descs[li];
    const int rel = blockIdx.x - d.block_begin;
    if (d.taps == 9) pack_tile<T, 9>(d, rel, tile);
    else if (d.taps == 3) pack_tile<T, 3>(d, rel, tile);  // 3x1 / 1x3 (FAL_netA.py:73-76)
    else pack_tile<T, 1>(d, rel, tile);
}

template <typename T>
__global__ __launch_bounds__(256) void adam_pack_batched_kernel(const falnet_pack_t* __restrict__ descs, int n, int64_t g_off, int64_t m_off, int64_t v_off,
                                                                const float* __restrict__ state, float b1, float b2, float eps, float grad_scale,
                                                                const float* __restrict__ scaler) {
    __shared__ float tile[32][32 * 9 + 1];
    __shared__ int entry_begin[64];
    if (scaler != nullptr) {
        if (scaler[2] != 0.f) return;  // non-finite gradient somewhere: the whole update is skipped, the packed copies stay valid (grid-uniform)
        grad_scale /= scaler[0];
    }
    const float t = state[1] + 1.0f;
    PackAdam ad;
    ad.g_off = g_off, ad.m_off = m_off, ad.v_off = v_off;
    ad.b1 = b1, ad.b2 = b2, ad.eps = eps, ad.grad_scale = grad_scale;
    ad.step_size = state[0] / (1.0f - powf(b1, t));
    ad.rsqrt_bc2 = rsqrtf(1.0f - powf(b2, t));
    const int li = find_entry(descs, n, entry_begin);
    const falnet_pack_t d = descs[li];
    const int rel = blockIdx.x - d.block_begin;
    if (d.no_update) {  // derived weights (their factors were updated by falnet_adam_ranges and re-composed before this launch)
        if (d.taps == 9) pack_tile<T, 9>(d, rel, tile);
        else if (d.taps == 3) pack_tile<T, 3>(d, rel, tile);
        else pack_tile<T, 1>(d, rel, tile);
    } else if (d.taps == 9) pack_tile<T, 9, true>(d, rel, tile, &ad);
    else if (d.taps == 3) pack_tile<T, 3, true>(d, rel, tile, &ad);
    else pack_tile<T, 1, true>(d, rel, tile, &ad);
}

// Sub-pixel weights of the deconv layers (conv_dma.hip: conv3x3_up2_dma_kernel): wu[co][pair][ci], pair = 4 (2 py + px) + 2 a + b
template <typename T>
__global__ __launch_bounds__(256) void pack_up2_batched_kernel(const falnet_pack_up2_t* __restrict__ descs, int n) {
    __shared__ int entry_begin[64];
    const int li = find_entry(descs, n, entry_begin);
    const falnet_pack_up2_t d = descs[li];
    const int rel = blockIdx.x - d.block_begin;
    const int ncb = d.cin_pad / 32;
    const int co0 = (rel / ncb) * 32, ci0 = (rel % ncb) * 32;
    T* wu = reinterpret_cast<T*>(d.wu);
    // one (co, ci) weight per thread and pass: its nine taps are loaded once and feed all sixteen (class, tap) sums (the first version looped over
    // the 16384 outputs of the block with up to four dependent loads each: 40 us per step for three layers)
    for (int e = threadIdx.x; e < 32 * 32; e += blockDim.x) {
        const int ci = ci0 + (e & 31), co = co0 + (e >> 5);
        float w[9];
        const bool real = co < d.cout && ci < d.cin;
#pragma unroll
        for (int t = 0; t < 9; ++t) w[t] = real ? d.w[((int64_t)co * d.cin + ci) * 9 + t] : 0.f;
#pragma unroll
        for (int pair = 0; pair < 16; ++pair) {
            const int cls = pair >> 2, a = (pair >> 1) & 1, b = pair & 1, py = cls >> 1, px = cls & 1;
            // 3x3 taps that coincide on low-resolution neighbour a (rows) / b (columns) for output parity py / px
            const int ky0 = py == 0 ? (a == 0 ? 0 : 1) : (a == 0 ? 0 : 2), ky1 = py == 0 ? (a == 0 ? 0 : 2) : (a == 0 ? 1 : 2);
            const int kx0 = px == 0 ? (b == 0 ? 0 : 1) : (b == 0 ? 0 : 2), kx1 = px == 0 ? (b == 0 ? 0 : 2) : (b == 0 ? 1 : 2);
            float v = 0.f;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
                    if (ky >= ky0 && ky <= ky1 && kx >= kx0 && kx <= kx1) v += w[ky * 3 + kx];
            wu[((int64_t)co * 16 + pair) * d.cin_pad + ci] = from_f32<T>(v);
        }
        if (d.wdd) {
            // data-gradient form on the low-resolution grid (conv_dma.hip: conv2x2_up2d_dma16_kernel): wdd[ci][2 du + dv][e 2 Cp + f Cp + co], the
            // coefficient of upstream pixel (2 (i + du) - 1 + e, 2 (j + dv) - 1 + f) in input position (i, j): per axis t = 2 d + parity selects the
            // 3x3 taps {2}, {1, 2}, {0, 1}, {0}
            T* wdd = reinterpret_cast<T*>(d.wdd);
#pragma unroll
            for (int tap = 0; tap < 4; ++tap)
#pragma unroll
                for (int ef = 0; ef < 4; ++ef) {
                    const int ty = 2 * (tap >> 1) + (ef >> 1), tx = 2 * (tap & 1) + (ef & 1);
                    const int ky0 = ty == 0 ? 2 : (ty == 1 ? 1 : 0), ky1 = ty == 0 ? 2 : (ty == 1 ? 2 : (ty == 2 ? 1 : 0));
                    const int kx0 = tx == 0 ? 2 : (tx == 1 ? 1 : 0), kx1 = tx == 0 ? 2 : (tx == 1 ? 2 : (tx == 2 ? 1 : 0));
                    float v = 0.f;
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx)
                            if (ky >= ky0 && ky <= ky1 && kx >= kx0 && kx <= kx1) v += w[ky * 3 + kx];
                    wdd[((int64_t)ci * 4 + tap) * (4 * d.cout_pad) + ef * d.cout_pad + co] = from_f32<T>(v);
                }
        }
    }
}

extern "C" int falnet_pack_up2_batched(const falnet_pack_up2_t* descs_dev, int n, int total_blocks, int dtype, void* stream) {
    FALNET_ENTER(stream);
    FALNET_CHECK_ARG(descs_dev && n > 0 && n <= 64 && total_blocks > 0, "pack_up2_batched: bad argument");
    FALNET_CHECK_ARG(dtype == FALNET_BF16 || dtype == FALNET_F16, "pack_up2_batched: 16-bit operand types only");
#define PACKU_L(T) hipLaunchKernelGGL(pack_up2_batched_kernel<T>, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, descs_dev, n)
    FALNET_DISPATCH_16(dtype, PACKU_L);
#undef PACKU_L
    FALNET_RETURN_LAUNCH();
}

extern "C" int falnet_pack_weights_batched(const falnet_pack_t* descs_dev, int n, int total_blocks, int dtype, void* stream) {
    FALNET_ENTER(stream);
    FALNET_CHECK_ARG(descs_dev && n > 0 && n <= 64 && total_blocks > 0, "pack_weights_batched: bad argument (taps must be 9, 3 or 1, n <= 64)");
#define PACK_B(T) hipLaunchKernelGGL(pack_weights_batched_kernel<T>, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, descs_dev, n)
    FALNET_DISPATCH_DTYPE(dtype, PACK_B);
#undef PACK_B
    FALNET_RETURN_LAUNCH();
}

extern "C" int falnet_adam_pack_batched(const falnet_pack_t* descs_dev, int n, int total_blocks, int dtype, int64_t g_off, int64_t m_off, int64_t v_off,
                                        const float* state, float b1, float b2, float eps, float grad_scale, const float* scaler, void* stream) {
    FALNET_ENTER(stream);
    FALNET_CHECK_ARG(descs_dev && n > 0 && n <= 64 && total_blocks > 0 && state, "adam_pack_batched: bad argument (n <= 64)");
    FALNET_CHECK_ARG(g_off != 0 && m_off != 0 && v_off != 0 && g_off != m_off && g_off != v_off && m_off != v_off,
                     "adam_pack_batched: gradient / moment buffers must be distinct from the weights and from each other");
#define APACK_B(T) hipLaunchKernelGGL(adam_pack_batched_kernel<T>, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, descs_dev, n, g_off, m_off, v_off, \
                                      state, b1, b2, eps, grad_scale, scaler)
    FALNET_DISPATCH_DTYPE(dtype, APACK_B);
#undef APACK_B
    FALNET_RETURN_LAUNCH();
}

extern "C" int falnet_wgrad_reduce_blocks(int cout, int cin_total, int groups) {
    if (cout <= 0 || cin_total <= 0 || groups <= 0) return -1;
    const int cob = cin_total >= 1024 ? 1 : 1024 / cin_total;
    return (cout + cob - 1) / cob * groups;
}

extern "C" int falnet_wgrad_reduce_batched(const falnet_reduce_t* descs_dev, int n, int total_blocks, int accumulate, void* stream) {
    FALNET_ENTER(stream);
    FALNET_CHECK_ARG(descs_dev && n > 0 && total_blocks > 0, "wgrad_reduce_batched: bad argument");
    hipLaunchKernelGGL(wgrad_reduce_batched_kernel, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, descs_dev, n, accumulate ? 1 : 0);
    FALNET_RETURN_LAUNCH();
}

extern "C" int falnet_bias_grad_batched(const falnet_biasgrad_t* descs_dev, int n, int total_blocks, int dtype, void* stream) {
    FALNET_ENTER(stream);
    FALNET_CHECK_ARG(descs_dev && n > 0 && total_blocks > 0, "bias_grad_batched: bad argument");
    FALNET_CHECK_ARG(!falnet_deterministic(), "bias_grad_batched: f32 atomics -- use falnet_bias_grad_batched_det in deterministic mode");
#define BIAS_B(T) hipLaunchKernelGGL(bias_grad_batched_kernel<T>, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, descs_dev, n, (float*)nullptr)
    FALNET_DISPATCH_DTYPE(dtype, BIAS_B);
#undef BIAS_B
    FALNET_RETURN_LAUNCH();
}

extern "C" int falnet_bias_grad_batched_det(const falnet_biasgrad_t* descs_dev, int n, int total_blocks, int dtype, float* ws, int64_t ws_floats,
                                            void* stream) {
    FALNET_ENTER(stream);
    FALNET_CHECK_ARG(descs_dev && n > 0 && total_blocks > 0 && ws, "bias_grad_batched_det: bad argument");
    FALNET_CHECK_ARG(ws_floats >= (int64_t)total_blocks * 512, "bias_grad_batched_det: workspace of %lld floats needed (512 per block)", (long long)total_blocks * 512);
#define BIAS_B(T) hipLaunchKernelGGL(bias_grad_batched_kernel<T>, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, descs_dev, n, ws)
    FALNET_DISPATCH_DTYPE(dtype, BIAS_B);
#undef BIAS_B
    hipLaunchKernelGGL(bias_grad_finish_kernel, dim3(n), dim3(512), 0, (hipStream_t)stream, descs_dev, (const float*)ws);
    FALNET_RETURN_LAUNCH();
}

// OIHW f32 -> packed operands (see falnet_hip.h)
template <typename T>
__global__ __launch_bounds__(256) void pack_weights_kernel(const float* __restrict__ w, int cout, int cin, int taps,
                                                           int c0_real, int c0_pad, int cin_pad, int cout_pad,
                                                           T* __restrict__ wf, T* __restrict__ wd) {
    const int64_t total = (int64_t)cout_pad * taps * cin_pad;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int cp = (int)(i % cin_pad), t = (int)((i / cin_pad) % taps), co = (int)(i / ((int64_t)cin_pad * taps));
        int ci = -1;
        if (cp < c0_pad) {
            if (cp < c0_real) ci = cp;
        } else if (c0_real + (cp - c0_pad) < cin) {
            ci = c0_real + (cp - c0_pad);
        }
        const float v = (co < cout && ci >= 0) ? w[((int64_t)co * cin + ci) * taps + t] : 0.f;
        if (wf) wf[i] = from_f32<T>(v);
        if (wd) wd[((int64_t)cp * taps + t) * cout_pad + co] = from_f32<T>(v);
    }
}

// ------------------------------------------------------------------------------------------ C-ABI
static int check_src(const falnet_src_t& s, int kc, const char* who) {
    FALNET_CHECK_ARG(s.ptr && s.C > 0 && s.C % kc == 0, "%s: source channels %d must be a positive multiple of %d", who, s.C, kc);
    FALNET_CHECK_ARG(s.H > 0 && s.W > 0, "%s: empty source", who);
    FALNET_CHECK_ARG((((uintptr_t)s.ptr) & 15) == 0, "%s: source pointer must be 16-B aligned", who);
    return 0;
}

template <typename T, bool SWAP>
static void launch_conv(const falnet_conv_t& p, int bn, dim3 grid, hipStream_t st) {
    if (bn == 128)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_igemm_kernel<T, 128, SWAP>), grid, dim3(CONV_THREADS), 0, st, p);
    else if (bn == 64)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_igemm_kernel<T, 64, SWAP>), grid, dim3(CONV_THREADS), 0, st, p);
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_igemm_kernel<T, 32, SWAP>), grid, dim3(CONV_THREADS), 0, st, p);
}

// v_mfma_f32_16x16x32 forms (conv_epilogue.h: Acc16) of the weight-stationary and LDS-DMA kernels; FALNET_MFMA16=0 selects the
// 32x32x16 forms (A/B).  Launched alone with warm caches the two forms are equal (64 -> 64 @256x512: 96.0 vs 93.9 us), inside the training
// step the 16x16x32 form is faster: same-box A/B, three alternations, weight-stationary kernel only: 1216 -> 1231 pairs/s (+1.2 %),
// kernel-time sum 7.45 -> 7.35 ms.  (MI355X_MICROARCH.md, MFMA shape: the chip holds a higher clock on this shape under load.)
// The same conversion of the LDS-DMA kernel (half steps (tap, channel half), positions of a tap shared by both halves) passed every test and
// measured neutral in the step (1264 vs 1268 pairs/s), and so did the nine-tap stages of the halo-patch kernel (1258 vs 1257): not kept.
static bool falnet_mfma16_enabled() {
    static const bool on = [] { const char* e = falnet_ab_env("FALNET_MFMA16"); return !(e && e[0] == '0'); }();
    return on;
}

// The dispatcher's decision for one launch; shared by falnet_conv2d and falnet_conv2d_kernel_name.
struct ConvChoice {
    int patch;               // 1: conv3x3_patch_kernel, 0: conv_igemm_kernel, 2: conv3x3_ws_kernel (kcb = chunks)
    int bn, kcb, tps, adb, th, nwaves, flip, swap;
};

// A/B switch for tests and profiling: FALNET_DISABLE_PATCH=1 routes every launch to the gather kernel
static bool g_disable_patch = [] { const char* e = falnet_ab_env("FALNET_DISABLE_PATCH"); return e && e[0] == '1'; }();
// K bytes per chunk of the pipelined patch kernel: 128 (1 workgroup/CU), 64 (2 workgroups/CU), 0 = mode S only
static int g_patch_kcb = [] { const char* e = falnet_ab_env("FALNET_PATCH_KCB"); return e ? atoi(e) : 128; }();


bool falnet_conv_dma_applicable(const falnet_conv_t& p, int min_oh);        // conv_dma.hip
int falnet_conv_dma_launch(const falnet_conv_t& p, int flip, hipStream_t st, int th);
bool falnet_conv_dma2_applicable(const falnet_conv_t& p, int th);           // four rows per wave, 16-channel chunks: 16x32 tiles, two four-wave workgroups per CU (21) / 32x32 tiles, eight waves (22)
int falnet_conv_dma2_launch(const falnet_conv_t& p, int flip, hipStream_t st, int th);
int falnet_conv_dma16_launch(const falnet_conv_t& p, int flip, hipStream_t st, int th);  // variants 13 / 17 / 20 on v_mfma_f32_16x16x32 (variants 23 / 24 / 25)
bool falnet_conv_up2_dma_applicable(const falnet_conv_t& p);                // deconv forward in sub-pixel form (variant 18)
int falnet_conv_up2_dma_launch(const falnet_conv_t& p, hipStream_t st);
bool falnet_conv_wave32_applicable(const falnet_conv_t& p);                 // conv_wave.hip: wave-streaming 32 -> 32 channel kernel (variant 27)
int falnet_conv_wave32_launch(const falnet_conv_t& p, int flip, hipStream_t st);
bool falnet_conv_wave64p_applicable(const falnet_conv_t& p);                // conv_wave.hip: wave-streaming 64 -> (<= 4) channel kernel, planar f32 output (variant 29)
int falnet_conv_wave64p_launch(const falnet_conv_t& p, int flip, hipStream_t st);
bool falnet_conv_up2d_applicable(const falnet_conv_t& p);                   // deconv data gradient on the low-resolution grid (variant 26)
int falnet_conv_up2d_launch(const falnet_conv_t& p, hipStream_t st);
bool falnet_conv_deep_applicable(const falnet_conv_t& p);                   // maps of <= 128 positions: one-shot LDS-DMA, K slices, last-arriver epilogue (variant 19)
int falnet_conv_deep_launch(const falnet_conv_t& p, hipStream_t st);
int falnet_conv_deep_mtiles(const falnet_conv_t& p);
bool falnet_conv_s2d_dma_applicable(const falnet_conv_t* d, int n);         // four parity classes of a stride-2 data gradient in one pass
int falnet_conv_s2d_dma_launch(const falnet_conv_t* d, hipStream_t st);
bool falnet_conv_s2f_dma_applicable(const falnet_conv_t& p);                // forward 3x3 stride-2
int falnet_conv_s2f_dma_launch(const falnet_conv_t& p, hipStream_t st);

static int choose_conv_kernel(const falnet_conv_t& p, ConvChoice& c) {
    const bool planar = p.out_layout == FALNET_OUT_PLANAR_F32;
    int ctot = 0;
    for (int s = 0; s < p.nsrc; ++s) ctot += p.src[s].C;
    // (a planar f32 output is fine for the halo-patch / weight-stationary kernels: their epilogue has the pixels on the lanes)
    bool dense3x3 = p.ntaps == 9 && p.isy == 1 && p.isx == 1 && p.osy == 1 && p.osx == 1 && p.ooy == 0 && p.oox == 0 &&
                    p.TH == p.OH && p.TW == p.OW && p.TH == p.IH && p.TW == p.IW && p.TW >= 16;
    // the patch kernels walk taps in spatial order with weight tap t (forward) or 8-t (stride-1 dgrad)
    int flip = -1;
    if (dense3x3 && p.w_taps == 9) {
        bool fwd = true, bwd = true;
        for (int t = 0; t < 9; ++t) {
            const int sp = (p.tap_dy[t] + 1) * 3 + (p.tap_dx[t] + 1);
            if (p.tap_dy[t] < -1 || p.tap_dy[t] > 1 || p.tap_dx[t] < -1 || p.tap_dx[t] > 1) fwd = bwd = false;
            fwd = fwd && p.tap_w[t] == sp;
            bwd = bwd && p.tap_w[t] == 8 - sp;
        }
        flip = fwd ? 0 : (bwd ? 1 : -1);
    }
    dense3x3 = dense3x3 && flip >= 0;
    const int esz = p.dtype == FALNET_F32 ? 4 : 2;
    bool c128 = true;  // every source is a whole number of 128-B channel chunks
    for (int s = 0; s < p.nsrc; ++s) c128 = c128 && (p.src[s].C % (128 / esz) == 0);
    int variant = p.variant;
    // 11 / 12: the gather kernel with 32 / 64 output channels per workgroup (instead of the widest tile that divides the layer):
    // 4x / 2x the workgroups for the small deep layers, whose launches otherwise occupy a fraction of the 256 CUs
    int forced_bn = 0;
    if (variant == 11 || variant == 12) {
        forced_bn = variant == 11 ? 32 : 64;
        if (p.w_rows % forced_bn != 0 || p.pool_out) {
            falnet_set_error("conv2d: variant %d (gather, %d channels per workgroup) not applicable to this launch", variant, forced_bn);
            return -2;
        }
        variant = 1;
    }
    if (g_disable_patch) variant = 1;
    FALNET_CHECK_ARG((variant >= 0 && variant <= 10) || variant == 13 || variant == 15 || variant == 16 || variant == 17 || variant == 18 || variant == 19 || variant == 20 || (variant >= 21 && variant <= 27) || variant == 29, "conv2d: unknown variant %d", variant);
    if (variant == 19) {  // levels 5-6: K-sliced one-shot LDS-DMA kernel with the epilogue in the last slice (conv_dma.hip: conv3x3_deep_kernel)
        if (!falnet_conv_deep_applicable(p)) {
            falnet_set_error("conv2d: variant 19 needs a 16-bit nine-tap stride-1/2 launch on maps of at most 128 positions (128 %% (TH TW) == 0), dense NHWC output, "
                             "w_rows %% 64 == 0, ksplit = cin_total / 32 or / 64 (a multiple of 4), scratch for the partial tiles and the split-K workspace");
            return -2;
        }
        c.flip = 0;
        c.swap = 0;
        c.patch = 6;
        c.bn = 64; c.kcb = p.ksplit * 32 == p.cin_total ? 1 : 2; c.tps = 9; c.adb = 1; c.th = 0; c.nwaves = falnet_conv_deep_mtiles(p);
        return 0;
    }
    if (variant == 15) {  // LDS-DMA forward 3x3 stride-2 (conv_dma.hip)
        if (!falnet_conv_s2f_dma_applicable(p)) {
            falnet_set_error("conv2d: variant 15 needs a canonical 16-bit 3x3 stride-2 pad-1 NHWC launch (>= 8 x 32 outputs) with sources at the input size");
            return -2;
        }
        c.flip = 0;
        c.swap = 0;
        c.patch = 4;
        c.bn = 64; c.kcb = 32; c.tps = 9; c.adb = 1; c.th = 8; c.nwaves = 8;
        return 0;
    }
    if (variant == 29) {  // wave-streaming kernel for a 64-channel source and <= 4 planar-f32 output channels (conv_wave.hip: conv3x3_wave64p_kernel)
        if (!(dense3x3 && falnet_conv_wave64p_applicable(p))) {
            falnet_set_error("conv2d: variant 29 needs a 16-bit dense 3x3 stride-1 launch over ONE 64-channel source at the launch size, planar f32 output of <= 4 channels, no addend / activation-gradient operand / pool / split-K");
            return -2;
        }
        c.flip = flip;
        c.swap = 0;
        c.patch = 11;
        c.bn = 16; c.kcb = 32; c.tps = 9; c.adb = 1; c.th = 1; c.nwaves = 8;
        return 0;
    }
    if (variant == 27) {  // wave-streaming kernel for 32 -> (<= 32) channel layers at full resolution (conv_wave.hip: conv3x3_wave32_kernel)
        if (!(dense3x3 && falnet_conv_wave32_applicable(p))) {
            falnet_set_error("conv2d: variant 27 needs a 16-bit dense 3x3 stride-1 NHWC launch over ONE 32-channel source at the launch size with a 32-row packed weight, no pool / split-K");
            return -2;
        }
        c.flip = flip;
        c.swap = 0;
        c.patch = 10;
        c.bn = 32; c.kcb = 32; c.tps = 9; c.adb = 1; c.th = 1; c.nwaves = 8;
        return 0;
    }
    if (variant == 26) {  // deconv data gradient on the low-resolution grid: 2x2 taps over pair pixels of the upstream gradient (conv_dma.hip: conv2x2_up2d_dma16_kernel)
        if (!(falnet_conv_up2d_applicable(p) && p.ntaps == 4 && p.isy == 1 && p.isx == 1 && p.osy == 1 && p.osx == 1 && p.ooy == 0 && p.oox == 0 && p.TH == p.OH && p.TW == p.OW)) {
            falnet_set_error("conv2d: variant 26 needs a 16-bit NHWC launch over ONE source at exactly twice the output map (>= 16 x 32), w_taps = ntaps = 4, cin_total = 4 C, no pool / split-K");
            return -2;
        }
        c.flip = 0;
        c.swap = 0;
        c.patch = 9;
        c.bn = 64; c.kcb = 32; c.tps = 4; c.adb = 1; c.th = 16; c.nwaves = 8;
        return 0;
    }
    if (variant == 18) {  // deconv forward in sub-pixel form (conv_dma.hip: conv3x3_up2_dma_kernel)
        if (!(dense3x3 && flip == 0 && !p.addend && !p.actout && falnet_conv_up2_dma_applicable(p))) {
            falnet_set_error("conv2d: variant 18 needs a 16-bit dense 3x3 forward launch over ONE source at exactly half the launch size, weight_up2 set, no residual / activation-gradient operand");
            return -2;
        }
        c.flip = 0;
        c.swap = 0;
        c.patch = 5;
        c.bn = 32; c.kcb = 64; c.tps = 16; c.adb = 1; c.th = 16; c.nwaves = 8;
        return 0;
    }
    if (variant >= 23 && variant <= 25) {  // variants 13 / 17 / 20 on v_mfma_f32_16x16x32 (conv_dma.hip: conv3x3_dma16_kernel)
        const int th = variant == 23 ? 16 : (variant == 24 ? 4 : 8);
        if (!(dense3x3 && falnet_conv_dma_applicable(p, th)) || (th != 16 && (p.pool_out || planar)) || (planar && p.pool_out)) {
            falnet_set_error("conv2d: variant %d needs a 16-bit dense 3x3 stride-1 launch (>= %d x 32 positions%s) with sources at the launch size or half of it",
                             variant, th, th != 16 ? ", NHWC output, no fused pool" : "");
            return -2;
        }
        c.flip = flip;
        c.swap = 0;
        c.patch = 8;
        c.bn = 64; c.kcb = 32; c.tps = 9; c.adb = 1; c.th = th; c.nwaves = th == 4 ? 4 : 8;
        return 0;
    }
    if (variant == 21 || variant == 22) {  // LDS-DMA, four rows per wave, 16-channel chunks: 16x32 tiles on two four-wave workgroups per CU (21), 32x32 tiles on eight waves (22) (conv_dma.hip: conv3x3_dma2_kernel)
        const int th = variant == 21 ? 16 : 32;
        if (!(dense3x3 && falnet_conv_dma2_applicable(p, th))) {
            falnet_set_error("conv2d: variant %d needs a 16-bit dense 3x3 stride-1 NHWC launch (>= %d x 32 positions) with sources at the launch size or half of it", variant, th);
            return -2;
        }
        c.flip = flip;
        c.swap = 0;
        c.patch = 7;
        c.bn = 64; c.kcb = 16; c.tps = 9; c.adb = 1; c.th = th; c.nwaves = th / 4;
        return 0;
    }
    if (variant == 13 || variant == 17 || variant == 20) {  // LDS-DMA, double-buffered, persistent: 16x32 (13), 4x32 (17) or 8x32 (20) positions x 64 channels per workgroup (conv_dma.hip)
        const int th = variant == 17 ? 4 : (variant == 20 ? 8 : 16);
        if (!(dense3x3 && falnet_conv_dma_applicable(p, th)) || (th != 16 && p.pool_out)) {
            falnet_set_error("conv2d: variant %d needs a 16-bit dense 3x3 stride-1 launch (>= %d x 32 positions%s) with sources at the launch size or half of it",
                             variant, th, th != 16 ? ", no fused pool" : "");
            return -2;
        }
        c.flip = flip;
        c.swap = 0;
        c.patch = 3;
        c.bn = 64; c.kcb = 64; c.tps = 9; c.adb = 1; c.th = th; c.nwaves = th == 4 ? 4 : 8;
        return 0;
    }
    if (variant == 10 || variant == 16) {
        // weight-stationary persistent kernel: every K chunk of a BN-channel weight slice stays in LDS (16: two-phase form, c.adb = 1)
        bool ok = dense3x3 && ctot * esz <= 128 && ctot * esz % 64 == 0 && p.src[0].C * esz % 64 == 0;
        for (int s = 0; s < p.nsrc && ok; ++s)  // sources at the launch size or exactly 2x upsampled
            ok = (p.src[s].H == p.IH || 2 * p.src[s].H == p.IH) && (p.src[s].W == p.IW || 2 * p.src[s].W == p.IW) && p.src[s].C < 4096;
        if (!ok) {
            falnet_set_error("conv2d: variant 10 needs a dense 3x3 launch with <= 128 B of input channels per pixel");
            return -2;
        }
        c.flip = flip;
        c.swap = 0;
        c.patch = 2;
        c.bn = (p.w_rows % 64 == 0 && p.Cout > 32) ? 64 : 32;
        c.kcb = ctot * esz / 64;  // chunks
        c.tps = 9; c.adb = variant == 16; c.th = (c.bn == 64 ? 8 : 16) >> c.adb; c.nwaves = 8;
        return 0;
    }
    if (variant >= 2 && (!dense3x3 || (variant == 2 && !c128) || ((variant == 3 || variant == 5 || variant == 6 || variant == 8) && ctot / (64 / esz) < 2) ||
                         ((variant == 6 || variant == 7) && p.OH < 16))) {
        falnet_set_error("conv2d: variant %d not applicable to this launch", variant);
        return -2;
    }
    if (variant == 0) {
        if (!dense3x3 || p.TW < 32) variant = 1;
        else if (c128 && g_patch_kcb == 128) variant = 2;
        else if (ctot / (64 / esz) >= 2 && g_patch_kcb != 0) variant = 3;
        else variant = 4;
    }
    const bool bn64 = p.w_rows % 64 == 0 && p.Cout > 32;
    c.flip = flip;
    c.swap = planar ? 1 : 0;
    c.patch = variant >= 2;
    if (p.pool_out && !c.patch) {
        falnet_set_error("conv2d: the fused max pool needs a halo-patch variant (dense 3x3 stride-1 launch)");
        return -2;
    }
    if (!c.patch) {
        c.bn = forced_bn ? forced_bn : (p.w_rows % 128 == 0 && p.Cout > 64) ? 128 : (bn64 ? 64 : 32);
        c.kcb = 64; c.tps = 1; c.adb = 0; c.th = 0; c.nwaves = 4;
        return 0;
    }
    c.th = variant >= 8 ? 4 : (variant >= 6 ? 16 : 8);  // variants 8/9: 4x32-position blocks for small images (more workgroups)
    c.nwaves = (variant == 6 || variant == 7) ? 8 : 4;
    const int tiles = p.B * ((p.OW + PT_TW - 1) / PT_TW) * ((p.OH + c.th - 1) / c.th);
    const bool bn128 = p.w_rows % 128 == 0 && p.Cout > 64 && (c.th == 16 || tiles >= 256);
    switch (variant) {
        case 2: c.bn = bn128 ? 128 : (bn64 ? 64 : 32); c.kcb = 128; c.tps = 1; c.adb = ctot / (128 / esz) > 1; break;
        case 3: case 6: c.bn = bn128 ? 128 : (bn64 ? 64 : 32); c.kcb = 64; c.tps = 1; c.adb = 1; break;
        case 8: c.bn = bn64 ? 64 : 32; c.kcb = 64; c.tps = 1; c.adb = 1; break;
        case 5: c.bn = bn64 ? 64 : 32; c.kcb = 64; c.tps = 9; c.adb = 1; break;
        default: c.bn = bn64 ? 64 : 32; c.kcb = 64; c.tps = 9; c.adb = 0; break;  // 4, 7, 9
    }
    if (p.pool_out && (c.th / (c.nwaves / (c.bn >= 128 ? 2 : 1))) % 2 != 0) {
        falnet_set_error("conv2d: variant %d gives every wave an odd number of rows: no fused max pool", variant);
        return -2;
    }
    return 0;
}

// conv3x3_ws2_kernel's FULL form: both 64-B chunks from ONE source of exactly 128 B per pixel -> one load stream of whole 128-B lines
static bool ws2_whole_lines(const falnet_conv_t& p) {
    static const bool on = !(falnet_ab_env("FALNET_WS2_FULL") && atoi(falnet_ab_env("FALNET_WS2_FULL")) == 0);
    return on && p.nsrc == 1 && (int64_t)p.src[0].C * (p.dtype == FALNET_F32 ? 4 : 2) == 128;
}

// Symbol of the kernel falnet_conv2d will launch for this descriptor (the name rocprofv3 reports): lets a harness
// group launches by the real instantiation.
extern "C" int falnet_conv2d_kernel_name(const falnet_conv_t* pp, char* buf, int len) {
    FALNET_CHECK_ARG(pp && buf && len > 0, "conv2d_kernel_name: bad argument");
    ConvChoice c;
    if (int r = choose_conv_kernel(*pp, c)) return r;
    const char* t = pp->dtype == FALNET_BF16 ? "DF16b" : pp->dtype == FALNET_F16 ? "DF16_" : "f";
    if (c.patch == 11)
        snprintf(buf, len, "_Z22conv3x3_wave64p_kernelI%sEv13falnet_conv_tiiii", t);
    else if (c.patch == 10)
        snprintf(buf, len, "_Z21conv3x3_wave32_kernelI%sEv13falnet_conv_tiiii", t);
    else if (c.patch == 9)
        snprintf(buf, len, "_Z25conv2x2_up2d_dma16_kernelI%sEv13falnet_conv_tiii", t);
    else if (c.patch == 8)
        snprintf(buf, len, "_Z20conv3x3_dma16_kernelI%sLb%dELi%dELi%dELb%dEEv13falnet_conv_tiiii", t, pp->pool_out ? 1 : 0, c.th, c.nwaves, pp->out_layout == FALNET_OUT_PLANAR_F32 ? 1 : 0);
    else if (c.patch == 7)
        snprintf(buf, len, "_Z19conv3x3_dma2_kernelI%sLb%dELi%dEEv13falnet_conv_tiiiii", t, pp->pool_out ? 1 : 0, c.nwaves);
    else if (c.patch == 6)
        snprintf(buf, len, "_Z19conv3x3_deep_kernelI%sLi%dELi%dEEv13falnet_conv_t18falnet_deep_geom_t", t, c.kcb, c.nwaves);
    else if (c.patch == 5)
        snprintf(buf, len, "_Z22conv3x3_up2_dma_kernelI%sEv13falnet_conv_tiiiii", t);
    else if (c.patch == 4)
        snprintf(buf, len, "_Z22conv3x3_s2f_dma_kernelI%sLi%dEEv13falnet_conv_tiii", t, c.bn);
    else if (c.patch == 3)
        snprintf(buf, len, "_Z18conv3x3_dma_kernelI%sLi%dELi%dEEv13falnet_conv_tiiii", t, c.th, c.nwaves);
    else if (c.patch == 2)
    {
        const int m16 = (int)(falnet_mfma16_enabled() && pp->dtype != FALNET_F32);
        if (c.adb) snprintf(buf, len, "_Z18conv3x3_ws2_kernelI%sLi%dELi%dELb%dELb%dEEv13falnet_conv_tiii", t, c.bn, c.kcb, m16, (int)(c.kcb == 2 && ws2_whole_lines(*pp)));
        else snprintf(buf, len, "_Z17conv3x3_ws_kernelI%sLi%dELi%dELb%dEEv13falnet_conv_tiii", t, c.bn, c.kcb, m16);
    }
    else if (c.patch)
        snprintf(buf, len, "_Z20conv3x3_patch_kernelI%sLi%dELi%dELi%dELb%dELi%dELi%dEEv13falnet_conv_tiii", t, c.bn, c.kcb, c.tps, c.adb, c.th, c.nwaves);
    else
        snprintf(buf, len, "_Z17conv_igemm_kernelI%sLi%dELb%dEEv13falnet_conv_t", t, c.bn, c.swap);
    return 0;
}

extern "C" int falnet_conv2d(const falnet_conv_t* pp, void* stream) {
    FALNET_ENTER(stream);
    FALNET_CHECK_ARG(pp, "conv2d: null descriptor");
    const falnet_conv_t& p = *pp;
    FALNET_CHECK_ARG(p.dtype == FALNET_F32 || p.dtype == FALNET_BF16 || p.dtype == FALNET_F16, "conv2d: bad dtype %d", p.dtype);
    const int kc = p.dtype == FALNET_F32 ? 16 : 32;
    FALNET_CHECK_ARG(p.nsrc == 1 || p.nsrc == 2, "conv2d: nsrc=%d", p.nsrc);
    int ctot = 0;
    for (int s = 0; s < p.nsrc; ++s) {
        if (int r = check_src(p.src[s], kc, "conv2d")) return r;
        ctot += p.src[s].C;
    }
    FALNET_CHECK_ARG(ctot <= p.cin_total, "conv2d: sources carry %d channels, packed weight row holds %d", ctot, p.cin_total);
    FALNET_CHECK_ARG(p.ntaps >= 1 && p.ntaps <= 9 && p.w_taps >= 1, "conv2d: ntaps=%d", p.ntaps);
    for (int t = 0; t < p.ntaps; ++t) FALNET_CHECK_ARG(p.tap_w[t] >= 0 && p.tap_w[t] < p.w_taps, "conv2d: tap_w[%d] out of range", t);
    FALNET_CHECK_ARG(p.B > 0 && p.TH > 0 && p.TW > 0 && p.IH > 0 && p.IW > 0 && p.OH > 0 && p.OW > 0, "conv2d: empty shape");
    FALNET_CHECK_ARG(p.weight && (p.out || p.pool_out) && p.Cout > 0 && p.w_rows >= p.Cout && p.w_rows % 32 == 0, "conv2d: bad weight/out (Cout=%d w_rows=%d)", p.Cout, p.w_rows);
    FALNET_CHECK_ARG(!p.pool_out || (p.out_layout == FALNET_OUT_NHWC && !p.actout && p.ksplit <= 1 && (p.pool_mode == 0 || p.pool_mode == 1)),
                     "conv2d: pool_out needs a plain NHWC launch (no actout / split-K) and pool_mode 0|1");
    FALNET_CHECK_ARG((int64_t)p.B * p.OH * p.OW * (p.out_layout == FALNET_OUT_PLANAR_F32 ? p.Cout : 1) < (1ll << 31), "conv2d: output too large for 32-bit pixel index");
    const bool planar = p.out_layout == FALNET_OUT_PLANAR_F32;
    FALNET_CHECK_ARG(!planar || (!p.addend && !p.actout), "conv2d: planar output supports bias/act epilogue only");
    if (p.ksplit > 1 && p.variant != 19 && falnet_deterministic()) {  // (variant 19 sums its K slices in a fixed order)
        falnet_set_error("conv2d: split-K (f32 atomics) is not available in deterministic mode");
        return -2;
    }
    hipStream_t st = (hipStream_t)stream;
    ConvChoice c;
    if (int r = choose_conv_kernel(p, c)) return r;
    if (c.patch == 11) return falnet_conv_wave64p_launch(p, c.flip, st);
    if (c.patch == 10) return falnet_conv_wave32_launch(p, c.flip, st);
    if (c.patch == 9) return falnet_conv_up2d_launch(p, st);
    if (c.patch == 8) return falnet_conv_dma16_launch(p, c.flip, st, c.th);
    if (c.patch == 7) return falnet_conv_dma2_launch(p, c.flip, st, c.th);
    if (c.patch == 6) return falnet_conv_deep_launch(p, st);
    if (c.patch == 5) return falnet_conv_up2_dma_launch(p, st);
    if (c.patch == 4) return falnet_conv_s2f_dma_launch(p, st);
    if (c.patch == 3) return falnet_conv_dma_launch(p, c.flip, st, c.th);
    if (c.patch == 2) {
        const int ws_th = c.th;
        const int tiles_x = (p.OW + PT_TW - 1) / PT_TW, tiles_y = (p.OH + ws_th - 1) / ws_th;
        const int ny = (p.Cout + c.bn - 1) / c.bn, ntiles = p.B * tiles_x * tiles_y;
        int gx = 256 / ny;  // one persistent workgroup per CU (146 KB of LDS each)
        if (gx < 1) gx = 1;
        if (gx > ntiles) gx = ntiles;
        const dim3 grid((unsigned)gx, (unsigned)ny);
        const bool m16 = falnet_mfma16_enabled();  // v_mfma_f32_16x16x32 form (16-bit types)
        const bool ws2_full = ws2_whole_lines(p);
#define LAUNCH_WS1(K, T, BN, NCH) do { if (m16 && sizeof(T) == 2) hipLaunchKernelGGL(HIP_KERNEL_NAME(K<T, BN, NCH, true>), grid, dim3(512), 0, st, p, tiles_x, tiles_y, c.flip); \
                                       else hipLaunchKernelGGL(HIP_KERNEL_NAME(K<T, BN, NCH, false>), grid, dim3(512), 0, st, p, tiles_x, tiles_y, c.flip); } while (0)
#define LAUNCH_WS2(T, BN, NCH) do { if (NCH == 2 && ws2_full) { if (m16 && sizeof(T) == 2) hipLaunchKernelGGL(HIP_KERNEL_NAME(conv3x3_ws2_kernel<T, BN, 2, true, true>), grid, dim3(512), 0, st, p, tiles_x, tiles_y, c.flip); \
                                         else hipLaunchKernelGGL(HIP_KERNEL_NAME(conv3x3_ws2_kernel<T, BN, 2, false, true>), grid, dim3(512), 0, st, p, tiles_x, tiles_y, c.flip); } \
                                     else LAUNCH_WS1(conv3x3_ws2_kernel, T, BN, NCH); } while (0)
#define LAUNCH_WS(T, BN, NCH) do { if (c.adb) LAUNCH_WS2(T, BN, NCH); else LAUNCH_WS1(conv3x3_ws_kernel, T, BN, NCH); } while (0)
#define WS_TABLE(T)                                                                                 \
    if (c.bn == 64) { if (c.kcb == 2) LAUNCH_WS(T, 64, 2); else LAUNCH_WS(T, 64, 1); }              \
    else { if (c.kcb == 2) LAUNCH_WS(T, 32, 2); else LAUNCH_WS(T, 32, 1); }
        FALNET_DISPATCH_DTYPE(p.dtype, WS_TABLE);
#undef WS_TABLE
#undef LAUNCH_WS
#undef LAUNCH_WS1
#undef LAUNCH_WS2
        FALNET_RETURN_LAUNCH();
    }
    if (c.patch) {
        const int tiles_x = (p.OW + PT_TW - 1) / PT_TW, tiles_y = (p.OH + c.th - 1) / c.th;
        const dim3 grid((unsigned)(p.B * tiles_x * tiles_y), (unsigned)((p.Cout + c.bn - 1) / c.bn));
        bool launched = false;
#define TRY_PATCH(T, BN, KCB, TPS, ADB, TH, NW)                                                                              \
    if (!launched && c.bn == BN && c.kcb == KCB && c.tps == TPS && c.adb == (ADB ? 1 : 0) && c.th == TH) {                     \
        hipLaunchKernelGGL(HIP_KERNEL_NAME(conv3x3_patch_kernel<T, BN, KCB, TPS, ADB, TH, NW>), grid, dim3(NW * 64), 0, st, p, \
                           tiles_x, tiles_y, c.flip);                                                                          \
        launched = true;                                                                                                       \
    }
#define PATCH_TABLE(T)                                                                                   \
    TRY_PATCH(T, 128, 128, 1, true, 8, 4) TRY_PATCH(T, 64, 128, 1, true, 8, 4) TRY_PATCH(T, 32, 128, 1, true, 8, 4)      \
    TRY_PATCH(T, 128, 128, 1, false, 8, 4) TRY_PATCH(T, 64, 128, 1, false, 8, 4) TRY_PATCH(T, 32, 128, 1, false, 8, 4)   \
    TRY_PATCH(T, 128, 64, 1, true, 8, 4) TRY_PATCH(T, 64, 64, 1, true, 8, 4) TRY_PATCH(T, 32, 64, 1, true, 8, 4)         \
    TRY_PATCH(T, 64, 64, 9, false, 8, 4) TRY_PATCH(T, 32, 64, 9, false, 8, 4)                                             \
    TRY_PATCH(T, 64, 64, 9, true, 8, 4) TRY_PATCH(T, 32, 64, 9, true, 8, 4)                                               \
    TRY_PATCH(T, 128, 64, 1, true, 16, 8) TRY_PATCH(T, 64, 64, 1, true, 16, 8) TRY_PATCH(T, 32, 64, 1, true, 16, 8)      \
    TRY_PATCH(T, 64, 64, 9, false, 16, 8) TRY_PATCH(T, 32, 64, 9, false, 16, 8)                                           \
    TRY_PATCH(T, 64, 64, 1, true, 4, 4) TRY_PATCH(T, 32, 64, 1, true, 4, 4)                                               \
    TRY_PATCH(T, 64, 64, 9, false, 4, 4) TRY_PATCH(T, 32, 64, 9, false, 4, 4)
        FALNET_DISPATCH_DTYPE(p.dtype, PATCH_TABLE);
#undef PATCH_TABLE
#undef TRY_PATCH
        FALNET_CHECK_ARG(launched, "conv2d: no instantiation for bn=%d kcb=%d tps=%d adb=%d th=%d", c.bn, c.kcb, c.tps, c.adb, c.th);
        FALNET_RETURN_LAUNCH();
    }
    const int bn = c.bn;
    const int64_t M = (int64_t)p.B * p.TH * p.TW;
    int ksplit = p.ksplit > 1 ? p.ksplit : 1;
    if (ksplit > 1) {
        FALNET_CHECK_ARG(!planar && p.splitk_ws && p.Cout % 8 == 0, "conv2d: split-K needs an NHWC output and a workspace");
        FALNET_CHECK_ARG(M * p.w_rows * 4 <= p.splitk_ws_bytes, "conv2d: split-K workspace too small (%lld needed)", (long long)(M * p.w_rows * 4));
        // no memset: the workspace is zero on entry by contract (the epilogue kernel below re-zeroes what it consumes)
    }
    const dim3 grid((unsigned)((M + CONV_BM - 1) / CONV_BM), (unsigned)((p.Cout + bn - 1) / bn), (unsigned)ksplit);
#define GATHER_L(T)                                          \
    if (planar) launch_conv<T, true>(p, bn, grid, st);       \
    else launch_conv<T, false>(p, bn, grid, st);
    FALNET_DISPATCH_DTYPE(p.dtype, GATHER_L);
#undef GATHER_L
    if (ksplit > 1) {
        const int64_t total = M * (p.w_rows / 8);
        const unsigned eg = (unsigned)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
#define SPLITK_E(T) hipLaunchKernelGGL(splitk_epilogue_kernel<T>, dim3(eg), dim3(256), 0, st, p)
        FALNET_DISPATCH_DTYPE(p.dtype, SPLITK_E);
#undef SPLITK_E
    }
    FALNET_RETURN_LAUNCH();
}

extern "C" int falnet_conv3x3_c3(const float* x_nchw, const float* w_oihw, const float* bias, void* out, int B, int H, int W, int Cout,
                                 int act, int dtype, void* stream) {
    FALNET_ENTER(stream);
    FALNET_CHECK_ARG(x_nchw && w_oihw && out && B > 0 && H > 0 && W > 0, "conv3x3_c3: bad argument");
    FALNET_CHECK_ARG(Cout == 32 || Cout == 64, "conv3x3_c3: Cout must be 32 or 64 (got %d)", Cout);
    falnet_conv_t p = {};
    p.IH = p.OH = H;
    p.IW = p.OW = W;
    p.out = out;
    p.Cout = Cout;
    p.out_cstride = Cout;
    p.bias = bias;
    p.act = act;
#ifdef C3_STAMPS  // profiling build (tools/c3_stamps.py): where the kernel writes its phase stamps
    if (const char* e = getenv("FALNET_C3_STAMP_PTR")) p.pool_actout = reinterpret_cast<const void*>(strtoull(e, nullptr, 10));
#endif
    const int tiles_x = ((W + PT_TW - 1) / PT_TW + C3_TPW - 1) / C3_TPW /* groups of C3_TPW tiles */, tiles_y = (H + PT_TH - 1) / PT_TH;
    const dim3 grid((unsigned)(B * tiles_x * tiles_y));
    hipStream_t st = (hipStream_t)stream;
#define C3_L(T)                                                                                                                                          \
    if (Cout == 64) hipLaunchKernelGGL(HIP_KERNEL_NAME(conv3x3_c3_kernel<T, 2>), grid, dim3(CONV_THREADS), 0, st, x_nchw, w_oihw, p, tiles_x, tiles_y); \
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(conv3x3_c3_kernel<T, 1>), grid, dim3(CONV_THREADS), 0, st, x_nchw, w_oihw, p, tiles_x, tiles_y);
    FALNET_CHECK_ARG(dtype == FALNET_F32 || dtype == FALNET_BF16 || dtype == FALNET_F16, "conv3x3_c3: bad dtype %d", dtype);
    FALNET_DISPATCH_DTYPE(dtype, C3_L);
#undef C3_L
    FALNET_RETURN_LAUNCH();
}

extern "C" int falnet_conv2d_multi(const falnet_conv_t* descs, int n, void* stream) {
    FALNET_ENTER(stream);
    FALNET_CHECK_ARG(descs && n >= 1 && n <= 4, "conv2d_multi: 1..4 descriptors");
    if (descs[0].variant == 14) {  // fused LDS-DMA kernel for the canonical 3x3 stride-2 data gradient (conv_dma.hip)
        for (int i = 0; i < n; ++i)
            for (int s = 0; s < descs[i].nsrc && s < 2; ++s)
                if (int r = check_src(descs[i].src[s], 32, "conv2d_multi")) return r;
        FALNET_CHECK_ARG(falnet_conv_s2d_dma_applicable(descs, n), "conv2d_multi: variant 14 needs the four parity classes of one 16-bit 3x3 stride-2 data gradient");
        return falnet_conv_s2d_dma_launch(descs, (hipStream_t)stream);
    }
    if (descs[0].ksplit > 1 && falnet_deterministic()) {
        falnet_set_error("conv2d_multi: split-K (f32 atomics) is not available in deterministic mode");
        return -2;
    }
    falnet_conv4_t pp;
    int64_t maxM = 0;
    ConvChoice c0;
    for (int i = 0; i < n; ++i) {
        const falnet_conv_t& p = descs[i];
        FALNET_CHECK_ARG(p.dtype == descs[0].dtype && p.w_rows == descs[0].w_rows && p.Cout == descs[0].Cout && p.out_layout == FALNET_OUT_NHWC &&
                         p.ksplit == descs[0].ksplit && p.nsrc >= 1 && p.nsrc <= 2 && p.weight && p.out && p.ntaps >= 1 && p.ntaps <= 9,
                         "conv2d_multi: members must share dtype / Cout / w_rows / ksplit and be plain NHWC gather launches");
        if (p.ksplit > 1) {  // every member accumulates into its OWN all-zero workspace region
            const int64_t need = (int64_t)p.B * p.TH * p.TW * p.w_rows * 4;
            FALNET_CHECK_ARG(p.splitk_ws && p.Cout % 8 == 0 && need <= p.splitk_ws_bytes, "conv2d_multi: split-K needs a workspace of %lld bytes per member", (long long)need);
            for (int j = 0; j < i; ++j) {
                const char* a = (const char*)p.splitk_ws; const char* bq = (const char*)descs[j].splitk_ws;
                const int64_t needj = (int64_t)descs[j].B * descs[j].TH * descs[j].TW * descs[j].w_rows * 4;
                FALNET_CHECK_ARG(a + need <= bq || bq + needj <= a, "conv2d_multi: split-K workspace regions of members %d and %d overlap", j, i);
            }
        }
        for (int s = 0; s < p.nsrc; ++s)
            if (int r = check_src(p.src[s], p.dtype == FALNET_F32 ? 16 : 32, "conv2d_multi")) return r;
        ConvChoice c;
        falnet_conv_t q = p;
        q.variant = (p.variant == 11 || p.variant == 12) ? p.variant : 1;  // member 0's tile width is the launch's
        if (int r = choose_conv_kernel(q, c)) return r;
        if (i == 0) c0 = c;
        const int64_t M = (int64_t)p.B * p.TH * p.TW;
        maxM = M > maxM ? M : maxM;
        pp.c[i] = p;
    }
    for (int i = n; i < 4; ++i) pp.c[i] = descs[0];
    const int ksplit = descs[0].ksplit > 1 ? descs[0].ksplit : 1;
    const dim3 grid((unsigned)((maxM + CONV_BM - 1) / CONV_BM), (unsigned)((descs[0].Cout + c0.bn - 1) / c0.bn), (unsigned)(n * ksplit));
    hipStream_t st = (hipStream_t)stream;
#define LAUNCH_MULTI(T)                                                                                              \
    do {                                                                                                             \
        if (c0.bn == 128) hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_igemm_multi_kernel<T, 128>), grid, dim3(CONV_THREADS), 0, st, pp, ksplit); \
        else if (c0.bn == 64) hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_igemm_multi_kernel<T, 64>), grid, dim3(CONV_THREADS), 0, st, pp, ksplit); \
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_igemm_multi_kernel<T, 32>), grid, dim3(CONV_THREADS), 0, st, pp, ksplit);  \
    } while (0)
    FALNET_DISPATCH_DTYPE(descs[0].dtype, LAUNCH_MULTI);
#undef LAUNCH_MULTI
    if (ksplit > 1) {
        const int64_t total = maxM * (descs[0].w_rows / 8);
        const dim3 eg((unsigned)((total + 255) / 256 > 1024 ? 1024 : (total + 255) / 256), (unsigned)n);
#define SPLITK_M(T) hipLaunchKernelGGL(splitk_epilogue_multi_kernel<T>, eg, dim3(256), 0, st, pp)
        FALNET_DISPATCH_DTYPE(descs[0].dtype, SPLITK_M);
#undef SPLITK_M
    }
    FALNET_RETURN_LAUNCH();
}

static inline int round32(int v) { return (v + 31) / 32 * 32; }

extern "C" int64_t falnet_wgrad_workspace_bytes(const falnet_wgrad_t* p) {
    if (!p) return -1;
    return (int64_t)p->nsplit * p->ntaps * round32(p->gC) * p->cin_total * (int64_t)sizeof(float);
}

// kernel selection of falnet_wgrad -- ONE place, also behind falnet_wgrad_fuses_bias (the host must not re-derive it)
enum WgradKernel { WGK_BAD = -1, WGK_TAP = 0, WGK_PATCH11, WGK_PATCH12, WGK_PATCH21, WGK_S2, WGK_C3, WGK_ROWS, WGK_ROWS_S2, WGK_WAVE };
bool falnet_wgrad_rows_applicable(const falnet_wgrad_t& p);           // wgrad_rows.hip
int falnet_wgrad_rows_launch(const falnet_wgrad_t& p, hipStream_t st);
bool falnet_wgrad_rows_s2_applicable(const falnet_wgrad_t& p);        // wgrad_rows.hip: row-streaming form of the stride-2 weight gradient (variant 8)
int falnet_wgrad_rows_s2_launch(const falnet_wgrad_t& p, hipStream_t st);
bool falnet_wgrad_wave_applicable(const falnet_wgrad_t& p);           // wgrad_wave.hip: wave-streaming kernel for 32-channel inputs (variant 9)
int falnet_wgrad_wave_launch(const falnet_wgrad_t& p, hipStream_t st);
bool falnet_wgrad_c3wave_applicable(const falnet_wgrad_t& p);         // wgrad_wave.hip: the first layer's gradient in the wave-streaming form (variant 6, IW % 4 == 0)
int falnet_wgrad_c3wave_launch(const falnet_wgrad_t& p, hipStream_t st);

static bool canonical_taps9(const falnet_wgrad_t& p) {
    if (p.ntaps != 9) return false;
    for (int t = 0; t < 9; ++t)
        if (p.tap_dy[t] != t / 3 - 1 || p.tap_dx[t] != t % 3 - 1) return false;
    return true;
}

// returns the kernel; on WGK_BAD the error text is set
static WgradKernel choose_wgrad_kernel(const falnet_wgrad_t& p) {
    const int w_rows = round32(p.gC);
    const bool h16 = p.dtype == FALNET_BF16 || p.dtype == FALNET_F16;
    const bool canon = canonical_taps9(p);
    if (p.up2 && p.variant != 7) { falnet_set_error("wgrad: up2 (a deconv layer's gradient on the low-resolution grid) is a mode of variant 7 only"); return WGK_BAD; }
    if (p.variant == 6) {  // first layer: planar f32 3-channel source (src[0].ptr = [B][3][IH][IW] f32), 16-bit gout, Cout 32
        const bool ok = h16 && canon && p.isy == 1 && p.isx == 1 && p.TH == p.IH && p.TW == p.IW && p.gC == 32 && w_rows == 32 && p.cin_total == 32 && p.nsrc == 1;
        if (!ok) { falnet_set_error("wgrad: variant 6 is the Cin=3 / Cout=32 first layer in bf16 / f16 (dense 3x3, cin_total 32)"); return WGK_BAD; }
        return WGK_C3;
    }
    if (p.variant == 5) {  // stride-2 3x3 (16-bit): parity-plane halo kernel
        bool ok = h16 && canon && p.isy == 2 && p.isx == 2 && p.TW >= 16 && p.TH == (p.IH + 1) / 2 && p.TW == (p.IW + 1) / 2;
        for (int s = 0; s < p.nsrc && ok; ++s) ok = p.src[s].C % 32 == 0 && ((p.src[s].H == p.IH && p.src[s].W == p.IW) || (p.src[s].sy == 0 && p.src[s].sx == 0));
        if (!ok) { falnet_set_error("wgrad: variant 5 needs a 16-bit 3x3 stride-2 pad-1 launch with sources at the input size"); return WGK_BAD; }
        return WGK_S2;
    }
    if (p.variant == 8) {
        if (!falnet_wgrad_rows_s2_applicable(p)) { falnet_set_error("wgrad: variant 8 needs a 16-bit 3x3 stride-2 pad-1 launch with ONE source at the input size"); return WGK_BAD; }
        return WGK_ROWS_S2;
    }
    if (p.variant == 9) {
        if (!falnet_wgrad_wave_applicable(p)) { falnet_set_error("wgrad: variant 9 needs a 16-bit dense 3x3 stride-1 launch over ONE 32-channel NHWC source at the launch size, gC 32 or 64, TW >= 32"); return WGK_BAD; }
        return WGK_WAVE;
    }
    if (p.variant == 7) {
        if (!falnet_wgrad_rows_applicable(p)) { falnet_set_error("wgrad: variant 7 needs a 16-bit dense 3x3 stride-1 launch with sources at the launch size or half of it (up2: ONE source at the launch size, nsplit a multiple of 4)"); return WGK_BAD; }
        return WGK_ROWS;
    }
    // dense 3x3 stride-1 -> halo-patch kernel (one slab per workgroup; nsplit = pixel-range splits)
    const bool dense = canon && p.isy == 1 && p.isx == 1 && p.TH == p.IH && p.TW == p.IW && p.TW >= 16 && !g_disable_patch && p.variant != 1;
    if (dense) {
        if (p.variant == 3 || p.variant == 4) {  // 32 x 64 / 64 x 32 channels per workgroup (register staged, two workgroups per CU)
            const bool co2 = p.variant == 3;
            if (!(h16 && (co2 ? w_rows : p.cin_total) % 64 == 0)) { falnet_set_error("wgrad: variant %d needs 16-bit operands and a channel count that is a multiple of 64", p.variant); return WGK_BAD; }
            return co2 ? WGK_PATCH12 : WGK_PATCH21;
        }
        return WGK_PATCH11;
    }
    return WGK_TAP;
}

static int check_wgrad_desc(const falnet_wgrad_t& p) {
    FALNET_CHECK_ARG(p.dtype == FALNET_F32 || p.dtype == FALNET_BF16 || p.dtype == FALNET_F16, "wgrad: bad dtype %d", p.dtype);
    FALNET_CHECK_ARG(p.nsrc == 1 || p.nsrc == 2, "wgrad: nsrc=%d", p.nsrc);
    int ctot = 0;
    if (p.variant != 6) {  // (variant 6 reads a planar f32 3-channel image: its own checks)
        for (int s = 0; s < p.nsrc; ++s) {
            if (int r = check_src(p.src[s], 32, "wgrad")) return r;
            ctot += p.src[s].C;
        }
        FALNET_CHECK_ARG(ctot == p.cin_total, "wgrad: sources carry %d channels, cin_total=%d", ctot, p.cin_total);
    } else {
        FALNET_CHECK_ARG(p.src[0].ptr && p.src[0].C == 3, "wgrad: variant 6 needs a 3-channel planar f32 source");
    }
    FALNET_CHECK_ARG(p.gout && p.gC > 0 && p.gC % 32 == 0 && p.nsplit >= 1 && p.ntaps >= 1 && p.ntaps <= 9, "wgrad: bad argument");
    FALNET_CHECK_ARG(p.B > 0 && p.TH > 0 && p.TW > 0, "wgrad: empty shape");
    FALNET_CHECK_ARG(p.cout >= 0 && p.cout <= p.gC, "wgrad: cout=%d exceeds gC=%d", p.cout, p.gC);
    return 0;
}

static bool wgrad_kernel_fuses_bias(WgradKernel k) {
    if (falnet_deterministic()) return false;  // the fused form adds with f32 atomics from every workgroup
    return k == WGK_PATCH11 || k == WGK_PATCH12 || k == WGK_PATCH21 || k == WGK_S2 || k == WGK_C3 || k == WGK_ROWS || k == WGK_ROWS_S2 || k == WGK_WAVE;
}

extern "C" int falnet_wgrad_fuses_bias(const falnet_wgrad_t* pp) {
    if (!pp || check_wgrad_desc(*pp) != 0) return 0;
    return wgrad_kernel_fuses_bias(choose_wgrad_kernel(*pp)) ? 1 : 0;
}

extern "C" int falnet_wgrad(const falnet_wgrad_t* pp, void* stream) {
    FALNET_ENTER(stream);
    FALNET_CHECK_ARG(pp, "wgrad: null descriptor");
    falnet_wgrad_t p = *pp;
    if (int r = check_wgrad_desc(p)) return r;
    FALNET_CHECK_ARG(p.partial, "wgrad: no workspace");
    if (p.cout == 0) p.cout = p.gC;
    const int w_rows = round32(p.gC);
    const WgradKernel k = choose_wgrad_kernel(p);
    if (k == WGK_BAD) return -1;
    if (p.bias_grad && !wgrad_kernel_fuses_bias(k)) {
        falnet_set_error("wgrad: bias_grad is set but the selected kernel (%d) cannot fuse it -- ask falnet_wgrad_fuses_bias first", (int)k);
        return -3;
    }
    hipStream_t st = (hipStream_t)stream;
    const int tiles_x = (p.TW + WP_TW - 1) / WP_TW, tiles_y = (p.TH + WP_TH - 1) / WP_TH;
    const int npatch = p.B * tiles_x * tiles_y;
    const int pps = (npatch + p.nsplit - 1) / p.nsplit;
    switch (k) {
    case WGK_ROWS:
        return falnet_wgrad_rows_launch(p, st);
    case WGK_ROWS_S2:
        return falnet_wgrad_rows_s2_launch(p, st);
    case WGK_WAVE:
        return falnet_wgrad_wave_launch(p, st);
    case WGK_C3:
        if (falnet_wgrad_c3wave_applicable(p)) return falnet_wgrad_c3wave_launch(p, st);
#define WG_C3(T) hipLaunchKernelGGL(HIP_KERNEL_NAME(wgrad3x3_c3_kernel<T>), dim3(1, 1, p.nsplit), dim3(WC3_THREADS), 0, st, p, w_rows, tiles_x, tiles_y, pps)
        FALNET_DISPATCH_16(p.dtype, WG_C3);
        break;
    case WGK_S2:
#define WG_S2_2(T) hipLaunchKernelGGL(HIP_KERNEL_NAME(wgrad3x3_s2_kernel<T, 2>), dim3(p.cin_total / 32, w_rows / 64, p.nsplit), dim3(WP_THREADS), 0, st, p, w_rows, tiles_x, tiles_y, pps)
#define WG_S2_1(T) hipLaunchKernelGGL(HIP_KERNEL_NAME(wgrad3x3_s2_kernel<T, 1>), dim3(p.cin_total / 32, w_rows / 32, p.nsplit), dim3(WP_THREADS), 0, st, p, w_rows, tiles_x, tiles_y, pps)
        if (w_rows % 64 == 0) FALNET_DISPATCH_16(p.dtype, WG_S2_2);
        else FALNET_DISPATCH_16(p.dtype, WG_S2_1);
        break;
    case WGK_PATCH12:
#define WG_P12(T) hipLaunchKernelGGL(HIP_KERNEL_NAME(wgrad3x3_patch_kernel<T, 1, 2>), dim3(p.cin_total / 32, w_rows / 64, p.nsplit), dim3(WP_THREADS), 0, st, p, w_rows, tiles_x, tiles_y, pps)
        FALNET_DISPATCH_16(p.dtype, WG_P12);
        break;
    case WGK_PATCH21:
#define WG_P21(T) hipLaunchKernelGGL(HIP_KERNEL_NAME(wgrad3x3_patch_kernel<T, 2, 1>), dim3(p.cin_total / 64, w_rows / 32, p.nsplit), dim3(WP_THREADS), 0, st, p, w_rows, tiles_x, tiles_y, pps)
        FALNET_DISPATCH_16(p.dtype, WG_P21);
        break;
    case WGK_PATCH11:
#define WG_P11(T) hipLaunchKernelGGL(HIP_KERNEL_NAME(wgrad3x3_patch_kernel<T, 1, 1>), dim3(p.cin_total / 32, w_rows / 32, p.nsplit), dim3(WP_THREADS), 0, st, p, w_rows, tiles_x, tiles_y, pps)
        FALNET_DISPATCH_DTYPE(p.dtype, WG_P11);
        break;
    default: {
        const dim3 grid((p.cin_total + WG_BN - 1) / WG_BN, (w_rows + WG_BM - 1) / WG_BM, p.ntaps * p.nsplit);
#define WG_TAP(T) hipLaunchKernelGGL(wgrad_kernel<T>, grid, dim3(CONV_THREADS), 0, st, p, w_rows)
        FALNET_DISPATCH_DTYPE(p.dtype, WG_TAP);
    }
    }
    FALNET_RETURN_LAUNCH();
}

extern "C" int falnet_wgrad_reduce(const float* partial, int nsplit, int ntaps, int cout_pad, int cin_total, float* grad,
                                   int cout, int cin, int c0_real, int c0_pad, int accumulate, void* stream) {
    FALNET_ENTER(stream);
    FALNET_CHECK_ARG(partial && grad && nsplit >= 1 && ntaps >= 1 && ntaps <= 9 && cout > 0 && cin > 0 && cout <= cout_pad, "wgrad_reduce: bad argument");
    FALNET_CHECK_ARG(c0_real <= cin && c0_real <= c0_pad && c0_pad + (cin - c0_real) <= cin_total, "wgrad_reduce: channel groups do not fit");
    // slab groups: enough blocks to fill the chip when the weight tensor is small and the slab count large
    const int blocks = cout * ((cin_total + 63) / 64);
    int groups = 1;
    if (nsplit >= 16 && blocks < 1024) groups = (1024 + blocks - 1) / blocks;
    if (groups > nsplit / 8) groups = nsplit / 8 > 0 ? nsplit / 8 : 1;
    if (falnet_deterministic()) groups = 1;  // one writer per gradient element: slabs summed in slab order
    const int use_atomics = (groups > 1 || accumulate) ? 1 : 0;
    if (groups > 1 && !accumulate) {
        hipError_t e = hipMemsetAsync(grad, 0, sizeof(float) * (size_t)cout * cin * ntaps, (hipStream_t)stream);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(cout, (cin_total + 63) / 64, groups), dim3(256), 0, (hipStream_t)stream, partial,
                       nsplit, ntaps, cout_pad, cin_total, grad, cout, cin, c0_real, c0_pad, use_atomics);
    FALNET_RETURN_LAUNCH();
}

extern "C" int falnet_bias_grad(const void* g, int64_t npix, int gC, int cout, float* db, int accumulate, int dtype,
                                void* stream) {
    FALNET_ENTER(stream);
    FALNET_CHECK_ARG(g && db && npix > 0 && cout > 0 && cout <= gC, "bias_grad: bad argument");
    FALNET_CHECK_ARG(gC % 32 == 0 && gC <= 2048, "bias_grad: unsupported channel count %d", gC);
    if (!accumulate) {
        hipError_t e = hipMemsetAsync(db, 0, sizeof(float) * cout, (hipStream_t)stream);
        if (e != hipSuccess) return (int)e;
    }
    const int segs = gC / 8, spb = segs < 256 ? segs : 256, rows = 256 / spb;
    int64_t gx = (npix + rows * 16 - 1) / (rows * 16);
    gx = gx < 1 ? 1 : (gx > 512 ? 512 : gx);
    if (falnet_deterministic()) gx = 1;  // one block = one add per channel (slow; the batched _det form is the training path)
    const dim3 grid((unsigned)gx, 1);
#define BIAS_L(T) hipLaunchKernelGGL(bias_grad_kernel<T>, grid, dim3(256), 0, (hipStream_t)stream, (const T*)g, npix, gC, cout, db)
    FALNET_DISPATCH_DTYPE(dtype, BIAS_L);
#undef BIAS_L
    FALNET_RETURN_LAUNCH();
}

extern "C" int falnet_pack_weights(const float* w_oihw, int cout, int cin, int taps, int c0_real, int c0_pad,
                                   int cin_pad_total, int cout_pad, void* wf, void* wd, int dtype, void* stream) {
    FALNET_ENTER(stream);
    FALNET_CHECK_ARG(w_oihw && (wf || wd) && cout > 0 && cin > 0 && taps >= 1, "pack_weights: bad argument");
    FALNET_CHECK_ARG(cout_pad >= cout && cout_pad % 32 == 0 && cin_pad_total % 32 == 0, "pack_weights: pads must be multiples of 32");
    FALNET_CHECK_ARG(c0_real <= cin && c0_real <= c0_pad && c0_pad + (cin - c0_real) <= cin_pad_total, "pack_weights: channel groups do not fit");
    const int64_t total = (int64_t)cout_pad * taps * cin_pad_total;
    const int grid = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
#define PACK_L(T) hipLaunchKernelGGL(pack_weights_kernel<T>, dim3(grid), dim3(256), 0, (hipStream_t)stream, w_oihw, cout, cin, taps, \
                                   c0_real, c0_pad, cin_pad_total, cout_pad, (T*)wf, (T*)wd)
    FALNET_DISPATCH_DTYPE(dtype, PACK_L);
#undef PACK_L
    FALNET_RETURN_LAUNCH();
}
