// Direct (register) epilogue of the kernels that keep the PIXEL on the lane (MFMA operands exchanged: A = weights, B = pixels):
// shared by conv.hip (halo-patch, weight-stationary, first-layer kernels) and conv_dma.hip.
#pragma once
#include <type_traits>
#include "common.h"

typedef f32x16_t f32x16;

// ELU / ReLU in the fewest VALU issue slots (the epilogue arithmetic of the 64-channel full-resolution layers is ~11 % of their launch):
//   elu(v) = v > 0 ? v : exp(v) - 1 = max(v, min(exp(v), 1) - 1)   [v > 0: the clamped exponential is 1, max(v, 0) = v; v <= 0: exp(v) <= 1 and
//   exp(v) - 1 >= v] -- the same values as the select form, with the min as the CLAMP output modifier of v_exp_f32 (fmed3(x, 0, 1) folds into it)
//   and one v_max instead of v_cmp + v_cndmask; the v_max is written in asm so hipcc does not put a canonicalising v_max v, v in front of it.
__device__ __forceinline__ float max_raw(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float elu_f(float v) {
    const float e = __builtin_amdgcn_fmed3f(__builtin_amdgcn_exp2f(v * 1.44269504088896340736f), 0.f, 1.f);
    return max_raw(v, e - 1.f);
}
__device__ __forceinline__ float relu_f(float v) { return max_raw(v, 0.f); }
// d elu / dx from the OUTPUT y (>= -1): y > 0 ? 1 : y + 1 = clamp(y + 1, 0, 1) (the clamp modifier of the v_add)
__device__ __forceinline__ float elu_grad_from_out(float y) { return __builtin_amdgcn_fmed3f(y + 1.f, 0.f, 1.f); }
__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == FALNET_ACT_ELU) return elu_f(v);
    if (act == FALNET_ACT_RELU) return relu_f(v);
    return v;
}
__device__ __forceinline__ float act_grad_from_out(float y, int kind) {
    if (kind == FALNET_ACT_ELU) return elu_grad_from_out(y);
    if (kind == FALNET_ACT_RELU) return y > 0.f ? 1.f : 0.f;
    return 1.f;
}


struct NoPool { static constexpr bool enabled = false; __device__ int64_t operator()(int, int) const { return -1; } };

// ---- direct epilogue (halo-patch and first-layer kernels) ---------------------------------------------------------
// With the MFMA operands exchanged (A = weights, B = pixels) the 32x32 C/D tile puts one PIXEL on each lane and sixteen
// channels in its accumulators: acc[j] = channel 8*(j>>2) + 4*h + (j&3) of the tile, i.e. four groups of four consecutive
// channels, lane halves h = 0/1 interleaved.  f32: every group is one 16-B store.  bf16: a group is 8 B; one
// v_permlane32_swap per dword exchanges the upper half's group k with the lower half's group k+1, after which lanes 0-31
// hold channels 8k..8k+7 and lanes 32-63 channels 8k+8..8k+15 of their pixel: one 16-B store per group pair
// (cdna_hip_programming.md T21).  No LDS staging, no wave barriers; residual / activation-output operands are read in
// the same 16-B chunks and un-swapped with the same (involutive) exchange.
__device__ __forceinline__ void half_swap(unsigned& a, unsigned& b) {  // lanes 32-63 of a <-> lanes 0-31 of b
    const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    a = r[0];
    b = r[1];
}
// this lane's 16 channels (accumulator order) of pixel offset o (-1: none) from an NHWC tensor; cbase = first channel of the tile
template <typename T>
__device__ __forceinline__ void tile_load(const T* __restrict__ base, int64_t o, int cbase, int h, int Cout, float (&v)[16]) {
    if constexpr (sizeof(T) == 2) {
        uint4 c[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int cb = cbase + 16 * q + 8 * h;
            c[q] = (o >= 0 && cb < Cout) ? *reinterpret_cast<const uint4*>(base + o + cb) : make_uint4(0, 0, 0, 0);
            half_swap(c[q].x, c[q].z);
            half_swap(c[q].y, c[q].w);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned w0 = (k & 1) ? c[k >> 1].z : c[k >> 1].x, w1 = (k & 1) ? c[k >> 1].w : c[k >> 1].y;
            v[4 * k + 0] = H16<T>::lo(w0);
            v[4 * k + 1] = H16<T>::hi(w0);
            v[4 * k + 2] = H16<T>::lo(w1);
            v[4 * k + 3] = H16<T>::hi(w1);
        }
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int cb = cbase + 8 * k + 4 * h;
            const float4 f = (o >= 0 && cb < Cout) ? *reinterpret_cast<const float4*>(base + o + cb) : make_float4(0.f, 0.f, 0.f, 0.f);
            v[4 * k + 0] = f.x;
            v[4 * k + 1] = f.y;
            v[4 * k + 2] = f.z;
            v[4 * k + 3] = f.w;
        }
    }
}
// The same load in two steps: tile_fetch issues this lane's loads WITHOUT a branch (an invalid lane reads element 0 of the tensor and is
// zeroed in tile_unpack), so that all loads of a tile -- and of the next one -- are in flight before the first s_waitcnt; the conditional form
// above compiles to one exec-masked block per 16-B load, each followed by s_waitcnt vmcnt(0).
template <typename T>
struct TileRaw {
    static constexpr int N = sizeof(T) == 2 ? 2 : 4;
    uint4 c[N];
    unsigned okmask;
};
template <typename T>
__device__ __forceinline__ void tile_fetch(const T* __restrict__ base, int64_t o, int cbase, int h, int Cout, TileRaw<T>& r) {
    r.okmask = 0;
#pragma unroll
    for (int q = 0; q < TileRaw<T>::N; ++q) {
        const int cb = sizeof(T) == 2 ? cbase + 16 * q + 8 * h : cbase + 8 * q + 4 * h;
        const bool ok = o >= 0 && cb < Cout;
        r.okmask |= ok ? 1u << q : 0u;
        r.c[q] = *reinterpret_cast<const uint4*>(base + (ok ? o + cb : (int64_t)0));
    }
}
template <typename T>
__device__ __forceinline__ void tile_unpack(TileRaw<T>& r, float (&v)[16]) {
#pragma unroll
    for (int q = 0; q < TileRaw<T>::N; ++q)
        if (!((r.okmask >> q) & 1)) r.c[q] = make_uint4(0, 0, 0, 0);
    if constexpr (sizeof(T) == 2) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            half_swap(r.c[q].x, r.c[q].z);
            half_swap(r.c[q].y, r.c[q].w);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned w0 = (k & 1) ? r.c[k >> 1].z : r.c[k >> 1].x, w1 = (k & 1) ? r.c[k >> 1].w : r.c[k >> 1].y;
            v[4 * k + 0] = H16<T>::lo(w0);
            v[4 * k + 1] = H16<T>::hi(w0);
            v[4 * k + 2] = H16<T>::lo(w1);
            v[4 * k + 3] = H16<T>::hi(w1);
        }
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            v[4 * k + 0] = __uint_as_float(r.c[k].x);
            v[4 * k + 1] = __uint_as_float(r.c[k].y);
            v[4 * k + 2] = __uint_as_float(r.c[k].z);
            v[4 * k + 3] = __uint_as_float(r.c[k].w);
        }
    }
}
template <typename T>
__device__ __forceinline__ void tile_store(T* __restrict__ base, int64_t o, int cbase, int h, int Cout, const float (&v)[16]) {
    if constexpr (sizeof(T) == 2) {
        unsigned w[4][2];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            w[k][0] = pack16x2<T>(v[4 * k + 0], v[4 * k + 1]);
            w[k][1] = pack16x2<T>(v[4 * k + 2], v[4 * k + 3]);
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            half_swap(w[2 * q][0], w[2 * q + 1][0]);
            half_swap(w[2 * q][1], w[2 * q + 1][1]);
            const int cb = cbase + 16 * q + 8 * h;
            if (o >= 0 && cb < Cout) *reinterpret_cast<uint4*>(base + o + cb) = make_uint4(w[2 * q][0], w[2 * q][1], w[2 * q + 1][0], w[2 * q + 1][1]);
        }
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int cb = cbase + 8 * k + 4 * h;
            if (o >= 0 && cb < Cout) *reinterpret_cast<float4*>(base + o + cb) = make_float4(v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]);
        }
    }
}
// A finished 32 x 32 tile as it goes to memory, still in accumulator order (no lane-half exchange yet): what a wave keeps when the
// arithmetic and the stores of its epilogue run in different phases (conv3x3_ws2_kernel).
template <typename T>
struct PackedTile {
    static constexpr int N = sizeof(T) == 2 ? 8 : 16;
    unsigned w[N];
};
template <typename T>
__device__ __forceinline__ void tile_pack(const float (&v)[16], PackedTile<T>& t) {
    if constexpr (sizeof(T) == 2) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            t.w[2 * k + 0] = pack16x2<T>(v[4 * k + 0], v[4 * k + 1]);
            t.w[2 * k + 1] = pack16x2<T>(v[4 * k + 2], v[4 * k + 3]);
        }
    } else {
#pragma unroll
        for (int j = 0; j < 16; ++j) t.w[j] = __float_as_uint(v[j]);
    }
}
template <typename T>
__device__ __forceinline__ void tile_store_packed(T* __restrict__ base, int64_t o, int cbase, int h, int Cout, PackedTile<T>& t) {
    if constexpr (sizeof(T) == 2) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            half_swap(t.w[4 * q + 0], t.w[4 * q + 2]);
            half_swap(t.w[4 * q + 1], t.w[4 * q + 3]);
            const int cb = cbase + 16 * q + 8 * h;
            if (o >= 0 && cb < Cout) *reinterpret_cast<uint4*>(base + o + cb) = make_uint4(t.w[4 * q + 0], t.w[4 * q + 1], t.w[4 * q + 2], t.w[4 * q + 3]);
        }
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int cb = cbase + 8 * k + 4 * h;
            if (o >= 0 && cb < Cout) *reinterpret_cast<uint4*>(base + o + cb) = make_uint4(t.w[4 * k], t.w[4 * k + 1], t.w[4 * k + 2], t.w[4 * k + 3]);
        }
    }
}
template <typename T, int MT, int NT>
struct PackedOut {  // epilogue_direct<..., DEFER = true> fills it, epilogue_store_packed writes it out
    PackedTile<T> out[MT][NT];
    PackedTile<T> pool[(MT + 1) / 2][NT];
};

// ---- 16x16x32 MFMA form of a 32 (channels) x 32 (positions) tile -------------------------------------------------------------
// v_mfma_f32_16x16x32 holds the chip's clock higher than 32x32x16 at equal cycles per FLOP (MI355X_MICROARCH.md, MFMA shape: 1.12-1.15x
// on random data; measured here by issuing the same operands through two 16x16x32: weight-stationary kernel 850 -> 1040 TFLOP/s).
// A 32x32 tile becomes 2 (channel halves ct) x 2 (position halves pt) MFMAs, K = 32 each; lane (p = lane & 15, g = lane >> 4) supplies
// row / column p and K block g of an operand and receives D[4g + f][p] in register f.  To land in epilogue_direct's layout (position
// 16 pt + p on lane 32 h + 16 pt + p, register 4 m + f = channel 8 m + 4 h + f) WITHOUT a data-dependent shuffle network:
//   * the weight operand of half ct puts channel 16 ct + 8 e + 4 h' + f on its row i = 8 h' + 4 e + f (m16_row_channel below), so lane row
//     g of the result holds the registers of lane half h' = g >> 1, group e = g & 1;
//   * one v_permlane16_swap per register exchanges the odd lane rows of the pt = 0 result with the even lane rows of the pt = 1 result,
//     after which every lane holds values of ITS OWN position only: four swaps per (ct, 32x32 tile).
__device__ __forceinline__ int m16_row_channel(int i) { return ((i >> 2) & 1) * 8 + (i >> 3) * 4 + (i & 3); }  // within a 16-channel half
struct Acc16 {        // one 32x32 tile as four 16x16 results
    f32x4_t t[2][2];  // [ct][pt]
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int pt = 0; pt < 2; ++pt) t[ct][pt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
    __device__ __forceinline__ void to32(f32x16_t& v) const {  // -> the 32x32x16 accumulator layout epilogue_direct expects
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(t[ct][0][f]), __float_as_uint(t[ct][1][f]), false, false);
                v[4 * (2 * ct) + f] = __uint_as_float(r[0]);
                v[4 * (2 * ct + 1) + f] = __uint_as_float(r[1]);
            }
    }
};

// activation / activation-gradient on one 16-value tile: the (workgroup-uniform) kind is switched ONCE per tile
__device__ __forceinline__ void act16(float (&v)[16], int act) {
    if (act == FALNET_ACT_ELU) {
#pragma unroll
        for (int j0 = 0; j0 < 16; j0 += 8) {  // stage by stage over eight values: packed multiplies / adds, no wait state behind the exponentials
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            f32x2 e[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) e[j] = f32x2{v[j0 + 2 * j], v[j0 + 2 * j + 1]} * 1.44269504088896340736f;  // v_pk_mul_f32
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                e[j].x = __builtin_amdgcn_fmed3f(__builtin_amdgcn_exp2f(e[j].x), 0.f, 1.f);
                e[j].y = __builtin_amdgcn_fmed3f(__builtin_amdgcn_exp2f(e[j].y), 0.f, 1.f);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) e[j] = e[j] - 1.f;  // v_pk_add_f32
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v[j0 + 2 * j] = max_raw(v[j0 + 2 * j], e[j].x);
                v[j0 + 2 * j + 1] = max_raw(v[j0 + 2 * j + 1], e[j].y);
            }
        }
    } else if (act == FALNET_ACT_RELU) {
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = relu_f(v[j]);
    }
}
__device__ __forceinline__ void actgrad16(float (&v)[16], const float (&y)[16], int kind) {
    if (kind == FALNET_ACT_ELU) {
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] *= elu_grad_from_out(y[j]);
    } else if (kind == FALNET_ACT_RELU) {
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = y[j] > 0.f ? v[j] : 0.f;
    }
}
// bias of this lane's 16 channels per 32-channel tile (accumulator order)
template <int NT>
__device__ __forceinline__ void load_bias16(const falnet_conv_t& p, int nbase, int h, float (&bias)[NT][16]) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int c = nbase + nt * 32 + 8 * (j >> 2) + 4 * h + (j & 3);
            bias[nt][j] = (p.bias && c < p.Cout) ? p.bias[c] : 0.f;
        }
}
// The same values from a copy of the workgroup's channel block in LDS (stage_bias_lds once per kernel, visible behind the kernel's first barrier): the
// persistent LDS-DMA kernels fetch the bias per TILE and slice -- registers that must not stay live across the MFMA loop -- and sixteen global loads
// per lane in front of every epilogue were 1-1.5 k cycles of exposed latency per slice; four ds_read_b128 are ~100.
__device__ __forceinline__ void stage_bias_lds(const falnet_conv_t& p, int n0, int bn, float* lds_bias) {
    for (int i = threadIdx.x; i < bn; i += blockDim.x) lds_bias[i] = (p.bias && n0 + i < p.Cout) ? p.bias[n0 + i] : 0.f;
}
__device__ __forceinline__ void load_bias16_lds(const float* lds_bias, int cb, int h, float (&bias)[1][16]) {  // cb: slice offset inside the staged block
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const float4 v = *reinterpret_cast<const float4*>(lds_bias + cb + 8 * m + 4 * h);
        bias[0][4 * m + 0] = v.x;
        bias[0][4 * m + 1] = v.y;
        bias[0][4 * m + 2] = v.z;
        bias[0][4 * m + 3] = v.w;
    }
}
__device__ __forceinline__ float lane_xor1(float v) {  // value of lane ^ 1 (DPP quad_perm [1,0,3,2])
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true));
}
// PixOff(mt) -> element offset of channel 0 of this lane's pixel in slab mt (or -1); PoolOff(mt) (odd mt, even column)
// -> offset of the 2x2-reduced pixel in p.pool_out (or -1).  Slabs are consecutive image rows, lanes consecutive columns.
// AHEAD = operand slabs in flight beyond the current one (1: conv.hip's kernels; -1: the LDS-DMA kernels, which have no registers to spare:
// conditional loads at the point of use)
// DEFER: nothing is stored; the finished tiles go to *defer (NHWC outputs only), epilogue_store_packed writes them later.
// NHWC_ONLY: the planar-f32 output form is compiled out (kernels whose dispatcher admits NHWC launches only): its sixteen 64-bit store addresses are
// ~60 registers at a kernel's high-water mark even when the branch is never taken.
// PLANAR_PTR: the planar-f32 stores walk ONE running 64-bit pointer (channel stride added between stores, pinned by an empty asm) instead of sixteen
// independent addresses: 2 address registers instead of 32 (conv3x3_dma16_kernel's planar instantiation spilled 136 B per lane without it).
template <typename T, int MT, int NT, typename PixOff, typename PoolOff = NoPool, int AHEAD = 1, bool DEFER = false, bool NHWC_ONLY = false, bool PLANAR_PTR = false>
__device__ __forceinline__ void epilogue_direct(const falnet_conv_t& p, f32x16 (&acc)[MT][NT], const float (&bias)[NT][16], int nbase, int lane,
                                                PixOff pixoff, PoolOff pooloff = PoolOff(), PackedOut<T, MT, NT>* defer = nullptr) {
    constexpr bool POOL = !std::is_same<PoolOff, NoPool>::value;
    const int h = lane >> 5;
    const T* addend = reinterpret_cast<const T*>(p.addend);
    const T* actout = reinterpret_cast<const T*>(p.actout);
    T* out = reinterpret_cast<T*>(p.out);
    T* pool_out = reinterpret_cast<T*>(p.pool_out);
    const bool pooling = POOL && pool_out != nullptr;
    const bool psum = p.pool_mode == 1;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int cbase = nbase + nt * 32;
        float hp[16];
        // residual / activation-output operands: fetched one row slab AHEAD (branch-free loads, conv_epilogue.h: tile_fetch), so their latency
        // sits behind the previous slab's arithmetic and stores instead of in front of every 16-B load
        constexpr int NR = AHEAD < 0 ? 1 : 1 + AHEAD;
        TileRaw<T> ra[NR], rb[NR];
        if (AHEAD > 0) {
            if (addend) tile_fetch<T>(addend, pixoff(0), cbase, h, p.Cout, ra[0]);  // workgroup-uniform branches: the half swaps inside need every lane
            if (actout) tile_fetch<T>(actout, pixoff(0), cbase, h, p.Cout, rb[0]);
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int64_t o = pixoff(mt);
            if (AHEAD >= 0 && (AHEAD ? mt + 1 < MT : true)) {
                const int mf = mt + AHEAD;
                if (addend) tile_fetch<T>(addend, pixoff(mf), cbase, h, p.Cout, ra[mf & AHEAD]);
                if (actout) tile_fetch<T>(actout, pixoff(mf), cbase, h, p.Cout, rb[mf & AHEAD]);
            }
            float v[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = acc[mt][nt][j] + bias[nt][j];
            if (addend) {
                float a[16];
                if constexpr (AHEAD < 0) tile_load<T>(addend, o, cbase, h, p.Cout, a);
                else tile_unpack<T>(ra[mt & (AHEAD < 0 ? 0 : AHEAD)], a);
#pragma unroll
                for (int j = 0; j < 16; ++j) v[j] += a[j];
            }
            act16(v, p.act);
            if (actout) {
                float a[16];
                if constexpr (AHEAD < 0) tile_load<T>(actout, o, cbase, h, p.Cout, a);
                else tile_unpack<T>(rb[mt & (AHEAD < 0 ? 0 : AHEAD)], a);
                actgrad16(v, a, p.actout_kind);
            }
            if constexpr (DEFER) {
                tile_pack<T>(v, defer->out[mt][nt]);
            } else if (!NHWC_ONLY && p.out_layout == FALNET_OUT_PLANAR_F32) {
                // planar f32 [B][Cout][OH][OW] (the MED logits): o = offset of channel 0 of this lane's pixel, channel stride
                // OH*OW; the 32 lanes of a half are consecutive columns -> 128-B runs per channel
                float* po = reinterpret_cast<float*>(p.out);
                const int64_t cs = (int64_t)p.OH * p.OW;
                if constexpr (PLANAR_PTR) {
                    if (o >= 0) {
                        float* q = po + o + (int64_t)(cbase + 4 * h) * cs;
#pragma unroll
                        for (int j = 0; j < 16; ++j) {
                            const int c = cbase + 8 * (j >> 2) + 4 * h + (j & 3);
                            asm volatile("" : "+v"(q));  // keep ONE address: hipcc otherwise materialises all sixteen up front
                            if (c < p.Cout) *q = v[j];
                            q += ((j & 3) == 3) ? 5 * cs : cs;  // next channel; after four, the lane half's next group of four (8 channels on)
                        }
                    }
                } else if (o >= 0) {
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        const int c = cbase + 8 * (j >> 2) + 4 * h + (j & 3);
                        if (c < p.Cout) po[o + c * cs] = v[j];
                    }
                }
            } else if (!POOL || out) tile_store<T>(out, o, cbase, h, p.Cout, v);
            if constexpr (POOL) {
                if (pooling) {
                    float m[16];
                    if (psum) {
#pragma unroll
                        for (int j = 0; j < 16; ++j) m[j] = v[j] + lane_xor1(v[j]);  // column neighbour (same half, same row)
                    } else {
#pragma unroll
                        for (int j = 0; j < 16; ++j) m[j] = fmaxf(v[j], lane_xor1(v[j]));
                    }
                    if ((mt & 1) == 0) {
#pragma unroll
                        for (int j = 0; j < 16; ++j) hp[j] = m[j];
                    } else {
                        if (psum) {
#pragma unroll
                            for (int j = 0; j < 16; ++j) m[j] += hp[j];
                        } else {
#pragma unroll
                            for (int j = 0; j < 16; ++j) m[j] = fmaxf(m[j], hp[j]);
                        }
                        const int64_t po = (lane & 1) ? (int64_t)-1 : pooloff(mt);
                        if (p.pool_actout) {
                            float a[16];
                            tile_load<T>(reinterpret_cast<const T*>(p.pool_actout), po, cbase, h, p.Cout, a);
                            actgrad16(m, a, p.pool_actout_kind);
                        }
                        if constexpr (DEFER) tile_pack<T>(m, defer->pool[mt >> 1][nt]);
                        else tile_store<T>(pool_out, po, cbase, h, p.Cout, m);
                    }
                }
            }
        }
    }
}

// Second half of a deferred epilogue: lane-half exchange + 16-B stores of the tiles epilogue_direct<..., DEFER> left in `d`.
template <typename T, int MT, int NT, typename PixOff, typename PoolOff>
__device__ __forceinline__ void epilogue_store_packed(const falnet_conv_t& p, PackedOut<T, MT, NT>& d, int nbase, int lane, PixOff pixoff, PoolOff pooloff) {
    const int h = lane >> 5;
    T* out = reinterpret_cast<T*>(p.out);
    T* pool_out = reinterpret_cast<T*>(p.pool_out);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int cbase = nbase + nt * 32;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            if (out) tile_store_packed<T>(out, pixoff(mt), cbase, h, p.Cout, d.out[mt][nt]);
            if ((mt & 1) && pool_out) {
                const int64_t po = (lane & 1) ? (int64_t)-1 : pooloff(mt);
                tile_store_packed<T>(pool_out, po, cbase, h, p.Cout, d.pool[mt >> 1][nt]);
            }
        }
    }
}
