// Row-streaming weight gradient of the dense 3x3 / stride-1 convolutions (16-bit operands: bf16 or f16), gfx950.
//
//   dW[co][ky][kx][ci] = sum over pixels of gout[y][x][co] * in[y + ky - 1][x + kx - 1][ci]
//
// (autograd of the conv2d call sites models/FAL_netB.py:38,45,55,73,75,127 of the reference).  GEMM view: M = cout,
// N = 9 * cin, K = pixels.  A workgroup owns a 64 (cout) x 64 (cin) channel block for ALL nine taps -- its four waves one
// 32 x 32 sub-block each, nine f32 accumulator tiles per wave -- and streams a 32-pixel-wide column strip of the image
// top to bottom, ONE image row per step:
//   * the step's gout row (32 px x 64 ch) and input row (34 px x 64 ch, the +-1 column halo) arrive by LDS-DMA
//     (global_load_lds_dwordx4: whole 128-B pixel lines, no VGPR staging) into a ring of D+1 row slots, D steps ahead;
//     out-of-image pixels are fetched from a 128-B page of zeros, so every wave issues the same piece count per step
//     (counted vmcnt + raw s_barrier, one barrier per step);
//   * input row i meets gout rows i+1, i, i-1 (ky = 0, 1, 2): the three gout fragments live in a rolling REGISTER
//     window, so a step reads one new gout fragment and three shifted input fragments (kx = 0, 1, 2) per 16-pixel half
//     and issues nine MFMAs on them: 0.44 fragments (0.9 transposed 8-B LDS reads) per v_mfma_f32_32x32x16, against
//     1.7-2.7 reads per MFMA of the patch kernel (wgrad3x3_patch_kernel) whose waves re-read both operands per tap row;
//   * rows are never re-fetched for a halo inside a strip (the patch form re-staged 6 rows for 4): 282 FLOP per
//     fetched byte.
// LDS image: [pixel][64 channels], 128-B rows as the DMA writes them; the two 64-B halves of pixels with bit 1 set are
// exchanged (on the SOURCE address: the DMA destination is lane-linear), which makes the ds_read_b64_tr_b16 fragment
// reads of any four consecutive pixels conflict-free.
// Split-K: the (sample, strip, row) units are cut into nsplit contiguous ranges; every workgroup writes one f32 slab
// [9][64][64] into partial[split][tap][co][ci] (falnet_wgrad_reduce_batched sums them).  blockIdx -> (split, tile) keeps
// the channel tiles of one pixel range on one XCD (they stream the same rows: L2 hits).
#include <stdlib.h>
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

template <typename T> struct Mma16;
template <> struct Mma16<bf16_t> {
    static __device__ __forceinline__ f32x16 mma(s16x8 a, s16x8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ float f32(short v) { return __uint_as_float(((unsigned)(unsigned short)v) << 16); }
};
template <> struct Mma16<f16_t> {
    static __device__ __forceinline__ f32x16 mma(s16x8 a, s16x8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ float f32(short v) { return (float)__builtin_bit_cast(_Float16, v); }
};

#define WR_THREADS 256
#define WR_TW 32                 // output pixels per strip row
#define WR_GROW (WR_TW * 128)    // gout row: 32 px x 64 ch x 2 B
#define WR_XPX 40                // input row: 34 px used, rounded up to whole 8-pixel DMA pieces
#define WR_XROW (WR_XPX * 128)
#define WR_SLOT (WR_GROW + WR_XROW)

__device__ uint4 g_wr_zero[8] = {};  // 128 B of zeros: source of every out-of-image / out-of-range 16-B piece

typedef __attribute__((address_space(1))) const void* wr_gptr_t;
typedef __attribute__((address_space(3))) void* wr_lptr_t;
typedef s16x4 __attribute__((address_space(3))) * wr_lds_v4;

// One 1-KiB LDS-DMA piece: lane l's 16 B from its own global address to LDS byte lds_dst + 16 l.  Inline asm on purpose:
// hipcc orders every ds_read behind ALL outstanding builtin global_load_lds with s_waitcnt vmcnt(0) (its LDS-DMA alias
// tracking), which drains the row ring at every step; an asm statement is outside that bookkeeping, the counted
// wr_vmcnt<> waits below are the only ordering (cdna_hip_programming.md section 5.7).  M0 is saved and restored inside
// the statement (compiler-reserved register).
__device__ __forceinline__ void wr_glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}

template <int K>
__device__ __forceinline__ void wr_vmcnt() {
    if constexpr (K == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (K == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if constexpr (K == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if constexpr (K == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (K == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else if constexpr (K == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (K == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    else if constexpr (K == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (K == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
    else static_assert(K == 0, "unsupported vmcnt");
}

struct WrCursor { int u, n, s, b, x0, y0; };  // item = rows [y0, y0+n) of column strip x0 of sample b; s = step inside it (0..n+1)

template <typename T, int D, int ABL = 0>  // ABL: timing-only ablations (1 no DMA in the loop, 2 no MFMAs, 3 no fragment reads / MFMAs)
__global__ __launch_bounds__(WR_THREADS, 2) void wgrad3x3_rows_kernel(const falnet_wgrad_t p, int w_rows, int ntci, int ntiles, int nstrips) {
    constexpr int NS = D + 1;
    static_assert(D >= 1 && D <= 3, "prefetch distance");
    __shared__ __attribute__((aligned(1024))) char lds[NS * WR_SLOT];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds_base = (unsigned)(unsigned long)(wr_lptr_t)lds;  // LDS byte address of the ring
    const int nsplit = p.nsplit;
    int tile, split;
    {
        const int wg = blockIdx.x;
        if ((nsplit & 7) == 0) {  // the channel tiles of one pixel range share an XCD (blocks b and b + 8 do)
            const int idx = wg >> 3;
            split = (idx / ntiles) * 8 + (wg & 7);
            tile = idx % ntiles;
        } else {
            tile = wg % ntiles;
            split = wg / ntiles;
        }
    }
    const int ci0 = (tile % ntci) * 64, co0 = (tile / ntci) * 64;
    const int a_t = wave >> 1, c_t = wave & 1;  // this wave's 32-channel sub-tiles (cin, cout)
    const int H = p.TH, TW = p.TW, gC = p.gC;
    const int R = p.B * nstrips * H;
    const int u0 = (int)((int64_t)R * split / nsplit), u1 = (int)((int64_t)R * (split + 1) / nsplit);

    // ---- per-lane DMA geometry: lane = (pixel pl of an 8-pixel piece, 16-B chunk position cpos of its 128-B line) ----
    const int pl = lane >> 3, cpos = lane & 7;
    const int gch = cpos ^ (((pl >> 1) & 1) << 2);  // channel chunk this lane FETCHES (64-B halves exchanged on pixels with bit 1 set)
    const char* const zero_page = reinterpret_cast<const char*>(g_wr_zero);
    const bool g_chok = co0 + 8 * gch < gC;
    const int ch = ci0 + 8 * gch;
    const bool x_chok = ch < p.cin_total;
    const int c_first = p.src[0].C;
    const bool second = p.nsrc > 1 && ch >= c_first;
    const falnet_src_t& S = second ? p.src[1] : p.src[0];
    const int l_hs = S.H != p.IH ? 1 : 0, l_ws = S.W != p.IW ? 1 : 0;  // exact 2x nearest upsampling (dispatcher checks)
    const int64_t l_sy = S.sy, l_sx = S.sx, l_sb = S.sb;
    const T* const l_ptr = reinterpret_cast<const T*>(S.ptr) + (second ? ch - c_first : ch);

    // lane bases of the current ISSUE item (recomputed when the issue cursor enters a new item)
    const T* gbase = nullptr;
    const T* xbase0 = nullptr;
    const T* xbase4 = nullptr;
    bool g_ok = false, x_ok0 = false, x_ok4 = false;
    auto item_bases = [&](const WrCursor& c) {
        const int gx = c.x0 + 8 * wave + pl;
        g_ok = g_chok && gx < TW;
        gbase = reinterpret_cast<const T*>(p.gout) + (((int64_t)c.b * H) * TW + gx) * gC + co0 + 8 * gch;
        const int x0i = c.x0 - 1 + 8 * wave + pl;
        x_ok0 = x_chok && x0i >= 0 && x0i < p.IW;
        xbase0 = l_ptr + (int64_t)c.b * l_sb + (int64_t)(x0i >> l_ws) * l_sx;
        const int x4 = c.x0 - 1 + 32 + pl;
        x_ok4 = x_chok && pl < 2 && x4 < p.IW;
        xbase4 = l_ptr + (int64_t)c.b * l_sb + (int64_t)(x4 >> l_ws) * l_sx;
    };
    auto load_item = [&](WrCursor& c, int u) {
        c.u = u;
        const int bs = u / H;
        c.y0 = u - bs * H;
        c.n = min((bs + 1) * H, u1) - u;
        c.b = bs / nstrips;
        c.x0 = (bs - c.b * nstrips) * WR_TW;
        c.s = 0;
    };
    auto advance = [&](WrCursor& c) -> bool {  // next step; false once the range is exhausted.  true = entered a new item
        if (++c.s < c.n + 2) return false;
        load_item(c, c.u + c.n);
        return true;
    };
    const int nst = u1 > u0 ? (u1 - u0) + 2 * ((u1 - 1) / H - u0 / H + 1) : 0;  // n + 2 steps per item

    auto issue = [&](const WrCursor& c, int slot) {
        const unsigned Gs = lds_base + slot * WR_SLOT, Xs = Gs + WR_GROW;
        const int yg = c.y0 + c.s;
        const bool gv = c.s < c.n;
        const char* ga = (gv && g_ok) ? reinterpret_cast<const char*>(gbase + (int64_t)yg * TW * gC) : zero_page;
        wr_glds16(ga, Gs + wave * 1024);
        const int i = c.y0 - 1 + c.s;
        const bool xv = i >= 0 && i < p.IH;
        const int64_t ro = (int64_t)(i >> l_hs) * l_sy;
        const char* xa = (xv && x_ok0) ? reinterpret_cast<const char*>(xbase0 + ro) : zero_page;
        wr_glds16(xa, Xs + wave * 1024);
        if (wave == 0) {
            const char* xb = (xv && x_ok4) ? reinterpret_cast<const char*>(xbase4 + ro) : zero_page;
            wr_glds16(xb, Xs + 4 * 1024);
        }
    };

    // ---- fragment read geometry (as the patch kernels: channel on lane & 31, k = pixel on lane >> 5 and element) ----
    const int i16 = lane & 15, g16 = lane >> 4;
    const int kh = g16 >> 1, cb = g16 & 1, q = i16 >> 2, pc = i16 & 3;
    auto frag_off = [&](int tile32, int pshift) {  // byte offset of this lane's first transposed read: pixel pshift + q + 8 kh
        const int gc = tile32 * 4 + cb * 2 + (pc >> 1);
        const int sw = ((pshift + q) >> 1) & 1;
        return (q + 8 * kh) * 128 + ((gc ^ (sw << 2)) * 16) + (pc & 1) * 8;
    };
    const int offA = frag_off(c_t, 0);
    const int offB0 = frag_off(a_t, 0), offB1 = frag_off(a_t, 1) + 128, offB2 = frag_off(a_t, 2) + 256;
    auto frag = [&](const char* base) -> s16x8 {
        const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((wr_lds_v4)(base));
        const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((wr_lds_v4)(base + 4 * 128));
        return __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
    };

    f32x16 acc[3][3];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[dy][dx][j] = 0.f;
    s16x8 aw1[2], aw2[2];  // gout fragments of the two previous rows (per 16-pixel half)
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int j = 0; j < 8; ++j) aw1[hf][j] = aw2[hf][j] = 0;
    const bool a_live = ci0 + 32 * a_t < p.cin_total && co0 + 32 * c_t < gC;  // a 64-channel block may be half empty
    const bool do_bias = p.bias_grad != nullptr && (tile % ntci) == 0 && a_t == 0;
    float bsum = 0.f;

    WrCursor ci_, cc_;  // issue / compute cursors
    if (nst > 0) {
        load_item(ci_, u0);
        cc_ = ci_;
        item_bases(ci_);
#pragma unroll
        for (int d = 0; d < D; ++d) {
            if (d < nst) {
                issue(ci_, d);
                if (d + 1 < nst && advance(ci_)) item_bases(ci_);
            }
        }
    }
    int slot = 0, islot = D;  // slot of step g; slot the step g + D is issued into
    for (int g = 0; g < nst; ++g) {
        // this wave's pieces of step g have landed once at most k steps' worth of younger pieces are outstanding
        const int k = min(D - 1, nst - 1 - g);
        if (ABL == 1) {
        } else if (wave == 0) {
            if (k >= 2) wr_vmcnt<(D >= 3 ? 6 : 0)>();
            else if (k == 1) wr_vmcnt<(D >= 2 ? 3 : 0)>();
            else wr_vmcnt<0>();
        } else {
            if (k >= 2) wr_vmcnt<(D >= 3 ? 4 : 0)>();
            else if (k == 1) wr_vmcnt<(D >= 2 ? 2 : 0)>();
            else wr_vmcnt<0>();
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // own LDS reads of step g - 1 are done (slot reuse below)
        __builtin_amdgcn_s_barrier();
        if (ABL != 1 && g + D < nst) {
            issue(ci_, islot);
            if (g + D + 1 < nst && advance(ci_)) item_bases(ci_);
        }
        const char* Gs = lds + slot * WR_SLOT;
        const char* Xs = Gs + WR_GROW;
        const int s = cc_.s, n = cc_.n;
        const int i = cc_.y0 - 1 + s;
        const bool xv = i >= 0 && i < p.IH;
        const bool m0 = s < n, m1 = s >= 1 && s - 1 < n, m2 = s >= 2;  // gout rows y0+s, y0+s-1, y0+s-2 inside the item
        s16x8 a0[2], bf[2][3];
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            if (ABL == 3) {
                a0[hf] = aw1[hf];
                bf[hf][0] = bf[hf][1] = bf[hf][2] = aw2[hf];
                continue;
            }
            a0[hf] = frag(Gs + offA + hf * 16 * 128);
            bf[hf][0] = frag(Xs + offB0 + hf * 16 * 128);
            bf[hf][1] = frag(Xs + offB1 + hf * 16 * 128);
            bf[hf][2] = frag(Xs + offB2 + hf * 16 * 128);
        }
        if (ABL < 2 && a_live && xv) {  // wave-uniform branches around groups of three MFMAs (a fast path duplicating the eighteen made hipcc spill)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                if (m0) {
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) acc[0][dx] = Mma16<T>::mma(a0[hf], bf[hf][dx], acc[0][dx]);
                }
                if (m1) {
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) acc[1][dx] = Mma16<T>::mma(aw1[hf], bf[hf][dx], acc[1][dx]);
                }
                if (m2) {
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) acc[2][dx] = Mma16<T>::mma(aw2[hf], bf[hf][dx], acc[2][dx]);
                }
            }
        }
        if (do_bias && m0) {
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                for (int j = 0; j < 8; ++j) bsum += Mma16<T>::f32(a0[hf][j]);
        }
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            aw2[hf] = aw1[hf];
            aw1[hf] = a0[hf];
        }
        if (g + 1 < nst && advance(cc_)) {  // new item: the window restarts (m1 / m2 are false on its first steps anyway)
        }
        slot = slot + 1 == NS ? 0 : slot + 1;
        islot = islot + 1 == NS ? 0 : islot + 1;
    }

    // ---- slab: partial[split][tap][co][ci] (every workgroup writes its whole block, zeros included) ----
    const int r = lane & 31, h = lane >> 5;
    const int ci = ci0 + 32 * a_t + r;
    if (ci < p.cin_total) {
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                float* dst = p.partial + (((int64_t)split * 9 + dy * 3 + dx) * w_rows) * p.cin_total;
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const int co = co0 + 32 * c_t + (j & 3) + 8 * (j >> 2) + 4 * h;
                    if (co < w_rows) dst[(int64_t)co * p.cin_total + ci] = acc[dy][dx][j];
                }
            }
    }
    if (do_bias) {  // lane (r, h) summed the pixels 8h..8h+7 (+16) of every row for channel co0 + 32 c_t + r
        bsum += __shfl_xor(bsum, 32, 64);
        const int co = co0 + 32 * c_t + r;
        if (h == 0 && co < p.cout) atomicAdd(p.bias_grad + co, bsum);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Eight-wave form (the one the dispatcher launches): ONE workgroup per CU.  The four (cin, cout) sub-tile roles exist twice,
// once per 16-pixel half of the strip, so both wave groups consume the SAME rows in LDS (half the L2 -> LDS traffic per MFMA
// of two independent four-wave workgroups) and their accumulators are summed through LDS before one slab leaves the CU:
// 256 slabs per layer instead of 512.  A step moves TWO image rows (18 MFMAs per wave and barrier); every wave keeps running
// per-lane source pointers (one 64-bit add per piece and step; lanes outside the image or the channel range sit on the zero
// page with increment 0), so the steady-state loop has no address arithmetic beyond that.
#define WR8_THREADS 512
#define WR8_SLOT (2 * WR_GROW + 2 * WR_XROW)   // 18 KiB: gout rows r0, r1, input rows j0, j1
#define WR8_RED (4 * 9 * 4 * 1024)             // epilogue: 4 waves x 9 accumulator tiles x 4 KiB

struct Wr8Item { int u, n, T, t, b, x0, y0; };  // rows [y0, y0+n) of strip x0 of sample b; t = step inside it (two rows per step)

// SPB = steps per barrier.  A step (two image rows) is only 18 MFMAs per wave -- the LDS-DMA conv kernels run 72 between barriers --, and
// every barrier costs the arrival skew of eight waves plus the refill of the fragment pipeline.  With SPB = 2 the ring has D + 2 slots
// (D = 4 steps of prefetch: 6 x 18 KB, inside the 147 KB the epilogue reduction needs anyway), a barrier stands only in front of the even
// steps, and the counted vmcnt in front of it retires the pieces of BOTH steps of the pair: step g issues into slot (g + D) mod NS, last
// read during step g - SPB, i.e. before the most recent barrier.
template <typename T, int D, int ABL = 0, bool STAGGER = true, bool ILV = true, int SPB = 1>
__global__ __launch_bounds__(WR8_THREADS, 2) void wgrad3x3_rows8_kernel(const falnet_wgrad_t p, int w_rows, int ntci, int ntiles, int nstrips) {
    constexpr int NS = D + SPB;
    static_assert(D >= 1 && D <= 4 && (SPB == 1 || SPB == 2 || SPB == 4) && D >= SPB, "prefetch distance / steps per barrier");
    constexpr int LDS_BYTES = NS * WR8_SLOT > WR8_RED ? NS * WR8_SLOT : WR8_RED;
    __shared__ __attribute__((aligned(1024))) char lds[LDS_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds_base = (unsigned)(unsigned long)(wr_lptr_t)lds;
    const int nsplit = p.nsplit;
    int tile, split;
    {
        const int wg = blockIdx.x;
        if ((nsplit & 7) == 0) {  // the channel tiles of one pixel range share an XCD (blocks b and b + 8 do)
            const int idx = wg >> 3;
            split = (idx / ntiles) * 8 + (wg & 7);
            tile = idx % ntiles;
        } else {
            tile = wg % ntiles;
            split = wg / ntiles;
        }
    }
    const int ci0 = (tile % ntci) * 64, co0 = (tile / ntci) * 64;
    // 16-pixel half, 32-channel sub-tiles (cin, cout).  (Measured and rejected: the cin sub-tile as the high bit, so that the two waves of
    // a SIMD differ in cin and a half-empty block -- 96 input channels -- keeps every SIMD busy: 8-10 % slower on every layer.)
    const int hf = wave >> 2, a_t = (wave >> 1) & 1, c_t = wave & 1;
    const int rsel = wave >> 2, pw = wave & 3;                        // DMA role: row of the step's pair, 8-pixel piece
    const int H = p.TH, TW = p.TW, gC = p.gC, IH = p.IH;
    const int R = p.B * nstrips * H;
    const int u0 = (int)((int64_t)R * split / nsplit), u1 = (int)((int64_t)R * (split + 1) / nsplit);

    // ---- per-lane DMA geometry (as the four-wave kernel) ----
    const int pl = lane >> 3, cpos = lane & 7;
    const int gch = cpos ^ (((pl >> 1) & 1) << 2);
    const char* const zero_page = reinterpret_cast<const char*>(g_wr_zero);
    const bool g_chok = co0 + 8 * gch < gC;
    const int ch = ci0 + 8 * gch;
    const bool x_chok = ch < p.cin_total;
    const int c_first = p.src[0].C;
    const bool second = p.nsrc > 1 && ch >= c_first;
    const falnet_src_t& S = second ? p.src[1] : p.src[0];
    const int l_hs = S.H != IH ? 1 : 0, l_ws = S.W != p.IW ? 1 : 0;
    const int64_t l_sy = S.sy, l_sx = S.sx, l_sb = S.sb;
    const T* const l_ptr = reinterpret_cast<const T*>(S.ptr) + (second ? ch - c_first : ch);

    auto load_item = [&](Wr8Item& c, int u) {
        c.u = u;
        const int bs = u / H;
        c.y0 = u - bs * H;
        c.n = min((bs + 1) * H, u1) - u;
        c.T = (c.n + 3) >> 1;
        c.b = bs / nstrips;
        c.x0 = (bs - c.b * nstrips) * WR_TW;
        c.t = 0;
    };
    int nst = 0;
    for (int u = u0; u < u1;) {
        const int e = min((u / H + 1) * H, u1);
        nst += (e - u + 3) >> 1;
        u = e;
    }

    // running per-lane source pointers of THIS wave's pieces (rows y0 + rsel + 2t / y0 - 1 + rsel + 2t) and their per-step increments
    const char* gptr = zero_page;
    const char* xptr = zero_page;
    const char* x4ptr = zero_page;
    unsigned g_inc = 0, x_inc = 0, x4_inc = 0;
    auto item_pointers = [&](const Wr8Item& c) {
        const int gx = c.x0 + 8 * pw + pl;
        const bool g_ok = g_chok && gx < TW;
        gptr = g_ok ? reinterpret_cast<const char*>(reinterpret_cast<const T*>(p.gout) + (((int64_t)c.b * H + c.y0 + rsel) * TW + gx) * gC + co0 + 8 * gch) : zero_page;
        g_inc = g_ok ? (unsigned)(2 * TW * gC * (int)sizeof(T)) : 0u;
        const int i = c.y0 - 1 + rsel;
        const int64_t rowoff = (int64_t)c.b * l_sb + (int64_t)(i >> l_hs) * l_sy;
        const unsigned xi = (unsigned)((l_hs ? l_sy : 2 * l_sy) * (int)sizeof(T));
        const int xa = c.x0 - 1 + 8 * pw + pl;
        const bool x_ok = x_chok && xa >= 0 && xa < p.IW;
        xptr = x_ok ? reinterpret_cast<const char*>(l_ptr + rowoff + (int64_t)(xa >> l_ws) * l_sx) : zero_page;
        x_inc = x_ok ? xi : 0u;
        const int xb = c.x0 - 1 + 32 + pl;
        const bool x4_ok = x_chok && pl < 2 && xb < p.IW;
        x4ptr = x4_ok ? reinterpret_cast<const char*>(l_ptr + rowoff + (int64_t)(xb >> l_ws) * l_sx) : zero_page;
        x4_inc = x4_ok ? xi : 0u;
    };
    auto issue_piece = [&](const Wr8Item& c, int slot, int k) {  // k = 0 gout row, 1 input row, 2 the input row's two halo pixels
        const unsigned base = lds_base + slot * WR8_SLOT;
        if (k == 0) {
            const bool gv = 2 * c.t + rsel < c.n;                    // gout row inside the item
            wr_glds16(gv ? gptr : zero_page, base + rsel * WR_GROW + pw * 1024);
            gptr += g_inc;
            return;
        }
        const int i = c.y0 - 1 + 2 * c.t + rsel;
        const bool xv = i >= 0 && i < IH;                             // input row inside the image
        if (k == 1) {
            wr_glds16(xv ? xptr : zero_page, base + 2 * WR_GROW + rsel * WR_XROW + pw * 1024);
            xptr += x_inc;
        } else if (pw == 0) {
            wr_glds16(xv ? x4ptr : zero_page, base + 2 * WR_GROW + rsel * WR_XROW + 4 * 1024);
            x4ptr += x4_inc;
        }
    };
    auto issue = [&](const Wr8Item& c, int slot) {
        issue_piece(c, slot, 0);
        issue_piece(c, slot, 1);
        issue_piece(c, slot, 2);
    };

    // ---- fragment read geometry ----
    const int i16 = lane & 15, g16 = lane >> 4;
    const int kh = g16 >> 1, cb = g16 & 1, q = i16 >> 2, pc = i16 & 3;
    auto frag_off = [&](int tile32, int pshift) {
        const int gc = tile32 * 4 + cb * 2 + (pc >> 1);
        const int sw = ((pshift + q) >> 1) & 1;
        return (q + 8 * kh) * 128 + ((gc ^ (sw << 2)) * 16) + (pc & 1) * 8;
    };
    const int offA = frag_off(c_t, 0) + hf * 16 * 128;
    const int offB0 = 2 * WR_GROW + frag_off(a_t, 0) + hf * 16 * 128, offB1 = 2 * WR_GROW + frag_off(a_t, 1) + 128 + hf * 16 * 128,
              offB2 = 2 * WR_GROW + frag_off(a_t, 2) + 256 + hf * 16 * 128;
    auto frag = [&](const char* base) -> s16x8 {
        const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((wr_lds_v4)(base));
        const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((wr_lds_v4)(base + 4 * 128));
        return __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
    };

    f32x16 acc[3][3];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[dy][dx][j] = 0.f;
    s16x8 w_a, w_b;  // gout fragments of the two previous rows (relative rows 2t-2, 2t-1)
#pragma unroll
    for (int j = 0; j < 8; ++j) w_a[j] = w_b[j] = 0;
    const bool a_live = ci0 + 32 * a_t < p.cin_total && co0 + 32 * c_t < gC;
    const bool do_bias = p.bias_grad != nullptr && (tile % ntci) == 0 && a_t == 0;
    float bsum = 0.f;

    Wr8Item ci_, cc_;
    if (nst > 0) {
        load_item(ci_, u0);
        cc_ = ci_;
        item_pointers(ci_);
#pragma unroll
        for (int d = 0; d < D; ++d) {
            if (d < nst) {
                issue(ci_, d);
                if (d + 1 < nst && ++ci_.t == ci_.T) {
                    load_item(ci_, ci_.u + ci_.n);
                    item_pointers(ci_);
                }
            }
        }
    }
    int slot = 0, islot = D;
    unsigned long long st_sum[5] = {0, 0, 0, 0, 0}, st_prev = 0;
    auto stamp = [&](int seg) {  // ABL == 9 (diagnostic build): cycles of the loop segment that ends here
        if constexpr (ABL == 9) {
            unsigned long long t;
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
            __builtin_amdgcn_sched_barrier(0);
            if (seg >= 0) st_sum[seg] += t - st_prev;
            st_prev = t;
        }
    };
    stamp(-1);
    const unsigned long long rt0 = ABL == 9 ? __builtin_amdgcn_s_memrealtime() : 0ull, ct0 = st_prev;
    // Stagger (MI355X_MICROARCH.md, "Two waves per SIMD" item 9): waves w and w + 4 share a SIMD and run the same program
    // between the same barriers, so un-staggered their MFMA phases collide and their DMA-issue / fragment-read phases leave
    // the matrix pipe idle together (stamps: 1150 of 2450 cycles per step in MFMAs).  Waves 4-7 (`lag`) therefore run the
    // MFMAs of step g - 1 right AFTER barrier g and only then issue their DMA pieces and read the fragments of step g, while
    // waves 0-3 issue / read first and multiply last: matrix work of one group beside the memory work of the other.  The
    // fragments and their validity flags simply stay in registers across the barrier; nothing else changes.
    const bool lag = STAGGER && hf == 1;
    s16x8 n_a, n_b, xa[3], xb[3];
    int m_t2 = 0, m_n = 0;          // relative row 2t and row count of the step whose fragments are in registers
    bool m_vx0 = false, m_vx1 = false, m_have = false;
#pragma unroll
    for (int j = 0; j < 8; ++j) n_a[j] = n_b[j] = 0;
    auto advance_issue = [&](int g) {  // the issue cursor moves to step g + D + 1
        if (g + D + 1 < nst && ++ci_.t == ci_.T) {
            load_item(ci_, ci_.u + ci_.n);
            item_pointers(ci_);
        }
        islot = islot + 1 == NS ? 0 : islot + 1;
    };
    auto read_frags = [&](int g) {  // this wave's fragments of step g (and the flags its MFMAs need)
        const char* sb = lds + slot * WR8_SLOT;
        m_t2 = 2 * cc_.t;
        m_n = cc_.n;
        const int i0 = cc_.y0 - 1 + m_t2;
        m_vx0 = i0 >= 0 && i0 < IH;
        m_vx1 = i0 + 1 >= 0 && i0 + 1 < IH;
        m_have = true;
        if (ABL == 3 || ABL == 4) {
            n_a = w_a; n_b = w_b;
            xa[0] = xa[1] = xa[2] = xb[0] = xb[1] = xb[2] = w_a;
        } else {
            n_a = frag(sb + offA);
            n_b = frag(sb + WR_GROW + offA);
            xa[0] = frag(sb + offB0);
            xa[1] = frag(sb + offB1);
            xa[2] = frag(sb + offB2);
            xb[0] = frag(sb + WR_XROW + offB0);
            xb[1] = frag(sb + WR_XROW + offB1);
            xb[2] = frag(sb + WR_XROW + offB2);
        }
        if (g + 1 < nst && ++cc_.t == cc_.T) load_item(cc_, cc_.u + cc_.n);
        slot = slot + 1 == NS ? 0 : slot + 1;
        stamp(3);
    };
    for (int g = 0; g <= nst; ++g) {
        const bool do_issue = ABL != 1 && ABL != 4 && ABL != 5 && g + D < nst;
        if (g < nst) {
          if (SPB == 1 || (g & (SPB - 1)) == 0) {
            // this wave's pieces of steps g .. g + SPB - 1 have landed once at most k steps' worth of younger pieces are outstanding
            // (issued so far: steps <= g + D - 1)
            const int k = max(0, min(D - SPB, nst - g - SPB));
            if (ABL == 1 || ABL == 4 || ABL == 5) {
            } else if (pw == 0) {
                if (k >= 2) wr_vmcnt<(D - SPB >= 2 ? 6 : 0)>();
                else if (k == 1) wr_vmcnt<(D - SPB >= 1 ? 3 : 0)>();
                else wr_vmcnt<0>();
            } else {
                if (k >= 2) wr_vmcnt<(D - SPB >= 2 ? 4 : 0)>();
                else if (k == 1) wr_vmcnt<(D - SPB >= 1 ? 2 : 0)>();
                else wr_vmcnt<0>();
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // own LDS reads of the previous steps are done (slot reuse)
            stamp(0);
            __builtin_amdgcn_s_barrier();
            stamp(1);
          }
            if (!ILV && do_issue && !lag) {  // un-interleaved order: every piece of step g + D before the fragment reads
                issue(ci_, islot);
                advance_issue(g);
            }
            stamp(2);
            if (!lag) read_frags(g);
        }
        // DMA pieces of step g + D go BETWEEN the MFMA groups (ILV): a piece holds its wave ~150 cycles in issue, which the
        // matrix pipe spends on the MFMAs issued just before it instead of waiting behind three pieces in a row
        auto piece = [&](int kk) {
            if (ILV && do_issue) {
                __builtin_amdgcn_sched_barrier(0);
                issue_piece(ci_, islot, kk);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        {  // the step whose fragments are in registers: g for waves 0-3, g - 1 for the lagging waves 4-7
            const int t2 = m_t2;
            auto vr = [&](int r) { return m_have && (unsigned)r < (unsigned)m_n; };
            const bool live = (ABL < 2 || ABL == 9 || ABL == 4) && a_live;
            if (live && m_vx0 && vr(t2)) {  // input row j = 2t meets gout rows 2t (ky 0), 2t-1 (ky 1), 2t-2 (ky 2)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) acc[0][dx] = Mma16<T>::mma(n_a, xa[dx], acc[0][dx]);
            }
            piece(0);
            if (live && m_vx0 && vr(t2 - 1)) {
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) acc[1][dx] = Mma16<T>::mma(w_b, xa[dx], acc[1][dx]);
            }
            if (live && m_vx0 && vr(t2 - 2)) {
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) acc[2][dx] = Mma16<T>::mma(w_a, xa[dx], acc[2][dx]);
            }
            piece(1);
            if (live && m_vx1 && vr(t2 + 1)) {  // input row j = 2t+1 meets gout rows 2t+1, 2t, 2t-1
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) acc[0][dx] = Mma16<T>::mma(n_b, xb[dx], acc[0][dx]);
            }
            if (live && m_vx1 && vr(t2)) {
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) acc[1][dx] = Mma16<T>::mma(n_a, xb[dx], acc[1][dx]);
            }
            piece(2);
            if (live && m_vx1 && vr(t2 - 1)) {
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) acc[2][dx] = Mma16<T>::mma(w_b, xb[dx], acc[2][dx]);
            }
            if (ILV && do_issue) advance_issue(g);
            if (m_have) {
                if (do_bias) {  // rows outside the item arrive as zeros
#pragma unroll
                    for (int j = 0; j < 8; ++j) bsum += Mma16<T>::f32(n_a[j]) + Mma16<T>::f32(n_b[j]);
                }
                w_a = n_a;
                w_b = n_b;
                m_have = false;
            }
        }
        stamp(4);
        if (lag && g < nst) {
            if (!ILV && do_issue) {
                issue(ci_, islot);
                advance_issue(g);
            }
            read_frags(g);
        }
    }

    // ---- the two halves' accumulators are summed through LDS: waves 4-7 deposit, waves 0-3 add and write the slab ----
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // every wave is past its last ring read: the ring space is reused
    float* red = reinterpret_cast<float*>(lds) + (wave & 3) * (9 * 4 * 256) + lane * 4;
    if (hf == 1) {
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                for (int j4 = 0; j4 < 4; ++j4)
                    *reinterpret_cast<float4*>(red + ((dy * 3 + dx) * 4 + j4) * 256) =
                        make_float4(acc[dy][dx][4 * j4], acc[dy][dx][4 * j4 + 1], acc[dy][dx][4 * j4 + 2], acc[dy][dx][4 * j4 + 3]);
    }
    __syncthreads();
    if (do_bias) {  // lane (r, h) summed pixels 8h..8h+7 of its half of every row for channel co0 + 32 c_t + r
        bsum += __shfl_xor(bsum, 32, 64);
        const int co = co0 + 32 * c_t + (lane & 31);
        if ((lane >> 5) == 0 && co < p.cout) atomicAdd(p.bias_grad + co, bsum);
    }
    if (hf == 1 && ABL != 9) return;
    const int r = lane & 31, h = lane >> 5;
    const int ci = ci0 + 32 * a_t + r;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            float* dst = p.partial + (((int64_t)split * 9 + dy * 3 + dx) * w_rows) * p.cin_total;
#pragma unroll
            for (int j4 = 0; j4 < 4; ++j4) {
                const float4 o = *reinterpret_cast<const float4*>(red + ((dy * 3 + dx) * 4 + j4) * 256);
                const float v[4] = {acc[dy][dx][4 * j4] + o.x, acc[dy][dx][4 * j4 + 1] + o.y, acc[dy][dx][4 * j4 + 2] + o.z, acc[dy][dx][4 * j4 + 3] + o.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int j = 4 * j4 + e;
                    const int co = co0 + 32 * c_t + (j & 3) + 8 * (j >> 2) + 4 * h;
                    if (ci < p.cin_total && co < w_rows) dst[(int64_t)co * p.cin_total + ci] = v[e];
                }
            }
        }
    __syncthreads();
    if constexpr (ABL == 9) {  // per-wave segment sums + step count into the head of this workgroup's slab (timing-only build)
        if (lane == 0) {
            unsigned long long* o = reinterpret_cast<unsigned long long*>(p.partial + (int64_t)split * 9 * w_rows * p.cin_total) + (tile * 8 + wave) * 8;
            for (int q2 = 0; q2 < 5; ++q2) o[q2] = st_sum[q2];
            o[5] = (unsigned long long)nst;
            o[6] = __builtin_amdgcn_s_memrealtime() - rt0;  // 100 MHz ticks over the loop
            o[7] = st_prev - ct0;                            // shader cycles over the loop
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// wgrad3x3_rows16_kernel (round 5): the eight-wave row-streaming kernel on v_mfma_f32_16x16x32.
//
// profiles/r05_sq_counters.txt / r05_dma16.txt: the chip holds a higher clock on the 16x16x32 shape (the LDS-DMA convolutions gained 8-12 % from
// nothing else), and with K = 32 ONE wave multiplies a whole 32-pixel strip row, so the strip no longer has to be split into two 16-pixel halves
// whose accumulators are summed through LDS at the end:
//   * wave = (c_t: 32-channel half of the 64 gout channels, a_q: 16-channel quarter of the 64 input channels); its output is 32 (cout) x 16 (cin)
//     x 9 taps = 18 tiles of 16 x 16: 72 accumulator registers instead of 144, every slab element written by exactly one wave (no reduction);
//   * a step still moves two image rows through the same LDS ring by the same LDS-DMA pieces; per row a wave reads two gout fragments (its two
//     16-channel tiles x 32 pixels) and three input fragments (column offsets 0, 1, 2) -- ten transposed 8-B reads -- and issues 18 MFMAs
//     (3 tap rows x 3 tap columns x 2 gout tiles; the gout fragments of the two previous rows stay in a register window as before);
//   * operand = [16 channels x 32 pixels]: 16-lane group g reads pixels 8g .. 8g+7 (two ds_read_b64_tr_b16 of four pixel rows each), so a 32-lane
//     half reads two blocks 8 pixels apart in the same columns: besides the 64-B half exchange on pixel bit 1 the 32-B quarters are exchanged on
//     pixel bit 3 (source-side, as everything the DMA writes): conflict-free for every column offset and both reads (exhaustive check).
//
// UP2 (falnet_wgrad_t::up2, round 5): the weight gradient of a `deconv` layer (nearest 2x upsampling + 3x3 convolution, FAL_netB.py:52-58) on the
// LOW-resolution grid.  up(x)[2y + py + ky - 1] is x[y + ((py + ky - 1) >> 1)]: for output parity py the taps ky that read the same x row fall
// together, so  dW[ky][kx] = sum over the four parities (py, px) of S[py, px][ky', kx']  with
//   S[py, px][ky', kx'] = sum_{y, x} g[2y + py, 2x + px] x[y + ky' - 1, x + kx' - 1],  ky' in {py, py + 1}, kx' in {px, px + 1}
// (py = 0: ky 0 <- ky' 0, ky 1 and 2 <- ky' 1;  py = 1: ky 0 and 1 <- ky' 1, ky 2 <- ky' 2) -- 16 instead of 36 tap products per low-resolution
// position.  A split is (parity class, pixel range): its workgroups stream the class's quarter of g (pixel stride 2 gC, row stride 2 rows: the
// LDS-DMA names every source pixel anyway) against x at its own size, issue 4 of the 9 tap products, and write the slab of the FULL 3x3 gradient's
// share of that class -- every slab element once -- so the batched slab reduce sums classes and pixel ranges alike.
template <typename T, int D, int SPB, bool UP2>
__global__ __launch_bounds__(WR8_THREADS, 2) void wgrad3x3_rows16_kernel(const falnet_wgrad_t p, int w_rows, int ntci, int ntiles, int nstrips) {
    constexpr int NS = D + SPB;
    static_assert(D >= 1 && D <= 4 && (SPB == 1 || SPB == 2) && D >= SPB, "prefetch distance / steps per barrier");
    __shared__ __attribute__((aligned(1024))) char lds[NS * WR8_SLOT];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds_base = (unsigned)(unsigned long)(wr_lptr_t)lds;
    const int nsplit = p.nsplit;
    int tile, split;
    {
        const int wg = blockIdx.x;
        if ((nsplit & 7) == 0) {  // the channel tiles of one pixel range share an XCD (blocks b and b + 8 do)
            const int idx = wg >> 3;
            split = (idx / ntiles) * 8 + (wg & 7);
            tile = idx % ntiles;
        } else {
            tile = wg % ntiles;
            split = wg / ntiles;
        }
    }
    const int ci0 = (tile % ntci) * 64, co0 = (tile / ntci) * 64;
    const int c_t = wave & 1, a_q = wave >> 1;   // MFMA role: 32 gout channels (two 16-channel tiles), 16 input channels
    const int rsel = wave >> 2, pw = wave & 3;   // DMA role: row of the step's pair, 8-pixel piece
    const int H = p.TH, TW = p.TW, gC = p.gC, IH = p.IH;
    const int R = p.B * nstrips * H;
    const int py = UP2 ? (split >> 1) & 1 : 0, px = UP2 ? split & 1 : 0;  // parity class of this split (UP2: splits = (pixel range, class))
    const int sub = UP2 ? split >> 2 : split, nsub = UP2 ? nsplit >> 2 : nsplit;
    const int u0 = (int)((int64_t)R * sub / nsub), u1 = (int)((int64_t)R * (sub + 1) / nsub);

    // ---- per-lane DMA geometry: lane = (pixel pl of an 8-pixel piece, 16-B position cpos of its 128-B line); the chunk it FETCHES is swizzled ----
    const int pl = lane >> 3, cpos = lane & 7;
    const int gch = cpos ^ ((((pl >> 1) & 1) << 2) | ((pw & 1) << 1));  // pixel 8 pw + pl of the LDS row: bit 1 -> 64-B halves, bit 3 -> 32-B quarters
    const char* const zero_page = reinterpret_cast<const char*>(g_wr_zero);
    const bool g_chok = co0 + 8 * gch < gC;
    const int ch = ci0 + 8 * gch;
    const bool x_chok = ch < p.cin_total;
    const int c_first = p.src[0].C;
    const bool second = p.nsrc > 1 && ch >= c_first;
    const falnet_src_t& S = second ? p.src[1] : p.src[0];
    const int l_hs = S.H != IH ? 1 : 0, l_ws = S.W != p.IW ? 1 : 0;
    const int64_t l_sy = S.sy, l_sx = S.sx, l_sb = S.sb;
    const T* const l_ptr = reinterpret_cast<const T*>(S.ptr) + (second ? ch - c_first : ch);

    auto load_item = [&](Wr8Item& c, int u) {
        c.u = u;
        const int bs = u / H;
        c.y0 = u - bs * H;
        c.n = min((bs + 1) * H, u1) - u;
        c.T = (c.n + 3) >> 1;
        c.b = bs / nstrips;
        c.x0 = (bs - c.b * nstrips) * WR_TW;
        c.t = 0;
    };
    int nst = 0;
    for (int u = u0; u < u1;) {
        const int e = min((u / H + 1) * H, u1);
        nst += (e - u + 3) >> 1;
        u = e;
    }
    const char* gptr = zero_page;
    const char* xptr = zero_page;
    const char* x4ptr = zero_page;
    unsigned g_inc = 0, x_inc = 0, x4_inc = 0;
    auto item_pointers = [&](const Wr8Item& c) {
        const int gx = c.x0 + 8 * pw + pl;
        const bool g_ok = g_chok && gx < TW;
        if constexpr (UP2) {  // row 2 (y0 + rsel) + py, pixel 2 gx + px of the upstream gradient at 2H x 2W
            gptr = g_ok ? reinterpret_cast<const char*>(reinterpret_cast<const T*>(p.gout) +
                                                        ((((int64_t)c.b * 2 * H + 2 * (c.y0 + rsel) + py) * (2 * TW)) + 2 * gx + px) * gC + co0 + 8 * gch) : zero_page;
            g_inc = g_ok ? (unsigned)(8 * TW * gC * (int)sizeof(T)) : 0u;
        } else {
            gptr = g_ok ? reinterpret_cast<const char*>(reinterpret_cast<const T*>(p.gout) + (((int64_t)c.b * H + c.y0 + rsel) * TW + gx) * gC + co0 + 8 * gch) : zero_page;
            g_inc = g_ok ? (unsigned)(2 * TW * gC * (int)sizeof(T)) : 0u;
        }
        const int i = c.y0 - 1 + rsel;
        const int64_t rowoff = (int64_t)c.b * l_sb + (int64_t)(i >> l_hs) * l_sy;
        const unsigned xi = (unsigned)((l_hs ? l_sy : 2 * l_sy) * (int)sizeof(T));
        const int xa = c.x0 - 1 + 8 * pw + pl;
        const bool x_ok = x_chok && xa >= 0 && xa < p.IW;
        xptr = x_ok ? reinterpret_cast<const char*>(l_ptr + rowoff + (int64_t)(xa >> l_ws) * l_sx) : zero_page;
        x_inc = x_ok ? xi : 0u;
        const int xb = c.x0 - 1 + 32 + pl;  // (issued by the pw == 0 waves only: pixels 32 .. 39 of the LDS row have bit 3 clear, like their gch)
        const bool x4_ok = x_chok && pl < 2 && xb < p.IW;
        x4ptr = x4_ok ? reinterpret_cast<const char*>(l_ptr + rowoff + (int64_t)(xb >> l_ws) * l_sx) : zero_page;
        x4_inc = x4_ok ? xi : 0u;
    };
    auto issue_piece = [&](const Wr8Item& c, int slot, int k) {  // k = 0 gout row, 1 input row, 2 the input row's two halo pixels
        const unsigned base = lds_base + slot * WR8_SLOT;
        if (k == 0) {
            const bool gv = 2 * c.t + rsel < c.n;
            wr_glds16(gv ? gptr : zero_page, base + rsel * WR_GROW + pw * 1024);
            gptr += g_inc;
            return;
        }
        const int i = c.y0 - 1 + 2 * c.t + rsel;
        const bool xv = i >= 0 && i < IH;
        if (k == 1) {
            wr_glds16(xv ? xptr : zero_page, base + 2 * WR_GROW + rsel * WR_XROW + pw * 1024);
            xptr += x_inc;
        } else if (pw == 0) {
            wr_glds16(xv ? x4ptr : zero_page, base + 2 * WR_GROW + rsel * WR_XROW + 4 * 1024);
            x4ptr += x4_inc;
        }
    };
    auto issue = [&](const Wr8Item& c, int slot) {
        issue_piece(c, slot, 0);
        issue_piece(c, slot, 1);
        issue_piece(c, slot, 2);
    };

    // ---- fragment read geometry: 16-lane group g16 reads pixels 8 g16 + q (+ 4 in the second read) + pshift of channel block cb16 ----
    const int i16 = lane & 15, g16 = lane >> 4;
    const int q = i16 >> 2, pc = i16 & 3;
    auto frag_off = [&](int cb16, int pshift, int second_read) {
        const int px = 8 * g16 + q + pshift + 4 * second_read;
        const int chunk = 2 * cb16 + (pc >> 1);
        const int sw = (((px >> 1) & 1) << 2) | (((px >> 3) & 1) << 1);
        return px * 128 + ((chunk ^ sw) * 16) + (pc & 1) * 8;
    };
    int offA[2][2], offB[3][2];  // [gout tile cc][read], [column offset][read]
#pragma unroll
    for (int rd = 0; rd < 2; ++rd) {
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) offA[cc][rd] = frag_off(2 * c_t + cc, 0, rd);
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) offB[dx][rd] = 2 * WR_GROW + frag_off(a_q, dx, rd);
    }
    auto frag = [&](const char* base, const int (&off)[2]) -> s16x8 {
        const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((wr_lds_v4)(base + off[0]));
        const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((wr_lds_v4)(base + off[1]));
        return __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
    };

    f32x4_t acc[3][3][2];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
#pragma unroll
            for (int cc = 0; cc < 2; ++cc) acc[dy][dx][cc] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    s16x8 w_a[2], w_b[2];  // gout fragments of the two previous rows (relative rows 2t-2, 2t-1), per 16-channel tile
#pragma unroll
    for (int cc = 0; cc < 2; ++cc)
#pragma unroll
        for (int j = 0; j < 8; ++j) w_a[cc][j] = w_b[cc][j] = 0;
    const bool a_live = ci0 + 16 * a_q < p.cin_total && co0 + 32 * c_t < gC;
    const bool do_bias = p.bias_grad != nullptr && (tile % ntci) == 0 && a_q == 0;
    float bsum[2] = {0.f, 0.f};

    Wr8Item ci_, cc_;
    if (nst > 0) {
        load_item(ci_, u0);
        cc_ = ci_;
        item_pointers(ci_);
#pragma unroll
        for (int d = 0; d < D; ++d) {
            if (d < nst) {
                issue(ci_, d);
                if (d + 1 < nst && ++ci_.t == ci_.T) {
                    load_item(ci_, ci_.u + ci_.n);
                    item_pointers(ci_);
                }
            }
        }
    }
    int slot = 0, islot = D;
    // stagger (as wgrad3x3_rows8_kernel): waves w and w + 4 share a SIMD; waves 4-7 run the MFMAs of step g - 1 right after barrier g and only then
    // issue their DMA pieces and read the fragments of step g
    const bool lag = (wave >> 2) == 1;
    s16x8 n_a[2], n_b[2], xa[3], xb[3];
    int m_t2 = 0, m_n = 0;
    bool m_vx0 = false, m_vx1 = false, m_have = false;
#pragma unroll
    for (int cc = 0; cc < 2; ++cc)
#pragma unroll
        for (int j = 0; j < 8; ++j) n_a[cc][j] = n_b[cc][j] = 0;
    auto advance_issue = [&](int g) {
        if (g + D + 1 < nst && ++ci_.t == ci_.T) {
            load_item(ci_, ci_.u + ci_.n);
            item_pointers(ci_);
        }
        islot = islot + 1 == NS ? 0 : islot + 1;
    };
    auto read_frags = [&](int g) {
        const char* sb = lds + slot * WR8_SLOT;
        m_t2 = 2 * cc_.t;
        m_n = cc_.n;
        const int i0 = cc_.y0 - 1 + m_t2;
        m_vx0 = i0 >= 0 && i0 < IH;
        m_vx1 = i0 + 1 >= 0 && i0 + 1 < IH;
        m_have = true;
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
            n_a[cc] = frag(sb, offA[cc]);
            n_b[cc] = frag(sb + WR_GROW, offA[cc]);
        }
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            xa[dx] = frag(sb, offB[dx]);
            xb[dx] = frag(sb + WR_XROW, offB[dx]);
        }
        if (g + 1 < nst && ++cc_.t == cc_.T) load_item(cc_, cc_.u + cc_.n);
        slot = slot + 1 == NS ? 0 : slot + 1;
    };
    for (int g = 0; g <= nst; ++g) {
        const bool do_issue = g + D < nst;
        if (g < nst) {
            if (SPB == 1 || (g & (SPB - 1)) == 0) {
                const int k = max(0, min(D - SPB, nst - g - SPB));
                if (pw == 0) {
                    if (k >= 2) wr_vmcnt<(D - SPB >= 2 ? 6 : 0)>();
                    else if (k == 1) wr_vmcnt<(D - SPB >= 1 ? 3 : 0)>();
                    else wr_vmcnt<0>();
                } else {
                    if (k >= 2) wr_vmcnt<(D - SPB >= 2 ? 4 : 0)>();
                    else if (k == 1) wr_vmcnt<(D - SPB >= 1 ? 2 : 0)>();
                    else wr_vmcnt<0>();
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
            if (!lag) read_frags(g);
        }
        auto piece = [&](int kk) {
            if (do_issue) {
                __builtin_amdgcn_sched_barrier(0);
                issue_piece(ci_, islot, kk);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        {
            const int t2 = m_t2;
            auto vr = [&](int r) { return m_have && (unsigned)r < (unsigned)m_n; };
            auto tap_on = [&](int par, int k) { return !UP2 || k == par || k == par + 1; };  // UP2: the class's two tap rows / columns
            auto mm = [&](int dy, const s16x8 (&A)[2], const s16x8 (&X)[3]) {
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    if (!tap_on(px, dx)) continue;  // (workgroup-uniform)
#pragma unroll
                    for (int cc = 0; cc < 2; ++cc) acc[dy][dx][cc] = H16<T>::mma16(A[cc], X[dx], acc[dy][dx][cc]);
                }
            };
            if (a_live && m_vx0 && vr(t2) && tap_on(py, 0)) mm(0, n_a, xa);      // input row j = 2t meets gout rows 2t (ky 0), 2t-1 (ky 1), 2t-2 (ky 2)
            piece(0);
            if (a_live && m_vx0 && vr(t2 - 1) && tap_on(py, 1)) mm(1, w_b, xa);
            if (a_live && m_vx0 && vr(t2 - 2) && tap_on(py, 2)) mm(2, w_a, xa);
            piece(1);
            if (a_live && m_vx1 && vr(t2 + 1) && tap_on(py, 0)) mm(0, n_b, xb);  // input row j = 2t+1 meets gout rows 2t+1, 2t, 2t-1
            if (a_live && m_vx1 && vr(t2) && tap_on(py, 1)) mm(1, n_a, xb);
            piece(2);
            if (a_live && m_vx1 && vr(t2 - 1) && tap_on(py, 2)) mm(2, w_b, xb);
            if (do_issue) advance_issue(g);
            if (m_have) {
                if (do_bias) {  // rows outside the item arrive as zeros
#pragma unroll
                    for (int cc = 0; cc < 2; ++cc)
#pragma unroll
                        for (int j = 0; j < 8; ++j) bsum[cc] += Mma16<T>::f32(n_a[cc][j]) + Mma16<T>::f32(n_b[cc][j]);
                }
#pragma unroll
                for (int cc = 0; cc < 2; ++cc) {
                    w_a[cc] = n_a[cc];
                    w_b[cc] = n_b[cc];
                }
                m_have = false;
            }
        }
        if (lag && g < nst) read_frags(g);
    }

    // ---- every (tap, cout, cin) element of the slab belongs to exactly one wave: tile (cc) of D[co 16 x ci 16], lane (ci = i16, rows 4 g16 + f) ----
    if (do_bias) {  // lane (i16, g16) summed pixels 8 g16 .. 8 g16 + 7 of every row for channel co0 + 32 c_t + 16 cc + i16
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
            float b = bsum[cc];
            b += __shfl_xor(b, 16, 64);
            b += __shfl_xor(b, 32, 64);
            const int co = co0 + 32 * c_t + 16 * cc + i16;
            if (g16 == 0 && co < p.cout) atomicAdd(p.bias_grad + co, b);
        }
    }
    const int ci = ci0 + 16 * a_q + i16;
    if (ci < p.cin_total) {
        // UP2: tap k of the 3x3 gradient takes the class's product k' (par = 0: 0 <- 0, 1 and 2 <- 1; par = 1: 0 and 1 <- 1, 2 <- 2)
        auto feeds = [&](int par, int kp, int k) { return !UP2 ? kp == k : (par == 0 ? (kp == 0 ? k == 0 : (kp == 1 && k >= 1)) : (kp == 1 ? k <= 1 : (kp == 2 && k == 2))); };
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        if (!UP2 && (ky != dy || kx != dx)) continue;
                        if (!(feeds(py, dy, ky) && feeds(px, dx, kx))) continue;  // (workgroup-uniform)
                        float* dst = p.partial + (((int64_t)split * 9 + ky * 3 + kx) * w_rows) * p.cin_total;
#pragma unroll
                        for (int cc = 0; cc < 2; ++cc)
#pragma unroll
                            for (int f = 0; f < 4; ++f) {
                                const int co = co0 + 32 * c_t + 16 * cc + 4 * g16 + f;
                                if (co < w_rows) dst[(int64_t)co * p.cin_total + ci] = acc[dy][dx][cc][f];
                            }
                    }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Stride-2 form of the eight-wave kernel (encoder convs conv2..conv4, FAL_netB.py:103-107: 3x3, stride 2, pad 1):
//   dW[co][ky][kx][ci] = sum over output pixels (i, j) of gout[i][j][co] * in[2 i + ky - 1][2 j + kx - 1][ci]
// Same roles (16-pixel half x 32-channel sub-tiles, one workgroup per CU, accumulators summed through LDS), same ring of row slots
// filled by LDS-DMA, but a step moves TWO gout rows (i0, i1 = i0 + 1) and the FOUR input rows R0..R3 = 2 i0 - 1 .. 2 i0 + 2:
//   gout i0 meets R0 (ky 0), R1 (ky 1), R2 (ky 2);  gout i1 meets R2 (ky 0), R3 (ky 1) and -- one step later -- the next R0 (ky 2),
// so one gout fragment stays in a register window (w_b) and an item of n rows takes n / 2 + 1 steps: 18 MFMAs per wave and step,
// as in the stride-1 kernel.  The DMA DE-INTERLEAVES every input row by column parity while fetching (each lane names its own source
// pixel): E[s] = in[2 (x0 + s)] (32 pixels), O[s] = in[2 (x0 + s) - 1] (33 pixels), so the sixteen pixels a fragment needs for tap
// kx are consecutive LDS rows -- kx 0: O[s], kx 1: E[s], kx 2: O[s + 1] -- and the transposed reads / XOR swizzle are unchanged.
// (The parity-plane halo kernel it replaces, wgrad3x3_s2_kernel, stages a 9 x 65-pixel region per 4 x 32 outputs through registers with
// two barriers per block: 96-171 TFLOP/s on these layers.)
#define WS2R_XROW (WR_GROW + WR_XROW)                 // one input row: E half 4 KiB + O half 5 KiB (33 pixels used)
#define WS2R_SLOT (2 * WR_GROW + 4 * WS2R_XROW)       // 44 KiB: gout rows i0, i1, input rows R0..R3

struct Ws2Item { int u, n, T, t, b, x0, y0; };

// (__launch_bounds__' second argument is HIP's MIN WAVES PER EXECUTION UNIT, not CUDA's blocks per SM: 2 = the eight waves of ONE workgroup on four
// SIMDs = a 256-register budget -- what a 512-thread workgroup gets anyway; the 132 KiB of LDS allow one workgroup per CU and the bound does not say otherwise)
template <typename T, int D>
__global__ __launch_bounds__(WR8_THREADS, 2) void wgrad3x3_rows8s2_kernel(const falnet_wgrad_t p, int w_rows, int ntci, int ntiles, int nstrips) {
    constexpr int NS = D + 1;
    static_assert(D >= 1 && D <= 2, "prefetch distance");
    constexpr int LDS_BYTES = NS * WS2R_SLOT > WR8_RED ? NS * WS2R_SLOT : WR8_RED;
    static_assert(LDS_BYTES <= 160 * 1024, "ring + reduction space must fit in LDS");
    __shared__ __attribute__((aligned(1024))) char lds[LDS_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds_base = (unsigned)(unsigned long)(wr_lptr_t)lds;
    const int nsplit = p.nsplit;
    int tile, split;
    {
        const int wg = blockIdx.x;
        if ((nsplit & 7) == 0) {
            const int idx = wg >> 3;
            split = (idx / ntiles) * 8 + (wg & 7);
            tile = idx % ntiles;
        } else {
            tile = wg % ntiles;
            split = wg / ntiles;
        }
    }
    const int ci0 = (tile % ntci) * 64, co0 = (tile / ntci) * 64;
    const int hf = wave >> 2, a_t = (wave >> 1) & 1, c_t = wave & 1;
    const int rsel = wave >> 2, pw = wave & 3;  // DMA role: gout row / input rows rsel and rsel + 2 of the step, 8-pixel piece
    const int H = p.TH, TW = p.TW, gC = p.gC, IH = p.IH, IW = p.IW;
    const int R = p.B * nstrips * H;
    const int u0 = (int)((int64_t)R * split / nsplit), u1 = (int)((int64_t)R * (split + 1) / nsplit);

    const int pl = lane >> 3, cpos = lane & 7;
    const int gch = cpos ^ (((pl >> 1) & 1) << 2);
    const char* const zero_page = reinterpret_cast<const char*>(g_wr_zero);
    const bool g_chok = co0 + 8 * gch < gC;
    const int ch = ci0 + 8 * gch;
    const bool x_chok = ch < p.cin_total;
    const falnet_src_t& S = p.src[0];
    const int64_t l_sy = S.sy, l_sx = S.sx, l_sb = S.sb;
    const T* const l_ptr = reinterpret_cast<const T*>(S.ptr) + ch;

    auto load_item = [&](Ws2Item& c, int u) {
        c.u = u;
        const int bs = u / H;
        c.y0 = u - bs * H;
        c.n = min((bs + 1) * H, u1) - u;
        c.T = (c.n >> 1) + 1;
        c.b = bs / nstrips;
        c.x0 = (bs - c.b * nstrips) * WR_TW;
        c.t = 0;
    };
    int nst = 0;
    for (int u = u0; u < u1;) {
        const int e = min((u / H + 1) * H, u1);
        nst += ((e - u) >> 1) + 1;
        u = e;
    }

    // running per-lane source pointers: gout row y0 + rsel + 2 t; input row 2 y0 - 1 + rsel + 4 t (the piece of row rsel + 2 sits two
    // image rows further: a constant byte offset); column pieces E (pw), O (pw) and the 33rd O pixel (pw == 0, lane pixel 0)
    const char* gptr = zero_page;
    const char* eptr = zero_page;
    const char* optr = zero_page;
    const char* hptr = zero_page;
    bool e_ok = false, o_ok = false, h_ok = false;
    unsigned g_inc = 0;
    const unsigned x_inc = (unsigned)(4 * l_sy * (int)sizeof(T));
    const int64_t x_row2 = 2 * l_sy * (int64_t)sizeof(T);
    auto item_pointers = [&](const Ws2Item& c) {
        const int gx = c.x0 + 8 * pw + pl;
        const bool g_ok = g_chok && gx < TW;
        gptr = g_ok ? reinterpret_cast<const char*>(reinterpret_cast<const T*>(p.gout) + (((int64_t)c.b * H + c.y0 + rsel) * TW + gx) * gC + co0 + 8 * gch) : zero_page;
        g_inc = g_ok ? (unsigned)(2 * TW * gC * (int)sizeof(T)) : 0u;
        const int64_t rowoff = (int64_t)c.b * l_sb + (int64_t)(2 * c.y0 - 1 + rsel) * l_sy;  // (may point in front of the image: only dereferenced for valid rows)
        const int ce = 2 * gx, co = 2 * gx - 1, chh = 2 * (c.x0 + 32 + pl) - 1;
        e_ok = x_chok && ce < IW;
        o_ok = x_chok && co >= 0 && co < IW;
        h_ok = x_chok && pl == 0 && chh < IW;
        eptr = reinterpret_cast<const char*>(l_ptr + rowoff + (int64_t)ce * l_sx);
        optr = reinterpret_cast<const char*>(l_ptr + rowoff + (int64_t)co * l_sx);
        hptr = reinterpret_cast<const char*>(l_ptr + rowoff + (int64_t)chh * l_sx);
    };
    // piece k of a step: 0 gout; 1 / 2: E of rows rsel / rsel + 2; 3 / 4: O of those rows; 5 / 6: their 33rd O pixel (pw == 0 only)
    auto issue_piece = [&](const Ws2Item& c, int slot, int k) {
        const unsigned base = lds_base + slot * WS2R_SLOT;
        if (k == 0) {
            const bool gv = 2 * c.t + rsel < c.n;
            wr_glds16(gv ? gptr : zero_page, base + rsel * WR_GROW + pw * 1024);
            gptr += g_inc;
            return;
        }
        const int far = (k - 1) & 1;                       // 0: row rsel, 1: row rsel + 2
        const int row = rsel + 2 * far;
        const int i = 2 * (c.y0 + 2 * c.t) - 1 + row;
        const bool xv = i >= 0 && i < IH;
        const unsigned xb = base + 2 * WR_GROW + row * WS2R_XROW;
        const int64_t off = far ? x_row2 : 0;
        if (k <= 2) wr_glds16((xv && e_ok) ? eptr + off : zero_page, xb + pw * 1024);
        else if (k <= 4) wr_glds16((xv && o_ok) ? optr + off : zero_page, xb + WR_GROW + pw * 1024);
        else if (pw == 0) wr_glds16((xv && h_ok) ? hptr + off : zero_page, xb + WR_GROW + 4 * 1024);
    };
    auto advance_pointers = [&]() {  // after the last piece of a step
        eptr += x_inc;
        optr += x_inc;
        hptr += x_inc;
    };
    auto issue = [&](const Ws2Item& c, int slot) {
#pragma unroll
        for (int k = 0; k < 7; ++k) issue_piece(c, slot, k);
        advance_pointers();
    };

    const int i16 = lane & 15, g16 = lane >> 4;
    const int kh = g16 >> 1, cb = g16 & 1, q = i16 >> 2, pc = i16 & 3;
    auto frag_off = [&](int tile32, int pshift) {
        const int gc = tile32 * 4 + cb * 2 + (pc >> 1);
        const int sw = ((pshift + q) >> 1) & 1;
        return (q + 8 * kh) * 128 + ((gc ^ (sw << 2)) * 16) + (pc & 1) * 8;
    };
    const int offA = frag_off(c_t, 0) + hf * 16 * 128;
    const int offE = 2 * WR_GROW + frag_off(a_t, 0) + hf * 16 * 128;
    const int offO0 = 2 * WR_GROW + WR_GROW + frag_off(a_t, 0) + hf * 16 * 128;
    const int offO1 = 2 * WR_GROW + WR_GROW + frag_off(a_t, 1) + 128 + hf * 16 * 128;
    auto frag = [&](const char* base) -> s16x8 {
        const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((wr_lds_v4)(base));
        const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((wr_lds_v4)(base + 4 * 128));
        return __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
    };

    f32x16 acc[3][3];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[dy][dx][j] = 0.f;
    s16x8 w_b;  // gout fragment of the previous step's second row
#pragma unroll
    for (int j = 0; j < 8; ++j) w_b[j] = 0;
    bool w_valid = false;
    const bool a_live = ci0 + 32 * a_t < p.cin_total && co0 + 32 * c_t < gC;
    const bool do_bias = p.bias_grad != nullptr && (tile % ntci) == 0 && a_t == 0;
    float bsum = 0.f;

    Ws2Item ci_, cc_;
    if (nst > 0) {
        load_item(ci_, u0);
        cc_ = ci_;
        item_pointers(ci_);
#pragma unroll
        for (int d = 0; d < D; ++d) {
            if (d < nst) {
                issue(ci_, d);
                if (d + 1 < nst && ++ci_.t == ci_.T) {
                    load_item(ci_, ci_.u + ci_.n);
                    item_pointers(ci_);
                }
            }
        }
    }
    int slot = 0, islot = D;
    // stagger as in the stride-1 kernel: waves 4-7 multiply the step they read before the last barrier while waves 0-3 issue / read first
    const bool lag = hf == 1;
    s16x8 n_a, n_b, xr[4][3];
    int m_t2 = 0, m_n = 0, m_i0 = 0;
    bool m_have = false, m_first = false;
#pragma unroll
    for (int j = 0; j < 8; ++j) n_a[j] = n_b[j] = 0;
    auto advance_issue = [&](int g) {
        advance_pointers();
        if (g + D + 1 < nst && ++ci_.t == ci_.T) {
            load_item(ci_, ci_.u + ci_.n);
            item_pointers(ci_);
        }
        islot = islot + 1 == NS ? 0 : islot + 1;
    };
    auto read_frags = [&](int g) {
        const char* sb = lds + slot * WS2R_SLOT;
        m_t2 = 2 * cc_.t;
        m_n = cc_.n;
        m_i0 = 2 * (cc_.y0 + m_t2) - 1;  // image row of R0
        m_first = cc_.t == 0;            // first step of an item: no previous gout row
        m_have = true;
        n_a = frag(sb + offA);
        n_b = frag(sb + WR_GROW + offA);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            xr[r][0] = frag(sb + r * WS2R_XROW + offO0);
            xr[r][1] = frag(sb + r * WS2R_XROW + offE);
            xr[r][2] = frag(sb + r * WS2R_XROW + offO1);
        }
        if (g + 1 < nst && ++cc_.t == cc_.T) load_item(cc_, cc_.u + cc_.n);
        slot = slot + 1 == NS ? 0 : slot + 1;
    };
    for (int g = 0; g <= nst; ++g) {
        const bool do_issue = g + D < nst;
        if (g < nst) {
            const int k = max(0, min(D - 1, nst - g - 1));  // younger steps whose pieces may still be in flight
            if (pw == 0) {
                if (k >= 1) wr_vmcnt<7>();
                else wr_vmcnt<0>();
            } else {
                if (k >= 1) wr_vmcnt<5>();
                else wr_vmcnt<0>();
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (!lag) read_frags(g);
        }
        auto piece = [&](int kk) {
            if (do_issue) {
                __builtin_amdgcn_sched_barrier(0);
                issue_piece(ci_, islot, kk);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        {
            const int t2 = m_t2;
            auto vr = [&](int r) { return (unsigned)r < (unsigned)m_n; };       // gout row r of the item exists
            auto vx = [&](int k) { return (unsigned)(m_i0 + k) < (unsigned)IH; };  // input row R_k inside the image
            const bool live = a_live && m_have;
            if (live && vx(0) && !m_first && w_valid) {  // previous step's second gout row, ky = 2
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) acc[2][dx] = Mma16<T>::mma(w_b, xr[0][dx], acc[2][dx]);
            }
            piece(0);
            if (live && vx(0) && vr(t2)) {
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) acc[0][dx] = Mma16<T>::mma(n_a, xr[0][dx], acc[0][dx]);
            }
            piece(1);
            piece(2);
            if (live && vx(1) && vr(t2)) {
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) acc[1][dx] = Mma16<T>::mma(n_a, xr[1][dx], acc[1][dx]);
            }
            piece(3);
            if (live && vx(2) && vr(t2)) {
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) acc[2][dx] = Mma16<T>::mma(n_a, xr[2][dx], acc[2][dx]);
            }
            piece(4);
            if (live && vx(2) && vr(t2 + 1)) {
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) acc[0][dx] = Mma16<T>::mma(n_b, xr[2][dx], acc[0][dx]);
            }
            piece(5);
            piece(6);
            if (live && vx(3) && vr(t2 + 1)) {
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) acc[1][dx] = Mma16<T>::mma(n_b, xr[3][dx], acc[1][dx]);
            }
            if (do_issue) advance_issue(g);
            if (m_have) {
                if (do_bias) {  // rows outside the item arrive as zeros
#pragma unroll
                    for (int j = 0; j < 8; ++j) bsum += Mma16<T>::f32(n_a[j]) + Mma16<T>::f32(n_b[j]);
                }
                w_b = n_b;
                w_valid = vr(t2 + 1);
                m_have = false;
            }
        }
        if (lag && g < nst) read_frags(g);
    }

    // ---- the two halves' accumulators are summed through LDS: waves 4-7 deposit, waves 0-3 add and write the slab ----
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    float* red = reinterpret_cast<float*>(lds) + (wave & 3) * (9 * 4 * 256) + lane * 4;
    if (hf == 1) {
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                for (int j4 = 0; j4 < 4; ++j4)
                    *reinterpret_cast<float4*>(red + ((dy * 3 + dx) * 4 + j4) * 256) =
                        make_float4(acc[dy][dx][4 * j4], acc[dy][dx][4 * j4 + 1], acc[dy][dx][4 * j4 + 2], acc[dy][dx][4 * j4 + 3]);
    }
    __syncthreads();
    if (do_bias) {
        bsum += __shfl_xor(bsum, 32, 64);
        const int co = co0 + 32 * c_t + (lane & 31);
        if ((lane >> 5) == 0 && co < p.cout) atomicAdd(p.bias_grad + co, bsum);
    }
    if (hf == 1) return;
    const int r = lane & 31, h = lane >> 5;
    const int ci = ci0 + 32 * a_t + r;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            float* dst = p.partial + (((int64_t)split * 9 + dy * 3 + dx) * w_rows) * p.cin_total;
#pragma unroll
            for (int j4 = 0; j4 < 4; ++j4) {
                const float4 o = *reinterpret_cast<const float4*>(red + ((dy * 3 + dx) * 4 + j4) * 256);
                const float v[4] = {acc[dy][dx][4 * j4] + o.x, acc[dy][dx][4 * j4 + 1] + o.y, acc[dy][dx][4 * j4 + 2] + o.z, acc[dy][dx][4 * j4 + 3] + o.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int j = 4 * j4 + e;
                    const int co = co0 + 32 * c_t + (j & 3) + 8 * (j >> 2) + 4 * h;
                    if (ci < p.cin_total && co < w_rows) dst[(int64_t)co * p.cin_total + ci] = v[e];
                }
            }
        }
}

bool falnet_wgrad_rows_s2_applicable(const falnet_wgrad_t& p) {
    if (p.dtype != FALNET_BF16 && p.dtype != FALNET_F16) return false;
    if (p.ntaps != 9 || p.isy != 2 || p.isx != 2 || p.TH != (p.IH + 1) / 2 || p.TW != (p.IW + 1) / 2) return false;
    for (int t = 0; t < 9; ++t)
        if (p.tap_dy[t] != t / 3 - 1 || p.tap_dx[t] != t % 3 - 1) return false;
    if (p.gC % 32 || p.cin_total % 32 || p.nsrc != 1) return false;
    const falnet_src_t& S = p.src[0];
    if (S.C != p.cin_total || S.H != p.IH || S.W != p.IW || S.sx == 0 || S.sy == 0) return false;
    if ((int64_t)p.B * ((p.TW + WR_TW - 1) / WR_TW) * p.TH >= (1ll << 30)) return false;
    return true;
}

int falnet_wgrad_rows_s2_launch(const falnet_wgrad_t& p, hipStream_t st) {
    const int w_rows = (p.gC + 31) / 32 * 32;
    const int ntci = (p.cin_total + 63) / 64, ntco = (w_rows + 63) / 64;
    const int ntiles = ntci * ntco;
    const int nstrips = (p.TW + WR_TW - 1) / WR_TW;
    const dim3 grid((unsigned)(ntiles * p.nsplit));
    if (p.dtype == FALNET_F16) hipLaunchKernelGGL(HIP_KERNEL_NAME(wgrad3x3_rows8s2_kernel<f16_t, 2>), grid, dim3(WR8_THREADS), 0, st, p, w_rows, ntci, ntiles, nstrips);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(wgrad3x3_rows8s2_kernel<bf16_t, 2>), grid, dim3(WR8_THREADS), 0, st, p, w_rows, ntci, ntiles, nstrips);
    FALNET_RETURN_LAUNCH();
}

// does this launch fit the row-streaming kernel?  (16-bit operands, canonical dense 3x3 stride 1, sources at the launch
// size or exactly half of it, at least one 64-channel side)
bool falnet_wgrad_rows_applicable(const falnet_wgrad_t& p) {
    if (p.dtype != FALNET_BF16 && p.dtype != FALNET_F16) return false;
    if (p.ntaps != 9 || p.isy != 1 || p.isx != 1 || p.TH != p.IH || p.TW != p.IW) return false;
    for (int t = 0; t < 9; ++t)
        if (p.tap_dy[t] != t / 3 - 1 || p.tap_dx[t] != t % 3 - 1) return false;
    if (p.gC % 32 || p.cin_total % 32 || p.nsrc < 1 || p.nsrc > 2) return false;
    int ctot = 0;
    for (int s = 0; s < p.nsrc; ++s) {
        const falnet_src_t& S = p.src[s];
        if (S.C % 32) return false;
        if (!((S.H == p.IH || 2 * S.H == p.IH) && (S.W == p.IW || 2 * S.W == p.IW))) return false;
        ctot += S.C;
    }
    if (ctot != p.cin_total) return false;
    if ((int64_t)p.B * ((p.TW + WR_TW - 1) / WR_TW) * p.TH >= (1ll << 30)) return false;
    // up2: gout is the upstream gradient at [B][2 TH][2 TW][gC], ONE source at the launch size, splits in whole groups of the four parity classes
    if (p.up2 && (p.nsrc != 1 || p.src[0].H != p.IH || p.src[0].W != p.IW || p.nsplit < 4 || (p.nsplit & 3) || (int64_t)8 * p.TW * p.gC * 2 >= (1ll << 31))) return false;
    return true;
}

int falnet_wgrad_rows_launch(const falnet_wgrad_t& p, hipStream_t st) {
    const int w_rows = (p.gC + 31) / 32 * 32;
    const int ntci = (p.cin_total + 63) / 64, ntco = (w_rows + 63) / 64;
    const int ntiles = ntci * ntco;
    const int nstrips = (p.TW + WR_TW - 1) / WR_TW;
    const dim3 grid((unsigned)(ntiles * p.nsplit));
#define WR_LAUNCH8(TT, DD, AA) hipLaunchKernelGGL(HIP_KERNEL_NAME(wgrad3x3_rows8_kernel<TT, DD, AA>), grid, dim3(WR8_THREADS), 0, st, p, w_rows, ntci, ntiles, nstrips)
#ifdef FALNET_AB
    // Experiment build only (python -m fal_net_amd._build --ab): timing ablations / stamp builds that do NOT compute the gradient
    // (FALNET_WR_ABL) and the four-wave form (FALNET_WR_FORM=4).  The product library contains none of these instantiations.
    static const int abl = [] { const char* e = falnet_ab_env("FALNET_WR_ABL"); return e ? atoi(e) : 0; }();
    static const int form = [] { const char* e = falnet_ab_env("FALNET_WR_FORM"); return e ? atoi(e) : 8; }();
#define WR_LAUNCH4(TT, DD, AA) hipLaunchKernelGGL(HIP_KERNEL_NAME(wgrad3x3_rows_kernel<TT, DD, AA>), grid, dim3(WR_THREADS), 0, st, p, w_rows, ntci, ntiles, nstrips)
    if (form == 4) {
        if (p.dtype == FALNET_F16) WR_LAUNCH4(f16_t, 3, 0);
        else WR_LAUNCH4(bf16_t, 3, 0);
        FALNET_RETURN_LAUNCH();
    }
    if (p.dtype != FALNET_F16 && abl != 0) {
        if (abl == 1) WR_LAUNCH8(bf16_t, 2, 1);
        else if (abl == 2) WR_LAUNCH8(bf16_t, 2, 2);
        else if (abl == 3) WR_LAUNCH8(bf16_t, 2, 3);
        else if (abl == 13) WR_LAUNCH8(bf16_t, 3, 0);
        else if (abl == 4) WR_LAUNCH8(bf16_t, 2, 4);
        else if (abl == 5) WR_LAUNCH8(bf16_t, 2, 5);
        else if (abl == 9) WR_LAUNCH8(bf16_t, 2, 9);
        else if (abl == 21) hipLaunchKernelGGL(HIP_KERNEL_NAME(wgrad3x3_rows8_kernel<bf16_t, 2, 0, true, false>), grid, dim3(WR8_THREADS), 0, st, p, w_rows, ntci, ntiles, nstrips);
        else if (abl == 22) hipLaunchKernelGGL(HIP_KERNEL_NAME(wgrad3x3_rows8_kernel<bf16_t, 2, 0, false, true>), grid, dim3(WR8_THREADS), 0, st, p, w_rows, ntci, ntiles, nstrips);
        else if (abl == 20) hipLaunchKernelGGL(HIP_KERNEL_NAME(wgrad3x3_rows8_kernel<bf16_t, 2, 0, false, false>), grid, dim3(WR8_THREADS), 0, st, p, w_rows, ntci, ntiles, nstrips);
        else if (abl == 29) hipLaunchKernelGGL(HIP_KERNEL_NAME(wgrad3x3_rows8_kernel<bf16_t, 2, 9, false, false>), grid, dim3(WR8_THREADS), 0, st, p, w_rows, ntci, ntiles, nstrips);
        else {
            falnet_set_error("wgrad rows: unknown FALNET_WR_ABL=%d", abl);
            return -1;
        }
        FALNET_RETURN_LAUNCH();
    }
#endif
#ifdef FALNET_AB  // (experiment builds only) FALNET_WR_FORM=8: the round-2..4 kernel on v_mfma_f32_32x32x16 (two 16-pixel wave groups, LDS reduction);
    // FALNET_WR_FORM=84: the same with FOUR steps per barrier over an eight-slot ring (neutral on the step: profiles/r05_ab_rows8_spb4.txt)
    if (form == 8 && falnet_ab_env("FALNET_WR_FORM")) {
        if (p.dtype == FALNET_F16) hipLaunchKernelGGL(HIP_KERNEL_NAME(wgrad3x3_rows8_kernel<f16_t, 4, 0, true, true, 2>), grid, dim3(WR8_THREADS), 0, st, p, w_rows, ntci, ntiles, nstrips);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(wgrad3x3_rows8_kernel<bf16_t, 4, 0, true, true, 2>), grid, dim3(WR8_THREADS), 0, st, p, w_rows, ntci, ntiles, nstrips);
        FALNET_RETURN_LAUNCH();
    }
    if (form == 84) {
        if (p.dtype == FALNET_F16) hipLaunchKernelGGL(HIP_KERNEL_NAME(wgrad3x3_rows8_kernel<f16_t, 4, 0, true, true, 4>), grid, dim3(WR8_THREADS), 0, st, p, w_rows, ntci, ntiles, nstrips);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(wgrad3x3_rows8_kernel<bf16_t, 4, 0, true, true, 4>), grid, dim3(WR8_THREADS), 0, st, p, w_rows, ntci, ntiles, nstrips);
        FALNET_RETURN_LAUNCH();
    }
#endif
    // the product kernel: v_mfma_f32_16x16x32 (round 5: -7 % on the launches alone, -0.9 % on the step: profiles/r05_ab_rows16.txt)
    if (p.up2) {  // a `deconv` layer's weight gradient on the low-resolution grid: splits = (pixel range, parity class)
        if (p.dtype == FALNET_F16) hipLaunchKernelGGL(HIP_KERNEL_NAME(wgrad3x3_rows16_kernel<f16_t, 4, 2, true>), grid, dim3(WR8_THREADS), 0, st, p, w_rows, ntci, ntiles, nstrips);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(wgrad3x3_rows16_kernel<bf16_t, 4, 2, true>), grid, dim3(WR8_THREADS), 0, st, p, w_rows, ntci, ntiles, nstrips);
        FALNET_RETURN_LAUNCH();
    }
    if (p.dtype == FALNET_F16) hipLaunchKernelGGL(HIP_KERNEL_NAME(wgrad3x3_rows16_kernel<f16_t, 4, 2, false>), grid, dim3(WR8_THREADS), 0, st, p, w_rows, ntci, ntiles, nstrips);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(wgrad3x3_rows16_kernel<bf16_t, 4, 2, false>), grid, dim3(WR8_THREADS), 0, st, p, w_rows, ntci, ntiles, nstrips);
    FALNET_RETURN_LAUNCH();
}
