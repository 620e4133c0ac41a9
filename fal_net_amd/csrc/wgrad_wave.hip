// Wave-streaming weight gradient of the dense 3x3 / stride-1 convolutions with a 32-channel input (16-bit operands), gfx950
// (falnet_wgrad variant 9, round 6).
//
//   dW[co][ky][kx][ci] = sum over pixels of gout[y][x][co] * in[y + ky - 1][x + kx - 1][ci]
//
// (autograd of the conv2d call sites models/FAL_netB.py:38,45,127 of the reference for the layers whose input has 32 channels: the two
// residual convolutions of conv0_1 -- 32 -> 32 at full resolution -- and the skip-connection group of the logits convolution, 32 -> N.)
//
// Why a kernel of its own.  These layers are HBM-bound (conv0_1 at B = 8, 256 x 512: 19.3 GFLOP over 134 MB of operands -- 144 FLOP / B, the
// ridge of the chip is ~310) and their whole 32 x 32 x 9 block of the gradient is 9 216 f32: 144 accumulator registers of ONE wave.  The
// row-streaming kernel (wgrad3x3_rows16_kernel) shares a 64 x 64 block between the eight waves of a workgroup -- one barrier per two image rows,
// half of its waves without work on a 32-channel input -- and the halo-patch kernel that served these layers until round 5 stages every row 1.5
// times through registers with two barriers per 4 x 32 pixels (255-297 TFLOP/s on the whole chip, 74-100 us per launch, on the step's tail).
// Here NOTHING is shared between waves until the end:
//   * a wave owns a 32 (cout) x 32 (cin) x 9 block and a contiguous range of (sample, 32-pixel column strip, row) units; it streams its rows top
//     to bottom through a private ring of three row slots in LDS, filled by LDS-DMA (global_load_lds_dwordx4: whole 64-B pixel lines, no VGPR
//     staging) two rows ahead, ordered by a counted s_waitcnt vmcnt only -- no s_barrier in the loop, no wave ever waits for another;
//   * per row: the gout fragments of rows i+1, i, i-1 (ky = 0, 1, 2) live in a rolling register window, the input row is read as three shifted
//     fragments (kx = 0, 1, 2): sixteen transposed 8-B reads (ds_read_b64_tr_b16) and 36 v_mfma_f32_16x16x32 (K = the strip row's 32 pixels);
//   * out-of-image rows / columns and the rows past the end of a range come from a 128-B page of zeros, so every step issues the same five
//     pieces and the vmcnt arithmetic is a constant;
//   * eight waves per workgroup (one workgroup per CU, grid = 256): 8 / NCO pixel ranges x NCO 32-channel halves of the output channels; the
//     ranges' accumulators are summed through LDS in a fixed tree (deterministic) and ONE slab per workgroup goes to partial[wg][9][cout][32]
//     for falnet_wgrad_reduce_batched.
// S2 (round 6): the 3x3 / STRIDE-2 / pad-1 form (conv1, FAL_netB.py:101: 32 + 1 -> 64 channels; the constant `flow` channel has its own kernel,
// falnet_wgrad_const_plane).  in[2 y + ky - 1][2 x + kx - 1] is a pixel of one of the four PARITY PLANES P_ab[u][v] = in[2 u + a][2 v + b] at offset
// (ky, kx) -> plane a = (ky + 1) & 1, row offset -1 (ky 0) or 0 (ky 1, 2), the same per column: plane (0,0) carries tap (1,1), (0,1) taps (1,0) (1,2),
// (1,0) taps (0,1) (2,1), (1,1) the four corner taps.  A plane is just another (pointer, 2 sy, 2 sx) view of the source -- the LDS-DMA names every pixel
// anyway -- so the SAME loop runs: the four pixel-range slots of a workgroup become the four planes of ONE pixel range (two output-channel halves
// each), a wave accumulates the (row offset, column offset) products of its plane -- at most 2 x 2 -- and writes them to their taps of the
// workgroup's slab itself: the planes' taps are disjoint and cover the 3x3 block, no reduction.  Items take n + 1 steps (row offsets -1, 0 only).
// LDS image of a row: [pixel][32 channels], 64-B rows as the DMA writes them; the two 32-B halves of the pixels with bit 3 set are exchanged
// (on the SOURCE address: the DMA destination is lane-linear), which makes the transposed reads of the two 4-pixel blocks a 32-lane half takes
// (8 pixels apart) conflict-free for every column offset.
#include "common.h"

typedef short wv_s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* wv_lptr_t;
typedef wv_s16x4 __attribute__((address_space(3))) * wv_lds_v4;

#define WV_THREADS 512
#define WV_GROW 2048                 // gout row: 32 px x 32 ch x 2 B
#define WV_XROW 3072                 // input row: 34 px used (the +-1 column halo), three 16-pixel DMA pieces
#define WV_SLOT (WV_GROW + WV_XROW)
#define WV_D 2                       // rows of prefetch
#define WV_NS (WV_D + 1)
#define WV_RING (WV_NS * WV_SLOT)    // 15 KiB per wave
#define WV_PIECES 5                  // DMA pieces per step and wave
#define WV_RED (9 * 4096)            // one wave's accumulators: 9 taps x 32 x 32 f32
#define WV_LDS (4 * WV_RED)          // 144 KiB: the first round of the reduction tree holds four waves' accumulators (>= 8 rings = 120 KiB)

__device__ uint4 g_wv_zero[8] = {};  // 128 B of zeros: source of every out-of-image / out-of-range 16-B piece

// One 1-KiB LDS-DMA piece: lane l's 16 B from its own global address to LDS byte lds_dst + 16 l (inline asm: outside hipcc's LDS-DMA alias
// bookkeeping, which would drain the ring with vmcnt(0) in front of every ds_read; the counted waits below are the only ordering.  M0 is
// saved and restored inside the statement: compiler-reserved).
__device__ __forceinline__ void wv_glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}

struct WvItem { int u, n, s, b, x0, y0; };  // rows [y0, y0 + n) of column strip x0 of sample b, first unit u; s = step inside it (0 .. n + 1)

template <typename T, int NCO, bool S2>
__global__ __launch_bounds__(WV_THREADS) void wgrad3x3_wave32_kernel(const falnet_wgrad_t p, int nstrips) {
    static_assert(sizeof(T) == 2 && (NCO == 1 || NCO == 2) && (!S2 || NCO == 2), "16-bit operands; 32 or 64 output channels; stride 2: 64");
    constexpr int XS = S2 ? 1 : 2;  // steps of an item beyond its n rows (input rows y0 - 1 .. y0 + n - 1 | y0 + n)
    constexpr int NP = 8 / NCO;  // pixel ranges per workgroup
    __shared__ __attribute__((aligned(1024))) char lds[WV_LDS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ch = wave % NCO, part = wave / NCO;
    const unsigned ring = (unsigned)(unsigned long)(wv_lptr_t)lds + wave * WV_RING;
    const char* const ring_p = lds + wave * WV_RING;
    const int H = p.TH, TW = p.TW, gC = p.gC;
    const int pa = S2 ? part >> 1 : 0, pb = S2 ? part & 1 : 0;                      // S2: this wave's parity plane of the source
    const int IH = S2 ? (p.IH - pa + 1) >> 1 : p.IH, IW = S2 ? (p.IW - pb + 1) >> 1 : p.IW;  // (the plane's size)
    const int R = p.B * nstrips * H;
    const int nparts = S2 ? (int)gridDim.x : (int)gridDim.x * NP, gp = S2 ? (int)blockIdx.x : (int)blockIdx.x * NP + part;
    const int u0 = (int)((int64_t)R * gp / nparts), u1 = (int)((int64_t)R * (gp + 1) / nparts);

    // ---- per-lane DMA geometry: lane = (pixel pl of a 16-pixel piece, 16-B position seg of its 64-B line); the segment it FETCHES is swizzled
    // (LDS pixel 16 k + pl has bit 3 of pl: the same for every piece) ----
    const int pl = lane >> 2, seg = lane & 3;
    const int gseg = seg ^ (((pl >> 3) & 1) << 1);
    const char* const zero_page = reinterpret_cast<const char*>(g_wv_zero);
    const falnet_src_t& S = p.src[0];
    const int64_t l_sy = S2 ? 2 * S.sy : S.sy, l_sx = S2 ? 2 * S.sx : S.sx, l_sb = S.sb;
    const T* const x_base = reinterpret_cast<const T*>(S.ptr) + 8 * gseg + (S2 ? pa * S.sy + pb * S.sx : 0);
    const T* const g_base = reinterpret_cast<const T*>(p.gout) + 32 * ch + 8 * gseg;
    const unsigned g_rowb = (unsigned)(TW * gC * (int)sizeof(T)), x_rowb = (unsigned)(l_sy * (int)sizeof(T));

    auto load_item = [&](WvItem& c, int u) {
        c.u = u;
        const int bs = u / H;
        c.y0 = u - bs * H;
        c.n = min((bs + 1) * H, u1) - u;
        c.b = bs / nstrips;
        c.x0 = (bs - c.b * nstrips) * 32;
        c.s = 0;
    };
    int nst = 0;
    for (int u = u0; u < u1;) {
        const int e = min((u / H + 1) * H, u1);
        nst += e - u + XS;
        u = e;
    }
    const char* gptr[2];
    const char* xptr[3];
    unsigned ginc[2], xinc[3];
    auto item_pointers = [&](const WvItem& c) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int gx = c.x0 + 16 * k + pl;
            const bool ok = gx < TW;
            gptr[k] = ok ? reinterpret_cast<const char*>(g_base + (((int64_t)c.b * H + c.y0) * TW + gx) * gC) : zero_page;
            ginc[k] = ok ? g_rowb : 0u;
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int lp = 16 * k + pl, xa = c.x0 - 1 + lp;
            const bool ok = lp < 34 && xa >= 0 && xa < IW;
            xptr[k] = ok ? reinterpret_cast<const char*>(x_base + (int64_t)c.b * l_sb + (int64_t)(c.y0 - 1) * l_sy + (int64_t)xa * l_sx) : zero_page;
            xinc[k] = ok ? x_rowb : 0u;
        }
    };
    // step s of an item: input row i = y0 - 1 + s (zeros outside the image) and gout row y0 + s (zeros past the item's last row)
    auto issue = [&](const WvItem& c, int slot, bool real) {
        const unsigned base = ring + slot * WV_SLOT;
        const bool gv = real && c.s < c.n;
        const int i = c.y0 - 1 + c.s;
        const bool xv = real && i >= 0 && i < IH;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            wv_glds16(gv ? gptr[k] : zero_page, base + k * 1024);
            gptr[k] += ginc[k];
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            wv_glds16(xv ? xptr[k] : zero_page, base + WV_GROW + k * 1024);
            xptr[k] += xinc[k];
        }
    };

    // ---- fragment read geometry: 16-lane group g16 reads pixels 8 g16 + q (+ 4 in the second read) + column offset of a 16-channel tile ----
    const int i16 = lane & 15, g16 = lane >> 4;
    const int q = i16 >> 2, pc = i16 & 3;
    auto frag_off = [&](int tile, int pshift, int second_read) {
        const int px = 8 * g16 + q + pshift + 4 * second_read;
        const int sg = (2 * tile + (pc >> 1)) ^ (((px >> 3) & 1) << 1);
        return px * 64 + sg * 16 + (pc & 1) * 8;
    };
    int offA[2][2], offB[3][2][2];  // [tile][read], [column offset][tile][read]
#pragma unroll
    for (int rd = 0; rd < 2; ++rd)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            offA[t][rd] = frag_off(t, 0, rd);
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) offB[dx][t][rd] = WV_GROW + frag_off(t, dx, rd);
        }
    auto frag = [&](const char* base, const int (&off)[2]) -> s16x8_t {
        const wv_s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((wv_lds_v4)(base + off[0]));
        const wv_s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((wv_lds_v4)(base + off[1]));
        return __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
    };

    constexpr int NK = S2 ? 2 : 3;
    f32x4_t acc[NK][NK][2][2];  // [ky][kx][cout tile][cin tile]; S2: [row offset -1, 0][column offset -1, 0] of this wave's parity plane
#pragma unroll
    for (int ky = 0; ky < NK; ++ky)
#pragma unroll
        for (int kx = 0; kx < NK; ++kx)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int at = 0; at < 2; ++at) acc[ky][kx][ct][at] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    s16x8_t g1[2], g2[2];  // gout fragments of rows i and i - 1 (the rolling window), per 16-channel tile
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int j = 0; j < 8; ++j) g1[t][j] = g2[t][j] = 0;
    // fused bias gradient (db[co] += sum of gout[., co]; f32 atomics: falnet_wgrad_fuses_bias says when) from the gout fragments every range reads
    // anyway; S2: the four planes of a range read the same rows -- plane (0, 0) sums them
    const bool do_bias = p.bias_grad != nullptr && (!S2 || part == 0);
    float bsum[2] = {0.f, 0.f};

    WvItem ci_, cc_;
    ci_ = WvItem{0, 0, 0, 0, 0, 0};
    if (nst > 0) {
        load_item(ci_, u0);
        item_pointers(ci_);
    }
    cc_ = ci_;
    int issued = 0;
    auto issue_next = [&](int slot) {  // every step issues exactly WV_PIECES pieces (zeros past the end): the counted waits are constants
        const bool real = issued < nst;
        issue(ci_, slot, real);
        if (real && ++issued < nst && ++ci_.s == ci_.n + XS) {
            load_item(ci_, ci_.u + ci_.n);
            item_pointers(ci_);
        }
    };
#pragma unroll
    for (int d = 0; d < WV_D; ++d) issue_next(d);
    int slot = 0, islot = WV_D;
    for (int g = 0; g < nst; ++g) {
        issue_next(islot);  // step g + D into the slot step g - 1 has finished reading
        islot = islot + 1 == WV_NS ? 0 : islot + 1;
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WV_D * WV_PIECES) : "memory");  // all but the 2 x 5 youngest pieces: step g's rows have landed
        const char* sb = ring_p + slot * WV_SLOT;
        slot = slot + 1 == WV_NS ? 0 : slot + 1;
        const int s = cc_.s, n = cc_.n;
        const int i = cc_.y0 - 1 + s;
        const bool xv = i >= 0 && i < IH;
        // (no window reset at a new item: ky 1 / ky 2 are skipped until the window holds this item's rows)
        s16x8_t g0[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) g0[t] = frag(sb, offA[t]);
        if (do_bias && s < n) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const unsigned w = ((unsigned)(unsigned short)g0[t][2 * j]) | ((unsigned)(unsigned short)g0[t][2 * j + 1] << 16);
                    bsum[t] += H16<T>::lo(w) + H16<T>::hi(w);
                }
        }
        auto mm = [&](f32x4_t (&A)[2][2], const s16x8_t (&G)[2], const s16x8_t (&X)[2]) {
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int at = 0; at < 2; ++at) A[ct][at] = H16<T>::mma16(G[ct], X[at], A[ct][at]);
        };
        if constexpr (S2) {
            // plane row i = y0 - 1 + s meets gout rows i + 1 (row offset -1: s < n) and i (offset 0: 1 <= s <= n); offset -1 exists on the odd planes
            // only (a = 1: tap ky 0; their offset 0 is ky 2), the even planes carry ky 1 at offset 0 -- the same per column (kx); wave-uniform tests
            if (xv) {
                const bool r0 = pa != 0 && s < n, r1 = s >= 1 && s <= n;
                s16x8_t x1[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) x1[t] = frag(sb, offB[1][t]);  // column offset 0
                if (r0) mm(acc[0][1], g0, x1);
                if (r1) mm(acc[1][1], g1, x1);
                if (pb != 0) {
                    s16x8_t x0[2];
#pragma unroll
                    for (int t = 0; t < 2; ++t) x0[t] = frag(sb, offB[0][t]);  // column offset -1
                    if (r0) mm(acc[0][0], g0, x0);
                    if (r1) mm(acc[1][0], g1, x0);
                }
            }
        } else if (xv) {
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                s16x8_t xf[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) xf[t] = frag(sb, offB[kx][t]);
                // input row i meets gout rows i + 1 (ky 0: step s < n), i (ky 1: 1 <= s <= n), i - 1 (ky 2: s >= 2); the tests are wave-uniform
                if (s < n) mm(acc[0][kx], g0, xf);
                if (s >= 1 && s <= n) mm(acc[1][kx], g1, xf);
                if constexpr (!S2) {
                    if (s >= 2) mm(acc[2][kx], g2, xf);
                }
            }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            g2[t] = g1[t];
            g1[t] = g0[t];
        }
        if (g + 1 < nst && ++cc_.s == cc_.n + XS) load_item(cc_, cc_.u + cc_.n);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the trailing zero pieces: nothing may land in LDS once the rings are re-used below

    const int w_rows = gC;  // (32 or 64: checked by falnet_wgrad_wave_applicable)
    if (p.bias_grad != nullptr) {  // (workgroup-uniform) lane (i16, g16) summed pixels 8 g16 .. + 7 of channel 32 ch + 16 t + i16: 64 sums per workgroup in LDS
        float* const bred = reinterpret_cast<float*>(lds);
        __syncthreads();  // every wave is done with its ring
        if (tid < 64) bred[tid] = 0.f;
        __syncthreads();
        if (do_bias) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                float b = bsum[t];
                b += __shfl_xor(b, 16, 64);
                b += __shfl_xor(b, 32, 64);
                if (g16 == 0) atomicAdd(&bred[32 * ch + 16 * t + i16], b);
            }
        }
        __syncthreads();
        if (tid < 32 * NCO && tid < p.cout) atomicAdd(p.bias_grad + tid, bred[tid]);
        __syncthreads();  // (the reduction below re-uses the bytes)
    }
    auto store_tap = [&](int tap, const f32x4_t (&A)[2][2]) {  // tile (ct, at) of D[co 16 x ci 16]: lane (ci = i16, rows 4 g16 + f)
        float* dst = p.partial + (((int64_t)blockIdx.x * 9 + tap) * w_rows) * 32;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int at = 0; at < 2; ++at)
#pragma unroll
                for (int f = 0; f < 4; ++f) {
                    const int co = 32 * ch + 16 * ct + 4 * g16 + f;
                    dst[(int64_t)co * 32 + 16 * at + i16] = A[ct][at][f];
                }
    };
    if constexpr (S2) {
        // every wave writes its plane's taps of the workgroup's slab (disjoint, together the whole 3x3 block): plane a -> ky 1 (a = 0) or ky 0, 2
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                if ((r == 0 && pa == 0) || (c == 0 && pb == 0)) continue;  // (wave-uniform)
                const int ky = pa ? 2 * r : 1, kx = pb ? 2 * c : 1;
                store_tap(ky * 3 + kx, acc[r][c]);
            }
    } else {
        // ---- sum the NP ranges of every output-channel half through LDS (fixed tree: deterministic), then one slab per workgroup ----
        float* const red = reinterpret_cast<float*>(lds);
        auto red_off = [&](int buf, int ky, int kx, int ct, int at) {  // f32x4 per lane, lane-linear 1-KiB blocks
            return (size_t)buf * (WV_RED / 4) + (size_t)((((ky * 3 + kx) * 2 + ct) * 2 + at) * 256) + lane * 4;
        };
#pragma unroll
        for (int stride = NP / 2; stride >= 1; stride >>= 1) {
            __syncthreads();  // (first round: every wave is done with its ring; later rounds: the previous round's readers)
            if (part >= stride && part < 2 * stride) {
                const int buf = (part - stride) * NCO + ch;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                            for (int at = 0; at < 2; ++at) *reinterpret_cast<f32x4_t*>(red + red_off(buf, ky, kx, ct, at)) = acc[ky][kx][ct][at];
            }
            __syncthreads();
            if (part < stride) {
                const int buf = part * NCO + ch;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                            for (int at = 0; at < 2; ++at) acc[ky][kx][ct][at] += *reinterpret_cast<const f32x4_t*>(red + red_off(buf, ky, kx, ct, at));
            }
        }
        if (part == 0) {
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) store_tap(ky * 3 + kx, acc[ky][kx]);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// wgrad3x3_c3wave_kernel (falnet_wgrad variant 6 when the image width is a multiple of 4, round 6): the FIRST layer's weight gradient
// (FAL_netB.py:99 conv0: 3 -> 32, planar f32 [B][3][H][W] image source) in the same wave-streaming form.
//   dW[co][ky][kx][c] = sum over pixels of gout[y][x][co] * img[c][y + ky - 1][x + kx - 1]
// is HBM-bound by the 32-channel gout tensor (67 MB at B = 8, 256 x 512, against 12.6 MB of image): 1.8 GFLOP.  GEMM view per tap row ky:
// D[co 32][n 16] += A[co][p] B[p][n], n = 3 kx + c (9 used), K = the 32 pixels of a strip row.  A = the transposed gout reads of the kernel above
// (rolling window over ky); B = eight consecutive image columns of the lane's (c, kx) from the f32 row in LDS, converted in registers.  A step
// moves one gout row (two pieces) and ONE image row (one piece: 3 channels x 40 columns from column x0 - 4, so that every 16-B lane of the DMA is
// aligned and wholly inside or outside the image) and issues six MFMAs: the kernel is a copy loop with a product attached.  The halo-patch kernel
// it replaces (wgrad3x3_c3_kernel: one barrier per 4 x 32 pixels, register staging) ran at 1.6 TB/s (56 us), this form at the chip's LDS-DMA rate.
// The bias gradient (sum of gout) is taken from the A fragments as before (f32 atomics, one per channel and workgroup; not in deterministic mode).
#define WC_XROW 1024                 // image row: one piece, lane 10 c + j = columns x0 - 4 + 4 j .. + 3 of channel c (lanes 30-63: zeros)
#define WC_SLOT (WV_GROW + WC_XROW)
#define WC_D 3
#define WC_NS (WC_D + 1)
#define WC_RING (WC_NS * WC_SLOT)    // 12 KiB per wave
#define WC_PIECES 3

template <typename T>
__global__ __launch_bounds__(WV_THREADS) void wgrad3x3_c3wave_kernel(const falnet_wgrad_t p, int nstrips) {
    static_assert(sizeof(T) == 2, "16-bit gout");
    __shared__ __attribute__((aligned(1024))) char lds[8 * WC_RING];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned ring = (unsigned)(unsigned long)(wv_lptr_t)lds + wave * WC_RING;
    const char* const ring_p = lds + wave * WC_RING;
    const int H = p.TH, TW = p.TW, gC = p.gC, IH = p.IH, IW = p.IW;
    const int R = p.B * nstrips * H;
    const int nparts = (int)gridDim.x * 8, gp = (int)blockIdx.x * 8 + wave;
    const int u0 = (int)((int64_t)R * gp / nparts), u1 = (int)((int64_t)R * (gp + 1) / nparts);

    const int pl = lane >> 2, seg = lane & 3;
    const int gseg = seg ^ (((pl >> 3) & 1) << 1);
    const char* const zero_page = reinterpret_cast<const char*>(g_wv_zero);
    const T* const g_base = reinterpret_cast<const T*>(p.gout) + 8 * gseg;
    const float* const img = reinterpret_cast<const float*>(p.src[0].ptr);
    const unsigned g_rowb = (unsigned)(TW * gC * (int)sizeof(T)), x_rowb = (unsigned)(IW * 4);
    const int xc = lane / 10, xj = lane - 10 * xc;  // image piece: lane -> (channel, 16-B granule); lanes >= 30 carry zeros

    auto load_item = [&](WvItem& c, int u) {
        c.u = u;
        const int bs = u / H;
        c.y0 = u - bs * H;
        c.n = min((bs + 1) * H, u1) - u;
        c.b = bs / nstrips;
        c.x0 = (bs - c.b * nstrips) * 32;
        c.s = 0;
    };
    int nst = 0;
    for (int u = u0; u < u1;) {
        const int e = min((u / H + 1) * H, u1);
        nst += e - u + 2;
        u = e;
    }
    const char* gptr[2];
    const char* xptr;
    unsigned ginc[2], xinc;
    auto item_pointers = [&](const WvItem& c) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int gx = c.x0 + 16 * k + pl;
            const bool ok = gx < TW;
            gptr[k] = ok ? reinterpret_cast<const char*>(g_base + (((int64_t)c.b * H + c.y0) * TW + gx) * gC) : zero_page;
            ginc[k] = ok ? g_rowb : 0u;
        }
        const int col = c.x0 - 4 + 4 * xj;  // (IW is a multiple of 4: a granule is wholly inside or outside the image)
        const bool ok = lane < 30 && col >= 0 && col < IW;
        xptr = ok ? reinterpret_cast<const char*>(img + ((int64_t)c.b * 3 + xc) * IH * IW + (int64_t)(c.y0 - 1) * IW + col) : zero_page;
        xinc = ok ? x_rowb : 0u;
    };
    auto issue = [&](const WvItem& c, int slot, bool real) {
        const unsigned base = ring + slot * WC_SLOT;
        const bool gv = real && c.s < c.n;
        const int i = c.y0 - 1 + c.s;
        const bool xv = real && i >= 0 && i < IH;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            wv_glds16(gv ? gptr[k] : zero_page, base + k * 1024);
            gptr[k] += ginc[k];
        }
        wv_glds16(xv ? xptr : zero_page, base + WV_GROW);
        xptr += xinc;
    };

    const int i16 = lane & 15, g16 = lane >> 4;
    const int q = i16 >> 2, pc = i16 & 3;
    auto frag_off = [&](int tile, int second_read) {
        const int px = 8 * g16 + q + 4 * second_read;
        const int sg = (2 * tile + (pc >> 1)) ^ (((px >> 3) & 1) << 1);
        return px * 64 + sg * 16 + (pc & 1) * 8;
    };
    int offA[2][2];
#pragma unroll
    for (int rd = 0; rd < 2; ++rd)
#pragma unroll
        for (int t = 0; t < 2; ++t) offA[t][rd] = frag_off(t, rd);
    auto frag = [&](const char* base, const int (&off)[2]) -> s16x8_t {
        const wv_s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((wv_lds_v4)(base + off[0]));
        const wv_s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((wv_lds_v4)(base + off[1]));
        return __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
    };
    // B fragment: lane n = i16 = 3 kx + c (n < 9), pixels 8 g16 .. + 7 -> image columns x0 + 8 g16 + j + kx - 1 = float 3 + kx + 8 g16 + j of channel c's 40
    const int nkx = i16 / 3, nc = i16 - 3 * nkx;
    const bool nvalid = i16 < 9;
    const int offX = WV_GROW + (nvalid ? (nc * 40 + 3 + nkx + 8 * g16) * 4 : 0);

    f32x4_t acc[3][2];  // [ky][cout tile]: D[co 16][n 16]
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) acc[ky][ct] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    s16x8_t g1[2], g2[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int j = 0; j < 8; ++j) g1[t][j] = g2[t][j] = 0;
    const bool do_bias = p.bias_grad != nullptr;
    float bsum[2] = {0.f, 0.f};

    WvItem ci_, cc_;
    ci_ = WvItem{0, 0, 0, 0, 0, 0};
    if (nst > 0) {
        load_item(ci_, u0);
        item_pointers(ci_);
    }
    cc_ = ci_;
    int issued = 0;
    auto issue_next = [&](int slot) {
        const bool real = issued < nst;
        issue(ci_, slot, real);
        if (real && ++issued < nst && ++ci_.s == ci_.n + 2) {
            load_item(ci_, ci_.u + ci_.n);
            item_pointers(ci_);
        }
    };
#pragma unroll
    for (int d = 0; d < WC_D; ++d) issue_next(d);
    int slot = 0, islot = WC_D;
    for (int g = 0; g < nst; ++g) {
        issue_next(islot);
        islot = islot + 1 == WC_NS ? 0 : islot + 1;
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WC_D * WC_PIECES) : "memory");
        const char* sb = ring_p + slot * WC_SLOT;
        slot = slot + 1 == WC_NS ? 0 : slot + 1;
        const int s = cc_.s, n = cc_.n;
        const int i = cc_.y0 - 1 + s;
        const bool xv = i >= 0 && i < IH;
        s16x8_t g0[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) g0[t] = frag(sb, offA[t]);
        if (do_bias && s < n) {  // (rows past the item arrive as zeros anyway)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const unsigned w = ((unsigned)(unsigned short)g0[t][2 * j]) | ((unsigned)(unsigned short)g0[t][2 * j + 1] << 16);
                    bsum[t] += H16<T>::lo(w) + H16<T>::hi(w);
                }
        }
        if (xv) {
            const float* xr = reinterpret_cast<const float*>(sb + offX);
            s16x8_t bv;
#pragma unroll
            for (int j = 0; j < 8; ++j) bv[j] = (short)H16<T>::bits(nvalid ? xr[j] : 0.f);
            if (s < n) {
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) acc[0][ct] = H16<T>::mma16(g0[ct], bv, acc[0][ct]);
            }
            if (s >= 1 && s <= n) {
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) acc[1][ct] = H16<T>::mma16(g1[ct], bv, acc[1][ct]);
            }
            if (s >= 2) {
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) acc[2][ct] = H16<T>::mma16(g2[ct], bv, acc[2][ct]);
            }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            g2[t] = g1[t];
            g1[t] = g0[t];
        }
        if (g + 1 < nst && ++cc_.s == cc_.n + 2) load_item(cc_, cc_.u + cc_.n);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- the eight waves' tiles and bias sums meet in LDS (f32 atomics on LDS: 24 + 2 per lane), one slab and 32 global atomics per workgroup ----
    float* const red = reinterpret_cast<float*>(lds);  // [ky][ct][lane][4] = 1536 floats, then 32 bias sums
    __syncthreads();
    for (int e = tid; e < 1536 + 32; e += WV_THREADS) red[e] = 0.f;
    __syncthreads();
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int f = 0; f < 4; ++f) atomicAdd(&red[((ky * 2 + ct) * 64 + lane) * 4 + f], acc[ky][ct][f]);
    if (do_bias) {  // lane (i16, g16) summed pixels 8 g16 .. + 7 of channel 16 t + i16
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float b = bsum[t];
            b += __shfl_xor(b, 16, 64);
            b += __shfl_xor(b, 32, 64);
            if (g16 == 0) atomicAdd(&red[1536 + 16 * t + i16], b);
        }
    }
    __syncthreads();
    // D tile (ky, ct): lane (n = i16, rows co = 16 ct + 4 g16 + f); slab [tap = 3 ky + kx][co][cin_total], column c
    for (int e = tid; e < 1536; e += WV_THREADS) {
        const int f = e & 3, ln = (e >> 2) & 63, kc = e >> 8;  // kc = 2 ky + ct
        const int n = ln & 15, gg = ln >> 4;
        if (n < 9) {
            const int ky = kc >> 1, ct = kc & 1, kx = n / 3, c = n - 3 * kx;
            const int co = 16 * ct + 4 * gg + f;
            p.partial[(((int64_t)blockIdx.x * 9 + 3 * ky + kx) * gC + co) * p.cin_total + c] = red[e];
        }
    }
    if (do_bias && tid < 32 && tid < p.cout) atomicAdd(p.bias_grad + tid, red[1536 + tid]);
}

bool falnet_wgrad_c3wave_applicable(const falnet_wgrad_t& p) {  // (the caller has checked the variant-6 contract: 16-bit, canonical taps, gC 32, cin_total 32)
    return p.IW % 4 == 0 && p.TW >= 32 && p.TH == p.IH && p.TW == p.IW && (((uintptr_t)p.src[0].ptr) & 15) == 0 && !falnet_deterministic() &&
           (int64_t)p.B * ((p.TW + 31) / 32) * p.TH < (1ll << 30) && (int64_t)p.TW * p.gC * 2 < (1ll << 31);
}

int falnet_wgrad_c3wave_launch(const falnet_wgrad_t& p, hipStream_t st) {
    const int nstrips = (p.TW + 31) / 32;
#define WC_L(T) hipLaunchKernelGGL(HIP_KERNEL_NAME(wgrad3x3_c3wave_kernel<T>), dim3((unsigned)p.nsplit), dim3(WV_THREADS), 0, st, p, nstrips)
    FALNET_DISPATCH_16(p.dtype, WC_L);
#undef WC_L
    FALNET_RETURN_LAUNCH();
}

// dense 3x3 / pad 1, 16-bit, ONE 32-channel NHWC source at the input size, strips of 32 output pixels; stride 1: 32 or 64 (padded) output channels;
// stride 2 (TH = ceil(IH / 2), TW = ceil(IW / 2)): 64
bool falnet_wgrad_wave_applicable(const falnet_wgrad_t& p) {
    if (p.dtype != FALNET_BF16 && p.dtype != FALNET_F16) return false;
    if (p.ntaps != 9 || p.up2 || p.isy != p.isx || (p.isy != 1 && p.isy != 2)) return false;
    if (p.isy == 1 && (p.TH != p.IH || p.TW != p.IW)) return false;
    if (p.isy == 2 && (p.TH != (p.IH + 1) / 2 || p.TW != (p.IW + 1) / 2 || p.gC != 64)) return false;
    for (int t = 0; t < 9; ++t)
        if (p.tap_dy[t] != t / 3 - 1 || p.tap_dx[t] != t % 3 - 1) return false;
    if (p.nsrc != 1 || p.cin_total != 32 || p.src[0].C != 32 || p.src[0].H != p.IH || p.src[0].W != p.IW) return false;
    if (p.gC != 32 && p.gC != 64) return false;
    if (p.TW < 32 || p.nsplit < 1) return false;
    if ((int64_t)p.B * ((p.TW + 31) / 32) * p.TH >= (1ll << 30)) return false;
    if ((int64_t)p.TW * p.gC * 2 >= (1ll << 31) || p.src[0].sy * 4 >= (1ll << 31)) return false;  // (32-bit row increments)
    return true;
}

int falnet_wgrad_wave_launch(const falnet_wgrad_t& p, hipStream_t st) {
    const int nstrips = (p.TW + 31) / 32;
    const dim3 grid((unsigned)p.nsplit);  // one slab per workgroup
#define WV_K(T, NCO, S2) hipLaunchKernelGGL(HIP_KERNEL_NAME(wgrad3x3_wave32_kernel<T, NCO, S2>), grid, dim3(WV_THREADS), 0, st, p, nstrips)
#define WV_L(T)                                \
    do {                                       \
        if (p.isy == 2) WV_K(T, 2, true);      \
        else if (p.gC == 64) WV_K(T, 2, false); \
        else WV_K(T, 1, false);                \
    } while (0)
    FALNET_DISPATCH_16(p.dtype, WV_L);
#undef WV_L
#undef WV_K
    FALNET_RETURN_LAUNCH();
}
