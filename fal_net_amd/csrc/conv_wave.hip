// Wave-streaming 3x3 / stride-1 convolution for layers with 32 input AND at most 32 output channels (16-bit operands), gfx950
// (falnet_conv2d variant 27, round 6): conv0_1's two residual convolutions (models/FAL_netB.py:38-47,100 of the reference) forward and as data
// gradients -- 32 -> 32 at full resolution, four launches per step.
//
// Why a kernel of its own.  At 32 -> 32 the layer moves 134 MB (B = 8, 256 x 512: 67 MB in, 67 MB out) for 19.3 GFLOP -- 144 FLOP / B, HBM-bound
// (the chip's ridge is ~310) -- and its whole packed weight is 18 KB: 72 registers of ONE wave hold every tap's fragment for the 16x16x32 MFMA.
// The persistent tile kernels (weight-stationary conv3x3_ws_kernel: 44-61 us per launch, 390-440 TFLOP/s; the LDS-DMA kernel computes 64 output
// channels per workgroup, half of them padding here) share a tile between eight waves behind barriers.  Here, as in wgrad_wave.hip, nothing is shared:
//   * a wave owns a contiguous range of (sample, 32-pixel column strip, row) units and streams its strip top to bottom: ONE input row per step
//     arrives by LDS-DMA (global_load_lds_dwordx4: three 16-pixel pieces of whole 64-B pixel lines, the +-1 column halo included) into a private
//     ring of three row slots, two rows ahead, ordered by a counted s_waitcnt vmcnt only -- no s_barrier in the loop;
//   * the row's six pixel fragments (column offsets 0, 1, 2 x two 16-position halves: ds_read_b128, K = the 32 channels) join a rolling REGISTER
//     window of three rows; once row y + 1 is in, output row y is 36 MFMAs (9 taps x 2 channel halves x 2 position halves) against the resident
//     weight fragments, then conv_epilogue.h's direct epilogue (bias / residual / ELU / ReLU / activation gradient, lane = pixel, 16-B stores);
//   * out-of-image rows / columns come from a page of zeros; every step issues the same three pieces, the vmcnt arithmetic is a constant.
// MFMA roles as in conv3x3_dma16_kernel (A = weights with the rows permuted by m16_row_channel, B = pixels; Acc16::to32 lands the 32 x 32 tile in
// the epilogue's layout); LDS image of a row [pixel][32 channels], the 16-B segments exchanged in pairs when bit 2 of the pixel index is set (on the
// DMA's SOURCE address): conflict-free ds_read_b128 for the (position lane & 15, K block lane >> 4) pattern at every column offset.
#include "conv_epilogue.h"

typedef __attribute__((address_space(3))) void* cw_lptr_t;

#define CW_THREADS 512
#define CW_XROW 3072                 // input row: 34 px used, three 16-pixel DMA pieces
#define CW_D 2
#define CW_NS (CW_D + 1)
#define CW_RING (CW_NS * CW_XROW)    // 9 KiB per wave
#define CW_PIECES 3

__device__ uint4 g_cw_zero[8] = {};

__device__ __forceinline__ void cw_glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}

struct CwItem { int u, len, n, s, b, x0, y0; };  // units [u, u + len): output rows [y0, y0 + n) of strip x0 of sample b; s = step (input row y0 - 1 + s)

template <typename T>
__global__ __launch_bounds__(CW_THREADS) void conv3x3_wave32_kernel(const falnet_conv_t p, int nstrips, int flip, int RB, int nyb) {
    static_assert(sizeof(T) == 2, "16-bit operands");
    __shared__ __attribute__((aligned(1024))) char lds[8 * CW_RING];
    __shared__ __attribute__((aligned(16))) float lds_bias[32];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned ring = (unsigned)(unsigned long)(cw_lptr_t)lds + wave * CW_RING;
    const char* const ring_p = lds + wave * CW_RING;
    const int H = p.OH, W = p.OW, IH = p.IH, IW = p.IW;
    stage_bias_lds(p, 0, 32, lds_bias);
    __syncthreads();  // (the only barrier: the bias block)
    // unit = (sample, block of RB rows, strip, row of the block), row fastest, THEN the strip: neighbouring waves stream neighbouring strips of the
    // same rows -- a workgroup pulls 8 x 2 KiB of consecutive bytes per image row instead of eight rows 64 KiB apart (DRAM page locality)
    const int R = p.B * nyb * nstrips * RB;
    const int nparts = (int)gridDim.x * 8, gp = (int)blockIdx.x * 8 + wave;
    const int u0 = (int)((int64_t)R * gp / nparts), u1 = (int)((int64_t)R * (gp + 1) / nparts);

    // ---- resident weights: A operand of tap t, channel half ct: lane (lp, lg) holds row m16_row_channel(lp) of the half, K block lg ----
    const int lp = lane & 15, lg = lane >> 4;
    s16x8_t wf[9][2];
    {
        const T* wptr = reinterpret_cast<const T*>(p.weight);
        const int wrow = m16_row_channel(lp);
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                const int co = 16 * ct + wrow;
                wf[t][ct] = *reinterpret_cast<const s16x8_t*>(wptr + ((int64_t)co * 9 + (flip ? 8 - t : t)) * 32 + 8 * lg);
            }
    }

    // ---- per-lane DMA geometry: lane = (pixel pl of a 16-pixel piece, 16-B position seg); LDS pixel 16 k + pl has bit 2 of pl ----
    const int pl = lane >> 2, seg = lane & 3;
    const int gseg = seg ^ (((pl >> 2) & 1) << 1);
    const char* const zero_page = reinterpret_cast<const char*>(g_cw_zero);
    const falnet_src_t& S = p.src[0];
    const int64_t l_sy = S.sy, l_sx = S.sx, l_sb = S.sb;
    const T* const x_base = reinterpret_cast<const T*>(S.ptr) + 8 * gseg;
    const unsigned x_rowb = (unsigned)(l_sy * (int)sizeof(T));

    auto load_item = [&](CwItem& c, int u) {
        c.u = u;
        const int bs = u / RB, r = u - bs * RB;
        const int bb = bs / nstrips;
        c.b = bb / nyb;
        c.x0 = (bs - bb * nstrips) * 32;
        c.y0 = (bb - c.b * nyb) * RB + r;
        c.len = min(RB - r, u1 - u);
        c.n = max(0, min(c.len, H - c.y0));  // (the last block of a map whose height is no multiple of RB has rows below the image)
        c.s = 0;
    };
    auto steps_of = [](const CwItem& c) { return c.n > 0 ? c.n + 2 : 0; };
    int nst = 0;
    for (int u = u0; u < u1;) {
        CwItem t;
        load_item(t, u);
        nst += steps_of(t);
        u += t.len;
    }
    auto first_item = [&](CwItem& c, int u) {  // the first item from u on that has rows (nst > 0 guarantees one)
        load_item(c, u);
        while (c.n == 0) load_item(c, c.u + c.len);
    };
    const char* xptr[3];
    unsigned xinc[3];
    auto item_pointers = [&](const CwItem& c) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int lpx = 16 * k + pl, xa = c.x0 - 1 + lpx;
            const bool ok = lpx < 34 && xa >= 0 && xa < IW;
            xptr[k] = ok ? reinterpret_cast<const char*>(x_base + (int64_t)c.b * l_sb + (int64_t)(c.y0 - 1) * l_sy + (int64_t)xa * l_sx) : zero_page;
            xinc[k] = ok ? x_rowb : 0u;
        }
    };
    auto issue = [&](const CwItem& c, int slot, bool real) {
        const unsigned base = ring + slot * CW_XROW;
        const int i = c.y0 - 1 + c.s;
        const bool xv = real && i >= 0 && i < IH;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            cw_glds16(xv ? xptr[k] : zero_page, base + k * 1024);
            xptr[k] += xinc[k];
        }
    };

    // ---- fragment read offsets: position 16 pt + lp at column offset dx -> LDS pixel col = dx + 16 pt + lp, segment lg (pair-swapped on bit 2) ----
    int offX[3][2];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) {
            const int col = dx + 16 * pt + lp;
            offX[dx][pt] = col * 64 + ((lg ^ (((col >> 2) & 1) << 1)) << 4);
        }

    s16x8_t xw[3][3][2];  // [window row: y - 1, y, y + 1][dx][pt]
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
#pragma unroll
            for (int pt = 0; pt < 2; ++pt)
#pragma unroll
                for (int j = 0; j < 8; ++j) xw[r][dx][pt][j] = 0;

    CwItem ci_, cc_;
    ci_ = CwItem{0, 0, 0, 0, 0, 0, 0};
    if (nst > 0) {
        first_item(ci_, u0);
        item_pointers(ci_);
    }
    cc_ = ci_;
    int issued = 0;
    auto issue_next = [&](int slot) {
        const bool real = issued < nst;
        issue(ci_, slot, real);
        if (real && ++issued < nst && ++ci_.s == ci_.n + 2) {
            first_item(ci_, ci_.u + ci_.len);
            item_pointers(ci_);
        }
    };
#pragma unroll
    for (int d = 0; d < CW_D; ++d) issue_next(d);
    int slot = 0, islot = CW_D;
    const int r32 = lane & 31, h = lane >> 5;
    const int cstride = p.out_cstride;
    for (int g = 0; g < nst; ++g) {
        issue_next(islot);
        islot = islot + 1 == CW_NS ? 0 : islot + 1;
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CW_D * CW_PIECES) : "memory");
        const char* sb = ring_p + slot * CW_XROW;
        slot = slot + 1 == CW_NS ? 0 : slot + 1;
        const int s = cc_.s;
        // roll the window and read the new row's fragments
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
#pragma unroll
            for (int pt = 0; pt < 2; ++pt) {
                xw[0][dx][pt] = xw[1][dx][pt];
                xw[1][dx][pt] = xw[2][dx][pt];
                xw[2][dx][pt] = *reinterpret_cast<const s16x8_t*>(sb + offX[dx][pt]);
            }
        if (s >= 2) {  // output row y = y0 + s - 2: rows y - 1, y, y + 1 are in the window (wave-uniform)
            Acc16 acc;
            acc.zero();
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                        for (int pt = 0; pt < 2; ++pt) acc.t[ct][pt] = H16<T>::mma16(wf[dy * 3 + dx][ct], xw[dy][dx][pt], acc.t[ct][pt]);
            const int y = cc_.y0 + s - 2, x = cc_.x0 + r32;
            const int64_t o = x < W ? (((int64_t)cc_.b * H + y) * W + x) * cstride : (int64_t)-1;
            auto pixoff = [&](int) -> int64_t { return o; };
            float bias[1][16];
            load_bias16_lds(lds_bias, 0, h, bias);
            f32x16 v[1][1];
            acc.to32(v[0][0]);
            epilogue_direct<T, 1, 1, decltype(pixoff), NoPool, -1, false, true>(p, v, bias, 0, lane, pixoff);
        }
        if (g + 1 < nst && ++cc_.s == cc_.n + 2) first_item(cc_, cc_.u + cc_.len);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// conv3x3_wave64p_kernel (falnet_conv2d variant 29, round 6): the same wave-streaming form for a 64-channel source and at most FOUR output channels
// written as planar f32 -- the data gradient of VGG19's first convolution with respect to the synthesised view (loss_functions.py:21: 64 -> 3, the
// last launch of the perceptual term's adjoint; 3.6 GFLOP over a 134 MB gradient tensor: HBM-bound, 60 us on the weight-stationary kernel whose
// 32-channel output block is 29 / 32 padding).  A wave owns 16-pixel strips (18 x 128-B pixel lines per row = three 8-pixel pieces), keeps the nine taps'
// weight fragments for ONE 16-row output tile (rows 0 .. 3 real) and both 32-channel halves in 72 registers, the three-row pixel window in 72 more, and
// issues 18 MFMAs per output row; lanes 0-15 end up with their pixel's (up to) four channels and store 64-B runs per channel plane.
// LDS image of a row: [pixel][64 channels], the eight 16-B segments of pixel p XOR-ed with p & 7 (on the DMA's source address): conflict-free
// ds_read_b128 for the (position lane & 15, K block lane >> 4) pattern at column offsets 0..2 and both halves (exhaustive check).
#define CP_XROW 3072                 // input row: 18 px used of three 8-pixel pieces
#define CP_RING (CW_NS * CP_XROW)

template <typename T>
__global__ __launch_bounds__(CW_THREADS) void conv3x3_wave64p_kernel(const falnet_conv_t p, int nstrips, int flip, int RB, int nyb) {
    static_assert(sizeof(T) == 2, "16-bit operands");
    __shared__ __attribute__((aligned(1024))) char lds[8 * CP_RING];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned ring = (unsigned)(unsigned long)(cw_lptr_t)lds + wave * CP_RING;
    const char* const ring_p = lds + wave * CP_RING;
    const int H = p.OH, W = p.OW, IH = p.IH, IW = p.IW;
    const int R = p.B * nyb * nstrips * RB;
    const int nparts = (int)gridDim.x * 8, gp = (int)blockIdx.x * 8 + wave;
    const int u0 = (int)((int64_t)R * gp / nparts), u1 = (int)((int64_t)R * (gp + 1) / nparts);

    const int lp = lane & 15, lg = lane >> 4;
    s16x8_t wf[9][2];  // [tap][32-channel half]: A operand rows = output channels 0 .. 15 (no row permutation: the results are read per lane below)
    {
        const T* wptr = reinterpret_cast<const T*>(p.weight);
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int kc = 0; kc < 2; ++kc)
                wf[t][kc] = *reinterpret_cast<const s16x8_t*>(wptr + ((int64_t)lp * 9 + (flip ? 8 - t : t)) * 64 + 32 * kc + 8 * lg);
    }

    // ---- per-lane DMA geometry: lane = (pixel pl of an 8-pixel piece, 16-B position seg of its 128-B line); LDS pixel 8 k + pl has pl in its low bits ----
    const int pl = lane >> 3, seg = lane & 7;
    const int gseg = seg ^ pl;
    const char* const zero_page = reinterpret_cast<const char*>(g_cw_zero);
    const falnet_src_t& S = p.src[0];
    const int64_t l_sy = S.sy, l_sx = S.sx, l_sb = S.sb;
    const T* const x_base = reinterpret_cast<const T*>(S.ptr) + 8 * gseg;
    const unsigned x_rowb = (unsigned)(l_sy * (int)sizeof(T));

    auto load_item = [&](CwItem& c, int u) {
        c.u = u;
        const int bs = u / RB, r = u - bs * RB;
        const int bb = bs / nstrips;
        c.b = bb / nyb;
        c.x0 = (bs - bb * nstrips) * 16;
        c.y0 = (bb - c.b * nyb) * RB + r;
        c.len = min(RB - r, u1 - u);
        c.n = max(0, min(c.len, H - c.y0));
        c.s = 0;
    };
    int nst = 0;
    for (int u = u0; u < u1;) {
        CwItem t;
        load_item(t, u);
        nst += t.n > 0 ? t.n + 2 : 0;
        u += t.len;
    }
    auto first_item = [&](CwItem& c, int u) {
        load_item(c, u);
        while (c.n == 0) load_item(c, c.u + c.len);
    };
    const char* xptr[3];
    unsigned xinc[3];
    auto item_pointers = [&](const CwItem& c) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int lpx = 8 * k + pl, xa = c.x0 - 1 + lpx;
            const bool ok = lpx < 18 && xa >= 0 && xa < IW;
            xptr[k] = ok ? reinterpret_cast<const char*>(x_base + (int64_t)c.b * l_sb + (int64_t)(c.y0 - 1) * l_sy + (int64_t)xa * l_sx) : zero_page;
            xinc[k] = ok ? x_rowb : 0u;
        }
    };
    auto issue = [&](const CwItem& c, int slot, bool real) {
        const unsigned base = ring + slot * CP_XROW;
        const int i = c.y0 - 1 + c.s;
        const bool xv = real && i >= 0 && i < IH;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            cw_glds16(xv ? xptr[k] : zero_page, base + k * 1024);
            xptr[k] += xinc[k];
        }
    };
    int offX[3][2];  // [dx][32-channel half]
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {
            const int col = dx + lp;
            offX[dx][kc] = col * 128 + (((4 * kc + lg) ^ (col & 7)) << 4);
        }
    s16x8_t xw[3][3][2];  // [window row][dx][half]
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
#pragma unroll
            for (int kc = 0; kc < 2; ++kc)
#pragma unroll
                for (int j = 0; j < 8; ++j) xw[r][dx][kc][j] = 0;

    CwItem ci_, cc_;
    ci_ = CwItem{0, 0, 0, 0, 0, 0, 0};
    if (nst > 0) {
        first_item(ci_, u0);
        item_pointers(ci_);
    }
    cc_ = ci_;
    int issued = 0;
    auto issue_next = [&](int slot) {
        const bool real = issued < nst;
        issue(ci_, slot, real);
        if (real && ++issued < nst && ++ci_.s == ci_.n + 2) {
            first_item(ci_, ci_.u + ci_.len);
            item_pointers(ci_);
        }
    };
#pragma unroll
    for (int d = 0; d < CW_D; ++d) issue_next(d);
    int slot = 0, islot = CW_D;
    float* const out = reinterpret_cast<float*>(p.out);
    const int64_t plane = (int64_t)H * W;
    float bias4[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) bias4[f] = (p.bias && f < p.Cout) ? p.bias[f] : 0.f;
    for (int g = 0; g < nst; ++g) {
        issue_next(islot);
        islot = islot + 1 == CW_NS ? 0 : islot + 1;
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CW_D * CW_PIECES) : "memory");
        const char* sb = ring_p + slot * CP_XROW;
        slot = slot + 1 == CW_NS ? 0 : slot + 1;
        const int s = cc_.s;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
#pragma unroll
            for (int kc = 0; kc < 2; ++kc) {
                xw[0][dx][kc] = xw[1][dx][kc];
                xw[1][dx][kc] = xw[2][dx][kc];
                xw[2][dx][kc] = *reinterpret_cast<const s16x8_t*>(sb + offX[dx][kc]);
            }
        if (s >= 2) {
            f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                    for (int kc = 0; kc < 2; ++kc) acc = H16<T>::mma16(wf[dy * 3 + dx][kc], xw[dy][dx][kc], acc);
            // D[4 lg + f][lp]: lanes 0 .. 15 (lg = 0) hold channels 0 .. 3 of position lp
            const int y = cc_.y0 + s - 2, x = cc_.x0 + lp;
            if (lg == 0 && x < W) {
                float* o = out + (int64_t)cc_.b * p.Cout * plane + (int64_t)y * W + x;
#pragma unroll
                for (int f = 0; f < 4; ++f)
                    if (f < p.Cout) o[f * plane] = apply_act(acc[f] + bias4[f], p.act);
            }
        }
        if (g + 1 < nst && ++cc_.s == cc_.n + 2) first_item(cc_, cc_.u + cc_.len);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

bool falnet_conv_wave64p_applicable(const falnet_conv_t& p) {
    if (p.dtype != FALNET_BF16 && p.dtype != FALNET_F16) return false;
    if (p.nsrc != 1 || p.cin_total != 64 || p.src[0].C != 64 || p.src[0].H != p.IH || p.src[0].W != p.IW) return false;
    if (p.w_rows < 16 || p.w_taps != 9 || p.Cout < 1 || p.Cout > 4 || p.out_layout != FALNET_OUT_PLANAR_F32 || p.pool_out || p.ksplit > 1) return false;
    if (p.addend || p.actout || p.OW < 16 || !p.out) return false;
    if ((int64_t)p.B * ((p.OW + 15) / 16) * (p.OH + 32) >= (1ll << 30) || p.src[0].sy * 2 >= (1ll << 31)) return false;
    return true;
}

int falnet_conv_wave64p_launch(const falnet_conv_t& p, int flip, hipStream_t st) {
    const int nstrips = (p.OW + 15) / 16;
    const int RB = p.OH >= 32 ? 32 : p.OH;
    const int nyb = (p.OH + RB - 1) / RB;
    const int64_t units = (int64_t)p.B * nyb * nstrips * RB;
    int wgs = (int)((units + 8 * RB - 1) / (8 * RB));
    if (wgs > 256) wgs = 256;
    if (wgs < 1) wgs = 1;
#define CP_L(T) hipLaunchKernelGGL(HIP_KERNEL_NAME(conv3x3_wave64p_kernel<T>), dim3((unsigned)wgs), dim3(CW_THREADS), 0, st, p, nstrips, flip, RB, nyb)
    FALNET_DISPATCH_16(p.dtype, CP_L);
#undef CP_L
    FALNET_RETURN_LAUNCH();
}

// dense 3x3 / stride 1 / pad 1 over ONE 32-channel NHWC source at the launch size, packed weight of 32 rows x 9 taps x 32 channels, NHWC output,
// no fused pool / split-K (the caller has established dense3x3 and the tap order: `flip`)
bool falnet_conv_wave32_applicable(const falnet_conv_t& p) {
    if (p.dtype != FALNET_BF16 && p.dtype != FALNET_F16) return false;
    if (p.nsrc != 1 || p.cin_total != 32 || p.src[0].C != 32 || p.src[0].H != p.IH || p.src[0].W != p.IW) return false;
    if (p.w_rows != 32 || p.w_taps != 9 || p.Cout > 32 || p.out_layout != FALNET_OUT_NHWC || p.pool_out || p.ksplit > 1) return false;
    if (p.OW < 32 || !p.out) return false;
    if ((int64_t)p.B * ((p.OW + 31) / 32) * (p.OH + 16) >= (1ll << 30) || p.src[0].sy * 2 >= (1ll << 31)) return false;
    return true;
}

int falnet_conv_wave32_launch(const falnet_conv_t& p, int flip, hipStream_t st) {
    const int nstrips = (p.OW + 31) / 32;
    const int RB = p.OH >= 16 ? 16 : p.OH;  // rows per block = one wave's range when the map is large enough
    const int nyb = (p.OH + RB - 1) / RB;
    const int64_t units = (int64_t)p.B * nyb * nstrips * RB;
    int wgs = (int)((units + 8 * RB - 1) / (8 * RB));  // one block-strip column per wave (two of a range's steps are halo rows)
    if (wgs > 256) wgs = 256;
    if (wgs < 1) wgs = 1;
#define CW_L(T) hipLaunchKernelGGL(HIP_KERNEL_NAME(conv3x3_wave32_kernel<T>), dim3((unsigned)wgs), dim3(CW_THREADS), 0, st, p, nstrips, flip, RB, nyb)
    FALNET_DISPATCH_16(p.dtype, CW_L);
#undef CW_L
    FALNET_RETURN_LAUNCH();
}
