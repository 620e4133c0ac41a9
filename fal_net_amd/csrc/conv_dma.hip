// Dense 3x3 / stride-1 convolution (forward, stride-1 data gradient; fused nearest-2x upsample and two-source concat), 16-bit
// operands, 64 output channels per workgroup -- LDS-DMA, double-buffered, persistent.
//
// Replaces, where the plan's autotuner finds it faster, the single-stage halo-patch kernel (conv3x3_patch_kernel<.., 64, 64, 9, ..>)
// of the cuDNN call sites models/FAL_netB.py:38-58,145-173 and loss_functions.py:21-29.  That kernel stages a K chunk
// (32 input channels: the (TH+2) x 34 halo patch and the nine 64 x 32 weight tiles, 59-76 KB) through REGISTERS with one
// exposed memory round trip per chunk, and relies on a second resident workgroup to cover it: the matrix pipe idles half the
// time (585 TFLOP/s).  Here
//   * a chunk arrives by LDS-DMA (global_load_lds_dwordx4, 1-KiB pieces, no VGPR staging, no ds_write) into one of TWO LDS
//     buffers: chunk i+1 is requested right after the barrier that publishes chunk i and lands behind chunk i's 72 MFMAs per
//     wave; ONE barrier per chunk;
//   * the workgroup is persistent (one per CU: 150 KB of LDS) and walks (tile, chunk) pairs as one flat sequence, so the
//     first chunk of the next tile is in flight during the last chunk -- and the epilogue -- of the current one: no exposed
//     prologue per 16x32 block;
//   * 16 x 32 output positions x 64 channels per workgroup, 8 waves (two 32-position rows x two 32-channel tiles each): the
//     nine weight tiles are fetched once per 512 positions.
// LDS image: 64-B rows ([pixel][32 ch] / [tap][cout][32 ch]) exactly as the DMA writes them; the four 16-B segments of row r
// are XOR-swizzled with (r >> 2) & 3 on the SOURCE address (the DMA destination is lane-linear), which makes every
// ds_read_b128 fragment read conflict-free for any row base (its 16-lane groups cover all 16 residues of r mod 16).
// MFMA operands are exchanged (A = weights, B = pixels): the pixel sits on the lane, epilogue_direct (conv_epilogue.h) writes
// 16-B stores with bias / residual / activation / activation-gradient / 2x2 pooling fused.
#include <stdlib.h>
#include "conv_epilogue.h"

#ifndef FALNET_DMA_EPI_AHEAD
#define FALNET_DMA_EPI_AHEAD -1  // conv_epilogue.h: epilogue_direct's operand prefetch depth (-1: conditional loads at the point of use)
#endif

// 32 KiB of zeros: the 'pixel' / 'weight row' every out-of-image or out-of-range 16-B piece is fetched from.  Invalid lanes carry
// the OFFSET of this page relative to their tensor, so a piece's address is always base + offset (+ channel offset < 16 K
// elements): no compare / select per piece and chunk.
#define CD_ZERO_BYTES 32768
__device__ uint4 g_cd_zero[CD_ZERO_BYTES / 16] = {};

typedef __attribute__((address_space(3))) void* cd_lptr_t;

// one 1-KiB LDS-DMA piece (inline asm: outside hipcc's LDS-DMA alias tracking, which would drain the prefetch with
// s_waitcnt vmcnt(0) in front of every fragment read; see wgrad_rows.hip)
__device__ __forceinline__ void cd_glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}

#define CD_PW 34  // patch columns: 32 + the +-1 halo

// <T, 16, 8>: the 16 x 32 tile described above.  <T, 4, 4> (variant 17): 4 x 32 positions, four waves of ONE row each, 2 x 49 KB of LDS -- the
// 16 x 32-pixel maps of level 4 are a single 16-row tile per sample (32 workgroups at B = 8), with four-row tiles 128: 29 -> 17 us on the
// 256-channel layers there.  (Measured on that form and dropped: a third chunk buffer with two chunks in flight, 18.6 vs 17.4 us -- a one-row
// wave is held by the ISSUE of its 12-13 DMA pieces per chunk, ~200 cycles each beside 36 MFMAs, not by their latency.)
template <typename T, int TH, int NWAVES>
__global__ __launch_bounds__(NWAVES * 64) void conv3x3_dma_kernel(const falnet_conv_t p, int tiles_x, int tiles_y, int flip, int ntiles) {
    constexpr int BN = 64, MT = TH / NWAVES, NT = BN / 32;
    static_assert((MT == 1 || MT == 2) && sizeof(T) == 2, "one or two 32-position rows per wave, 16-bit operands");
    constexpr int KCV = 32;                                   // input channels per chunk (64 B per pixel / weight row)
    constexpr int NPIX = (TH + 2) * CD_PW;
    constexpr int A_PIECES = (NPIX + 15) / 16, B_PIECES = 9 * BN / 16, NPIECES = A_PIECES + B_PIECES;
    constexpr int A_BYTES = A_PIECES * 1024, BUF = NPIECES * 1024;
    __shared__ __attribute__((aligned(1024))) char lds[2 * BUF];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds_base = (unsigned)(unsigned long)(cd_lptr_t)lds;
    const int r = lane & 31, h = lane >> 5;
    const int n0 = blockIdx.y * BN;
    const char* const zero_page = reinterpret_cast<const char*>(g_cd_zero);

    // ---- K walk: chunks over (source, channel offset) ----
    const int nsrc = p.nsrc, IH = p.IH, IW = p.IW;
    const int C0 = p.src[0].C, C1 = nsrc > 1 ? p.src[1].C : 0;
    const int nchunks = (C0 + C1) / KCV;
    int my_tiles = 0;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) ++my_tiles;
    const int total = my_tiles * nchunks;

    // ---- DMA geometry of this lane: (row of a 16-row piece, 16-B segment position); the segment it FETCHES is swizzled ----
    // Patch pieces a = wave + NWAVES k and weight pieces w = wave + NWAVES k (k < KP) belong to this wave.  Everything
    // lane-dependent is computed ONCE per tile (patch: element offset per source, -1 = zero fill) or once per kernel (weights):
    // a chunk then costs one 64-bit add and one select per piece (PMC of the first version: 7 VALU + 5 SALU per MFMA, most of
    // them address arithmetic of this loop).
    constexpr int KP = (A_PIECES + NWAVES - 1) / NWAVES, KW = (B_PIECES + NWAVES - 1) / NWAVES;
    const int l4 = lane >> 2, segpos = lane & 3;
    const T* const wptr = reinterpret_cast<const T*>(p.weight);
    const T* const zero_t = reinterpret_cast<const T*>(zero_page);
    int64_t w_off[KW];  // element offsets from the packed weight (rows beyond w_rows: the zero page)
#pragma unroll
    for (int k = 0; k < KW; ++k) {
        const int wid = wave + NWAVES * k;
        const int tap = wid >> 2, co = n0 + ((wid & 3) << 4) + l4;  // four 16-row pieces per tap tile
        const int gseg = segpos ^ ((lane >> 4) & 3);                 // (row >> 2) & 3 = (l4 >> 2) & 3: pieces start at multiples of 16
        w_off[k] = (wid < B_PIECES && co < p.w_rows) ? (int64_t)(co * p.w_taps + (flip ? 8 - tap : tap)) * p.cin_total + gseg * 8 : (int64_t)(zero_t - wptr);
    }
    int64_t a_off[KP];  // of the issue cursor's (tile, source), from the sample's base: recomputed when either changes (twice per tile at most)
    const T* sptr[2] = {reinterpret_cast<const T*>(p.src[0].ptr), reinterpret_cast<const T*>(nsrc > 1 ? p.src[1].ptr : p.src[0].ptr)};
    int64_t sbat[2] = {0, 0};  // sample offset of the issue cursor's tile, per source
    auto tile_coords = [&](int tile, int& b, int& ty0, int& tx0) {
        const int tix = tile % tiles_x;
        const int q = tile / tiles_x;
        ty0 = (q % tiles_y) * TH;
        tx0 = tix * 32;
        b = q / tiles_y;
    };
    auto tile_offsets = [&](int tile, int s2) {  // patch offsets of the issue cursor's (tile, source)
        int b, ty0, tx0;
        tile_coords(tile, b, ty0, tx0);
        const falnet_src_t& S = s2 == 0 ? p.src[0] : p.src[1];
        sbat[s2] = (int64_t)b * S.sb;
        const int hs = S.H != IH ? 1 : 0, ws = S.W != IW ? 1 : 0;  // exact 2x nearest upsampling (dispatcher checks)
#pragma unroll
        for (int k = 0; k < KP; ++k) {
            const int pix = 16 * (wave + NWAVES * k) + l4;
            const int pr = pix / CD_PW, pc = pix - pr * CD_PW;
            const int vy = ty0 - 1 + pr, vx = tx0 - 1 + pc;
            const bool ok = pix < NPIX && vy >= 0 && vy < IH && vx >= 0 && vx < IW;
            a_off[k] = ok ? (int64_t)((vy >> hs) * (int)S.sy + (vx >> ws) * (int)S.sx + (segpos ^ ((pix >> 2) & 3)) * 8)
                          : (int64_t)(zero_t - (reinterpret_cast<const T*>(S.ptr) + sbat[s2]));
        }
    };
    struct Cur { int tile, c, s, c0, kofs; };      // issue cursor: tile, chunk, source, channel offset, weight-row offset
    auto advance = [&](Cur& q) {
        if (++q.c == nchunks) {
            q.c = 0; q.s = 0; q.c0 = 0; q.kofs = 0;
            q.tile += gridDim.x;
            if (q.tile < ntiles) tile_offsets(q.tile, 0);
            return;
        }
        q.c0 += KCV;
        q.kofs += KCV;
        if (q.s == 0 && q.c0 >= C0) {
            q.s = 1;
            q.c0 = 0;
            tile_offsets(q.tile, 1);
        }
    };
    // one piece of the issue cursor's chunk (i = 0 .. KP + KW - 1: patch pieces, then weight pieces); the pieces of chunk it + 1 are
    // issued one per MFMA step inside chunk it's loop -- in-kernel stamps of the all-at-once form: 2 000-2 500 of a chunk's 9 000 cycles
    // were spent in the issue of ~10 gathering DMA instructions with the matrix pipe idle (both waves of a SIMD issued at the same time)
    auto issue_piece = [&](const Cur& q, int buf, int i) {
        const unsigned dst0 = lds_base + buf * BUF;
        if (i < KP) {
            const int id = wave + NWAVES * i;
            const T* sbase = (q.s == 0 ? sptr[0] + sbat[0] : sptr[1] + sbat[1]) + q.c0;
            if (id < A_PIECES) cd_glds16(sbase + a_off[i], dst0 + id * 1024);
        } else {
            const int wid = wave + NWAVES * (i - KP);
            if (wid < B_PIECES) cd_glds16(wptr + q.kofs + w_off[i - KP], dst0 + A_BYTES + wid * 1024);
        }
    };
    auto issue = [&](const Cur& q, int buf) {
        const T* sbase = (q.s == 0 ? sptr[0] + sbat[0] : sptr[1] + sbat[1]) + q.c0;
        const T* wbase = wptr + q.kofs;
        const unsigned dst0 = lds_base + buf * BUF;
#pragma unroll
        for (int k = 0; k < KP; ++k) {
            const int id = wave + NWAVES * k;
            if (id < A_PIECES) {  // wave-uniform
                cd_glds16(sbase + a_off[k], dst0 + id * 1024);
            }
        }
#pragma unroll
        for (int k = 0; k < KW; ++k) {
            const int wid = wave + NWAVES * k;
            if (wid < B_PIECES)
                cd_glds16(wbase + w_off[k], dst0 + A_BYTES + wid * 1024);
        }
    };

    // ---- fragment read addresses (bytes inside a buffer): pixel rows  p = (2 wave + mt + dy) * 34 + dx + r ----
    int a_addr[MT + 2][3];  // k-step 0; k-step 1 is the same address with bit 5 flipped (segment (2 ks + h) ^ q)
#pragma unroll
    for (int rs = 0; rs < MT + 2; ++rs)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int pp = (wave * MT + rs) * CD_PW + dx + r;
            a_addr[rs][dx] = pp * 64 + ((h ^ ((pp >> 2) & 3)) << 4);
        }
    int b_lane[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) b_lane[ks] = A_BYTES + r * 64 + (((2 * ks + h) ^ ((r >> 2) & 3)) << 4);

    f32x16 acc[NT][MT][1];  // [nt][mt]: the epilogue runs per 32-channel slice (one slice's bias / operands in registers at a time)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[nt][mt][0][j] = 0.f;
    Cur qi = {(int)blockIdx.x, 0, 0, 0, 0};
    if (total > 0) {
        tile_offsets(qi.tile, 0);
        issue(qi, 0);
        advance(qi);
    }
    int ctile = blockIdx.x, cc = 0;  // compute cursor
#ifdef FALNET_CD_STAMPS
    // profiling build (tools/cd_stamps.py): s_memtime at the phase boundaries of the first 48 chunks, every wave of workgroup (0, 0) -> p.splitk_ws
    unsigned long long* stamp_out = reinterpret_cast<unsigned long long*>(p.splitk_ws);
    int stamp_i = 0;
#define CD_STAMP()                                                                                   \
    do {                                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                           \
        unsigned long long t_;                                                                       \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                  \
        __builtin_amdgcn_sched_barrier(0);                                                           \
        if (stamp_out && blockIdx.x == 0 && blockIdx.y == 0 && lane == 0 && stamp_i < 256) stamp_out[wave * 256 + stamp_i] = t_; \
        ++stamp_i;                                                                                   \
    } while (0)
#else
#define CD_STAMP() do {} while (0)
#endif
    for (int it = 0; it < total; ++it) {
        CD_STAMP();  // 0: loop top
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // own pieces of chunk `it` landed; own reads of chunk it-1 done
        CD_STAMP();  // 1: own DMA landed
        __builtin_amdgcn_s_barrier();                                // chunk `it` complete for every wave; buffer (it+1)&1 is free
        CD_STAMP();  // 2: barrier passed
        const bool more = it + 1 < total;
        CD_STAMP();  // 3: (nothing issued up front any more)
        // fragment bases of this chunk's buffer: ONE add per lane address (hipcc otherwise keeps every (address + tap offset)
        // of both buffers in registers -- ~70 VGPRs of loop-invariant sums -- and spills)
        int bo = (it & 1) * BUF;
        asm volatile("" : "+s"(bo));
        const char* const Bf = lds;
        int aa[MT + 2][3], bb[2];
#pragma unroll
        for (int rs = 0; rs < MT + 2; ++rs)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) aa[rs][dx] = a_addr[rs][dx] + bo;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) bb[ks] = b_lane[ks] + bo;
        // 18 steps (tap, 16-channel half): the fragment reads of step i + 1 are issued in front of the four MFMAs of step i (two
        // register sets); sched_group_barrier pins that order -- left alone, hipcc hoists a dozen steps of reads and spills
        s16x8_t fa[2][MT], fb[2][NT];
        auto load_step = [&](int st, int set) {
            const int t = st >> 1, ks = st & 1;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) fa[set][mt] = *reinterpret_cast<const s16x8_t*>(Bf + (aa[mt + t / 3][t % 3] ^ (ks << 5)));
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) fb[set][nt] = *reinterpret_cast<const s16x8_t*>(Bf + bb[ks] + (t * BN + nt * 32) * 64);
        };
        load_step(0, 0);
#pragma unroll
        for (int st = 0; st < 18; ++st) {
            if (st + 1 < 18) load_step(st + 1, (st + 1) & 1);
            if (st < KP + KW && more) issue_piece(qi, (it + 1) & 1, st);  // (wave-uniform)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[nt][mt][0] = H16<T>::mma(fb[st & 1][nt], fa[st & 1][mt], acc[nt][mt][0]);
            __builtin_amdgcn_sched_group_barrier(0x100, MT + NT, 0);  // DS reads of the next step
            __builtin_amdgcn_sched_group_barrier(0x008, MT * NT, 0);  // this step's MFMAs
        }
        if (more) advance(qi);
        CD_STAMP();  // 4: MFMAs issued
        if (++cc == nchunks) {  // tile finished: epilogue straight from the accumulators, then the next tile starts from zero
            cc = 0;
            int b, ty0, tx0;
            tile_coords(ctile, b, ty0, tx0);
            ctile += gridDim.x;
            const int cstride = p.out_cstride;
            const int x = tx0 + r;
            const bool planar_out = p.out_layout == FALNET_OUT_PLANAR_F32;
            auto pixoff = [&](int mt) -> int64_t {
                const int y = ty0 + wave * MT + mt;
                if (!(y < p.OH && x < p.OW)) return (int64_t)-1;
                return planar_out ? ((int64_t)b * p.Cout * p.OH + y) * p.OW + x : (((int64_t)b * p.OH + y) * p.OW + x) * cstride;
            };
            auto pooloff = [&](int mt) -> int64_t {
                const int py = (ty0 + wave * MT + mt) >> 1, px = x >> 1, PH = p.OH >> 1, PW = p.OW >> 1;
                return (py < PH && px < PW) ? (((int64_t)b * PH + py) * PW + px) * cstride : (int64_t)-1;
            };
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                float bias[1][16];  // (loaded per tile and slice: registers that would otherwise stay live across the MFMA loop)
                load_bias16<1>(p, n0 + 32 * nt, h, bias);
                epilogue_direct<T, MT, 1, decltype(pixoff), decltype(pooloff), FALNET_DMA_EPI_AHEAD>(p, acc[nt], bias, n0 + 32 * nt, lane, pixoff, pooloff);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int j = 0; j < 16; ++j) acc[nt][mt][0][j] = 0.f;
            }
        }
        CD_STAMP();  // 5: (epilogue) done
    }
#undef CD_STAMP
}

// dense 3x3 stride-1 launch in bf16 / f16 with 32-channel-granular sources at the launch size or exactly half of it (the
// caller has verified the canonical tap order and passes flip)
bool falnet_conv_dma_applicable(const falnet_conv_t& p, int min_oh) {
    // (32-bit element offsets inside a sample / the packed weight)
    for (int s = 0; s < p.nsrc; ++s)
        if ((int64_t)p.src[s].H * p.src[s].sy >= (1ll << 31)) return false;
    if ((int64_t)p.w_rows * p.w_taps * p.cin_total >= (1ll << 31)) return false;
    if ((p.cin_total + 64) * 2 > CD_ZERO_BYTES) return false;  // the zero page must cover one row of channels
    if (p.dtype != FALNET_BF16 && p.dtype != FALNET_F16) return false;
    if (p.nsrc < 1 || p.nsrc > 2 || p.w_taps != 9 || p.OH < min_oh || p.OW < 32) return false;
    for (int s = 0; s < p.nsrc; ++s) {
        const falnet_src_t& S = p.src[s];
        if (S.C % 32 || S.C <= 0) return false;
        if (!((S.H == p.IH || 2 * S.H == p.IH) && (S.W == p.IW || 2 * S.W == p.IW))) return false;
    }
    return true;
}

// small_tile: variant 17 -- 4 x 32 positions per workgroup, four waves of ONE row each (conv3x3_dma_kernel<T, 4, 4>): the 16 x 32-pixel maps of
// encoder / decoder level 4 are a single 16-row tile per sample (32 workgroups at B = 8 for 256 channels), with four-row tiles 128.
int falnet_conv_dma_launch(const falnet_conv_t& p, int flip, hipStream_t st, bool small_tile) {
    if (small_tile) {
        const int tiles_x = (p.OW + 31) / 32, tiles_y = (p.OH + 3) / 4;
        const int ntiles = p.B * tiles_x * tiles_y;
        const int ny = (p.Cout + 63) / 64;
        int gx = 256 / ny;
        if (gx < 1) gx = 1;
        if (gx > ntiles) gx = ntiles;
        const dim3 grid((unsigned)gx, (unsigned)ny);
        if (p.dtype == FALNET_F16)
            hipLaunchKernelGGL(HIP_KERNEL_NAME(conv3x3_dma_kernel<f16_t, 4, 4>), grid, dim3(256), 0, st, p, tiles_x, tiles_y, flip, ntiles);
        else
            hipLaunchKernelGGL(HIP_KERNEL_NAME(conv3x3_dma_kernel<bf16_t, 4, 4>), grid, dim3(256), 0, st, p, tiles_x, tiles_y, flip, ntiles);
        FALNET_RETURN_LAUNCH();
    }
    constexpr int TH = 16;
    const int tiles_x = (p.OW + 31) / 32, tiles_y = (p.OH + TH - 1) / TH;
    const int ntiles = p.B * tiles_x * tiles_y;
    const int ny = (p.Cout + 63) / 64;
    int gx = 256 / ny;  // one persistent workgroup per CU (150 KB of LDS each)
    if (gx < 1) gx = 1;
    if (gx > ntiles) gx = ntiles;
    const dim3 grid((unsigned)gx, (unsigned)ny);
    if (p.dtype == FALNET_F16)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(conv3x3_dma_kernel<f16_t, 16, 8>), grid, dim3(512), 0, st, p, tiles_x, tiles_y, flip, ntiles);
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(conv3x3_dma_kernel<bf16_t, 16, 8>), grid, dim3(512), 0, st, p, tiles_x, tiles_y, flip, ntiles);
    FALNET_RETURN_LAUNCH();
}

// ============================================================================================================================
// Data gradient of a 3x3 / stride-2 / pad-1 convolution (models/FAL_netB.py:101-111 conv1..conv6), all four output-parity
// classes in ONE pass over the upstream gradient.
//
// Input-gradient pixel (2i + py, 2j + px) gets the taps (kh, kw) with kh = py + 1 (mod 2), kw = px + 1 (mod 2), read at gout
// position (i + oy, j + ox), oy = (py + 1 - kh) / 2 in {0, 1}: class (0,0) one tap, (0,1) and (1,0) two, (1,1) four -- nine (tap, class)
// pairs, i.e. exactly the MFMA work of a stride-1 3x3 convolution on the gout grid.  The gather kernel ran the classes as four
// independent implicit GEMMs (conv_igemm_multi_kernel): every 128-position tile re-fetched its gout operand once per tap and per
// 32-channel output slice -- PMC: 488 MB fetched for a 34 MB tensor on conv1, 107 TFLOP/s.  Here a (16+1) x (32+1) gout patch and the
// nine 32 x 32 weight tiles of a K chunk arrive by LDS-DMA exactly as in conv3x3_dma_kernel (double buffer, one barrier per chunk,
// persistent workgroups) and feed FOUR accumulator sets (class x 2 rows x 16 registers = 128 per wave); the epilogue (residual add,
// ELU', store) runs once per class on the interleaved output positions.
#define SD_PW 33  // patch columns: 32 + the +1 halo

template <typename T>
__global__ __launch_bounds__(512) void conv3x3_s2d_dma_kernel(const falnet_conv_t p, int tiles_x, int tiles_y, int ntiles, int GH, int GW) {
    constexpr int TH = 16, NWAVES = 8, BN = 32, MT = 2;
    constexpr int KCV = 32;
    constexpr int NPIX = (TH + 1) * SD_PW;
    constexpr int A_PIECES = (NPIX + 15) / 16, B_PIECES = 9 * BN / 16, NPIECES = A_PIECES + B_PIECES;
    constexpr int A_BYTES = A_PIECES * 1024, BUF = NPIECES * 1024;
    __shared__ __attribute__((aligned(1024))) char lds[2 * BUF];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds_base = (unsigned)(unsigned long)(cd_lptr_t)lds;
    const int r = lane & 31, h = lane >> 5;
    const int n0 = blockIdx.y * BN;
    const char* const zero_page = reinterpret_cast<const char*>(g_cd_zero);
    const falnet_src_t& S = p.src[0];
    const int nchunks = S.C / KCV;
    int my_tiles = 0;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) ++my_tiles;
    const int total = my_tiles * nchunks;

    constexpr int KP = (A_PIECES + NWAVES - 1) / NWAVES, KW = (B_PIECES + NWAVES - 1) / NWAVES;
    const int l4 = lane >> 2, segpos = lane & 3;
    const T* const wptr = reinterpret_cast<const T*>(p.weight);
    const T* const sptr = reinterpret_cast<const T*>(S.ptr);
    const T* const zero_t = reinterpret_cast<const T*>(zero_page);
    int64_t w_off[KW];
#pragma unroll
    for (int k = 0; k < KW; ++k) {
        const int wid = wave + NWAVES * k;
        const int tap = wid >> 1, co = n0 + ((wid & 1) << 4) + l4;  // two 16-row pieces per tap tile
        const int gseg = segpos ^ ((lane >> 4) & 3);
        w_off[k] = (wid < B_PIECES && co < p.w_rows) ? (int64_t)(co * p.w_taps + tap) * p.cin_total + gseg * 8 : (int64_t)(zero_t - wptr);
    }
    int64_t a_off[KP];
    int64_t sbat = 0;
    auto tile_coords = [&](int tile, int& b, int& ty0, int& tx0) {
        const int tix = tile % tiles_x;
        const int q = tile / tiles_x;
        ty0 = (q % tiles_y) * TH;
        tx0 = tix * 32;
        b = q / tiles_y;
    };
    auto tile_offsets = [&](int tile) {
        int b, ty0, tx0;
        tile_coords(tile, b, ty0, tx0);
        sbat = (int64_t)b * S.sb;
#pragma unroll
        for (int k = 0; k < KP; ++k) {
            const int pix = 16 * (wave + NWAVES * k) + l4;
            const int pr = pix / SD_PW, pc = pix - pr * SD_PW;
            const int vy = ty0 + pr, vx = tx0 + pc;
            const bool ok = pix < NPIX && vy < GH && vx < GW;
            a_off[k] = ok ? (int64_t)(vy * (int)S.sy + vx * (int)S.sx + (segpos ^ ((pix >> 2) & 3)) * 8) : (int64_t)(zero_t - (sptr + sbat));
        }
    };
    struct Cur { int tile, c; };
    auto advance = [&](Cur& q) {
        if (++q.c == nchunks) {
            q.c = 0;
            q.tile += gridDim.x;
            if (q.tile < ntiles) tile_offsets(q.tile);
        }
    };
    auto issue_piece = [&](const Cur& q, int buf, int i) {  // one piece per MFMA step (see conv3x3_dma_kernel)
        const unsigned dst0 = lds_base + buf * BUF;
        if (i < KP) {
            const int id = wave + NWAVES * i;
            if (id < A_PIECES) cd_glds16(sptr + sbat + q.c * KCV + a_off[i], dst0 + id * 1024);
        } else {
            const int wid = wave + NWAVES * (i - KP);
            if (wid < B_PIECES) cd_glds16(wptr + q.c * KCV + w_off[i - KP], dst0 + A_BYTES + wid * 1024);
        }
    };
    auto issue = [&](const Cur& q, int buf) {
        const T* sbase = sptr + sbat + q.c * KCV;
        const T* wbase = wptr + q.c * KCV;
        const unsigned dst0 = lds_base + buf * BUF;
#pragma unroll
        for (int k = 0; k < KP; ++k) {
            const int id = wave + NWAVES * k;
            if (id < A_PIECES) cd_glds16(sbase + a_off[k], dst0 + id * 1024);
        }
#pragma unroll
        for (int k = 0; k < KW; ++k) {
            const int wid = wave + NWAVES * k;
            if (wid < B_PIECES) cd_glds16(wbase + w_off[k], dst0 + A_BYTES + wid * 1024);
        }
    };
    // fragment read addresses: gout position (2 wave + rs, ox + r) of the tile, rs = mt + oy in 0..2, ox in 0..1
    int a_addr[3][2];
#pragma unroll
    for (int rs = 0; rs < 3; ++rs)
#pragma unroll
        for (int ox = 0; ox < 2; ++ox) {
            const int pp = (wave * MT + rs) * SD_PW + ox + r;
            a_addr[rs][ox] = pp * 64 + ((h ^ ((pp >> 2) & 3)) << 4);
        }
    int b_lane[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) b_lane[ks] = A_BYTES + r * 64 + (((2 * ks + h) ^ ((r >> 2) & 3)) << 4);

    f32x16 acc[4][MT][1];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[c][mt][0][j] = 0.f;
    Cur qi = {(int)blockIdx.x, 0};
    if (total > 0) {
        tile_offsets(qi.tile);
        issue(qi, 0);
        advance(qi);
    }
    int ctile = blockIdx.x, cc = 0;
    for (int it = 0; it < total; ++it) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const bool more = it + 1 < total;
        int bo = (it & 1) * BUF;
        asm volatile("" : "+s"(bo));
        const char* const Bf = lds;
        int aa[3][2], bb[2];
#pragma unroll
        for (int rs = 0; rs < 3; ++rs)
#pragma unroll
            for (int ox = 0; ox < 2; ++ox) aa[rs][ox] = a_addr[rs][ox] + bo;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) bb[ks] = b_lane[ks] + bo;
        s16x8_t fa[2][MT], fb[2];
        auto load_step = [&](int st, int set) {
            const int t = st >> 1, ks = st & 1;
            const int kh = t / 3, kw = t % 3;
            const int oy = kh == 0 ? 1 : 0, ox = kw == 0 ? 1 : 0;  // (py + 1 - kh) / 2 with py = (kh + 1) & 1
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) fa[set][mt] = *reinterpret_cast<const s16x8_t*>(Bf + (aa[mt + oy][ox] ^ (ks << 5)));
            fb[set] = *reinterpret_cast<const s16x8_t*>(Bf + bb[ks] + (t * BN) * 64);
        };
        load_step(0, 0);
#pragma unroll
        for (int st = 0; st < 18; ++st) {
            if (st + 1 < 18) load_step(st + 1, (st + 1) & 1);
            if (st < KP + KW && more) issue_piece(qi, (it + 1) & 1, st);
            const int t = st >> 1;
            const int cls = (((t / 3) + 1) & 1) * 2 + (((t % 3) + 1) & 1);  // (py, px) of this tap
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[cls][mt][0] = H16<T>::mma(fb[st & 1], fa[st & 1][mt], acc[cls][mt][0]);
            __builtin_amdgcn_sched_group_barrier(0x100, MT + 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, MT, 0);
        }
        if (more) advance(qi);
        if (++cc == nchunks) {
            cc = 0;
            int b, ty0, tx0;
            tile_coords(ctile, b, ty0, tx0);
            ctile += gridDim.x;
            const int cstride = p.out_cstride;
            float bias[1][16];
            load_bias16<1>(p, n0, h, bias);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int py = c >> 1, px = c & 1;
                const int x = 2 * (tx0 + r) + px;
                auto pixoff = [&](int mt) -> int64_t {
                    const int y = 2 * (ty0 + wave * MT + mt) + py;
                    if (!(y < p.OH && x < p.OW)) return (int64_t)-1;
                    return (((int64_t)b * p.OH + y) * p.OW + x) * cstride;
                };
                epilogue_direct<T, MT, 1, decltype(pixoff), NoPool, FALNET_DMA_EPI_AHEAD>(p, acc[c], bias, n0, lane, pixoff);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int j = 0; j < 16; ++j) acc[c][mt][0][j] = 0.f;
            }
        }
    }
}

// The four members of a falnet_conv2d_multi launch are the canonical parity classes of one 3x3 stride-2 data gradient?
bool falnet_conv_s2d_dma_applicable(const falnet_conv_t* d, int n) {
    if (n != 4) return false;
    const falnet_conv_t& q = d[3];
    if (q.dtype != FALNET_BF16 && q.dtype != FALNET_F16) return false;
    if (q.nsrc != 1 || q.w_taps != 9 || q.ksplit > 1 || q.out_layout != FALNET_OUT_NHWC || q.bias || q.pool_out) return false;
    const falnet_src_t& S = q.src[0];
    if (S.C % 32 || S.C <= 0 || S.H != q.IH || S.W != q.IW) return false;
    if ((q.OH & 1) || (q.OW & 1) || q.IH * 2 != q.OH || q.IW * 2 != q.OW || q.IH < 16 || q.IW < 32) return false;
    if ((int64_t)S.H * S.sy >= (1ll << 31) || (int64_t)q.w_rows * 9 * q.cin_total >= (1ll << 31) || (q.cin_total + 64) * 2 > CD_ZERO_BYTES) return false;
    if (q.cin_total != S.C) return false;
    for (int c = 0; c < 4; ++c) {
        const falnet_conv_t& p = d[c];
        const int py = c >> 1, px = c & 1;
        if (p.src[0].ptr != S.ptr || p.weight != q.weight || p.out != q.out || p.addend != q.addend || p.actout != q.actout || p.act != q.act ||
            p.actout_kind != q.actout_kind || p.Cout != q.Cout || p.w_rows != q.w_rows || p.osy != 2 || p.osx != 2 || p.ooy != py || p.oox != px ||
            p.TH != q.OH / 2 || p.TW != q.OW / 2 || p.B != q.B || p.OH != q.OH || p.OW != q.OW || p.isy != 1 || p.isx != 1)
            return false;
        int k = 0;
        for (int kh = 0; kh < 3; ++kh) {
            if ((py + 1 - kh) % 2) continue;
            for (int kw = 0; kw < 3; ++kw) {
                if ((px + 1 - kw) % 2) continue;
                if (k >= p.ntaps || p.tap_dy[k] != (py + 1 - kh) / 2 || p.tap_dx[k] != (px + 1 - kw) / 2 || p.tap_w[k] != kh * 3 + kw) return false;
                ++k;
            }
        }
        if (k != p.ntaps) return false;
    }
    return true;
}

int falnet_conv_s2d_dma_launch(const falnet_conv_t* d, hipStream_t st) {
    const falnet_conv_t& p = d[3];
    const int GH = p.IH, GW = p.IW;  // the upstream-gradient grid
    const int tiles_x = (GW + 31) / 32, tiles_y = (GH + 15) / 16;
    const int ntiles = p.B * tiles_x * tiles_y;
    const int ny = (p.Cout + 31) / 32;
    int gx = 256 / ny;
    if (gx < 1) gx = 1;
    if (gx > ntiles) gx = ntiles;
    const dim3 grid((unsigned)gx, (unsigned)ny);
    if (p.dtype == FALNET_F16)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(conv3x3_s2d_dma_kernel<f16_t>), grid, dim3(512), 0, st, p, tiles_x, tiles_y, ntiles, GH, GW);
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(conv3x3_s2d_dma_kernel<bf16_t>), grid, dim3(512), 0, st, p, tiles_x, tiles_y, ntiles, GH, GW);
    FALNET_RETURN_LAUNCH();
}

// ============================================================================================================================
// Forward 3x3 / stride-2 / pad-1 convolution (models/FAL_netB.py:101-111 conv1..conv4), 16-bit operands, LDS-DMA double buffer.
//
// A stride-2 output tile reads (2 TH + 1) x 65 input pixels for TH x 32 outputs -- four times the footprint of a stride-1 tile -- so
// the K chunk is 16 channels (32 B per pixel): 17 x 65 pixels (35 KB) + nine BN x 16 weight tiles per buffer, two buffers, one
// persistent 8-wave workgroup per CU, wave w = output row w of the 8 x 32 tile.  The DMA de-interleaves the patch columns by parity
// while it fetches (every lane names its own source pixel): LDS row = [even columns 0..32 | odd columns 0..31], so the 32 lanes of a
// fragment read consecutive 32-B pixel rows for every tap (kw = 0, 2: even plane at i, i + 1; kw = 1: odd plane at i).  The two 16-B
// halves of row R are exchanged when bit 3 of R is set (on the SOURCE address): lanes i and i + 8 of a ds_read_b128 then hit
// different banks.  One MFMA K step (16 channels) per tap and chunk; the gather kernel this replaces ran these layers at 116-253 TFLOP/s.
// PMC (128 -> 256 @64x128, 25.9 us, 373 TFLOP/s): 0 bank conflicts, per wave and chunk 214 VALU + 139 SALU + 27 LDS + 18 MFMA + 7 DMA
// instructions in ~6.3 k cycles, waves waiting 42 % of the time, MFMA pipe busy 18 %.  Measured neutral and not kept: a third buffer
// (two chunks in flight, counted vmcnt) and fragment reads three taps ahead -- the loop is bound by neither the DMA latency nor the LDS
// round trip of a tap, but by the per-chunk barrier + issue phases that all eight waves go through in lock step.
#define SF_COLS 65

template <typename T, int BN>
__global__ __launch_bounds__(512) void conv3x3_s2f_dma_kernel(const falnet_conv_t p, int tiles_x, int tiles_y, int ntiles) {
    constexpr int TH = 8, NWAVES = 8, NT = BN / 32;
    constexpr int KCV = 16;  // input channels per chunk (32 B per pixel / weight row)
    constexpr int NPIX = (2 * TH + 1) * SF_COLS;
    constexpr int A_PIECES = (NPIX + 31) / 32, B_PIECES = 9 * BN / 32, NPIECES = A_PIECES + B_PIECES;
    constexpr int A_BYTES = A_PIECES * 1024, BUF = NPIECES * 1024;
    __shared__ __attribute__((aligned(1024))) char lds[2 * BUF];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds_base = (unsigned)(unsigned long)(cd_lptr_t)lds;
    const int r = lane & 31, h = lane >> 5;
    const int n0 = blockIdx.y * BN;
    const char* const zero_page = reinterpret_cast<const char*>(g_cd_zero);
    const int nsrc = p.nsrc, IH = p.IH, IW = p.IW;
    const int C0 = p.src[0].C, C1 = nsrc > 1 ? p.src[1].C : 0;
    const int nchunks = (C0 + C1) / KCV;
    int my_tiles = 0;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) ++my_tiles;
    const int total = my_tiles * nchunks;

    // ---- DMA geometry: a 1-KiB piece = 32 rows of 32 B; lane -> (row lane >> 1, physical half lane & 1) ----
    constexpr int KP = (A_PIECES + NWAVES - 1) / NWAVES, KW = (B_PIECES + NWAVES - 1) / NWAVES;
    const int l2 = lane >> 1, half = lane & 1;
    const T* const wptr = reinterpret_cast<const T*>(p.weight);
    const T* const zero_t = reinterpret_cast<const T*>(zero_page);
    int64_t w_off[KW];
#pragma unroll
    for (int k = 0; k < KW; ++k) {
        const int wid = wave + NWAVES * k;
        const int R = 32 * wid + l2;  // row of the [tap][BN] weight image
        const int tap = R / BN, co = n0 + R % BN;
        const int lseg = half ^ ((R >> 3) & 1);
        w_off[k] = (wid < B_PIECES && co < p.w_rows) ? (int64_t)(co * p.w_taps + tap) * p.cin_total + lseg * 8 : (int64_t)(zero_t - wptr);
    }
    int64_t a_off[KP];
    const T* sptr[2] = {reinterpret_cast<const T*>(p.src[0].ptr), reinterpret_cast<const T*>(nsrc > 1 ? p.src[1].ptr : p.src[0].ptr)};
    int64_t sbat[2] = {0, 0};
    auto tile_coords = [&](int tile, int& b, int& ty0, int& tx0) {
        const int tix = tile % tiles_x;
        const int q = tile / tiles_x;
        ty0 = (q % tiles_y) * TH;
        tx0 = tix * 32;
        b = q / tiles_y;
    };
    auto tile_offsets = [&](int tile, int s2) {
        int b, ty0, tx0;
        tile_coords(tile, b, ty0, tx0);
        const falnet_src_t& S = s2 == 0 ? p.src[0] : p.src[1];
        sbat[s2] = (int64_t)b * S.sb;
#pragma unroll
        for (int k = 0; k < KP; ++k) {
            const int lp = 32 * (wave + NWAVES * k) + l2;  // LDS pixel row: [patch row][even cols 0..32 | odd cols 0..31]
            const int pr = lp / SF_COLS, q = lp - pr * SF_COLS;
            const int c = q < 33 ? 2 * q : 2 * (q - 33) + 1;  // patch column
            const int vy = 2 * ty0 - 1 + pr, vx = 2 * tx0 - 1 + c;
            const bool ok = lp < NPIX && vy >= 0 && vy < IH && vx >= 0 && vx < IW;
            const int lseg = half ^ ((lp >> 3) & 1);
            a_off[k] = ok ? (int64_t)(vy * (int)S.sy + vx * (int)S.sx + lseg * 8) : (int64_t)(zero_t - (reinterpret_cast<const T*>(S.ptr) + sbat[s2]));
        }
    };
    struct Cur { int tile, c, s, c0, kofs; };
    auto advance = [&](Cur& q) {
        if (++q.c == nchunks) {
            q.c = 0; q.s = 0; q.c0 = 0; q.kofs = 0;
            q.tile += gridDim.x;
            if (q.tile < ntiles) tile_offsets(q.tile, 0);
            return;
        }
        q.c0 += KCV;
        q.kofs += KCV;
        if (q.s == 0 && q.c0 >= C0) {
            q.s = 1;
            q.c0 = 0;
            tile_offsets(q.tile, 1);
        }
    };
    auto issue_piece = [&](const Cur& q, int buf, int i) {  // one piece per MFMA step (see conv3x3_dma_kernel)
        const unsigned dst0 = lds_base + buf * BUF;
        if (i < KP) {
            const int id = wave + NWAVES * i;
            const T* sbase = (q.s == 0 ? sptr[0] + sbat[0] : sptr[1] + sbat[1]) + q.c0;
            if (id < A_PIECES) cd_glds16(sbase + a_off[i], dst0 + id * 1024);
        } else {
            const int wid = wave + NWAVES * (i - KP);
            if (wid < B_PIECES) cd_glds16(wptr + q.kofs + w_off[i - KP], dst0 + A_BYTES + wid * 1024);
        }
    };
    auto issue = [&](const Cur& q, int buf) {
        const T* sbase = (q.s == 0 ? sptr[0] + sbat[0] : sptr[1] + sbat[1]) + q.c0;
        const T* wbase = wptr + q.kofs;
        const unsigned dst0 = lds_base + buf * BUF;
#pragma unroll
        for (int k = 0; k < KP; ++k) {
            const int id = wave + NWAVES * k;
            if (id < A_PIECES) cd_glds16(sbase + a_off[k], dst0 + id * 1024);
        }
#pragma unroll
        for (int k = 0; k < KW; ++k) {
            const int wid = wave + NWAVES * k;
            if (wid < B_PIECES) cd_glds16(wbase + w_off[k], dst0 + A_BYTES + wid * 1024);
        }
    };
    // fragment read addresses: tap (kh, kw) of output (wave, r) = patch row 2 wave + kh, plane kw & 1, entry r + (kw >> 1)
    int a_addr[3][3];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int lp = (2 * wave + kh) * SF_COLS + ((kw & 1) ? 33 : 0) + r + (kw >> 1);
            a_addr[kh][kw] = lp * 32 + ((h ^ ((lp >> 3) & 1)) << 4);
        }
    int b_lane[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int R = nt * 32 + r;
        b_lane[nt] = A_BYTES + R * 32 + ((h ^ ((R >> 3) & 1)) << 4);  // + tap * BN * 32 (BN a multiple of 16: the swizzle bit is the same)
    }

    f32x16 acc[NT][1][1];  // [nt]: the epilogue runs per 32-channel slice (one slice's bias / operands in registers at a time)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[nt][0][0][j] = 0.f;
    Cur qi = {(int)blockIdx.x, 0, 0, 0, 0};
    if (total > 0) {
        tile_offsets(qi.tile, 0);
        issue(qi, 0);
        advance(qi);
    }
    int ctile = blockIdx.x, cc = 0;
    for (int it = 0; it < total; ++it) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const bool more = it + 1 < total;
        int bo = (it & 1) * BUF;
        asm volatile("" : "+s"(bo));
        const char* const Bf = lds;
        int aa[3][3], bb[NT];  // ONE add per lane address and chunk (see conv3x3_dma_kernel); tap offsets of the weights are immediates
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) aa[kh][kw] = a_addr[kh][kw] + bo;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bb[nt] = b_lane[nt] + bo;
        s16x8_t fa[2], fb[2][NT];
        auto load_step = [&](int t, int set) {
            fa[set] = *reinterpret_cast<const s16x8_t*>(Bf + aa[t / 3][t % 3]);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) fb[set][nt] = *reinterpret_cast<const s16x8_t*>(Bf + bb[nt] + t * BN * 32);
        };
        load_step(0, 0);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            if (t + 1 < 9) load_step(t + 1, (t + 1) & 1);
            if (t < KP + KW && more) issue_piece(qi, (it + 1) & 1, t);
            if (t == 8 && KP + KW > 9 && more) {  // (BN = 128: more pieces than taps)
#pragma unroll
                for (int i = 9; i < KP + KW; ++i) issue_piece(qi, (it + 1) & 1, i);
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[nt][0][0] = H16<T>::mma(fb[t & 1][nt], fa[t & 1], acc[nt][0][0]);
            __builtin_amdgcn_sched_group_barrier(0x100, 1 + NT, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, NT, 0);
        }
        if (more) advance(qi);
        if (++cc == nchunks) {
            cc = 0;
            int b, ty0, tx0;
            tile_coords(ctile, b, ty0, tx0);
            ctile += gridDim.x;
            const int cstride = p.out_cstride;
            const int x = tx0 + r, y = ty0 + wave;
            auto pixoff = [&](int) -> int64_t {
                if (!(y < p.OH && x < p.OW)) return (int64_t)-1;
                return (((int64_t)b * p.OH + y) * p.OW + x) * cstride;
            };
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                float bias[1][16];
                load_bias16<1>(p, n0 + 32 * nt, h, bias);
                epilogue_direct<T, 1, 1, decltype(pixoff), NoPool, FALNET_DMA_EPI_AHEAD>(p, acc[nt], bias, n0 + 32 * nt, lane, pixoff);
#pragma unroll
                for (int j = 0; j < 16; ++j) acc[nt][0][0][j] = 0.f;
            }
        }
    }
}

// canonical forward 3x3 stride-2 pad-1 launch (taps (kh - 1, kw - 1, kh * 3 + kw)), 16-bit, NHWC output, sources at the input size
bool falnet_conv_s2f_dma_applicable(const falnet_conv_t& p) {
    if (p.dtype != FALNET_BF16 && p.dtype != FALNET_F16) return false;
    if (p.ntaps != 9 || p.w_taps != 9 || p.isy != 2 || p.isx != 2 || p.osy != 1 || p.osx != 1 || p.ooy || p.oox) return false;
    if (p.out_layout != FALNET_OUT_NHWC || p.pool_out || p.ksplit > 1 || p.nsrc < 1 || p.nsrc > 2) return false;
    for (int t = 0; t < 9; ++t)
        if (p.tap_dy[t] != t / 3 - 1 || p.tap_dx[t] != t % 3 - 1 || p.tap_w[t] != t) return false;
    if (p.TH != p.OH || p.TW != p.OW || p.OH != (p.IH + 1) / 2 || p.OW != (p.IW + 1) / 2 || p.OH < 8 || p.OW < 32) return false;
    for (int s = 0; s < p.nsrc; ++s) {
        const falnet_src_t& S = p.src[s];
        if (S.C % 16 || S.C <= 0 || S.H != p.IH || S.W != p.IW) return false;
        if ((int64_t)S.H * S.sy >= (1ll << 31)) return false;
    }
    if ((int64_t)p.w_rows * 9 * p.cin_total >= (1ll << 31) || (p.cin_total + 64) * 2 > CD_ZERO_BYTES) return false;
    return true;
}

int falnet_conv_s2f_dma_launch(const falnet_conv_t& p, hipStream_t st) {
    const int tiles_x = (p.OW + 31) / 32, tiles_y = (p.OH + 7) / 8;
    const int ntiles = p.B * tiles_x * tiles_y;
    static const bool allow128 = [] { const char* e = falnet_ab_env("FALNET_S2F_BN128"); return e && e[0] == '1'; }();  // (128-channel form: 36-41 spilled VGPRs)
    const bool wide = allow128 && p.w_rows % 128 == 0 && p.Cout > 64;
    const int bn = wide ? 128 : 64;
    const int ny = (p.Cout + bn - 1) / bn;
    int gx = 256 / ny;
    if (gx < 1) gx = 1;
    if (gx > ntiles) gx = ntiles;
    const dim3 grid((unsigned)gx, (unsigned)ny);
#define SF_L(T) do { if (wide) hipLaunchKernelGGL(HIP_KERNEL_NAME(conv3x3_s2f_dma_kernel<T, 128>), grid, dim3(512), 0, st, p, tiles_x, tiles_y, ntiles); \
                     else hipLaunchKernelGGL(HIP_KERNEL_NAME(conv3x3_s2f_dma_kernel<T, 64>), grid, dim3(512), 0, st, p, tiles_x, tiles_y, ntiles); } while (0)
    FALNET_DISPATCH_16(p.dtype, SF_L);
#undef SF_L
    FALNET_RETURN_LAUNCH();
}
