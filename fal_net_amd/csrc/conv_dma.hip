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

#ifndef FALNET_DMA_ROW_REUSE
#define FALNET_DMA_ROW_REUSE 1  // 16 x 32 tiles: pixel fragments read once per (row, dx, channel half) instead of once per tap (0: A/B builds)
#endif
#ifndef FALNET_DMA_EPI_AHEAD
#define FALNET_DMA_EPI_AHEAD -1  // conv_epilogue.h: epilogue_direct's operand prefetch depth (-1: conditional loads at the point of use)
#endif
#ifndef FALNET_DMA16_EPI_AHEAD
#define FALNET_DMA16_EPI_AHEAD 0  // the same for conv3x3_dma16_kernel: branch-free fetch of the slab's residual / activation operands in front of its arithmetic (218 registers; -1 -> 0: 3-10 % on launches with such operands, 1 no better: profiles/r05_dma16.txt)
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
    __shared__ __attribute__((aligned(16))) float lds_bias[BN];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds_base = (unsigned)(unsigned long)(cd_lptr_t)lds;
    const int r = lane & 31, h = lane >> 5;
    const int n0 = blockIdx.y * BN;
    const char* const zero_page = reinterpret_cast<const char*>(g_cd_zero);
    stage_bias_lds(p, n0, BN, lds_bias);  // (read by the tile epilogues: at least one barrier later)

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
    // Patch piece geometry (as conv3x3_dma16_kernel): a lane's patch pixel is the same in every tile -- element offset = (tile origin, wave-uniform)
    // + (lane constant per source) -- and the cursors carry (b, ty, tx) forward by the decomposed grid step: no division and no 64-bit vector
    // multiply per tile.
    const T* sptr[2] = {reinterpret_cast<const T*>(p.src[0].ptr), reinterpret_cast<const T*>(nsrc > 1 ? p.src[1].ptr : p.src[0].ptr)};
    struct TPos { int b, ty, tx; };
    TPos step_;
    {
        const int g = (int)gridDim.x;
        step_.tx = g % tiles_x;
        const int q = g / tiles_x;
        step_.ty = q % tiles_y;
        step_.b = q / tiles_y;
    }
    auto pos_of = [&](int tile) {
        TPos t;
        t.tx = tile % tiles_x;
        const int q = tile / tiles_x;
        t.ty = q % tiles_y;
        t.b = q / tiles_y;
        return t;
    };
    auto pos_next = [&](TPos& t) {
        t.tx += step_.tx;
        if (t.tx >= tiles_x) { t.tx -= tiles_x; ++t.ty; }
        t.ty += step_.ty;
        if (t.ty >= tiles_y) { t.ty -= tiles_y; ++t.b; }
        t.b += step_.b;
    };
    int pr_[KP], pc_[KP], a_lc[2][KP];
#pragma unroll
    for (int k = 0; k < KP; ++k) {
        const int pix = 16 * (wave + NWAVES * k) + l4;
        const int pr = pix / CD_PW, pc = pix - pr * CD_PW;
        pr_[k] = pix < NPIX ? pr : -(1 << 20);
        pc_[k] = pc;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const falnet_src_t& S = s2 == 0 ? p.src[0] : p.src[(nsrc > 1) ? 1 : 0];
            const int hs = S.H != IH ? 1 : 0, ws = S.W != IW ? 1 : 0;  // exact 2x nearest upsampling (dispatcher checks)
            a_lc[s2][k] = ((pr - 1) >> hs) * (int)S.sy + ((pc - 1) >> ws) * (int)S.sx + (segpos ^ ((pix >> 2) & 3)) * 8;
        }
    }
    int a_off[KP];      // element offset of piece k from a_base (valid when bit k of a_ok is set; the zero page otherwise)
    unsigned a_ok = 0;
    const T* a_base = sptr[0];  // source pointer + image offset + tile origin of the issue cursor
    auto tile_offsets = [&](const TPos& t, int s2) {  // patch offsets of the issue cursor's (tile, source)
        const falnet_src_t& S = s2 == 0 ? p.src[0] : p.src[1];
        const int hs = S.H != IH ? 1 : 0, ws = S.W != IW ? 1 : 0;
        const int ty0 = t.ty * TH, tx0 = t.tx * 32;
        a_base = sptr[s2] + ((int64_t)t.b * S.sb + (int64_t)((ty0 >> hs) * (int)S.sy + (tx0 >> ws) * (int)S.sx));
        a_ok = 0;
#pragma unroll
        for (int k = 0; k < KP; ++k) {
            const int vy = ty0 - 1 + pr_[k], vx = tx0 - 1 + pc_[k];
            a_ok |= (vy >= 0 && vy < IH && vx >= 0 && vx < IW) ? 1u << k : 0u;
            a_off[k] = a_lc[s2][k];
        }
    };
    struct Cur { int tile, c, s, c0, kofs; TPos t; };  // issue cursor: tile, chunk, source, channel offset, weight-row offset, tile position
    auto advance = [&](Cur& q) {
        if (++q.c == nchunks) {
            q.c = 0; q.s = 0; q.c0 = 0; q.kofs = 0;
            q.tile += gridDim.x;
            pos_next(q.t);
            if (q.tile < ntiles) tile_offsets(q.t, 0);
            return;
        }
        q.c0 += KCV;
        q.kofs += KCV;
        if (q.s == 0 && q.c0 >= C0) {
            q.s = 1;
            q.c0 = 0;
            tile_offsets(q.t, 1);
        }
    };
    // one piece of the issue cursor's chunk (i = 0 .. KP + KW - 1: patch pieces, then weight pieces); the pieces of chunk it + 1 are
    // issued one per MFMA step inside chunk it's loop -- in-kernel stamps of the all-at-once form: 2 000-2 500 of a chunk's 9 000 cycles
    // were spent in the issue of ~10 gathering DMA instructions with the matrix pipe idle (both waves of a SIMD issued at the same time)
    auto issue_piece = [&](const Cur& q, int buf, int i) {
        const unsigned dst0 = lds_base + buf * BUF;
        if (i < KP) {
            const int id = wave + NWAVES * i;
            const T* src = ((a_ok >> i) & 1) ? a_base + q.c0 + a_off[i] : zero_t;
            if (id < A_PIECES) cd_glds16(src, dst0 + id * 1024);
        } else {
            const int wid = wave + NWAVES * (i - KP);
            if (wid < B_PIECES) cd_glds16(wptr + q.kofs + w_off[i - KP], dst0 + A_BYTES + wid * 1024);
        }
    };
    auto issue = [&](const Cur& q, int buf) {
        const T* wbase = wptr + q.kofs;
        const unsigned dst0 = lds_base + buf * BUF;
#pragma unroll
        for (int k = 0; k < KP; ++k) {
            const int id = wave + NWAVES * k;
            if (id < A_PIECES) {  // wave-uniform
                const T* src = ((a_ok >> k) & 1) ? a_base + q.c0 + a_off[k] : zero_t;
                cd_glds16(src, dst0 + id * 1024);
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
    Cur qi = {(int)blockIdx.x, 0, 0, 0, 0, pos_of((int)blockIdx.x)};
    TPos ct = qi.t;  // the compute cursor's tile
    if (total > 0) {
        tile_offsets(qi.t, 0);
        issue(qi, 0);
        advance(qi);
    }
    int cc = 0;  // chunk of the compute cursor's tile
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
        // 18 steps of four MFMAs; the fragment reads of step i + 1 are issued in front of the MFMAs of step i; sched_group_barrier pins that order --
        // left alone, hipcc hoists a dozen steps of reads and spills.
        if constexpr (MT == 2 && FALNET_DMA_ROW_REUSE) {
            // Two rows per wave: the pixel fragment of patch row rs, column offset dx, channel half ks serves output row mt under tap row dy = rs - mt,
            // i.e. up to TWO taps.  Steps walk (dx, ks) groups x dy, and a group's four row fragments are read once: 24 + 36 fragment reads per chunk
            // instead of 36 + 36 (the weight fragments have no such reuse).  Row r of a group is first needed at dy = max(0, r - 1) and its registers
            // are free from the step after its last use, so the rows rotate through FOUR fragment registers with every read one step ahead.
            s16x8_t fr[4], fb[2][NT];
            auto a_read = [&](int g, int rs) { return *reinterpret_cast<const s16x8_t*>(Bf + (aa[rs][g >> 1] ^ ((g & 1) << 5))); };
            auto b_read = [&](int st, int set) {
                const int g = st / 3, dy = st % 3, t = dy * 3 + (g >> 1);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) fb[set][nt] = *reinterpret_cast<const s16x8_t*>(Bf + bb[g & 1] + (t * BN + nt * 32) * 64);
            };
            fr[0] = a_read(0, 0);
            fr[1] = a_read(0, 1);
            b_read(0, 0);
#pragma unroll
            for (int st = 0; st < 18; ++st) {
                const int g = st / 3, dy = st % 3;
                if (st + 1 < 18) b_read(st + 1, (st + 1) & 1);
                if (dy == 0) fr[2] = a_read(g, 2);
                else if (dy == 1) fr[3] = a_read(g, 3);
                else if (g + 1 < 6) {
                    fr[0] = a_read(g + 1, 0);
                    fr[1] = a_read(g + 1, 1);
                }
                if (st < KP + KW && more) issue_piece(qi, (it + 1) & 1, st);  // (wave-uniform)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[nt][mt][0] = H16<T>::mma(fb[st & 1][nt], fr[mt + dy], acc[nt][mt][0]);
                if (st + 1 == 18) __builtin_amdgcn_sched_group_barrier(0x100, 0, 0);
                else if (dy == 2) __builtin_amdgcn_sched_group_barrier(0x100, NT + 2, 0);
                else __builtin_amdgcn_sched_group_barrier(0x100, NT + 1, 0);  // DS reads of the next step
                __builtin_amdgcn_sched_group_barrier(0x008, MT * NT, 0);      // this step's MFMAs
            }
        } else {
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
        }
        if (more) advance(qi);
        CD_STAMP();  // 4: MFMAs issued
        if (++cc == nchunks) {  // tile finished: epilogue straight from the accumulators, then the next tile starts from zero
            cc = 0;
            const int b = ct.b, ty0 = ct.ty * TH, tx0 = ct.tx * 32;
            pos_next(ct);
            const int cstride = p.out_cstride;
            const int x = tx0 + r;
            const bool planar_out = p.out_layout == FALNET_OUT_PLANAR_F32;
            // (wave-uniform 64-bit origin of the tile + a 32-bit lane part: no 64-bit vector multiply per slab)
            const int64_t obase = planar_out ? ((int64_t)b * p.Cout * p.OH + ty0) * p.OW + tx0 : (((int64_t)b * p.OH + ty0) * p.OW + tx0) * cstride;
            const int lstride = planar_out ? 1 : cstride;
            auto pixoff = [&](int mt) -> int64_t {
                const int y = ty0 + wave * MT + mt;
                if (!(y < p.OH && x < p.OW)) return (int64_t)-1;
                return obase + ((wave * MT + mt) * p.OW + r) * lstride;
            };
            auto pooloff = [&](int mt) -> int64_t {
                const int py = (ty0 + wave * MT + mt) >> 1, px = x >> 1, PH = p.OH >> 1, PW = p.OW >> 1;
                return (py < PH && px < PW) ? (((int64_t)b * PH + py) * PW + px) * cstride : (int64_t)-1;
            };
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                float bias[1][16];  // (loaded per tile and slice: registers that would otherwise stay live across the MFMA loop)
                load_bias16_lds(lds_bias, 32 * nt, h, bias);
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

// th = 16: variant 13 (8 waves of two rows).  th = 4: variant 17 -- four waves of ONE row each: the 16 x 32-pixel maps of encoder / decoder
// level 4 are a single 16-row tile per sample (32 workgroups at B = 8 for 256 channels), with four-row tiles 128.  th = 8: variant 20 -- eight
// waves of one row each: the 32 x 64 maps of level 3 are 128 sixteen-row tiles x 4 channel blocks (two tiles per CU, half the CUs of the second
// round idle) or 512 four-row tiles whose nine weight tiles (37 KB per 32-channel chunk) are staged for only 128 positions; eight-row tiles
// are 256 x 4: every CU busy and 2x the positions per staged weight byte of the four-row form.
template <typename T, int TH, int NWAVES>
static void dma_launch_t(const falnet_conv_t& p, int flip, hipStream_t st) {
    const int tiles_x = (p.OW + 31) / 32, tiles_y = (p.OH + TH - 1) / TH;
    const int ntiles = p.B * tiles_x * tiles_y;
    const int ny = (p.Cout + 63) / 64;
    int gx = 256 / ny;  // one persistent workgroup per CU
    if (gx < 1) gx = 1;
    if (gx > ntiles) gx = ntiles;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(conv3x3_dma_kernel<T, TH, NWAVES>), dim3((unsigned)gx, (unsigned)ny), dim3(NWAVES * 64), 0, st, p, tiles_x, tiles_y, flip, ntiles);
}

int falnet_conv_dma_launch(const falnet_conv_t& p, int flip, hipStream_t st, int th) {
#define DMA_L(T) do { if (th == 4) dma_launch_t<T, 4, 4>(p, flip, st); else if (th == 8) dma_launch_t<T, 8, 8>(p, flip, st); else dma_launch_t<T, 16, 8>(p, flip, st); } while (0)
    FALNET_DISPATCH_16(p.dtype, DMA_L);
#undef DMA_L
    FALNET_RETURN_LAUNCH();
}

// ============================================================================================================================
// conv3x3_dma2_kernel (falnet_conv2d variant 21, round 5): the same convolution as conv3x3_dma_kernel<T, 16, 8>, re-cut so that the two waves
// of a SIMD belong to DIFFERENT workgroups.
//
// What bounds variant 13 (profiles/r04_dma_epilogue_probes.txt, profiles/r05_sq_counters.txt): its eight waves pass one barrier per K chunk, so
// both waves of every SIMD are in their MFMA loop, at the chunk barrier and in the tile epilogue (5 000-7 400 cycles of VALU + stores, 8-30 % of a
// tile) AT THE SAME TIME: the matrix pipe is 40-61 % busy (SQ_VALU_MFMA_BUSY_CYCLES) and nothing of the same workgroup can cover for it.  Four
// waves of four rows each on the same tile (r04 probe G: 0.5 instead of 0.83 fragment reads per MFMA) lost 15-30 % for the same reason from the
// other side: one wave per SIMD, nothing covers a DMA-issue stall or the epilogue.  Here:
//   * 16 x 32 positions x 64 channels per workgroup of FOUR waves (one per SIMD), four output rows per wave: the six patch-row fragments of a
//     column offset serve four output rows, a weight fragment four MFMAs -- 36 fragment reads per 72 MFMAs;
//   * K in 16-channel chunks (32-B LDS rows): a chunk is 20 + 18 one-KiB pieces = 38 KiB, two buffers = 76 KiB, so TWO workgroups are resident
//     per CU (__launch_bounds__(256, 2): 256 registers per wave) and the SIMD's second wave runs another tile at another phase: one workgroup's
//     epilogue, chunk barrier and DMA issue overlap the other's MFMAs;
//   * swizzle for 32-B rows: the two 16-B halves of a pixel row are exchanged when bit 3 of its patch COLUMN (of the channel row for weights) is
//     set -- applied on the DMA's source address, the destination stays lane-linear -- which makes every ds_read_b128 of 32 consecutive rows
//     conflict-free AND independent of the patch row, so the fragment of patch row rs is the row-0 address plus an immediate offset.
// Same DMA / cursor / zero-page machinery as conv3x3_dma_kernel; NHWC outputs only (the planar-f32 logits launch stays on variant 13).
#define C2_PW 34
// NWAVES = 8 (variant 22): the same wave program on 32 x 32-position tiles, ONE workgroup of eight waves per CU (2 x 55 KiB of LDS) -- the tile
// VERDICT r4 next #1 prescribes: 55 instead of 75 KiB staged per 576 MFMAs (the 34 x 34 patch serves 32 rows) AND 0.5 instead of 0.83 fragment
// reads per MFMA.  Both cuts of the 16 x 32 tile are paced at ~10 B/clk of LDS-DMA per CU (variant 13: 75 KiB per ~7 500-cycle chunk; variant
// 21 the same bytes per MFMA); fewer reads alone (variant 21) or fewer bytes alone (r03's 128-channel tile) each left the other bound standing.
template <typename T, bool POOL, int NWAVES>
__global__ __launch_bounds__(NWAVES * 64, 2) void conv3x3_dma2_kernel(const falnet_conv_t p, int tiles_x, int tiles_y, int flip, int ntiles, int skew) {
    constexpr int MT = 4, TH = NWAVES * MT, BN = 64, NT = BN / 32;
    static_assert(sizeof(T) == 2, "16-bit operands");
    constexpr int KCV = 16;  // input channels per chunk (32 B per pixel / weight row)
    constexpr int NPIX = (TH + 2) * C2_PW;
    constexpr int A_PIECES = (NPIX + 31) / 32, B_PIECES = 9 * BN / 32, NPIECES = A_PIECES + B_PIECES;
    constexpr int A_BYTES = A_PIECES * 1024, BUF = NPIECES * 1024;
    constexpr int ROWB = C2_PW * 32;  // bytes between patch rows
    static_assert(2 * BUF + 256 <= (NWAVES == 4 ? 80 : 160) * 1024, "two workgroups (four waves) / one workgroup (eight waves) per CU");
    __shared__ __attribute__((aligned(1024))) char lds[2 * BUF];
    __shared__ __attribute__((aligned(16))) float lds_bias[BN];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds_base = (unsigned)(unsigned long)(cd_lptr_t)lds;
    const int r = lane & 31, h = lane >> 5;
    const int n0 = blockIdx.y * BN;
    const char* const zero_page = reinterpret_cast<const char*>(g_cd_zero);
    stage_bias_lds(p, n0, BN, lds_bias);  // (read by the tile epilogues: at least one barrier later)

    const int nsrc = p.nsrc, IH = p.IH, IW = p.IW;
    const int C0 = p.src[0].C, C1 = nsrc > 1 ? p.src[1].C : 0;
    const int nchunks = (C0 + C1) / KCV;
    int my_tiles = 0;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) ++my_tiles;
    const int total = my_tiles * nchunks;
    // Phase skew (experiment switch, 0 in the product): the second half of the grid (dispatched as the CUs' SECOND residents) starts
    // `skew` x 64 cycles late, so that the two residents of a CU do not reach their epilogues together.
    if (skew > 0 && (int)(blockIdx.y * gridDim.x + blockIdx.x) >= (int)(gridDim.x * gridDim.y + 1) / 2) {
        const long long t0 = (long long)__builtin_amdgcn_s_memtime();
        while ((long long)__builtin_amdgcn_s_memtime() - t0 < (long long)skew * 64) __builtin_amdgcn_s_sleep(32);
    }

    // ---- DMA geometry of this lane: row l2 of a 32-row piece, 16-B half `seg`; the half it FETCHES is swizzled ----
    constexpr int KP = (A_PIECES + NWAVES - 1) / NWAVES, KW = (B_PIECES + NWAVES - 1) / NWAVES;
    const int l2 = lane >> 1, seg = lane & 1;
    const T* const wptr = reinterpret_cast<const T*>(p.weight);
    const T* const zero_t = reinterpret_cast<const T*>(zero_page);
    int64_t w_off[KW];
#pragma unroll
    for (int k = 0; k < KW; ++k) {
        const int wid = wave + NWAVES * k;
        const int tap = wid >> 1, co = n0 + ((wid & 1) << 5) + l2;  // two 32-row pieces per tap tile
        const int gseg = seg ^ ((l2 >> 3) & 1);
        w_off[k] = (wid < B_PIECES && co < p.w_rows) ? (int64_t)(co * p.w_taps + (flip ? 8 - tap : tap)) * p.cin_total + gseg * 8 : (int64_t)(zero_t - wptr);
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
        const int hs = S.H != IH ? 1 : 0, ws = S.W != IW ? 1 : 0;  // exact 2x nearest upsampling (dispatcher checks)
#pragma unroll
        for (int k = 0; k < KP; ++k) {
            const int pix = 32 * (wave + NWAVES * k) + l2;
            const int pr = pix / C2_PW, pc = pix - pr * C2_PW;
            const int vy = ty0 - 1 + pr, vx = tx0 - 1 + pc;
            const bool ok = pix < NPIX && vy >= 0 && vy < IH && vx >= 0 && vx < IW;
            a_off[k] = ok ? (int64_t)((vy >> hs) * (int)S.sy + (vx >> ws) * (int)S.sx + (seg ^ ((pc >> 3) & 1)) * 8)
                          : (int64_t)(zero_t - (reinterpret_cast<const T*>(S.ptr) + sbat[s2]));
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
    auto issue_piece = [&](const Cur& q, int buf, int i) {  // i = 0 .. KP + KW - 1: patch pieces, then weight pieces (wave-uniform conditions)
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

    // ---- fragment read addresses inside a buffer: patch row 0 of this wave per column offset (row rs: + rs * ROWB), weight row r ----
    int a_lane[3];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) a_lane[dx] = ((wave * MT) * C2_PW + dx + r) * 32 + ((h ^ (((dx + r) >> 3) & 1)) << 4);
    const int b_lane = A_BYTES + r * 32 + ((h ^ ((r >> 3) & 1)) << 4);

    f32x16 acc[NT][MT][1];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[nt][mt][0][j] = 0.f;
    Cur qi = {(int)blockIdx.x, 0, 0, 0, 0};
    if (total > 0) {
        tile_offsets(qi.tile, 0);
#pragma unroll
        for (int i = 0; i < KP + KW; ++i) issue_piece(qi, 0, i);
        advance(qi);
    }
    int ctile = blockIdx.x, cc = 0;  // compute cursor
    for (int it = 0; it < total; ++it) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // own pieces of chunk `it` landed; own reads of chunk it-1 done
        __builtin_amdgcn_s_barrier();                                // chunk `it` complete for every wave; buffer (it+1)&1 is free
        const bool more = it + 1 < total;
        int bo = (it & 1) * BUF;
        asm volatile("" : "+s"(bo));
        const char* const Bf = lds;
        int aa[3];
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) aa[dx] = a_lane[dx] + bo;
        const int bb = b_lane + bo;
        // Nine steps (column offset dx outer, tap row dy inner) of eight MFMAs.  Patch row rs of group dx is first used at dy = max(0, rs - 3):
        // rows 0, 1 of the NEXT group are read during step dy = 1, rows 2, 3 during dy = 2, rows 4 / 5 during the next dy = 0 / 1 -- at most eight
        // row fragments live; the two weight fragments of step st + 1 are read during step st.
        s16x8_t fa[3][MT + 2], fb[2][NT];
        auto a_read = [&](int dx, int rs) { return *reinterpret_cast<const s16x8_t*>(Bf + aa[dx] + rs * ROWB); };
        auto b_read = [&](int st, int set) {
            const int t = (st % 3) * 3 + st / 3;  // tap (dy, dx) of step st = 3 dx + dy
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) fb[set][nt] = *reinterpret_cast<const s16x8_t*>(Bf + bb + (t * BN + nt * 32) * 32);
        };
#pragma unroll
        for (int rs = 0; rs < 4; ++rs) fa[0][rs] = a_read(0, rs);
        b_read(0, 0);
#pragma unroll
        for (int st = 0; st < 9; ++st) {
            const int dx = st / 3, dy = st % 3;
            if (st + 1 < 9) b_read(st + 1, (st + 1) & 1);
            if (dy == 0) fa[dx][4] = a_read(dx, 4);
            if (dy == 1) fa[dx][5] = a_read(dx, 5);
            if (dx + 1 < 3) {
                if (dy == 1) { fa[dx + 1][0] = a_read(dx + 1, 0); fa[dx + 1][1] = a_read(dx + 1, 1); }
                if (dy == 2) { fa[dx + 1][2] = a_read(dx + 1, 2); fa[dx + 1][3] = a_read(dx + 1, 3); }
            }
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int i = 2 * st + half;
                if (i < KP + KW && more) issue_piece(qi, (it + 1) & 1, i);  // (wave-uniform)
#pragma unroll
                for (int m2 = 0; m2 < 2; ++m2) {
                    const int mt = 2 * half + m2;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[nt][mt][0] = H16<T>::mma(fb[st & 1][nt], fa[dx][mt + dy], acc[nt][mt][0]);
                }
            }
            // DS reads for the coming steps (the builtin wants literals): 3 5 4 | 3 5 4 | 3 3 0
            if (st == 8) __builtin_amdgcn_sched_group_barrier(0x100, 0, 0);
            else if (dy == 1 && dx < 2) __builtin_amdgcn_sched_group_barrier(0x100, 5, 0);
            else if (dy == 2) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
            else __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, MT * NT, 0);  // this step's MFMAs
        }
        if (more) advance(qi);
        if (++cc == nchunks) {  // tile finished: epilogue straight from the accumulators, then the next tile starts from zero
            cc = 0;
            int b, ty0, tx0;
            tile_coords(ctile, b, ty0, tx0);
            ctile += gridDim.x;
            const int cstride = p.out_cstride;
            const int x = tx0 + r;
            auto pixoff = [&](int mt) -> int64_t {
                const int y = ty0 + wave * MT + mt;
                if (!(y < p.OH && x < p.OW)) return (int64_t)-1;
                return (((int64_t)b * p.OH + y) * p.OW + x) * cstride;
            };
            auto pooloff = [&](int mt) -> int64_t {
                const int py = (ty0 + wave * MT + mt) >> 1, px = x >> 1, PH = p.OH >> 1, PW = p.OW >> 1;
                return (py < PH && px < PW) ? (((int64_t)b * PH + py) * PW + px) * cstride : (int64_t)-1;
            };
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                float bias[1][16];
                load_bias16_lds(lds_bias, 32 * nt, h, bias);
                if constexpr (POOL) epilogue_direct<T, MT, 1, decltype(pixoff), decltype(pooloff), -1, false, true>(p, acc[nt], bias, n0 + 32 * nt, lane, pixoff, pooloff);
                else epilogue_direct<T, MT, 1, decltype(pixoff), NoPool, -1, false, true>(p, acc[nt], bias, n0 + 32 * nt, lane, pixoff);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int j = 0; j < 16; ++j) acc[nt][mt][0][j] = 0.f;
            }
        }
    }
}

bool falnet_conv_dma2_applicable(const falnet_conv_t& p, int th) {
    return falnet_conv_dma_applicable(p, th) && p.out_layout == FALNET_OUT_NHWC;
}

int falnet_conv_dma2_launch(const falnet_conv_t& p, int flip, hipStream_t st, int th) {
    const int tiles_x = (p.OW + 31) / 32, tiles_y = (p.OH + th - 1) / th;
    const int ntiles = p.B * tiles_x * tiles_y;
    const int ny = (p.Cout + 63) / 64;
    int gx = (th == 32 ? 256 : 512) / ny;  // one persistent eight-wave workgroup per CU, or two four-wave ones
    if (gx < 1) gx = 1;
    if (gx > ntiles) gx = ntiles;
    const dim3 grid((unsigned)gx, (unsigned)ny);
    // start skew of the CUs' second residents: FALNET_DMA2_SKEW percent (experiment builds only; default 0) of a tile's time, taken as 5 200
    // cycles per 16-channel chunk; none when the grid has a single resident per CU.  Measured (profiles/r05_dma2_probes.txt): 25 % +-0, 50 % and
    // 100 % cost what they delay -- a workgroup alone on its CU does not run faster than beside its partner, so the two are not fighting over
    // the matrix pipe and de-phasing them buys nothing.
    static const int skew_pct = [] { const char* e = falnet_ab_env("FALNET_DMA2_SKEW"); return e ? atoi(e) : 0; }();
    int cin = 0;
    for (int s = 0; s < p.nsrc; ++s) cin += p.src[s].C;
    const int skew = (th == 16 && gx * ny > 256) ? (int)((int64_t)skew_pct * (cin / 16) * 5200 / 100 / 64) : 0;
#define DMA2_K(T, PL, NW) hipLaunchKernelGGL(HIP_KERNEL_NAME(conv3x3_dma2_kernel<T, PL, NW>), grid, dim3(NW * 64), 0, st, p, tiles_x, tiles_y, flip, ntiles, skew)
#define DMA2_L(T)                                              \
    do {                                                       \
        if (th == 32) {                                        \
            if (p.pool_out) DMA2_K(T, true, 8);                \
            else DMA2_K(T, false, 8);                          \
        } else {                                               \
            if (p.pool_out) DMA2_K(T, true, 4);                \
            else DMA2_K(T, false, 4);                          \
        }                                                      \
    } while (0)
    FALNET_DISPATCH_16(p.dtype, DMA2_L);
#undef DMA2_L
#undef DMA2_K
    FALNET_RETURN_LAUNCH();
}

// ============================================================================================================================
// conv3x3_dma16_kernel (falnet_conv2d variant 23, round 5): conv3x3_dma_kernel<T, 16, 8> on v_mfma_f32_16x16x32.
//
// profiles/r05_sq_counters.txt: on the 128- and 256-channel layers the dominant kernel runs at the clock the chip holds under bf16 MFMA load
// (the same launches on all-zero activations are 1.15-1.26x faster; cuts that need 5-6 % fewer shader cycles are not faster), and
// MI355X_MICROARCH.md (DVFS give-back, item 7) measures the 16x16x32 shape at 1.12-1.15x the FLOP/s of 32x32x16 at equal cycles per FLOP on
// random data, operands re-read from LDS: the chip holds a higher clock on it.  Same tile (16 x 32 positions x 64 channels, eight waves of two rows),
// same LDS-DMA machinery and 32-channel chunks (K = 32 = ONE 16x16x32 per tap), same bytes read per FLOP; what changes:
//   * a 32 x 32 (channels x positions) tile is 2 x 2 MFMAs (conv_epilogue.h: Acc16); the weight fragment of channel half ct reads row
//     m16_row_channel(lane & 15), so that Acc16::to32 lands the results in epilogue_direct's layout with one v_permlane16_swap per register;
//   * lane = (row / position lane & 15, K block lane >> 4) reads 16 B of a 64-B row: the 16-B segments are exchanged in pairs (seg ^ 2) when bit 2
//     of the patch COLUMN (of the weight row) is set -- on the DMA's source address -- which makes every ds_read_b128 of this lane pattern
//     conflict-free for the column offsets 0..2 and both 16-position halves (found by exhaustive search over the 4-entry tables; the (row >> 2) & 3
//     swizzle of the 32x32x16 kernel is 2-way conflicted under this pattern), and independent of the patch row: rows are immediate offsets;
//   * nine steps (column offset outer, tap row inner) of 16 MFMAs; a pixel fragment serves the two output rows it meets, as in variant 13.
// <T, POOL, 16, 8, PLANAR>: variant 23 (PLANAR: the planar-f32 output form of the logits launch, an instantiation of its own -- its sixteen 64-bit
// store addresses cost ~60 registers).  <T, false, 4, 4, false> / <T, false, 8, 8, false>: variants 24 / 25, the 4 x 32 / 8 x 32 tiles of variants
// 17 / 20 (one row per wave) for the maps of levels 4 / 3.
template <typename T, bool POOL, int TH, int NWAVES, bool PLANAR>
__global__ __launch_bounds__(NWAVES * 64, NWAVES == 8 ? 2 : 1) void conv3x3_dma16_kernel(const falnet_conv_t p, int tiles_x, int tiles_y, int flip, int ntiles) {
    constexpr int MT = TH / NWAVES, BN = 64, NT = BN / 32;
    static_assert(sizeof(T) == 2 && (MT == 1 || MT == 2) && (MT == 2 || !POOL), "16-bit operands; one or two rows per wave; the fused pool needs row pairs");
    constexpr int KCV = 32;
    constexpr int NPIX = (TH + 2) * CD_PW;
    constexpr int A_PIECES = (NPIX + 15) / 16, B_PIECES = 9 * BN / 16, NPIECES = A_PIECES + B_PIECES;
    constexpr int A_BYTES = A_PIECES * 1024, BUF = NPIECES * 1024;
    constexpr int ROWB = CD_PW * 64;  // bytes between patch rows
    __shared__ __attribute__((aligned(1024))) char lds[2 * BUF];
    __shared__ __attribute__((aligned(16))) float lds_bias[BN];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds_base = (unsigned)(unsigned long)(cd_lptr_t)lds;
    const int r = lane & 31, h = lane >> 5;  // epilogue layout (position, lane half)
    const int lp = lane & 15, lg = lane >> 4;  // MFMA operand layout (row / position, K block)
    const int n0 = blockIdx.y * BN;
    const char* const zero_page = reinterpret_cast<const char*>(g_cd_zero);
    stage_bias_lds(p, n0, BN, lds_bias);

    const int nsrc = p.nsrc, IH = p.IH, IW = p.IW;
    const int C0 = p.src[0].C, C1 = nsrc > 1 ? p.src[1].C : 0;
    const int nchunks = (C0 + C1) / KCV;
    int my_tiles = 0;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) ++my_tiles;
    const int total = my_tiles * nchunks;

    constexpr int KP = (A_PIECES + NWAVES - 1) / NWAVES, KW = (B_PIECES + NWAVES - 1) / NWAVES;
    const int l4 = lane >> 2, segpos = lane & 3;
    const T* const wptr = reinterpret_cast<const T*>(p.weight);
    const T* const zero_t = reinterpret_cast<const T*>(zero_page);
    int64_t w_off[KW];
#pragma unroll
    for (int k = 0; k < KW; ++k) {
        const int wid = wave + NWAVES * k;
        const int tap = wid >> 2, co = n0 + ((wid & 3) << 4) + l4;  // four 16-row pieces per tap tile
        const int gseg = segpos ^ (((l4 >> 2) & 1) << 1);            // (row >> 2) & 1: pieces start at multiples of 16 rows
        w_off[k] = (wid < B_PIECES && co < p.w_rows) ? (int64_t)(co * p.w_taps + (flip ? 8 - tap : tap)) * p.cin_total + gseg * 8 : (int64_t)(zero_t - wptr);
    }
    // ---- patch piece geometry.  A lane's patch pixel (pr, pc) is the same in every tile: its element offset is (tile origin, wave-uniform) +
    // (lane constant per source), and only the in-image test depends on the tile.  Tiles advance by gridDim.x: the step is decomposed ONCE into
    // (images, tile rows, tile columns) and the cursors carry (b, ty, tx) forward with two compare-and-wrap steps -- no division, no 64-bit
    // multiply per tile (in-kernel stamps, 64 -> 64 channels at 128 x 256: 2 100 of a tile's 22 700 cycles went into re-deriving these). ----
    const T* sptr[2] = {reinterpret_cast<const T*>(p.src[0].ptr), reinterpret_cast<const T*>(nsrc > 1 ? p.src[1].ptr : p.src[0].ptr)};
    struct TPos { int b, ty, tx; };
    TPos step_;
    {
        const int g = (int)gridDim.x;
        step_.tx = g % tiles_x;
        const int q = g / tiles_x;
        step_.ty = q % tiles_y;
        step_.b = q / tiles_y;
    }
    auto pos_of = [&](int tile) {
        TPos t;
        t.tx = tile % tiles_x;
        const int q = tile / tiles_x;
        t.ty = q % tiles_y;
        t.b = q / tiles_y;
        return t;
    };
    auto pos_next = [&](TPos& t) {
        t.tx += step_.tx;
        if (t.tx >= tiles_x) { t.tx -= tiles_x; ++t.ty; }
        t.ty += step_.ty;
        if (t.ty >= tiles_y) { t.ty -= tiles_y; ++t.b; }
        t.b += step_.b;
    };
    int pr_[KP], pc_[KP], a_lc[2][KP];  // patch row / column of piece k's pixel; its offset inside source s2 relative to the tile origin
#pragma unroll
    for (int k = 0; k < KP; ++k) {
        const int pix = 16 * (wave + NWAVES * k) + l4;
        const int pr = pix / CD_PW, pc = pix - pr * CD_PW;
        pr_[k] = pix < NPIX ? pr : -(1 << 20);  // (a piece beyond the patch never passes the in-image test)
        pc_[k] = pc;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const falnet_src_t& S = s2 == 0 ? p.src[0] : p.src[(nsrc > 1) ? 1 : 0];
            const int hs = S.H != IH ? 1 : 0, ws = S.W != IW ? 1 : 0;
            a_lc[s2][k] = ((pr - 1) >> hs) * (int)S.sy + ((pc - 1) >> ws) * (int)S.sx + (segpos ^ (((pc >> 2) & 1) << 1)) * 8;
        }
    }
    int a_off[KP];      // element offset of piece k inside the sample (valid when bit k of a_ok is set; the zero page otherwise)
    unsigned a_ok = 0;
    const T* a_base = sptr[0];  // source pointer + image offset + tile origin of the issue cursor
    auto tile_offsets = [&](const TPos& t, int s2) {
        const falnet_src_t& S = s2 == 0 ? p.src[0] : p.src[1];
        const int hs = S.H != IH ? 1 : 0, ws = S.W != IW ? 1 : 0;
        const int ty0 = t.ty * TH, tx0 = t.tx * 32;
        a_base = sptr[s2] + ((int64_t)t.b * S.sb + (int64_t)((ty0 >> hs) * (int)S.sy + (tx0 >> ws) * (int)S.sx));
        a_ok = 0;
#pragma unroll
        for (int k = 0; k < KP; ++k) {
            const int vy = ty0 - 1 + pr_[k], vx = tx0 - 1 + pc_[k];
            a_ok |= (vy >= 0 && vy < IH && vx >= 0 && vx < IW) ? 1u << k : 0u;
            a_off[k] = a_lc[s2][k];
        }
    };
    struct Cur { int tile, c, s, c0, kofs; TPos t; };
    auto advance = [&](Cur& q) {
        if (++q.c == nchunks) {
            q.c = 0; q.s = 0; q.c0 = 0; q.kofs = 0;
            q.tile += gridDim.x;
            pos_next(q.t);
            if (q.tile < ntiles) tile_offsets(q.t, 0);
            return;
        }
        q.c0 += KCV;
        q.kofs += KCV;
        if (q.s == 0 && q.c0 >= C0) {
            q.s = 1;
            q.c0 = 0;
            tile_offsets(q.t, 1);
        }
    };
    auto issue_piece = [&](const Cur& q, int buf, int i) {
        const unsigned dst0 = lds_base + buf * BUF;
        if (i < KP) {
            const int id = wave + NWAVES * i;
            const T* src = ((a_ok >> i) & 1) ? a_base + q.c0 + a_off[i] : zero_t;
            if (id < A_PIECES) cd_glds16(src, dst0 + id * 1024);
        } else {
            const int wid = wave + NWAVES * (i - KP);
            if (wid < B_PIECES) cd_glds16(wptr + q.kofs + w_off[i - KP], dst0 + A_BYTES + wid * 1024);
        }
    };

    // ---- fragment read addresses: position 16 pt + lp of patch row 0 of this wave per (column offset, half); weight row m16_row_channel(lp) ----
    int a_lane[3][2];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) {
            const int col = dx + 16 * pt + lp;
            a_lane[dx][pt] = ((wave * MT) * CD_PW + col) * 64 + ((lg ^ (((col >> 2) & 1) << 1)) << 4);
        }
    const int wrow = m16_row_channel(lp);
    const int b_lane = A_BYTES + wrow * 64 + ((lg ^ (((wrow >> 2) & 1) << 1)) << 4);

    Acc16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt].zero();
    Cur qi = {(int)blockIdx.x, 0, 0, 0, 0, pos_of((int)blockIdx.x)};
    TPos ct = qi.t;  // the compute cursor's tile
    if (total > 0) {
        tile_offsets(qi.t, 0);
#pragma unroll
        for (int i = 0; i < KP + KW; ++i) issue_piece(qi, 0, i);
        advance(qi);
    }
    int cc = 0;
#ifdef FALNET_CD_STAMPS  // profiling build (tools/cd_stamps.py): s_memtime at the phase boundaries, every wave of workgroup (0, 0) -> p.splitk_ws
    unsigned long long* stamp_out = reinterpret_cast<unsigned long long*>(p.splitk_ws);
    int stamp_i = 0;
#define CD16_STAMP()                                                                                 \
    do {                                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                           \
        unsigned long long t_;                                                                       \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                  \
        __builtin_amdgcn_sched_barrier(0);                                                           \
        if (stamp_out && blockIdx.x == 0 && blockIdx.y == 0 && lane == 0 && stamp_i < 256) stamp_out[wave * 256 + stamp_i] = t_; \
        ++stamp_i;                                                                                   \
    } while (0)
#else
#define CD16_STAMP() do {} while (0)
#endif
#if defined(FALNET_CD_STAMPS) && FALNET_CD_STAMPS == 2  // second stamp set: inside the tile end (advance / slice 0 / slice 1)
#define CD16_A() do {} while (0)
#define CD16_B() CD16_STAMP()
#else
#define CD16_A() CD16_STAMP()
#define CD16_B() do {} while (0)
#endif
    for (int it = 0; it < total; ++it) {
        CD16_STAMP();  // 0: loop top
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        CD16_A();  // 1: own DMA landed (and the previous tile's stores)
        __builtin_amdgcn_s_barrier();
        CD16_STAMP();  // 2 (B: 1): barrier passed
        CD16_A();  // 3
        const bool more = it + 1 < total;
        int bo = (it & 1) * BUF;
        asm volatile("" : "+s"(bo));
        const char* const Bf = lds;
        int aa[3][2];
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
#pragma unroll
            for (int pt = 0; pt < 2; ++pt) aa[dx][pt] = a_lane[dx][pt] + bo;
        const int bb = b_lane + bo;
        // step st = 3 dx + dy: tap (dy, dx); patch row rs of group dx serves output row mt under dy = rs - mt.  Rows 0 .. MT-1 of the NEXT group are
        // read during step dy = 2, row MT during dy = 0, row MT + 1 during dy = 1; the four weight fragments of step st + 1 during step st.
        s16x8_t fa[3][MT + 2][2], fb[2][NT][2];
        auto a_read = [&](int dx, int rs, int pt) { return *reinterpret_cast<const s16x8_t*>(Bf + aa[dx][pt] + rs * ROWB); };
        auto b_read = [&](int st, int set) {
            const int t = (st % 3) * 3 + st / 3;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) fb[set][nt][ct] = *reinterpret_cast<const s16x8_t*>(Bf + bb + (t * BN + nt * 32 + ct * 16) * 64);
        };
        static_assert(KP + KW <= 18, "two DMA issue slots per step");
#pragma unroll
        for (int rs = 0; rs < MT; ++rs)
#pragma unroll
            for (int pt = 0; pt < 2; ++pt) fa[0][rs][pt] = a_read(0, rs, pt);
        b_read(0, 0);
#pragma unroll
        for (int st = 0; st < 9; ++st) {
            const int dx = st / 3, dy = st % 3;
            if (st + 1 < 9) b_read(st + 1, (st + 1) & 1);
            if (dy < 2) { fa[dx][MT + dy][0] = a_read(dx, MT + dy, 0); fa[dx][MT + dy][1] = a_read(dx, MT + dy, 1); }
            if (dy == 2 && dx + 1 < 3) {
#pragma unroll
                for (int rs = 0; rs < MT; ++rs)
#pragma unroll
                    for (int pt = 0; pt < 2; ++pt) fa[dx + 1][rs][pt] = a_read(dx + 1, rs, pt);
            }
#pragma unroll
            for (int half = 0; half < 2; ++half) {  // two DMA issue slots per step, each in front of half of the step's MFMAs
                const int i = 2 * st + half;
                if (i < KP + KW && more) issue_piece(qi, (it + 1) & 1, i);  // (wave-uniform)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = (MT == 2 ? 0 : half); nt < (MT == 2 ? NT : half + 1); ++nt) {
                        if (MT == 2 && mt != half) continue;
#pragma unroll
                        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                            for (int pt = 0; pt < 2; ++pt)
                                acc[mt][nt].t[ct][pt] = H16<T>::mma16(fb[st & 1][nt][ct], fa[dx][mt + dy][pt], acc[mt][nt].t[ct][pt]);
                    }
            }
            // DS reads for the coming steps (the builtin wants literals): two rows per wave 6 6 8 | 6 6 8 | 6 6 0, one row 6 6 6 | 6 6 6 | 6 6 0
            if (st == 8) __builtin_amdgcn_sched_group_barrier(0x100, 0, 0);
            else if (dy == 2 && MT == 2) __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
            else __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
            if constexpr (MT == 2) __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
            else __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
        }
        CD16_STAMP();  // 4 (B: 2): MFMAs issued
        if (more) advance(qi);
        CD16_B();  // B 3: issue cursor advanced
        if (++cc == nchunks) {
            cc = 0;
            const int b = ct.b, ty0 = ct.ty * TH, tx0 = ct.tx * 32;
            pos_next(ct);
            const int cstride = p.out_cstride;
            const int x = tx0 + r;
            // (wave-uniform 64-bit origin of the tile + a 32-bit lane part: no 64-bit vector multiply per slab)
            const int64_t obase = PLANAR ? ((int64_t)b * p.Cout * p.OH + ty0) * p.OW + tx0 : (((int64_t)b * p.OH + ty0) * p.OW + tx0) * cstride;
            auto pixoff = [&](int mt) -> int64_t {
                const int y = ty0 + wave * MT + mt;
                if (!(y < p.OH && x < p.OW)) return (int64_t)-1;
                if constexpr (PLANAR) return obase + ((wave * MT + mt) * p.OW + r);  // planar f32 [B][Cout][OH][OW]: offset of channel 0
                else return obase + ((wave * MT + mt) * p.OW + r) * cstride;
            };
            auto pooloff = [&](int mt) -> int64_t {
                const int py = (ty0 + wave * MT + mt) >> 1, px = x >> 1, PH = p.OH >> 1, PW = p.OW >> 1;
                return (py < PH && px < PW) ? (((int64_t)b * PH + py) * PW + px) * cstride : (int64_t)-1;
            };
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                float bias[1][16];
                load_bias16_lds(lds_bias, 32 * nt, h, bias);
                f32x16 v[MT][1];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    acc[mt][nt].to32(v[mt][0]);
                    acc[mt][nt].zero();
                }
                if constexpr (POOL) epilogue_direct<T, MT, 1, decltype(pixoff), decltype(pooloff), -1, false, true>(p, v, bias, n0 + 32 * nt, lane, pixoff, pooloff);
                else epilogue_direct<T, MT, 1, decltype(pixoff), NoPool, (PLANAR ? -1 : FALNET_DMA16_EPI_AHEAD), false, !PLANAR, PLANAR>(p, v, bias, n0 + 32 * nt, lane, pixoff);
                if (nt == 0) CD16_B();  // B 4: slice 0 done
            }
        } else {
            CD16_B();
        }
        CD16_STAMP();  // 5: (epilogue) done
    }
#undef CD16_STAMP
#undef CD16_A
#undef CD16_B
}

int falnet_conv_dma16_launch(const falnet_conv_t& p, int flip, hipStream_t st, int th) {
    const int tiles_x = (p.OW + 31) / 32, tiles_y = (p.OH + th - 1) / th;
    const int ntiles = p.B * tiles_x * tiles_y;
    const int ny = (p.Cout + 63) / 64;
    int gx = 256 / ny;  // one persistent workgroup per CU
    if (gx < 1) gx = 1;
    if (gx > ntiles) gx = ntiles;
    const dim3 grid((unsigned)gx, (unsigned)ny);
    const bool planar = p.out_layout == FALNET_OUT_PLANAR_F32;
#define DMA16_K(T, PL, TH_, NW, PN) hipLaunchKernelGGL(HIP_KERNEL_NAME(conv3x3_dma16_kernel<T, PL, TH_, NW, PN>), grid, dim3(NW * 64), 0, st, p, tiles_x, tiles_y, flip, ntiles)
#define DMA16_L(T)                                        \
    do {                                                  \
        if (th == 4) DMA16_K(T, false, 4, 4, false);      \
        else if (th == 8) DMA16_K(T, false, 8, 8, false); \
        else if (planar) DMA16_K(T, false, 16, 8, true);  \
        else if (p.pool_out) DMA16_K(T, true, 16, 8, false); \
        else DMA16_K(T, false, 16, 8, false);             \
    } while (0)
    FALNET_DISPATCH_16(p.dtype, DMA16_L);
#undef DMA16_L
#undef DMA16_K
    FALNET_RETURN_LAUNCH();
}

// ============================================================================================================================
// Data gradient of a `deconv` layer (nearest 2x upsampling + 3x3 convolution, models/FAL_netB.py:52-58) ON THE LOW-RESOLUTION GRID
// (falnet_conv2d variant 26, round 5).
//
// y[Y, X] = sum_k W[k] up(x)[Y + kh - 1, X + kw - 1], up(x)[r, c] = x[r >> 1, c >> 1].  The high-resolution form (3x3 data gradient at 2H x 2W,
// 2x2 sums fused into its epilogue: 36 MACs per low-resolution position and channel pair) adds taps that meet the SAME upstream pixel: per
// axis, input position i receives  W[2] g[2i-1] + (W[1] + W[2]) g[2i] + (W[0] + W[1]) g[2i+1] + W[0] g[2i+2]  -- a 4x4 / stride-2 convolution of the
// upstream gradient, 16 MACs (2.25x fewer).  Grouped by PAIRS of upstream rows / columns (pair u = rows 2u-1, 2u; columns alike) it is a dense
// 2x2-tap stride-1 convolution over "pair pixels" of 4 C channels: K index = e 2C + f C + c for row parity e, column parity f (falnet_pack_up2_batched
// writes the summed weights wdd[ci][2 du + dv][K]).  The kernel is conv3x3_dma16_kernel's machinery with four tap steps per 32-channel chunk:
//   * a chunk belongs to one (e, f): its patch piece of pair pixel (u, v) is the 64 B at upstream pixel (2u - 1 + e, 2v - 1 + f) -- the LDS-DMA
//     gathers the strided view directly, pixels outside the upstream map (row -1 of pair 0, row 2H of pair H) read the zero page;
//   * patch (16 + 1) x (32 + 1) pair pixels (CD_PW columns kept for the segment swizzle), weights 4 x 64 rows: 53 KB per buffer;
//   * per chunk and wave 4 steps x 16 MFMAs on six + six pixel fragments and four weight fragments per step; the epilogue is the plain NHWC one
//     (residual addend, activation gradient of the layer's input).
template <typename T>
__global__ __launch_bounds__(512, 2) void conv2x2_up2d_dma16_kernel(const falnet_conv_t p, int tiles_x, int tiles_y, int ntiles) {
    constexpr int TH = 16, NWAVES = 8, MT = 2, BN = 64, NT = 2, KCV = 32;
    constexpr int NPIX = (TH + 1) * CD_PW;
    constexpr int A_PIECES = (NPIX + 15) / 16, B_PIECES = 4 * BN / 16, NPIECES = A_PIECES + B_PIECES;
    constexpr int A_BYTES = A_PIECES * 1024, BUF = NPIECES * 1024;
    constexpr int ROWB = CD_PW * 64;
    __shared__ __attribute__((aligned(1024))) char lds[2 * BUF];
    __shared__ __attribute__((aligned(16))) float lds_bias[BN];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds_base = (unsigned)(unsigned long)(cd_lptr_t)lds;
    const int r = lane & 31;
    const int lp = lane & 15, lg = lane >> 4;
    const int n0 = blockIdx.y * BN;
    const char* const zero_page = reinterpret_cast<const char*>(g_cd_zero);
    stage_bias_lds(p, n0, BN, lds_bias);

    const falnet_src_t& S = p.src[0];          // the upstream gradient at 2 OH x 2 OW
    const int GH = S.H, GW = S.W, CP = S.C;     // its map and (padded) channel count
    const int cpc = CP / KCV;                   // chunks per (e, f)
    const int nchunks = 4 * cpc;
    int my_tiles = 0;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) ++my_tiles;
    const int total = my_tiles * nchunks;

    constexpr int KP = (A_PIECES + NWAVES - 1) / NWAVES, KW = (B_PIECES + NWAVES - 1) / NWAVES;
    static_assert(KP + KW <= 8, "two DMA issue slots per step");
    const int l4 = lane >> 2, segpos = lane & 3;
    const T* const wptr = reinterpret_cast<const T*>(p.weight);
    const T* const zero_t = reinterpret_cast<const T*>(zero_page);
    const T* const sptr = reinterpret_cast<const T*>(S.ptr);
    int64_t w_off[KW];
#pragma unroll
    for (int k = 0; k < KW; ++k) {
        const int wid = wave + NWAVES * k;
        const int tap = wid >> 2, co = n0 + ((wid & 3) << 4) + l4;
        const int gseg = segpos ^ (((l4 >> 2) & 1) << 1);
        w_off[k] = (wid < B_PIECES && co < p.w_rows) ? (int64_t)(co * 4 + tap) * p.cin_total + gseg * 8 : (int64_t)(zero_t - wptr);
    }
    struct TPos { int b, ty, tx; };
    TPos step_;
    {
        const int g = (int)gridDim.x;
        step_.tx = g % tiles_x;
        const int q = g / tiles_x;
        step_.ty = q % tiles_y;
        step_.b = q / tiles_y;
    }
    auto pos_of = [&](int tile) {
        TPos t;
        t.tx = tile % tiles_x;
        const int q = tile / tiles_x;
        t.ty = q % tiles_y;
        t.b = q / tiles_y;
        return t;
    };
    auto pos_next = [&](TPos& t) {
        t.tx += step_.tx;
        if (t.tx >= tiles_x) { t.tx -= tiles_x; ++t.ty; }
        t.ty += step_.ty;
        if (t.ty >= tiles_y) { t.ty -= tiles_y; ++t.b; }
        t.b += step_.b;
    };
    int pr_[KP], pc_[KP], a_off[KP];  // pair-pixel row / column of piece k inside the patch; its element offset from the (tile, e, f) origin
#pragma unroll
    for (int k = 0; k < KP; ++k) {
        const int pix = 16 * (wave + NWAVES * k) + l4;
        const int pr = pix / CD_PW, pc = pix - pr * CD_PW;
        pr_[k] = pix < NPIX ? pr : (1 << 20);  // (a piece beyond the patch never passes the in-map test)
        pc_[k] = pc;
        a_off[k] = 2 * pr * (int)S.sy + 2 * pc * (int)S.sx + (segpos ^ (((pc >> 2) & 1) << 1)) * 8;
    }
    unsigned a_ok = 0;
    const T* a_base = sptr;
    auto tile_offsets = [&](const TPos& t, int e, int f) {  // origin = upstream pixel (2 ty0 - 1 + e, 2 tx0 - 1 + f) of image b (only ever dereferenced with a piece that passed the test)
        const int y0 = 2 * t.ty * TH - 1 + e, x0 = 2 * t.tx * 32 - 1 + f;
        a_base = sptr + ((int64_t)t.b * S.sb + (int64_t)y0 * (int)S.sy + (int64_t)x0 * (int)S.sx);
        a_ok = 0;
#pragma unroll
        for (int k = 0; k < KP; ++k) {
            const int vy = y0 + 2 * pr_[k], vx = x0 + 2 * pc_[k];
            a_ok |= (vy >= 0 && vy < GH && vx >= 0 && vx < GW) ? 1u << k : 0u;
        }
    };
    struct Cur { int tile, c, cin, ef, kofs; TPos t; };  // chunk of the tile, chunk inside its (e, f), (e, f) index, weight K offset
    auto advance = [&](Cur& q) {
        q.kofs += KCV;
        ++q.c;
        if (++q.cin == cpc) {
            q.cin = 0;
            if (++q.ef == 4) {
                q.ef = 0; q.c = 0; q.kofs = 0;
                q.tile += gridDim.x;
                pos_next(q.t);
                if (q.tile >= ntiles) return;
            }
            tile_offsets(q.t, q.ef >> 1, q.ef & 1);
        }
    };
    auto issue_piece = [&](const Cur& q, int buf, int i) {
        const unsigned dst0 = lds_base + buf * BUF;
        if (i < KP) {
            const int id = wave + NWAVES * i;
            const T* src = ((a_ok >> i) & 1) ? a_base + q.cin * KCV + a_off[i] : zero_t;
            if (id < A_PIECES) cd_glds16(src, dst0 + id * 1024);
        } else {
            const int wid = wave + NWAVES * (i - KP);
            if (wid < B_PIECES) cd_glds16(wptr + q.kofs + w_off[i - KP], dst0 + A_BYTES + wid * 1024);
        }
    };

    int a_lane[2][2];
#pragma unroll
    for (int dv = 0; dv < 2; ++dv)
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) {
            const int col = dv + 16 * pt + lp;
            a_lane[dv][pt] = ((wave * MT) * CD_PW + col) * 64 + ((lg ^ (((col >> 2) & 1) << 1)) << 4);
        }
    const int wrow = m16_row_channel(lp);
    const int b_lane = A_BYTES + wrow * 64 + ((lg ^ (((wrow >> 2) & 1) << 1)) << 4);

    Acc16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt].zero();
    Cur qi = {(int)blockIdx.x, 0, 0, 0, 0, pos_of((int)blockIdx.x)};
    TPos ct = qi.t;
    if (total > 0) {
        tile_offsets(qi.t, 0, 0);
#pragma unroll
        for (int i = 0; i < KP + KW; ++i) issue_piece(qi, 0, i);
        advance(qi);
    }
    int cc = 0;
    for (int it = 0; it < total; ++it) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const bool more = it + 1 < total;
        int bo = (it & 1) * BUF;
        asm volatile("" : "+s"(bo));
        const char* const Bf = lds;
        int aa[2][2];
#pragma unroll
        for (int dv = 0; dv < 2; ++dv)
#pragma unroll
            for (int pt = 0; pt < 2; ++pt) aa[dv][pt] = a_lane[dv][pt] + bo;
        const int bb = b_lane + bo;
        // step st = 2 dv + du: tap (du, dv) = weight tile 2 du + dv; patch row rs of column group dv serves output row mt under du = rs - mt
        s16x8_t fa[2][MT + 1][2], fb[2][NT][2];
        auto a_read = [&](int dv, int rs, int pt) { return *reinterpret_cast<const s16x8_t*>(Bf + aa[dv][pt] + rs * ROWB); };
        auto b_read = [&](int st, int set) {
            const int t = (st & 1) * 2 + (st >> 1);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int ct2 = 0; ct2 < 2; ++ct2) fb[set][nt][ct2] = *reinterpret_cast<const s16x8_t*>(Bf + bb + (t * BN + nt * 32 + ct2 * 16) * 64);
        };
#pragma unroll
        for (int rs = 0; rs < MT; ++rs)
#pragma unroll
            for (int pt = 0; pt < 2; ++pt) fa[0][rs][pt] = a_read(0, rs, pt);
        b_read(0, 0);
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            const int dv = st >> 1, du = st & 1;
            if (st + 1 < 4) b_read(st + 1, (st + 1) & 1);
            if (du == 0) { fa[dv][MT][0] = a_read(dv, MT, 0); fa[dv][MT][1] = a_read(dv, MT, 1); }
            if (du == 1 && dv == 0) {
#pragma unroll
                for (int rs = 0; rs < MT; ++rs)
#pragma unroll
                    for (int pt = 0; pt < 2; ++pt) fa[1][rs][pt] = a_read(1, rs, pt);
            }
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int i = 2 * st + half;
                if (i < KP + KW && more) issue_piece(qi, (it + 1) & 1, i);  // (wave-uniform)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int ct2 = 0; ct2 < 2; ++ct2)
#pragma unroll
                        for (int pt = 0; pt < 2; ++pt)
                            acc[half][nt].t[ct2][pt] = H16<T>::mma16(fb[st & 1][nt][ct2], fa[dv][half + du][pt], acc[half][nt].t[ct2][pt]);
            }
            if (st == 3) __builtin_amdgcn_sched_group_barrier(0x100, 0, 0);
            else if (st == 1) __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
            else __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
        }
        if (more) advance(qi);
        if (++cc == nchunks) {
            cc = 0;
            const int b = ct.b, ty0 = ct.ty * TH, tx0 = ct.tx * 32;
            pos_next(ct);
            const int cstride = p.out_cstride;
            const int x = tx0 + r;
            const int64_t obase = (((int64_t)b * p.OH + ty0) * p.OW + tx0) * cstride;
            auto pixoff = [&](int mt) -> int64_t {
                const int y = ty0 + wave * MT + mt;
                if (!(y < p.OH && x < p.OW)) return (int64_t)-1;
                return obase + ((wave * MT + mt) * p.OW + r) * cstride;
            };
            const int h = lane >> 5;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                float bias[1][16];
                load_bias16_lds(lds_bias, 32 * nt, h, bias);
                f32x16 v[MT][1];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    acc[mt][nt].to32(v[mt][0]);
                    acc[mt][nt].zero();
                }
                epilogue_direct<T, MT, 1, decltype(pixoff), NoPool, FALNET_DMA16_EPI_AHEAD, false, true>(p, v, bias, n0 + 32 * nt, lane, pixoff);
            }
        }
    }
}

// variant 26: one upstream-gradient source at exactly twice the output map, 2x2-tap weights [w_rows][4][4 C] (falnet_pack_up2_batched: wdd)
bool falnet_conv_up2d_applicable(const falnet_conv_t& p) {
    if (p.dtype != FALNET_BF16 && p.dtype != FALNET_F16) return false;
    if (p.nsrc != 1 || p.w_taps != 4 || p.out_layout != FALNET_OUT_NHWC || p.pool_out || p.ksplit > 1 || !p.out || p.weight_up2) return false;
    const falnet_src_t& S = p.src[0];
    if (S.C % 32 || S.C <= 0 || p.cin_total != 4 * S.C) return false;
    if (S.H != 2 * p.OH || S.W != 2 * p.OW || p.OH < 16 || p.OW < 32) return false;
    if ((int64_t)S.H * S.sy >= (1ll << 31) || (int64_t)p.w_rows * 4 * p.cin_total >= (1ll << 31) || (S.C + 64) * 2 > CD_ZERO_BYTES) return false;
    return true;
}

int falnet_conv_up2d_launch(const falnet_conv_t& p, hipStream_t st) {
    const int tiles_x = (p.OW + 31) / 32, tiles_y = (p.OH + 15) / 16;
    const int ntiles = p.B * tiles_x * tiles_y;
    const int ny = (p.Cout + 63) / 64;
    int gx = 256 / ny;
    if (gx < 1) gx = 1;
    if (gx > ntiles) gx = ntiles;
    const dim3 grid((unsigned)gx, (unsigned)ny);
    if (p.dtype == FALNET_F16) hipLaunchKernelGGL(HIP_KERNEL_NAME(conv2x2_up2d_dma16_kernel<f16_t>), grid, dim3(512), 0, st, p, tiles_x, tiles_y, ntiles);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(conv2x2_up2d_dma16_kernel<bf16_t>), grid, dim3(512), 0, st, p, tiles_x, tiles_y, ntiles);
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

// MT = rows per wave: 2 -> 16 x 32 gout positions per tile; 1 -> 8 x 32 (round 4): conv3 / conv4 have 32 / 8 sixteen-row tiles per
// channel block at B = 8 -- 128 / 64 workgroups for 256 CUs -- and twice as many eight-row ones.
template <typename T, int MT = 2>
__global__ __launch_bounds__(512) void conv3x3_s2d_dma_kernel(const falnet_conv_t p, int tiles_x, int tiles_y, int ntiles, int GH, int GW) {
    constexpr int NWAVES = 8, BN = 32, TH = NWAVES * MT;
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
    __shared__ __attribute__((aligned(16))) float lds_bias[BN];
    stage_bias_lds(p, n0, BN, lds_bias);  // (read by the tile epilogues: at least one barrier later)
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
            load_bias16_lds(lds_bias, 0, h, bias);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int py = c >> 1, px = c & 1;
                const int x = 2 * (tx0 + r) + px;
                auto pixoff = [&](int mt) -> int64_t {
                    const int y = 2 * (ty0 + wave * MT + mt) + py;
                    if (!(y < p.OH && x < p.OW)) return (int64_t)-1;
                    return (((int64_t)b * p.OH + y) * p.OW + x) * cstride;
                };
                epilogue_direct<T, MT, 1, decltype(pixoff), NoPool, FALNET_DMA_EPI_AHEAD, false, true>(p, acc[c], bias, n0, lane, pixoff);
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
    const int ny = (p.Cout + 31) / 32;
    const int tiles_x = (GW + 31) / 32;
    static const int force_mt = [] { const char* e = falnet_ab_env("FALNET_S2D_MT"); return e ? atoi(e) : 0; }();
    // eight-row tiles when sixteen-row ones leave CUs idle (fewer workgroups than CUs)
    const bool small = force_mt ? force_mt == 1 : p.B * tiles_x * ((GH + 15) / 16) * ny < 256;
    const int th = small ? 8 : 16;
    const int tiles_y = (GH + th - 1) / th;
    const int ntiles = p.B * tiles_x * tiles_y;
    int gx = 256 / ny;
    if (gx < 1) gx = 1;
    if (gx > ntiles) gx = ntiles;
    const dim3 grid((unsigned)gx, (unsigned)ny);
#define S2D_L(T)                                                                                                                                     \
    do {                                                                                                                                             \
        if (small) hipLaunchKernelGGL(HIP_KERNEL_NAME(conv3x3_s2d_dma_kernel<T, 1>), grid, dim3(512), 0, st, p, tiles_x, tiles_y, ntiles, GH, GW);  \
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(conv3x3_s2d_dma_kernel<T, 2>), grid, dim3(512), 0, st, p, tiles_x, tiles_y, ntiles, GH, GW);        \
    } while (0)
    FALNET_DISPATCH_16(p.dtype, S2D_L);
#undef S2D_L
    FALNET_RETURN_LAUNCH();
}

// ============================================================================================================================
// `deconv` forward (models/FAL_netB.py:52-58: F.interpolate(scale_factor=2, nearest) -> 3x3 conv) in SUB-PIXEL form (variant 18).
//
// A 3x3 convolution over a 2x nearest-upsampled map U(Y, X) = L(Y >> 1, X >> 1) reads, for output pixel (2i + py, 2j + px), only a 2 x 2
// neighbourhood of the low-resolution map L: rows i + py - 1 + a, columns j + px - 1 + b (a, b in {0, 1}), because two of the three taps of a
// row always fall on the same low-resolution pixel.  With the weights of coinciding taps summed,
//     Weff[py][a] = { py = 0: (W[0], W[1] + W[2]);  py = 1: (W[0] + W[1], W[2]) }   (rows; columns alike),
// the layer is four 2x2 convolutions of L, one per output parity class: 16 tap-MACs per output quad instead of 36 -- the same values (the
// sums are formed in f32 before the weights are rounded to the compute type), 2.25x fewer MFMAs, and the upsampled map is never addressed.
// Structure: the four-class stride-2 data-gradient kernel above.  A workgroup owns 16 x 32 LOW-resolution positions (= 32 x 64 outputs) x 32
// output channels; per 32-channel chunk the (16+2) x (32+2) patch of L and the sixteen 32 x 32 (class, tap) weight tiles arrive by LDS-DMA
// (double buffer, one barrier per chunk, persistent); four accumulator sets (class x 2 rows); bias / activation epilogue per class on the
// interleaved output positions.  Weights: packed [CoutPad][16][CinTot] by falnet_pack_up2_batched, pair index = 4 class + 2 a + b.
template <typename T>
__global__ __launch_bounds__(512) void conv3x3_up2_dma_kernel(const falnet_conv_t p, int tiles_x, int tiles_y, int ntiles, int GH, int GW) {
    constexpr int TH = 16, NWAVES = 8, BN = 32, MT = 2, NPAIR = 16;
    constexpr int KCV = 32;
    constexpr int NPIX = (TH + 2) * CD_PW;
    constexpr int A_PIECES = (NPIX + 15) / 16, B_PIECES = NPAIR * BN / 16, NPIECES = A_PIECES + B_PIECES;
    constexpr int A_BYTES = A_PIECES * 1024, BUF = NPIECES * 1024;
    __shared__ __attribute__((aligned(1024))) char lds[2 * BUF];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds_base = (unsigned)(unsigned long)(cd_lptr_t)lds;
    const int r = lane & 31, h = lane >> 5;
    const int n0 = blockIdx.y * BN;
    const char* const zero_page = reinterpret_cast<const char*>(g_cd_zero);
    __shared__ __attribute__((aligned(16))) float lds_bias[BN];
    stage_bias_lds(p, n0, BN, lds_bias);  // (read by the tile epilogues: at least one barrier later)
    const falnet_src_t& S = p.src[0];
    const int nchunks = S.C / KCV;
    int my_tiles = 0;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) ++my_tiles;
    const int total = my_tiles * nchunks;

    constexpr int KP = (A_PIECES + NWAVES - 1) / NWAVES, KW = (B_PIECES + NWAVES - 1) / NWAVES;
    const int l4 = lane >> 2, segpos = lane & 3;
    const T* const wptr = reinterpret_cast<const T*>(p.weight_up2);
    const T* const sptr = reinterpret_cast<const T*>(S.ptr);
    const T* const zero_t = reinterpret_cast<const T*>(zero_page);
    // weight piece wid = wave + NWAVES k is rows (wid & 1) 16 + l4 of (class, tap) pair wid >> 1 = (wave >> 1) + 4 k: one lane offset, uniform step
    int64_t w_off0;
    {
        const int pair = wave >> 1, co = n0 + ((wave & 1) << 4) + l4;
        const int gseg = segpos ^ ((lane >> 4) & 3);
        w_off0 = co < p.w_rows ? (int64_t)(co * NPAIR + pair) * p.cin_total + gseg * 8 : (int64_t)(zero_t - wptr);
    }
    const int w_step = (n0 + 32 <= p.w_rows) ? (NWAVES / 2) * p.cin_total : 0;  // (a partial last row block reads the zero page for every k)
    static_assert(B_PIECES == KW * NWAVES, "weight pieces divide over the waves");
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
            const int pr = pix / CD_PW, pc = pix - pr * CD_PW;
            const int vy = ty0 - 1 + pr, vx = tx0 - 1 + pc;
            const bool ok = pix < NPIX && vy >= 0 && vy < GH && vx >= 0 && vx < GW;
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
            cd_glds16(wptr + q.c * KCV + w_off0 + (i - KP) * w_step, dst0 + A_BYTES + wid * 1024);
        }
    };
    auto issue = [&](const Cur& q, int buf) {
#pragma unroll
        for (int i = 0; i < KP + KW; ++i) issue_piece(q, buf, i);
    };
    // fragment read addresses: patch row (2 wave + rs), rs = mt + dy + 1 in 0..3; patch column r + dx + 1, dx + 1 in 0..2
    int a_addr[4][3];
#pragma unroll
    for (int rs = 0; rs < 4; ++rs)
#pragma unroll
        for (int dxi = 0; dxi < 3; ++dxi) {
            const int pp = (wave * MT + rs) * CD_PW + dxi + r;
            a_addr[rs][dxi] = pp * 64 + ((h ^ ((pp >> 2) & 3)) << 4);
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
        const char* const Bf = lds;
        int (&aa)[4][3] = a_addr;  // the addresses already point into buffer it & 1 (toggled in place below: twelve registers less than a copy)
        int (&bb)[2] = b_lane;
        s16x8_t fa[2][MT], fb[2];
        auto load_step = [&](int st, int set) {
            const int pr_ = st >> 1, ks = st & 1;
            const int cls = pr_ >> 2, a = (pr_ >> 1) & 1, b2 = pr_ & 1;
            const int dyi = (cls >> 1) + a, dxi = (cls & 1) + b2;  // (py - 1 + a) + 1, (px - 1 + b) + 1
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) fa[set][mt] = *reinterpret_cast<const s16x8_t*>(Bf + (aa[mt + dyi][dxi] ^ (ks << 5)));
            fb[set] = *reinterpret_cast<const s16x8_t*>(Bf + bb[ks] + (pr_ * BN) * 64);
        };
        load_step(0, 0);
#pragma unroll
        for (int st = 0; st < 2 * NPAIR; ++st) {
            if (st + 1 < 2 * NPAIR) load_step(st + 1, (st + 1) & 1);
            if (st < KP + KW && more) issue_piece(qi, (it + 1) & 1, st);
            const int cls = st >> 3;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[cls][mt][0] = H16<T>::mma(fb[st & 1], fa[st & 1][mt], acc[cls][mt][0]);
            __builtin_amdgcn_sched_group_barrier(0x100, MT + 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, MT, 0);
        }
        if (more) advance(qi);
        {
            const int d = (it & 1) ? -BUF : BUF;
#pragma unroll
            for (int rs = 0; rs < 4; ++rs)
#pragma unroll
                for (int dxi = 0; dxi < 3; ++dxi) a_addr[rs][dxi] += d;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) b_lane[ks] += d;
        }
        if (++cc == nchunks) {
            cc = 0;
            int b, ty0, tx0;
            tile_coords(ctile, b, ty0, tx0);
            ctile += gridDim.x;
            const int cstride = p.out_cstride;
            float bias[1][16];
            load_bias16_lds(lds_bias, 0, h, bias);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int py = c >> 1, px = c & 1;
                const int x = 2 * (tx0 + r) + px;
                auto pixoff = [&](int mt) -> int64_t {
                    const int y = 2 * (ty0 + wave * MT + mt) + py;
                    if (!(y < p.OH && x < p.OW)) return (int64_t)-1;
                    return (((int64_t)b * p.OH + y) * p.OW + x) * cstride;
                };
                epilogue_direct<T, MT, 1, decltype(pixoff), NoPool, FALNET_DMA_EPI_AHEAD, false, true>(p, acc[c], bias, n0, lane, pixoff);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int j = 0; j < 16; ++j) acc[c][mt][0][j] = 0.f;
            }
        }
    }
}

// variant 18: a dense 3x3 stride-1 forward launch whose ONE source is the launch size halved exactly (2x nearest upsampling), 16-bit, with
// sub-pixel weights (falnet_conv_t::weight_up2)
bool falnet_conv_up2_dma_applicable(const falnet_conv_t& p) {
    if (p.dtype != FALNET_BF16 && p.dtype != FALNET_F16) return false;
    if (!p.weight_up2 || p.nsrc != 1 || p.w_taps != 9 || p.out_layout != FALNET_OUT_NHWC || p.pool_out || p.ksplit > 1 || !p.out) return false;
    const falnet_src_t& S = p.src[0];
    if (S.C % 32 || S.C <= 0 || S.C != p.cin_total) return false;
    if (2 * S.H != p.IH || 2 * S.W != p.IW || p.OH != p.IH || p.OW != p.IW || S.H < 4 || S.W < 32) return false;
    if ((int64_t)S.H * S.sy >= (1ll << 31) || (int64_t)p.w_rows * 16 * p.cin_total >= (1ll << 31) || (p.cin_total + 64) * 2 > CD_ZERO_BYTES) return false;
    return true;
}

int falnet_conv_up2_dma_launch(const falnet_conv_t& p, hipStream_t st) {
    const int GH = p.src[0].H, GW = p.src[0].W;  // the low-resolution grid
    const int tiles_x = (GW + 31) / 32, tiles_y = (GH + 15) / 16;
    const int ntiles = p.B * tiles_x * tiles_y;
    const int ny = (p.Cout + 31) / 32;
    int gx = 256 / ny;
    if (gx < 1) gx = 1;
    if (gx > ntiles) gx = ntiles;
    const dim3 grid((unsigned)gx, (unsigned)ny);
    if (p.dtype == FALNET_F16)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(conv3x3_up2_dma_kernel<f16_t>), grid, dim3(512), 0, st, p, tiles_x, tiles_y, ntiles, GH, GW);
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(conv3x3_up2_dma_kernel<bf16_t>), grid, dim3(512), 0, st, p, tiles_x, tiles_y, ntiles, GH, GW);
    FALNET_RETURN_LAUNCH();
}

// ============================================================================================================================
// The deepest levels (models/FAL_netB.py:107-119,130-137: conv5, conv5_1, conv6, conv6_1, deconv6, iconv6 and their data gradients at
// 256 x 512 input: 8 x 16 and 4 x 8 maps, 512 channels) -- variant 19.
//
// These layers are 0.6-5 GFLOP with 2.4-7 MB of weights and at most 1024 output positions in the whole batch: no tiling of M x N fills 256
// CUs, and the gather kernel's answer (split-K over blockIdx.z, one exposed global -> register -> LDS round trip per 32-channel K step,
// atomics, a second launch for the epilogue) spends 24-31 us on 18 dependent memory round trips per workgroup.  A launch this small is a
// LATENCY chain, not a throughput problem; this kernel keeps the chain at one memory round trip per stage:
//   * a workgroup owns 128 or 256 output positions (whole images: 8 x 16 maps, or 4 x 8 maps four to the M tile) x 64 output channels x ONE
//     K slice of 32 or 64 input channels over all nine taps.  Everything it will ever read -- the halo patch of its images for that slice and
//     the nine 64-row weight tiles, 50-150 KB -- is requested by LDS-DMA up front, no buffer is reused.  SIXTEEN waves issue the pieces (a
//     wave issues one 1-KiB piece per ~100 ns: four waves needed 2 us for 90 pieces), the first four or eight then run the MFMAs, the
//     others leave at the barrier;
//   * the K slices of a tile add their f32 partial sums into the all-zero split-K workspace (row = position, 32 consecutive channels per
//     atomic instruction), and the slice that arrives LAST at the tile's counter applies the epilogue (bias / residual / activation /
//     activation gradient; its operands were fetched at the start, behind the DMA), writes the NHWC output and returns workspace and
//     counter to zero with the same atomic exchanges that read the sums: no second launch, no fence;
//   * stride-2 forward launches (conv5, conv6) use the same kernel with a (2 TH + 1) x (2 TW + 1) patch per image; a source at half the
//     launch size is read through the nearest-upsample map (deconv6).
// Phase times of one workgroup (tools/deep_stamps.py, 512 -> 512 channels at 8 x 16): see DESIGN.md section 4.
typedef unsigned deep_v4u __attribute__((ext_vector_type(4)));
struct falnet_deep_geom_t {
    int imgs, mtiles;        // images per M tile (MTILES x 32 positions), M tiles in the batch
    int PH, PW;              // halo patch of one image
    unsigned m_ppi, m_pw;    // floor(2^32 / d) + 1 for d = PH * PW and PW: x / d = umulhi(x, m) for x < 2^16
    int a_pieces;            // 1-KiB pieces of one 32-channel patch plane
};

template <typename T, int NPL, int MTILES>
__global__ __launch_bounds__(1024) void conv3x3_deep_kernel(const falnet_conv_t p, const falnet_deep_geom_t g) {
    constexpr int BN = 64, NT = 2, NW = 16, BM = 32 * MTILES, ITEMS = BM * (BN / 8) / 1024;  // 8-channel epilogue items per thread: 1 or 2
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef FALNET_DEEP_STAMPS  // profiling build (tools/deep_stamps.py): s_memrealtime (100 MHz) at the phase boundaries of every workgroup -> p.pool_actout
    unsigned long long* const stamp_out = reinterpret_cast<unsigned long long*>(const_cast<void*>(p.pool_actout)) + (size_t)blockIdx.x * 8;
#define DEEP_STAMP(k) do { unsigned long long t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); if (threadIdx.x == 0) stamp_out[k] = t_; } while (0)
#else
#define DEEP_STAMP(k) do { } while (0)
#endif
    DEEP_STAMP(0);
    const unsigned lds_base = (unsigned)(unsigned long)(cd_lptr_t)lds;
    const int r = lane & 31, h = lane >> 5;
    const int nblocks = p.w_rows / BN, ksplits = p.ksplit;
    // blockIdx.x = 8 q + x: tile (q / ksplits) 8 + x, K slice q % ksplits: all slices of a tile, and with tile = mtile * nblocks + nblock one
    // or two channel blocks of weights, on ONE XCD (blockIdx.x % 8)
    const int ntiles = g.mtiles * nblocks;
    const int xq = blockIdx.x >> 3;
    const int tile = (xq / ksplits) * 8 + (blockIdx.x & 7), ks_ = xq % ksplits;
    if (tile >= ntiles) return;
    const int mt = tile / nblocks, nb = tile - mt * nblocks;
    const int n0 = nb * BN;
    constexpr int KC = 32 * NPL;
    const int cglob = ks_ * KC;  // first channel of the slice in the concatenated K axis
    const int C0 = p.src[0].C;
    const bool second = p.nsrc > 1 && cglob >= C0;
    const T* const sp = reinterpret_cast<const T*>(second ? p.src[1].ptr : p.src[0].ptr);
    const int64_t Ssb = second ? p.src[1].sb : p.src[0].sb, Ssy = second ? p.src[1].sy : p.src[0].sy, Ssx = second ? p.src[1].sx : p.src[0].sx;
    const int SH = second ? p.src[1].H : p.src[0].H, SW = second ? p.src[1].W : p.src[0].W;
    const int cloc = cglob - (second ? C0 : 0);
    const int hs = SH != p.IH ? 1 : 0, wsft = SW != p.IW ? 1 : 0;  // exact 2x nearest upsampling (dispatcher checks)
    const int ppi = g.PH * g.PW, npix = g.imgs * ppi;
    const int A_PLANE = g.a_pieces * 1024, W_BASE = NPL * A_PLANE;
    const T* const zero_t = reinterpret_cast<const T*>(g_cd_zero);
    const int l4 = lane >> 2, segpos = lane & 3;

    // ---- every byte this workgroup reads, requested now by all sixteen waves ----
    for (int a = wave; a < g.a_pieces; a += NW) {
        const int pix = 16 * a + l4;
        const int pi = (int)__umulhi((unsigned)pix, g.m_ppi);
        const int rem = pix - pi * ppi;
        const int pr = (int)__umulhi((unsigned)rem, g.m_pw);
        const int pc = rem - pr * g.PW;
        const int vy = pr - 1, vx = pc - 1, b = mt * g.imgs + pi;
        const bool ok = pix < npix && b < p.B && vy >= 0 && vy < p.IH && vx >= 0 && vx < p.IW;
        const T* src = ok ? sp + (int64_t)b * Ssb + (int64_t)(vy >> hs) * Ssy + (int64_t)(vx >> wsft) * Ssx + cloc + (segpos ^ ((pix >> 2) & 3)) * 8 : zero_t;
        const int step = ok ? 32 : 0;
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) cd_glds16(src + pl * step, lds_base + pl * A_PLANE + a * 1024);
    }
    {
        // weight piece id = (tap * NPL + plane) * 4 + quarter: rows 16 quarter + l4 of that 64-row tile; wave w takes ids w, w + 16, ...
        const int gseg = segpos ^ ((l4 >> 2) & 3);
        const T* const wbase = reinterpret_cast<const T*>(p.weight) + (int64_t)(n0 + l4) * p.w_taps * p.cin_total + cglob + gseg * 8;
        const int64_t qstride = (int64_t)16 * p.w_taps * p.cin_total;
        for (int id = wave; id < 9 * NPL * 4; id += NW) {
            const int quarter = id & 3, tp = id >> 2, t = tp / NPL, pl = tp - t * NPL;
            cd_glds16(wbase + quarter * qstride + (int64_t)p.tap_w[t] * p.cin_total + pl * 32, lds_base + W_BASE + id * 1024);
        }
    }
    DEEP_STAMP(1);
    const int hw = p.TH * p.TW;
    const int64_t M = (int64_t)p.B * hw;
    // epilogue operands of this thread's 8-channel items (needed only by the tile's last slice; fetched here, behind the DMA, because a
    // dependent load after the counter is another exposed round trip)
    const T* addend = reinterpret_cast<const T*>(p.addend);
    const T* actout = reinterpret_cast<const T*>(p.actout);
    uint4 e_add[ITEMS] = {}, e_act[ITEMS] = {};
    float e_bias[8] = {};
    int e_off[ITEMS];  // element offsets in the output (the dispatcher bounds the output at 2^31 elements)
    {
        const int n = n0 + (tid & 7) * 8;
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) {
            e_off[it] = -1;
            const int64_t m = (int64_t)mt * BM + ((tid + it * 1024) >> 3);
            if (m >= M || n >= p.Cout) continue;
            const int b = (int)(m / hw), qr = (int)(m - (int64_t)b * hw);
            const int ty = qr / p.TW, tx = qr - ty * p.TW;
            e_off[it] = ((b * p.OH + ty) * p.OW + tx) * p.out_cstride + n;
            if (addend) e_add[it] = *reinterpret_cast<const uint4*>(addend + e_off[it]);
            if (actout) e_act[it] = *reinterpret_cast<const uint4*>(actout + e_off[it]);
        }
        if (p.bias && n < p.Cout) {
#pragma unroll
            for (int k = 0; k < 8; ++k) e_bias[k] = p.bias[n + k];
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // the whole LDS image is in place
    DEEP_STAMP(2);

    // partial tiles: scratch[slice][position][w_rows] f32
    const __amdgpu_buffer_rsrc_t slab_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.scratch, 0, (int)p.scratch_bytes, 0x00020000);
    const int w_rows = p.w_rows;
    const int64_t slab_stride = (int64_t)g.mtiles * BM * w_rows;
    if (wave < MTILES) {
        // ---- fragment addresses: lane r of wave w holds position 32 w + r of the tile (operand B), weight row 32 nt + r (operand A) ----
        int pp0;
        {
            const int q = 32 * wave + r;
            const int qi = q / hw, qr = q - qi * hw;
            const int ty = qr / p.TW, tx = qr - ty * p.TW;
            pp0 = qi * ppi + ty * p.isy * g.PW + tx * p.isx;  // patch pixel under tap (-1, -1)
        }
        int b_addr[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) b_addr[nt] = W_BASE + (32 * nt + r) * 64 + ((h ^ ((r >> 2) & 3)) << 4);
        f32x16 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[nt][j] = 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int pp = pp0 + (p.tap_dy[t] + 1) * g.PW + p.tap_dx[t] + 1;
            const int a_addr = pp * 64 + ((h ^ ((pp >> 2) & 3)) << 4);
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const s16x8_t fa = *reinterpret_cast<const s16x8_t*>(lds + pl * A_PLANE + (a_addr ^ (ks << 5)));
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const s16x8_t fb = *reinterpret_cast<const s16x8_t*>(lds + (b_addr[nt] ^ (ks << 5)) + (t * NPL + pl) * (BN * 64));
                        acc[nt] = H16<T>::mma(fb, fa, acc[nt]);  // D rows = output channels, D columns (lanes) = positions
                    }
                }
        }
        DEEP_STAMP(3);
        // ---- this slice's partial tile -> scratch: the position on the lane, channels 8 (j >> 2) + 4 h + (j & 3) in its accumulators: 16-B
        // agent-coherent stores (sc1: written through to the point of coherence, no L2 write-back / invalidate as a fence would need; 8-B
        // stores cost 2.7x per byte) ----
        const int64_t m = (int64_t)mt * BM + 32 * wave + r;
        if (m < M) {
            const unsigned row = (unsigned)(((int64_t)ks_ * slab_stride + m * w_rows + n0 + 4 * h) * 4);  // byte offset (scratch <= 2^31 bytes: dispatcher)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int jg = 0; jg < 4; ++jg) {
                    deep_v4u v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = __float_as_uint(acc[nt][4 * jg + e]);
                    __builtin_amdgcn_raw_buffer_store_b128(v, slab_rsrc, row + (32 * nt + 8 * jg) * 4, 0, 16);  // aux 16 = sc1
                }
        }
        DEEP_STAMP(4);
    }
    // ---- the last slice to arrive finishes the tile ----
    // No release / acquire fence here: an agent-scope fence on gfx950 writes back and invalidates the XCD's whole L2 (buffer_wbl2 / buffer_inv
    // sc1), per workgroup -- measured 99-146 us per launch instead of 20.  The protocol needs none: the partial tiles are written and read
    // with agent-coherent accesses (sc1), a wave's stores are acknowledged when its s_waitcnt vmcnt(0) returns, and the counter increment
    // follows the barrier that follows every wave's wait.  (First version: f32 atomics into the zeroed split-K workspace -- 4.2 M atomic lanes
    // per 512-channel layer saturate the L2 atomic units: 8 us to acknowledge them, another 10 us for counter and read-back behind them.)
    unsigned* const counter = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(p.splitk_ws) + p.splitk_ws_bytes) - 4096 + tile;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    DEEP_STAMP(5);
    __builtin_amdgcn_s_barrier();  // (also: every wave is past its last fragment read, the LDS image is dead)
    volatile __attribute__((address_space(3))) int* const flag = reinterpret_cast<volatile __attribute__((address_space(3))) int*>((cd_lptr_t)lds);
    if (tid == 0) {
        const unsigned old = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        flag[0] = old + 1 == (unsigned)ksplits;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    DEEP_STAMP(6);
    if (!flag[0]) return;
    T* out = reinterpret_cast<T*>(p.out);
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
        if (e_off[it] < 0) continue;
        const int i = tid + it * 1024;
        const int64_t m = (int64_t)mt * BM + (i >> 3);
        const unsigned src = (unsigned)((m * w_rows + n0 + (i & 7) * 8) * 4), sstep = (unsigned)(slab_stride * 4);
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = 0.f;
        for (int ks = 0; ks < ksplits; ks += 4) {  // slices in order: the sum does not depend on which slice finishes (ksplits % 4 == 0: dispatcher)
            deep_v4u x[4][2];                        // eight 16-B agent-coherent loads in flight
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int k = 0; k < 2; ++k) x[q][k] = __builtin_amdgcn_raw_buffer_load_b128(slab_rsrc, src + (ks + q) * sstep + k * 16, 0, 16);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] += __uint_as_float(x[q][k >> 2][k & 3]);
        }
        if (addend) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                v[2 * k] += H16<T>::lo((&e_add[it].x)[k]);
                v[2 * k + 1] += H16<T>::hi((&e_add[it].x)[k]);
            }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = apply_act(v[k] + e_bias[k], p.act);
        if (actout) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                v[2 * k] *= act_grad_from_out(H16<T>::lo((&e_act[it].x)[k]), p.actout_kind);
                v[2 * k + 1] *= act_grad_from_out(H16<T>::hi((&e_act[it].x)[k]), p.actout_kind);
            }
        }
        uint4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) (&o.x)[k] = pack16x2<T>(v[2 * k], v[2 * k + 1]);
        *reinterpret_cast<uint4*>(out + e_off[it]) = o;
    }
    DEEP_STAMP(7);
    if (tid == 0) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
#undef DEEP_STAMP

// geometry of a variant-19 launch; MTILES = 8 (256 positions per workgroup) when the 128-position form would need more than one round of
// workgroups on the 256 CUs and the image fits
static bool deep_geometry(const falnet_conv_t& p, falnet_deep_geom_t& g, int& npl, int& mtiles_wg, size_t& lds_bytes) {
    const int hw = p.TH * p.TW;
    if (hw <= 0 || hw > 128 || 128 % hw) return false;
    if (p.ksplit * 32 == p.cin_total) npl = 1;
    else if (p.ksplit * 64 == p.cin_total) npl = 2;
    else return false;
    g.PH = (p.TH - 1) * p.isy + 3;
    g.PW = (p.TW - 1) * p.isx + 3;
    const int ppi = g.PH * g.PW;
    g.m_ppi = (unsigned)((1ull << 32) / (unsigned)ppi) + 1u;
    g.m_pw = (unsigned)((1ull << 32) / (unsigned)g.PW) + 1u;
    for (mtiles_wg = 8; mtiles_wg >= 4; mtiles_wg -= 4) {
        g.imgs = 32 * mtiles_wg / hw;
        g.mtiles = (p.B + g.imgs - 1) / g.imgs;
        g.a_pieces = (g.imgs * ppi + 15) / 16;
        lds_bytes = (size_t)npl * g.a_pieces * 1024 + (size_t)9 * npl * 4096;
        const bool fits = lds_bytes <= 160 * 1024 && g.a_pieces * 16 < 65536;
        if (mtiles_wg == 4) return fits;
        const int wgs4 = (p.B * hw + 127) / 128 * (p.w_rows / 64) * p.ksplit;
        if (fits && wgs4 > 256) return true;
    }
    return false;
}

// variant 19: canonical nine-tap launch on a map of at most 128 positions whose images tile 128 exactly, stride 1 or 2, dense NHWC output,
// ksplit = cin_total / 32 or cin_total / 64 K slices (a multiple of 4), scratch for the partial tiles, the split-K workspace (all-zero between
// launches) for the tile counters in its last 16 KiB
bool falnet_conv_deep_applicable(const falnet_conv_t& p) {
    if (p.dtype != FALNET_BF16 && p.dtype != FALNET_F16) return false;
    if (p.ntaps != 9 || p.nsrc < 1 || p.nsrc > 2 || p.out_layout != FALNET_OUT_NHWC || p.pool_out || !p.out || !p.splitk_ws || !p.scratch) return false;
    if (p.ksplit < 4 || p.ksplit % 4 || p.splitk_ws_bytes < 16384 || p.splitk_ws_bytes % 4) return false;
    if (p.isy != p.isx || (p.isy != 1 && p.isy != 2) || p.osy != 1 || p.osx != 1 || p.ooy || p.oox) return false;
    if (p.OH != p.TH || p.OW != p.TW || p.IH != p.TH * p.isy || p.IW != p.TW * p.isx) return false;
    if (p.w_rows % 64 || p.Cout % 8 || p.Cout > p.w_rows || (int64_t)p.B * p.OH * p.OW * p.out_cstride >= (1ll << 31)) return false;
    for (int t = 0; t < 9; ++t)
        if (p.tap_dy[t] < -1 || p.tap_dy[t] > 1 || p.tap_dx[t] < -1 || p.tap_dx[t] > 1 || p.tap_w[t] < 0 || p.tap_w[t] >= p.w_taps) return false;
    falnet_deep_geom_t g;
    int npl, mtw;
    size_t lds_bytes;
    if (!deep_geometry(p, g, npl, mtw, lds_bytes)) return false;
    int csum = 0;
    for (int s = 0; s < p.nsrc; ++s) {
        const falnet_src_t& S = p.src[s];
        if (S.C <= 0 || S.C % (32 * npl)) return false;
        if (!((S.H == p.IH && S.W == p.IW) || (2 * S.H == p.IH && 2 * S.W == p.IW))) return false;
        csum += S.C;
    }
    if (csum != p.cin_total || (p.cin_total + 64) * 2 > CD_ZERO_BYTES) return false;
    if (g.mtiles * (p.w_rows / 64) > 4096) return false;  // tile counters: the last 16 KiB of the split-K workspace
    const int64_t need = (int64_t)p.ksplit * g.mtiles * 32 * mtw * p.w_rows * 4;
    return need <= p.scratch_bytes && p.scratch_bytes < (1ll << 31);
}

template <typename T, int NPL, int MTILES>
static void deep_launch(const falnet_conv_t& p, const falnet_deep_geom_t& g, size_t lds_bytes, dim3 grid, hipStream_t st) {
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_deep_kernel<T, NPL, MTILES>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)attr;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(conv3x3_deep_kernel<T, NPL, MTILES>), grid, dim3(1024), lds_bytes, st, p, g);
}

int falnet_conv_deep_mtiles(const falnet_conv_t& p) {  // 4 or 8: the instantiation falnet_conv_deep_launch uses (kernel-name query)
    falnet_deep_geom_t g;
    int npl, mtw = 4;
    size_t lds_bytes;
    deep_geometry(p, g, npl, mtw, lds_bytes);
    return mtw;
}

int falnet_conv_deep_launch(const falnet_conv_t& p, hipStream_t st) {
    falnet_deep_geom_t g;
    int npl, mtw;
    size_t lds_bytes;
    deep_geometry(p, g, npl, mtw, lds_bytes);
    const int ntiles = g.mtiles * (p.w_rows / 64);
    const dim3 grid((unsigned)((ntiles + 7) / 8 * 8 * p.ksplit));
#define DEEP_L(T) do {                                                          \
        if (npl == 1 && mtw == 4) deep_launch<T, 1, 4>(p, g, lds_bytes, grid, st);      \
        else if (npl == 1) deep_launch<T, 1, 8>(p, g, lds_bytes, grid, st);             \
        else if (mtw == 4) deep_launch<T, 2, 4>(p, g, lds_bytes, grid, st);             \
        else deep_launch<T, 2, 8>(p, g, lds_bytes, grid, st);                           \
    } while (0)
    FALNET_DISPATCH_16(p.dtype, DEEP_L);
#undef DEEP_L
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
    __shared__ __attribute__((aligned(16))) float lds_bias[BN];
    stage_bias_lds(p, n0, BN, lds_bias);  // (read by the tile epilogues: at least one barrier later)
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
                load_bias16_lds(lds_bias, 32 * nt, h, bias);
                epilogue_direct<T, 1, 1, decltype(pixoff), NoPool, FALNET_DMA_EPI_AHEAD, false, true>(p, acc[nt], bias, n0 + 32 * nt, lane, pixoff);
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
