/*
 * falnet_hip.h -- C-ABI of libfalnet_hip.so: the MI355X (gfx950) kernels behind the
 * FAL_netB hot path.
 *
 * The reference (JuanLuisGonzalez/FAL_net) has no native code: its "FFI" for this path is
 * torch.nn.functional -> aten/cuDNN.  Each entry point below replaces the aten call sites
 * named in its comment (paths relative to the reference root).  The library is loaded with
 * ctypes by fal_net_amd/_lib.py; INTEGRATION.md shows the binding.
 *
 * Conventions
 *   - plain `extern "C"`, raw device pointers, explicit sizes; no torch types.
 *   - every launch function takes the HIP stream (as void*) and enqueues only on it; it never
 *     allocates, never synchronises and never touches the default stream; on entry it makes the
 *     stream's device current (hipStreamGetDevice + hipSetDevice).  Workspaces are caller-owned:
 *     falnet_wgrad_workspace_bytes (split-K slabs), falnet_conv_t::splitk_ws (+ _bytes, M * w_rows
 *     floats for a split-K launch); no other entry point needs scratch memory.
 *   - return value: 0 ok, <0 bad argument (see falnet_last_error()), >0 hipError_t.
 *   - re-entrant: forward runs on the Python main thread, backward on autograd's thread.
 *   - dtype enum: activations/weights are FALNET_F32 (exact-f32 MFMA, parity path), FALNET_BF16
 *     (bf16 MFMA, f32 accumulate; throughput path) or FALNET_F16 (IEEE half MFMA, f32 accumulate:
 *     11 significant bits instead of bf16's 8; BASELINE configs[4]).  Reductions, losses, the MED
 *     head and Adam are f32 in all three.
 *   - activation layout inside the network: NHWC ("pixel-major"), channels padded to a
 *     multiple of falnet_channel_pad(dtype) with zeros.  The boundary tensors of the
 *     reference API (images, disparity, synthesised view, MED logits) are planar NCHW f32.
 *   - kernel-selection switches (A/B and tests; read once from the environment): FALNET_DISABLE_PATCH,
 *     FALNET_PATCH_KCB (conv.hip), FALNET_WR_FORM / FALNET_WR_ABL (wgrad_rows.hip), FALNET_HEAD_V1,
 *     FALNET_HEAD_BWD_V1, FALNET_HEAD_PAIR, FALNET_HEAD_FWD2, FALNET_HEAD_BWD2, FALNET_HEAD_WAVE (MED head), FALNET_MFMA16 (16x16x32 form of the
 *     weight-stationary kernel), FALNET_WS2_FULL (whole-line loads of the two-phase weight-stationary kernel, variant 16), FALNET_GEMM_WAVE
 *     (small f32 GEMM).  Host side (fal_net_amd/ops.py): FALNET_WS2 / FALNET_NO_DMA / FALNET_S2F_DMA gate autotune candidates.
 */
#ifndef FALNET_HIP_H
#define FALNET_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { FALNET_F32 = 0, FALNET_BF16 = 1, FALNET_F16 = 2 };
enum { FALNET_ACT_NONE = 0, FALNET_ACT_ELU = 1, FALNET_ACT_RELU = 2 };
enum { FALNET_OUT_NHWC = 0, FALNET_OUT_PLANAR_F32 = 1 };

int falnet_version(void);
const char* falnet_last_error(void);
/* channel padding granule (elements) of NHWC tensors and packed weights: 32 for every dtype */
int falnet_channel_pad(int dtype);
/* Every launch function makes the device that owns its `stream` argument current on the calling thread before it enqueues
 * (forward runs on the Python main thread, backward on autograd's worker thread: no thread-local HIP state is assumed).
 * falnet_set_device is for callers that use the NULL stream from a thread whose current device is not the tensors'. */
int falnet_set_device(int device);
/* Deterministic mode (process-wide switch, default off): every result becomes bit-identical from run to run for identical inputs,
 * launch sequence and kernel choices.  What changes: scalar loss reductions add their per-workgroup partials in workgroup order
 * (last-arriver pattern over a static scratch: run them on ONE stream per device); falnet_wgrad_fuses_bias reports 0 (the fused
 * bias gradient adds with atomics); split-K convolution launches (ksplit > 1) are refused with -2; falnet_bias_grad_batched is
 * refused in favour of falnet_bias_grad_batched_det; falnet_wgrad_reduce / falnet_bias_grad use a single writer per element.  The
 * CALLER keeps `groups` = 1 in its falnet_reduce_t tables and makes its kernel choices reproducible (fal_net_amd.ops with
 * FALNET_DETERMINISTIC=1: cached or heuristic variants, no timing).  Throughput drops (no split-K on the deep levels, a second
 * pass for the bias gradients): a debugging / regression-hunting switch, like torch.use_deterministic_algorithms. */
int falnet_set_deterministic(int on);
int falnet_get_deterministic(void);

/* One input of a convolution: NHWC tensor (or a per-sample constant when sy = sx = 0). */
typedef struct {
    const void* ptr;
    int32_t C;          /* channels consumed (multiple of the channel pad) */
    int32_t H, W;       /* physical spatial size; != (IH, IW) means nearest-upsampled on the fly */
    int64_t sb, sy, sx; /* element strides: sample, row, pixel */
} falnet_src_t;

/*
 * Implicit-GEMM convolution on MFMA: out[p, co] = epilogue(sum_{tap, src, ci} in_src[nbr(p, tap), ci] * w[co, tap, ci]).
 * One kernel serves
 *   - forward 3x3 / 1x1, stride 1 / 2            (models/FAL_netB.py:38,45,55,73,75,127,190; loss_functions.py:21-29)
 *   - the fused nearest upsample of `deconv`     (FAL_netB.py:58)   via src.H/W != IH/IW
 *   - the fused channel concat                   (FAL_netB.py:145-173) via two sources
 *   - data gradients (the autograd of the above): a dgrad is the same gather with another tap table;
 *     stride-2 dgrad is 4 launches, one per output parity class.
 * Tile space (TH x TW positions per sample): position (ty, tx) reads virtual input pixel
 * (ty*isy + dy[t], tx*isx + dx[t]) for tap t (zero outside [0,IH)x[0,IW)) and writes output pixel
 * (ty*osy + ooy, tx*osx + oox).
 * Epilogue: v = acc + bias[co]; v += addend[p, co]; v = act(v); v *= act'(actout[p, co]); store.
 */
typedef struct {
    falnet_src_t src[2];
    int32_t nsrc;
    int32_t IH, IW;            /* virtual input size */
    const void* weight;        /* packed [CoutPad][ntaps_w][CinTot] in `dtype` (falnet_pack_weights) */
    int32_t cin_total;         /* CinTot: row length per tap of the packed weight */
    int32_t ntaps;             /* taps walked by this launch (<= 9) */
    int32_t tap_dy[9], tap_dx[9], tap_w[9]; /* offsets and packed-weight tap index */
    int32_t w_taps;            /* ntaps_w of the packed weight */
    int32_t w_rows;            /* CoutPad: rows of the packed weight (multiple of 32) */
    int32_t isy, isx;
    int32_t B, TH, TW;
    int32_t osy, osx, ooy, oox;
    void* out;                 /* NHWC `dtype` [B][OH][OW][out_cstride] or planar f32 [B][Cout][OH][OW] */
    int32_t OH, OW;
    int32_t Cout;              /* channels written: NHWC -> padded count, planar -> real count */
    int32_t out_cstride;
    int32_t out_layout;
    const float* bias;         /* [CoutPad] or NULL */
    const void* addend;        /* NHWC like out, or NULL (may alias out: accumulate) */
    int32_t act;               /* FALNET_ACT_* applied to v */
    const void* actout;        /* NHWC like out: multiply by d act / d pre computed from the activation output */
    int32_t actout_kind;       /* FALNET_ACT_ELU / RELU / NONE */
    int32_t dtype;
    int32_t ksplit;            /* gather kernel: > 1 splits the K loop over blockIdx.z (f32 atomics into
                                  splitk_ws [B*TH*TW][w_rows], then an epilogue launch); for small-M layers.
                                  Variant 19: the number of 32- or 64-channel K slices (cin_total / 32 or / 64, a multiple of 4) */
    float* splitk_ws;          /* must be ALL-ZERO on entry; the split-K epilogue leaves it all-zero again (no memset per launch).
                                  Variant 19 keeps its tile counters in the last 16 KiB (zero on entry, zero again on exit) */
    int64_t splitk_ws_bytes;
    int32_t variant;           /* kernel choice: 0 heuristic, 1 gather, 2/3 halo-patch with 128-/64-B K chunks,
                                  4 halo-patch single-stage, 5 single-stage double-buffered,
                                  6/7 = 3/4 on a 16x32-position block with 8 waves, 8/9 = 3/4 on a 4x32 block;
                                  10 weight-stationary persistent kernel (<= 128 B of input channels per pixel);
                                  11 / 12 gather with 32 / 64 output channels per workgroup (more workgroups for small layers);
                                  13 LDS-DMA double-buffered persistent kernel (16-bit operands, 16x32 positions x 64 channels per
                                  workgroup, sources at the launch size or exactly half of it);
                                  15 forward 3x3 stride-2 by LDS-DMA (parity-de-interleaved patch, 16-channel chunks);
                                  18 = `deconv` forward in sub-pixel form (one source at exactly half the launch size, weight_up2 set): four 2x2
                                  convolutions of the low-resolution map, 16 instead of 36 tap-MACs per output quad (conv_dma.hip);
                                  17 = 13 on 4x32-position tiles with four waves (one row each): small maps (levels 4-6) get 4x the workgroups;
                                  20 = 13 on 8x32-position tiles with eight waves (one row each): the 32x64 maps of level 3;
                                  16 = 10 with two groups of four waves half a period apart (one in its MFMAs while the other stores,
                                  loads and runs the epilogue), half-height tiles;
                                  19 = the deepest levels (maps of at most 128 positions, 128 % (TH TW) == 0; stride 1 or 2; nine taps; 16-bit):
                                  one-shot LDS-DMA of a 32- or 64-channel K slice per workgroup, `ksplit` = cin_total / 32 or / 64 slices,
                                  f32 partial tiles in `scratch`, summed IN SLICE ORDER (deterministic) with the epilogue by the slice that
                                  arrives last at the tile's counter (the last 16 KiB of splitk_ws) -- one launch (conv_dma.hip);
                                  21 = 13 re-cut for two resident workgroups per CU: 16x32 positions x 64 channels on FOUR waves of four rows
                                  each, 16-channel K chunks (2 x 38 KiB of LDS), NHWC outputs only -- the two waves of a SIMD belong to different
                                  workgroups, so one's epilogue / chunk barrier overlaps the other's MFMAs (conv_dma.hip: conv3x3_dma2_kernel);
                                  22 = the wave program of 21 on 32x32-position tiles, eight waves, one workgroup per CU (2 x 55 KiB of LDS): 27 % fewer
                                  staged bytes and 40 % fewer fragment reads per MFMA than 13;
                                  23 = 13 on v_mfma_f32_16x16x32 (a 32x32 tile as 2 x 2 MFMAs of K = 32): the MFMA shape the chip holds a higher clock on
                                  (conv_dma.hip: conv3x3_dma16_kernel); 24 / 25 = 17 / 20 in the same way (NHWC outputs, no fused pool);
                                  26 = data gradient of a `deconv` layer (nearest 2x upsampling + 3x3 convolution, FAL_netB.py:52-58) on the LOW-resolution
                                  grid: ONE source = the upstream gradient at exactly twice the output map, weight = falnet_pack_up2_t::wdd
                                  ([w_rows][4][4 C], w_taps = ntaps = 4, cin_total = 4 C) -- a 2x2-tap convolution over pairs of upstream rows /
                                  columns, 16 instead of 36 tap-MACs per input position (conv_dma.hip: conv2x2_up2d_dma16_kernel);
                                  27 = wave-streaming kernel for 32 -> (<= 32) channel layers (conv_wave.hip: conv3x3_wave32_kernel): ONE 32-channel
                                  source at the launch size, 32-row packed weight, NHWC output, no pool / split-K; every wave streams its own
                                  32-pixel strip rows through a private LDS-DMA ring (no workgroup barrier), weights resident in registers;
                                  29 = the same wave-streaming form for ONE 64-channel source and <= 4 output channels written as planar f32, no addend /
                                  activation-gradient operand (conv_wave.hip: conv3x3_wave64p_kernel: the data gradient of VGG19's first convolution);
                                  -2 is returned when the variant does not apply */
    void* pool_out;            /* optional fused 2x2/stride-2 reduction of the (activated) output: NHWC `dtype`
                                  [B][OH/2][OW/2][out_cstride].  pool_mode 0: max (nn.MaxPool2d(2,2) after the VGG slices,
                                  loss_functions.py:21-29); pool_mode 1: sum, then multiplied by d act / d pre taken from
                                  pool_actout (the adjoint of F.interpolate(scale 2, nearest) in front of a deconv,
                                  FAL_netB.py:58, fused into that conv's data-gradient launch).  Halo-patch variants with an
                                  even number of rows per wave only (-2 otherwise); with pool_out set, `out` may be NULL
                                  (only the reduced map is kept) */
    int32_t pool_mode;
    int32_t pool_actout_kind;  /* FALNET_ACT_* of pool_actout */
    const void* pool_actout;   /* NHWC like pool_out, or NULL */
    const void* weight_up2;    /* variant 18 only, else NULL: sub-pixel weights [CoutPad][16][CinTot] in `dtype` of a 3x3 convolution over a 2x
                                  nearest-upsampled source (falnet_pack_up2_batched): pair 4 (2 py + px) + 2 a + b holds the sum of the 3x3
                                  taps that fall on low-resolution neighbour (a, b) for output parity (py, px) */
    void* scratch;             /* variant 19 only, else NULL: uninitialised device scratch for the K-slice partial tiles,
                                  ksplit * ceil(B TH TW / 128 | 256) * 128 | 256 * w_rows f32; contents are meaningless between launches;
                                  launches that share it must be stream-ordered */
    int64_t scratch_bytes;
} falnet_conv_t;
int falnet_conv2d(const falnet_conv_t* p, void* stream);
/* First layer: 3x3 / stride 1 / pad 1 convolution of a 3-channel planar f32 image (FAL_netB.py:99 conv0, VGG19 features[0];
 * loss_functions.py:21) with the f32 OIHW weights as they are -> NHWC `dtype` [B][H][W][Cout], Cout in {32, 64}, bias + act fused. */
int falnet_conv3x3_c3(const float* x_nchw, const float* w_oihw, const float* bias, void* out, int B, int H, int W, int Cout,
                      int act, int dtype, void* stream);
/* n <= 4 gather launches of one family (same dtype / Cout / packed rows / ksplit, NHWC output) in ONE grid: the four
 * output-parity classes of a stride-2 data gradient.  ksplit > 1: every member needs its OWN all-zero splitk_ws region
 * (B*TH*TW*w_rows floats, non-overlapping); one fused epilogue launch follows and leaves the regions zero. */
/* variant 14 in descs[0]: the members must be the four output-parity classes (0,0) (0,1) (1,0) (1,1) of ONE 3x3 / stride-2 / pad-1
 * data gradient in bf16 / f16 (same gout source, packed weight, output tensor; taps as fal_net_amd/ops.py:dgrad_taps_s2); they then
 * run as one LDS-DMA halo kernel (csrc/conv_dma.hip: conv3x3_s2d_dma_kernel) instead of four gather GEMMs. */
int falnet_conv2d_multi(const falnet_conv_t* descs, int n, void* stream);
/* symbol (as rocprofv3 reports it) of the kernel falnet_conv2d launches for this descriptor */
int falnet_conv2d_kernel_name(const falnet_conv_t* p, char* buf, int len);

/*
 * Weight gradient (autograd of the conv2d call sites above):
 *   dW[co, tap, ci] = sum_p gout[p, co] * in[nbr(p, tap), ci]
 * split over pixel ranges into `nsplit` f32 partial slabs in the caller's workspace
 * [nsplit][ntaps][CoutPad][CinTot], then falnet_wgrad_reduce sums the slabs into the
 * OIHW f32 gradient (+= when accumulate).  `gout` is NHWC [B][TH][TW][gC] (tile space = the
 * conv's output grid); sources/taps as in falnet_conv_t.
 */
typedef struct {
    falnet_src_t src[2];
    int32_t nsrc;
    int32_t IH, IW;
    const void* gout;
    int32_t gC;                /* padded channel count of gout (CoutPad) */
    int32_t ntaps;
    int32_t tap_dy[9], tap_dx[9];
    int32_t isy, isx;
    int32_t B, TH, TW;
    int32_t cin_total;
    int32_t nsplit;
    float* partial;            /* workspace */
    int32_t dtype;
    int32_t variant;           /* 0 heuristic (halo-patch kernel, 32x32 channels per workgroup, when dense 3x3 stride 1),
                                  1 force the per-tap kernel, 3 / 4 halo-patch with 32x64 / 64x32 (cin x cout) channels per
                                  workgroup (16-bit operands), 5 parity-plane halo kernel for 3x3 stride-2 launches (16-bit),
                                  6 first layer: src[0].ptr = planar f32 [B][3][IH][IW] image (src[0].C = 3), 16-bit gout with 32
                                  channels, cin_total 32 (slab layout),
                                  7 row-streaming kernel (bf16 / f16, dense 3x3 stride 1, sources at the launch size or exactly
                                  half of it): 64 x 64 channels per workgroup, LDS-DMA row ring, gout fragments in a rolling
                                  register window; nsplit = ranges of (sample, 32-pixel column strip, row) units,
                                  9 wave-streaming kernel (csrc/wgrad_wave.hip; bf16 / f16, dense 3x3, ONE 32-channel NHWC source at the input size):
                                  stride 1 with gC 32 or 64, or stride 2 (TH = ceil(IH / 2), TW = ceil(IW / 2)) with gC 64; nsplit = workgroups =
                                  slabs, every wave streams its own (sample, strip, row) range (stride 2: its own parity plane of one range).
                                  Variant 6 takes the same wave-streaming form when IW % 4 == 0 (not in deterministic mode) */
    float* bias_grad;          /* optional, halo kernels (dense 3x3 stride 1, variant 5): db[co] += sum over positions of gout[.,co]
                                  (f32 atomics, [gC]) from the gout tiles the kernel stages anyway -- replaces a falnet_bias_grad
                                  pass over the same tensor.  Only the kernels falnet_wgrad_fuses_bias() reports fuse it;
                                  falnet_wgrad returns -3 when it is set for a launch whose kernel cannot */
    int32_t cout;              /* real output channels (<= gC): bound of the fused bias gradient (bias_grad holds `cout` floats);
                                  0 = gC */
    int32_t up2;               /* variant 7 only, else 0.  != 0: the weight gradient of a `deconv` layer (nearest 2x upsampling + 3x3 convolution,
                                  FAL_netB.py:52-58) on the LOW-resolution grid: `gout` is the upstream gradient at [B][2 TH][2 TW][gC], the one source
                                  the layer's input at TH x TW (= IH x IW); nsplit = 4 x pixel ranges (split = 4 range + 2 py + px): a split multiplies
                                  the gout pixels of its parity class with the 2 x 2 input offsets they meet (16 instead of 36 tap products per
                                  position) and writes its share of the full 3x3 slab, so the slab reduce is unchanged */
} falnet_wgrad_t;
/* 1 when the kernel falnet_wgrad selects for this descriptor sums the bias gradient itself (bias_grad honoured), else 0 */
int falnet_wgrad_fuses_bias(const falnet_wgrad_t* p);
int64_t falnet_wgrad_workspace_bytes(const falnet_wgrad_t* p);
int falnet_wgrad(const falnet_wgrad_t* p, void* stream);
/* partial [nsplit][ntaps][CoutPad][CinTot] -> grad OIHW f32 [Cout][Cin][kh][kw] (taps in kh*kw+kw order) */
/* channel groups as in falnet_pack_weights: packed column cp holds real channel cp (cp < c0_real) or
 * c0_real + (cp - c0_pad) */
int falnet_wgrad_reduce(const float* partial, int nsplit, int ntaps, int cout_pad, int cin_total,
                        float* grad, int cout, int cin, int c0_real, int c0_pad, int accumulate, void* stream);
/* db[co] (+)= sum_p g[p, co]; g NHWC [npix][gC], gC a multiple of 32 dividing 256 or a multiple of 256 */
int falnet_bias_grad(const void* g, int64_t npix, int gC, int cout, float* db,
                     int accumulate, int dtype, void* stream);

/* Batched forms: device-resident descriptor tables, ONE launch for all layers of a step.  The bias form ADDS into the
 * gradient buffers (f32 atomics), the reduce form as its `accumulate` argument says: the caller zeroes the flat gradient
 * buffer once per step (or keeps it to accumulate).
 * block_begin = first blockIdx.x of the entry; a reduce entry uses ceil(cout / cob) * groups blocks with
 * cob = max(1, 1024 / cin_total) output channels per block (falnet_wgrad_reduce_blocks; block -> (channel block, slab group);
 * groups == 1: plain read-modify-write of the gradient, else f32 atomics); ntaps in {1, 3, 9}; bias entry uses `blocks` blocks. */
int falnet_wgrad_reduce_blocks(int cout, int cin_total, int groups);
typedef struct {
    const float* partial; float* grad;
    int32_t nsplit, ntaps, w_rows, cin_total, cout, cin, c0_real, c0_pad, groups, block_begin;
} falnet_reduce_t;
typedef struct {
    const void* g; float* db; int64_t npix;
    int32_t gC, cout, blocks, block_begin;
} falnet_biasgrad_t;
typedef struct {
    const float* w; void* wf; void* wd;
    int32_t cout, cin, taps, c0_real, c0_pad, cin_pad, cout_pad, block_begin;  /* entry uses (cout_pad/32)*(cin_pad/32) blocks, taps 9, 3 or 1 */
    int32_t no_update, reserved;  /* falnet_adam_pack_batched: != 0 -> this entry is only packed (a derived weight, e.g. composed logits weights) */
} falnet_pack_t;
int falnet_pack_weights_batched(const falnet_pack_t* descs_dev, int n, int total_blocks, int dtype, void* stream);
/* The optimiser step of every packed layer AND its re-pack in one pass over the f32 masters (Train_Stage1_K.py:177-180 torch.optim.Adam; the
 * arithmetic of falnet_adam_step_dev): a block updates the 32 x 32 x taps master tile it is about to pack.  g_off / m_off / v_off: element
 * offsets from a master weight to its gradient and moments (the four flat buffers share one layout).  state = {lr, t} on the device (t is NOT
 * advanced: falnet_adam_tick); scaler: NULL, or the f16 loss-scale state {scale, ., overflow flag, .} -- the update divides by the scale and
 * is skipped as a whole when the flag is set.  Parameters outside the packed layers (biases, factors of composed weights): falnet_adam_ranges. */
int falnet_adam_pack_batched(const falnet_pack_t* descs_dev, int n, int total_blocks, int dtype, int64_t g_off, int64_t m_off, int64_t v_off,
                             const float* state, float b1, float b2, float eps, float grad_scale, const float* scaler, void* stream);
/* the same update over n_ranges element ranges (ranges_dev[2 r] = first element, [2 r + 1] = count) of the flat parameter buffer p */
int falnet_adam_ranges(float* p, int64_t g_off, int64_t m_off, int64_t v_off, const int64_t* ranges_dev, int n_ranges, const float* state,
                       float b1, float b2, float eps, float grad_scale, const float* scaler, void* stream);
/* t += 1 unless the scaler's overflow flag is set (a skipped step does not count) */
int falnet_adam_tick(float* state, const float* scaler, void* stream);
/* Sub-pixel weights of the `deconv` layers (falnet_conv_t::weight_up2; FAL_netB.py:52-58), all layers in one launch: entry i sums the f32 OIHW
 * 3x3 master weights w into wu [cout_pad][16][cin_pad] (`dtype`; pair index 4 (2 py + px) + 2 a + b, rows / columns: py = 0 -> (W[0], W[1] + W[2]),
 * py = 1 -> (W[0] + W[1], W[2])); an entry uses (cout_pad / 32) * (cin_pad / 32) blocks from block_begin. */
typedef struct {
    const float* w; void* wu;
    int32_t cout, cin, cin_pad, cout_pad, block_begin;
    int32_t reserved;
    void* wdd;  /* NULL, or [cin_pad][4][4 cout_pad]: the layer's data-gradient weights on the low-resolution grid (falnet_conv2d variant 26):
                 * wdd[ci][2 du + dv][e 2 cout_pad + f cout_pad + co] = sum of the 3x3 taps that carry upstream pixel (2 (i + du) - 1 + e,
                 * 2 (j + dv) - 1 + f) into input position (i, j); per axis t = 2 d + parity selects taps {2}, {1, 2}, {0, 1}, {0} */
} falnet_pack_up2_t;
int falnet_pack_up2_batched(const falnet_pack_up2_t* descs_dev, int n, int total_blocks, int dtype, void* stream);
/* accumulate == 0: entries with groups == 1 OVERWRITE their gradient (plain stores), entries with groups > 1 add into it
 * (atomics: the caller zeroes those); accumulate != 0: every entry adds */
int falnet_wgrad_reduce_batched(const falnet_reduce_t* descs_dev, int n, int total_blocks, int accumulate, void* stream);
int falnet_bias_grad_batched(const falnet_biasgrad_t* descs_dev, int n, int total_blocks, int dtype, void* stream);
/* the same sums without atomics: every block stores its partial sums to ws[block][512] (ws_floats >= 512 * total_blocks, channels
 * <= 512), a second launch adds them to db in block order */
int falnet_bias_grad_batched_det(const falnet_biasgrad_t* descs_dev, int n, int total_blocks, int dtype, float* ws, int64_t ws_floats,
                                 void* stream);

/*
 * OIHW f32 master weights -> packed compute-dtype operands.
 *   fwd  : wf[co][tap][ci]  (CoutPad x taps x CinPad), channel groups padded per source:
 *          real input channels [0,c0) go to [0,c0), [c0,Cin) to [c0pad, ...)  (concat sources)
 *   dgrad: wd[ci][tap][co]  (CinPad' x taps x CoutPad) -- the same numbers, transposed
 */
int falnet_pack_weights(const float* w_oihw, int cout, int cin, int taps,
                        int c0_real, int c0_pad, int cin_pad_total, int cout_pad,
                        void* wf, void* wd, int dtype, void* stream);

/* planar NCHW f32 [B][C][H][W] -> NHWC `dtype` [B][H][W][Cpad] (zero padded).  Boundary of the
 * reference API: input image (FAL_netB.py:200), VGG input (loss_functions.py:36), MED-head grads. */
int falnet_nchw_to_nhwc(const float* src, void* dst, int B, int C, int H, int W, int Cpad,
                        int dtype, void* stream);
/* NHWC `dtype` [B][H][W][Cpad] -> planar NCHW f32 [B][C][H][W] (first C channels) */
int falnet_nhwc_to_nchw(const void* src, float* dst, int B, int C, int H, int W, int Cpad,
                        int dtype, void* stream);
/* Adjoint of the fused nearest upsample (F.interpolate backward, FAL_netB.py:58):
 * gsrc[b, sy, sx, c] = (sum over virtual pixels mapping to (sy,sx) of gup[b, vy, vx, c]) * elu'(actout) */
int falnet_upsample_bwd(const void* gup, void* gsrc, const void* actout, int B, int IH, int IW,
                        int H, int W, int C, int dtype, void* stream);
/* Weight gradient of a 3x3 pad-1 conv (stride 1 or 2) with respect to an input plane that is CONSTANT per sample -- the `flow` input of
 * conv1 (FAL_netB.py:208-209, :101): dW[co][ky][kx] += sum_b plane[b * plane_stride] * sum over the output pixels whose tap (ky,kx) is in
 * bounds of gout[b][i][j][co].  gout NHWC `dtype` [B][TH][TW][gC]; grad = &dW_oihw[0][ci][0][0] of the f32 gradient, grad_co_stride = Cin * 9;
 * ws: B * 9 * gC floats, ZERO on entry and left zero.  Adds with f32 atomics (refused in deterministic mode). */
int falnet_wgrad_const_plane(const void* gout, const void* plane, int64_t plane_stride, float* grad, int64_t grad_co_stride, float* ws,
                             int B, int TH, int TW, int gC, int cout, int IH, int IW, int stride, int dtype, void* stream);
/* 2x2/2 max pool on NHWC (torchvision VGG19 features[4,9,18]; loss_functions.py:21-29) and its adjoint */
int falnet_maxpool2_fwd(const void* x, void* y, int B, int H, int W, int C, int dtype, void* stream);
int falnet_maxpool2_bwd(const void* x, const void* y, const void* gy, void* gx, int B, int H, int W, int C,
                        int dtype, void* stream);
/* maxpool2_bwd: x is the (ReLU) pool input; the gradient goes to the first maximum of each window
 * (aten tie rule) and is zero where that maximum is 0 (fused relu'); H, W even. */
/* gx = g * act'(y) elementwise on NHWC, act' from the activation OUTPUT y (ELU: y>0?1:y+1) */
int falnet_act_bwd(const void* g, const void* y, void* gx, int64_t n, int kind, int dtype, void* stream);

/*
 * MED head (FAL_netB.py:216-282): per-pixel softmax over the N disparity planes, expectation ->
 * disp; plane-sweep warp of the logits by d_n (W-1)/W pixels (2-tap, zero padded), second softmax,
 * blend of the equally shifted left image -> p_im0.  All planar f32.
 *   dlog0 [B][N][H][W], left [B][3][H][W], min_disp/max_disp [B]
 *   disp [B][1][H][W] (or NULL), p_im0 [B][3][H][W] (or NULL)
 *   stats [B][4][H][W]: (max0, sum0, maxW, sumW) of the two softmaxes, kept for backward (or NULL)
 */
int falnet_med_head_fwd(const float* dlog0, const float* left, const float* min_disp, const float* max_disp,
                        float* disp, float* p_im0, float* stats, int B, int N, int H, int W, void* stream);
/* grad_dlog0 [B][N][H][W] from grad_disp / grad_p_im0 (either may be NULL) */
int falnet_med_head_bwd(const float* dlog0, const float* left, const float* min_disp, const float* max_disp,
                        const float* disp, const float* p_im0, const float* stats,
                        const float* grad_disp, const float* grad_p_im0, float* grad_dlog0,
                        int B, int N, int H, int W, void* stream);
/* the same gradient written pixel-major in the conv stack's layout: `dtype` NHWC [B][H][W][cpad], channels >= N zero
 * (cpad % 8 == 0) -- feeds the 1x1 logits conv's data / weight gradient launches without a planar round trip */
int falnet_med_head_bwd_nhwc(const float* dlog0, const float* left, const float* min_disp, const float* max_disp,
                             const float* disp, const float* p_im0, const float* stats,
                             const float* grad_disp, const float* grad_p_im0, void* grad_dlog0_nhwc, int cpad, int dtype,
                             int B, int N, int H, int W, void* stream);
/* Name of the kernel the three head entry points dispatch to for a shape (tests assert that every form is covered; same predicates as
 * the launches, 16-byte aligned operands assumed).  pass: 0 = falnet_med_head_fwd, 1 = falnet_med_head_bwd, 2 = falnet_med_head_bwd_nhwc. */
int falnet_med_head_kernel_name(int pass, int dtype, int N, int W, char* buf, int len);
/* occlusion masks (FAL_netB.py:264-273,291-292), no grad: maskR = min(1, sum_n shift_{+s_n}(softmax(dlog0)_n)),
 * maskL = min(1, sum_n shift_{-s_n}(Dprob_n)).  Dprob is rebuilt from the logits and `stats`. */
int falnet_med_masks_fwd(const float* dlog0, const float* min_disp, const float* max_disp, const float* stats,
                         float* maskL, float* maskR, int B, int N, int H, int W, void* stream);
/* FAL_netA's right mask (models/FAL_netA.py:264): the same sum, but each plane is sampled with grid_sample's default
 * align_corners=False on the align_corners=True grid (:231,:241-242) -- bilinear in x AND y at
 * (x W/(W-1) + d_n - 1/2, y H/(H-1) - 1/2), zero padding.  Overwrites maskR; `stats` from falnet_med_head_fwd. */
int falnet_med_maskr_acfalse_fwd(const float* dlog0, const float* min_disp, const float* max_disp, const float* stats,
                                 float* maskR, int B, int N, int H, int W, void* stream);

/* ---- losses (loss_functions.py) ; all write/accumulate a scalar in `out` (f32, device) ---- */
/* out[0] (+)= scale * sum(mask * |a - b|) ; mask NULL or [B][1][H][W] broadcast over C (loss_functions.py:53) */
int falnet_l1_fwd(const float* a, const float* b, const float* mask, int B, int C, int64_t HW,
                  float scale, float* out, int accumulate, void* stream);
/* ga (+)= gscale[0] * scale * mask * sign(a - b) */
int falnet_l1_bwd(const float* a, const float* b, const float* mask, int B, int C, int64_t HW,
                  float scale, const float* gscale, float* ga, int accumulate, void* stream);
/* perceptual term: out[0] (+)= scale * sum over real channels of (a-b)^2 on NHWC `dtype` (loss_functions.py:61-65) */
int falnet_mse_fwd(const void* a, const void* b, int64_t npix, int Cpad, float scale, float* out,
                   int accumulate, int dtype, void* stream);
/* ga = gscale[0] * 2 * scale * (a - b) */
int falnet_mse_bwd(const void* a, const void* b, int64_t npix, int Cpad, float scale, const float* gscale,
                   void* ga, int dtype, void* stream);
/* edge-aware smoothness on the column window [x0, x1) (loss_functions.py:70-101; crop at
 * Train_Stage1_K.py:255): out[0] (+)= scale * sum(...). img [B][3][H][W], disp [B][1][H][W] planar f32 */
int falnet_smooth_fwd(const float* img, const float* disp, int B, int H, int W, int x0, int x1,
                      float gamma, float scale, float* out, int accumulate, void* stream);
int falnet_smooth_bwd(const float* img, const float* disp, int B, int H, int W, int x0, int x1,
                      float gamma, float scale, const float* gscale, float* gdisp, int accumulate, void* stream);
/* Loss AND gradient in one pass over the operands (a training step that knows its upstream scalars before the forward runs:
 * fal_net_amd/train.py:_stage1_fused): out[0] += the loss term exactly as the *_fwd entry point with accumulate = 1 adds it; the
 * gradient exactly as the *_bwd entry point writes it (assigned, not accumulated). */
int falnet_l1_fwd_bwd(const float* a, const float* b, int B, int C, int64_t HW, float scale, float* out, const float* gscale, float* ga,
                      void* stream);
/* falnet_l1_fwd_bwd with another gradient of the same tensor added in the same pass: ga = gscale * scale * sign(a - b) + gadd (gadd may not
 * alias ga) -- the perceptual (VGG) gradient of the synthesised view joins the L1 gradient without an extra add launch */
int falnet_l1_fwd_bwd_add(const float* a, const float* b, int B, int C, int64_t HW, float scale, float* out, const float* gscale,
                          const float* gadd, float* ga, void* stream);
int falnet_mse_fwd_bwd(const void* a, const void* b, int64_t npix, int Cpad, float scale_out, float* out, float scale_grad,
                       const float* gscale, void* ga, int dtype, void* stream);
/* Three falnet_mse_fwd_bwd in ONE launch (the perceptual term's three VGG slices, loss_functions.py:61-65): tensor k has numel[k] elements (a multiple
 * of 8, pointers 32-B aligned), out += scale_out[k] * sum (a_k - b_k)^2, ga_k = gscale[0] * 2 * scale_grad[k] * (a_k - b_k).  The pointer / scalar
 * arrays are HOST arrays of three entries (read at the call). */
int falnet_mse3_fwd_bwd(const void* const* a, const void* const* b, const int64_t* numel, const float* scale_out, float* out,
                        const float* scale_grad, const float* gscale, void* const* ga, int dtype, void* stream);
int falnet_smooth_fwd_bwd(const float* img, const float* disp, int B, int H, int W, int x0, int x1, float gamma, float scale, float* out,
                          const float* gscale, float* gdisp, void* stream);
/* Tail of a fused step (Train_Stage1_K.py:258 `loss = rec_loss + a_sm * sm_loss`): S = {rec, sm} as accumulated by the *_fwd_bwd entry
 * points above -> out = {rec + a * sm, rec, sm}; S is left {0, 0} for the next step. */
int falnet_step_scalars(float* S, float a, float* out, void* stream);
/* mixing for masked perceptual input: out = m*a + (1-m)*b (loss_functions.py:55); and grad wrt a: ga = m*g */
int falnet_mask_mix(const float* a, const float* b, const float* m, float* out, int B, int C, int64_t HW, void* stream);

/* Fused flat-buffer Adam (torch.optim.Adam semantics, Train_Stage1_K.py:180): p, g, m, v f32 [n];
 * step_size = lr / (1 - b1^t), bias2 = sqrt(1 - b2^t) computed by the caller. */
int falnet_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2,
                     float eps, int step, float grad_scale, void* stream);

/* Same update with the hyper-state on the device: state = {lr, t} (f32).  The kernel uses t+1 and a one-thread
 * launch then advances t, so a captured hipGraph replays correctly.  n must be a multiple of 4. */
int falnet_adam_step_dev(float* p, const float* g, float* m, float* v, int64_t n, float* state, float b1, float b2,
                         float eps, float grad_scale, void* stream);

/* Dynamic loss scale of the f16 compute path (torch.cuda.amp.GradScaler semantics; the reference is f32-only, Train_Stage1_K.py:258-262
 * `loss.backward(); optimizer.step()`), device-resident so that no step synchronises with the host:
 *   scaler = {scale, clean steps since the last change, overflow flag, skipped steps} (f32[4], caller-initialised {S0, 0, 0, 0}).
 * falnet_grad_guard: scaler[2] = 1 when any of g[0, n) is inf / NaN (run AFTER the gradient all-reduce: every rank sees the same sum).
 * falnet_adam_step_guarded: falnet_adam_step_dev with gradients multiplied by grad_scale / scaler[0]; when scaler[2] != 0 the whole
 *   update (p, m, v, step count) is skipped.
 * falnet_loss_scale_update (after Adam): overflow -> scale = max(scale * backoff, min_scale), skipped++; else clean++ and scale =
 *   min(scale * growth, max_scale) every `interval` clean steps; clears the flag.
 * falnet_loss_seeds: seeds[i] = scaler[0] * coef[i] -- the upstream-gradient scalars (`gscale`) of the loss kernels above. */
int falnet_grad_guard(const float* g, int64_t n, float* scaler, void* stream);
int falnet_adam_step_guarded(float* p, const float* g, float* m, float* v, int64_t n, float* state, float b1, float b2,
                             float eps, float grad_scale, const float* scaler, void* stream);
int falnet_loss_scale_update(float* scaler, float growth, float backoff, int interval, float min_scale, float max_scale, void* stream);
int falnet_loss_seeds(const float* scaler, const float* coef, float* seeds, int n, void* stream);

/* Stage-2 occlusion mask (Train_Stage2_K.py:296-302): out = a * b, forced to 1 in the column window [x0, x1); planar f32
 * [B][1][H][W], no gradient */
int falnet_occlusion_mask(const float* a, const float* b, float* out, int B, int H, int W, int x0, int x1, void* stream);
/* mirror-loss weight (Train_Stage2_K.py:319-324): out = (1 - occ) / rowmax[b] inside [x0, x1), 0 outside; with it the mirror
 * loss is falnet_l1_fwd(disp, teacher_disp, mask = out, scale = 1 / (B H (x1 - x0))) */
int falnet_mirror_weight(const float* occ, const float* rowmax, float* out, int B, int H, int W, int x0, int x1, void* stream);
/* horizontal flip of planar f32 [n_rows][W] (Train_Stage2_K.py:248-253 flip grid) */
int falnet_hflip(const float* src, float* dst, int64_t n_rows, int W, void* stream);
/* per-sample max of planar f32 [B][n] -> out[B]  (F.max_pool2d(kernel=(H,W)), Train_Stage2_K.py:319) */
int falnet_rowmax(const float* src, float* out, int B, int64_t n, void* stream);

/* Small f32 matrix product C (+)= A B with element strides (A[m * sam + k * sak], B[k * sbk + n * sbn], C row-major M x N):
 * the composed logits conv Wc = W1x1 . W3x3 (FAL_netB.py:127,190; once per weight update) and the split of its gradient,
 * dW3x3 = W1x1^T dWc, dW1x1 = dWc W3x3^T.  M * N * K <= 2^32. */
int falnet_gemm_f32_small(const float* A, int64_t sam, int64_t sak, const float* B, int64_t sbk, int64_t sbn, float* C,
                          int M, int N, int K, int accumulate, void* stream);
/* Resampling of planar f32 maps [planes][H][W] -> [planes][OH][OW] times `scale` (Test_KITTI.py:287-300 ms_pp): bilinear != 0 ->
 * F.interpolate(mode='bilinear', align_corners=True), else mode='nearest' */
int falnet_resize_planar(const float* src, float* dst, int64_t planes, int H, int W, int OH, int OW, int bilinear, float scale,
                         void* stream);

/* Per-sample scalars of one step in one launch (Train_Stage1_K.py:237, FAL_netB.py:208-209): max_out = max_disp, min_out = min_disp
 * (or max_disp * mul / div when min_disp is NULL), flow[b * flow_stride] = max_disp / 100 in `dtype` (the constant input plane). */
int falnet_disp_prologue(const float* max_disp, const float* min_disp, float mul, float div, float* min_out, float* max_out,
                         void* flow, int flow_stride, int B, int dtype, void* stream);

/* ---- training-data augmentation (data_transforms.py:46-157, Train_Stage1_K.py:116-128) ---- */
/* One pass of Pillow's 8-bit bicubic resampling (Image.resize(size, BICUBIC), data_transforms.py:68) over an interleaved uint8
 * image [H][W][C]: along x (horizontal != 0: dst [H][out_size][C]) or along y (dst [out_size][W][C]).  bounds [out_size][2] =
 * (first source index, tap count), kk [out_size][ksize] = 22-bit fixed-point coefficients (host side:
 * fal_net_amd.data_transforms.resample_coeffs, the precompute_coeffs of Pillow's Resample.c).  Bit-exact with Pillow. */
int falnet_resample_u8(const uint8_t* src, uint8_t* dst, int H, int W, int C, int out_size, int horizontal,
                       const int32_t* bounds, const int32_t* kk, int ksize, void* stream);
/* crop [y1,y1+th) x [x1,x1+tw) of a uint8 [H][W][3] image, optional np.fliplr, RandomGamma / RandomBrightness /
 * RandomCBrightness with the given factors (<= 0: not applied; float64 arithmetic as numpy), ArrayToTensor, Normalize(0,255),
 * Normalize(mean,1) -> planar f32 [3][th][tw] */
int falnet_augment_normalize(const uint8_t* src, int H, int W, int x1, int y1, int th, int tw, int flip, double gamma,
                             double bright, double cb0, double cb1, double cb2, float mean0, float mean1, float mean2,
                             float* dst, void* stream);

/* ---------------------------------------------------------------------------------------------------------------------------------
 * Host launch path in C (csrc/replay.cpp).  A command = one call of a launch entry point of this header (every function that ends in
 * `void* stream`), an event record or a stream wait.  `op`: index from falnet_replay_op_index("falnet_conv2d") ..., or FALNET_CMD_RECORD /
 * FALNET_CMD_WAIT; `stream` / `event`: indices into the tables passed to falnet_replay; iarg: the pointer / integer arguments in
 * declaration order (nint of them), farg: the float / double ones (nflt), the trailing stream argument is supplied by the replay.
 * falnet_replay issues commands 0..n-1 in order and stops at the first failure (its index in *failed_at, -1 when all were issued); the
 * launches are ordinary launches -- same kernels, same streams, same dependencies as the eager sequence they were recorded from. */
#define FALNET_CMD_RECORD (-1)
#define FALNET_CMD_WAIT (-2)
typedef struct {
    int32_t op, stream, event, nint, nflt, reserved;
    uint64_t iarg[18];
    double farg[8];
} falnet_cmd_t;
#define FALNET_CMD_MAX_INT 18
#define FALNET_CMD_MAX_FLT 8
int falnet_replay_op_index(const char* name);
int falnet_replay_op_args(int op, int* nint, int* nflt);
int falnet_replay(const falnet_cmd_t* cmds, int n, void* const* streams, int nstreams, void* const* events, int nevents, int* failed_at);
/* p[0..n) = value (hipMemsetAsync / hipMemsetD32Async) and a device-to-device copy: the two aten launches of a static step, replayable */
int falnet_fill_f32(float* p, int64_t n, float value, void* stream);
int falnet_copy_bytes(void* dst, const void* src, int64_t nbytes, void* stream);
/* One wave that does nothing for `microseconds` (s_memrealtime, the 100 MHz constant clock) on `stream`: the probe of the plan's stream self-test
 * (fal_net_amd/plan.py: stream_selftest) -- two of them on two streams take ONE period when the streams sit on different hardware queues and TWO
 * when HIP has mapped both onto the same queue.  Not part of any step. */
int falnet_spin(int microseconds, void* stream);
/* The matrix-pipe rate the board SUSTAINS: 2 048 workgroups x 4 waves, each issuing iters x 16 register-resident v_mfma_f32_16x16x32 on the
 * caller's operands `ab` (64 x 8 x 64 x 8 16-bit values = 512 KiB; random data for the ceiling under real switching activity); FLOPs of one launch =
 * 2048 * 4 * iters * 16 * 16384.  bench.py's `roofline.sustained_mfma` (measurement only: no step launches it); `out` is never written on finite data. */
int falnet_mfma_probe(const void* ab, float* out, int iters, int dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif
