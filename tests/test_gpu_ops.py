"""GPU parity tests, op level: every libfalnet_hip.so kernel against the CPU oracle / torch-CPU fp32
on the same seeded inputs.  Calls go through the C-ABI (fal_net_amd.ops / _lib)."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from fal_net_amd import _lib as L  # noqa: E402
from fal_net_amd import ops  # noqa: E402
from oracle import falnet_oracle as O  # noqa: E402

DEV = "cuda"
F32_TOL = 1e-4   # north_star gate for the exact-f32 path (relative, max-norm)
BF16_TOL = 3e-2  # bf16 operands carry 8 significant bits; reported, not gated at 1e-4
F16_TOL = 4e-3   # IEEE half operands: 11 significant bits (BASELINE configs[4])
TOL = {torch.float32: F32_TOL, torch.bfloat16: BF16_TOL, torch.float16: F16_TOL}
DTYPES = [torch.float32, torch.bfloat16, torch.float16]


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))


def to_nhwc(x, dtype):
    B, C, H, W = x.shape
    out = torch.empty(B, H, W, ops.pad_c(C), dtype=dtype, device=DEV)
    xs = x.to(DEV).contiguous()
    L.check(L.lib().falnet_nchw_to_nhwc(L.ptr(xs), L.ptr(out), B, C, H, W, ops.pad_c(C), L.dtype_code(dtype), L.stream_ptr()))
    return out


def to_nchw(t, C):
    B, H, W, Cp = t.shape
    out = torch.empty(B, C, H, W, dtype=torch.float32, device=DEV)
    L.check(L.lib().falnet_nhwc_to_nchw(L.ptr(t), L.ptr(out), B, C, H, W, Cp, L.dtype_code(t.dtype), L.stream_ptr()))
    return out.cpu()


def packed(w, b, groups, stride, dtype):
    wp = torch.nn.Parameter(w.to(DEV))
    bp = None if b is None else torch.nn.Parameter(b.to(DEV))
    pc = ops.PackedConv("t", wp, bp, groups, stride)
    pc.alloc(dtype, torch.device(DEV))
    pc.pack_call()()
    return pc


@pytest.mark.parametrize("dtype", DTYPES)
def test_layout_roundtrip(dtype):
    x = torch.randn(2, 5, 7, 70)
    y = to_nchw(to_nhwc(x, dtype), 5)
    assert rel(y, x) < {torch.float32: 1e-7, torch.bfloat16: 1e-2, torch.float16: 1e-3}[dtype]
    t = to_nhwc(x, dtype)
    assert float(t[..., 5:].abs().max()) == 0.0  # padded channels are zero


CONV_CASES = [
    # B, Cin groups, Cout, H, W, stride, ksize, bias, act, residual
    (2, [3], 32, 16, 24, 1, 3, True, L.ACT_ELU, False),
    (1, [32], 32, 9, 13, 1, 3, False, L.ACT_ELU, True),
    (2, [64], 128, 12, 20, 2, 3, True, L.ACT_ELU, False),
    (1, [32], 64, 11, 15, 2, 3, True, L.ACT_ELU, False),      # odd sizes, stride 2
    (2, [64, 32], 49, 8, 16, 1, 3, False, L.ACT_NONE, False),   # concat + Cout=49
    (1, [128, 256], 256, 4, 8, 1, 3, True, L.ACT_ELU, False),   # bottleneck-like, M < tile
    (2, [64], 64, 8, 8, 1, 3, True, L.ACT_RELU, False),         # VGG style
    (1, [49], 49, 6, 40, 1, 1, True, L.ACT_NONE, False),        # 1x1
    (2, [32, 32], 64, 24, 71, 2, 3, True, L.ACT_ELU, False),    # stride 2, output wide enough for the parity-plane wgrad (odd W)
    (1, [64], 128, 16, 64, 2, 3, True, L.ACT_ELU, False),       # stride 2, even sizes, Cout 128
    (1, [128], 96, 9, 66, 2, 3, False, L.ACT_ELU, False),       # stride 2, odd H, Cout 96 (32-wide cout tiles)
    (2, [64], 64, 12, 40, 1, (3, 1), False, L.ACT_ELU, False),  # FAL_netA residual block, first conv: 3x1 (FAL_netA.py:73)
    (1, [128], 128, 9, 21, 1, (1, 3), False, L.ACT_ELU, True),  # FAL_netA residual block, second conv: 1x3 + residual (:75,:79)
]


def _khw(k):
    return (k, k) if isinstance(k, int) else k


def _pad(k):
    return (_khw(k)[0] // 2, _khw(k)[1] // 2)
# W >= 32, dense 3x3 stride 1 -> the halo-patch kernel (conv.hip: conv3x3_patch_kernel), every instantiation
PATCH_CASES = [
    (2, [32], 32, 12, 40, 1, 3, True, L.ACT_ELU, True),         # mode S, BN=32, ragged tiles
    (1, [64, 32], 49, 10, 70, 1, 3, False, L.ACT_NONE, False),  # mode S, BN=64, concat, Cout=49
    (2, [64], 64, 9, 33, 1, 3, True, L.ACT_ELU, False),         # mode P single chunk (bf16) / two chunks (f32), BN=64
    (1, [128], 256, 8, 64, 1, 3, True, L.ACT_ELU, False),       # mode P multi chunk, BN=64
    (1, [128, 256], 256, 16, 32, 1, 3, True, L.ACT_ELU, False), # mode P, two sources
    (2, [64], 128, 128, 256, 1, 3, True, L.ACT_RELU, False),    # BN=128 (>=256 tiles), single chunk in bf16
    (2, [128], 128, 128, 256, 1, 3, False, L.ACT_ELU, True),    # BN=128, double-buffered patch
    (1, [32], 96, 16, 48, 1, 3, False, L.ACT_NONE, False),      # Cout=96 (N=96 planes)
    (2, [32, 32], 49, 12, 40, 1, 3, True, L.ACT_NONE, False),   # iconv1-like: a 64-channel weight-gradient block straddles both sources
]


def _conv_inputs(case, seed=0):
    B, groups, Cout, H, W, stride, k, bias, act, res = case
    g = torch.Generator().manual_seed(seed)
    xs = [torch.randn(B, c, H, W, generator=g) for c in groups]
    cin = sum(groups)
    kh, kw = _khw(k)
    w = torch.randn(Cout, cin, kh, kw, generator=g) * (2.0 / (cin * kh * kw)) ** 0.5
    b = torch.randn(Cout, generator=g) * 0.1 if bias else None
    return xs, w, b


def _ref_conv(case, xs, w, b, addend=None):
    B, groups, Cout, H, W, stride, k, bias, act, res = case
    y = F.conv2d(torch.cat(xs, 1), w, b, stride=stride, padding=_pad(k))
    if addend is not None:
        y = y + addend
    if act == L.ACT_ELU:
        y = F.elu(y)
    elif act == L.ACT_RELU:
        y = F.relu(y)
    return y


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", CONV_CASES + PATCH_CASES)
def test_conv_forward(case, dtype):
    B, groups, Cout, H, W, stride, k, bias, act, res = case
    xs, w, b = _conv_inputs(case)
    OH, OW = (H - 1) // stride + 1, (W - 1) // stride + 1
    addend = torch.randn(B, Cout, OH, OW) if res else None
    ref = _ref_conv(case, xs, w, b, addend)
    pc = packed(w, b, groups, stride, dtype)
    srcs_t = [to_nhwc(x, dtype) for x in xs]
    out = torch.full((B, OH, OW, pc.cout_pad), float("nan"), dtype=dtype, device=DEV)
    add_t = to_nhwc(addend, dtype) if res else None
    bias_t = None
    if b is not None:
        bias_t = torch.zeros(pc.cout_pad, device=DEV)
        bias_t[:Cout] = pc.bias.data
    ops.conv_call(dtype, [ops.nhwc_src(t) for t in srcs_t], H, W, pc.wf, pc.cin_pad, ops.fwd_taps(k), pc.taps,
                  pc.cout_pad, stride, B, OH, OW, out, OH, OW, pc.cout_pad, pc.cout_pad, bias=bias_t, addend=add_t,
                  act=act)()
    torch.cuda.synchronize()
    assert torch.isfinite(out.float()).all()
    if pc.cout_pad > Cout:
        assert float(out[..., Cout:].float().abs().max()) == 0.0
    got = to_nchw(out, Cout)
    tol = TOL[dtype]
    assert rel(got, ref) < tol


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [PATCH_CASES[0], PATCH_CASES[1], PATCH_CASES[3], PATCH_CASES[4],
                                  (2, [64], 128, 40, 70, 1, 3, True, L.ACT_ELU, True), (1, [32, 64], 64, 33, 64, 1, 3, False, L.ACT_ELU, False),
                                  (3, [32, 32], 49, 21, 75, 1, 3, True, L.ACT_NONE, True), (2, [64], 64, 64, 96, 1, 3, True, L.ACT_RELU, False),
                                  # stride 2 (variant 15 = LDS-DMA form): ragged tiles / odd sizes, two sources, 8x32 minimum, Cout 96 and 256
                                  (2, [32, 32], 64, 48, 142, 2, 3, True, L.ACT_ELU, False), (1, [64], 128, 16, 64, 2, 3, True, L.ACT_ELU, False),
                                  (1, [128], 96, 33, 130, 2, 3, False, L.ACT_ELU, False), (2, [48], 256, 20, 64, 2, 3, True, L.ACT_NONE, False),
                                  (1, [32], 64, 11, 15, 2, 3, True, L.ACT_ELU, False)])
def test_conv_every_kernel_variant(case, dtype):
    """Force each kernel variant (gather, halo-patch 128/64-B chunks, single/double stage, 16x32-block forms) on the
    same launch: all must agree with torch-CPU (the autotuner may pick any of them)."""
    B, groups, Cout, H, W, stride, k, bias, act, res = case
    xs, w, b = _conv_inputs(case, seed=11)
    addend = torch.randn(B, Cout, H, W, generator=torch.Generator().manual_seed(12)) if res else None
    ref = _ref_conv(case, xs, w, b, addend)
    pc = packed(w, b, groups, stride, dtype)
    srcs_t = [to_nhwc(x, dtype) for x in xs]
    add_t = to_nhwc(addend, dtype) if res else None
    bias_t = None
    if b is not None:
        bias_t = torch.zeros(pc.cout_pad, device=DEV)
        bias_t[:Cout] = pc.bias.data
    old = ops.AUTOTUNE
    ops.AUTOTUNE = False
    try:
        OH, OW = (H + stride - 1) // stride, (W + stride - 1) // stride
        out = torch.empty(B, OH, OW, pc.cout_pad, dtype=dtype, device=DEV)
        call = ops.conv_call(dtype, [ops.nhwc_src(t) for t in srcs_t], H, W, pc.wf, pc.cin_pad, ops.fwd_taps(k), pc.taps, pc.cout_pad,
                             stride, B, OH, OW, out, OH, OW, pc.cout_pad, pc.cout_pad, bias=bias_t, addend=add_t, act=act)
    finally:
        ops.AUTOTUNE = old
    ran = []
    for variant in list(range(1, 14)) + [15, 16, 17, 20, 21, 22, 23, 24, 25]:  # 21 / 22: four rows per wave, 16-channel chunks, on 16x32 tiles (two workgroups per CU) / 32x32 tiles; 11 / 12: gather with 32 / 64 output channels per workgroup; 13: LDS-DMA double-buffered persistent; 15: LDS-DMA stride 2; 16: two-phase weight-stationary; 17 / 20: LDS-DMA on 4x32 / 8x32 tiles
        call.desc.variant = variant
        out.fill_(float("nan"))
        rc = L.lib().falnet_conv2d(call.ref, L.stream_ptr())
        if rc == -2:
            continue
        assert rc == 0, (variant, L.lib().falnet_last_error())
        got = to_nchw(out, Cout)
        assert rel(got, ref) < TOL[dtype], variant
        ran.append(variant)
    if stride == 2:
        assert 1 in ran and (15 in ran) == (dtype != torch.float32 and k == 3 and (H + 1) // 2 >= 8 and (W + 1) // 2 >= 32)
        return
    assert 1 in ran and 4 in ran and (7 in ran or H < 16) and 11 in ran
    assert (13 in ran) == (21 in ran) == (23 in ran) == (dtype != torch.float32 and H >= 16)
    assert (22 in ran) == (dtype != torch.float32 and H >= 32)
    assert (17 in ran) == (24 in ran) == (dtype != torch.float32 and H >= 4 and W >= 32)
    assert (20 in ran) == (25 in ran) == (dtype != torch.float32 and H >= 8 and W >= 32)
    assert (10 in ran) == (16 in ran) == (sum(ops.pad_c(c) for c in groups) * (4 if dtype == torch.float32 else 2) <= 128)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,groups,up,cout,H,W,mode", [
    (8, [64], (False,), 64, 128, 256, "fwd"),            # persistent walk: 512 tiles over 512 workgroups x 1 channel block, several chunks per tile
    (8, [128], (False,), 128, 64, 128, "dgrad"),         # residual addend + activation-gradient operand (the data-gradient epilogue), 2 channel blocks
    (2, [256], (True,), 128, 64, 128, "fwd"),            # deconv forward: one source at exactly half the launch size
    (2, [64, 64], (True, False), 64, 48, 96, "dgrad"),   # iconv-like: upsampled source + skip source, 16-channel chunks crossing the source boundary
    (3, [32, 96], (False, False), 49, 37, 75, "fwd"),    # ragged tile rows / columns, Cout 49 of 64, three samples over the tile walk
    (2, [64], (False,), 64, 40, 96, "pool"),             # fused 2x2 max pool (VGG slices), full map kept
    (1, [128], (False,), 128, 33, 70, "pool_only"),      # pooled map only, odd size (floor semantics)
    (2, [64], (False,), 64, 32, 64, "sum2x2"),           # 2x2 block sums x elu'(low-resolution activation): adjoint of the nearest upsample
])
def test_conv_dma2_variant21(B, groups, up, cout, H, W, mode, dtype):
    """falnet_conv2d variant 21 (conv3x3_dma2_kernel: 16x32 tiles on four waves of four rows, 16-channel chunks, two workgroups per CU) against
    torch-CPU fp32 on the operand forms of its call sites (models/FAL_netB.py:38-58,145-173 forward and stride-1 data gradients,
    loss_functions.py:21-29 VGG slices with the fused pool), and against variant 13 on the same operands."""
    g = torch.Generator().manual_seed(B * 100 + H + cout)
    cin = sum(groups)
    xs = [torch.randn(B, c, H // 2 if u else H, W // 2 if u else W, generator=g) for c, u in zip(groups, up)]
    w = torch.randn(cout, cin, 3, 3, generator=g) * (1.5 / (9 * cin) ** 0.5)
    b = torch.randn(cout, generator=g) * 0.1
    pc = packed(w, b, groups, 1, dtype)
    srcs_t = [to_nhwc(x, dtype) for x in xs]
    xr = torch.cat([F.interpolate(to_nchw(t, c), scale_factor=2, mode="nearest") if u else to_nchw(t, c) for t, c, u in zip(srcs_t, groups, up)], 1)
    wref = w.to(dtype).float()  # the rounded weights the kernel reads
    conv = F.conv2d(xr, wref, b, padding=1)
    kw = dict(bias=None, act=L.ACT_NONE)
    bias_t = torch.zeros(pc.cout_pad, device=DEV)
    bias_t[:cout] = b.to(DEV)
    out = torch.full((B, H, W, pc.cout_pad), float("nan"), dtype=dtype, device=DEV)
    pooled = None
    if mode == "fwd":
        kw = dict(bias=bias_t, act=L.ACT_ELU)
        ref = F.elu(conv)
    elif mode == "dgrad":
        add = torch.randn(B, cout, H, W, generator=g)
        y = F.elu(torch.randn(B, cout, H, W, generator=g))
        add_t, y_t = to_nhwc(add, dtype), to_nhwc(y, dtype)
        yr = to_nchw(y_t, cout)
        kw = dict(bias=None, addend=add_t, actout=y_t, actout_kind=L.ACT_ELU)
        ref = (F.conv2d(xr, wref, None, padding=1) + to_nchw(add_t, cout)) * torch.where(yr > 0, torch.ones_like(yr), yr + 1)
    elif mode in ("pool", "pool_only"):
        pooled = torch.full((B, H // 2, W // 2, pc.cout_pad), float("nan"), dtype=dtype, device=DEV)
        kw = dict(bias=bias_t, act=L.ACT_RELU, pool_out=pooled)
        ref = F.relu(conv)
        if mode == "pool_only":
            out = None
    else:
        low = F.elu(torch.randn(B, cout, H // 2, W // 2, generator=g))
        low_t = to_nhwc(low, dtype)
        lr = to_nchw(low_t, cout)
        pooled = torch.full((B, H // 2, W // 2, pc.cout_pad), float("nan"), dtype=dtype, device=DEV)
        kw = dict(pool_out=pooled, pool_mode=1, pool_actout=low_t, pool_actout_kind=L.ACT_ELU)
        ref = None
        pref = F.avg_pool2d(F.conv2d(xr, wref, None, padding=1), 2, 2) * 4 * torch.where(lr > 0, torch.ones_like(lr), lr + 1)
        out = None
    old = ops.AUTOTUNE
    ops.AUTOTUNE = False
    try:
        call = ops.conv_call(dtype, [ops.nhwc_src(t) for t in srcs_t], H, W, pc.wf, pc.cin_pad, ops.fwd_taps(3), 9, pc.cout_pad, 1, B, H, W, out,
                             H, W, pc.cout_pad, pc.cout_pad, **kw)  # (Cout = the padded count, as the plans pass it: padded channels are written as zeros)
    finally:
        ops.AUTOTUNE = old
    res = {}
    for variant in (21, 22, 23, 13):
        call.desc.variant = variant
        if out is not None:
            out.fill_(float("nan"))
        if pooled is not None:
            pooled.fill_(float("nan"))
        assert L.lib().falnet_conv2d(call.ref, L.stream_ptr()) == 0, (variant, L.lib().falnet_last_error())
        name = C.create_string_buffer(160)
        assert L.lib().falnet_conv2d_kernel_name(call.ref, name, 160) == 0
        assert (b"conv3x3_dma2_kernel" in name.value) == (variant in (21, 22)) and (b"conv3x3_dma16_kernel" in name.value) == (variant == 23), name.value
        torch.cuda.synchronize()
        res[variant] = (None if out is None else out.float().cpu(), None if pooled is None else pooled.float().cpu())
        if out is not None:
            got = to_nchw(out, cout)
            assert torch.isfinite(got).all(), variant
            assert rel(got, ref) < TOL[dtype], (variant, rel(got, ref))
            if pc.cout_pad > cout:
                assert float(out[..., cout:].float().abs().max()) == 0.0
        if pooled is not None:
            pr = pref if mode == "sum2x2" else F.max_pool2d(ref, 2, 2)
            assert rel(to_nchw(pooled, cout), pr) < TOL[dtype], (variant, mode)
            if mode == "pool":
                assert torch.equal(pooled.float(), F.max_pool2d(out.float().permute(0, 3, 1, 2), 2, 2).permute(0, 2, 3, 1)), variant
    for v in (21, 22, 23):
        for a, b2 in zip(res[v], res[13]):  # the same products summed in another order: a few ulps of the 16-bit output apart
            if a is not None:
                assert float((a - b2).abs().max()) <= 2.0 ** (-6 if dtype == torch.bfloat16 else -9) * float(b2.abs().max()), (mode, v)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,cout,H,W,mode", [
    (2, 32, 24, 64, "fwd"),       # bias + ELU
    (2, 32, 37, 70, "res"),       # residual block tail: bias-free conv + addend + ELU; ragged third strip, odd row count
    (8, 32, 24, 96, "dgrad"),     # data gradient: flipped taps of the transposed weight, addend + activation-gradient operand; many units
    (1, 17, 5, 32, "fwd"),        # fewer rows than waves (empty ranges); 17 real output channels of 32
    (1, 32, 256, 512, "dgrad"),   # the call site's own map size (one sample)
])
def test_conv_wave32_variant27(B, cout, H, W, mode, dtype):
    """falnet_conv2d variant 27 (conv3x3_wave32_kernel, csrc/conv_wave.hip: every wave streams its own strip rows, weights resident in registers) on
    the operand forms of its call sites -- conv0_1's two convolutions forward and as stride-1 data gradients (models/FAL_netB.py:38-47,100) -- against
    torch-CPU fp32 on the rounded operands and against the weight-stationary kernel (variant 10) it replaces; launches it cannot take are refused."""
    g = torch.Generator().manual_seed(B * 100 + H + cout)
    x = torch.randn(B, 32, H, W, generator=g)
    w = torch.randn(cout, 32, 3, 3, generator=g) * (1.5 / (9 * 32) ** 0.5)
    b = torch.randn(cout, generator=g) * 0.1
    pc = packed(w, b, [32], 1, dtype)
    x_t = to_nhwc(x, dtype)
    xr, wref = to_nchw(x_t, 32), w.to(dtype).float()
    bias_t = torch.zeros(pc.cout_pad, device=DEV)
    bias_t[:cout] = b.to(DEV)
    out = torch.full((B, H, W, pc.cout_pad), float("nan"), dtype=dtype, device=DEV)
    taps, weight = ops.fwd_taps(3), pc.wf
    if mode == "fwd":
        kw = dict(bias=bias_t, act=L.ACT_ELU)
        ref = F.elu(F.conv2d(xr, wref, b, padding=1))
    elif mode == "res":
        add = torch.randn(B, cout, H, W, generator=g)
        add_t = to_nhwc(add, dtype)
        kw = dict(bias=None, addend=add_t, act=L.ACT_ELU)
        ref = F.elu(F.conv2d(xr, wref, None, padding=1) + to_nchw(add_t, cout))
    else:  # x plays the upstream gradient of a cout = 32 -> cin = 32 convolution: conv_transpose with the same weight
        assert cout == 32
        add = torch.randn(B, 32, H, W, generator=g)
        y = F.elu(torch.randn(B, 32, H, W, generator=g))
        add_t, y_t = to_nhwc(add, dtype), to_nhwc(y, dtype)
        yr = to_nchw(y_t, 32)
        kw = dict(bias=None, addend=add_t, actout=y_t, actout_kind=L.ACT_ELU)
        taps, weight = ops.dgrad_taps_s1(3), pc.wd
        ref = (F.conv_transpose2d(xr, wref, None, padding=1) + to_nchw(add_t, 32)) * torch.where(yr > 0, torch.ones_like(yr), yr + 1)
    old = ops.AUTOTUNE
    ops.AUTOTUNE = False
    try:
        call = ops.conv_call(dtype, [ops.nhwc_src(x_t)], H, W, weight, 32, taps, 9, pc.cout_pad, 1, B, H, W, out, H, W, pc.cout_pad, pc.cout_pad, **kw)
    finally:
        ops.AUTOTUNE = old
    res = {}
    for variant in (27, 10):
        call.desc.variant = variant
        out.fill_(float("nan"))
        assert L.lib().falnet_conv2d(call.ref, L.stream_ptr()) == 0, (variant, L.lib().falnet_last_error())
        name = C.create_string_buffer(160)
        assert L.lib().falnet_conv2d_kernel_name(call.ref, name, 160) == 0
        assert (b"conv3x3_wave32_kernel" in name.value) == (variant == 27), name.value
        torch.cuda.synchronize()
        got = to_nchw(out, cout)
        assert torch.isfinite(out.float()).all(), variant
        assert rel(got, ref) < TOL[dtype], (variant, rel(got, ref))
        if pc.cout_pad > cout:
            assert float(out[..., cout:].float().abs().max()) == 0.0
        res[variant] = out.float().cpu()
    assert float((res[27] - res[10]).abs().max()) <= 2.0 ** (-6 if dtype == torch.bfloat16 else -9) * float(res[10].abs().max())
    # a 64-channel source is not this kernel's
    x64 = to_nhwc(torch.randn(1, 64, 8, 32), dtype)
    w64 = torch.randn(32, 64, 3, 3) * 0.05
    pc64 = packed(w64, None, [64], 1, dtype)
    o64 = torch.empty(1, 8, 32, 32, dtype=dtype, device=DEV)
    ops.AUTOTUNE = False
    try:
        c64 = ops.conv_call(dtype, [ops.nhwc_src(x64)], 8, 32, pc64.wf, 64, ops.fwd_taps(3), 9, 32, 1, 1, 8, 32, o64, 8, 32, 32, 32)
    finally:
        ops.AUTOTUNE = old
    c64.desc.variant = 27
    assert L.lib().falnet_conv2d(c64.ref, L.stream_ptr()) == -2


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,H,W", [(2, 24, 64), (1, 37, 70), (8, 32, 96), (1, 5, 16), (1, 256, 512)])
def test_conv_wave64p_variant29(B, H, W, dtype):
    """falnet_conv2d variant 29 (conv3x3_wave64p_kernel, csrc/conv_wave.hip): the data gradient of VGG19's first convolution with respect to the
    synthesised view (loss_functions.py:21: features[0] is 3 -> 64, so its adjoint is 64 -> 3 with flipped taps of the transposed weight, planar f32
    output) -- against torch-CPU conv_transpose2d on the rounded operands and against the weight-stationary kernel it replaces; ragged strips, odd
    sizes, fewer rows than waves."""
    g = torch.Generator().manual_seed(B * 100 + H)
    gout = torch.randn(B, 64, H, W, generator=g)
    w = torch.randn(64, 3, 3, 3, generator=g) * 0.1  # VGG features[0]: OIHW (64, 3, 3, 3)
    pc = packed(w, None, [3], 1, dtype)
    g_t = to_nhwc(gout, dtype)
    gr, wref = to_nchw(g_t, 64), w.to(dtype).float()
    ref = F.conv_transpose2d(gr, wref, None, padding=1)
    out = torch.full((B, 3, H, W), float("nan"), device=DEV)
    old = ops.AUTOTUNE
    ops.AUTOTUNE = False
    try:
        call = ops.conv_call(dtype, [ops.nhwc_src(g_t)], H, W, pc.wd, pc.cout_pad, ops.dgrad_taps_s1(3), 9, ops.pad_c(3), 1, B, H, W, out, H, W, 3, 0,
                             out_layout=L.OUT_PLANAR_F32)
    finally:
        ops.AUTOTUNE = old
    res = {}
    for variant in (29, 10):
        call.desc.variant = variant
        out.fill_(float("nan"))
        assert L.lib().falnet_conv2d(call.ref, L.stream_ptr()) == 0, (variant, L.lib().falnet_last_error())
        name = C.create_string_buffer(160)
        assert L.lib().falnet_conv2d_kernel_name(call.ref, name, 160) == 0
        assert (b"conv3x3_wave64p_kernel" in name.value) == (variant == 29), name.value
        torch.cuda.synchronize()
        assert torch.isfinite(out).all(), variant
        assert rel(out, ref) < TOL[dtype], (variant, rel(out, ref))
        res[variant] = out.cpu().clone()
    assert float((res[29] - res[10]).abs().max()) <= 1e-4 * float(res[10].abs().max())  # f32 outputs: only the summation order differs


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,cin,cout,H,W,keep_full", [(2, 64, 64, 16, 64, True), (1, 64, 128, 24, 40, False), (2, 128, 128, 128, 256, True),
                                                     (1, 256, 256, 16, 32, False), (1, 64, 64, 11, 37, True)])
def test_conv_fused_maxpool(B, cin, cout, H, W, keep_full, dtype):
    """pool_out: relu(conv) and nn.MaxPool2d(2,2) of it from ONE launch (VGG slices, loss_functions.py:21-29), on every
    halo-patch variant that supports it; with out=NULL only the pooled map is produced.  Odd H/W: floor semantics."""
    case = (B, [cin], cout, H, W, 1, 3, True, L.ACT_RELU, False)
    xs, w, b = _conv_inputs(case, seed=5)
    ref = _ref_conv(case, xs, w, b)
    pc = packed(w, b, [cin], 1, dtype)
    x_t = to_nhwc(xs[0], dtype)
    old = ops.AUTOTUNE
    ops.AUTOTUNE = False
    try:
        out = torch.empty(B, H, W, cout, dtype=dtype, device=DEV) if keep_full else None
        pooled = torch.empty(B, H // 2, W // 2, cout, dtype=dtype, device=DEV)
        call = ops.conv_call(dtype, [ops.nhwc_src(x_t)], H, W, pc.wf, pc.cin_pad, ops.fwd_taps(3), 9, pc.cout_pad, 1, B, H, W, out, H, W,
                             cout, cout, bias=pc.bias, act=L.ACT_RELU, pool_out=pooled)
    finally:
        ops.AUTOTUNE = old
    ran = []
    for variant in list(range(1, 11)) + [13, 16, 21, 22, 23]:
        call.desc.variant = variant
        pooled.fill_(float("nan"))
        if out is not None:
            out.fill_(float("nan"))
        rc = L.lib().falnet_conv2d(call.ref, L.stream_ptr())
        if rc == -2:
            continue
        assert rc == 0, (variant, L.lib().falnet_last_error())
        tol = TOL[dtype]
        assert rel(to_nchw(pooled, cout), F.max_pool2d(ref, 2, 2)) < tol, variant
        if out is not None:
            assert rel(to_nchw(out, cout), ref) < tol, variant
            # the pooled map is exactly the max of the stored (rounded) full-resolution values
            assert torch.equal(pooled.float(), F.max_pool2d(out.float().permute(0, 3, 1, 2), 2, 2).permute(0, 2, 3, 1)), variant
        ran.append(variant)
    assert 1 not in ran and 4 in ran and len(ran) >= 3, ran


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,cin,cout,H,W", [(2, 64, 32, 16, 64), (1, 128, 64, 24, 96), (1, 32, 96, 128, 256)])
def test_conv_fused_sum2x2(B, cin, cout, H, W, dtype):
    """pool_mode 1: the 2x2 block sums of the conv output times elu'(low-resolution activation) -- the adjoint of the exact 2x
    nearest upsampling (FAL_netB.py:58) folded into the deconv's data-gradient launch; the full-resolution map is not stored."""
    case = (B, [cin], cout, H, W, 1, 3, False, L.ACT_NONE, False)
    xs, w, b = _conv_inputs(case, seed=9)
    full = _ref_conv(case, xs, w, None)
    low_act = torch.randn(B, cout, H // 2, W // 2, generator=torch.Generator().manual_seed(3))
    low_act = torch.where(low_act > 0, low_act, torch.expm1(low_act))  # an ELU output
    ref = F.avg_pool2d(full, 2, 2) * 4 * torch.where(low_act > 0, torch.ones_like(low_act), low_act + 1)
    pc = packed(w, None, [cin], 1, dtype)
    x_t, act_t = to_nhwc(xs[0], dtype), to_nhwc(low_act, dtype)
    ref = F.avg_pool2d(full, 2, 2) * 4 * torch.where(to_nchw(act_t, cout) > 0, torch.ones_like(low_act), to_nchw(act_t, cout) + 1)
    old = ops.AUTOTUNE
    ops.AUTOTUNE = False
    try:
        pooled = torch.empty(B, H // 2, W // 2, pc.cout_pad, dtype=dtype, device=DEV)
        call = ops.conv_call(dtype, [ops.nhwc_src(x_t)], H, W, pc.wf, pc.cin_pad, ops.fwd_taps(3), 9, pc.cout_pad, 1, B, H, W, None, H, W,
                             pc.cout_pad, pc.cout_pad, pool_out=pooled, pool_mode=1, pool_actout=act_t, pool_actout_kind=L.ACT_ELU)
    finally:
        ops.AUTOTUNE = old
    ran = []
    for variant in list(range(1, 11)) + [13, 16, 21, 22, 23]:
        call.desc.variant = variant
        pooled.fill_(float("nan"))
        rc = L.lib().falnet_conv2d(call.ref, L.stream_ptr())
        if rc == -2:
            continue
        assert rc == 0, (variant, L.lib().falnet_last_error())
        assert rel(to_nchw(pooled, cout), ref) < TOL[dtype], variant
        ran.append(variant)
    assert 4 in ran and len(ran) >= 2, ran


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,H,W,cout,act", [(2, 16, 64, 32, L.ACT_ELU), (1, 9, 13, 64, L.ACT_RELU), (2, 37, 70, 32, L.ACT_NONE),
                                            (1, 75, 250, 64, L.ACT_RELU)])
def test_conv_first_layer_c3(B, H, W, cout, act, dtype):
    """falnet_conv3x3_c3: the Cin=3 first layers (FAL_netB.py:127 conv0, VGG features[0]) straight from the planar f32 image."""
    g = torch.Generator().manual_seed(B * 1000 + H * 10 + cout)
    x = torch.randn(B, 3, H, W, generator=g)
    w = torch.randn(cout, 3, 3, 3, generator=g) * 0.2
    b = torch.randn(cout, generator=g) * 0.1
    pc = packed(w, b, [3], 1, dtype)
    out = torch.full((B, H, W, cout), float("nan"), dtype=dtype, device=DEV)
    ops.conv_c3_call(dtype, x.to(DEV), pc, out, act)()
    ref = F.conv2d(x, w, b, padding=1)
    ref = F.elu(ref) if act == L.ACT_ELU else F.relu(ref) if act == L.ACT_RELU else ref
    assert rel(to_nchw(out, cout), ref) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
def test_conv_planar_output(dtype):
    case = (2, [49], 49, 6, 40, 1, 1, True, L.ACT_NONE, False)
    xs, w, b = _conv_inputs(case)
    ref = _ref_conv(case, xs, w, b)
    pc = packed(w, b, [49], 1, dtype)
    src = to_nhwc(xs[0], dtype)
    out = torch.full((2, 49, 6, 40), float("nan"), device=DEV)
    ops.conv_call(dtype, [ops.nhwc_src(src)], 6, 40, pc.wf, pc.cin_pad, ops.fwd_taps(1), 1, pc.cout_pad, 1, 2, 6, 40, out,
                  6, 40, 49, 0, out_layout=L.OUT_PLANAR_F32, bias=pc.bias)()
    assert rel(out, ref) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cout", [3, 16, 17])
def test_conv_few_output_channels_planar(cout, dtype):
    """Dense 3x3 over 64 channels into <= 17 planar f32 output channels (the VGG data gradient into the 3-channel image, loss_functions.py:
    autograd of features[0]): every applicable kernel, incl. the weight-stationary kernel's half-tile path (v_mfma_f32_16x16x32, only the first
    16-channel half of the 32-wide tile computed when Cout <= 16)."""
    B, H, W = 2, 24, 64
    case = (B, [64], cout, H, W, 1, 3, False, L.ACT_NONE, False)
    xs, w, b = _conv_inputs(case, seed=5)
    ref = _ref_conv(case, xs, w, b)
    pc = packed(w, None, [64], 1, dtype)
    src = to_nhwc(xs[0], dtype)
    out = torch.empty(B, cout, H, W, device=DEV)
    ops.AUTOTUNE = False
    call = ops.conv_call(dtype, [ops.nhwc_src(src)], H, W, pc.wf, pc.cin_pad, ops.fwd_taps(3), 9, pc.cout_pad, 1, B, H, W, out,
                         H, W, cout, 0, out_layout=L.OUT_PLANAR_F32)
    ran = []
    for variant in (1, 2, 4, 10, 16, 13, 23):  # (23: the planar-f32 instantiation of the 16x16x32 LDS-DMA kernel)
        call.desc.variant = variant
        out.fill_(float("nan"))
        rc = L.lib().falnet_conv2d(call.ref, L.stream_ptr())
        if rc == -2:
            continue
        assert rc == 0, (variant, L.lib().falnet_last_error())
        assert rel(out, ref) < TOL[dtype], variant
        ran.append(variant)
    assert 1 in ran and ((10 in ran) == (dtype != torch.float32))  # (64 f32 channels are 256 B per pixel: beyond the weight-stationary kernel)


@pytest.mark.parametrize("dtype", DTYPES)
def test_conv_fused_upsample(dtype):
    """deconv: nearest resize to an arbitrary size then conv (FAL_netB.py:57-60), incl. non-x2 ratio."""
    g = torch.Generator().manual_seed(3)
    for (h, w, IH, IW) in ((4, 6, 8, 12), (6, 12, 11, 23), (16, 24, 32, 48), (9, 20, 17, 39)):
        x = torch.randn(2, 64, h, w, generator=g)
        wt = torch.randn(32, 64, 3, 3, generator=g) * 0.06
        ref = F.elu(F.conv2d(F.interpolate(x, size=(IH, IW), mode="nearest"), wt, None, padding=1))
        pc = packed(wt, None, [64], 1, dtype)
        src = to_nhwc(x, dtype)
        out = torch.empty(2, IH, IW, 32, dtype=dtype, device=DEV)
        ops.conv_call(dtype, [ops.nhwc_src(src)], IH, IW, pc.wf, pc.cin_pad, ops.fwd_taps(3), 9, pc.cout_pad, 1, 2, IH, IW,
                      out, IH, IW, 32, 32, act=L.ACT_ELU)()
        assert rel(to_nchw(out, 32), ref) < TOL[dtype]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,cin,cout,h,w,act,with_bias", [
    (2, 64, 64, 16, 32, "elu", False),    # one tile exactly
    (1, 128, 64, 24, 80, "elu", True),    # ragged tiles both ways, two row blocks of low-res tiles
    (3, 256, 128, 5, 33, "none", True),   # minimum-ish heights, a single ragged column, 4 channel blocks, 8 chunks
    (2, 96, 49, 9, 40, "elu", True),      # Cout not a multiple of 32 (zero rows of the packed weights), 3 chunks
])
def test_conv_subpixel_deconv_forward(B, cin, cout, h, w, act, with_bias, dtype):
    """falnet_conv2d variant 18 (conv3x3_up2_dma_kernel + falnet_pack_up2_batched): the `deconv` forward (FAL_netB.py:52-58, nearest x2 then
    3x3 conv) as four 2x2 convolutions of the low-resolution map with summed weights, against F.conv2d of the upsampled map AND against
    the ordinary kernel (variant 13 / 1) on the same operands; the automatic choice must be able to run it."""
    g = torch.Generator().manual_seed(B * 1000 + cin + cout + h)
    x = torch.randn(B, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) * (0.5 / (cin ** 0.5))
    bias = torch.randn(cout, generator=g) * 0.3 if with_bias else None
    xr = x.to(dtype).float()  # what the kernels see
    ref = F.conv2d(F.interpolate(xr, scale_factor=2, mode="nearest"), wt, bias, padding=1)
    ref = F.elu(ref) if act == "elu" else ref
    wp = torch.nn.Parameter(wt.to(DEV))
    bp = None if bias is None else torch.nn.Parameter(bias.to(DEV))
    pc = ops.PackedConv("deconvT", wp, bp, [cin], 1)
    bias_t = None
    if bias is not None:  # the launches below run over the padded channel count: the bias vector is padded alike (zeros)
        bias_t = torch.zeros(ops.pad_c(cout), device=DEV)
        bias_t[:cout] = bias.to(DEV)
    pc.up2 = True
    pc.alloc(dtype, torch.device(DEV))
    assert pc.wu is not None and tuple(pc.wu.shape) == (pc.cout_pad, 16, pc.cin_pad)
    pc.pack_call()()
    ops.pack_up2_call([pc], dtype, torch.device(DEV))()
    # the packed sub-pixel weights: pair = 4 (2 py + px) + 2 a + b, row / column taps of coinciding source pixels summed in f32
    rows = {(0, 0): [0], (0, 1): [1, 2], (1, 0): [0, 1], (1, 1): [2]}
    wu_ref = torch.zeros(pc.cout_pad, 16, pc.cin_pad)
    for py in range(2):
        for px in range(2):
            for a in range(2):
                for b in range(2):
                    acc = sum(wt[:, :, ky, kx] for ky in rows[(py, a)] for kx in rows[(px, b)])
                    wu_ref[:cout, 4 * (2 * py + px) + 2 * a + b, :cin] = acc
    assert torch.equal(pc.wu.float().cpu(), wu_ref.to(dtype).float())
    src = to_nhwc(x, dtype)
    H, W = 2 * h, 2 * w
    outs = {}
    ops.AUTOTUNE = False
    base = 13 if (H >= 16 and W >= 32) else 1  # the ordinary LDS-DMA kernel where it applies (>= 16 rows), else the gather kernel
    for variant in (base, 18):
        out = torch.full((B, H, W, pc.cout_pad), float("nan"), dtype=dtype, device=DEV)
        call = ops.conv_call(dtype, [ops.nhwc_src(src)], H, W, pc.wf, pc.cin_pad, ops.fwd_taps(3), 9, pc.cout_pad, 1, B, H, W,
                             out, H, W, pc.cout_pad, pc.cout_pad, bias=bias_t, act=L.ACT_ELU if act == "elu" else L.ACT_NONE, weight_up2=pc.wu)
        call.desc.variant = variant
        rc = L.lib().falnet_conv2d(call.ref, L.stream_ptr())
        assert rc == 0, (variant, L.lib().falnet_last_error())
        torch.cuda.synchronize()
        assert torch.isfinite(out.float()).all(), variant  # every output position and padded channel written
        assert float(out[..., cout:].float().abs().max() if pc.cout_pad > cout else 0.0) == 0.0
        outs[variant] = to_nchw(out, cout)
        assert rel(outs[variant], ref) < TOL[dtype], variant
    # the two kernels round differently summed weights: same tolerance class, not bit-equal
    assert rel(outs[18], outs[base]) < 2 * TOL[dtype]
    # without the sub-pixel weights the variant is refused, not silently replaced
    call = ops.conv_call(dtype, [ops.nhwc_src(src)], H, W, pc.wf, pc.cin_pad, ops.fwd_taps(3), 9, pc.cout_pad, 1, B, H, W,
                         out, H, W, pc.cout_pad, pc.cout_pad, bias=bias_t)
    call.desc.variant = 18
    assert L.lib().falnet_conv2d(call.ref, L.stream_ptr()) != 0


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,cin,cout,h,w,with_act", [
    (2, 64, 64, 16, 32, True),     # one tile exactly
    (1, 128, 64, 24, 80, True),    # ragged tiles both ways (second tile row of 8, third tile column of 16)
    (3, 256, 128, 16, 33, False),  # a single ragged column, 4 output-channel blocks, 16 chunks
    (2, 96, 49, 17, 40, True),     # gradient channels not a multiple of 32 (49 -> 64: zero weight columns), input channels 96 (two blocks, ragged)
    (8, 64, 64, 128, 256, True),   # deconv1 at the benchmark's size
])
def test_conv_deconv_dgrad_lowres(B, cin, cout, h, w, with_act, dtype):
    """falnet_conv2d variant 26 (conv2x2_up2d_dma16_kernel + falnet_pack_up2_batched's wdd): the data gradient of a `deconv` layer (FAL_netB.py:52-58,
    nearest x2 then 3x3 conv) computed ON THE LOW-RESOLUTION GRID -- the 3x3 taps that meet one upstream pixel summed beforehand: a 4x4 / stride-2
    convolution of the upstream gradient, 16 instead of 36 tap-MACs per position -- against autograd through F.interpolate + F.conv2d AND against the
    product's other form (3x3 data gradient at 2H x 2W with the 2x2 sums in its epilogue) on the same operands."""
    g = torch.Generator().manual_seed(B * 977 + cin + 3 * cout + h)
    wt = torch.randn(cout, cin, 3, 3, generator=g) * (0.5 / (cout ** 0.5))
    gout = torch.randn(B, cout, 2 * h, 2 * w, generator=g)
    low = torch.randn(B, cin, h, w, generator=g)
    low = torch.where(low > 0, low, torch.expm1(low))  # the layer's input: an ELU output
    gr = gout.to(dtype).float()
    x = torch.zeros(B, cin, h, w, requires_grad=True)
    (F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), wt, None, padding=1) * gr).sum().backward()
    act_t = to_nhwc(low, dtype)
    lowr = to_nchw(act_t, cin)
    ref = x.grad * (torch.where(lowr > 0, torch.ones_like(lowr), lowr + 1) if with_act else 1.0)
    wp = torch.nn.Parameter(wt.to(DEV))
    pc = ops.PackedConv("deconvT", wp, None, [cin], 1)
    pc.up2 = True
    pc.alloc(dtype, torch.device(DEV))
    assert pc.wdd is not None and tuple(pc.wdd.shape) == (pc.cin_pad, 4, 4 * pc.cout_pad)
    pc.pack_call()()
    ops.pack_up2_call([pc], dtype, torch.device(DEV))()
    # the packed weights: wdd[ci][2 du + dv][e 2 Cp + f Cp + co], per axis t = 2 d + parity selects the 3x3 taps {2}, {1, 2}, {0, 1}, {0}
    sel = {0: [2], 1: [1, 2], 2: [0, 1], 3: [0]}
    wdd_ref = torch.zeros(pc.cin_pad, 4, 4 * pc.cout_pad)
    for du in range(2):
        for dv in range(2):
            for e in range(2):
                for f in range(2):
                    acc = sum(wt[:, :, ky, kx] for ky in sel[2 * du + e] for kx in sel[2 * dv + f])  # [cout][cin]
                    k0 = e * 2 * pc.cout_pad + f * pc.cout_pad
                    wdd_ref[:cin, 2 * du + dv, k0:k0 + cout] = acc.t()
    assert torch.equal(pc.wdd.float().cpu(), wdd_ref.to(dtype).float())
    g_t = to_nhwc(gout, dtype)
    gin = torch.full((B, h, w, pc.cin_pad), float("nan"), dtype=dtype, device=DEV)
    call = ops.deconv_dgrad_call(dtype, g_t, pc, B, gin, act_t if with_act else None, name="t")
    assert call.desc.variant == 26 and "up2d" in call.tag
    call()
    torch.cuda.synchronize()
    assert torch.isfinite(gin.float()).all()
    if pc.cin_pad > cin:  # padded output channels: zero weight rows -> zeros (times elu' of a zero activation = 1)
        assert float(gin[..., cin:].float().abs().max()) == 0.0
    got = to_nchw(gin, cin)
    assert rel(got, ref) < TOL[dtype], rel(got, ref)
    # the other form on the same operands: 3x3 data gradient over the 2H x 2W map, 2x2 block sums x elu' in the epilogue
    old = ops.AUTOTUNE
    ops.AUTOTUNE = False
    try:
        pooled = torch.full((B, h, w, pc.cin_pad), float("nan"), dtype=dtype, device=DEV)
        hi = ops.conv_call(dtype, [ops.nhwc_src(g_t)], 2 * h, 2 * w, pc.wd, pc.cout_pad, ops.dgrad_taps_s1(3), 9, pc.cin_pad, 1, B, 2 * h, 2 * w, None,
                           2 * h, 2 * w, pc.cin_pad, pc.cin_pad, pool_out=pooled, pool_mode=1, pool_actout=act_t if with_act else None,
                           pool_actout_kind=L.ACT_ELU if with_act else L.ACT_NONE)
    finally:
        ops.AUTOTUNE = old
    hi()
    torch.cuda.synchronize()
    assert rel(got, to_nchw(pooled, cin)) < 2 * TOL[dtype]
    # refused, not silently replaced, when the map is below one tile or the weights are the 3x3 ones
    small = torch.zeros(B, 16, 32, pc.cout_pad, dtype=dtype, device=DEV)
    with pytest.raises(ValueError):
        ops.deconv_dgrad_call(dtype, small, pc, B, torch.zeros(B, 8, 16, pc.cin_pad, dtype=dtype, device=DEV), None)


DEEP_CASES = [  # B, groups, Cout, H, W (input), stride, upsampled first source, bias, act, residual, activation-gradient operand
    (8, [512], 512, 8, 16, 1, False, True, L.ACT_ELU, False, False),    # conv5_1.conv1 at the bench size: one image per tile
    (8, [512], 512, 4, 8, 1, False, True, L.ACT_NONE, True, True),      # conv6_1.conv2: four images per tile, residual + activation gradient
    (3, [128], 64, 4, 8, 1, False, False, L.ACT_ELU, False, False),     # a partly filled tile (three of four images)
    (5, [256, 512], 256, 8, 16, 1, False, True, L.ACT_ELU, False, False),  # iconv6: two sources, K slices never straddle them
    (2, [512], 256, 8, 16, 1, True, True, L.ACT_ELU, False, False),     # deconv6: source at half size through the nearest-upsample map
    (8, [256], 512, 16, 32, 2, False, True, L.ACT_ELU, False, False),   # conv5: stride 2, 17 x 33 patch per image
    (6, [512], 512, 8, 16, 2, False, True, L.ACT_ELU, False, False),    # conv6: stride 2, four images per tile, ragged batch
    (2, [128], 49, 8, 8, 1, False, True, L.ACT_NONE, False, False),     # 64 positions per image, Cout below the 64-row block
]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", DEEP_CASES)
def test_conv_deep_levels_kernel(case, dtype):
    """falnet_conv2d variant 19 (conv3x3_deep_kernel: maps of at most 128 positions, one-shot LDS-DMA, K-slice partial tiles in scratch, ordered
    sum + epilogue in the last slice to arrive) for both slice widths against F.conv2d on the rounded operands and against the gather kernel;
    the split-K workspace (tile counters) must be all-zero again afterwards, and a second launch must give the SAME BITS (the slices are
    summed in a fixed order, whichever finishes last)."""
    B, groups, Cout, H, W, stride, ups, with_bias, act, res, actg = case
    g = torch.Generator().manual_seed(B * 100 + Cout + H + stride)
    cin = sum(groups)
    xs = [torch.randn(B, c, (H // 2 if (ups and i == 0) else H), (W // 2 if (ups and i == 0) else W), generator=g) for i, c in enumerate(groups)]
    w = torch.randn(Cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5
    b = torch.randn(Cout, generator=g) * 0.1 if with_bias else None
    OH, OW = H // stride, W // stride
    addend = torch.randn(B, Cout, OH, OW, generator=g) if res else None
    yact = torch.randn(B, Cout, OH, OW, generator=g) if actg else None
    full = [F.interpolate(x, scale_factor=2, mode="nearest") if (ups and i == 0) else x for i, x in enumerate(xs)]
    ref = F.conv2d(torch.cat([x.to(dtype).float() for x in full], 1), w.to(dtype).float(), b, stride=stride, padding=1)
    if res:
        ref = ref + addend.to(dtype).float()
    ref = F.elu(ref) if act == L.ACT_ELU else ref
    if actg:  # ELU'(x) from the stored output y: clamp(y + 1, 0, 1) (conv_epilogue.h: act_grad_from_out)
        ya = yact.to(dtype).float()
        ref = ref * torch.clamp(ya + 1, 0, 1)
    pc = packed(w, b, groups, stride, dtype)
    srcs_t = [to_nhwc(x, dtype) for x in xs]
    bias_t = None
    if b is not None:
        bias_t = torch.zeros(pc.cout_pad, device=DEV)
        bias_t[:Cout] = pc.bias.data
    add_t = to_nhwc(addend, dtype) if res else None
    act_t = to_nhwc(yact, dtype) if actg else None
    out = torch.empty(B, OH, OW, pc.cout_pad, dtype=dtype, device=DEV)
    ops.AUTOTUNE = False
    call = ops.conv_call(dtype, [ops.nhwc_src(t) for t in srcs_t], H, W, pc.wf, pc.cin_pad, ops.fwd_taps(3), pc.taps, pc.cout_pad, stride, B, OH, OW,
                         out, OH, OW, pc.cout_pad, pc.cout_pad, bias=bias_t, addend=add_t, act=act, actout=act_t, actout_kind=L.ACT_ELU if actg else L.ACT_NONE,
                         ws_owner=("test-deep", 0))
    ws = ops._splitk_workspace(torch.device(DEV, torch.cuda.current_device()), ("test-deep", 0))
    assert call.desc.splitk_ws == ws.data_ptr()
    outs = {}
    for variant, ksplit in [(1, 1), (19, pc.cin_pad // 32), (19, pc.cin_pad // 64)]:
        if variant == 19 and (ksplit == 0 or ksplit % 4 or any(gp % (pc.cin_pad // ksplit) for gp in pc.groups_pad)):
            continue  # (slice counts are multiples of 4; a 64-channel slice must not straddle two sources)
        call.desc.variant, call.desc.ksplit = variant, ksplit
        for rep in range(2):
            out.fill_(float("nan"))
            rc = L.lib().falnet_conv2d(call.ref, L.stream_ptr())
            assert rc == 0, (variant, ksplit, L.lib().falnet_last_error())
            torch.cuda.synchronize()
            assert torch.isfinite(out.float()).all(), (variant, ksplit)
            got = to_nchw(out, Cout)
            assert rel(got, ref) < TOL[dtype], (variant, ksplit, rep)
            assert float(ws.abs().max()) == 0.0, (variant, ksplit, "workspace / counters not returned to zero")
            if variant == 19:
                if rep == 1:
                    assert torch.equal(out, first), (ksplit, "run-to-run difference")
                first = out.clone()
        if pc.cout_pad > Cout:
            assert float(out[..., Cout:].float().abs().max()) == 0.0
        outs[(variant, ksplit)] = got
    assert any(v == 19 for v, _ in outs), "variant 19 did not run"
    for key, got in outs.items():
        assert rel(got, outs[(1, 1)]) < TOL[dtype], key
    # refused where it does not apply: a map the tile does not divide, and deterministic-mode semantics are covered by tests/_deterministic.py
    call.desc.variant, call.desc.ksplit = 19, 3
    assert L.lib().falnet_conv2d(call.ref, L.stream_ptr()) == -2


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", CONV_CASES[:7] + CONV_CASES[8:] + PATCH_CASES[:5] + PATCH_CASES[7:])
def test_conv_backward(case, dtype):
    """dgrad (per concat group, stride 1 and the 4 stride-2 parity launches) and wgrad/bias-grad vs autograd."""
    B, groups, Cout, H, W, stride, k, bias, act, res = case
    xs, w, b = _conv_inputs(case, seed=5)
    xs = [x.requires_grad_(True) for x in xs]
    w.requires_grad_(True)
    if b is not None:
        b.requires_grad_(True)
    y = F.conv2d(torch.cat(xs, 1), w, b, stride=stride, padding=_pad(k))
    gy = torch.randn(y.shape, generator=torch.Generator().manual_seed(6))
    y.backward(gy)
    OH, OW = y.shape[2], y.shape[3]
    pc = packed(w.detach(), None if b is None else b.detach(), groups, stride, dtype)
    g_t = to_nhwc(gy, dtype)
    tol = TOL[dtype]
    # dgrad per group
    for gi, x in enumerate(xs):
        cg = pc.groups_pad[gi]
        gin = torch.full((B, H, W, cg), float("nan"), dtype=dtype, device=DEV)
        off = sum(pc.groups_pad[:gi]) * pc.taps * pc.cout_pad
        src = [ops.nhwc_src(g_t)]
        if stride == 1:
            ops.conv_call(dtype, src, OH, OW, pc.wd, pc.cout_pad, ops.dgrad_taps_s1(k), pc.taps, cg, 1, B, H, W, gin, H, W,
                          cg, cg, weight_offset_elems=off)()
        else:
            for py in range(2):
                for px in range(2):
                    th, tw = (H - py + 1) // 2, (W - px + 1) // 2
                    ops.conv_call(dtype, src, OH, OW, pc.wd, pc.cout_pad, ops.dgrad_taps_s2(py, px), pc.taps, cg, 1, B, th,
                                  tw, gin, H, W, cg, cg, out_step=(2, 2, py, px), weight_offset_elems=off)()
        assert torch.isfinite(gin.float()).all()
        assert rel(to_nchw(gin, groups[gi]), x.grad) < tol
    # wgrad + bias grad
    ws = torch.empty(8 << 20, device=DEV)
    gw = torch.full(w.shape, float("nan"), device=DEV)
    gb = torch.full((Cout,), float("nan"), device=DEV) if b is not None else None
    srcs_t = [to_nhwc(x.detach(), dtype) for x in xs]
    call = ops.wgrad_calls(dtype, [ops.nhwc_src(t) for t in srcs_t], H, W, g_t,
                           [(dy, dx, 0) for dy, dx, _ in ops.fwd_taps(k)], stride, B, OH, OW, pc, gw, gb, ws, target_wgs=64)
    call(0)
    assert rel(gw, w.grad) < tol
    if b is not None:
        assert rel(gb, b.grad) < tol
    call(1)  # accumulate
    assert rel(gw, 2 * w.grad) < tol


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,cin,cout,H,W,nsplit", [(2, 64, 128, 16, 64, 1), (1, 128, 96, 9, 66, 2), (3, 256, 256, 31, 70, 5), (8, 64, 128, 64, 128, 16)])
def test_stride2_wgrad_row_streaming_kernel(B, cin, cout, H, W, nsplit, dtype):
    """falnet_wgrad variant 8 (wgrad3x3_rows8s2_kernel: LDS-DMA row ring, input rows de-interleaved by column parity while fetched, one gout
    fragment in a register window) against autograd of a 3x3 / stride-2 / pad-1 conv2d (FAL_netB.py:103-107), with the fused bias gradient:
    even and odd sizes, ragged 32-pixel strips, half-empty 64-channel blocks (cout 96), items that start mid-image (nsplit does not divide
    the rows), and the parity-plane kernel (variant 5) on the same launch as a second opinion."""
    g = torch.Generator().manual_seed(B * 7 + W)
    x = torch.randn(B, cin, H, W, generator=g)
    w = (torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5).requires_grad_(True)
    b = torch.zeros(cout, requires_grad=True)
    y = F.conv2d(x.to(dtype).float(), w, b, stride=2, padding=1)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy.to(dtype).float())
    OH, OW = y.shape[2], y.shape[3]
    pc = packed(w.detach(), b.detach(), [cin], 2, dtype)
    x_t, g_t = to_nhwc(x, dtype), to_nhwc(gy, dtype)
    results = {}
    for variant in (8, 5):
        d = L.Wgrad()
        ops._fill_wgrad(d, dtype, [ops.nhwc_src(x_t)], H, W, g_t, [(dy, dx, 0) for dy, dx, _ in ops.fwd_taps(3)], 2, B, OH, OW, pc)
        d.variant, d.nsplit = variant, nsplit
        ws = torch.empty(int(L.lib().falnet_wgrad_workspace_bytes(C.byref(d))) // 4, device=DEV)
        gb = torch.zeros(cout, device=DEV)
        d.partial, d.bias_grad = ws.data_ptr(), gb.data_ptr()
        assert L.lib().falnet_wgrad_fuses_bias(C.byref(d)) == 1
        L.check(L.lib().falnet_wgrad(C.byref(d), L.stream_ptr()), f"wgrad variant {variant}")
        gw = torch.full(w.shape, float("nan"), device=DEV)
        c0_real, c0_pad = pc.group_channels()
        L.check(L.lib().falnet_wgrad_reduce(L.ptr(ws), nsplit, 9, ops.pad_c(g_t.shape[-1]), pc.cin_pad, L.ptr(gw), cout, cin, c0_real, c0_pad, 0, L.stream_ptr()))
        results[variant] = (gw, gb)
        assert rel(gw, w.grad) < 3e-3 and rel(gb, b.grad) < 3e-3, variant  # (operands rounded alike: f32 accumulation order is the only difference)
    assert rel(results[8][0], results[5][0]) < 1e-4


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("ksplit", [1, 4])
@pytest.mark.parametrize("B,cin,cout,H,W", [(2, 64, 128, 12, 20), (1, 128, 256, 9, 15), (2, 32, 64, 16, 32)])
def test_stride2_dgrad_fused_launch(B, cin, cout, H, W, ksplit, dtype):
    """falnet_conv2d_multi: the four output-parity classes of a stride-2 data gradient in ONE launch (optionally split-K with one
    fused epilogue, every member accumulating into its own workspace region), with addend and activation gradient, vs autograd."""
    g = torch.Generator().manual_seed(B * H + W)
    x = torch.randn(B, cin, H, W, generator=g).requires_grad_(True)
    w = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5
    y = F.conv2d(F.elu(x), w, None, stride=2, padding=1)
    gy = torch.randn(y.shape, generator=g)
    skip = torch.randn(B, cin, H, W, generator=g)
    y.backward(gy)
    # (dgrad wrt the conv input + addend) * elu'(pre-activation), the epilogue contract of falnet_conv_t
    ref = (F.conv_transpose2d(gy, w, stride=2, padding=1, output_padding=(1 - H % 2, 1 - W % 2)) + skip) * torch.where(x > 0, torch.ones_like(x), F.elu(x) + 1.0)
    pc = packed(w, None, [cin], 2, dtype)
    OH, OW = y.shape[2], y.shape[3]
    g_t, add_t, act_t = to_nhwc(gy, dtype), to_nhwc(skip, dtype), to_nhwc(F.elu(x.detach()), dtype)
    cg = pc.groups_pad[0]
    gin = torch.full((B, H, W, cg), float("nan"), dtype=dtype, device=DEV)
    members = []
    for py in range(2):
        for px in range(2):
            th, tw = (H - py + 1) // 2, (W - px + 1) // 2
            members.append(ops.conv_call(dtype, [ops.nhwc_src(g_t)], OH, OW, pc.wd, pc.cout_pad, ops.dgrad_taps_s2(py, px), pc.taps, cg, 1, B, th,
                                         tw, gin, H, W, cg, cg, out_step=(2, 2, py, px), addend=add_t, actout=act_t, actout_kind=L.ACT_ELU,
                                         autotune=False))
    call = ops.conv_multi_call(members, ksplit=ksplit)
    for _ in range(2):  # twice: the split-K epilogue must leave the workspace zero for the next launch
        gin.fill_(float("nan"))
        call()
        torch.cuda.synchronize()
        assert rel(to_nchw(gin, cin), ref.detach()) < TOL[dtype]
    ws = ops._splitk_workspace(torch.device(DEV, torch.cuda.current_device()))
    assert float(ws.abs().max()) == 0.0


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,cin,cout,H,W", [(2, 32, 64, 64, 128), (1, 64, 128, 40, 72), (2, 33, 64, 32, 64), (1, 256, 256, 32, 64)])
def test_stride2_dgrad_dma_four_classes(B, cin, cout, H, W, dtype):
    """falnet_conv2d_multi variant 14 (conv3x3_s2d_dma_kernel): the four parity classes of a stride-2 data gradient from ONE staged
    gout patch, with addend and activation gradient -- vs autograd and vs the gather form; partial tiles (gout 20x36) included."""
    g = torch.Generator().manual_seed(B * H + W + cin)
    x = torch.randn(B, cin, H, W, generator=g).requires_grad_(True)
    w = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5
    y = F.conv2d(F.elu(x), w, None, stride=2, padding=1)
    gy = torch.randn(y.shape, generator=g)
    skip = torch.randn(B, cin, H, W, generator=g)
    ref = (F.conv_transpose2d(gy, w, stride=2, padding=1, output_padding=1) + skip) * torch.where(x > 0, torch.ones_like(x), F.elu(x) + 1.0)
    pc = packed(w, None, [cin], 2, dtype)
    OH, OW = y.shape[2], y.shape[3]
    g_t, add_t, act_t = to_nhwc(gy, dtype), to_nhwc(skip, dtype), to_nhwc(F.elu(x.detach()), dtype)
    cg = pc.groups_pad[0]
    gin = torch.full((B, H, W, cg), float("nan"), dtype=dtype, device=DEV)
    members = []
    for py in range(2):
        for px in range(2):
            members.append(ops.conv_call(dtype, [ops.nhwc_src(g_t)], OH, OW, pc.wd, pc.cout_pad, ops.dgrad_taps_s2(py, px), pc.taps, cg, 1, B, H // 2,
                                         W // 2, gin, H, W, cg, cg, out_step=(2, 2, py, px), addend=add_t, actout=act_t, actout_kind=L.ACT_ELU,
                                         autotune=False))
    ops.conv_multi_call(members, s2d=True)()
    torch.cuda.synchronize()
    got = gin.clone()
    assert rel(to_nchw(got, cin), ref.detach()) < TOL[dtype]
    assert cg == cin or float(got[..., cin:].float().abs().max()) == 0.0  # padding channels stay zero
    gin.fill_(float("nan"))
    ops.conv_multi_call(members)()
    assert rel(got.float(), gin.float()) < 2 * TOL[dtype]
    # not the canonical pattern (a member's tap table altered): rejected, not mis-computed
    bad = ops.conv_multi_call(members[:3] + [members[2]], s2d=True)
    with pytest.raises(RuntimeError):
        bad()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,H,W", [(2, 16, 64), (1, 37, 70), (1, 75, 250), (2, 37, 68), (8, 24, 96), (1, 5, 32), (1, 64, 512)])
def test_wgrad_first_layer_planar(B, H, W, dtype):
    """falnet_wgrad variant 6: conv0's weight (+ fused bias) gradient straight from the planar f32 image (no NHWC copy), through
    the batched path the plans use (slab reduce un-pads the 3 real input channels).  Widths that are a multiple of 4 take the wave-streaming
    form (csrc/wgrad_wave.hip: wgrad3x3_c3wave_kernel -- ragged last strip, ranges that cut columns, empty ranges), the others the halo-patch form."""
    g = torch.Generator().manual_seed(H * W)
    x = torch.randn(B, 3, H, W, generator=g)
    w = (torch.randn(32, 3, 3, 3, generator=g) * 0.2).requires_grad_(True)
    b = (torch.randn(32, generator=g) * 0.1).requires_grad_(True)
    go = torch.randn(B, 32, H, W, generator=g)
    (F.conv2d(x, w, b, padding=1) * go).sum().backward()
    pc = packed(w.detach(), b.detach(), [3], 1, dtype)
    xd, g_t = x.to(DEV).contiguous(), to_nhwc(go, dtype)
    gw, gb = torch.zeros(32, 3, 3, 3, device=DEV), torch.zeros(32, device=DEV)
    wb = ops.WgradBatch(dtype, torch.device(DEV))
    call = wb.add([ops.planar_src(xd)], H, W, g_t, [(dy, dx, 0) for dy, dx, _ in ops.fwd_taps(3)], 1, B, H, W, pc, gw, gb, name="wgrad conv0")
    fin = wb.finalize()
    call()
    for c in fin[0]:
        c()
    tol = BF16_TOL if dtype == torch.bfloat16 else 2e-3
    assert rel(gw, w.grad) < tol
    assert rel(gb, b.grad) < tol


def test_upsample_bwd_and_pool():
    g = torch.Generator().manual_seed(9)
    for dtype, tol in ((torch.float32, 1e-6), (torch.bfloat16, 2e-2), (torch.float16, 2e-3)):
        x = torch.randn(2, 32, 6, 12, generator=g, requires_grad=True)
        y = F.elu(x)
        up = F.interpolate(y, size=(11, 23), mode="nearest")
        gu = torch.randn(up.shape, generator=g)
        up.backward(gu)
        gsrc = torch.empty(2, 6, 12, 32, dtype=dtype, device=DEV)
        gu_t, y_t = to_nhwc(gu, dtype), to_nhwc(y.detach(), dtype)  # keep alive: raw pointers are passed
        L.check(L.lib().falnet_upsample_bwd(L.ptr(gu_t), L.ptr(gsrc), L.ptr(y_t), 2, 11, 23,
                                            6, 12, 32, L.dtype_code(dtype), L.stream_ptr()))
        assert rel(to_nchw(gsrc, 32), x.grad) < tol * 5
        # relu -> maxpool fwd/bwd
        x = torch.randn(2, 64, 8, 12, generator=g)
        if dtype != torch.float32:  # argmax routing is discontinuous: compare on the input rounded to the compute dtype
            x = x.to(dtype).float()
        x.requires_grad_(True)
        r = F.relu(x)
        p = F.max_pool2d(r, 2)
        gp = torch.randn(p.shape, generator=g)
        p.backward(gp)
        rt = to_nhwc(r.detach(), dtype)
        pt = torch.empty(2, 4, 6, 64, dtype=dtype, device=DEV)
        L.check(L.lib().falnet_maxpool2_fwd(L.ptr(rt), L.ptr(pt), 2, 8, 12, 64, L.dtype_code(dtype), L.stream_ptr()))
        assert rel(to_nchw(pt, 64), p) < tol
        gx = torch.empty_like(rt)
        gp_t = to_nhwc(gp, dtype)
        L.check(L.lib().falnet_maxpool2_bwd(L.ptr(rt), L.ptr(pt), L.ptr(gp_t), L.ptr(gx), 2, 8, 12, 64,
                                            L.dtype_code(dtype), L.stream_ptr()))
        assert rel(to_nchw(gx, 64), x.grad) < tol


# The last three rows are the wide shapes: 1280 = BASELINE configs[4] (384 x 1280, N = 96: the 512-thread strided forms and, for a 16-bit
# gradient, the wave-neighbour backward), 1242 = native KITTI width (W % 4 = 2: the first-generation kernels take it), 2100 = wider than any
# staged form.  test_med_head_cases_cover_every_head_kernel checks that the list reaches every kernel the three entry points can dispatch to.
@pytest.mark.parametrize("B,IH,IW,stride,gC,cout", [(2, 64, 128, 2, 64, 64), (3, 75, 250, 2, 64, 49), (2, 20, 36, 1, 32, 32), (1, 9, 11, 2, 32, 17)])
@pytest.mark.parametrize("dtype", DTYPES)
def test_wgrad_const_plane(B, IH, IW, stride, gC, cout, dtype):
    """Weight gradient with respect to a per-sample constant input plane (conv1's `flow` channel, FAL_netB.py:101,208-209) from nine masked
    sums of the output gradient, against autograd of F.conv2d over the broadcast plane (zero padding 1; even and odd sizes, both strides)."""
    g = torch.Generator().manual_seed(B * 1000 + IW)
    TH, TW = (IH + stride - 1) // stride, (IW + stride - 1) // stride
    f = (torch.rand(B, generator=g) * 3 + 0.5).to(dtype).float()
    gout = torch.randn(B, TH, TW, gC, generator=g).to(dtype)
    w = torch.zeros(cout, 1, 3, 3, requires_grad=True)
    y = F.conv2d(f.view(B, 1, 1, 1).expand(B, 1, IH, IW), w, stride=stride, padding=1)
    (y * gout.float()[..., :cout].permute(0, 3, 1, 2)).sum().backward()
    cin = 5  # the plane is input channel 3 of a 5-channel OIHW gradient: only that column may change
    grad = torch.full((cout, cin, 3, 3), 7.0, device=DEV)
    plane = torch.zeros(B, 32, dtype=dtype, device=DEV)
    plane[:, 0] = f.to(DEV).to(dtype)
    ws = torch.zeros(B * 9 * gC, device=DEV)
    gd = gout.to(DEV)
    for rep in range(2):  # twice: the kernel must leave its workspace zero, and it ADDS
        L.check(L.lib().falnet_wgrad_const_plane(L.ptr(gd), L.ptr(plane), plane.stride(0), L.ptr(grad[:, 3:]), cin * 9, L.ptr(ws), B, TH, TW, gC, cout,
                                                 IH, IW, stride, L.dtype_code(dtype), L.stream_ptr()))
    assert float(ws.abs().max()) == 0.0
    got = (grad[:, 3] - 7.0) / 2
    assert rel(got, w.grad[:, 0]) < 2e-5
    assert float((grad[:, [0, 1, 2, 4]] - 7.0).abs().max()) == 0.0


HEAD_CASES = [(2, 7, 6, 40, 30.0), (2, 49, 4, 128, 300.0), (1, 49, 3, 512, 300.0), (1, 96, 2, 320, 300.0), (2, 33, 5, 77, 120.0),
              (1, 96, 2, 1280, 300.0), (1, 49, 2, 1242, 300.0), (1, 7, 1, 2100, 300.0)]
HEAD_KERNELS = {"med_head_fwd_lds2_kernel", "med_head_fwd_lds2_kernel<512 threads>", "med_head_fwd_lds_kernel", "med_head_fwd_kernel",
                "med_head_bwd_kernel<planar>", "med_head_bwd_lds2_kernel", "med_head_bwd_lds2_kernel<512 threads>", "med_head_bwd_wave_kernel",
                "med_head_bwd_lds_kernel", "med_head_bwd_kernel<nhwc>"}


def head_kernel(pas, dt, N, W):
    import ctypes
    buf = ctypes.create_string_buffer(96)
    L.check(L.lib().falnet_med_head_kernel_name(pas, L.dtype_code(dt), N, W, buf, 96))
    return buf.value.decode()


def test_med_head_cases_cover_every_head_kernel():
    seen = set()
    for B, N, H, W, maxd in HEAD_CASES:
        seen.add(head_kernel(0, torch.float32, N, W))
        seen.add(head_kernel(1, torch.float32, N, W))
        seen |= {head_kernel(2, dt, N, W) for dt in (torch.float32, torch.bfloat16, torch.float16)}
    assert seen == HEAD_KERNELS, (seen ^ HEAD_KERNELS)
    # BASELINE configs[4]: 16-bit gradient on the wave-neighbour kernel, f32 and the forward on the 512-thread strided forms
    assert head_kernel(2, torch.float16, 96, 1280) == "med_head_bwd_wave_kernel" == head_kernel(2, torch.bfloat16, 96, 1280)
    assert head_kernel(2, torch.float32, 96, 1280) == "med_head_bwd_lds2_kernel<512 threads>"
    assert head_kernel(0, torch.float32, 96, 1280) == "med_head_fwd_lds2_kernel<512 threads>"
    assert head_kernel(0, torch.float32, 49, 1242) == "med_head_fwd_lds_kernel" and head_kernel(2, torch.float16, 49, 1242) == "med_head_bwd_kernel<nhwc>"


@pytest.mark.parametrize("B,N,H,W,maxd", HEAD_CASES)
def test_med_head(B, N, H, W, maxd):
    g = torch.Generator().manual_seed(N * 1000 + W)
    dlog0 = (torch.randn(B, N, H, W, generator=g) * 2.0).requires_grad_(True)
    left = torch.rand(B, 3, H, W, generator=g) - 0.43
    mx = torch.full((B, 1, 1), maxd) * (1 - 0.07 * torch.arange(B).view(B, 1, 1))
    mn = mx * 2 / 300
    out = O.med_head(dlog0, left, mn, mx, True, True, True)
    gd = torch.randn(B, 1, H, W, generator=g)
    gp = torch.randn(B, 3, H, W, generator=g)
    ((out["disp"] * gd).sum() + (out["p_im0"] * gp).sum()).backward()

    lib = L.lib()
    d0, lf = dlog0.detach().to(DEV), left.to(DEV)
    mnd, mxd = mn.reshape(-1).to(DEV), mx.reshape(-1).to(DEV)
    disp, pan, st = (torch.empty(B, 1, H, W, device=DEV), torch.empty(B, 3, H, W, device=DEV), torch.empty(B, 4, H, W, device=DEV))
    L.check(lib.falnet_med_head_fwd(L.ptr(d0), L.ptr(lf), L.ptr(mnd), L.ptr(mxd), L.ptr(disp), L.ptr(pan), L.ptr(st), B, N, H, W, L.stream_ptr()))
    assert rel(disp, out["disp"]) < F32_TOL
    assert rel(pan, out["p_im0"]) < F32_TOL
    ml, mr = torch.empty(B, 1, H, W, device=DEV), torch.empty(B, 1, H, W, device=DEV)
    L.check(lib.falnet_med_masks_fwd(L.ptr(d0), L.ptr(mnd), L.ptr(mxd), L.ptr(st), L.ptr(ml), L.ptr(mr), B, N, H, W, L.stream_ptr()))
    assert rel(ml, out["maskL"]) < F32_TOL
    assert rel(mr, out["maskR"]) < F32_TOL
    gl = torch.empty(B, N, H, W, device=DEV)
    gd_d, gp_d = gd.to(DEV), gp.to(DEV)
    L.check(lib.falnet_med_head_bwd(L.ptr(d0), L.ptr(lf), L.ptr(mnd), L.ptr(mxd), L.ptr(disp), L.ptr(pan), L.ptr(st),
                                    L.ptr(gd_d), L.ptr(gp_d), L.ptr(gl), B, N, H, W, L.stream_ptr()))
    assert rel(gl, dlog0.grad) < F32_TOL
    # the same gradient written pixel-major (what the logits conv's gradient launches consume); padding channels zero
    for dt in (torch.float32, torch.bfloat16, torch.float16):
        cp = ops.pad_c(N)
        gn = torch.full((B, H, W, cp), float("nan"), dtype=dt, device=DEV)
        L.check(lib.falnet_med_head_bwd_nhwc(L.ptr(d0), L.ptr(lf), L.ptr(mnd), L.ptr(mxd), L.ptr(disp), L.ptr(pan), L.ptr(st),
                                             L.ptr(gd_d), L.ptr(gp_d), L.ptr(gn), cp, L.dtype_code(dt), B, N, H, W, L.stream_ptr()))
        # (the NHWC form is a different kernel -- LDS-staged plane rows -- so equal up to contraction order, not bitwise)
        assert rel(gn[..., :N].float(), gl.permute(0, 2, 3, 1).float()) < {torch.float32: 2e-6, torch.bfloat16: 1e-2, torch.float16: 2e-3}[dt]
        assert rel(gn[..., :N].float().permute(0, 3, 1, 2), dlog0.grad) < {torch.float32: F32_TOL, torch.bfloat16: 1e-2, torch.float16: 2e-3}[dt]
        assert cp == N or float(gn[..., N:].float().abs().max()) == 0.0
    # disparity-only backward
    dlog0.grad = None
    (O.med_head(dlog0, left, mn, mx)["disp"] * gd).sum().backward()
    L.check(lib.falnet_med_head_bwd(L.ptr(d0), L.ptr(lf), L.ptr(mnd), L.ptr(mxd), L.ptr(disp), L.ptr(pan), L.ptr(st),
                                    L.ptr(gd_d), L.ptr(None), L.ptr(gl), B, N, H, W, L.stream_ptr()))
    assert rel(gl, dlog0.grad) < F32_TOL


def test_med_head_linearity_full_size():
    """Size-independent property at the benchmark shape: backward is linear in the upstream gradient and
    p_im0 is a convex blend (|p| <= max|left|); disp within [min_disp, max_disp]."""
    B, N, H, W = 2, 49, 256, 512
    g = torch.Generator(device="cpu").manual_seed(1)
    d0 = (torch.randn(B, N, H, W, generator=g) * 2).to(DEV)
    lf = (torch.rand(B, 3, H, W, generator=g) - 0.43).to(DEV)
    mx, mn = torch.full((B,), 300.0, device=DEV), torch.full((B,), 2.0, device=DEV)
    lib = L.lib()
    disp, pan, st = torch.empty(B, 1, H, W, device=DEV), torch.empty(B, 3, H, W, device=DEV), torch.empty(B, 4, H, W, device=DEV)
    L.check(lib.falnet_med_head_fwd(L.ptr(d0), L.ptr(lf), L.ptr(mn), L.ptr(mx), L.ptr(disp), L.ptr(pan), L.ptr(st), B, N, H, W, L.stream_ptr()))
    assert float(disp.min()) >= 2.0 - 1e-3 and float(disp.max()) <= 300.0 + 1e-3
    assert float(pan.abs().max()) <= float(lf.abs().max()) + 1e-5
    g1, g2 = torch.randn(B, 3, H, W, device=DEV), torch.randn(B, 3, H, W, device=DEV)
    outs = []
    for gp in (g1, g2, 2 * g1 - 3 * g2):
        gl = torch.empty(B, N, H, W, device=DEV)
        L.check(lib.falnet_med_head_bwd(L.ptr(d0), L.ptr(lf), L.ptr(mn), L.ptr(mx), L.ptr(disp), L.ptr(pan), L.ptr(st),
                                        L.ptr(None), L.ptr(gp), L.ptr(gl), B, N, H, W, L.stream_ptr()))
        outs.append(gl)
    assert rel(outs[2], 2 * outs[0] - 3 * outs[1]) < 1e-4


def test_losses():
    g = torch.Generator().manual_seed(4)
    lib = L.lib()
    B, H, W = 2, 12, 40
    a, b = torch.rand(B, 3, H, W, generator=g) - 0.4, torch.rand(B, 3, H, W, generator=g) - 0.4
    m = torch.rand(B, 1, H, W, generator=g)
    a.requires_grad_(True)
    out = torch.zeros(1, device=DEV)
    a_d, b_d, m_d = a.detach().to(DEV), b.to(DEV), m.to(DEV)  # keep device copies alive: raw pointers are passed
    for mask, mask_d in ((None, None), (m, m_d)):
        a.grad = None
        ref = torch.mean((1 if mask is None else mask) * (a - b).abs())
        ref.backward()
        sc = 1.0 / (B * 3 * H * W)
        L.check(lib.falnet_l1_fwd(L.ptr(a_d), L.ptr(b_d), L.ptr(mask_d), B, 3, H * W, sc, L.ptr(out), 0, L.stream_ptr()))
        assert abs(float(out) - float(ref)) < 1e-5 * abs(float(ref))
        ga = torch.empty(B, 3, H, W, device=DEV)
        L.check(lib.falnet_l1_bwd(L.ptr(a_d), L.ptr(b_d), L.ptr(mask_d), B, 3, H * W, sc, L.ptr(None), L.ptr(ga), 0, L.stream_ptr()))
        assert rel(ga, a.grad) < 1e-5
    # smoothness on a cropped window, both gammas
    img = torch.rand(B, 3, H, W, generator=g) - 0.43
    dsp = (torch.rand(B, 1, H, W, generator=g) * 30 + 2).requires_grad_(True)
    img_d, dsp_d = img.to(DEV), dsp.detach().to(DEV)
    for x0, x1, gamma in ((8, W, 2.0), (0, 32, 1.0)):
        dsp.grad = None
        ref = O.smoothness(img[:, :, :, x0:x1], dsp[:, :, :, x0:x1], gamma)
        ref.backward()
        sc = 1.0 / (B * H * (x1 - x0))
        L.check(lib.falnet_smooth_fwd(L.ptr(img_d), L.ptr(dsp_d), B, H, W, x0, x1, gamma, sc, L.ptr(out), 0, L.stream_ptr()))
        assert abs(float(out) - float(ref)) < 1e-5 * abs(float(ref))
        gd = torch.empty(B, 1, H, W, device=DEV)
        L.check(lib.falnet_smooth_bwd(L.ptr(img_d), L.ptr(dsp_d), B, H, W, x0, x1, gamma, sc, L.ptr(None), L.ptr(gd), 0, L.stream_ptr()))
        assert rel(gd, dsp.grad) < 1e-5
        # loss + gradient in one pass (the fused training step's entry point): same numbers, seeded upstream scalar
        seed, acc2, gd2 = torch.tensor([3.0], device=DEV), torch.zeros(1, device=DEV), torch.full((B, 1, H, W), float("nan"), device=DEV)
        L.check(lib.falnet_smooth_fwd_bwd(L.ptr(img_d), L.ptr(dsp_d), B, H, W, x0, x1, gamma, sc, L.ptr(acc2), L.ptr(seed), L.ptr(gd2), L.stream_ptr()))
        assert abs(float(acc2) - float(ref)) < 1e-5 * abs(float(ref)) and rel(gd2, 3.0 * dsp.grad) < 1e-5
    # L1 and MSE: loss + gradient in one pass
    a2, b2 = torch.randn(B, 3, H, W, generator=g), torch.randn(B, 3, H, W, generator=g)
    a2d, b2d = a2.to(DEV), b2.to(DEV)
    seed, acc2, ga2 = torch.tensor([0.5], device=DEV), torch.zeros(1, device=DEV), torch.full((B, 3, H, W), float("nan"), device=DEV)
    sc = 1.0 / a2.numel()
    L.check(lib.falnet_l1_fwd_bwd(L.ptr(a2d), L.ptr(b2d), B, 3, H * W, sc, L.ptr(acc2), L.ptr(seed), L.ptr(ga2), L.stream_ptr()))
    assert abs(float(acc2) - float((a2 - b2).abs().mean())) < 1e-5 * float((a2 - b2).abs().mean())
    assert rel(ga2, 0.5 * sc * torch.sign(a2 - b2)) < 1e-6
    for dtype, tol in ((torch.float32, 1e-5), (torch.bfloat16, 2e-2)):
        x, y = torch.randn(2, 64, 6, 8, generator=g), torch.randn(2, 64, 6, 8, generator=g)
        xt, yt = to_nhwc(x, dtype), to_nhwc(y, dtype)
        acc2.zero_()
        gx = torch.full_like(xt, float("nan"))
        scm = 1.0 / x.numel()
        L.check(lib.falnet_mse_fwd_bwd(L.ptr(xt), L.ptr(yt), 2 * 6 * 8, 64, 0.25 * scm, L.ptr(acc2), scm, L.ptr(seed), L.ptr(gx), L.dtype_code(dtype), L.stream_ptr()))
        assert abs(float(acc2) - 0.25 * float(((x - y) ** 2).mean())) < tol * float(((x - y) ** 2).mean())
        assert rel(to_nchw(gx, 64), 0.5 * 2 * scm * (x - y)) < tol
    # the three VGG slices' MSE (value + gradient) in ONE launch == three single launches (falnet_mse3_fwd_bwd; sizes of very different magnitude,
    # one of them smaller than a workgroup's share)
    import ctypes as _C
    for dtype, tol in ((torch.float32, 1e-5), (torch.bfloat16, 2e-2), (torch.float16, 3e-3)):
        shapes = [(2, 64, 32, 64), (2, 128, 16, 32), (1, 64, 2, 4)]
        xs, ys = [torch.randn(*sh, generator=g) for sh in shapes], [torch.randn(*sh, generator=g) for sh in shapes]
        xt, yt = [to_nhwc(t, dtype) for t in xs], [to_nhwc(t, dtype) for t in ys]
        gs3 = [torch.full_like(t, float("nan")) for t in xt]
        scs = [1.0 / t.numel() for t in xs]
        acc2.zero_()
        P3, L3, F3 = _C.c_void_p * 3, _C.c_int64 * 3, _C.c_float * 3
        L.check(lib.falnet_mse3_fwd_bwd(P3(*[t.data_ptr() for t in xt]), P3(*[t.data_ptr() for t in yt]), L3(*[t.numel() for t in xt]),
                                        F3(*[0.25 * s_ for s_ in scs]), L.ptr(acc2), F3(*scs), L.ptr(seed), P3(*[t.data_ptr() for t in gs3]),
                                        L.dtype_code(dtype), L.stream_ptr()))
        ref3 = sum(0.25 * float(((to_nchw(xa, xa.shape[3]) - to_nchw(xb, xb.shape[3])) ** 2).mean()) for xa, xb in zip(xt, yt))
        assert abs(float(acc2) - ref3) < tol * ref3
        for xa, xb, gq, sc3 in zip(xt, yt, gs3, scs):
            assert rel(gq.float(), 0.5 * 2 * sc3 * (xa.float() - xb.float())) < tol
    # mse on NHWC, flip, rowmax, mask mix
    for dtype, tol in ((torch.float32, 1e-5), (torch.bfloat16, 2e-2), (torch.float16, 3e-3)):
        x, y = torch.randn(2, 40, 5, 9, generator=g), torch.randn(2, 40, 5, 9, generator=g)
        xt, yt = to_nhwc(x, dtype), to_nhwc(y, dtype)
        sc = 1.0 / x.numel()
        L.check(lib.falnet_mse_fwd(L.ptr(xt), L.ptr(yt), 2 * 5 * 9, 64, sc, L.ptr(out), 0, L.dtype_code(dtype), L.stream_ptr()))
        assert abs(float(out) - float(((x - y) ** 2).mean())) < tol * float(((x - y) ** 2).mean())
        gx = torch.empty_like(xt)
        L.check(lib.falnet_mse_bwd(L.ptr(xt), L.ptr(yt), 2 * 5 * 9, 64, sc, L.ptr(None), L.ptr(gx), L.dtype_code(dtype), L.stream_ptr()))
        assert rel(to_nchw(gx, 40), 2 * sc * (x - y)) < tol
    fl = torch.empty(B, 3, H, W, device=DEV)
    L.check(lib.falnet_hflip(L.ptr(a_d), L.ptr(fl), B * 3 * H, W, L.stream_ptr()))
    assert torch.equal(fl.cpu(), torch.flip(a.detach(), [3]))
    rm = torch.empty(B, device=DEV)
    L.check(lib.falnet_rowmax(L.ptr(dsp_d), L.ptr(rm), B, H * W, L.stream_ptr()))
    assert torch.equal(rm.cpu(), dsp.detach().reshape(B, -1).max(1).values)
    mix = torch.empty(B, 3, H, W, device=DEV)
    L.check(lib.falnet_mask_mix(L.ptr(a_d), L.ptr(b_d), L.ptr(m_d), L.ptr(mix), B, 3, H * W, L.stream_ptr()))
    assert rel(mix, m * a.detach() + (1 - m) * b) < 1e-6


def test_adam_matches_torch():
    g = torch.Generator().manual_seed(8)
    n = 10007
    p0, gr = torch.randn(n, generator=g), torch.randn(n, generator=g) * 1e-3
    pt = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([pt], lr=1e-4, betas=(0.5, 0.999))
    npad = (n + 3) // 4 * 4
    p, m, v, gg = (torch.zeros(npad, device=DEV) for _ in range(4))
    p[:n] = p0.to(DEV)
    for step in range(1, 4):
        pt.grad = gr * step
        opt.step()
        gg[:n] = (gr * step).to(DEV)
        L.check(L.lib().falnet_adam_step(L.ptr(p), L.ptr(gg), L.ptr(m), L.ptr(v), n, 1e-4, 0.5, 0.999, 1e-8, step, 1.0, L.stream_ptr()))
    assert float((p[:n].cpu() - pt.detach()).abs().max()) < 1e-7


def test_data_augmentation_vs_reference_goldens(golden_dir):
    """GPU augmentation path (falnet_resample_u8 + falnet_augment_normalize behind fal_net_amd.data_transforms) against the
    goldens recorded from the reference's data_transforms + Pillow: the resize is bit-exact, the augmented normalised views agree
    to float32 rounding (device pow vs libm in RandomGamma)."""
    import os
    import random
    from fal_net_amd import data_transforms as DT
    g = np.load(os.path.join(golden_dir, "g8_data_aug.npz"))
    for i in range(int(g["n_resize"])):
        ow, oh = (int(v) for v in g[f"rs{i}_size"])
        got = DT.resize_bicubic_u8(torch.from_numpy(g[f"rs{i}_in"]).to(DEV), ow, oh).cpu().numpy()
        assert np.array_equal(got, g[f"rs{i}_out"]), i
    H, W, TH, TW = (int(v) for v in g["aug_shape"])
    aug = DT.StereoAugment(TH, TW)
    for k, seed in enumerate(g["aug_seeds"]):
        random.seed(int(seed))
        np.random.seed(int(seed))
        outs = aug([torch.from_numpy(g[f"aug{k}_left"]).to(DEV), torch.from_numpy(g[f"aug{k}_right"]).to(DEV)])
        for j, o in enumerate(outs):
            assert float((o.cpu() - torch.from_numpy(g[f"aug{k}_out{j}"])).abs().max()) <= 2e-6, (k, j)


def test_data_resize_full_size_properties():
    """Size-independent properties at KITTI size (375x1242 -> x1.3): constant images stay constant, output within the input range
    up to the bicubic overshoot, identity when the size does not change, and agreement with the CPU oracle on a strip."""
    from fal_net_amd import data_transforms as DT
    from oracle import data_oracle as D
    g = torch.Generator().manual_seed(5)
    img = torch.randint(0, 256, (375, 1242, 3), generator=g, dtype=torch.uint8)
    const = torch.full((375, 1242, 3), 77, dtype=torch.uint8)
    ow, oh = int(1242 * 1.3), int(375 * 1.3)
    assert int((DT.resize_bicubic_u8(const.to(DEV), ow, oh).cpu().int() - 77).abs().max()) == 0
    assert torch.equal(DT.resize_bicubic_u8(img.to(DEV), 1242, 375).cpu(), img)
    got = DT.resize_bicubic_u8(img.to(DEV), ow, oh).cpu().numpy()
    ref = D.pil_bicubic_resize_u8(img[:40].numpy(), ow, 40)  # horizontal pass only on a strip (the oracle is slow)
    hgot = DT.resize_bicubic_u8(img[:40].contiguous().to(DEV), ow, 40).cpu().numpy()
    assert np.array_equal(hgot, ref) and got.shape == (oh, ow, 3)


def _nhwc_torch(x, dtype):
    """NCHW f32 -> NHWC `dtype` with zero-padded channels, built with torch (dtypes the layout kernel does not take yet)."""
    B, C, H, W = x.shape
    out = torch.zeros(B, H, W, ops.pad_c(C), dtype=dtype, device=DEV)
    out[..., :C] = x.to(DEV).permute(0, 2, 3, 1).to(dtype)
    return out


ROWS_CASES = [
    # B, Cin groups, Cout, H, W, upsample-from (h, w) or None, bias
    (2, [64], 64, 9, 33, None, True),            # ragged strip (one valid column in the second), bias from the register window
    (1, [128, 256], 256, 16, 32, None, True),    # two sources, 6 x 4 channel tiles
    (2, [32, 32], 49, 12, 40, None, True),       # one 64-channel block straddles both sources; Cout 49 -> gC 64
    (1, [64, 32], 49, 10, 70, None, False),      # cin 96: second channel block half empty
    (2, [64], 32, 16, 40, (8, 20), False),       # fused 2x nearest upsample of the source; gC 32: cout sub-tile 1 dead
    (1, [64], 128, 37, 64, None, False),         # odd row count, ranges that cut columns at arbitrary rows
    (8, [64], 64, 24, 64, None, True),           # many units: 8-aligned split counts take the XCD-grouped block mapping
]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", ROWS_CASES)
def test_wgrad_rows(case, dtype):
    """falnet_wgrad variant 7 (row-streaming kernel, csrc/wgrad_rows.hip) against autograd of F.conv2d on the operands rounded
    to the compute dtype, for several split-K factors (ranges that start / end mid-column, empty ranges, 8-aligned counts)."""
    B, groups, Cout, H, W, up, bias = case
    g = torch.Generator().manual_seed(H * W + Cout)
    xs = [torch.randn(B, c, *(up or (H, W)), generator=g) for c in groups]
    go = torch.randn(B, Cout, H, W, generator=g)
    # reference on the rounded operands: what is left is the f32 summation order
    xr = [x.to(dtype).float() for x in xs]
    gr = go.to(dtype).float()
    w = torch.zeros(Cout, sum(groups), 3, 3, requires_grad=True)
    b = torch.zeros(Cout, requires_grad=True)
    xin = [F.interpolate(x, size=(H, W), mode="nearest") if up else x for x in xr]
    (F.conv2d(torch.cat(xin, 1), w, b, padding=1) * gr).sum().backward()
    pc = packed(w.detach(), b.detach() if bias else None, groups, 1, torch.bfloat16)
    srcs_t = [_nhwc_torch(x, dtype) for x in xs]
    g_t = _nhwc_torch(go, dtype)
    ws = torch.empty(16 << 20, device=DEV)
    taps = [(dy, dx, 0) for dy, dx, _ in ops.fwd_taps(3)]
    for nsplit in (1, 3, 8, 16, 40):
        gw = torch.full(w.shape, float("nan"), device=DEV)
        gb = torch.full((Cout,), float("nan"), device=DEV) if bias else None
        call = ops.wgrad_calls(dtype, [ops.nhwc_src(t) for t in srcs_t], H, W, g_t, taps, 1, B, H, W, pc, gw, gb, ws)
        assert call.desc.variant == 7
        call.desc.nsplit = min(nsplit, ws.numel() * 4 // (9 * ops.pad_c(Cout) * pc.cin_pad * 4))
        call(0)
        assert rel(gw, w.grad) < 2e-5, (nsplit, rel(gw, w.grad))
        if bias:
            assert rel(gb, b.grad) < 2e-5


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,cin,cout,h,w,bias", [
    (2, 64, 64, 9, 33, True),      # ragged strip, odd row count
    (1, 128, 64, 16, 40, True),    # two input-channel tiles
    (2, 256, 128, 12, 32, False),  # 4 x 2 channel tiles
    (1, 96, 49, 10, 70, True),     # gradient channels 49 -> 64, input channels 96: second block half empty
    (8, 64, 64, 24, 64, True),     # many units: 8-aligned split counts take the XCD-grouped block mapping
])
def test_wgrad_rows_deconv_lowres(B, cin, cout, h, w, bias, dtype):
    """falnet_wgrad_t::up2 (wgrad3x3_rows16_kernel<..., true>): the weight gradient of a `deconv` layer (FAL_netB.py:52-58, nearest x2 then 3x3 conv)
    on the LOW-resolution grid -- per parity class of the upstream gradient the four tap products that class meets, written as that class's share
    of the full 3x3 slab -- against autograd through F.interpolate + F.conv2d on the rounded operands, for several split counts (whole groups of
    the four classes; ranges that cut columns at arbitrary rows), AND against the high-resolution form of the same kernel."""
    g = torch.Generator().manual_seed(h * w + cout + 7 * cin)
    x = torch.randn(B, cin, h, w, generator=g)
    go = torch.randn(B, cout, 2 * h, 2 * w, generator=g)
    xr, gr = x.to(dtype).float(), go.to(dtype).float()
    wt = torch.zeros(cout, cin, 3, 3, requires_grad=True)
    bt = torch.zeros(cout, requires_grad=True)
    (F.conv2d(F.interpolate(xr, scale_factor=2, mode="nearest"), wt, bt, padding=1) * gr).sum().backward()
    pc = packed(wt.detach(), bt.detach() if bias else None, [cin], 1, torch.bfloat16)
    x_t, g_t = _nhwc_torch(x, dtype), _nhwc_torch(go, dtype)
    ws = torch.empty(16 << 20, device=DEV)
    taps = [(dy, dx, 0) for dy, dx, _ in ops.fwd_taps(3)]
    max_slabs = ws.numel() * 4 // (9 * ops.pad_c(cout) * pc.cin_pad * 4)
    for nsplit in (4, 8, 12, 40, 64):
        gw = torch.full(wt.shape, float("nan"), device=DEV)
        gb = torch.full((cout,), float("nan"), device=DEV) if bias else None
        call = ops.wgrad_calls(dtype, [ops.nhwc_src(x_t)], h, w, g_t, taps, 1, B, h, w, pc, gw, gb, ws, up2=True)
        assert call.desc.variant == 7 and call.desc.up2 == 1 and call.desc.nsplit % 4 == 0
        call.desc.nsplit = min(nsplit, max_slabs - max_slabs % 4)
        call(0)
        assert rel(gw, wt.grad) < 2e-5, (nsplit, rel(gw, wt.grad))
        if bias:
            assert rel(gb, bt.grad) < 2e-5
    # the same layer through the high-resolution form (source upsampled on the fly by the same kernel)
    gw2 = torch.full(wt.shape, float("nan"), device=DEV)
    hi = ops.wgrad_calls(dtype, [ops.nhwc_src(x_t)], 2 * h, 2 * w, g_t, taps, 1, B, 2 * h, 2 * w, pc, gw2, None, ws)
    assert hi.desc.variant == 7 and hi.desc.up2 == 0
    hi(0)
    assert rel(gw, gw2) < 2e-5
    # a split count that is not a whole number of class groups is refused
    call.desc.nsplit = 6
    assert L.lib().falnet_wgrad(C.byref(call.desc), L.stream_ptr()) != 0


WAVE_CASES = [
    # B, Cout, H, W, bias
    (2, 32, 9, 33, False),     # ragged second strip (one valid column), odd row count
    (1, 49, 12, 40, True),     # gC 64: two output-channel halves per workgroup (the logits convolution's skip group), fused bias gradient
    (2, 32, 37, 64, False),    # ranges that cut columns at arbitrary rows
    (8, 32, 24, 96, False),    # many units
    (1, 64, 5, 32, True),      # fewer rows than waves: empty ranges
]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", WAVE_CASES)
def test_wgrad_wave32(case, dtype):
    """falnet_wgrad variant 9 (wave-streaming kernel for 32-channel inputs, csrc/wgrad_wave.hip: conv0_1's two convolutions FAL_netB.py:38-47,100 and
    the skip group of the logits convolution :127) against autograd of F.conv2d on the operands rounded to the compute dtype, for several workgroup
    counts (ranges that start / end mid-column, empty ranges); a source that is not 32 channels wide is refused."""
    B, Cout, H, W, bias = case
    g = torch.Generator().manual_seed(H * W + Cout)
    x = torch.randn(B, 32, H, W, generator=g)
    go = torch.randn(B, Cout, H, W, generator=g)
    xr, gr = x.to(dtype).float(), go.to(dtype).float()
    w = torch.zeros(Cout, 32, 3, 3, requires_grad=True)
    b = torch.zeros(Cout, requires_grad=True)
    (F.conv2d(xr, w, b, padding=1) * gr).sum().backward()
    pc = packed(w.detach(), b.detach() if bias else None, [32], 1, torch.bfloat16)
    x_t, g_t = _nhwc_torch(x, dtype), _nhwc_torch(go, dtype)
    ws = torch.empty(16 << 20, device=DEV)
    taps = [(dy, dx, 0) for dy, dx, _ in ops.fwd_taps(3)]
    for nsplit in (1, 3, 8, 40, 256):
        gw = torch.full(w.shape, float("nan"), device=DEV)
        gb = torch.full((Cout,), float("nan"), device=DEV) if bias else None
        call = ops.wgrad_calls(dtype, [ops.nhwc_src(x_t)], H, W, g_t, taps, 1, B, H, W, pc, gw, gb, ws)
        assert call.desc.variant == 9, call.desc.variant
        call.desc.nsplit = min(nsplit, ws.numel() * 4 // (9 * ops.pad_c(Cout) * 32 * 4))
        call(0)
        assert rel(gw, w.grad) < 2e-5, (nsplit, rel(gw, w.grad))
        if bias:
            assert rel(gb, b.grad) < 2e-5
    bad = L.Wgrad.from_buffer_copy(call.desc)
    bad.cin_total = 64
    assert L.lib().falnet_wgrad(C.byref(bad), L.stream_ptr()) != 0


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,Cout,IH,IW", [
    (2, 64, 24, 80),    # TW 40: ragged second strip
    (1, 64, 37, 65),    # odd input size: the parity planes differ in size, the last output row / column meets one tap row / column only
    (8, 49, 32, 128),   # many units; gC 64 with 49 real channels
    (1, 64, 6, 64),     # fewer rows than workgroups: empty ranges
])
def test_wgrad_wave32_stride2(B, Cout, IH, IW, dtype):
    """falnet_wgrad variant 9, stride-2 form (conv1's image source, FAL_netB.py:101: 3x3 / stride 2 / pad 1 over a 32-channel input): four parity-plane
    waves per pixel range, each writing its own taps of the slab -- against autograd of F.conv2d(stride=2) on the rounded operands, for several
    workgroup counts."""
    TH, TW = (IH + 1) // 2, (IW + 1) // 2
    g = torch.Generator().manual_seed(IH * IW + Cout)
    x = torch.randn(B, 32, IH, IW, generator=g)
    go = torch.randn(B, Cout, TH, TW, generator=g)
    xr, gr = x.to(dtype).float(), go.to(dtype).float()
    w = torch.zeros(Cout, 32, 3, 3, requires_grad=True)
    b = torch.zeros(Cout, requires_grad=True)
    (F.conv2d(xr, w, b, stride=2, padding=1) * gr).sum().backward()
    pc = packed(w.detach(), b.detach(), [32], 2, torch.bfloat16)
    x_t, g_t = _nhwc_torch(x, dtype), _nhwc_torch(go, dtype)
    ws = torch.empty(16 << 20, device=DEV)
    taps = [(dy, dx, 0) for dy, dx, _ in ops.fwd_taps(3)]
    for nsplit in (1, 3, 8, 40, 256):
        gw = torch.full(w.shape, float("nan"), device=DEV)
        gb = torch.full((Cout,), float("nan"), device=DEV)
        call = ops.wgrad_calls(dtype, [ops.nhwc_src(x_t)], IH, IW, g_t, taps, 2, B, TH, TW, pc, gw, gb, ws)
        assert call.desc.variant == 9 and call.desc.isy == 2, (call.desc.variant, call.desc.isy)
        assert call.desc.bias_grad  # (the bias gradient is summed by the kernel: plane (0, 0)'s waves)
        call.desc.nsplit = nsplit
        call(0)
        assert rel(gw, w.grad) < 2e-5, (nsplit, rel(gw, w.grad))
        assert rel(gb, b.grad) < 2e-5


def test_mfma_probe_runs_and_rejects_bad_arguments():
    """falnet_mfma_probe (bench.py: roofline.sustained_mfma): a measurement kernel -- it must launch on both 16-bit types, leave `out` alone on finite
    data and refuse f32 / null operands / a non-positive iteration count."""
    out = torch.full((4,), 7.0, device=DEV)
    for dt in (torch.bfloat16, torch.float16):
        ab = torch.randn(64 * 8 * 64 * 8, device=DEV).to(dt)
        assert L.lib().falnet_mfma_probe(L.ptr(ab), L.ptr(out), 4, L.dtype_code(dt), L.stream_ptr()) == 0
    torch.cuda.synchronize()
    assert torch.equal(out.cpu(), torch.full((4,), 7.0))
    assert L.lib().falnet_mfma_probe(L.ptr(ab), L.ptr(out), 0, L.dtype_code(torch.bfloat16), L.stream_ptr()) != 0
    assert L.lib().falnet_mfma_probe(L.ptr(ab), L.ptr(out), 4, L.dtype_code(torch.float32), L.stream_ptr()) != 0
    assert L.lib().falnet_mfma_probe(L.ptr(None), L.ptr(out), 4, L.dtype_code(torch.bfloat16), L.stream_ptr()) != 0


def test_launch_from_a_fresh_thread_and_side_stream():
    """SURVEY 8b threading contract: launches are issued from autograd's worker thread too.  Every entry point makes the device of
    the stream it is given current on the calling thread (hipStreamGetDevice + hipSetDevice), so a brand-new thread works."""
    import threading
    x = torch.randn(4, 37, device=DEV)
    out = torch.empty_like(x)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    err = []

    def work():
        try:
            import ctypes
            L.check(L.lib().falnet_set_device(torch.cuda.current_device()))
            L.check(L.lib().falnet_hflip(L.ptr(x), L.ptr(out), 4, 37, ctypes.c_void_p(side.cuda_stream)))
        except Exception as e:  # pragma: no cover
            err.append(e)
    t = threading.Thread(target=work)
    t.start()
    t.join()
    side.synchronize()
    assert not err and torch.equal(out, x.flip(1))
    assert L.lib().falnet_set_device(10 ** 6) != 0  # errors are reported, not swallowed


def test_stage2_mask_and_mirror_kernels():
    """falnet_occlusion_mask / falnet_rowmax / falnet_mirror_weight + the masked L1: Train_Stage2_K.py:296-302, :316-324."""
    from fal_net_amd import loss_functions as LF
    g = torch.Generator().manual_seed(3)
    B, H, W = 3, 9, 40
    a, b = torch.rand(B, 1, H, W, generator=g), torch.rand(B, 1, H, W, generator=g)
    c2, c8 = int(0.2 * W), int(0.8 * W)
    ref = a * b
    ref[:, :, :, 0:c2] = 1
    O_L = LF.occlusion_mask(a.to(DEV), b.to(DEV), 0, c2)
    assert torch.equal(O_L.cpu(), ref)
    ref_r = a * b
    ref_r[:, :, :, c8:] = 1
    assert torch.equal(LF.occlusion_mask(a.to(DEV), b.to(DEV), c8, W).cpu(), ref_r)
    disp = (torch.rand(B, 1, H, W, generator=g) * 50).requires_grad_(True)
    tdisp = torch.rand(B, 1, H, W, generator=g) * 60
    nmax = 1 / F.max_pool2d(tdisp, kernel_size=(H, W))
    want = torch.mean(nmax * (1 - ref)[:, :, :, c2:] * torch.abs(disp - tdisp)[:, :, :, c2:])
    want.backward()
    d = disp.detach().to(DEV).requires_grad_(True)
    got = LF.mirror_loss_fnc(d, tdisp.to(DEV), O_L, c2, W)
    got.backward()
    assert abs(float(got) - float(want)) < 1e-5 * abs(float(want))
    assert rel(d.grad, disp.grad) < 1e-5


def test_small_gemm_and_resize():
    """falnet_gemm_f32_small (composed logits weights and their gradient split) and falnet_resize_planar (ms_pp resampling)."""
    from fal_net_amd import inference
    g = torch.Generator().manual_seed(4)
    n, k = 49, 864
    w1, w3, gc = torch.randn(n, n, generator=g), torch.randn(n, k, generator=g), torch.randn(n, k, generator=g)
    w1d, w3d, gcd = w1.to(DEV), w3.to(DEV), gc.to(DEV)
    lib, st = L.lib(), L.stream_ptr()
    wc = torch.empty(n, k, device=DEV)
    L.check(lib.falnet_gemm_f32_small(L.ptr(w1d), n, 1, L.ptr(w3d), k, 1, L.ptr(wc), n, k, n, 0, st))
    assert rel(wc, w1 @ w3) < 1e-5
    g3 = torch.ones(n, k, device=DEV)
    L.check(lib.falnet_gemm_f32_small(L.ptr(w1d), 1, n, L.ptr(gcd), k, 1, L.ptr(g3), n, k, n, 1, st))
    assert rel(g3, 1 + w1.t() @ gc) < 1e-5
    g1 = torch.zeros(n, n, device=DEV)
    L.check(lib.falnet_gemm_f32_small(L.ptr(gcd), k, 1, L.ptr(w3d), 1, k, L.ptr(g1), n, n, k, 1, st))
    assert rel(g1, gc @ w3.t()) < 1e-5
    x = torch.randn(2, 3, 37, 124, generator=g)
    up = F.interpolate(x, scale_factor=2 / 3, mode="bilinear", align_corners=True)
    got = inference.resize_planar(x.to(DEV), up.shape[2:], bilinear=True)
    assert got.shape == up.shape and rel(got, up) < 1e-5
    dn = 1.5 * F.interpolate(up, size=(37, 124), mode="nearest")
    assert torch.equal(inference.resize_planar(up.to(DEV), (37, 124), bilinear=False, scale=1.5).cpu(), dn)
