#!/usr/bin/env python3
"""A/B of the kernels for the 32 -> 32 channel full-resolution convolutions (conv0_1, B = 8, 256 x 512): forward (+ residual, ELU) and data
gradient (addend + activation gradient) per falnet_conv2d variant, interleaved rounds in one process.  Tuning tool only."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from fal_net_amd import _lib as L, ops

DEV, B = "cuda", 8
dtype = torch.float16 if (len(sys.argv) > 1 and sys.argv[1] == "f16") else torch.bfloat16
for H, W in ((256, 512), (192, 640), (384, 1280)):
    w = torch.nn.Parameter(torch.randn(32, 32, 3, 3, device=DEV) * 0.06)
    pc = ops.PackedConv("t", w, None, [32], 1)
    pc.alloc(dtype, torch.device(DEV))
    pc.pack_call()()
    x = torch.randn(B, H, W, 32, device=DEV).to(dtype)
    add = torch.randn(B, H, W, 32, device=DEV).to(dtype)
    y = torch.randn(B, H, W, 32, device=DEV).to(dtype)
    out = torch.empty(B, H, W, 32, dtype=dtype, device=DEV)
    ops.AUTOTUNE = False
    modes = {"fwd+res+elu": (ops.fwd_taps(3), pc.wf, dict(addend=add, act=L.ACT_ELU)),
             "dgrad+add+elu'": (ops.dgrad_taps_s1(3), pc.wd, dict(addend=add, actout=y, actout_kind=L.ACT_ELU))}
    for mname, (taps, weight, kw) in modes.items():
        call = ops.conv_call(dtype, [ops.nhwc_src(x)], H, W, weight, 32, taps, 9, 32, 1, B, H, W, out, H, W, 32, 32, **kw)
        line = f"{mname:16s} @{H}x{W}:"
        for v in (10, 16, 4, 23, 27):
            call.desc.variant = v
            if L.lib().falnet_conv2d(call.ref, L.stream_ptr()) != 0:
                continue
            ts = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    L.lib().falnet_conv2d(call.ref, L.stream_ptr())
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / 5)
            t = sorted(ts)[2]
            line += f"  v{v} {t*1e3:6.1f}us {2.0*B*H*W*32*32*9/t/1e9:5.0f}TF"
        print(line, flush=True)

# the VGG adjoint's last launch: 64 -> 3 channels, planar f32 output (variant 29 against the weight-stationary kernel)
for H, W in ((256, 512), (384, 1280)):
    w = torch.nn.Parameter(torch.randn(64, 3, 3, 3, device=DEV) * 0.1)
    pc = ops.PackedConv("t", w, None, [3], 1)
    pc.alloc(dtype, torch.device(DEV))
    pc.pack_call()()
    gout = torch.randn(B, H, W, 64, device=DEV).to(dtype)
    out = torch.empty(B, 3, H, W, device=DEV)
    ops.AUTOTUNE = False
    call = ops.conv_call(dtype, [ops.nhwc_src(gout)], H, W, pc.wd, pc.cout_pad, ops.dgrad_taps_s1(3), 9, ops.pad_c(3), 1, B, H, W, out, H, W, 3, 0, out_layout=L.OUT_PLANAR_F32)
    line = f"vgg dgrad0 64->3  @{H}x{W}:"
    for v in (10, 16, 4, 1, 29):
        call.desc.variant = v
        if L.lib().falnet_conv2d(call.ref, L.stream_ptr()) != 0:
            continue
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                L.lib().falnet_conv2d(call.ref, L.stream_ptr())
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 5)
        line += f"  v{v} {sorted(ts)[2]*1e3:6.1f}us"
    print(line, flush=True)
