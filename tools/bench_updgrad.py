#!/usr/bin/env python3
"""Data gradient of the `deconv` layers at the benchmark's sizes (B = 8, bf16): the 3x3 form at 2H x 2W with the 2x2 sums in its epilogue
(falnet_conv2d variant 23 / autotuned) against the low-resolution form (variant 26).  Interleaved rounds in one process, HIP events."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from fal_net_amd import _lib as L, ops
DEV, dtype, B = "cuda", torch.bfloat16, int(os.environ.get("BENCH_B", "8"))
LAYERS = [("deconv1 64->64 grid 128x256", 64, 64, 128, 256), ("deconv2 128->64 grid 64x128", 128, 64, 64, 128), ("deconv3 256->128 grid 32x64", 256, 128, 32, 64),
          ("deconv4 256->128 grid 16x32", 256, 128, 16, 32)]
for name, cin, cout, h, w in LAYERS:
    wt = torch.nn.Parameter(torch.randn(cout, cin, 3, 3, device=DEV) * 0.05)
    pc = ops.PackedConv("t", wt, None, [cin], 1)
    pc.up2 = True
    pc.alloc(dtype, torch.device(DEV))
    pc.pack_call()()
    ops.pack_up2_call([pc], dtype, torch.device(DEV))()
    g = torch.randn(B, 2 * h, 2 * w, pc.cout_pad, device=DEV).to(dtype)
    act = torch.randn(B, h, w, pc.cin_pad, device=DEV).to(dtype)
    o1 = torch.empty(B, h, w, pc.cin_pad, dtype=dtype, device=DEV)
    o2 = torch.empty_like(o1)
    ops.AUTOTUNE = False
    hi = ops.conv_call(dtype, [ops.nhwc_src(g)], 2 * h, 2 * w, pc.wd, pc.cout_pad, ops.dgrad_taps_s1(3), 9, pc.cin_pad, 1, B, 2 * h, 2 * w, None, 2 * h, 2 * w,
                       pc.cin_pad, pc.cin_pad, pool_out=o1, pool_mode=1, pool_actout=act, pool_actout_kind=L.ACT_ELU)
    hi.desc.variant = 23
    lo = ops.deconv_dgrad_call(dtype, g, pc, B, o2, act)
    times = {"3x3 at 2H x 2W + 2x2 sums (v23)": [], "low-resolution grid (v26)": []}
    for rnd in range(6):
        for k, c in zip(times, (hi, lo)):
            c()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                c()
            e1.record()
            torch.cuda.synchronize()
            times[k].append(e0.elapsed_time(e1) * 1e3 / 5)
    fl = 2.0 * B * 4 * h * w * cin * cout * 9
    err = float((o1.float() - o2.float()).abs().max() / o1.float().abs().max())
    print(f"{name:34s} | " + " | ".join(f"{k} {sorted(v)[len(v) // 2]:7.1f}us {fl / sorted(v)[len(v) // 2] / 1e6:6.0f}TF(alg)" for k, v in times.items()) + f" | max diff {err:.3g}")
