#!/usr/bin/env python3
"""Per-layer A/B of the weight-gradient kernels on FAL_netB's 32-channel-input layers (B = 8, 256 x 512): the halo-patch kernels of rounds 1-5
against the wave-streaming kernel (falnet_wgrad variant 9, csrc/wgrad_wave.hip), interleaved rounds in ONE process; slab reduce timed separately.
Tuning tool only (FALNET_AB=1 python tools/bench_wgrad32.py [f16])."""
import os, sys
os.environ["FALNET_AB"] = "1"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from fal_net_amd import _lib as L, ops

DEV, B = "cuda", 8
dtype = torch.float16 if (len(sys.argv) > 1 and sys.argv[1] == "f16") else torch.bfloat16
LAYERS = [("conv0_1 32->32 @256x512", 32, 256, 512, 1), ("logits[skip] 32->49 @256x512", 49, 256, 512, 1), ("32->32 @192x640", 32, 192, 640, 1),
          ("conv1 32->64 s2 @256x512", 64, 256, 512, 2), ("conv1 32->64 s2 @384x1280", 64, 384, 1280, 2)]
taps = [(dy, dx, 0) for dy, dx, _ in ops.fwd_taps(3)]
wgs_list = [int(x) for x in os.environ.get("WAVE_WGS", "128,256").split(",")]
for name, cout, H, W, stride in LAYERS:
    TH, TW = (H + stride - 1) // stride, (W + stride - 1) // stride
    w = torch.nn.Parameter(torch.randn(cout, 32, 3, 3, device=DEV) * 0.05)
    pc = ops.PackedConv("t", w, None, [32], stride)
    x = torch.randn(B, H, W, 32, device=DEV).to(dtype)
    gout = torch.randn(B, TH, TW, pc.cout_pad, device=DEV).to(dtype)
    ws = torch.empty(40 << 20, device=DEV)
    gw = torch.empty_like(w)
    srcs = [ops.nhwc_src(x)]
    cands = []
    os.environ["FALNET_WGRAD_WAVE"] = "0"
    c_old = ops.wgrad_calls(dtype, srcs, H, W, gout, taps, stride, B, TH, TW, pc, gw, None, ws)
    cands.append((f"old v{c_old.desc.variant} n{c_old.desc.nsplit}", c_old))
    os.environ["FALNET_WGRAD_WAVE"] = "1"
    for wgs in wgs_list:
        ops._WGRAD_WAVE_WGS = wgs
        c = ops.wgrad_calls(dtype, srcs, H, W, gout, taps, stride, B, TH, TW, pc, gw, None, ws)
        assert c.desc.variant == 9
        cands.append((f"wave{wgs} n{c.desc.nsplit}", c))
    flops = 2.0 * B * TH * TW * cout * 32 * 9
    times, full = {k: [] for k, _ in cands}, {k: [] for k, _ in cands}
    ref = None
    lib = L.lib()
    for rnd in range(6):
        for k, c in cands:
            refd = c.desc
            def run():
                L.check(lib.falnet_wgrad(refd, L.stream_ptr()), "wgrad")
            run()
            e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            e0.record()
            for _ in range(5):
                run()
            e1.record()
            for _ in range(5):
                c(0)  # launch + slab reduce
            e2.record()
            torch.cuda.synchronize()
            if rnd > 0:
                times[k].append(e0.elapsed_time(e1) / 5)
                full[k].append(e1.elapsed_time(e2) / 5)
            if rnd == 0:
                if ref is None:
                    ref = gw.clone()
                else:
                    err = float((gw - ref).abs().max() / ref.abs().max())
                    assert err < 2e-2, (name, k, err)
    line = f"{name:30s}"
    for k, _ in cands:
        t, f = sorted(times[k])[len(times[k]) // 2], sorted(full[k])[len(full[k]) // 2]
        line += f" | {k:14s} {t*1e3:6.1f}us {flops/t/1e9:5.0f}TF +reduce {f*1e3:6.1f}us"
    print(line, flush=True)

# first layer (conv0: planar f32 image, 3 -> 32): the wave-streaming form (IW % 4 == 0) against the halo-patch form (forced through the deterministic switch)
for H, W in ((256, 512), (384, 1280)):
    w = torch.nn.Parameter(torch.randn(32, 3, 3, 3, device=DEV) * 0.05)
    b = torch.nn.Parameter(torch.zeros(32, device=DEV))
    pc = ops.PackedConv("t", w, b, [3], 1)
    img = torch.randn(B, 3, H, W, device=DEV)
    gout = torch.randn(B, H, W, 32, device=DEV).to(dtype)
    ws = torch.empty(40 << 20, device=DEV)
    gw, gb = torch.empty_like(w), torch.zeros(32, device=DEV)
    res = {}
    for label, det in (("patch", True), ("wave", False)):
        ops.DETERMINISTIC = det
        c = ops.wgrad_calls(dtype, [ops.planar_src(img)], H, W, gout, taps, 1, B, H, W, pc, gw, gb, ws)
        L.lib().falnet_set_deterministic(1 if det else 0)
        refd = c.desc
        if det:
            refd.bias_grad = None
        ts = []
        for rnd in range(6):
            L.check(L.lib().falnet_wgrad(refd, L.stream_ptr()), "wgrad")
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                L.check(L.lib().falnet_wgrad(refd, L.stream_ptr()), "wgrad")
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 5)
        res[label] = (sorted(ts)[3], refd.nsplit)
    L.lib().falnet_set_deterministic(0)
    ops.DETERMINISTIC = False
    print(f"conv0 3->32 @{H}x{W}: " + " | ".join(f"{k} n{n} {t*1e3:6.1f}us" for k, (t, n) in res.items()), flush=True)
