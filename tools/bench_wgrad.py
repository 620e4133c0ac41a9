#!/usr/bin/env python3
"""Per-layer A/B of the weight-gradient kernels on the FAL_netB layer shapes (B=8, 256x512): the plan's previous choice
(patch kernels) against the row-streaming kernel (variant 7), interleaved rounds in ONE process, optional cache flush between
launches (COLD=1: the in-situ state of a training step).  Tuning tool only."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from fal_net_amd import _lib as L, ops

DEV, B = "cuda", 8
dtype = torch.float16 if (len(sys.argv) > 1 and sys.argv[1] == "f16") else torch.bfloat16
LAYERS = [  # name, groups, Cout, H, W, upsample_from  (BENCH_LAYERS: comma-separated substrings)
    ("deconv1 64->64 @256x512 (up)", [64], 64, 256, 512, (128, 256)),
    ("logits 64+32->49 @256x512", [64, 32], 49, 256, 512, None),
    ("conv1_1 64->64 @128x256", [64], 64, 128, 256, None),
    ("iconv2 64+64->64 @128x256", [64, 64], 64, 128, 256, None),
    ("deconv2 128->64 @128x256 (up)", [128], 64, 128, 256, (64, 128)),
    ("conv2_1 128->128 @64x128", [128], 128, 64, 128, None),
    ("iconv3 128+128->128 @64x128", [128, 128], 128, 64, 128, None),
    ("deconv3 256->128 @64x128 (up)", [256], 128, 64, 128, (32, 64)),
    ("conv3_1 256->256 @32x64", [256], 256, 32, 64, None),
    ("iconv4 128+256->256 @32x64", [128, 256], 256, 32, 64, None),
    ("deconv4 256->128 @32x64 (up)", [256], 128, 32, 64, (16, 32)),
    ("conv4_1 256->256 @16x32", [256], 256, 16, 32, None),
    ("iconv5 128+256->256 @16x32", [128, 256], 256, 16, 32, None),
]
if os.environ.get("BENCH_LAYERS"):
    LAYERS = [l for l in LAYERS if any(k in l[0] for k in os.environ["BENCH_LAYERS"].split(","))]
cold = os.environ.get("COLD") == "1"
flush = torch.empty(768 << 20, dtype=torch.uint8, device=DEV) if cold else None
splits = [int(x) for x in os.environ.get("ROWS_WGS", "512").split(",")]
taps = [(dy, dx, 0) for dy, dx, _ in ops.fwd_taps(3)]
tot = {}
for name, groups, cout, H, W, up in LAYERS:
    cin = sum(groups)
    w = torch.nn.Parameter(torch.randn(cout, cin, 3, 3, device=DEV) * 0.05)
    pc = ops.PackedConv("t", w, None, groups, 1)
    srcs_t = [torch.randn(B, *(up or (H, W)), ops.pad_c(g), device=DEV).to(dtype) for g in groups]
    gout = torch.randn(B, H, W, pc.cout_pad, device=DEV).to(dtype)
    ws = torch.empty(40 << 20, device=DEV)  # 160 MB
    gw = torch.empty_like(w)
    srcs = [ops.nhwc_src(t) for t in srcs_t]
    cands = []
    if dtype == torch.bfloat16:
        os.environ["FALNET_WGRAD_ROWS"] = "0"
        c_old = ops.wgrad_calls(dtype, srcs, H, W, gout, taps, 1, B, H, W, pc, gw, None, ws)
        cands.append((f"old v{c_old.desc.variant} n{c_old.desc.nsplit}", c_old))
    os.environ["FALNET_WGRAD_ROWS"] = "1"
    for wgs in splits:
        ops._WGRAD_ROWS_WGS = wgs
        c = ops.wgrad_calls(dtype, srcs, H, W, gout, taps, 1, B, H, W, pc, gw, None, ws)
        cands.append((f"rows{wgs} n{c.desc.nsplit}", c))
    if up and (2 * up[0], 2 * up[1]) == (H, W) and len(groups) == 1:  # the deconv layer's gradient on the low-resolution grid (falnet_wgrad_t::up2)
        for wgs in splits:
            ops._WGRAD_ROWS_WGS = wgs
            c = ops.wgrad_calls(dtype, srcs, up[0], up[1], gout, taps, 1, B, up[0], up[1], pc, gw, None, ws, up2=True)
            cands.append((f"lowres{wgs} n{c.desc.nsplit}", c))
    flops = 2.0 * B * H * W * cout * cin * 9
    times = {k: [] for k, _ in cands}
    ref = None
    for rnd in range(6):
        for k, c in cands:
            lib, refd = L.lib(), c.desc
            def run():
                L.check(lib.falnet_wgrad(refd, L.stream_ptr()), "wgrad")
            run()
            n = 1 if cold else 5
            t = 0.0
            for _ in range(5 if cold else 1):
                if cold:
                    flush.fill_(1)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(n):
                    run()
                e1.record()
                torch.cuda.synchronize()
                t += e0.elapsed_time(e1) / n
            if rnd > 0:
                times[k].append(t / (5 if cold else 1))
            if rnd == 0:
                c(0)
                torch.cuda.synchronize()
                if ref is None:
                    ref = gw.clone()
                else:
                    err = float((gw - ref).abs().max() / ref.abs().max())
                    assert err < 2e-2 or os.environ.get('NOCHECK') == '1', (name, k, err)
    line = f"{name:32s}"
    for k, _ in cands:
        t = sorted(times[k])[len(times[k]) // 2]
        tot[k.split()[0]] = tot.get(k.split()[0], 0.0) + t
        line += f" | {k:16s} {t*1e3:6.1f}us {flops/t/1e9:6.0f}TF"
    print(line, flush=True)
print("total ms:", {k: round(v, 3) for k, v in tot.items()})
