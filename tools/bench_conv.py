#!/usr/bin/env python3
"""Per-layer micro-benchmark of the conv kernels (forward launches of the hot layers of FAL_netB / VGG19 at
B=8, 256x512): interleaved rounds of the kernel variants inside ONE process (cdna guide rule 24), HIP-event
timing, algorithmic TFLOP/s.  Tuning tool only."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from fal_net_amd import _lib as L, ops

DEV = "cuda"
dtype = torch.bfloat16 if (len(sys.argv) < 2 or sys.argv[1] == "bf16") else torch.float32
B = int(os.environ.get("BENCH_B", "8"))
LAYERS = [  # name, groups, Cout, H, W, upsample_from
    ("conv0_1 32->32 @256x512", [32], 32, 256, 512, None),
    ("deconv1 64->64 @256x512 (up)", [64], 64, 256, 512, (128, 256)),
    ("iconv1 64+32->49 @256x512", [64, 32], 49, 256, 512, None),
    ("vgg 64->64 @256x512", [64], 64, 256, 512, None),
    ("conv1_1 64->64 @128x256", [64], 64, 128, 256, None),
    ("iconv2 64+64->64 @128x256", [64, 64], 64, 128, 256, None),
    ("vgg 128->128 @128x256", [128], 128, 128, 256, None),
    ("conv2_1 128->128 @64x128", [128], 128, 64, 128, None),
    ("vgg 256->256 @64x128", [256], 256, 64, 128, None),
    ("conv3_1 256->256 @32x64", [256], 256, 32, 64, None),
    ("iconv4 128+256->256 @32x64", [128, 256], 256, 32, 64, None),
    ("conv4_1 256->256 @16x32", [256], 256, 16, 32, None),
    ("iconv5 128+256->256 @16x32", [128, 256], 256, 16, 32, None),
    ("deconv4 256->128 @32x64 (up)", [256], 128, 32, 64, (16, 32)),
    ("conv5_1 512->512 @8x16", [512], 512, 8, 16, None),
    ("conv6_1 512->512 @4x8", [512], 512, 4, 8, None),
    ("iconv6 256+512->256 @8x16", [256, 512], 256, 8, 16, None),
    ("deconv6 512->256 @8x16 (up)", [512], 256, 8, 16, (4, 8)),
    ("deconv3 256->128 @64x128 (up)", [256], 128, 64, 128, (32, 64)),
    ("deconv2 128->64 @128x256 (up)", [128], 64, 128, 256, (64, 128)),
]
VARIANTS = [("p128", 2), ("p64", 3), ("S", 4), ("p64M512", 6), ("S512", 7), ("ws", 10), ("ws2", 16), ("dma", 13), ("dma2", 21), ("dma32", 22), ("dma16", 23), ("dma16t4", 24), ("dma16t8", 25), ("dma4", 17), ("dma8", 20), ("up2", 18), ("gather", 1), ("g8", 108), ("g16", 116), ("deep32", 1932), ("deep64", 1964)]
if os.environ.get("BENCH_VARIANTS"):  # e.g. BENCH_VARIANTS=dma,ws
    VARIANTS = [v for v in VARIANTS if v[0] in os.environ["BENCH_VARIANTS"].split(",")]
if os.environ.get("BENCH_LAYERS"):  # substring filter, comma separated
    LAYERS = [l for l in LAYERS if any(k in l[0] for k in os.environ["BENCH_LAYERS"].split(","))]
NOCHECK = os.environ.get("BENCH_NOCHECK") == "1"  # timing probes that deliberately compute wrong values
lib = L.lib()
for name, groups, cout, H, W, up in LAYERS:
    cin = sum(groups)
    w = torch.nn.Parameter(torch.randn(cout, cin, 3, 3, device=DEV) * 0.05)
    pc = ops.PackedConv("t", w, None, groups, 1)
    pc.up2 = up is not None and (2 * up[0], 2 * up[1]) == (H, W)
    pc.alloc(dtype, torch.device(DEV))
    pc.pack_call()()
    if pc.wu is not None:
        ops.pack_up2_call([pc], dtype, torch.device(DEV))()
    srcs_t = []
    for g in groups:
        h, ww = up if up else (H, W)
        srcs_t.append((torch.randn(B, h, ww, ops.pad_c(g), device=DEV) * float(os.environ.get('BENCH_SCALE', '1'))).to(dtype))  # BENCH_SCALE=0: all-zero activations (clock / power probe)
    out = torch.empty(B, H, W, pc.cout_pad, dtype=dtype, device=DEV)
    ops.AUTOTUNE = False
    pooled = torch.empty(B, H // 2, W // 2, pc.cout_pad, dtype=dtype, device=DEV) if os.environ.get('BENCH_POOL') else None  # BENCH_POOL=1: only the 2x2-pooled map is stored; 2: the full map AND the pooled one (the synthesised view's VGG pass)
    # BENCH_DGRAD=1: the data-gradient epilogue (residual addend + activation-gradient operand, no activation) instead of bias-free ELU
    dg = os.environ.get('BENCH_DGRAD') == '1' and pooled is None
    addend = torch.randn(B, H, W, pc.cout_pad, device=DEV).to(dtype) if dg else None
    actout = torch.randn(B, H, W, pc.cout_pad, device=DEV).to(dtype) if dg else None
    call = ops.conv_call(dtype, [ops.nhwc_src(t) for t in srcs_t], H, W, pc.wf, pc.cin_pad, ops.fwd_taps(3), 9, pc.cout_pad, 1, B,
                         H, W, None if (pooled is not None and os.environ.get('BENCH_POOL') != '2') else out, H, W, pc.cout_pad, pc.cout_pad, pool_out=pooled,
                         act=0 if dg else {'elu': L.ACT_ELU, 'relu': L.ACT_RELU, 'none': 0}[os.environ.get('BENCH_ACT', 'elu')], weight_up2=None if dg else pc.wu,
                         addend=addend, actout=actout, actout_kind=L.ACT_ELU if dg else L.ACT_NONE)
    flops = 2.0 * B * H * W * cout * cin * 9
    times = {v: [] for v, _ in VARIANTS}
    ref = None
    for rnd in range(6):
        for v, var in VARIANTS:
            call.desc.variant, call.desc.ksplit = var, 1
            if var in (108, 116):  # gather with split-K 8 / 16
                call.desc.variant, call.desc.ksplit = 1, var - 100
            elif var in (1932, 1964):  # K-sliced one-shot kernel, 32- / 64-channel slices
                call.desc.variant, call.desc.ksplit = 19, pc.cin_pad // (var - 1900)
            if lib.falnet_conv2d(call.ref, L.stream_ptr()) != 0:
                continue
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                call()
            e1.record()
            torch.cuda.synchronize()
            if rnd > 0:
                times[v].append(e0.elapsed_time(e1) / 5)
            if rnd == 0:
                if pooled is not None:
                    pass
                elif NOCHECK:
                    pass
                elif ref is None:
                    ref = out.float().clone()
                else:
                    assert float((out.float() - ref).abs().max()) <= 2e-2 * float(ref.abs().max()), (name, v)
    line = f"{name:34s}"
    for v, _ in VARIANTS:
        if not times[v]:
            line += f" | {v}    n/a         "
            continue
        t = sorted(times[v])[len(times[v]) // 2]
        line += f" | {v} {t*1e3:6.1f}us {flops/t/1e9:6.0f}TF"
    print(line, flush=True)
