#!/usr/bin/env python3
"""First-layer kernel (falnet_conv3x3_c3: 3-channel planar f32 image -> 32 / 64-channel NHWC activation) alone: time and output write rate.
usage: bench_c3.py [B H W]"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from fal_net_amd import _lib as L
B, H, W = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (8, 256, 512)
dev = "cuda"
x = torch.randn(B, 3, H, W, device=dev)
lib, st = L.lib(), L.stream_ptr()
for cout, act in ((64, L.ACT_RELU), (32, L.ACT_ELU)):
    w = torch.randn(cout, 3, 3, 3, device=dev) * 0.2
    b = torch.randn(cout, device=dev) * 0.1
    for dt in (torch.bfloat16, torch.float16):
        out = torch.empty(B, H, W, cout, device=dev, dtype=dt)
        def run():
            L.check(lib.falnet_conv3x3_c3(L.ptr(x), L.ptr(w), L.ptr(b), L.ptr(out), B, H, W, cout, act, L.dtype_code(dt), st), "c3")
        for _ in range(5): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30): run()
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 30 * 1e3
        ref = torch.nn.functional.conv2d(x[:1], w, b, padding=1)
        ref = torch.relu(ref) if act == L.ACT_RELU else torch.nn.functional.elu(ref)
        err = float((out[:1].float().permute(0, 3, 1, 2) - ref).abs().max() / ref.abs().max())
        print(f"Cout {cout} {str(dt)[6:]:9s} {t:7.1f} us  {out.numel() * 2 / t / 1e6:5.2f} TB/s written   max rel err {err:.1e}")
