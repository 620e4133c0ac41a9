#!/usr/bin/env python3
"""MED head forward / backward (NHWC 16-bit gradient) alone: time and algorithmic HBM rate.  usage: bench_head.py [B H W N]"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from fal_net_amd import _lib as L
B, H, W, N = (int(a) for a in sys.argv[1:5]) if len(sys.argv) > 4 else (8, 256, 512, 49)
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
dlog = torch.randn(B, N, H, W, device=dev, generator=g)
left = torch.randn(B, 3, H, W, device=dev, generator=g)
mx = torch.full((B,), 300.0 * W / 1242, device=dev)
mn = mx * 2 / 300
disp, pan, stats = torch.empty(B, 1, H, W, device=dev), torch.empty(B, 3, H, W, device=dev), torch.empty(B, 4, H, W, device=dev)
gd, gp = torch.randn(B, 1, H, W, device=dev, generator=g), torch.randn(B, 3, H, W, device=dev, generator=g)
cpad = (N + 31) // 32 * 32
lib, st = L.lib(), L.stream_ptr()
def fwd():
    L.check(lib.falnet_med_head_fwd(L.ptr(dlog), L.ptr(left), L.ptr(mn), L.ptr(mx), L.ptr(disp), L.ptr(pan), L.ptr(stats), B, N, H, W, st), "fwd")
outs = {dt: torch.empty(B, H, W, cpad, device=dev, dtype=dt) for dt in (torch.bfloat16, torch.float32)}
def bwd(dt):
    L.check(lib.falnet_med_head_bwd_nhwc(L.ptr(dlog), L.ptr(left), L.ptr(mn), L.ptr(mx), L.ptr(disp), L.ptr(pan), L.ptr(stats), L.ptr(gd), L.ptr(gp),
                                         L.ptr(outs[dt]), cpad, L.dtype_code(dt), B, N, H, W, st), "bwd")
def timeit(f, n=30):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
HW = H * W
t = timeit(fwd); print(f"fwd  {t:7.1f} us  {(N + 7) * HW * 4 * B / t / 1e6:6.2f} TB/s algorithmic")
for dt in outs:
    t = timeit(lambda: bwd(dt)); print(f"bwd {str(dt)[6:]:9s} {t:7.1f} us  {(2 * N + 7) * HW * 4 * B / t / 1e6:6.2f} TB/s algorithmic (planar-f32 accounting)")
