#!/usr/bin/env python3
"""Times of the loss launches of a Stage-1 step at B = 8, 256 x 512 (perceptual MSE of the three VGG slices, smoothness forward + adjoint, L1 + add).  Tuning tool."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from fal_net_amd import _lib as L
lib, DEV, B, H, W = L.lib(), "cuda", 8, 256, 512
dt = torch.bfloat16
S = torch.zeros(2, device=DEV); seed = torch.ones(1, device=DEV)
def timeit(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
flush = torch.empty(512 << 20, dtype=torch.uint8, device=DEV)
for name, (h, w, c) in {"mse slice1 64ch@128x256": (128, 256, 64), "mse slice2 128ch@64x128": (64, 128, 128), "mse slice3 256ch@32x64": (32, 64, 256)}.items():
    a, b, g = (torch.randn(B, h, w, c, device=DEV).to(dt) for _ in range(3))
    st = L.stream_ptr()
    t = timeit(lambda: L.check(lib.falnet_mse_fwd_bwd(L.ptr(a), L.ptr(b), B * h * w, c, 0.01, L.ptr(S), 0.01, L.ptr(seed), L.ptr(g), L.dtype_code(dt), st)))
    print(f"{name}: {t:6.1f} us  ({3 * a.numel() * 2 / t / 1e6:6.2f} TB/s)")
img, disp, gd = torch.randn(B, 3, H, W, device=DEV), torch.rand(B, 1, H, W, device=DEV), torch.empty(B, 1, H, W, device=DEV)
t = timeit(lambda: L.check(lib.falnet_smooth_fwd_bwd(L.ptr(img), L.ptr(disp), B, H, W, 102, W, 2.0, 1e-6, L.ptr(S[1:]), L.ptr(seed), L.ptr(gd), L.stream_ptr())))
print(f"smooth_fwd_bwd: {t:6.1f} us")
