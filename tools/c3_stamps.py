#!/usr/bin/env python3
"""Phase timeline of the first-layer kernel (profiling build: python -m fal_net_amd._build --ab c3st -DC3_STAMPS; the stamps go to
falnet_conv_t::pool_actout, which falnet_conv3x3_c3 leaves NULL -- so this tool calls the kernel through a descriptor hook: FALNET_C3_STAMP_PTR)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from fal_net_amd import _lib as L
B, H, W = 8, 256, 512
cout = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = "cuda"
x = torch.randn(B, 3, H, W, device=dev)
w = torch.randn(cout, 3, 3, 3, device=dev) * 0.2
b = torch.randn(cout, device=dev) * 0.1
out = torch.empty(B, H, W, cout, device=dev, dtype=torch.bfloat16)
stamps = torch.zeros(4096 * 8, dtype=torch.int64, device=dev)
os.environ["FALNET_C3_STAMP_PTR"] = str(stamps.data_ptr())
lib, st = L.lib(), L.stream_ptr()
for _ in range(3):
    stamps.zero_(); torch.cuda.synchronize()
    L.check(lib.falnet_conv3x3_c3(L.ptr(x), L.ptr(w), L.ptr(b), L.ptr(out), B, H, W, cout, L.ACT_ELU, L.dtype_code(torch.bfloat16), st), "c3")
torch.cuda.synchronize()
s = stamps.cpu().view(-1, 8)
n = int((s[:, 0] > 0).sum())
s = s[:n].double() * 0.01
t0 = float(s[:, 0].min())
print(f"{n} workgroups, span {float(s[:, :7].max()) - t0:.2f} us")
names = ["start", "patch loads issued", "all loads consumed", "barrier", "first tile done", "last tile done", "stores drained"]
for k in range(7):
    c = s[:, k] - t0
    print(f"  {names[k]:28s} min {float(c.min()):6.2f} median {float(c.median()):6.2f} max {float(c.max()):6.2f}")
d = s[:, 1:7] - s[:, 0:6]
for k in range(6):
    print(f"  phase -> {names[k + 1]:28s} median {float(d[:, k].median()):5.2f} max {float(d[:, k].max()):5.2f}")
order = torch.argsort(s[:, 0])
print("  start times every 64th:", [round(float(s[i, 0]) - t0, 1) for i in order[::64]])
