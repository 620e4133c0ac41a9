#!/usr/bin/env python3
"""Phase timeline of the levels-5/6 conv kernel (profiling build: python -m fal_net_amd._build --ab stamps -DFALNET_DEEP_STAMPS).
usage: FALNET_LIB=fal_net_amd/libfalnet_hip_stamps.so python tools/deep_stamps.py <cin> <cout> <H> <W> <slice channels 32|64>"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from fal_net_amd import _lib as L, ops
ops.AUTOTUNE = False
DEV, dtype, B = "cuda", torch.bfloat16, 8
cin, cout, H, W, kc = (int(a) for a in sys.argv[1:6])
w = torch.nn.Parameter(torch.randn(cout, cin, 3, 3, device=DEV) * 0.05)
pc = ops.PackedConv("t", w, None, [cin], 1)
pc.alloc(dtype, torch.device(DEV))
pc.pack_call()()
x = torch.randn(B, H, W, ops.pad_c(cin), device=DEV).to(dtype)
out = torch.empty(B, H, W, pc.cout_pad, dtype=dtype, device=DEV)
call = ops.conv_call(dtype, [ops.nhwc_src(x)], H, W, pc.wf, pc.cin_pad, ops.fwd_taps(3), 9, pc.cout_pad, 1, B, H, W, out, H, W, pc.cout_pad,
                     pc.cout_pad, act=L.ACT_ELU)
call.desc.variant, call.desc.ksplit = 19, pc.cin_pad // kc
stamps = torch.zeros(8192 * 8, dtype=torch.int64, device=DEV)
call.desc.pool_actout = stamps.data_ptr()
for _ in range(3):
    stamps.zero_()
    torch.cuda.synchronize()
    call()
torch.cuda.synchronize()
st = stamps.cpu().view(-1, 8)
nwg = int((st[:, 0] > 0).sum())
st = st[:nwg].double() * 0.01  # us (100 MHz)
t0 = float(st[:, 0].min())
names = ["start", "loads issued", "image in LDS", "MFMA done", "partials stored", "stores acked", "counter known", "epilogue done"]
print(f"{nwg} workgroups; kernel span {float(st[st > 0].max()) - t0:.2f} us")
for k in range(8):
    col = st[:, k]
    col = col[col > 0] - t0
    if len(col):
        print(f"  {names[k]:16s} n={len(col):5d}  min {float(col.min()):7.2f}  median {float(col.median()):7.2f}  max {float(col.max()):7.2f} us")
d = st[:, 1:6] - st[:, 0:5]
for k, nm in enumerate(["issue loads", "wait for image", "nine taps", "store partials", "ack stores"]):
    print(f"  phase {nm:14s} median {float(d[:, k].median()):6.2f}  max {float(d[:, k].max()):6.2f} us")
order = torch.argsort(st[:, 0])
print("  start times (sorted) every 32nd:", [round(float(st[i, 0]) - t0, 2) for i in order[::32]])
