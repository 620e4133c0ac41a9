#!/usr/bin/env python3
"""Per-phase cycle stamps of the two-phase weight-stationary conv kernel (profiling build: -DFALNET_WS_STAMPS).
usage: FALNET_LIB=fal_net_amd/libfalnet_hip_stamps.so python tools/ws2_stamps.py <cin> <cout> <H> <W>"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from fal_net_amd import _lib as L, ops
ops.AUTOTUNE = False
DEV, dtype, B = "cuda", torch.bfloat16, 8
cin, cout, H, W = (int(a) for a in sys.argv[1:5])
w = torch.nn.Parameter(torch.randn(cout, cin, 3, 3, device=DEV) * 0.05)
pc = ops.PackedConv("t", w, None, [cin], 1)
pc.alloc(dtype, torch.device(DEV))
pc.pack_call()()
x = torch.randn(B, H, W, ops.pad_c(cin), device=DEV).to(dtype)
out = torch.empty(B, H, W, pc.cout_pad, dtype=dtype, device=DEV)
call = ops.conv_call(dtype, [ops.nhwc_src(x)], H, W, pc.wf, pc.cin_pad, ops.fwd_taps(3), 9, pc.cout_pad, 1, B, H, W, out, H, W, pc.cout_pad,
                     pc.cout_pad, act=L.ACT_ELU)
call.desc.variant = 16
stamps = torch.zeros(8 * 256, dtype=torch.int64, device=DEV)
call.desc.splitk_ws = stamps.data_ptr()
for _ in range(3):
    call()
torch.cuda.synchronize()
st = stamps.cpu().view(8, 256)
t00 = int(st[:, 0].min())
for wv in range(8):
    t = [int(v) for v in st[wv] if int(v) > 0]
    nph = len(t) // 4
    print(f"wave {wv}: {nph} phases, total {t[-1] - t[0]} ticks")
    rows = []
    for k in range(4, min(nph - 2, 12)):
        b = 4 * k
        nxt = t[b + 4]
        rows.append((t[b] - t00, t[b + 1] - t[b], t[b + 2] - t[b + 1], t[b + 3] - t[b + 2], nxt - t[b + 3]))
    for r in rows:
        matrix = r[3] < 60
        if matrix:
            print(f"   t={r[0]:8d}  barrier wait {r[1]:6d} | prefetch issue {r[2]:6d} | fragments + MFMA {r[4]:6d}")
        else:
            print(f"   t={r[0]:8d}  barrier wait {r[1]:6d} | patch store   {r[2]:6d} | epilogue {r[3]:6d} | to phase end {r[4]:6d}")
