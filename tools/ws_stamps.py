#!/usr/bin/env python3
"""Per-phase cycle stamps of the weight-stationary conv kernel (profiling build: libfalnet_hip_stamps.so, -DFALNET_WS_STAMPS).
usage: FALNET_LIB=fal_net_amd/libfalnet_hip_stamps.so python tools/ws_stamps.py <cin> <cout> <H> <W>"""
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
call.desc.variant = 10
stamps = torch.zeros(4 * 256, dtype=torch.int64, device=DEV)
call.desc.splitk_ws = stamps.data_ptr()
for _ in range(3):
    call()
torch.cuda.synchronize()
st = stamps.cpu().view(4, 256)
names = ["barrierA", "patch_store(+vmcnt)", "barrierB", "prefetch issue", "mfma", "epilogue", "loop back"]
for wv in range(4):
    t = st[wv]
    n = int((t > 0).sum())
    tiles = n // 7
    print(f"wave {wv}: {tiles} tiles, total {(int(t[n-1]) - int(t[0]))} ticks")
    import collections
    agg = collections.defaultdict(list)
    for k in range(1, tiles - 1):  # skip first/last
        base = k * 7
        for ph in range(7):
            nxt = int(t[base + ph + 1]) if base + ph + 1 < n else None
            if nxt is not None:
                agg[ph].append(nxt - int(t[base + ph]))
    for ph in range(7):
        if agg[ph]:
            print(f"   {names[ph]:22s} avg {sum(agg[ph]) / len(agg[ph]):8.0f} ticks")
