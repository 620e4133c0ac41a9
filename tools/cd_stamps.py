#!/usr/bin/env python3
"""Per-phase cycle stamps of the LDS-DMA conv kernel (profiling build with -DFALNET_CD_STAMPS).
usage: FALNET_LIB=fal_net_amd/libfalnet_hip_cdst.so python tools/cd_stamps.py <cin> <cout> <H> <W>"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from fal_net_amd import _lib as L, ops
ops.AUTOTUNE = False
DEV, dtype, B = "cuda", torch.bfloat16, int(os.environ.get("BENCH_B", "8"))
cin, cout, H, W = (int(a) for a in sys.argv[1:5])
w = torch.nn.Parameter(torch.randn(cout, cin, 3, 3, device=DEV) * 0.05)
pc = ops.PackedConv("t", w, None, [cin], 1)
pc.alloc(dtype, torch.device(DEV))
pc.pack_call()()
x = torch.randn(B, H, W, ops.pad_c(cin), device=DEV).to(dtype)
out = torch.empty(B, H, W, pc.cout_pad, dtype=dtype, device=DEV)
call = ops.conv_call(dtype, [ops.nhwc_src(x)], H, W, pc.wf, pc.cin_pad, ops.fwd_taps(3), 9, pc.cout_pad, 1, B, H, W, out, H, W, pc.cout_pad,
                     pc.cout_pad, act=L.ACT_ELU)
call.desc.variant = int(os.environ.get("CD_VARIANT", "13"))
stamps = torch.zeros(8 * 256, dtype=torch.int64, device=DEV)
call.desc.splitk_ws = stamps.data_ptr()
for _ in range(5):
    call()
torch.cuda.synchronize()
st = stamps.cpu().view(8, 256)
names = ["wait own DMA", "barrier", "issue next chunk", "fragment reads + MFMAs", "epilogue (tile ends)", "loop back"]
for wv in range(8):
    t = st[wv]
    n = int((t > 0).sum())
    its = n // 6
    nch = ops.pad_c(cin) // 32
    tot = (int(t[(its - 1) * 6]) - int(t[6])) / max(its - 2, 1)
    print(f"wave {wv}: {its} chunks, {tot:.0f} ticks per chunk (s_memtime ticks: 100 MHz)")
    for cpos in range(nch):  # by position of the chunk inside its tile (the last one carries the epilogue)
        agg = {ph: [] for ph in range(6)}
        for k in range(1, its - 1):
            if k % nch != cpos:
                continue
            base = k * 6
            for ph in range(6):
                if base + ph + 1 < n:
                    agg[ph].append(int(t[base + ph + 1]) - int(t[base + ph]))
        print(f"   chunk {cpos} of {nch}: " + ", ".join(f"{names[ph]} {sum(v) / max(len(v), 1):.0f}" for ph, v in agg.items()))
