#!/usr/bin/env python3
"""Per-phase time of the halo-patch conv kernel's one-tap-per-stage loop (profiling build: -DFALNET_P_STAMPS).
usage: FALNET_LIB=fal_net_amd/libfalnet_hip_pstamps.so python tools/p_stamps.py <cin> <cout> <H> <W> [variant]"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from fal_net_amd import _lib as L, ops
ops.AUTOTUNE = False
DEV, dtype, B = "cuda", torch.bfloat16, 8
cin, cout, H, W = (int(a) for a in sys.argv[1:5])
variant = int(sys.argv[5]) if len(sys.argv) > 5 else 6
w = torch.nn.Parameter(torch.randn(cout, cin, 3, 3, device=DEV) * 0.05)
pc = ops.PackedConv("t", w, None, [cin], 1)
pc.alloc(dtype, torch.device(DEV))
pc.pack_call()()
x = torch.randn(B, H, W, ops.pad_c(cin), device=DEV).to(dtype)
out = torch.empty(B, H, W, pc.cout_pad, dtype=dtype, device=DEV)
call = ops.conv_call(dtype, [ops.nhwc_src(x)], H, W, pc.wf, pc.cin_pad, ops.fwd_taps(3), 9, pc.cout_pad, 1, B, H, W, out, H, W, pc.cout_pad,
                     pc.cout_pad, act=L.ACT_ELU)
call.desc.variant = variant
stamps = torch.zeros(64, dtype=torch.float32, device=DEV)
call.desc.splitk_ws = stamps.data_ptr()
for _ in range(3):
    call()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); call(); e1.record(); torch.cuda.synchronize()
print(f"launch {e0.elapsed_time(e1) * 1e3:.1f} us")
st = stamps.cpu().view(8, 8)
names = ["issue loads", "frag reads + MFMA issue", "wait vmcnt(0)", "LDS stores", "barrier"]
for wv in range(8):
    taps = st[wv, 5].item()
    if taps == 0:
        continue
    tot = st[wv, :5].sum().item()
    print(f"wave {wv}: {int(taps)} taps, {tot / taps * 10:.0f} ns/tap: " + ", ".join(f"{names[k]} {st[wv, k].item() / taps * 10:.0f}" for k in range(5)))
