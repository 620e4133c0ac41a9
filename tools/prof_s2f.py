#!/usr/bin/env python3
"""Run ONE forward stride-2 conv configuration repeatedly on a forced variant (target for tools/pmc_cmd.sh).
usage: prof_s2f.py <cin> <cout> <H> <W> [variant=15]"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from fal_net_amd import _lib as L, ops
ops.AUTOTUNE = False
DEV, dtype, B = "cuda", torch.bfloat16, 8
cin, cout, H, W = (int(a) for a in sys.argv[1:5])
variant = int(sys.argv[5]) if len(sys.argv) > 5 else 15
w = torch.nn.Parameter(torch.randn(cout, cin, 3, 3, device=DEV) * 0.05)
pc = ops.PackedConv("t", w, None, [cin], 2)
pc.alloc(dtype, torch.device(DEV))
pc.pack_call()()
x = torch.randn(B, H, W, ops.pad_c(cin), device=DEV).to(dtype)
OH, OW = (H + 1) // 2, (W + 1) // 2
out = torch.empty(B, OH, OW, pc.cout_pad, dtype=dtype, device=DEV)
call = ops.conv_call(dtype, [ops.nhwc_src(x)], H, W, pc.wf, pc.cin_pad, ops.fwd_taps(3), pc.taps, pc.cout_pad, 2, B, OH, OW, out, OH, OW,
                     pc.cout_pad, pc.cout_pad, act=L.ACT_ELU)
call.desc.variant = variant
def run():
    L.check(L.lib().falnet_conv2d(call.ref, L.stream_ptr()), "conv")
for _ in range(10):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(30):
    run()
e1.record()
torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 30 * 1e3
print(f"s2 conv {cin}->{cout} @{H}x{W} v{variant}: {t:.1f} us  {2 * B * OH * OW * cout * cin * 9 / t / 1e6:.0f} TF")
