#!/usr/bin/env python3
"""A few seconds of ONE conv launch back to back (power / clock probes): loop_conv.py <cin> <cout> <H> <W> [seconds] ; BENCH_SCALE=0 -> zero activations;
CD_VARIANT picks the kernel (default 23).  Prints the mean launch time."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from fal_net_amd import _lib as L, ops
ops.AUTOTUNE = False
DEV, dtype, B = "cuda", torch.bfloat16, int(os.environ.get("BENCH_B", "8"))
cin, cout, H, W = (int(a) for a in sys.argv[1:5])
secs = float(sys.argv[5]) if len(sys.argv) > 5 else 4.0
w = torch.nn.Parameter(torch.randn(cout, cin, 3, 3, device=DEV) * 0.05)
pc = ops.PackedConv("t", w, None, [cin], 1)
pc.alloc(dtype, torch.device(DEV))
pc.pack_call()()
x = (torch.randn(B, H, W, ops.pad_c(cin), device=DEV) * float(os.environ.get("BENCH_SCALE", "1"))).to(dtype)
out = torch.empty(B, H, W, pc.cout_pad, dtype=dtype, device=DEV)
call = ops.conv_call(dtype, [ops.nhwc_src(x)], H, W, pc.wf, pc.cin_pad, ops.fwd_taps(3), 9, pc.cout_pad, 1, B, H, W, out, H, W, pc.cout_pad, pc.cout_pad, act=L.ACT_ELU)
call.desc.variant = int(os.environ.get("CD_VARIANT", "23"))
for _ in range(10):
    call()
torch.cuda.synchronize()
n, t0 = 0, time.time()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
while time.time() - t0 < secs:
    for _ in range(200):
        call()
    n += 200
    torch.cuda.synchronize()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / n
print(f"{cin}->{cout} @{H}x{W} B={B} variant {call.desc.variant} scale {os.environ.get('BENCH_SCALE', '1')}: {us:.1f} us/launch, {2.0 * B * H * W * cin * cout * 9 / us / 1e6:.0f} TFLOP/s over {n} launches")
