#!/usr/bin/env python3
"""Run ONE conv / wgrad launch configuration repeatedly (target for rocprofv3 --pmc).  usage:
   prof_one.py conv <groups e.g. 256> <cout> <H> <W> <variant> | wgrad <groups> <cout> <H> <W>"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from fal_net_amd import _lib as L, ops
ops.AUTOTUNE = False
DEV, dtype, B = "cuda", torch.bfloat16, 8
kind = sys.argv[1]
if kind == "c3":  # first-layer kernel: prof_one.py c3 <cout 32|64> <H> <W>
    cout, H, W = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    w = torch.nn.Parameter(torch.randn(cout, 3, 3, 3, device=DEV) * 0.2)
    bias = torch.nn.Parameter(torch.randn(cout, device=DEV) * 0.1)
    pc = ops.PackedConv("t", w, bias, [3], 1)
    x = torch.randn(B, 3, H, W, device=DEV)
    out = torch.empty(B, H, W, cout, dtype=dtype, device=DEV)
    run = ops.conv_c3_call(dtype, x, pc, out, L.ACT_RELU)
    for _ in range(10):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record()
    torch.cuda.synchronize()
    print(f"c3 3->{cout} @{H}x{W}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us")
    sys.exit(0)
groups = [int(x) for x in sys.argv[2].split("+")]
cout, H, W = int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
cin = sum(groups)
w = torch.nn.Parameter(torch.randn(cout, cin, 3, 3, device=DEV) * 0.05)
pc = ops.PackedConv("t", w, None, groups, 1)
pc.alloc(dtype, torch.device(DEV))
pc.pack_call()()
srcs_t = [torch.randn(B, H, W, ops.pad_c(g), device=DEV).to(dtype) for g in groups]
if kind == "conv":
    out = torch.empty(B, H, W, pc.cout_pad, dtype=dtype, device=DEV)
    addend = torch.randn(B, H, W, pc.cout_pad, device=DEV).to(dtype) if os.environ.get("ADD") == "1" else None
    actout = torch.randn(B, H, W, pc.cout_pad, device=DEV).to(dtype) if os.environ.get("ACTOUT") == "1" else None
    call = ops.conv_call(dtype, [ops.nhwc_src(t) for t in srcs_t], H, W, pc.wf, pc.cin_pad, ops.fwd_taps(3), 9, pc.cout_pad, 1, B,
                         H, W, out, H, W, pc.cout_pad, pc.cout_pad, act=L.ACT_ELU if actout is None else L.ACT_NONE, addend=addend,
                         actout=actout, actout_kind=L.ACT_ELU if actout is not None else L.ACT_NONE)
    call.desc.variant = int(sys.argv[6])
    if os.environ.get("KSPLIT"):  # gather kernel only; the workspace conv_call attached must hold B*H*W*cout_pad floats
        call.desc.ksplit = int(os.environ["KSPLIT"])
    run = call
else:
    gout = torch.randn(B, H, W, pc.cout_pad, device=DEV).to(dtype)
    ws = torch.empty(24 << 20, device=DEV)
    gw = torch.empty_like(w)
    c = ops.wgrad_calls(dtype, [ops.nhwc_src(t) for t in srcs_t], H, W, gout, [(dy, dx, 0) for dy, dx, _ in ops.fwd_taps(3)], 1, B, H, W,
                        pc, gw, None, ws)
    run = lambda: c(0)
    if os.environ.get("WGRAD_VARIANT"):  # 2: 64x64 LDS-DMA, 3: 32(cin) x 64(cout), 4: 64 x 32 channels per workgroup
        c.desc.variant = int(os.environ["WGRAD_VARIANT"])
        if os.environ.get("WGRAD_NSPLIT"):  # must stay within the 24 MiB workspace allocated above
            c.desc.nsplit = min(int(os.environ["WGRAD_NSPLIT"]), (24 << 20) * 4 // (9 * pc.cout_pad * pc.cin_pad * 4))
for _ in range(10):
    run()
torch.cuda.synchronize()
cold = os.environ.get("COLD") == "1"
flush = torch.empty(768 << 20, dtype=torch.uint8, device=DEV) if cold else None
tt = 0.0
for _ in range(10):
    if cold:
        flush.fill_(1)  # evict L2 + Infinity Cache between launches: in-situ cache state of a training step
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    run()
    e1.record()
    torch.cuda.synchronize()
    tt += e0.elapsed_time(e1)
t = tt / 10
print(f"{kind} {groups}->{cout} @{H}x{W}: {t*1e3:.1f} us, {2.0*B*H*W*cout*cin*9/t/1e9:.1f} TF")
