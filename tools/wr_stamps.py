#!/usr/bin/env python3
"""Phase stamps of the row-streaming weight-gradient kernel (FALNET_WR_ABL=9 diagnostic build: s_memtime around the loop
segments, per-wave sums in the slab heads).  usage: wr_stamps.py <cin groups a+b> <cout> <H> <W> [up]"""
import os, sys
os.environ["FALNET_WR_ABL"] = "9"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from fal_net_amd import _lib as L, ops
DEV, B, dtype = "cuda", 8, torch.bfloat16
groups = [int(x) for x in sys.argv[1].split("+")]
cout, H, W = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
up = (H // 2, W // 2) if len(sys.argv) > 5 else None
w = torch.nn.Parameter(torch.randn(cout, sum(groups), 3, 3, device=DEV) * 0.05)
pc = ops.PackedConv("t", w, None, groups, 1)
srcs_t = [torch.randn(B, *(up or (H, W)), ops.pad_c(g), device=DEV).to(dtype) for g in groups]
gout = torch.randn(B, H, W, pc.cout_pad, device=DEV).to(dtype)
ws = torch.zeros(40 << 20, device=DEV)
gw = torch.empty_like(w)
c = ops.wgrad_calls(dtype, [ops.nhwc_src(t) for t in srcs_t], H, W, gout, [(dy, dx, 0) for dy, dx, _ in ops.fwd_taps(3)], 1, B, H, W, pc, gw, None, ws)
lib = L.lib()
for _ in range(3):
    L.check(lib.falnet_wgrad(c.desc, L.stream_ptr()))
torch.cuda.synchronize()
d = c.desc
slab = 9 * ops.pad_c(pc.cout_pad) * pc.cin_pad
ntiles = ((pc.cin_pad + 63) // 64) * ((pc.cout_pad + 63) // 64)
raw = ws[: d.nsplit * slab].view(d.nsplit, slab).cpu()
names = ["dma-wait", "barrier", "dma-issue", "frag-reads", "mfma+tail"]
tot = torch.zeros(6, dtype=torch.float64)
rt = ct = 0.0
cnt = 0
for sp in range(d.nsplit):
    v = raw[sp][: ntiles * 8 * 16].contiguous().view(torch.int64).view(ntiles * 8, 8)
    for r in v:
        if r[5] > 0:
            tot[:5] += r[:5].double() / float(r[5])
            tot[5] += float(r[5])
            rt += float(r[6])
            ct += float(r[7])
            cnt += 1
print(f"{sys.argv[1:]} nsplit {d.nsplit}: steps/wave {tot[5] / cnt:.1f}; cycles per step and wave: " +
      ", ".join(f"{n} {float(tot[i]) / cnt:.0f}" for i, n in enumerate(names)) + f"; sum {float(tot[:5].sum()) / cnt:.0f}; loop {rt / cnt / 100:.1f} us, in-kernel clock {ct / rt * 100:.0f} MHz")
