#!/usr/bin/env python3
"""Race hunting: the same forward+backward from the same state N times; prints, per parameter, the largest deviation of the
gradient from the first repetition (atomics only reorder f32 sums: ~1e-6; a data race shows as an outlier)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from fal_net_amd import loss_functions as LF, synthetic, train
from fal_net_amd.models import FAL_netB
dev = "cuda"
DT = torch.float32 if os.environ.get("GR_F32") == "1" else torch.bfloat16
LF.set_compute_dtype(DT)
m = FAL_netB({"state_dict": synthetic.seeded_falnetb_state_dict(49)}, 49, compute_dtype=DT).to(dev).train()
opt = train.FlatAdam(m, lr=1e-4, betas=(0.5, 0.999))
left, right, mn, mx = synthetic.synthetic_pair(8, 256, 512, seed=1234)
left, right, mx = left.to(dev), right.to(dev), mx.to(dev)
ref, worst = None, {}
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
for it in range(n):
    out = train.stage1_step(m, opt, left, right, mx, optimize=False)
    torch.cuda.synchronize()
    g = {k: p.grad.detach().float().clone() for k, p in m.named_parameters() if p.grad is not None}
    if ref is None:
        ref = g
        continue
    for k in g:
        d = float((g[k] - ref[k]).abs().max() / ref[k].abs().max().clamp_min(1e-20))
        worst[k] = max(worst.get(k, 0.0), d)
top = sorted(worst.items(), key=lambda kv: -kv[1])[:8]
print("loss", float(out["loss"]))
for k, v in top:
    print(f"{v:.3e}  {k}")
