import sys, os
sys.path.insert(0, os.getcwd())
import torch
from fal_net_amd import synthetic, train, ops
from fal_net_amd import loss_functions as LF
from fal_net_amd.models import FAL_netB
ops.AUTOTUNE = False
LF.set_compute_dtype(torch.bfloat16)
m = FAL_netB({"state_dict": synthetic.seeded_falnetb_state_dict(49)}, no_levels=49, compute_dtype=torch.bfloat16).cuda().train()
l, r, mn, mx = synthetic.synthetic_pair(8, 256, 512)
opt = train.FlatAdam(m)
train.stage1_step(m, opt, l.cuda(), r.cuda(), mx.cuda())
plan = next(iter(m._plans.values()))
tot = 0
for it in plan.wbatch.items:
    tot += it["bytes"]
    print(f'{it["bucket"]} nsplit {it["nsplit"]:4d} taps {it["ntaps"]} rows {it["w_rows"]:4d} cin {it["cin_total"]:4d} MB {it["bytes"]/1e6:7.2f}')
print("total slab MB", tot / 1e6)
