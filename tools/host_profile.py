#!/usr/bin/env python3
"""Where does the HOST spend its time issuing a Stage-1 step?  cProfile over K un-synchronised steps of the bench workload
(bench.py reports the total as config.host_issue_ms_per_step).  Tuning tool only."""
import cProfile, os, pstats, sys, io
os.environ.setdefault("GPU_MAX_HW_QUEUES", "5")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from fal_net_amd import loss_functions as LF, synthetic, train
from fal_net_amd.models import FAL_netB

dev = torch.device("cuda", 0)
LF.set_compute_dtype(torch.bfloat16)
model = FAL_netB({"state_dict": synthetic.seeded_falnetb_state_dict(49)}, no_levels=49, compute_dtype=torch.bfloat16).to(dev).train()
opt = train.FlatAdam(model)
left, right, mn, mx = synthetic.synthetic_pair(8, 256, 512, seed=1234)
left, right, mx = left.to(dev), right.to(dev), mx.to(dev)
for _ in range(5):
    train.stage1_step(model, opt, left, right, mx)
torch.cuda.synchronize()
K = 20
pr = cProfile.Profile()
pr.enable()
for _ in range(K):
    train.stage1_step(model, opt, left, right, mx)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
st = pstats.Stats(pr, stream=s).sort_stats("tottime")
st.print_stats(28)
print(s.getvalue())
print("per step: total profiled time / K =", st.total_tt / K * 1e3, "ms")
