"""Helper of tests/test_gpu_bench_config.py: ONE f32 Stage-1 step of the HIP path in a fresh process with FALNET_DETERMINISTIC=1 (the switch is
read when the library is loaded), so that the f32 side of a 16-bit-vs-f32 comparison carries no atomics-order noise and the bound is about the
16-bit kernels alone.  usage: _f32_det_step.py B H W N out.pt [stage2]   (loss scalars, disparity, synthesised view, every parameter's gradient;
`stage2`: one Stage-2 step, teacher = the same seeded weights, as tests/test_gpu_bench_config.py::_stage2 runs it)"""
import os
import sys

os.environ["FALNET_DETERMINISTIC"] = "1"
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))

import torch  # noqa: E402

from fal_net_amd import _lib as L  # noqa: E402
from fal_net_amd import loss_functions as LF  # noqa: E402
from fal_net_amd import synthetic, train  # noqa: E402
from fal_net_amd.models import FAL_netB  # noqa: E402


def main():
    b, h, w, n = (int(a) for a in sys.argv[1:5])
    assert L.lib().falnet_get_deterministic() == 1
    LF.set_compute_dtype(torch.float32)
    left, right, mn, mx = synthetic.synthetic_pair(b, h, w, seed=1234)
    m = FAL_netB({"state_dict": synthetic.seeded_falnetb_state_dict(n)}, no_levels=n, compute_dtype=torch.float32).to("cuda").train()
    if len(sys.argv) > 6 and sys.argv[6] == "stage2":
        fix = FAL_netB({"state_dict": synthetic.seeded_falnetb_state_dict(n)}, no_levels=n, compute_dtype=torch.float32).to("cuda").eval()
        for q in fix.parameters():
            q.requires_grad_(False)
        out = train.stage2_step(m, fix, train.FlatAdam(m, lr=5e-5), left.cuda(), right.cuda(), mx.cuda())
        torch.cuda.synchronize()
        res = {k: float(out[k]) for k in ("loss", "rec", "sm", "mirror")}
        res.update(ldisp=out["ldisp"].detach().cpu(), rdisp=out["rdisp"].detach().cpu())
    else:
        out = train.stage1_step(m, train.FlatAdam(m), left.cuda(), right.cuda(), mx.cuda(), optimize=False)
        torch.cuda.synchronize()
        res = {"loss": float(out["loss"]), "rec": float(out["rec"]), "sm": float(out["sm"]), "ldisp": out["ldisp"].detach().cpu(),
               "rpan": out["rpan"].detach().cpu()}
    res.update(flat_grad=m.flat_gradients().detach().cpu(), grads={k: p.grad.detach().cpu() for k, p in m.named_parameters() if p.grad is not None})
    res["gnorm"] = {k: float(g.norm()) for k, g in res["grads"].items()}
    torch.save(res, sys.argv[5])


if __name__ == "__main__":
    main()
