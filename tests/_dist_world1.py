"""Helper of tests/test_gpu_step.py::test_overlapped_allreduce_equals_single_allreduce_world1 (run as a subprocess: the collective
path is selected at import by FALNET_FORCE_DIST=1).  World size 1 over RCCL: the bucketed asynchronous all-reduce fired from the
weight-gradient side stream against ONE all-reduce of the whole flat buffer after backward -- same gradients, same weights."""
import json
import os
import sys

os.environ["FALNET_FORCE_DIST"] = "1"
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from fal_net_amd import loss_functions as LF  # noqa: E402
from fal_net_amd import synthetic, train  # noqa: E402
from fal_net_amd.models import FAL_netB  # noqa: E402


def run(dtype, overlap):
    LF.set_compute_dtype(dtype)
    m = FAL_netB({"state_dict": synthetic.seeded_falnetb_state_dict(49)}, no_levels=49, compute_dtype=dtype).to("cuda").train()
    if not overlap:
        m._no_overlap = True
    opt = train.FlatAdam(m)
    left, right, mn, mx = synthetic.synthetic_pair(2, 64, 128, seed=17, distinct=True)
    fired = []
    out = None
    for _ in range(1):
        out = train.stage1_step(m, opt, left.cuda(), right.cuda(), mx.cuda())
        if overlap:
            assert getattr(m, 'bucket_hook', None) is not None
    torch.cuda.synchronize()
    return float(out["loss"]), m.flat_gradients().clone(), m.flat_parameters().clone(), (getattr(m, 'bucket_hook', None) is not None)


def main():
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    res = {}
    for name, dt in (("f32", torch.float32), ("f16", torch.float16)):
        la, ga, wa, hooked = run(dt, True)
        lb, gb, wb, hooked_b = run(dt, False)
        assert hooked and not hooked_b
        res[name] = {"loss": [la, lb], "grad_rel": float((ga - gb).norm() / gb.norm()), "w_maxabs": float((wa - wb).abs().max()),
                     "finite": bool(torch.isfinite(ga).all() and torch.isfinite(gb).all())}
    dist.destroy_process_group()
    import ctypes
    ctypes.CDLL(None).fflush(None)
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
