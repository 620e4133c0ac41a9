"""Helper of tests/test_gpu_step.py::test_hooked_backward_replay_world1 (run as a subprocess: FALNET_FORCE_DIST / FALNET_DETERMINISTIC are read at
import).  The N > 1 step on one GPU -- world-size-1 RCCL group, bucket hooks installed -- issued by falnet_replay with the recorded backward CUT
at the hooks (fal_net_amd/_lib.py: SegmentChain) against the same steps issued launch by launch from Python: six steps each in deterministic mode,
every step's loss / gradients / weights / disparities bit-identical; the recorded sequence must really be a chain with the three in-backward
hooks as Python cuts, every hook must fire once per step in both forms, and the plan's stream self-test must have run with the collective."""
import hashlib
import json
import os
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "5")  # main + side + third + aux + the collective's stream: as the entry scripts set it (before HIP initialises)
os.environ["FALNET_FORCE_DIST"] = "1"
os.environ["FALNET_DETERMINISTIC"] = "1"
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29541")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from fal_net_amd import _lib as L  # noqa: E402
from fal_net_amd import loss_functions as LF  # noqa: E402
from fal_net_amd import synthetic, train  # noqa: E402
from fal_net_amd.models import FAL_netB  # noqa: E402


def digest(t):
    return hashlib.sha256(t.detach().contiguous().cpu().numpy().tobytes()).hexdigest()[:16]


def run(replay, steps=6):
    L.REPLAY = replay
    dtype = torch.bfloat16
    LF.set_compute_dtype(dtype)
    m = FAL_netB({"state_dict": synthetic.seeded_falnetb_state_dict(49)}, no_levels=49, compute_dtype=dtype).to("cuda").train()
    opt = train.FlatAdam(m)
    left, right, mn, mx = synthetic.synthetic_pair(2, 128, 256, seed=29, distinct=True)
    fired, out = [], []
    for s in range(steps):
        o = train.stage1_step(m, opt, left.cuda(), right.cuda(), mx.cuda())
        if s == 0:  # count the hook calls from the second step on (the hook is installed by the first step)
            inner = m.bucket_hook
            assert inner is not None

            def counting(bucket, view, inner=inner):
                fired.append(bucket)
                inner(bucket, view)
            m.bucket_hook = counting
        torch.cuda.synchronize()
        out.append((float(o["loss"]).hex(), digest(m.flat_gradients()), digest(m.flat_parameters()), digest(o["ldisp"])))
    plan = next(iter(m._plans.values()))
    segs = [type(s).__name__ for s in plan._bwd_segments.values()]
    cuts = [sum(1 for p in s.parts if not isinstance(p, L.Segment)) for s in plan._bwd_segments.values() if isinstance(s, L.SegmentChain)]
    return {"steps": out, "fired": fired, "segments": segs, "cuts": cuts, "selftest": plan.selftest, "hooked_selftest": plan._streams.hooked_tested}


def main():
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    assert L.lib().falnet_get_deterministic() == 1
    a, b = run(True), run(False)
    dist.destroy_process_group()
    import ctypes
    ctypes.CDLL(None).fflush(None)
    print(json.dumps({"replayed": a, "eager": b, "identical": a["steps"] == b["steps"]}), flush=True)


if __name__ == "__main__":
    main()
