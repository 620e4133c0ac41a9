"""Helper of tests/test_gpu_step.py::test_deterministic_mode_is_bit_identical (run as a subprocess with FALNET_DETERMINISTIC=1: the
switch is read when the library is loaded).  Two fresh model instances per dtype, two optimiser steps each, from the same seeded
weights and inputs: losses, flat gradients and weights must be BIT-identical."""
import hashlib
import json
import os
import sys

os.environ["FALNET_DETERMINISTIC"] = "1"
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))

import torch  # noqa: E402

from fal_net_amd import _lib as L  # noqa: E402
from fal_net_amd import loss_functions as LF  # noqa: E402
from fal_net_amd import synthetic, train  # noqa: E402
from fal_net_amd.models import FAL_netB  # noqa: E402


def digest(t):
    return hashlib.sha256(t.detach().contiguous().cpu().numpy().tobytes()).hexdigest()[:16]


def run(dtype, stage2=False, steps=2, replay=True):
    L.REPLAY = replay  # host launch path: recorded sequences through falnet_replay (after two eager passes) vs every launch from Python
    LF.set_compute_dtype(dtype)
    sd = synthetic.seeded_falnetb_state_dict(49)
    m = FAL_netB({"state_dict": sd}, no_levels=49, compute_dtype=dtype).to("cuda").train()
    fix = FAL_netB({"state_dict": sd}, no_levels=49, compute_dtype=dtype).to("cuda").eval() if stage2 else None
    opt = train.FlatAdam(m)
    left, right, mn, mx = synthetic.synthetic_pair(2, 128, 256, seed=23, distinct=True)
    out = []
    for _ in range(steps):
        if stage2:
            o = train.stage2_step(m, fix, opt, left.cuda(), right.cuda(), mx.cuda())
        else:
            o = train.stage1_step(m, opt, left.cuda(), right.cuda(), mx.cuda())
        torch.cuda.synchronize()
        out.append((float(o["loss"]).hex(), digest(m.flat_gradients()), digest(m.flat_parameters()), digest(o["ldisp"])))
    return out


def main():
    assert L.lib().falnet_get_deterministic() == 1
    res = {}
    for name, dt, s2 in (("f32", torch.float32, False), ("bf16", torch.bfloat16, False), ("f32_stage2", torch.float32, True)):
        a, b = run(dt, s2), run(dt, s2)
        res[name] = {"a": a, "b": b, "identical": a == b}
    # the C replay of the recorded launch sequences against the eager launch path: six steps each (recording happens after the second
    # one), every step's loss, gradients, weights and disparities bit-identical
    for name, dt, s2 in (("replay_bf16", torch.bfloat16, False), ("replay_f32_stage2", torch.float32, True)):
        a, b = run(dt, s2, steps=6, replay=True), run(dt, s2, steps=6, replay=False)
        res[name] = {"a": a, "b": b, "identical": a == b}
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
