#!/usr/bin/env python3
"""Does the reduced-precision path TRAIN like the f32 path?  (VERDICT r1 weak #1.)

N Stage-1 steps (Train_Stage1_K.py:233-262 body, fal_net_amd.train.stage1_step) on a cycled pool of seeded synthetic
batches, from the same seeded weights, once per compute dtype (f32 = the parity path, bf16, f16).  Reports per dtype the
loss curve (mean of the first / last `pool` steps), and against the f32-trained model on held-out seeded pairs: depth
abs_rel (myUtils.py:225 formula on f*b/disp) and the max-norm relative disparity difference.

    python tools/trajectory.py --steps 200 --height 128 --width 256 --batch 4 --pool 8
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402


def run(steps=200, height=128, width=256, batch=4, pool=8, levels=49, lr=1e-4, dtypes=("f32", "f32_again", "bf16", "f16"), device="cuda"):
    from fal_net_amd import loss_functions as LF
    from fal_net_amd import synthetic, train
    from fal_net_amd.models import FAL_netB
    # "f32_again": the CONTROL -- a second f32 run.  Split-K / loss reductions use f32 atomics, so two f32 runs sum in different
    # orders and their trajectories separate too (synthetic noise images leave the disparity field weakly determined): the
    # 16-bit runs are judged against that run-to-run distance, not against zero.
    DT = {"f32": torch.float32, "f32_again": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}
    batches = [tuple(t.to(device) for t in synthetic.synthetic_pair(batch, height, width, seed=500 + i)) for i in range(pool)]
    held = [tuple(t.to(device) for t in synthetic.synthetic_pair(2, height, width, seed=900 + i)) for i in range(2)]
    sd = synthetic.seeded_falnetb_state_dict(levels)
    out, disps = {}, {}
    for name in dtypes:
        dt = DT[name]
        LF.set_compute_dtype(dt)
        torch.manual_seed(0)
        m = FAL_netB({"state_dict": sd}, no_levels=levels, compute_dtype=dt).to(device).train()
        opt = train.FlatAdam(m, lr=lr)
        losses = []
        for s in range(steps):
            left, right, mn, mx = batches[s % pool]
            losses.append(train.stage1_step(m, opt, left, right, mx)["loss"])
        losses = [float(x) for x in torch.stack(losses).cpu()]
        m.eval()
        with torch.no_grad():
            disps[name] = torch.cat([m(l, mn, mx).float() for l, r, mn, mx in held]).cpu()
        k = min(pool, steps)
        out[name] = {"loss_first": sum(losses[:k]) / k, "loss_last": sum(losses[-k:]) / k, "finite": bool(all(x == x and abs(x) < 1e30 for x in losses))}
        del m, opt
    LF.set_compute_dtype(torch.float32)
    ref = disps.get("f32")
    if ref is not None:
        for name in dtypes:
            if name == "f32":
                continue
            d = disps[name]
            out[name]["depth_abs_rel_vs_f32_model"] = float(((1.0 / ref - 1.0 / d).abs() * ref).mean())  # |f b/d_ref - f b/d| / (f b/d_ref)
            out[name]["disp_max_rel_vs_f32_model"] = float((d - ref).abs().max() / ref.abs().max())
            out[name]["loss_last_rel_to_f32"] = out[name]["loss_last"] / out["f32"]["loss_last"] - 1.0
    out["config"] = {"steps": steps, "height": height, "width": width, "batch": batch, "pool": pool, "levels": levels, "lr": lr}
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--height", type=int, default=128)
    ap.add_argument("--width", type=int, default=256)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--pool", type=int, default=8)
    ap.add_argument("--levels", type=int, default=49)
    ap.add_argument("--lr", type=float, default=1e-4)
    a = ap.parse_args()
    print(json.dumps(run(a.steps, a.height, a.width, a.batch, a.pool, a.levels, a.lr)))
