"""GPU parity tests, model level: fal_net_amd.models.FAL_netB (HIP plan) vs the CPU oracle and the
golden fixtures generated from the reference."""
import os
import zlib

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from fal_net_amd import synthetic  # noqa: E402
from fal_net_amd.models import FAL_netB  # noqa: E402
from oracle import falnet_oracle as O  # noqa: E402

DEV = "cuda"
F32_TOL = 1e-4  # north_star: disparity maps and loss scalars within 1e-4 relative on the f32 path


def rel(a, b):
    a, b = torch.as_tensor(a).detach().double().cpu(), torch.as_tensor(b).detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))


def build(n_levels, dtype=torch.float32):
    m = FAL_netB({"state_dict": synthetic.seeded_falnetb_state_dict(n_levels)}, no_levels=n_levels, compute_dtype=dtype)
    return m.to(DEV)


@pytest.mark.parametrize("n_levels", [7, 49])
def test_forward_vs_golden_and_oracle(golden_dir, n_levels):
    g = np.load(os.path.join(golden_dir, f"g1_forward_n{n_levels}.npz"))
    st = int(g["stride"])
    left, right, mn, mx = synthetic.synthetic_pair(2, 64, 128, seed=int(g["seed"]), distinct=True)
    m = build(n_levels).eval()
    with torch.no_grad():
        pan, disp, maskL, maskR = m(left.to(DEV), mn.to(DEV), mx.to(DEV), ret_disp=True, ret_subocc=True, ret_pan=True)
        d_only = m(left.to(DEV), mn.to(DEV), mx.to(DEV))
    s = slice(None, None, st)
    assert rel(disp, g["disp"]) < F32_TOL
    assert rel(d_only, g["disp"]) < F32_TOL
    assert rel(pan[:, :, s, s], g["p_im0"]) < 2e-4  # + the reference's own fp32 grid noise (test_oracle_golden.WARP_TOL)
    assert rel(maskL[:, :, s, s], g["maskL"]) < 2e-4
    assert rel(maskR[:, :, s, s], g["maskR"]) < 2e-4
    with torch.no_grad():
        o = O.falnet_forward(synthetic.seeded_falnetb_state_dict(n_levels), left, mn, mx, True, True, True, return_dict=True)
    assert rel(m._plans[(2, 64, 128, torch.float32)].buf["dlog0"], o["dlog0"]) < F32_TOL
    assert rel(pan, o["p_im0"]) < F32_TOL
    assert rel(maskL, o["maskL"]) < F32_TOL and rel(maskR, o["maskR"]) < F32_TOL


def test_odd_size_forward(golden_dir):
    g = np.load(os.path.join(golden_dir, "g7_odd_75x250.npz"))
    left, right, mn, mx = synthetic.synthetic_pair(1, 75, 250, seed=int(g["seed"]))
    with torch.no_grad():
        disp = build(49).eval()(left.to(DEV), mn.to(DEV), mx.to(DEV))
    assert rel(disp, g["disp"]) < F32_TOL


@pytest.mark.parametrize("n_levels,B,H,W", [(7, 2, 64, 128), (7, 1, 75, 250)])
def test_backward_vs_oracle(n_levels, B, H, W):
    """Gradients of an arbitrary scalar of (p_im0, disp) wrt every parameter, f32 path.  75 x 250 (golden G7's odd size, forward-only there): every
    decoder level upsamples by a non-x2 ratio (2x4 -> 3x8 -> 5x16 -> 10x32 -> 19x63 -> 38x125 -> 75x250), so the full-model backward runs the
    general nearest-upsample adjoint (FAL_netB.py:57 F.interpolate(size=...)) and the odd-size stride-2 data / weight gradients."""
    left, right, mn, mx = synthetic.synthetic_pair(B, H, W, seed=77, distinct=True)
    gen = torch.Generator().manual_seed(78)
    gp, gd = torch.randn(B, 3, H, W, generator=gen), torch.randn(B, 1, H, W, generator=gen) * 0.01
    params = O.leaf_params(synthetic.seeded_falnetb_state_dict(n_levels))
    pan, disp = O.falnet_forward(params, left, mn, mx, ret_disp=True, ret_pan=True)
    ((pan * gp).sum() + (disp * gd).sum()).backward()
    m = build(n_levels).train()
    pan_h, disp_h = m(left.to(DEV), mn.to(DEV), mx.to(DEV), ret_disp=True, ret_pan=True)
    ((pan_h * gp.to(DEV)).sum() + (disp_h * gd.to(DEV)).sum()).backward()
    for k, p in m.named_parameters():
        if "amask_conv" in k:
            assert p.grad is None
            continue
        ref = params[k].grad
        assert p.grad is not None, k
        assert rel(p.grad, ref) < 2e-3, (k, rel(p.grad, ref))  # max-norm per tensor; fp32 sum-order noise on big reductions
        assert abs(float(p.grad.norm()) - float(ref.norm())) / float(ref.norm()) < 2e-4, k
    # second backward without zero_grad accumulates
    pan_h, disp_h = m(left.to(DEV), mn.to(DEV), mx.to(DEV), ret_disp=True, ret_pan=True)
    ((pan_h * gp.to(DEV)).sum() + (disp_h * gd.to(DEV)).sum()).backward()
    k = "backbone.iconv1.weight"
    assert rel(dict(m.named_parameters())[k].grad, 2 * params[k].grad) < 2e-3


def test_bf16_forward_deviation():
    """bf16 path: no 1e-4 gate (the reference has no reduced-precision path); report and bound loosely."""
    left, right, mn, mx = synthetic.synthetic_pair(2, 64, 128, seed=11, distinct=True)
    m32, m16 = build(49).eval(), build(49, torch.bfloat16).eval()
    with torch.no_grad():
        p32, d32 = m32(left.to(DEV), mn.to(DEV), mx.to(DEV), ret_disp=True, ret_pan=True)
        p16, d16 = m16(left.to(DEV), mn.to(DEV), mx.to(DEV), ret_disp=True, ret_pan=True)
    dev = rel(d16, d32)
    print("bf16 disp deviation (max-norm rel):", dev, " abs_rel:", float(((d16 - d32).abs() / d32).mean()))
    assert dev < 0.1


def test_highres_n96_forward_vs_oracle():
    """BASELINE configs[4] shape: 384x1280, N=96 planes (f32 path vs the CPU oracle; B=1 keeps the oracle in seconds)."""
    left, right, mn, mx = synthetic.synthetic_pair(1, 384, 1280, seed=96)
    sd = synthetic.seeded_falnetb_state_dict(96)
    with torch.no_grad():
        ref = O.falnet_forward(sd, left, mn, mx, ret_disp=True, ret_pan=True)
    m = FAL_netB({"state_dict": sd}, no_levels=96).to(DEV).eval()
    with torch.no_grad():
        pan, disp = m(left.to(DEV), mn.to(DEV), mx.to(DEV), ret_disp=True, ret_pan=True)
    assert rel(disp, ref[1]) < F32_TOL
    assert rel(pan, ref[0]) < F32_TOL


def test_composed_logits_conv_matches_two_launch_form(monkeypatch):
    """iconv1 (3x3, linear) followed by the 1x1 conv0 runs as ONE 3x3 convolution with composed weights (FAL_netB.py:127,174,
    190,215); outputs and the gradients of BOTH original weight tensors (split back from the composed weight gradient) must
    equal the two-launch form."""
    left, right, mn, mx = synthetic.synthetic_pair(2, 64, 96, seed=11)
    res = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("FALNET_AB", "1")  # experiment switches are honoured only with FALNET_AB=1
        monkeypatch.setenv("FALNET_COMPOSE_LOGITS", flag)
        m = FAL_netB({"state_dict": synthetic.seeded_falnetb_state_dict(7)}, 7).to(DEV).train()
        pan, disp = m(left.to(DEV), mn.to(DEV), mx.to(DEV), ret_disp=True, ret_pan=True)
        g = torch.Generator().manual_seed(3)
        ((pan * torch.randn(pan.shape, generator=g).to(DEV)).sum() + (disp * torch.randn(disp.shape, generator=g).to(DEV)).sum() * 0.01).backward()
        res[flag] = (disp.detach().clone(), pan.detach().clone(), {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None})
    assert rel(res["1"][0], res["0"][0]) < 1e-5 and rel(res["1"][1], res["0"][1]) < 1e-5
    for k in ("backbone.iconv1.weight", "conv0.weight", "conv0.bias", "backbone.deconv1.conv1.weight", "backbone.conv0.0.weight"):
        assert rel(res["1"][2][k], res["0"][2][k]) < 1e-4, k
