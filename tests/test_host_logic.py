"""CPU: host-side logic of the hot path (no kernels run): tap tables, channel-group packing math, module surface,
checkpoint keys, loud failure without a GPU."""
import itertools

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from fal_net_amd import ops, synthetic
from fal_net_amd.models import FAL_netB


def test_state_dict_surface_matches_reference_keys():
    m = FAL_netB(None, no_levels=49)
    keys = list(m.state_dict().keys())
    assert keys == list(synthetic.falnetb_param_shapes(49).keys())
    for k, shape in synthetic.falnetb_param_shapes(49).items():
        assert tuple(m.state_dict()[k].shape) == tuple(shape), k
    assert sum(p.numel() for p in m.parameters()) == 16974354  # FAL_netB N=49
    trainable = sum(p.numel() for n, p in m.named_parameters() if "amask_conv" not in n)
    assert trainable == 16932402
    assert len(m.weight_parameters()) + len(m.bias_parameters()) == len(list(m.parameters()))
    m2 = FAL_netB({"state_dict": synthetic.seeded_falnetb_state_dict(33)}, no_levels=33)
    assert m2.conv0.weight.shape == (33, 33, 1, 1)


def test_forward_on_cpu_fails_loudly():
    m = FAL_netB(None, no_levels=7)
    x = torch.zeros(1, 3, 64, 128)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(x, torch.ones(1, 1, 1), torch.ones(1, 1, 1) * 30)
    from fal_net_amd import loss_functions as LF
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        LF.vgg(x)


def test_stride2_dgrad_parity_tables_are_exact():
    """The 4 parity-class tap tables reproduce the full transposed conv (checked with torch on CPU)."""
    g = torch.Generator().manual_seed(0)
    for H, W in ((8, 10), (7, 9)):
        x = torch.randn(1, 2, H, W, generator=g, requires_grad=True)
        w = torch.randn(3, 2, 3, 3, generator=g)
        y = F.conv2d(x, w, stride=2, padding=1)
        gy = torch.randn(y.shape, generator=g)
        y.backward(gy)
        OH, OW = y.shape[2:]
        gin = torch.zeros(1, 2, H, W)
        for py, px in itertools.product(range(2), range(2)):
            for dy, dx, wi in ops.dgrad_taps_s2(py, px):
                kh, kw = wi // 3, wi % 3
                for ty in range((H - py + 1) // 2):
                    for tx in range((W - px + 1) // 2):
                        oy, ox = ty + dy, tx + dx
                        if 0 <= oy < OH and 0 <= ox < OW:
                            gin[0, :, 2 * ty + py, 2 * tx + px] += (gy[0, :, oy, ox, None] * w[:, :, kh, kw]).sum(0)
        assert torch.allclose(gin, x.grad, atol=1e-5)
    # every 3x3 tap appears in exactly one parity class
    seen = sorted(wi for py in range(2) for px in range(2) for _, _, wi in ops.dgrad_taps_s2(py, px))
    assert seen == list(range(9))


def test_tap_tables_forward_and_flipped():
    f, d = ops.fwd_taps(3), ops.dgrad_taps_s1(3)
    assert [(dy, dx) for dy, dx, _ in f] == [(kh - 1, kw - 1) for kh in range(3) for kw in range(3)]
    for (dy, dx, wi) in d:  # spatial tap (dy,dx) of a stride-1 dgrad uses packed weight tap 8 - spatial index
        assert wi == 8 - ((dy + 1) * 3 + (dx + 1))
    assert ops.fwd_taps(1) == [(0, 0, 0)]


def test_packed_conv_channel_groups():
    w = torch.nn.Parameter(torch.zeros(64, 33, 3, 3))
    pc = ops.PackedConv("conv1", w, None, [32, 1], stride=2)
    assert pc.groups_pad == [32, 32] and pc.cin_pad == 64 and pc.cout_pad == 64 and pc.group_channels() == (32, 32)
    w = torch.nn.Parameter(torch.zeros(49, 96, 3, 3))
    pc = ops.PackedConv("iconv1", w, None, [64, 32])
    assert pc.cin_pad == 96 and pc.cout_pad == 64
    w = torch.nn.Parameter(torch.zeros(32, 3, 3, 3))
    pc = ops.PackedConv("conv0", w, None, [3])
    assert pc.cin_pad == 32 and pc.group_channels() == (3, 32)
    assert ops.pad_c(49) == 64 and ops.pad_c(96) == 96 and ops.pad_c(1) == 32


def test_synthetic_is_deterministic():
    a, b = synthetic.seeded_falnetb_state_dict(7), synthetic.seeded_falnetb_state_dict(7)
    assert all(torch.equal(a[k], b[k]) for k in a)
    l1, r1, mn1, mx1 = synthetic.synthetic_pair(2, 8, 16, seed=3, distinct=True)
    l2, r2, mn2, mx2 = synthetic.synthetic_pair(2, 8, 16, seed=3, distinct=True)
    assert torch.equal(l1, l2) and torch.equal(r1, r2) and torch.equal(mx1, mx2)
    assert float(mx1[1]) < float(mx1[0]) and torch.allclose(mn1, mx1 * 2 / 300)
    v = synthetic.seeded_vgg19_state_dict()
    assert v["features.16.weight"].shape == (256, 256, 3, 3)


def test_kitti_metrics_against_golden(golden_dir):
    import os
    from fal_net_amd import myUtils as U
    g = np.load(os.path.join(golden_dir, "g6_losses_metrics.npz"))
    assert np.allclose(U.compute_kitti_errors(g["gt"], g["pr"]), g["kitti_errors"], rtol=1e-6)
    td, pd = U.disps_to_depths_kitti2015(g["gd"], g["pd"])
    assert np.allclose(np.asarray(td), g["gt_depth"], rtol=1e-6) and np.allclose(np.asarray(pd), g["pred_depth"], rtol=1e-6)
