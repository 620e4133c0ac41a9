"""Deterministic synthetic weights and inputs (``--synthetic`` mode, bench, tests).

There is no dataset and no checkpoint on the benchmark box, so everything that
feeds the hot path is generated from seeds.  The recipe is the one SURVEY.md
§8c fixes: every ``state_dict`` key gets its own generator seeded with
``crc32(key)`` so the 68 MB of weights never have to be committed, and the same
tensors can be loaded into the reference (fixture generation), into the oracle
and into the HIP-backed module.

Shapes/keys follow the reference's checkpoint layout
(``models/FAL_netB.py:92-138,180-192``; key list probed in SURVEY.md §8b).
"""
import zlib
from collections import OrderedDict

import torch

# (name, Cin, Cout, has_bias) of the 3x3 stride-2/1 "conv_elu" stages, FAL_netB.py:99-111
_ENC = [("conv0", 3, 32), ("conv1", 33, 64), ("conv2", 64, 128), ("conv3", 128, 256),
        ("conv4", 256, 256), ("conv5", 256, 256), ("conv6", 256, 512)]
# decoder, FAL_netB.py:116-127: (deconv name, Cin, Cout), (iconv name, Cin, Cout)
_DEC = [("deconv6", 512, 256, "iconv6", 512, 256), ("deconv5", 256, 128, "iconv5", 384, 256),
        ("deconv4", 256, 128, "iconv4", 384, 256), ("deconv3", 256, 128, "iconv3", 256, 128),
        ("deconv2", 128, 64, "iconv2", 128, 64)]


def falnetb_param_shapes(no_levels=49):
    """Ordered {key: shape} of FAL_netB's state_dict (51 tensors; FAL_netB.py:92-192)."""
    shapes = OrderedDict()
    for name, cin, cout in _ENC:
        shapes[f"backbone.{name}.0.weight"] = (cout, cin, 3, 3)
        shapes[f"backbone.{name}.0.bias"] = (cout,)
        shapes[f"backbone.{name}_1.conv1.weight"] = (cout, cout, 3, 3)
        shapes[f"backbone.{name}_1.conv2.weight"] = (cout, cout, 3, 3)
    for dname, dcin, dcout, iname, icin, icout in _DEC:
        shapes[f"backbone.{dname}.conv1.weight"] = (dcout, dcin, 3, 3)
        shapes[f"backbone.{iname}.0.weight"] = (icout, icin, 3, 3)
        shapes[f"backbone.{iname}.0.bias"] = (icout,)
    shapes["backbone.deconv1.conv1.weight"] = (64, 64, 3, 3)
    shapes["backbone.iconv1.weight"] = (no_levels, 96, 3, 3)
    shapes["backbone.amask_conv.0.weight"] = (48, 96, 3, 3)
    shapes["backbone.amask_conv.0.bias"] = (48,)
    shapes["backbone.amask_conv.2.weight"] = (1, 48, 3, 3)
    shapes["conv0.weight"] = (no_levels, no_levels, 1, 1)
    shapes["conv0.bias"] = (no_levels,)
    return shapes


def _seeded(key, shape, bias_std=0.05):
    g = torch.Generator().manual_seed(zlib.crc32(key.encode()))
    if len(shape) == 1:  # biases: small but non-zero so the bias path is exercised
        return torch.randn(shape, generator=g) * bias_std
    fan_in = shape[1] * shape[2] * shape[3]
    return torch.randn(shape, generator=g) * (2.0 / fan_in) ** 0.5  # Kaiming-normal (FAL_netB.py:133)


def seeded_state_dict(arch="B", no_levels=None):
    """Seeded weights of FAL_netA / B / C in the reference's checkpoint key order (models/FAL_net{A,B,C}.py)."""
    from .arch import ARCHS, conv_shapes
    sd = OrderedDict()
    for key, shape, has_bias in conv_shapes(arch, no_levels or ARCHS[arch]["no_levels"]):
        sd[key + ".weight"] = _seeded(key + ".weight", shape)
        if has_bias:
            sd[key + ".bias"] = _seeded(key + ".bias", (shape[0],))
    return sd


def seeded_falnetb_state_dict(no_levels=49):
    """Seeded FAL_netB weights in the reference's checkpoint key order."""
    # the module registers parameters in construction order (encoder interleaved as below)
    sd = OrderedDict()
    for key, shape in falnetb_param_shapes(no_levels).items():
        sd[key] = _seeded(key, shape)
    return sd


# torchvision VGG19 cfg "E", features[0:19] only (loss_functions.py:21-29): index -> (Cin, Cout)
VGG19_PC_CONVS = OrderedDict([(0, (3, 64)), (2, (64, 64)), (5, (64, 128)), (7, (128, 128)),
                              (10, (128, 256)), (12, (256, 256)), (14, (256, 256)), (16, (256, 256))])


def seeded_vgg19_state_dict():
    """Seeded stand-in for torchvision's ``vgg19(pretrained=True).features`` weights.

    Keys are torchvision's (``features.<idx>.weight``).  The real ImageNet weights
    are neither in the reference tree nor downloadable; perceptual-loss parity is
    pinned for these seeded weights only ("parity unpinned" for the
    pretrained ones).
    """
    sd = OrderedDict()
    for idx, (cin, cout) in VGG19_PC_CONVS.items():
        sd[f"features.{idx}.weight"] = _seeded(f"vgg19.features.{idx}.weight", (cout, cin, 3, 3))
        sd[f"features.{idx}.bias"] = _seeded(f"vgg19.features.{idx}.bias", (cout,))
    return sd


def synthetic_pair(batch, height, width, seed=1234, max_disp=300.0, distinct=False):
    """Seeded stereo pair + (B,1,1) disparity range, shaped like the loader's batch.

    ``rand - 0.43`` matches the reference normalisation (``/255`` then minus the RGB
    mean, Train_Stage1_K.py:124-128).  ``distinct`` gives every sample its own
    ``max_disp`` so the per-sample plane tables are exercised.
    """
    g = torch.Generator().manual_seed(seed)
    left = torch.rand(batch, 3, height, width, generator=g) - 0.43
    right = torch.rand(batch, 3, height, width, generator=g) - 0.43
    md = torch.full((batch, 1, 1), float(max_disp))
    if distinct:
        md = md * (1.0 - 0.1 * torch.arange(batch, dtype=torch.float32).view(batch, 1, 1) / max(batch, 1))
    min_disp = md * 2.0 / 300.0  # Train_Stage1_K.py:237 with the default --min_disp 2 --max_disp 300
    return left, right, min_disp, md


def structured_stereo(batch, height, width, seed=77, d_lo=None, d_hi=None, max_disp=300.0):
    """A stereo pair with a KNOWN disparity field (seeded, no dataset): a band-limited textured left image and a smooth right-view
    disparity d_r in [d_lo, d_hi] pixels (default 3 .. 24 px per 256 px of width; larger towards the bottom, as on a road scene, plus two low-frequency bumps per sample);
    the right image is the left one sampled at x + d_r(x, y) (the geometry the network's plane sweep synthesises:
    models/FAL_netB.py:255-262).  The Stage-1 loss then has a defined minimum and the trained disparity can be scored against ground
    truth (tools/trajectory.py).  Returns left, right, min_disp, max_disp as synthetic_pair does, plus the ground-truth LEFT-view
    disparity (B, 1, H, W): d_l(x + d_r(x)) = d_r(x), solved by fixed-point iteration (|d d_r / dx| << 1: no occlusions)."""
    g = torch.Generator().manual_seed(seed)
    F = torch.nn.functional
    # disparity range as a fraction of the width (3 .. 24 px at W = 256): the same geometry at every size
    d_lo = 3.0 * width / 256 if d_lo is None else d_lo
    d_hi = 24.0 * width / 256 if d_hi is None else d_hi

    def band(ch, cells, amp):  # smooth random field: coarse noise, bicubic up (CPU data synthesis, not on the hot path)
        z = torch.rand(batch, ch, max(2, height // cells), max(2, width // cells), generator=g)
        return amp * F.interpolate(z, size=(height, width), mode="bicubic", align_corners=True)
    left = (band(3, 32, 0.45) + band(3, 8, 0.35) + band(3, 3, 0.25) + 0.1 * torch.rand(batch, 3, height, width, generator=g)).clamp(0, 1.2) / 1.2
    yy = torch.linspace(0, 1, height).view(1, 1, height, 1)
    xx = torch.linspace(0, 1, width).view(1, 1, 1, width)
    cx, cy = torch.rand(batch, 2, 1, 1, generator=g), torch.rand(batch, 2, 1, 1, generator=g)
    bumps = sum(torch.exp(-(((xx - cx[:, i:i + 1]) / 0.25) ** 2 + ((yy - cy[:, i:i + 1]) / 0.3) ** 2)) for i in range(2))
    rel = (0.15 + 0.55 * yy ** 1.5 + 0.3 * bumps / 2).clamp(0, 1)
    d_r = d_lo + (d_hi - d_lo) * rel  # (B, 1, H, W)
    xs = torch.arange(width, dtype=torch.float32).view(1, 1, 1, width)

    def sample_x(img, pos):  # img (B, C, H, W) at fractional columns pos (B, 1, H, W), zero outside
        x0 = pos.floor()
        a = pos - x0
        i0, i1 = x0.long(), x0.long() + 1
        v0 = torch.gather(img, 3, i0.clamp(0, width - 1).expand_as(img)) * ((i0 >= 0) & (i0 < width))
        v1 = torch.gather(img, 3, i1.clamp(0, width - 1).expand_as(img)) * ((i1 >= 0) & (i1 < width))
        return v0 * (1 - a) + v1 * a
    right = sample_x(left, xs + d_r)
    d_l = d_r.clone()
    for _ in range(8):  # d_l(x) = d_r(x - d_l(x))
        d_l = sample_x(d_r, (xs - d_l).clamp(0, width - 1))
    md = torch.full((batch, 1, 1), float(max_disp))
    return left - 0.43, right - 0.43, md * 2.0 / 300.0, md, d_l
