"""Inference post-processing of Test_KITTI.py (ms_pp :287-300, flip post-process :200-203) around the HIP model.
The resampling of the 1- and 3-channel maps (bilinear x2/3, nearest back) goes through falnet_resize_planar; only the
host-side 95th percentile stays in numpy exactly as in the reference; the two network forwards are the HIP plan."""
import math

import numpy as np
import torch

from . import _lib as L
from .train import hflip


def resize_planar(x, size, bilinear, scale=1.0):
    """F.interpolate(x, size, mode='bilinear', align_corners=True) / mode='nearest' on planar f32 (B, C, H, W), times `scale`."""
    B, C, H, W = x.shape
    x = x.contiguous().float()
    out = torch.empty(B, C, size[0], size[1], device=x.device)
    L.check(L.lib().falnet_resize_planar(L.ptr(x), L.ptr(out), B * C, H, W, size[0], size[1], int(bilinear), float(scale), L.stream_ptr()),
            "resize_planar")
    return out


def ms_pp(input_view, pan_model, disp, min_disp, max_pix):
    """Test_KITTI.py:287-300: second forward on the flipped, x2/3-downscaled view; blend by normalised disparity."""
    B, C, H, W = input_view.shape
    up_fac = 2 / 3
    # F.interpolate(scale_factor=2/3) sizes the output as floor(in * scale) and (align_corners=True) samples at dst (in-1)/(out-1)
    upscaled = resize_planar(hflip(input_view), (int(math.floor(H * up_fac)), int(math.floor(W * up_fac))), bilinear=True)
    dwn_flip_disp = pan_model(upscaled, min_disp, max_pix, ret_disp=True, ret_pan=False, ret_subocc=False)
    dwn_flip_disp = resize_planar(dwn_flip_disp, (H, W), bilinear=False, scale=1 / up_fac)
    dwn_flip_disp = hflip(dwn_flip_disp)
    norm = disp / (np.percentile(disp.detach().cpu().numpy(), 95) + 1e-6)
    norm[norm > 1] = 1
    return (1 - norm) * disp + norm * dwn_flip_disp


def flip_post_process(input_view, pan_model, disp, min_disp, max_pix):
    """Test_KITTI.py:200-203."""
    flip_disp = pan_model(hflip(input_view), min_disp, max_pix, ret_disp=True, ret_pan=False, ret_subocc=False)
    return (disp + hflip(flip_disp)) / 2
