"""Inference post-processing of Test_KITTI.py (ms_pp :287-300, flip post-process :200-203) around the HIP model.
The resampling of the 1- and 3-channel maps (bilinear x2/3, nearest back) and the host-side percentile are
plumbing and stay in torch/numpy exactly as in the reference; the two network forwards are the HIP plan."""
import numpy as np
import torch
import torch.nn.functional as F

from .train import hflip


def ms_pp(input_view, pan_model, disp, min_disp, max_pix):
    """Test_KITTI.py:287-300: second forward on the flipped, x2/3-downscaled view; blend by normalised disparity."""
    B, C, H, W = input_view.shape
    up_fac = 2 / 3
    upscaled = F.interpolate(hflip(input_view), scale_factor=up_fac, mode='bilinear', align_corners=True)
    dwn_flip_disp = pan_model(upscaled.contiguous(), min_disp, max_pix, ret_disp=True, ret_pan=False, ret_subocc=False)
    dwn_flip_disp = (1 / up_fac) * F.interpolate(dwn_flip_disp, size=(H, W), mode='nearest')
    dwn_flip_disp = hflip(dwn_flip_disp)
    norm = disp / (np.percentile(disp.detach().cpu().numpy(), 95) + 1e-6)
    norm[norm > 1] = 1
    return (1 - norm) * disp + norm * dwn_flip_disp


def flip_post_process(input_view, pan_model, disp, min_disp, max_pix):
    """Test_KITTI.py:200-203."""
    flip_disp = pan_model(hflip(input_view), min_disp, max_pix, ret_disp=True, ret_pan=False, ret_subocc=False)
    return (disp + hflip(flip_disp)) / 2
