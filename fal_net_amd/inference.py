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


def evaluate(pan_model, loader, data_name="Kitti2015", max_disp=300.0, min_disp=2.0, rel_baseline=1.0, post="ms_pp", use_median=False,
             print_freq=10, log=print, with_metrics=True):
    """The evaluation loop of Test_KITTI.py:163-208,255-280 over a loader of full-size frames (batch size 1: KITTI mixes image
    sizes, :113): forward (+ flip or multi-scale post-processing, :196-205), then per image the KITTI depth errors and -- for
    KITTI 2015 -- the end-point error (:257-271).  `loader` yields lists of (left_u8, right_u8, gt) from
    datasets.StereoValDataset; gt is a disparity map (Kitti2015) or a depth map (Eigen split, listdataset_test.py:43-46 reads both
    as uint16 / 256).  Returns {'epe', 'kitti': {name: value}, 'n', 'sec_per_image'}."""
    import time
    from . import datasets as DS
    from . import myUtils as utils
    from .loss_functions import realEPE
    dev = next(pan_model.parameters()).device
    pan_model.eval()
    epes, kitti, batch_time = utils.AverageMeter(), utils.multiAverageMeter(utils.kitti_error_names), utils.AverageMeter()
    n = 0
    with torch.no_grad():
        for i, batch in enumerate(loader):
            for left_u8, right_u8, gt in batch:
                left = DS.to_model_input(left_u8, dev)
                mx = torch.full((1, 1, 1), float(max_disp) * rel_baseline, device=dev)  # :181-182
                mn = mx * min_disp / max_disp
                torch.cuda.synchronize()
                t0 = time.time()
                disp = pan_model(left, mn, mx, ret_disp=True, ret_subocc=False, ret_pan=False)  # :196
                if post == "flip":
                    disp = flip_post_process(left, pan_model, disp, mn, mx)
                elif post == "ms_pp":
                    disp = ms_pp(left, pan_model, disp, mn, mx)
                torch.cuda.synchronize()
                batch_time.update(time.time() - t0, 1)
                if gt is not None and with_metrics:  # `-eval False`: forward and timing only (:255)
                    target = gt.to(dev).view(1, 1, *gt.shape)
                    t_np, p_np = target.squeeze(1).cpu().numpy(), disp.float().squeeze(1).cpu().numpy()
                    if data_name == "Kitti2015":  # :265-271
                        epes.update(float(realEPE(disp, target, sparse=True)), 1)
                        gt_depth, pred_depth = utils.disps_to_depths_kitti2015(t_np, p_np)
                    else:  # Eigen split: :258-263
                        gt_depth, pred_depth = utils.disps_to_depths_kitti(t_np, p_np)
                    kitti.update(utils.compute_kitti_errors(gt_depth[0], pred_depth[0], use_median=use_median), 1)
                n += 1
            if log is not None and i % print_freq == 0:
                log('Test: [{0}/{1}]\t Time {2}\t a1 {3:.4f}'.format(i, len(loader), batch_time, kitti.avg[4]))  # :273-275
    return {"epe": epes.avg, "kitti": dict(zip(utils.kitti_error_names, [float(a) for a in kitti.avg])), "kitti_table": repr(kitti), "n": n,
            "sec_per_image": batch_time.avg}
