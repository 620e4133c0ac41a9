"""The part of the reference's myUtils.py the hot path needs: checkpoint writer, meters, and the KITTI metric
chain behind `abs_rel vs ref` (myUtils.py:10-13,59-110,177-277).  Host-side numpy, as in the reference."""
import os
import shutil

import numpy as np
import torch

kitti_error_names = ['abs_rel', 'sq_rel', 'rms', 'log_rms', 'a1', 'a2', 'a3']
width_to_focal = {1242: 721.5377, 1241: 718.856, 1224: 707.0493, 1238: 718.3351, 1226: 707.0912, 1280: 738.2355}
width_to_baseline = {1242: 0.9982 * 0.54, 1241: 0.9848 * 0.54, 1224: 1.0144 * 0.54, 1238: 0.9847 * 0.54,
                     1226: 0.9765 * 0.54, 1280: 0.54}


def save_checkpoint(state, is_best, save_path, filename='checkpoint.pth.tar'):
    """myUtils.py:10-13; `state` = {'epoch','m_model','state_dict','best_rmse'} (Train_Stage1_K.py:202-207)."""
    torch.save(state, os.path.join(save_path, filename))
    if is_best:
        shutil.copyfile(os.path.join(save_path, filename), os.path.join(save_path, 'model_best.pth.tar'))


class AverageMeter(object):
    """myUtils.py:59-78."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count

    def __repr__(self):
        return 'last:{:.3f} avg:({:.3f})'.format(float(self.val), float(self.avg))


class multiAverageMeter(object):
    """myUtils.py:81-110."""

    def __init__(self, labels):
        self.meter_no, self.labels = len(labels), labels
        self.reset()

    def reset(self):
        self.val, self.avg = np.zeros(self.meter_no), np.zeros(self.meter_no)
        self.sum, self.count = np.zeros(self.meter_no), np.zeros(self.meter_no)

    def update(self, val, n=1):
        for i in range(self.meter_no):
            self.val[i] = val[i]
            self.sum[i] += val[i] * n
            self.count[i] += n
            self.avg[i] = self.sum[i] / self.count[i]

    def __repr__(self):
        return "".join("{:>10}".format(l) for l in self.labels) + "\n" + "".join("{:10.4f}".format(a) for a in self.avg)


def get_rmse(output_right, label_right, mean=(0.411, 0.432, 0.45)):
    """RMSE of the synthesised view in 8-bit units, prediction clamped to [0, 255] (myUtils.py:138-150); device follows the input."""
    shift = torch.tensor(mean, device=output_right.device, dtype=output_right.dtype).view(1, 3, 1, 1)
    out = ((output_right + shift) * 255).clamp(0, 255)
    lab = (label_right + shift) * 255
    return torch.mean((out - lab) ** 2) ** 0.5


def get_n_params(model):
    return sum(p.numel() for p in model.parameters())


def compute_kitti_errors(gt, pred, use_median=False, min_d=1.0, max_d=80.0):
    """myUtils.py:196-231."""
    mask = gt > 0
    gt, pred = gt[mask].copy(), pred[mask].copy()
    if use_median:
        pred = np.median(gt) / np.median(pred) * pred
    pred = np.clip(pred, min_d, max_d)
    gt = np.clip(gt, min_d, max_d)
    thresh = np.maximum(gt / pred, pred / gt)
    a1, a2, a3 = (thresh < 1.25).mean(), (thresh < 1.25 ** 2).mean(), (thresh < 1.25 ** 3).mean()
    rmse = np.sqrt(((gt - pred) ** 2).mean())
    rmse_log = np.sqrt(((np.log(gt) - np.log(pred)) ** 2).mean())
    abs_rel = np.mean(np.abs(gt - pred) / gt)
    sq_rel = np.mean(((gt - pred) ** 2) / gt)
    return [abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3]


def disps_to_depths_kitti2015(gt_disparities, pred_disparities):
    """myUtils.py:234-253."""
    gt_depths, pred_depths = [], []
    for gt_disp, pred_disp in zip(gt_disparities, pred_disparities):
        width = gt_disp.shape[1]
        gt_mask, pred_mask = gt_disp > 0, pred_disp > 0
        gt_depth = width_to_focal[width] * 0.54 / (gt_disp + (1.0 - gt_mask))
        pred_depth = width_to_focal[width] * 0.54 / (pred_disp + (1.0 - pred_mask))
        gt_depths.append(gt_mask * gt_depth)
        pred_depths.append(pred_depth)
    return gt_depths, pred_depths


def disps_to_depths_kitti(gt_disparities, pred_disparities):
    """myUtils.py:256-277 (Eigen crop rows H-219:H-4, cols 44:1180; gt is already depth)."""
    gt_depths, pred_depths = [], []
    for gt_disp, pred_disp in zip(gt_disparities, pred_disparities):
        height, width = gt_disp.shape
        gt_disp = gt_disp[height - 219:height - 4, 44:1180]
        pred_disp = pred_disp[height - 219:height - 4, 44:1180]
        gt_mask, pred_mask = gt_disp > 0, pred_disp > 0
        pred_depth = width_to_focal[width] * width_to_baseline[width] / (pred_disp + (1.0 - pred_mask))
        gt_depths.append(gt_mask * gt_disp)
        pred_depths.append(pred_depth)
    return gt_depths, pred_depths
