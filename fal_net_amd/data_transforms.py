"""GPU mirror of the reference's training-data augmentation (data_transforms.py + Train_Stage1_K.py:116-128).

The reference augments every stereo pair on the host (PIL bicubic resize + numpy, 4 loader workers); at >1000 pairs/s per
MI355X that is the bottleneck, so here the decoded uint8 pair is uploaded once and everything runs as two kinds of HIP
launches (falnet_resample_u8 x2 per image, falnet_augment_normalize x1 per image): PIL-bit-exact bicubic resize, random crop,
left/right flip-swap, gamma / brightness / per-channel brightness, ArrayToTensor and both Normalize steps, producing the planar
f32 tensors the model consumes.  The random draws are made on the host IN THE REFERENCE'S ORDER from the same generators
(`np.random` for the scale factor, Python `random` for the rest), so a seeded run reproduces the reference's augmentation.

No CPU fallback: the tensors must live on the GPU.
"""
import math
import random

import numpy as np
import torch

from . import _lib as L

MEAN = (0.411, 0.432, 0.45)  # Train_Stage1_K.py:127
_PRECISION_BITS = 32 - 8 - 2


def _bicubic(x, a=-0.5):
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


_COEFF_CACHE = {}


def resample_coeffs(in_size, out_size, device):
    """Pillow's precompute_coeffs + normalize_coeffs_8bpc (src/libImaging/Resample.c) for a whole-axis bicubic resize:
    device tensors bounds [out][2] int32 and kk [out][ksize] int32 (22-bit fixed point), cached per (in, out, device)."""
    key = (in_size, out_size, str(device))
    if key in _COEFF_CACHE:
        return _COEFF_CACHE[key]
    if len(_COEFF_CACHE) > 4096:  # the random scale factor makes most sizes one-offs
        _COEFF_CACHE.clear()
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    inv = 1.0 / filterscale
    # vectorised over the output index, same float64 operations in the same order as the scalar C loop (the weight sum is
    # accumulated tap by tap, not with numpy's pairwise reduction, so the fixed-point coefficients are bit-identical)
    center = (np.arange(out_size, dtype=np.float64) + 0.5) * scale
    xmin = np.maximum((center - support + 0.5).astype(np.int64), 0)          # (int) truncation of a value >= -support+0.5: clamp after
    xmin = np.where(center - support + 0.5 < 0, 0, xmin)
    xmax = np.minimum((center + support + 0.5).astype(np.int64), in_size) - xmin
    xs = np.arange(ksize, dtype=np.float64)[None, :]
    arg = np.abs((xs + xmin[:, None] - center[:, None] + 0.5) * inv)
    a = -0.5
    w = np.where(arg < 1.0, ((a + 2.0) * arg - (a + 3.0)) * arg * arg + 1, np.where(arg < 2.0, (((arg - 5) * arg + 8) * arg - 4) * a, 0.0))
    valid = np.arange(ksize)[None, :] < xmax[:, None]
    w = np.where(valid, w, 0.0)
    ww = np.zeros(out_size, np.float64)
    for x in range(ksize):
        ww = ww + w[:, x]
    v = np.where(ww[:, None] != 0.0, w / np.where(ww[:, None] != 0.0, ww[:, None], 1.0), w)
    fx = v * float(1 << _PRECISION_BITS)
    kk = np.where(v < 0, np.trunc(-0.5 + fx), np.trunc(0.5 + fx)).astype(np.int32)
    kk = np.where(valid, kk, 0).astype(np.int32)
    bounds = np.stack([xmin, xmax], axis=1).astype(np.int32)
    out = (torch.from_numpy(bounds).to(device), torch.from_numpy(kk).to(device), ksize)
    _COEFF_CACHE[key] = out
    return out


def resize_bicubic_u8(img, out_w, out_h):
    """Image.fromarray(img).resize((out_w, out_h), BICUBIC) for a uint8 (H, W, 3) CUDA tensor (bit-exact with Pillow)."""
    if not (img.is_cuda and img.dtype == torch.uint8 and img.dim() == 3 and img.is_contiguous()):
        raise RuntimeError("resize_bicubic_u8 needs a contiguous uint8 (H, W, C) tensor on the GPU (no CPU fallback)")
    lib = L.lib()
    H, W, Cc = img.shape
    cur = img
    if out_w != W:
        b, k, ks = resample_coeffs(W, out_w, img.device)
        nxt = torch.empty(H, out_w, Cc, dtype=torch.uint8, device=img.device)
        L.check(lib.falnet_resample_u8(L.ptr(cur), L.ptr(nxt), H, W, Cc, out_w, 1, L.ptr(b), L.ptr(k), ks, L.stream_ptr()), "resample_u8(h)")
        cur, W = nxt, out_w
    if out_h != H:
        b, k, ks = resample_coeffs(H, out_h, img.device)
        nxt = torch.empty(out_h, W, Cc, dtype=torch.uint8, device=img.device)
        L.check(lib.falnet_resample_u8(L.ptr(cur), L.ptr(nxt), H, W, Cc, out_h, 0, L.ptr(b), L.ptr(k), ks, L.stream_ptr()), "resample_u8(v)")
        cur = nxt
    return cur


def draw_params(h, w, th, tw, down=0.75, up=1.5, gamma=(0.8, 1.2), bright=(0.5, 2.0), cbright=(0.8, 1.2)):
    """The co_transform chain's random draws, in the reference's order and from the same generators
    (data_transforms.py:63-65,76-77,97,124-126,139-141,153-157)."""
    min_factor = max(max((th + 1) / h, (tw + 1) / w), down)
    factor = np.random.uniform(low=min_factor, high=up)
    rw, rh = int(w * factor), int(h * factor)
    x1 = random.randint(0, rw - tw)
    y1 = random.randint(0, rh - th)
    flip = random.random() < 0.5
    g = random.uniform(*gamma) if random.random() < 0.5 else None
    b = random.uniform(*bright) if random.random() < 0.5 else None
    cb = [[random.uniform(*cbright) for _ in range(3)] for _ in range(2)] if random.random() < 0.5 else None
    return dict(factor=factor, rw=rw, rh=rh, x1=x1, y1=y1, flip=flip, gamma=g, bright=b, cbright=cb)


class StereoAugment:
    """co_transform + input_transform of Train_Stage1_K.py:116-128 for one stereo pair on the GPU.

    __call__([left_u8, right_u8]) with (H, W, 3) uint8 CUDA tensors returns [view0, view1], planar f32 (3, crop_h, crop_w),
    mean-shifted exactly like the reference's loader output (`input_data[0]`, listdataset_train.py:90-98)."""

    def __init__(self, crop_height, crop_width, down=0.75, up=1.5, gamma=(0.8, 1.2), brightness=(0.5, 2.0), cbrightness=(0.8, 1.2)):
        self.size, self.down, self.up = (int(crop_height), int(crop_width)), down, up
        self.gamma, self.brightness, self.cbrightness = gamma, brightness, cbrightness

    def __call__(self, inputs, params=None):
        left, right = inputs
        h, w, _ = left.shape
        th, tw = self.size
        prm = params or draw_params(h, w, th, tw, self.down, self.up, self.gamma, self.brightness, self.cbrightness)
        lib = L.lib()
        outs = []
        # RandomHorizontalFlip swaps the views as well as mirroring them (data_transforms.py:98-100); the per-image factors of
        # RandomCBrightness are drawn for the positions AFTER the swap
        order = (right, left) if prm["flip"] else (left, right)
        for i, img in enumerate(order):
            resized = resize_bicubic_u8(img, prm["rw"], prm["rh"])
            out = torch.empty(3, th, tw, dtype=torch.float32, device=img.device)
            cb = prm["cbright"][i] if prm["cbright"] is not None else (0.0, 0.0, 0.0)
            L.check(lib.falnet_augment_normalize(L.ptr(resized), prm["rh"], prm["rw"], prm["x1"], prm["y1"], th, tw, int(prm["flip"]),
                                                 float(prm["gamma"] or 0.0), float(prm["bright"] or 0.0), float(cb[0]), float(cb[1]), float(cb[2]),
                                                 MEAN[0], MEAN[1], MEAN[2], L.ptr(out), L.stream_ptr()), "augment_normalize")
            outs.append(out)
        return outs
