#!/usr/bin/env python3
"""Generates tests/golden/g8_data_aug.npz by IMPORTING the reference's data_transforms (path on the command line; Pillow from
this container) -- only runnable where /root/reference exists.  usage: python tests/golden/make_data_goldens.py /root/reference

Cases: (a) PIL bicubic resizes of seeded uint8 images (up- and down-scaling, odd sizes); (b) the full co_transform chain of
Train_Stage1_K.py:116-122 + the input_transform of :124-128 (torchvision's Normalize is (x - mean) / std in float32: restated,
torchvision is not installed) for seeds chosen to cover every branch: flip / no flip, gamma, brightness, per-channel brightness
on float AND on still-uint8 arrays (the truncating assignment of data_transforms.py:155)."""
import os, random, sys
import numpy as np
import torch

ref_root = sys.argv[1]
sys.path.insert(0, ref_root)
import data_transforms as R  # noqa: E402
from PIL import Image  # noqa: E402

out = {}
rng = np.random.RandomState(20260101)
resize_cases = [(40, 60, 75, 50), (40, 60, 45, 30), (37, 53, 61, 37), (48, 160, 236, 70), (50, 70, 52, 71), (48, 160, 146, 43)]
for i, (h, w, ow, oh) in enumerate(resize_cases):
    img = rng.randint(0, 256, (h, w, 3), dtype=np.uint8)
    out[f"rs{i}_in"] = img
    out[f"rs{i}_size"] = np.array([ow, oh])
    out[f"rs{i}_out"] = np.array(Image.fromarray(img).resize((ow, oh), resample=Image.BICUBIC))
out["n_resize"] = np.array(len(resize_cases))

TH, TW, H, W = 32, 96, 48, 160
mean = torch.tensor([0.411, 0.432, 0.45]).view(3, 1, 1)
seeds, seen = [], set()
for seed in range(200):
    random.seed(seed)
    np.random.seed(seed)
    L, Rr = rng.randint(0, 256, (H, W, 3), dtype=np.uint8), rng.randint(0, 256, (H, W, 3), dtype=np.uint8)
    co = R.Compose([R.RandomResizeCrop((TH, TW), down=0.75, up=1.5), R.RandomHorizontalFlip(), R.RandomGamma(min=0.8, max=1.2),
                    R.RandomBrightness(min=0.5, max=2.0), R.RandomCBrightness(min=0.8, max=1.2)])
    res, _ = co([L.copy(), Rr.copy()], None)
    # branch signature: replay the draws
    random.seed(seed)
    np.random.seed(seed)
    np.random.uniform(0, 1)
    random.randint(0, 1), random.randint(0, 1)
    sig = (res[0].dtype == np.uint8, )
    key = (str(res[0].dtype), len(seeds) % 2)
    tens = []
    for a in res:
        t = R.ArrayToTensor()(a)
        t = (t - 0.0) / 255.0
        t = (t - mean) / 1.0
        tens.append(t.numpy())
    kind = (str(res[0].dtype), bool(np.any(res[0] == 255)))
    if len(seeds) < 10 and (kind not in seen or len(seeds) < 6):
        seen.add(kind)
        k = len(seeds)
        seeds.append(seed)
        out[f"aug{k}_left"], out[f"aug{k}_right"] = L, Rr
        out[f"aug{k}_out0"], out[f"aug{k}_out1"] = tens[0], tens[1]
out["aug_seeds"] = np.array(seeds)
out["aug_shape"] = np.array([H, W, TH, TW])
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "g8_data_aug.npz"), **out)
print("seeds", seeds, "kinds", seen)
