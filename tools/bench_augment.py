#!/usr/bin/env python3
"""Throughput of the GPU augmentation path (fal_net_amd.data_transforms.StereoAugment) on KITTI-sized uint8 pairs resident in
HBM: pairs/s, to compare with the step rate of bench.py (the reference augments on the host with PIL/numpy, 4 workers)."""
import os, random, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
from fal_net_amd import data_transforms as DT
dev = "cuda"
H, W, TH, TW = 375, 1242, 256, 512
g = torch.Generator().manual_seed(0)
pairs = [[torch.randint(0, 256, (H, W, 3), generator=g, dtype=torch.uint8).to(dev) for _ in range(2)] for _ in range(8)]
aug = DT.StereoAugment(TH, TW)
random.seed(0); np.random.seed(0)
for p in pairs:
    aug(p)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 0
for _ in range(20):
    for p in pairs:
        aug(p)
        n += 1
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"StereoAugment {H}x{W} -> {TH}x{TW}: {n / dt:.0f} pairs/s ({dt / n * 1e3:.3f} ms/pair, host-issue bound if << GPU time)")
try:
    from PIL import Image
    import data_transforms  # noqa: F401  (only where the reference is on PYTHONPATH)
except Exception:
    pass
