"""Real-data input of the training / validation loops (reference: Datasets/Kitti.py:26-58, Datasets/Kitti2015.py:28-92,
Datasets/listdataset_train.py:50-98, Datasets/listdataset_test.py:52-113, Train_Stage1_K.py:137-160).

Division of labour on an MI355X box: loader WORKERS only read and decode files (PNG / JPEG -> uint8 HWC, Pillow); the decoded pair
is pinned, uploaded once, and everything the reference's workers did after the decode -- bicubic resize, crop, flip-swap, gamma /
brightness, tensor conversion, both Normalize steps (data_transforms.py:46-157, Train_Stage1_K.py:116-128) -- runs on the GPU
(fal_net_amd.data_transforms.StereoAugment).  KITTI frames differ in size (370-376 x 1224-1242), so a batch leaves the loader as
a LIST of pairs; it becomes one (B, 3, crop_h, crop_w) tensor after the crop.

Nothing here needs a GPU to import; the augmentation step does (no CPU fallback, like the rest of the path).
"""
import os
import random

import numpy as np
import torch
import torch.utils.data as data


def _imread(path):
    from PIL import Image
    with Image.open(path) as im:
        if im.mode in ("I;16", "I;16B", "I"):  # KITTI disparity maps: 16-bit PNG
            return np.array(im)
        return np.array(im.convert("RGB"))  # (a writable copy: torch.from_numpy shares it)


def read_pair_list(list_file, root):
    """One stereo pair per line, 'left_path right_path' relative to `root` (the format of the reference's Datasets/
    kitti_eigen_train.txt); pairs whose left image is missing under `root` are skipped (Kitti.py:38-41)."""
    if not os.path.isfile(list_file):
        raise FileNotFoundError(
            f"training list {list_file!r} not found: pass --train_list <file> with one 'left right' pair of paths (relative to --data/"
            "<dataName0>) per line.  The Eigen training split the reference defaults to (22 600 pairs) ships with the reference repository as "
            "Datasets/kitti_eigen_train.txt (github.com/JuanLuisGonzalez/FAL_net); it is not redistributed here: copy it next to the script or name it.")
    with open(list_file) as f:
        lines = [ln.split() for ln in f.read().splitlines() if ln.strip()]
    return [(ln[0], ln[1]) for ln in lines if len(ln) >= 2 and os.path.isfile(os.path.join(root, ln[0]))]


class StereoPairDataset(data.Dataset):
    """Decoded training pairs (listdataset_train.py:50-98).  __getitem__ -> (first_u8, second_u8, x_pix): uint8 (H, W, 3) tensors
    and the signed maximum disparity; with fix=False the views are swapped with probability 1/2 and x_pix negated (:70-79)."""

    def __init__(self, root, pairs, max_pix=300, fix=True):
        self.root, self.pairs, self.max_pix, self.fix = root, list(pairs), max_pix, fix

    def __len__(self):
        return len(self.pairs)

    def __getitem__(self, index):
        lp, rp = self.pairs[index]
        left, right = _imread(os.path.join(self.root, lp)), _imread(os.path.join(self.root, rp))
        if self.fix or random.random() < 0.5:
            views, x_pix = (left, right), self.max_pix
        else:
            views, x_pix = (right, left), -self.max_pix
        return torch.from_numpy(views[0]), torch.from_numpy(views[1]), float(x_pix)


def kitti2015_pairs(root, with_disp=True):
    """training/image_2|image_3/%06d_10.png (+ disp_occ_0) of KITTI 2015, the 200 validation pairs (Kitti2015.py:28-56)."""
    out = []
    for i in range(200):
        l, r = os.path.join("training", "image_2", "%06d_10.png" % i), os.path.join("training", "image_3", "%06d_10.png" % i)
        d = os.path.join("training", "disp_occ_0", "%06d_10.png" % i)
        if os.path.isfile(os.path.join(root, l)) and os.path.isfile(os.path.join(root, r)) and (not with_disp or os.path.isfile(os.path.join(root, d))):
            out.append((l, r, d if with_disp else None))
    return out


def eigen_test_triples(list_file, root):
    """The Eigen test split with improved ground truth (Datasets/Kitti_eigen_test_improved.py:33-45): every line of `list_file` is
    'left right' (or 'left right gt'); the ground truth of a two-column line is the projected depth map beside the drive,
    <drive>/proj_depth/groundtruth/image_02/<frame>.png, derived from the left path exactly as the reference slices it (the last 29
    characters are 'image_02/data/<10 digits>.png').  Lines whose image or ground truth is missing under `root` are skipped."""
    if not os.path.isfile(list_file):
        raise FileNotFoundError(f"test list {list_file!r} not found (one 'left right [gt]' line per frame, paths relative to <data>/<tdataName>).  "
                                "The improved Eigen test split (697 lines) ships with the reference repository as Datasets/kitti_eigen_test_improved.txt "
                                "(github.com/JuanLuisGonzalez/FAL_net); it is not redistributed here: copy it or pass --test_list <file>.")
    out = []
    with open(list_file) as f:
        for ln in f.read().splitlines():
            c = ln.split()
            if len(c) < 2:
                continue
            gt = c[2] if len(c) >= 3 else os.path.join(c[0][0:-29], "proj_depth", "groundtruth", "image_02", c[0][-14:])
            if os.path.isfile(os.path.join(root, c[0])) and os.path.isfile(os.path.join(root, gt)):
                out.append((c[0], c[1], gt))
    return out


class StereoValDataset(data.Dataset):
    """Full-size validation pairs with ground-truth disparity (listdataset_test.py:52-113; KITTI disparity PNGs are uint16 / 256,
    :43-46).  __getitem__ -> (left_u8, right_u8, disp_f32 (H, W) or None)."""

    def __init__(self, root, triples):
        self.root, self.triples = root, list(triples)

    def __len__(self):
        return len(self.triples)

    def __getitem__(self, index):
        lp, rp, dp = self.triples[index]
        left, right = _imread(os.path.join(self.root, lp)), _imread(os.path.join(self.root, rp))
        disp = None if dp is None else torch.from_numpy(_imread(os.path.join(self.root, dp)).astype(np.float32) / 256.0)
        return torch.from_numpy(left), torch.from_numpy(right), disp


def _list_collate(batch):
    return batch  # frames differ in size: keep the list


def make_loader(dataset, batch_size, workers, shuffle, rank=0, world=1, seed=0, drop_last=True):
    """DataLoader over decoded uint8 pairs: `workers` decode processes, pinned host memory, per-rank shard of the index space
    (DistributedSampler: every rank draws its own `batch_size` pairs)."""
    sampler = None
    if world > 1:
        sampler = data.distributed.DistributedSampler(dataset, num_replicas=world, rank=rank, shuffle=shuffle, seed=seed, drop_last=drop_last)
    return data.DataLoader(dataset, batch_size=batch_size, num_workers=workers, shuffle=shuffle and sampler is None, sampler=sampler,
                           pin_memory=torch.cuda.is_available(), collate_fn=_list_collate, drop_last=drop_last,
                           persistent_workers=workers > 0)


MEAN = (0.411, 0.432, 0.45)  # Train_Stage1_K.py:127


def to_model_input(img_u8, device):
    """ArrayToTensor + Normalize(0, 255) + Normalize(mean, 1) of a full-size frame (validation: no co_transform,
    Kitti2015.py:88-90): uint8 (H, W, 3) -> planar f32 (1, 3, H, W) on `device`."""
    x = img_u8.to(device, non_blocking=True).permute(2, 0, 1).float().div_(255.0)
    return (x - torch.tensor(MEAN, device=device).view(3, 1, 1)).unsqueeze(0).contiguous()
