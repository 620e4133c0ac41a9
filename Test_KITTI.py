#!/usr/bin/env python3
"""Inference entry point (reference: Test_KITTI.py): disparity forward + flip / multi-scale post-processing.

`--synthetic` runs on a seeded image of `--height x --width` (native KITTI 375x1242 by default; no dataset on the
box).  Image / PLY dumping of the reference (Test_KITTI.py:211-258) is I/O cosmetics and not provided.  The KITTI
metric chain (myUtils.compute_kitti_errors ...) is available in fal_net_amd.myUtils for real ground truth."""
import argparse
import json
import time

parser = argparse.ArgumentParser(description='FAL_net inference on MI355X', formatter_class=argparse.ArgumentDefaultsHelpFormatter)
parser.add_argument('-maxd', '--max_disp', type=float, default=300)
parser.add_argument('-mind', '--min_disp', type=float, default=2)
parser.add_argument('-relbase', '--rel_baselne', type=float, default=1)
parser.add_argument('-mm', '--m_model', default='FAL_netB', choices=['FAL_netA', 'FAL_netB', 'FAL_netC'])
parser.add_argument('-no_levels', '--no_levels', type=int, default=49)
parser.add_argument('--model', dest='model_dir', default=None, help='checkpoint (reference format); seeded weights if absent')
parser.add_argument('-fpp', '--f_post_process', action='store_true', help='flip post-processing (Test_KITTI.py:200-203)')
parser.add_argument('-mspp', '--ms_post_process', action='store_true', default=True, help='multi-scale post-processing (:287-300)')
parser.add_argument('--no-ms_post_process', dest='ms_post_process', action='store_false')
parser.add_argument('--synthetic', action='store_true', default=True)
parser.add_argument('--height', type=int, default=375)
parser.add_argument('--width', type=int, default=1242)
parser.add_argument('--iters', type=int, default=10)
parser.add_argument('--dtype', default='bf16', choices=['bf16', 'f32'])


def main():
    import torch
    from fal_net_amd import inference, synthetic
    import models
    dev = torch.device('cuda', 0)
    dtype = torch.bfloat16 if args.dtype == 'bf16' else torch.float32
    data = torch.load(args.model_dir, map_location='cpu') if args.model_dir else {'state_dict': synthetic.seeded_state_dict(args.m_model[-1], args.no_levels)}
    pan_model = models.__dict__[args.m_model](data, no_levels=args.no_levels, compute_dtype=dtype).to(dev).eval()
    left, _, _, _ = synthetic.synthetic_pair(1, args.height, args.width, seed=7)
    left = left.to(dev)
    max_disp = torch.tensor([args.max_disp * args.rel_baselne], device=dev).view(1, 1, 1)  # Test_KITTI.py:181-182
    min_disp = max_disp * args.min_disp / args.max_disp
    times = []
    with torch.no_grad():
        for _ in range(args.iters):
            torch.cuda.synchronize()
            t0 = time.time()
            disp = pan_model(left, min_disp, max_disp, ret_disp=True, ret_subocc=False, ret_pan=False)  # :196
            if args.f_post_process:
                disp = inference.flip_post_process(left, pan_model, disp, min_disp, max_disp)
            elif args.ms_post_process:
                disp = inference.ms_pp(left, pan_model, disp, min_disp, max_disp)
            torch.cuda.synchronize()
            times.append(time.time() - t0)
    print(json.dumps({'image': [args.height, args.width], 'dtype': args.dtype, 'post': 'flip' if args.f_post_process else ('ms_pp' if args.ms_post_process else 'none'),
                      'sec_per_image_median': sorted(times)[len(times) // 2], 'disp_mean': float(disp.mean()), 'disp_max': float(disp.max())}))


if __name__ == '__main__':
    args = parser.parse_args()
    main()
