#!/usr/bin/env python3
"""Inference / evaluation entry point (reference: Test_KITTI.py) on the MI355X implementation.

Two modes:
  * dataset mode (`-d <root> -tn Kitti2015`, or `-tn Kitti_eigen_test_improved --test_list <file>`): the reference's evaluation
    loop (Test_KITTI.py:103-117 file-list dataset at batch size 1, :163-208 forward + flip / multi-scale post-processing,
    :255-271 per-image KITTI depth errors and EPE, :277-280 `errors.txt`) over full-size frames of mixed sizes.  Frames are decoded
    by loader workers (Pillow) and normalised on the GPU; the network, `ms_pp` resampling and flips are HIP kernels; the metric
    chain is host-side numpy exactly as in the reference (fal_net_amd.myUtils).
  * `--synthetic`: seeded image of `--height x --width` (native KITTI 375x1242 by default), timing only -- no dataset on the box.
Image / PLY dumping of the reference (:211-253) is I/O cosmetics and not provided.  `--dtype f16` is the recommended 16-bit
inference type (depth abs_rel vs the f32 path 2e-3, bf16 1.7e-2, at the same speed)."""
import argparse
import json
import os
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # application-level choice, before HIP initialises (fal_net_amd/__init__.py)

parser = argparse.ArgumentParser(description='FAL_net inference on MI355X', formatter_class=argparse.ArgumentDefaultsHelpFormatter)
parser.add_argument('-d', '--data', metavar='DIR', default=None, help='dataset root; frames are read from <data>/<tdataName>')
parser.add_argument('-tn', '--tdataName', default='Kitti2015', choices=['Kitti2015', 'Kitti_eigen_test_improved'])
parser.add_argument('--test_list', default=os.path.join('Datasets', 'kitti_eigen_test_improved.txt'),
                    help="Eigen split: one 'left right [gt]' line per frame, paths relative to <data>/<tdataName> (the reference opens "
                         "Datasets/kitti_eigen_test_improved.txt relative to the working directory)")
parser.add_argument('-maxd', '--max_disp', type=float, default=300)
parser.add_argument('-mind', '--min_disp', type=float, default=2)
parser.add_argument('-relbase', '--rel_baselne', type=float, default=1)
parser.add_argument('-mm', '--m_model', default='FAL_netB', choices=['FAL_netA', 'FAL_netB', 'FAL_netC'])
parser.add_argument('-no_levels', '--no_levels', type=int, default=49)
parser.add_argument('--model', dest='model_dir', default=None, help='checkpoint (reference format); required in dataset mode')
parser.add_argument('-fpp', '--f_post_process', action='store_true', help='flip post-processing (Test_KITTI.py:200-203)')
parser.add_argument('-mspp', '--ms_post_process', action='store_true', default=True, help='multi-scale post-processing (:287-300)')
parser.add_argument('--no-ms_post_process', dest='ms_post_process', action='store_false')
parser.add_argument('--median', action='store_true', help='median scaling (not needed when training from stereo)')
parser.add_argument('-w', '--workers', type=int, default=4)
parser.add_argument('-p', '--print-freq', type=int, default=10)
parser.add_argument('--save-path', default=None, help='where errors.txt / settings.txt go (default Test_Results/<tdataName>/<model>[fpp|mspp])')
parser.add_argument('--synthetic', action='store_true', help='timing on a seeded image (default when no --data is given)')
parser.add_argument('--allow-seeded-weights', action='store_true', help='dataset mode without --model: evaluate SEEDED (untrained) weights (tests)')
parser.add_argument('--height', type=int, default=375)
parser.add_argument('--width', type=int, default=1242)
parser.add_argument('--iters', type=int, default=10)
parser.add_argument('--dtype', default='f16', choices=['f16', 'bf16', 'f32'])


def main():
    import torch
    from fal_net_amd import inference, synthetic
    from fal_net_amd import myUtils as utils
    import models
    dev = torch.device('cuda', 0)
    dtype = {'f16': torch.float16, 'bf16': torch.bfloat16, 'f32': torch.float32}[args.dtype]
    post = 'flip' if args.f_post_process else ('ms_pp' if args.ms_post_process else 'none')
    if args.data and not args.synthetic and not args.model_dir and not args.allow_seeded_weights:
        raise SystemExit('dataset mode evaluates a trained model: give --model <checkpoint.pth.tar> (the reference torch.loads it, Test_KITTI.py:120-121)')
    data = torch.load(args.model_dir, map_location='cpu') if args.model_dir else {'state_dict': synthetic.seeded_state_dict(args.m_model[-1], args.no_levels)}
    m_name = data.get('m_model', args.m_model) if isinstance(data, dict) else args.m_model  # :122
    pan_model = models.__dict__[m_name](data, no_levels=args.no_levels, compute_dtype=dtype).to(dev).eval()
    n_params = utils.get_n_params(pan_model)

    if args.data and not args.synthetic:
        from fal_net_amd import datasets as DS
        root = os.path.join(args.data, args.tdataName)
        triples = DS.kitti2015_pairs(root) if args.tdataName == 'Kitti2015' else DS.eigen_test_triples(args.test_list, root)
        if not triples:
            raise SystemExit('no test frame with ground truth found under {}'.format(root))
        loader = DS.make_loader(DS.StereoValDataset(root, triples), 1, args.workers, shuffle=False, drop_last=False)  # B = 1: KITTI mixes sizes (:113)
        save_path = args.save_path or (os.path.join('Test_Results', args.tdataName, os.path.basename(args.model_dir or 'seeded'))
                                       + ('fpp' if args.f_post_process else '') + ('mspp' if args.ms_post_process and not args.f_post_process else ''))
        os.makedirs(save_path, exist_ok=True)
        with open(os.path.join(save_path, 'settings.txt'), 'w') as f:  # :63-75
            f.write(''.join('%15s: %s\n' % (k, v) for k, v in vars(args).items()))
        print('=> {} test frames under {}; saving to {}'.format(len(triples), root, save_path))
        res = inference.evaluate(pan_model, loader, data_name=args.tdataName, max_disp=args.max_disp, min_disp=args.min_disp,
                                 rel_baseline=args.rel_baselne, post=post, use_median=args.median, print_freq=args.print_freq)
        with open(os.path.join(save_path, 'errors.txt'), 'w') as f:  # :277-280
            f.write('\nNumber of parameters {}\n'.format(n_params))
            f.write('\nEPE {}\n'.format(res['epe']))
            f.write('\nKitti metrics: \n{}\n'.format(res['kitti_table']))
        print('* EPE: {0}'.format(res['epe']))
        print(res['kitti_table'])
        print(json.dumps({'dataset': args.tdataName, 'frames': res['n'], 'dtype': args.dtype, 'post': post, 'epe': res['epe'], 'kitti': res['kitti'],
                          'sec_per_image': res['sec_per_image'], 'errors_txt': os.path.join(save_path, 'errors.txt')}))
        return

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
    print(json.dumps({'image': [args.height, args.width], 'dtype': args.dtype, 'post': post,
                      'sec_per_image_median': sorted(times)[len(times) // 2], 'disp_mean': float(disp.mean()), 'disp_max': float(disp.max())}))


if __name__ == '__main__':
    args = parser.parse_args()
    main()
