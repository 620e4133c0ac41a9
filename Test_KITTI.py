#!/usr/bin/env python3
"""Inference / evaluation entry point (reference: Test_KITTI.py) on the MI355X implementation.

Two modes:
  * dataset mode (`-d <root> -tn Kitti2015`, or `-tn Kitti_eigen_test_improved --test_list <file>`): the reference's evaluation
    loop (Test_KITTI.py:103-117 file-list dataset at batch size 1, :163-208 forward + flip / multi-scale post-processing,
    :255-271 per-image KITTI depth errors and EPE, :277-280 `errors.txt`) over full-size frames of mixed sizes.  Frames are decoded
    by loader workers (Pillow) and normalised on the GPU; the network, `ms_pp` resampling and flips are HIP kernels; the metric
    chain is host-side numpy exactly as in the reference (fal_net_amd.myUtils).
  * `--synthetic`: seeded image of `--height x --width` (native KITTI 375x1242 by default), timing only -- no dataset on the box.
The command line is the reference's (Test_KITTI.py:36-60): `-m` is the model NAME and the checkpoint is <-dt>/<-ts>/<-m><-dtl>
(:119-120; `--checkpoint <file>` names it directly).  Image / PLY dumping (:211-253) is I/O cosmetics and not provided: `-save*` parse
and are refused when true.  `--dtype f16` is the recommended 16-bit
inference type (depth abs_rel vs the f32 path 2e-3, bf16 1.7e-2, at the same speed)."""
import argparse
import json
import os
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "5")  # application-level choice, before HIP initialises (fal_net_amd/__init__.py)

def _flag(v):
    """The reference declares its switches as untyped options with a default (`-fpp True`, `-eval False`, Test_KITTI.py:41-60), so the
    value arrives as a string; there every non-empty string -- 'False' included -- is truthy.  Here the words mean what they say."""
    if isinstance(v, bool):
        return v
    if str(v).strip().lower() in ('1', 'true', 't', 'yes', 'y', 'on'):
        return True
    if str(v).strip().lower() in ('0', 'false', 'f', 'no', 'n', 'off', ''):
        return False
    raise argparse.ArgumentTypeError('expected True or False, got {!r}'.format(v))


def _switch(p, *names, default, help):
    """`-name` alone, `-name True` and `-name False` all parse (the reference form is the one with a value)."""
    p.add_argument(*names, nargs='?', const=True, default=default, type=_flag, metavar='BOOL', help=help)


# The reference's command line (Test_KITTI.py:36-60): same flag names, meanings and defaults, except that --data has no default
# (the reference's is the author's desktop) and the values are typed.
parser = argparse.ArgumentParser(description='Testing pan generation (FAL_net on MI355X)', formatter_class=argparse.ArgumentDefaultsHelpFormatter)
parser.add_argument('-d', '--data', metavar='DIR', default=None, help='path to dataset; frames are read from <data>/<tdataName>')
parser.add_argument('-tn', '--tdataName', metavar='Test Data Set Name', default='Kitti_eigen_test_improved', choices=['Kitti2015', 'Kitti_eigen_test_improved'])
parser.add_argument('-relbase', '--rel_baselne', type=float, default=1, help='Relative baseline of testing dataset')
parser.add_argument('-mdisp', '--max_disp', type=float, default=300, help='of the training patch W')
parser.add_argument('-mindisp', '--min_disp', type=float, default=2, help='of the training patch W')
parser.add_argument('-b', '--batch_size', metavar='Batch Size', type=int, default=1, help='accepted and set to 1 as the reference does (:112: KITTI mixes image sizes)')
_switch(parser, '-eval', '--evaluate', default=True, help='compute KITTI errors (+ EPE on Kitti2015) and write errors.txt')
_switch(parser, '-save', '--save', default=False, help='disparity PNG dumps (out of scope here: refused when true)')
_switch(parser, '-save_pc', '--save_pc', default=False, help='PLY point clouds (out of scope: refused when true)')
_switch(parser, '-save_pan', '--save_pan', default=False, help='synthesised views (out of scope: refused when true; crashes in the reference, :190)')
_switch(parser, '-save_input', '--save_input', default=False, help='input image dumps (out of scope: refused when true)')
parser.add_argument('-w', '--workers', metavar='Workers', type=int, default=4)
parser.add_argument('--sparse', default=False, action='store_true', help='Depth GT is sparse, automatically selected when choosing a KITTI dataset')
parser.add_argument('--print-freq', '-p', default=10, type=int, metavar='N', help='print frequency')
parser.add_argument('-gpu_no', '--gpu_no', default=None, help='GPU ID: exported as HIP_VISIBLE_DEVICES before HIP initialises (the reference sets '
                    "CUDA_VISIBLE_DEVICES, default '1'; here the default keeps the environment's device)")
parser.add_argument('-dt', '--dataset', help='Dataset and training stage directory', default='Kitti_stage2')
parser.add_argument('-ts', '--time_stamp', help='Model timestamp', default='10-18-15_42')
parser.add_argument('-m', '--model', help='Model name (FAL_netA / FAL_netB / FAL_netC); the checkpoint\'s own m_model entry wins, as in the reference (:122)', default='FAL_netB')
parser.add_argument('-no_levels', '--no_levels', type=int, default=49, help='Number of quantization levels in MED')
parser.add_argument('-dtl', '--details', help='details: the checkpoint is <dataset>/<time_stamp>/<model><details> (:119-120)', default=',e20es,b4,lr5e-05/checkpoint.pth.tar')
_switch(parser, '-fpp', '--f_post_process', default=False, help='Post-processing with flipped input')
_switch(parser, '-mspp', '--ms_post_process', default=True, help='Post-processing with multi-scale input')
_switch(parser, '-median', '--median', default=False, help='use median scaling (not needed when training from stereo)')
# additions of this implementation (no reference counterpart)
parser.add_argument('--checkpoint', default=None, help='checkpoint file given directly instead of composing it from -dt / -ts / -m / -dtl')
parser.add_argument('--test_list', default=os.path.join('Datasets', 'kitti_eigen_test_improved.txt'),
                    help="Eigen split: one 'left right [gt]' line per frame, paths relative to <data>/<tdataName> (the reference opens "
                         "Datasets/kitti_eigen_test_improved.txt relative to the working directory)")
parser.add_argument('--save-path', default=None, help='where errors.txt / settings.txt go (default Test_Results/<tdataName>/<model>/<time_stamp>[fpp][mspp], :81-85)')
parser.add_argument('--synthetic', action='store_true', help='timing on a seeded image (default when no --data is given)')
parser.add_argument('--allow-seeded-weights', action='store_true', help='dataset mode without a checkpoint: evaluate SEEDED (untrained) weights (tests)')
parser.add_argument('--height', type=int, default=375)
parser.add_argument('--width', type=int, default=1242)
parser.add_argument('--iters', type=int, default=10)
parser.add_argument('--dtype', default='f16', choices=['f16', 'bf16', 'f32'])


def checkpoint_path(a):
    """Test_KITTI.py:119-120: os.path.join(args.dataset, args.time_stamp, args.model + args.details); --checkpoint overrides."""
    return a.checkpoint or os.path.join(a.dataset, a.time_stamp, a.model + a.details)


def refuse_out_of_scope(a):
    on = [n for n in ('save', 'save_pc', 'save_pan', 'save_input') if getattr(a, n)]
    if on:
        raise SystemExit('{}: image / point-cloud dumps (reference Test_KITTI.py:211-253) are out of scope of this implementation; '
                         'run without them to evaluate'.format(', '.join('-' + n for n in on)))


def main():
    import torch
    from fal_net_amd import inference, synthetic
    from fal_net_amd import myUtils as utils
    import models
    dev = torch.device('cuda', 0)
    dtype = {'f16': torch.float16, 'bf16': torch.bfloat16, 'f32': torch.float32}[args.dtype]
    post = 'flip' if args.f_post_process else ('ms_pp' if args.ms_post_process else 'none')
    refuse_out_of_scope(args)
    dataset_mode = bool(args.data) and not args.synthetic
    model_dir = checkpoint_path(args)  # :119-120
    have_ckpt = os.path.isfile(model_dir)
    if args.checkpoint and not have_ckpt:
        raise SystemExit('--checkpoint {!r} does not exist'.format(args.checkpoint))
    if dataset_mode and not have_ckpt and not args.allow_seeded_weights:
        raise SystemExit('dataset mode evaluates a trained model, but the checkpoint {!r} does not exist: it is composed as '
                         '<-dt>/<-ts>/<-m><-dtl> (reference Test_KITTI.py:119-121), or give --checkpoint <file.pth.tar>'.format(model_dir))
    if have_ckpt:
        data = torch.load(model_dir, map_location='cpu')
        print("=> using pre-trained model for pan '{}'".format(data.get('m_model', args.model)))
    else:
        data = {'state_dict': synthetic.seeded_state_dict(args.model[-1], args.no_levels)}
    m_name = data.get('m_model', args.model) if isinstance(data, dict) else args.model  # :122
    pan_model = models.__dict__[m_name](data, no_levels=args.no_levels, compute_dtype=dtype).to(dev).eval()
    n_params = utils.get_n_params(pan_model)

    if dataset_mode:
        from fal_net_amd import datasets as DS
        args.batch_size = 1  # kitty mixes image sizes! (:112)
        args.sparse = True  # disparities are sparse (from lidar) (:113)
        root = os.path.join(args.data, args.tdataName)
        triples = DS.kitti2015_pairs(root) if args.tdataName == 'Kitti2015' else DS.eigen_test_triples(args.test_list, root)
        if not triples:
            raise SystemExit('no test frame with ground truth found under {}'.format(root))
        loader = DS.make_loader(DS.StereoValDataset(root, triples), 1, args.workers, shuffle=False, drop_last=False)  # B = 1: KITTI mixes sizes (:113)
        save_path = args.save_path or (os.path.join('Test_Results', args.tdataName, args.model, args.time_stamp)  # :81-85
                                       + ('fpp' if args.f_post_process else '') + ('mspp' if args.ms_post_process else ''))
        os.makedirs(save_path, exist_ok=True)
        with open(os.path.join(save_path, 'settings.txt'), 'w') as f:  # :63-75
            f.write(''.join('%15s: %s\n' % (k, v) for k, v in vars(args).items()))
        print('=> {} test frames under {}; saving to {}'.format(len(triples), root, save_path))
        res = inference.evaluate(pan_model, loader, data_name=args.tdataName, max_disp=args.max_disp, min_disp=args.min_disp,
                                 rel_baseline=args.rel_baselne, post=post, use_median=args.median, print_freq=args.print_freq,
                                 with_metrics=args.evaluate)
        with open(os.path.join(save_path, 'errors.txt'), 'w') as f:  # :277-280
            f.write('\nNumber of parameters {}\n'.format(n_params))
            f.write('\nEPE {}\n'.format(res['epe']))
            f.write('\nKitti metrics: \n{}\n'.format(res['kitti_table']))
        if args.evaluate:  # :282-284
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
    if args.gpu_no is not None:  # :363, before anything touches the GPU
        os.environ['HIP_VISIBLE_DEVICES'] = str(args.gpu_no)
    main()
